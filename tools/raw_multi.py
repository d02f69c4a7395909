#!/usr/bin/env python3
"""Kernel time of one shape through the C ABI of SEVERAL builds of the library in one process, launches interleaved (A B C A B
C ...) so that clock / box drift hits them alike; the outputs of every build must agree bit for bit with the first one's.
    python tools/raw_multi.py B T V beam reps lib1.so lib2.so ... [--lm tests/data/test.arpa] [--cu-sharing 1]
(--kind blank / peaky: other input distributions.  tools/build_variants.sh makes the builds; --lm times the LM tier -- labels blank, ', space, a..z, alpha 0.5, beta 1.0 -- and
needs builds that contain its kernels: CTC_QUICK_BUILD=2 or full builds; --cu-sharing 1 asks for the two-workgroups-per-CU
instantiations.)  The --lm / --cu-sharing forms were written without a GPU at the end of round 3: check them on first use."""
import ctypes
import statistics
import sys

import torch

args = sys.argv[1:]
lm_path, cu_sharing = None, None
if "--lm" in args:
    i = args.index("--lm"); lm_path = args[i + 1]; del args[i:i + 2]
if "--cu-sharing" in args:
    i = args.index("--cu-sharing"); cu_sharing = int(args[i + 1]); del args[i:i + 2]
threads = 0  # --threads N: ctcd_set_threads (the workgroup-size sweep's builds hold one size each: CTC_QUICK_BUILD=3)
if "--threads" in args:
    i = args.index("--threads"); threads = [int(v) for v in args[i + 1].split(",")]; del args[i:i + 2]
kind = "randn"  # --kind randn | blank (+6 on the blank logit) | peaky (one label +8 per frame, in runs of 5-15 frames)
if "--kind" in args:
    i = args.index("--kind"); kind = args[i + 1]; del args[i:i + 2]
B, T, V, K, reps = (int(a) for a in args[:5])
paths = args[5:]
g = torch.Generator(device="cpu").manual_seed(7)
lg = torch.randn((B, T, V), generator=g)
if kind == "blank":
    lg[:, :, 0] += 6.0
elif kind == "peaky":
    for b in range(B):
        t = 0
        while t < T:
            run = int(torch.randint(5, 16, (1,), generator=g)); c = int(torch.randint(0, V, (1,), generator=g))
            lg[b, t:t + run, c] += 8.0
            t += run
lp = lg.log_softmax(-1).cuda()
labels = ["_", "'", " "] + [chr(ord("a") + i) for i in range(26)]
I, D, P = ctypes.c_int, ctypes.c_double, ctypes.c_void_p
libs = []
for p in paths:
    lib = ctypes.CDLL(p)
    h = P()
    lib.ctcd_create.argtypes = [ctypes.POINTER(P), I]
    assert lib.ctcd_create(ctypes.byref(h), 0) == 0
    lib.ctcd_set_timing.argtypes = [P, I]
    lib.ctcd_set_timing(h, 1)
    if threads:  # one size for all builds, or one per build
        lib.ctcd_set_threads.argtypes = [P, I]
        assert lib.ctcd_set_threads(h, threads[len(libs)] if len(threads) > 1 else threads[0]) == 0
    if cu_sharing is not None:
        lib.ctcd_set_cu_sharing.argtypes = [P, I]
        assert lib.ctcd_set_cu_sharing(h, cu_sharing) == 0
    lib.ctcd_last_kernel_ms.argtypes = [P, ctypes.POINTER(ctypes.c_float)]
    lib.ctcd_beam_decode.argtypes = [P, P, P] + [I] * 5 + [D] + [I] * 3 + [P] * 6
    lib.ctcd_beam_decode_lm.argtypes = [P, P, P] + [I] * 5 + [D] + [I] * 3 + [P] + [P] * 6
    scorer = P()
    if lm_path:
        assert V == len(labels), "--lm uses the 29-label alphabet"
        lib.ctcd_scorer_create.argtypes = [ctypes.POINTER(P), D, D, ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p), I, I]
        arr = (ctypes.c_char_p * V)(*[s.encode() for s in labels])
        rc = lib.ctcd_scorer_create(ctypes.byref(scorer), 0.5, 1.0, lm_path.encode(), arr, V, 0)
        assert rc == 0, ("ctcd_scorer_create", rc)
    out = (torch.empty((B, K, T), dtype=torch.int32, device="cuda"), torch.empty((B, K, T), dtype=torch.int32, device="cuda"),
           torch.empty((B, K), dtype=torch.float32, device="cuda"), torch.empty((B, K), dtype=torch.int32, device="cuda"))
    libs.append((p.split("/")[-1], lib, h, out, [], scorer))
for r in range(reps + 1):
    for name, lib, h, out, times, scorer in libs:
        if lm_path:
            rc = lib.ctcd_beam_decode_lm(h, lp.data_ptr(), None, B, T, V, K, 4, 1.0, 40, 0, 1, scorer, out[0].data_ptr(), out[1].data_ptr(),
                                         out[2].data_ptr(), out[3].data_ptr(), None, None)
        else:
            rc = lib.ctcd_beam_decode(h, lp.data_ptr(), None, B, T, V, K, 4, 1.0, 40, 0, 1, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                      out[3].data_ptr(), None, None)
        assert rc == 0, (name, rc)
        torch.cuda.synchronize()
        ms = ctypes.c_float()
        lib.ctcd_last_kernel_ms(h, ctypes.byref(ms))
        if r:
            times.append(ms.value)
ref = libs[0][3]
for name, lib, h, out, times, scorer in libs:
    same = all(torch.equal(a, b) for a, b in zip(out, ref))
    print("%-28s min %.3f  median %.3f ms   outputs %s" % (name, min(times), statistics.median(times), "== first" if same else "DIFFER"))
