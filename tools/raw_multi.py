#!/usr/bin/env python3
"""Kernel time of one shape through the C ABI of SEVERAL builds of the library in one process, launches interleaved (A B C A B
C ...) so that clock / box drift hits them alike; the outputs of every build must agree bit for bit with the first one's.
    python tools/raw_multi.py B T V beam reps lib1.so lib2.so ...      (tools/build_variants.sh makes the builds)"""
import ctypes
import statistics
import sys

import torch

B, T, V, K, reps = (int(a) for a in sys.argv[1:6])
paths = sys.argv[6:]
g = torch.Generator(device="cpu").manual_seed(7)
lp = torch.randn((B, T, V), generator=g).log_softmax(-1).cuda()
libs = []
for p in paths:
    lib = ctypes.CDLL(p)
    h = ctypes.c_void_p()
    lib.ctcd_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]
    assert lib.ctcd_create(ctypes.byref(h), 0) == 0
    lib.ctcd_set_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.ctcd_set_timing(h, 1)
    lib.ctcd_last_kernel_ms.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
    lib.ctcd_beam_decode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_double] + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 6
    out = (torch.empty((B, K, T), dtype=torch.int32, device="cuda"), torch.empty((B, K, T), dtype=torch.int32, device="cuda"),
           torch.empty((B, K), dtype=torch.float32, device="cuda"), torch.empty((B, K), dtype=torch.int32, device="cuda"))
    libs.append((p.split("/")[-1], lib, h, out, []))
for r in range(reps + 1):
    for name, lib, h, out, times in libs:
        rc = lib.ctcd_beam_decode(h, lp.data_ptr(), None, B, T, V, K, 4, 1.0, 40, 0, 1, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                  out[3].data_ptr(), None, None)
        assert rc == 0, (name, rc)
        torch.cuda.synchronize()
        ms = ctypes.c_float()
        lib.ctcd_last_kernel_ms(h, ctypes.byref(ms))
        if r:
            times.append(ms.value)
ref = libs[0][3]
for name, lib, h, out, times in libs:
    same = all(torch.equal(a, b) for a, b in zip(out, ref))
    print("%-28s min %.3f  median %.3f ms   outputs %s" % (name, min(times), statistics.median(times), "== first" if same else "DIFFER"))
