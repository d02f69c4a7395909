#!/usr/bin/env python3
"""Batches per second with N launches in flight, straight through the C ABI of a given build of the library (any revision
that has ctcd_set_cu_sharing).  python tools/raw_inflight.py <lib.so> [inflight=3] [cu_sharing=1] [B=256]"""
import ctypes
import sys
import time

import torch

path = sys.argv[1]
nfl = int(sys.argv[2]) if len(sys.argv) > 2 else 3
share = int(sys.argv[3]) if len(sys.argv) > 3 else 1
B = int(sys.argv[4]) if len(sys.argv) > 4 else 256
T, V, K = 1000, 29, 100
lib = ctypes.CDLL(path)
lib.ctcd_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]
lib.ctcd_set_cu_sharing.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.ctcd_check_status.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.ctcd_beam_decode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_double] + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 6
g = torch.Generator(device="cpu").manual_seed(1234)
lp = torch.randn((B, T, V), generator=g).log_softmax(-1).cuda()
hs, outs, streams = [], [], []
for i in range(nfl):
    h = ctypes.c_void_p()
    assert lib.ctcd_create(ctypes.byref(h), 0) == 0
    lib.ctcd_set_cu_sharing(h, share)
    hs.append(h)
    tok = torch.empty((B, K, T), dtype=torch.int32, device="cuda")
    outs.append((tok, torch.empty_like(tok), torch.empty((B, K), dtype=torch.float32, device="cuda"), torch.empty((B, K), dtype=torch.int32, device="cuda")))
    streams.append(torch.cuda.Stream())


def run(n):
    for i in range(n):
        k = i % nfl
        tok, ts, sc, ln = outs[k]
        rc = lib.ctcd_beam_decode(hs[k], lp.data_ptr(), None, B, T, V, K, 4, 1.0, V, 0, 1, tok.data_ptr(), ts.data_ptr(), sc.data_ptr(), ln.data_ptr(), None,
                                  ctypes.c_void_p(streams[k].cuda_stream))
        assert rc == 0, rc
    torch.cuda.synchronize()


run(6)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); run(24); best = min(best, (time.perf_counter() - t0) / 24)
for h in hs:
    assert lib.ctcd_check_status(h, B) == 0
print("%s: %d in flight, cu_sharing %d, B %d: %.3f ms per batch, %.0f utt/s" % (path.split("/")[-1], nfl, share, B, best * 1e3, B / best), flush=True)
