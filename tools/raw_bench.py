#!/usr/bin/env python3
"""Kernel time of one shape straight through the C ABI of a given build of the library (any revision: only ctcd_create,
ctcd_beam_decode, ctcd_set_timing, ctcd_last_kernel_ms are used).  python tools/raw_bench.py <lib.so> B T V beam [reps]"""
import ctypes
import sys

import torch

path, B, T, V, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
lib = ctypes.CDLL(path)
h = ctypes.c_void_p()
lib.ctcd_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]
assert lib.ctcd_create(ctypes.byref(h), 0) == 0
lib.ctcd_set_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.ctcd_set_timing(h, 1)
lib.ctcd_last_kernel_ms.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
lib.ctcd_beam_decode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_double] + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 6
g = torch.Generator(device="cpu").manual_seed(7)
lp = torch.randn((B, T, V), generator=g).log_softmax(-1).cuda()
tok = torch.empty((B, K, T), dtype=torch.int32, device="cuda"); ts = torch.empty_like(tok)
sc = torch.empty((B, K), dtype=torch.float32, device="cuda"); ln = torch.empty((B, K), dtype=torch.int32, device="cuda")
best = 1e9
for _ in range(reps + 1):
    rc = lib.ctcd_beam_decode(h, lp.data_ptr(), None, B, T, V, K, 4, 1.0, 40, 0, 1, tok.data_ptr(), ts.data_ptr(), sc.data_ptr(), ln.data_ptr(), None, None)
    assert rc == 0, rc
    torch.cuda.synchronize()
    ms = ctypes.c_float()
    lib.ctcd_last_kernel_ms(h, ctypes.byref(ms))
    best = min(best, ms.value)
print("%s B%d T%d V%d K%d: kernel %.3f ms  (checksum %d)" % (path.split("/")[-1], B, T, V, K, best, int(tok.long().sum().item() + ln.long().sum().item())))
