#!/usr/bin/env python3
"""Kernel time of a streaming chunk call against the chunk length (256 streams, beam 100): the cost of a launch beyond its frames.
    python tools/chunk_probe.py   (on the GPU box)"""
import sys, os, ctypes, statistics
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import torch, ctcdecode_amd
B, T, V, K = 256, 1000, 29, 100
g = torch.Generator(device="cpu").manual_seed(1234)
lp = torch.randn((B, T, V), generator=g).log_softmax(-1).cuda()
dec = ctcdecode_amd.OnlineCTCBeamDecoder([str(i) for i in range(V)], cutoff_top_n=V, beam_width=K, log_probs_input=True)
ctcdecode_amd._native.check(ctcdecode_amd._native.lib.ctcd_set_timing(dec._handle, 1))
for chunk in (1, 2, 5, 10, 25, 50, 100):
    states = [ctcdecode_amd.DecoderState(dec) for _ in range(B)]
    ks = []
    f0 = 0
    for i in range(12):
        c = lp[:, f0:f0 + chunk].contiguous(); f0 += chunk
        dec.decode(c, states, [False] * B)
        ms = ctypes.c_float(); ctcdecode_amd._native.lib.ctcd_last_kernel_ms(dec._handle, ctypes.byref(ms)); ks.append(ms.value)
    print("chunk %4d frames: kernel %.1f us (median of calls 3..12)" % (chunk, statistics.median(ks[2:]) * 1e3))
    del states
# a call that feeds nothing (chunk lengths 0): launch + restore + park alone
states = [ctcdecode_amd.DecoderState(dec) for _ in range(B)]
dec.decode(lp[:, :50].contiguous(), states, [False] * B)
zl = torch.zeros((B,), dtype=torch.int32)
ks = []
for i in range(10):
    dec.decode(lp[:, 50:51].contiguous(), states, [False] * B, seq_lens=zl)
    ms = ctypes.c_float(); ctcdecode_amd._native.lib.ctcd_last_kernel_ms(dec._handle, ctypes.byref(ms)); ks.append(ms.value)
print("chunk    0 frames: kernel %.1f us" % (statistics.median(ks[2:]) * 1e3))
