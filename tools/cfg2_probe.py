#!/usr/bin/env python3
"""configs[2] per-GPU shape (B=256, T=2000, V=29, beam 500): kernel time of the library in CTCDECODE_AMD_LIB (default: the product),
several launches, and a hash of the outputs (two builds that print the same hash decoded the same tensors).
    python tools/cfg2_probe.py [--reps 3] [--B 256 --T 2000 --beam 500]"""
import argparse, hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctcdecode_amd
ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=3); ap.add_argument("--B", type=int, default=256); ap.add_argument("--T", type=int, default=2000); ap.add_argument("--beam", type=int, default=500); ap.add_argument("--runtime-layout", action="store_true")
a = ap.parse_args()
g = torch.Generator(device="cpu").manual_seed(7)
lp = torch.randn((a.B, a.T, 29), generator=g).log_softmax(-1).cuda()
dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(29)], cutoff_top_n=29, beam_width=a.beam, log_probs_input=True)
dec.set_timing(True)
if a.runtime_layout:
    dec.set_fixed_layout(False)
ms = []
for r in range(a.reps + 1):
    out = dec.decode_device(lp)
    torch.cuda.synchronize()
    if r:
        ms.append(dec.last_kernel_ms())
h = hashlib.sha1()
for t in out:
    h.update(t.cpu().numpy().tobytes())
print(("run-time layout " if a.runtime_layout else "") + "%s B=%d T=%d beam=%d kernel ms min %.3f median %.3f  outputs %s" % (os.path.basename(os.environ.get("CTCDECODE_AMD_LIB", "product")), a.B, a.T, a.beam, min(ms), sorted(ms)[len(ms) // 2], h.hexdigest()[:12]))
