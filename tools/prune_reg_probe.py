#!/usr/bin/env python3
"""The workgroup prune pass with the row kept in registers against the two-sweep form (ctcd_debug_set_prune_registers): kernel time by HIP
events and the pass's whole output compared.    python tools/prune_reg_probe.py [--V 10000 --B 64 --T 500 --top 40 --cp 0.99] [--probs]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctcdecode_amd
ap = argparse.ArgumentParser()
ap.add_argument("--V", type=int, default=10000); ap.add_argument("--B", type=int, default=64); ap.add_argument("--T", type=int, default=500)
ap.add_argument("--top", type=int, default=40); ap.add_argument("--cp", type=float, default=0.99); ap.add_argument("--probs", action="store_true")
ap.add_argument("--reps", type=int, default=6)
a = ap.parse_args()
g = torch.Generator(device="cpu").manual_seed(5)
x = torch.randn((a.B, a.T, a.V), generator=g).log_softmax(-1)
if a.probs:
    x = x.exp()
x = x.cuda()
res = {}
for reg in (True, False, True, False):
    dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(a.V)], cutoff_top_n=a.top, cutoff_prob=a.cp, beam_width=100, log_probs_input=not a.probs)
    dec.set_timing(True)
    dec.set_prune_registers(reg)
    ms = []
    for _ in range(a.reps):
        out = dec.decode_device(x, None)
        torch.cuda.synchronize()
        ms.append(dec.last_prune_ms())
    cnt, lab, val = (torch.from_numpy(t) for t in dec.last_prune_rows(a.B * a.T, min(a.top, a.V)))
    key = "registers" if reg else "two sweeps"
    keep = torch.arange(lab.shape[1])[None, :] < cnt[:, None]
    sig = (cnt.clone(), torch.where(keep, lab, torch.zeros_like(lab)), torch.where(keep, val, torch.zeros_like(val)).view(torch.int32), [t.cpu() for t in out])
    if key in res:
        assert all(torch.equal(p, q) for p, q in zip(res[key][1][:3], sig[:3]))
    res[key] = (min(ms[1:]), sig)
    print("%-10s V=%d rows=%d: prune kernel %.3f ms = %.2f TB/s" % (key, a.V, a.B * a.T, min(ms[1:]), a.B * a.T * a.V * 4 / min(ms[1:]) / 1e9))
r, t = res["registers"][1], res["two sweeps"][1]
same = all(torch.equal(p, q) for p, q in zip(r[:3], t[:3])) and all(torch.equal(p, q) for p, q in zip(r[3], t[3]))
print("outputs of the pass and of the decode equal:", same)
