#!/usr/bin/env python3
"""configs[3] (B=64, T=500, V=10000, beam 100, cutoff_top_n 40, cutoff_prob 0.99): decode-kernel and prune-pass time of the
compile-time layout of the pruned class (kernel <0,0,2,true,1024>) against the run-time layout it replaces
(ctcd_debug_set_fixed_layout(0): kernel <0,0,0,true,0>), outputs compared bit for bit; then other shapes of the class.
    python tools/cfg3_probe.py [--reps 8] [--logits]        (CTCDECODE_AMD_LIB=... picks another build of the library)"""
import argparse
import json
import statistics
import sys, os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import ctcdecode_amd


def run(B, T, V, K, top_n, cp, reps, fixed, logits=False, seed=3):
    g = torch.Generator(device="cpu").manual_seed(seed)
    lg = torch.randn((B, T, V), generator=g)
    x = (lg if logits else lg.log_softmax(-1)).cuda()
    dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], cutoff_top_n=top_n, cutoff_prob=cp, beam_width=K, log_probs_input=True,
                                       **({"logits_input": True} if logits else {}))
    dec.set_timing(True)
    dec.set_fixed_layout(fixed)
    ks, ps = [], []
    out = None
    for r in range(reps + 1):
        out = dec.decode_device(x)
        torch.cuda.synchronize()
        if r:
            ks.append(dec.last_kernel_ms()); ps.append(dec.last_prune_ms())
    return statistics.median(ks), statistics.median(ps), [o.clone() for o in out]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--logits", action="store_true")
    a = ap.parse_args()
    res = {}
    for name, (B, T, V, K, top_n, cp) in {"configs[3]": (64, 500, 10000, 100, 40, 0.99), "B=256 V=2000 top_n 40": (256, 500, 2000, 100, 40, 1.0),
                                          "V=64 top_n 40 (class 2)": (256, 1000, 64, 100, 40, 1.0), "V=29 top_n 20 (class 1, pruned)": (256, 1000, 29, 100, 20, 1.0)}.items():
        kf, pf, of = run(B, T, V, K, top_n, cp, a.reps, True, a.logits)
        kr, pr, orr = run(B, T, V, K, top_n, cp, a.reps, False, a.logits)
        same = all(torch.equal(x, y) for x, y in zip(of, orr))
        res[name] = dict(frames=B * T, compile_time_layout_kernel_ms=round(kf, 3), run_time_layout_kernel_ms=round(kr, 3), prune_ms=round(pf, 3),
                         us_per_frame=round(1e3 * kf / T, 3), us_per_frame_run_time_layout=round(1e3 * kr / T, 3), outputs_equal=same)
        print(name, res[name], flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
