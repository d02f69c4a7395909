#!/usr/bin/env python3
"""Timing of the other BASELINE.json configurations' per-GPU shapes and of a blank-dominated input (not bench lines;
bench.py stays on configs[1]).  Usage on the GPU box: python tools/bench_configs.py [--out file.json]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(torch, ctcdecode_amd, name, B, T, V, K, top_n=40, cutoff_prob=1.0, blank_bias=0.0, reps=3):
    torch.manual_seed(7)
    x = torch.randn((B, T, V))
    x[:, :, 0] += blank_bias
    lp = x.log_softmax(-1).cuda()
    dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], cutoff_top_n=top_n, cutoff_prob=cutoff_prob, beam_width=K, log_probs_input=True)
    dec.set_timing(True)
    res = dec.decode_device(lp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        res = dec.decode_device(lp, check=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    r = {"name": name, "B": B, "T": T, "V": V, "beam": K, "cutoff_top_n": top_n, "cutoff_prob": cutoff_prob, "blank_bias": blank_bias,
         "ms_per_batch": round(dt * 1e3, 3), "decode_kernel_ms": round(dec.last_kernel_ms(), 3), "utt_per_s": round(B / dt, 1),
         "us_per_frame": round(dec.last_kernel_ms() * 1e3 / T, 3), "mean_top_len": float(res[3][:, 0].float().mean()),
         "prune_flagged_rows": int(ctcdecode_amd._native.lib.ctcd_last_prune_flagged_rows(dec._handle)),
         "prune_host_rows": int(ctcdecode_amd._native.lib.ctcd_last_prune_host_rows(dec._handle))}
    print(json.dumps(r), flush=True)
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--only", default="", help="digits of the BASELINE.json configs to run (e.g. 23); default all")
    ap.add_argument("--reps", type=int, default=0)
    a = ap.parse_args()
    import torch

    import ctcdecode_amd

    out = []
    want = lambda i: not a.only or str(i) in a.only  # noqa: E731
    rp = lambda d: a.reps or d  # noqa: E731
    if want(0):
        out.append(run(torch, ctcdecode_amd, "configs[0] shape (log input)", 4, 100, 29, 10, reps=rp(3)))
    if want(1):
        out.append(run(torch, ctcdecode_amd, "configs[1]", 256, 1000, 29, 100, reps=rp(3)))
        out.append(run(torch, ctcdecode_amd, "configs[1] blank-dominated (+6 on the blank logit)", 256, 1000, 29, 100, blank_bias=6.0, reps=rp(3)))
        out.append(run(torch, ctcdecode_amd, "configs[1] blank +3", 256, 1000, 29, 100, blank_bias=3.0, reps=rp(3)))
    if want(2):
        out.append(run(torch, ctcdecode_amd, "configs[2] per-GPU shape (256 of 2048 utterances)", 256, 2000, 29, 500, reps=1))
    if want(3):
        out.append(run(torch, ctcdecode_amd, "configs[3]", 64, 500, 10000, 100, top_n=40, cutoff_prob=0.99, reps=rp(3)))
    if want(4):
        out.append(run(torch, ctcdecode_amd, "configs[4] shape without the LM (128 of 1024 utterances)", 128, 1500, 29, 100, reps=rp(3)))
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
