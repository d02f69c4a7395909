#!/usr/bin/env python3
"""Workload for counter passes: the headline shape (B=256, T=1000, V=29, beam 100) on random rows and on blank-dominated
rows (+6 on the blank logit; checked launches, so that the decoder picks the chain-shaped-beam build <3,...> for them), a few
launches each; optionally the configs[4] shape with the LM scorer and the beam-500 shape.
    rocprofv3 --pmc ... -- python tools/pmc_kinds.py [--reps 3] [--kinds randn,blank,lm,wide]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import ctcdecode_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--kinds", default="randn,blank")
a = ap.parse_args()
dev = torch.device("cuda", 0)
labels = ["_", "'", " "] + [chr(ord("a") + i) for i in range(26)]
for kind in a.kinds.split(","):
    kw, B, T, K = {}, 256, 1000, 100
    if kind == "lm":
        kw, B, T = dict(model_path=os.path.join(ROOT, "tests", "data", "test.arpa"), alpha=0.5, beta=1.0), 128, 1500
    if kind == "wide":
        T, K = 2000, 500
    lp = bench.synth_rows(torch, B, T, 29, 7, kind if kind in ("randn", "blank", "peaky") else "randn").to(dev)
    dec = ctcdecode_amd.CTCBeamDecoder(labels, cutoff_top_n=29, beam_width=K, blank_id=0, log_probs_input=True, device=dev, **kw)
    dec.set_timing(True)
    for r in range(a.reps + 1):
        res = dec.decode_device(lp, None, check=True)
        torch.cuda.synchronize()
        print(kind, "launch", r, "kernel ms %.3f" % dec.last_kernel_ms(), flush=True)
    del dec, lp, res
