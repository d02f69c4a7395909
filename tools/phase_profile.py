#!/usr/bin/env python3
"""Per-phase time of the decode kernel (instrumented build), averaged over the utterances of one batch.
Usage on the GPU box: python tools/phase_profile.py [--batch 256 --frames 1000 --vocab 29 --beam 100 --threads 512]"""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

NAMES = ["A per-entry LCP scans, slot offsets, existing children", "B1 beam entries (thread 0's own part of B)", "B wait for the new-children waves", "D ordered compaction of survivors", "finish: final sort",
         "C3 select: rank in bucket", "D' exact nth_element replay", "E emit next beam (work)", "E state update", "E end-of-frame fence (global writes)",
         "frame load", "finish: copy shared labels", "loop tail", "C1 select: find bucket", "C2 select: gather bucket", "finish: read back unshared labels"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--vocab", type=int, default=29)
    ap.add_argument("--beam", type=int, default=100)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--blank-bias", type=float, default=0.0)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import numpy as np
    import torch

    import ctcdecode_amd
    from ctcdecode_amd import _native as n

    torch.manual_seed(1234)
    x = torch.randn((a.batch, a.frames, a.vocab))
    x[:, :, 0] += a.blank_bias
    lp = x.log_softmax(-1).cuda()
    dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(a.vocab)], cutoff_top_n=a.vocab, beam_width=a.beam, log_probs_input=True)
    if a.threads:
        dec.set_threads(a.threads)
    dec.set_timing(True)
    n.check(n.lib.ctcd_debug_set_profile(dec._handle, 1))
    for _ in range(2):
        dec.decode_device(lp)
    torch.cuda.synchronize()
    ms = dec.last_kernel_ms()
    prof = np.zeros((a.batch, 16), np.int64)
    n.check(n.lib.ctcd_debug_get_profile(dec._handle, prof.ctypes.data, a.batch))
    tot = prof.sum(1).astype(np.float64)
    scale = ms * 1e3 / tot.max()  # us per tick, calibrated on the slowest workgroup == kernel duration
    mean = prof.mean(0) * scale
    res = {"kernel_ms_instrumented": ms, "us_per_frame": ms * 1e3 / max(a.frames, 1), "tick_us": scale,
           "phases_us_per_utterance": {NAMES[i]: round(float(mean[i]), 1) for i in range(len(NAMES))},
           "phases_percent": {NAMES[i]: round(float(100 * mean[i] / mean.sum()), 2) for i in range(len(NAMES))},
           "config": vars(a)}
    print(json.dumps(res, indent=1))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
