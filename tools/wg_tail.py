#!/usr/bin/env python3
"""What makes the slowest workgroups of a launch slow: per-phase time (instrumented build, tools/phase_profile.py) of the slowest
utterances against the batch mean.  A launch lasts as long as its slowest utterance.
    python tools/wg_tail.py [--beam 100 --seed 1234]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phase_profile import NAMES  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--beam", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1234)
    a = ap.parse_args()
    import numpy as np
    import torch

    import ctcdecode_amd
    from ctcdecode_amd import _native as n

    g = torch.Generator().manual_seed(a.seed)
    lp = torch.randn((a.batch, a.frames, 29), generator=g).log_softmax(-1).cuda()
    dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(29)], cutoff_top_n=29, beam_width=a.beam, log_probs_input=True)
    dec.set_timing(True)
    n.check(n.lib.ctcd_debug_set_profile(dec._handle, 1))
    for _ in range(2):
        dec.decode_device(lp)
    torch.cuda.synchronize()
    ms = dec.last_kernel_ms()
    prof = np.zeros((a.batch, 16), np.int64)
    n.check(n.lib.ctcd_debug_get_profile(dec._handle, prof.ctypes.data, a.batch))
    tot = prof.sum(1).astype(np.float64)
    scale = ms * 1e3 / tot.max()
    order = np.argsort(-tot)
    print("kernel %.3f ms (instrumented); workgroup totals: max %.0f us, p90 %.0f, mean %.0f, min %.0f" % (ms, tot.max() * scale, np.percentile(tot, 90) * scale, tot.mean() * scale, tot.min() * scale))
    mean = prof.mean(0) * scale
    print("%-60s %8s | slowest five workgroups" % ("phase (us per utterance)", "mean"))
    for i in np.argsort(-mean):
        print("%-60s %8.1f | %s" % (NAMES[i][:60], mean[i], " ".join("%8.1f" % (prof[w, i] * scale) for w in order[:5])))


if __name__ == "__main__":
    main()
