#!/usr/bin/env python3
"""How often a frame takes the paths that are rare on random input (host build of the product core, one thread).

    python tools/beam_stats.py [--T 1000] [--lm tests/data/test.arpa]

Prints per-frame averages of the events beam_core.h counts (enum Event): entries with in-beam descendants (phase A1's
searches), entries whose parent is in the beam (phase B's second log-sum-exp + pool updates), entries below a dead interior
node, revival candidates and revivals, pool walks.  Random log-softmax rows barely touch these paths; a beam shaped by a
dictionary (the LM tier) -- or by peaky acoustic posteriors -- lives on them, which is why its frame is longer.
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
NAMES = ["frames", "candidates", "exact replays", "entries with in-beam descendants", "entries whose parent is in the beam",
         "pool updates of a label probability", "entries below a dead interior node", "revival candidates", "revived nodes that survive",
         "pool walks", "hops of those walks", "selects on the fast path", "... whose bucket holds a single key", "keys in the K-th key's bucket",
         "speculative select: settled the frame", "... too few hot keys", "... too many hot keys", "... handed back (ties, last frame, danger mode)", "hot keys",
         "histogram select: K-th key below the window", "... crowded bucket (more than 128 keys), another round", "... ended on a single key value", "... rounds of the slow path",
         "frames with several candidates on the K-th score", "... such candidates", "... more than 128 of them", "selects on the fast path whose crowded bucket was ranked from the wide list", "selects on the fast path listed from phase B's pre-list"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--beam", type=int, default=100)
    ap.add_argument("--lm", default="")
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--blank-bias", type=float, default=0.0)
    a = ap.parse_args()
    import numpy as np
    import torch

    import oracle_util as ou

    labels = ["_", "'", " "] + [chr(ord("a") + i) for i in range(26)]
    torch.manual_seed(a.seed)
    lg = torch.randn((1, a.T, 29))
    lg[:, :, 0] += a.blank_bias
    lp = lg.log_softmax(-1).numpy()
    lib = ctypes.CDLL(ou.build_core_host())
    cnt = (ctypes.c_longlong * 32)()
    lib.ctccore_event_counts(cnt, 1)
    if a.lm:
        ou.decode_core_host_lm(lp, 0.5, 1.0, a.lm, labels, beam=a.beam, threads=1)
    else:
        ou.decode_core_host(lp, beam=a.beam, threads=1)
    k = lib.ctccore_event_counts(cnt, 0)
    frames = max(int(cnt[0]), 1)
    print("%s, T=%d, beam %d: per frame" % ("scorer " + os.path.basename(a.lm) if a.lm else "no scorer", a.T, a.beam))
    for i in range(1, k):
        print("  %-42s %9.2f" % (NAMES[i], cnt[i] / frames))


if __name__ == "__main__":
    main()
