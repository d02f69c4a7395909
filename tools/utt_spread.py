#!/usr/bin/env python3
"""Per-utterance decode time of the product kernel: a launch lasts as long as its slowest utterance, so each of the first N
utterances of the headline batch is decoded as a batch of 256 copies of itself (every workgroup then does the same work).
    python tools/utt_spread.py [--n 24] [--lib path]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=24)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--beam", type=int, default=100)
    a = ap.parse_args()
    import numpy as np
    import torch

    import ctcdecode_amd

    g = torch.Generator().manual_seed(a.seed)
    lp = torch.randn((256, 1000, 29), generator=g).log_softmax(-1).cuda()
    dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(29)], cutoff_top_n=29, beam_width=a.beam, log_probs_input=True)
    dec.set_timing(True)
    ks = []
    for _ in range(3):
        dec.decode_device(lp, None, check=False)
        torch.cuda.synchronize()
        ks.append(dec.last_kernel_ms())
    print("whole batch (256 different utterances): kernel %.3f ms" % min(ks[1:]))
    res = []
    for u in range(a.n):
        one = lp[u:u + 1].expand(256, -1, -1).contiguous()
        t = []
        for _ in range(3):
            dec.decode_device(one, None, check=False)
            torch.cuda.synchronize()
            t.append(dec.last_kernel_ms())
        res.append(min(t[1:]))
    r = np.array(res)
    print("256 copies of one utterance, %d utterances: min %.3f  median %.3f  mean %.3f  max %.3f ms" % (a.n, r.min(), np.median(r), r.mean(), r.max()))
    print(" ".join("%.2f" % v for v in sorted(res)))


if __name__ == "__main__":
    main()
