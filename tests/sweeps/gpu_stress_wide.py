#!/usr/bin/env python3
"""Randomised GPU-vs-oracle stress of the wide-beam layouts (not part of the pytest suite): beams of 130 .. 1000 over 3 .. 40 labels, the
compile-time wide layout and the run-time ones, ties (quantised rows), ragged lengths, pruning, streams with random chunking.
Exits non-zero on the first mismatch.    python tests/sweeps/gpu_stress_wide.py [--n 150] [--seed 0]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=150)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    import numpy as np
    import torch

    import ctcdecode_amd
    import oracle_util as ou

    rng = np.random.default_rng(a.seed)
    done = 0
    for it in range(a.n):
        V = int(rng.choice([3, 5, 12, 29, 29, 29, 30, 40]))
        K = int(rng.choice([130, 200, 257, 300, 400, 500, 500, 640, 800, 1000]))
        T = int(rng.integers(8, 110))
        B = int(rng.integers(1, 4))
        quant = [None, None, 0.5, 0.25, 1.0][int(rng.integers(0, 5))]
        bias = float(rng.choice([0, 0, 3, 6]))
        top_n = int(rng.choice([40, 40, 40, max(1, V // 2)]))
        cutoff = float(rng.choice([1.0, 1.0, 0.9]))
        if K * (min(V, top_n) + 2) > 34000:
            continue
        lp = ou.synth_logprobs(B, T, V, 8800 + 31 * a.seed + it, quant=quant, blank_bias=bias)
        sl = rng.integers(0, T + 3, size=B).astype(np.int32) if it % 3 == 0 else None
        kw = dict(beam=K, cutoff_top_n=top_n, cutoff_prob=cutoff)
        want = ou.decode(lp, sl, which="restated", **kw)
        tag = "it=%d V=%d K=%d T=%d B=%d q=%s bias=%s top_n=%d cutoff=%s" % (it, V, K, T, B, quant, bias, top_n, cutoff)
        labels = [str(i) for i in range(V)]
        try:
            if it % 2 == 0:
                dec = ctcdecode_amd.CTCBeamDecoder(labels, cutoff_top_n=top_n, cutoff_prob=cutoff, beam_width=K, log_probs_input=True)
                if it % 4 == 2:
                    dec.set_fixed_layout(False)
                out, sc, ts, ln = dec.decode(torch.from_numpy(lp), torch.from_numpy(sl) if sl is not None else None)
                got = dict(tokens=out.numpy(), timesteps=ts.numpy(), scores=sc.numpy(), lens=ln.numpy(), nres=want["nres"])
            else:
                dec = ctcdecode_amd.OnlineCTCBeamDecoder(labels, cutoff_top_n=top_n, cutoff_prob=cutoff, beam_width=K, log_probs_input=True)
                states = [ctcdecode_amd.DecoderState(dec) for _ in range(B)]
                cuts = sorted(set(int(v) for v in rng.integers(0, T + 1, size=int(rng.integers(0, 4)))))
                bounds = [0] + cuts + [T]
                x = torch.from_numpy(lp)
                for i in range(len(bounds) - 1):
                    f0, f1 = bounds[i], bounds[i + 1]
                    chunk = x[:, f0:f1]
                    ends = [i == len(bounds) - 2] * B
                    out, sc, ts, ln = dec.decode(chunk, states, ends)
                if sl is not None:
                    continue  # (streams take whole chunks: ragged lengths are the offline form's business)
                L = out.shape[2]
                got = dict(tokens=np.zeros((B, K, T), np.int32), timesteps=np.zeros((B, K, T), np.int32), scores=sc.numpy(), lens=ln.numpy(), nres=want["nres"])
                got["tokens"][:, : out.shape[1], :L] = out.numpy()
                got["timesteps"][:, : out.shape[1], :L] = ts.numpy()
        except NotImplementedError:
            continue
        ou.assert_same(got, want, tag)
        done += 1
    print("ok: %d wide-beam configurations" % done)


if __name__ == "__main__":
    main()
