#!/usr/bin/env python3
"""Randomised GPU-vs-oracle stress (not part of the pytest suite): many small configurations, all workgroup sizes,
pruned and unpruned, ragged lengths, streaming with random chunking.  Exits non-zero on the first mismatch."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--degenerate", action="store_true", help="whole frames of -inf, overflowing sums ... (tests/degenerate_util.py)")
    a = ap.parse_args()
    import numpy as np
    import torch

    import ctcdecode_amd
    import degenerate_util as du
    import oracle_util as ou

    rng = np.random.default_rng(a.seed)
    stats = np.zeros(5, np.int64)
    for it in range(a.n):
        V = int(rng.choice([2, 3, 5, 9, 29, 29, 64, 200]))
        K = int(rng.choice([1, 2, 5, 16, 50, 100, 128, 300]))
        T = int(rng.integers(1, 220))
        B = int(rng.integers(1, 5))
        quant = [None, None, 0.5, 1.0, 0.25, 2.0][int(rng.integers(0, 6))]
        bias = float(rng.choice([0, 0, 3, 6, -2]))
        blank = int(rng.integers(0, V))
        top_n = int(rng.choice([40, 40, 40, max(1, V // 2), 3]))
        cutoff = float(rng.choice([1.0, 1.0, 1.0, 0.5, 0.99]))
        threads = int(rng.choice([64, 128, 256, 512, 1024]))
        if K * (min(V, top_n) + 2) > 60000:
            continue
        lp = ou.synth_logprobs(B, T, V, 7000 + it, quant=quant, blank_bias=bias, blank_id=blank)
        if a.degenerate:
            meta, lp = du.make_case(rng, V=V if V <= 29 else 29, T=max(T, 2) if T < 120 else 60)
            K, blank, B = min(K, 128), meta["blank"], 2
            V, T = lp.shape[2], lp.shape[1]
        sl = rng.integers(0, T + 5, size=B).astype(np.int32) if it % 3 == 0 else None
        kw = dict(beam=K, blank_id=blank, cutoff_top_n=top_n, cutoff_prob=cutoff)
        want = ou.decode(lp, sl, which="restated", want_stats=True, **kw)
        stats += want["stats"].sum(0)
        tag = "it=%d V=%d K=%d T=%d B=%d q=%s bias=%s blank=%d top_n=%d cutoff=%s threads=%d" % (it, V, K, T, B, quant, bias, blank, top_n, cutoff, threads)
        labels = [str(i) for i in range(V)]
        lds = ctcdecode_amd._native.lib.ctcd_workgroup_lds_bytes(K, V, top_n, cutoff)
        if K * 4 * (min(V, top_n) + 2) + 30 * 4 * K > 150000:  # beyond one workgroup's LDS even with the HBM-scratch layout
            continue
        if it % 2 == 0:
            dec = ctcdecode_amd.CTCBeamDecoder(labels, cutoff_top_n=top_n, cutoff_prob=cutoff, beam_width=K, blank_id=blank, log_probs_input=True)
            dec.set_threads(threads)
            if it % 4 == 2:
                dec.set_fixed_layout(False)
            out, sc, ts, ln = dec.decode(torch.from_numpy(lp), torch.from_numpy(sl) if sl is not None else None)
            got = dict(tokens=out.numpy(), timesteps=ts.numpy(), scores=sc.numpy(), lens=ln.numpy(), nres=want["nres"])
        else:  # streaming with random chunk boundaries
            dec = ctcdecode_amd.OnlineCTCBeamDecoder(labels, cutoff_top_n=top_n, cutoff_prob=cutoff, beam_width=K, blank_id=blank, log_probs_input=True)
            states = [ctcdecode_amd.DecoderState(dec) for _ in range(B)]
            lens = sl if sl is not None else np.full((B,), T, np.int32)
            lens = np.clip(lens, 0, T)
            cuts = sorted(set(int(x) for x in rng.integers(0, T + 1, size=int(rng.integers(0, 4)))))
            bounds = [0] + cuts + [T]
            x = torch.from_numpy(lp)
            for i in range(len(bounds) - 1):
                lo, hi = bounds[i], bounds[i + 1]
                chunk_lens = torch.from_numpy(np.clip(lens - lo, 0, hi - lo).astype(np.int32))
                out, sc, ts, ln = dec.decode(x[:, lo:hi], states, [i == len(bounds) - 2] * B, seq_lens=chunk_lens)
            got = dict(tokens=np.zeros((B, K, T), np.int32), timesteps=np.zeros((B, K, T), np.int32), scores=sc.numpy(), lens=ln.numpy(), nres=want["nres"])
            got["tokens"][:, : out.shape[1], : out.shape[2]] = out.numpy()
            got["timesteps"][:, : out.shape[1], : out.shape[2]] = ts.numpy()
        try:
            ou.assert_same(got, want, tag)
        except AssertionError as e:
            print("MISMATCH", tag, str(e)[:200], flush=True)
            sys.exit(1)
    print("ok: %d configurations, oracle stats (steps, tie splits, revivals, child hits, lpc updates) = %s" % (a.n, stats.tolist()))


if __name__ == "__main__":
    main()
