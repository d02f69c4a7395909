#!/usr/bin/env python3
"""Randomised GPU-vs-oracle stress of the LM tier (not part of the pytest suite): word and character models, random alpha /
beta, beam widths, workgroup sizes, pruning, ragged lengths, probability input, streaming with random chunking.
Exits non-zero on the first mismatch."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--degenerate", action="store_true", help="whole frames of -inf, overflowing sums ... (tests/degenerate_util.py)")
    a = ap.parse_args()
    import numpy as np
    import torch

    import ctcdecode_amd
    import oracle_util as ou
    from test_lm import DATA, LABELS29

    models = [("abcd_words.arpa", ["_", "a", "b", "c", "d", "'", " "]), ("chars.arpa", ["_", "a", "b", "c", "d", "'", "é", " "]),
              ("test.arpa", LABELS29), ("abcd_words.arpa", ["a", "b", " ", "c", "d", "'", "_"])]
    rng = np.random.default_rng(a.seed)
    for it in range(a.n):
        arpa, labels = models[int(rng.integers(0, len(models)))]
        V = len(labels)
        blank = labels.index("_")
        K = int(rng.choice([1, 2, 5, 16, 50, 100, 128, 200]))
        T = int(rng.integers(1, 200))
        B = int(rng.integers(1, 5))
        alpha, beta = float(rng.choice([0.0, 0.3, 1.0, 2.5])), float(rng.choice([-1.0, 0.0, 0.5, 1.5]))
        quant = [None, None, 0.5, 1.0][int(rng.integers(0, 4))]
        top_n = int(rng.choice([40, 40, 5]))
        threads = int(rng.choice([128, 256, 512, 1024]))
        prob_in = it % 7 == 3
        lp = ou.synth_logprobs(B, T, V, 17000 + it, quant=quant, blank_id=blank)
        if " " in labels:
            lp[:, :, labels.index(" ")] += np.float32(rng.choice([0.0, 1.0, 2.0]))
        if a.degenerate and " " in labels and blank == 0:
            import degenerate_util as du
            meta, lp = du.make_case(rng, V=V, labels_space=labels.index(" "))
            B, T, K, prob_in = 2, lp.shape[1], min(K, 128), False
        x = np.exp(lp).astype(np.float32) if prob_in else lp
        sl = rng.integers(0, T + 5, size=B).astype(np.int32) if it % 3 == 0 else None
        path = os.path.join(DATA, arpa)
        sc = ou.Scorer(alpha, beta, path, labels, "restated")
        kw = dict(beam=K, blank_id=blank, cutoff_top_n=top_n, log_input=not prob_in)
        want = ou.decode(x, sl, scorer=sc, **kw)
        tag = "it=%d %s V=%d K=%d T=%d B=%d alpha=%g beta=%g q=%s top_n=%d threads=%d prob_in=%s" % (it, arpa, V, K, T, B, alpha, beta, quant, top_n, threads, prob_in)
        try:
            if it % 2 == 0 or prob_in:
                dec = ctcdecode_amd.CTCBeamDecoder(labels, model_path=path, alpha=alpha, beta=beta, cutoff_top_n=top_n, beam_width=K, blank_id=blank,
                                                   log_probs_input=not prob_in)
                dec.set_threads(threads)
                out, scs, ts, ln = dec.decode(torch.from_numpy(x), torch.from_numpy(sl) if sl is not None else None)
                got = dict(tokens=out.numpy(), timesteps=ts.numpy(), scores=scs.numpy(), lens=ln.numpy(), nres=want["nres"])
            else:
                dec = ctcdecode_amd.OnlineCTCBeamDecoder(labels, model_path=path, alpha=alpha, beta=beta, cutoff_top_n=top_n, beam_width=K, blank_id=blank,
                                                         log_probs_input=True)
                states = [ctcdecode_amd.DecoderState(dec) for _ in range(B)]
                lens = np.clip(sl if sl is not None else np.full((B,), T, np.int32), 0, T)
                cuts = sorted(set(int(v) for v in rng.integers(0, T + 1, size=int(rng.integers(0, 4)))))
                bounds = [0] + cuts + [T]
                xt = torch.from_numpy(x)
                for i in range(len(bounds) - 1):
                    lo, hi = bounds[i], bounds[i + 1]
                    chunk_lens = torch.from_numpy(np.clip(lens - lo, 0, hi - lo).astype(np.int32))
                    out, scs, ts, ln = dec.decode(xt[:, lo:hi], states, [i == len(bounds) - 2] * B, seq_lens=chunk_lens)
                got = dict(tokens=np.zeros((B, K, T), np.int32), timesteps=np.zeros((B, K, T), np.int32), scores=scs.numpy(), lens=ln.numpy(), nres=want["nres"])
                got["tokens"][:, : out.shape[1], : out.shape[2]] = out.numpy()
                got["timesteps"][:, : out.shape[1], : out.shape[2]] = ts.numpy()
            ou.assert_same(got, want, tag)
        except NotImplementedError:
            continue
        except AssertionError as e:
            print("MISMATCH", tag, str(e)[:200], flush=True)
            sys.exit(1)
    print("ok: %d LM configurations" % a.n)


if __name__ == "__main__":
    main()
