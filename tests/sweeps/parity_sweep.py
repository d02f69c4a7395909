#!/usr/bin/env python3
"""One-off large parity sweep on the GPU box: N utterances of the north-star shape (and a blank-dominated variant),
every one compared bit-exactly with the oracle (which uses the box's host cores).  Prints a JSON summary."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--beam", type=int, default=100)
    ap.add_argument("--out", default="")
    ap.add_argument("--head", default="", help="git revision of the tree under test (recorded in the output)")
    a = ap.parse_args()
    import numpy as np
    import torch

    import ctcdecode_amd
    import oracle_util as ou

    res = []
    for name, bias, quant in [("randn", 0.0, None), ("blank+4", 4.0, None), ("quantised 0.25", 0.0, 0.25)]:
        lp = ou.synth_logprobs(a.n, a.frames, 29, 4242, blank_bias=bias, quant=quant)
        dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(29)], beam_width=a.beam, log_probs_input=True)
        t0 = time.time()
        out, sc, ts, ln = dec.decode(torch.from_numpy(lp))
        tg = time.time() - t0
        got = dict(tokens=out.numpy(), timesteps=ts.numpy(), scores=sc.numpy(), lens=ln.numpy())
        t0 = time.time()
        want = ou.decode(lp, beam=a.beam, which="reference" if ou.have_reference() else "restated", want_stats=False)
        tc = time.time() - t0
        got["nres"] = want["nres"]
        bad = 0
        for b in range(a.n):
            try:
                ou.assert_same({k: v[b:b + 1] for k, v in got.items()}, {k: v[b:b + 1] for k, v in want.items()})
            except AssertionError:
                bad += 1
        r = {"input": name, "utterances": a.n, "frames": a.frames, "beam": a.beam, "mismatching_utterances": bad,
             "checker": "reference" if ou.have_reference() else "restated", "git_head": a.head, "gpu_decode_s_incl_pcie": round(tg, 3), "cpu_s": round(tc, 1)}
        print(json.dumps(r), flush=True)
        res.append(r)
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)
    sys.exit(1 if any(r["mismatching_utterances"] for r in res) else 0)


if __name__ == "__main__":
    main()
