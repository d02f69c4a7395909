#!/usr/bin/env python3
"""Generate tests/golden_lm/*.npz: the LM-tier cases of tests/test_lm.py decoded by oracle/_ref (= the reference's own
scorer.cpp and decoder LM hooks, compiled unmodified over the kenlm / OpenFST stand-ins of oracle/shim).  Runs only in the
build container; the fixtures travel with the repo and pin the restated oracle and the HIP path where oracle/_ref is absent."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu  # noqa: E402
import oracle_util as ou  # noqa: E402
import test_lm as t  # noqa: E402


def main():
    assert ou.have_reference(), "build oracle/_ref first (make -C oracle ref)"
    out_dir = os.path.join(ROOT, "tests", "golden_lm")
    os.makedirs(out_dir, exist_ok=True)
    cases = list(t.LM_CASES)
    args, _ = gu.load("ref_fixtures_prob")  # tests/test_decode.py:55-64 of the reference: "a a"
    for c in cases + [dict(name="reference_a_a", arpa="test.arpa", labels=t.VOCAB7, alpha=0.0, beta=0.0, fixed=args)]:
        if "fixed" in c:
            a = c["fixed"]
            x, kw = a["probs"], dict(seq_lens=None, beam=a["beam"], cutoff_top_n=a["cutoff_top_n"], blank_id=a["blank_id"], log_input=a["log_input"])
        else:
            x, kw = t.lm_case_inputs(c)
        sc = ou.Scorer(c["alpha"], c["beta"], os.path.join(t.DATA, c["arpa"]), c["labels"], "reference")
        r = ou.decode(x, scorer=sc, which="reference", **kw)
        np.savez_compressed(os.path.join(out_dir, c["name"] + ".npz"), probs=x.astype(np.float32),
                            seq_lens=kw["seq_lens"] if kw["seq_lens"] is not None else np.zeros((0,), np.int32),
                            params=np.array([kw["beam"], kw["cutoff_top_n"], kw["blank_id"], int(kw["log_input"])], np.int64),
                            alpha_beta=np.array([c["alpha"], c["beta"]], np.float64), arpa=np.array(c["arpa"]), labels=np.array(c["labels"]),
                            meta=np.array([int(sc.is_character_based()), sc.max_order(), sc.dict_size()], np.int64),
                            tokens=r["tokens"], timesteps=r["timesteps"], scores=r["scores"], lens=r["lens"], nres=r["nres"])
        print(c["name"], "nres", r["nres"].tolist(), "top lens", r["lens"][:, 0].tolist())


if __name__ == "__main__":
    main()
