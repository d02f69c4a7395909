"""Loader for the committed reference fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py)."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    beam, top_n, blank, log_input = (int(v) for v in z["params"])
    seq_lens = z["seq_lens"] if z["seq_lens"].size else None
    args = dict(probs=z["probs"], seq_lens=seq_lens, beam=beam, cutoff_top_n=top_n, blank_id=blank,
                log_input=bool(log_input), cutoff_prob=float(z["cutoff_prob"]))
    want = dict(tokens=z["tokens"], timesteps=z["timesteps"], scores=z["scores"], lens=z["lens"], nres=z["nres"])
    return args, want


GOLDEN_LM_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_lm")
DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def lm_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_LM_DIR, "*.npz")))


def load_lm(name):
    """LM-tier fixtures (tests/golden/make_golden_lm.py): decode arguments, scorer arguments, the reference's outputs."""
    z = np.load(os.path.join(GOLDEN_LM_DIR, name + ".npz"))
    beam, top_n, blank, log_input = (int(v) for v in z["params"])
    seq_lens = z["seq_lens"] if z["seq_lens"].size else None
    args = dict(probs=z["probs"], seq_lens=seq_lens, beam=beam, cutoff_top_n=top_n, blank_id=blank, log_input=bool(log_input))
    lm = dict(alpha=float(z["alpha_beta"][0]), beta=float(z["alpha_beta"][1]), lm_path=os.path.join(DATA_DIR, str(z["arpa"])),
              labels=[str(x) for x in z["labels"]], meta=tuple(int(v) for v in z["meta"]))
    want = dict(tokens=z["tokens"], timesteps=z["timesteps"], scores=z["scores"], lens=z["lens"], nres=z["nres"])
    return args, lm, want
