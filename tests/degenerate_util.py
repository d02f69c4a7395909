"""Degenerate inputs for the parity tests: whole frames of -inf, values near -FLT_MAX whose sums overflow, -inf tail
padding without seq_lens, hard-masked labels.  On such inputs log_sum_exp (decoder_utils.h:47-54) depends on the ORDER of
its arguments, i.e. on the permutation std::nth_element / std::sort leave in the reference's `prefixes` array
(ctc_beam_search_decoder.cpp:75-76,150-154): the product's "danger mode" (beam_core.h enter_danger) must reproduce it."""
import numpy as np

import oracle_util as ou

KINDS = ("inf_frames", "huge_frames", "inf_tail", "inf_frames_one_label", "masked_labels", "huge_values_and_inf_frames")


def make_case(rng, V=None, T=None, labels_space=None):
    """-> (meta, lp[2,T,V]).  labels_space: index of the space label to favour (LM cases)."""
    V = V or int(rng.choice([2, 3, 4, 5, 9, 29]))
    K = int(rng.choice([1, 2, 3, 4, 5, 8, 16, 50, 100]))
    T = T or int(rng.integers(2, 60))
    quant = [None, 0.5, 1.0, 2.0][int(rng.integers(0, 4))]
    blank = 0 if labels_space is not None else int(rng.integers(0, V))
    seed = int(rng.integers(0, 1 << 30))
    lp = ou.synth_logprobs(2, T, V, seed, quant=quant, blank_id=blank)
    if labels_space is not None:
        lp[:, :, labels_space] += np.float32(rng.choice([0.0, 1.0, 2.0]))
    kind = int(rng.integers(0, len(KINDS)))
    frames = rng.integers(0, T, size=int(rng.choice([1, 1, 3, 4])))
    if kind == 0:
        lp[:, frames, :] = -np.inf
    elif kind == 1:
        lp[:, frames, :] = -3.0e38
    elif kind == 2:
        lp[:, int(rng.integers(1, T)):, :] = -np.inf
    elif kind == 3:
        lp[:, frames, :] = -np.inf
        lp[:, frames[0], int(rng.integers(0, V))] = -1.0
    elif kind == 4:
        lp[rng.random((2, T, V)) < 0.3] = -np.inf
    else:
        lp[rng.random((2, T, V)) < 0.2] = np.float32(rng.choice([-3.0e38, -1e31, -1e25, -3.4028235e38]))
        lp[:, frames, :] = -np.inf
    return dict(V=V, K=K, T=T, blank=blank, seed=seed, kind=KINDS[kind]), lp
