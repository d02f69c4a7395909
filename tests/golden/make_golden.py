#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (oracle/_ref/libctcref.so, built by oracle/Makefile from the
unmodified sources under /root/reference).  Runs only in the build container; the fixtures travel with the repo.

Each fixture stores the inputs (so nothing depends on an RNG implementation) and the reference's four output
tensors restricted to the region the reference defines (rest zero), plus n_results per item.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_util as ou  # noqa: E402

# tests/test_decode.py:16-30 of the reference (probabilities, blank = index 6, beam 20)
SEQ1 = [
    [0.06390443, 0.21124858, 0.27323887, 0.06870235, 0.0361254, 0.18184413, 0.16493624],
    [0.03309247, 0.22866108, 0.24390638, 0.09699597, 0.31895462, 0.0094893, 0.06890021],
    [0.218104, 0.19992557, 0.18245131, 0.08503348, 0.14903535, 0.08424043, 0.08120984],
    [0.12094152, 0.19162472, 0.01473646, 0.28045061, 0.24246305, 0.05206269, 0.09772094],
    [0.1333387, 0.00550838, 0.00301669, 0.21745861, 0.20803985, 0.41317442, 0.01946335],
    [0.16468227, 0.1980699, 0.1906545, 0.18963251, 0.19860937, 0.04377724, 0.01457421],
]
SEQ2 = [
    [0.08034842, 0.22671944, 0.05799633, 0.36814645, 0.11307441, 0.04468023, 0.10903471],
    [0.09742457, 0.12959763, 0.09435383, 0.21889204, 0.15113123, 0.10219457, 0.20640612],
    [0.45033529, 0.09091417, 0.15333208, 0.07939558, 0.08649316, 0.12298585, 0.01654384],
    [0.02512238, 0.22079203, 0.19664364, 0.11906379, 0.07816055, 0.22538587, 0.13483174],
    [0.17928453, 0.06065261, 0.41153005, 0.1172041, 0.11880313, 0.07113197, 0.04139363],
    [0.15882358, 0.1235788, 0.23376776, 0.20510435, 0.00279306, 0.05294827, 0.22298418],
]


def cases():
    fx = np.array([SEQ1, SEQ2], np.float32)
    yield "ref_fixtures_prob", dict(probs=fx, beam=20, blank_id=6, log_input=False, cutoff_top_n=40, cutoff_prob=1.0)
    yield "ref_fixtures_log", dict(probs=np.log(fx), beam=20, blank_id=6, log_input=True, cutoff_top_n=40, cutoff_prob=1.0)
    lp = ou.synth_logprobs(4, 100, 29, 0)
    yield "cfg1_prob_b4_t100_k10", dict(probs=np.exp(lp), beam=10, blank_id=0, log_input=False, cutoff_top_n=40, cutoff_prob=1.0)
    yield "rand_b3_t160_k32", dict(probs=ou.synth_logprobs(3, 160, 29, 1), beam=32, blank_id=0, log_input=True, cutoff_top_n=29, cutoff_prob=1.0)
    yield "ties_q05_b3_t200_k40", dict(probs=ou.synth_logprobs(3, 200, 29, 2, quant=0.5), beam=40, blank_id=0, log_input=True, cutoff_top_n=40, cutoff_prob=1.0)
    yield "ties_q1_b2_t120_k100", dict(probs=ou.synth_logprobs(2, 120, 12, 3, quant=1.0), beam=100, blank_id=3, log_input=True, cutoff_top_n=40, cutoff_prob=1.0)
    yield "blanky_b3_t200_k25", dict(probs=ou.synth_logprobs(3, 200, 29, 4, blank_bias=6.0), beam=25, blank_id=0, log_input=True, cutoff_top_n=40, cutoff_prob=1.0)
    yield "blanky_ties_b2_t150_k30", dict(probs=ou.synth_logprobs(2, 150, 9, 5, blank_bias=3.0, quant=0.25, blank_id=8), beam=30, blank_id=8, log_input=True, cutoff_top_n=40, cutoff_prob=1.0)
    yield "topn_v64_n8_b2_t80_k20", dict(probs=ou.synth_logprobs(2, 80, 64, 6), beam=20, blank_id=0, log_input=True, cutoff_top_n=8, cutoff_prob=1.0)
    yield "ragged_b5_t60_k16", dict(probs=ou.synth_logprobs(5, 60, 29, 7), seq_lens=np.array([60, 0, 1, 37, 99], np.int32), beam=16, blank_id=0, log_input=True, cutoff_top_n=40, cutoff_prob=1.0)
    yield "tiny_b2_t2_v3_k50", dict(probs=ou.synth_logprobs(2, 2, 3, 8), beam=50, blank_id=1, log_input=True, cutoff_top_n=40, cutoff_prob=1.0)
    # N4: the cumulative-probability cut where it really triggers, probability input with top_n, and the
    # BASELINE.json configs[3] setting (top_n 40, cutoff_prob 0.99) at a four-digit vocabulary
    yield "cutprob05_b2_t120_k24", dict(probs=ou.synth_logprobs(2, 120, 29, 10), beam=24, blank_id=0, log_input=True, cutoff_top_n=40, cutoff_prob=0.5)
    yield "cutprob06_n10_b2_t120_k24", dict(probs=ou.synth_logprobs(2, 120, 29, 11), beam=24, blank_id=0, log_input=True, cutoff_top_n=10, cutoff_prob=0.6)
    yield "probin_n6_v40_b2_t100_k16", dict(probs=np.exp(ou.synth_logprobs(2, 100, 40, 12)), beam=16, blank_id=0, log_input=False, cutoff_top_n=6, cutoff_prob=1.0)
    yield "probin_cut055_v40_b2_t100_k16", dict(probs=np.exp(ou.synth_logprobs(2, 100, 40, 13)), beam=16, blank_id=0, log_input=False, cutoff_top_n=40, cutoff_prob=0.55)
    yield "v1000_n40_cut099_b2_t60_k20", dict(probs=ou.synth_logprobs(2, 60, 1000, 14), beam=20, blank_id=0, log_input=True, cutoff_top_n=40, cutoff_prob=0.99)
    yield "topn_ties_v64_n8_b2_t80_k20", dict(probs=ou.synth_logprobs(2, 80, 64, 15, quant=0.5), beam=20, blank_id=0, log_input=True, cutoff_top_n=8, cutoff_prob=1.0)
    yield "long_b1_t600_k60", dict(probs=ou.synth_logprobs(1, 600, 29, 9), beam=60, blank_id=0, log_input=True, cutoff_top_n=40, cutoff_prob=1.0)


def main():
    assert ou.have_reference(), "build oracle/_ref first (make -C oracle ref)"
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    for name, c in cases():
        r = ou.decode(c["probs"], c.get("seq_lens"), beam=c["beam"], cutoff_prob=c["cutoff_prob"], cutoff_top_n=c["cutoff_top_n"],
                      blank_id=c["blank_id"], log_input=c["log_input"], which="reference")
        np.savez_compressed(os.path.join(out_dir, name + ".npz"),
                            probs=c["probs"].astype(np.float32), seq_lens=c.get("seq_lens", np.zeros((0,), np.int32)),
                            params=np.array([c["beam"], c["cutoff_top_n"], c["blank_id"], int(c["log_input"])], np.int64),
                            cutoff_prob=np.float64(c["cutoff_prob"]),
                            tokens=r["tokens"], timesteps=r["timesteps"], scores=r["scores"], lens=r["lens"], nres=r["nres"])
        print(name, "nres", r["nres"].tolist(), "top lens", r["lens"][:, 0].tolist())


if __name__ == "__main__":
    main()
