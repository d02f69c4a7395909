"""Pins the CPU restatement (oracle/ctc_oracle.cpp) to the reference.

(a) the reference's own golden strings (tests/test_decode.py:31-32 of the reference),
(b) the survey's known-answer table,
(c) the committed fixtures produced by the real reference build (tests/golden, tests/golden/make_golden.py),
(d) live differential runs against oracle/_ref when it is present (build container and GPU box).
"""
import numpy as np
import pytest

import golden_util as gu
import oracle_util as ou

VOCAB = ["'", " ", "a", "b", "c", "d", "_"]


def _string(r, b, p):
    return "".join(VOCAB[x] for x in r["tokens"][b, p, : r["lens"][b, p]])


def test_reference_golden_strings():
    args, _ = gu.load("ref_fixtures_prob")
    r = ou.decode(which="restated", **args)
    assert _string(r, 0, 0) == "acdc"  # tests/test_decode.py:32, test_beam_search_decoder_1
    assert _string(r, 1, 0) == "b'a"   # test_beam_search_decoder_2
    args, _ = gu.load("ref_fixtures_log")  # test_beam_search_decoder_batch_log
    r = ou.decode(which="restated", **args)
    assert (_string(r, 0, 0), _string(r, 1, 0)) == ("acdc", "b'a")


def test_survey_known_answers():
    args, _ = gu.load("ref_fixtures_prob")
    r = ou.decode(which="restated", **args)
    want = [(0, 0, 6.480284, "acdc", [0, 1, 4, 5]), (0, 1, 6.483004, "acd ", [0, 1, 4, 5]), (0, 2, 6.521161, "acda", [0, 1, 4, 5]),
            (1, 0, 4.989980, "b'a", [0, 2, 4]), (1, 1, 5.298550, "b'da", [0, 2, 3, 4]), (1, 2, 5.337018, "b' a", [0, 2, 3, 4])]
    for b, p, sc, s, ts in want:
        assert _string(r, b, p) == s
        assert abs(float(r["scores"][b, p]) - sc) < 1e-5
        assert r["timesteps"][b, p, : len(ts)].tolist() == ts
    assert r["nres"].tolist() == [20, 20]


@pytest.mark.parametrize("name", gu.names())
def test_restatement_matches_committed_reference_fixtures(name):
    args, want = gu.load(name)
    got = ou.decode(which="restated", **args)
    ou.assert_same(got, want, name)


@pytest.mark.skipif(not ou.have_reference(), reason="oracle/_ref not built (needs the reference checkout)")
@pytest.mark.parametrize("name", gu.names())
def test_fixtures_reproduce_from_live_reference(name):
    args, want = gu.load(name)
    got = ou.decode(which="reference", **args)
    ou.assert_same(got, want, name)


CASES = [
    dict(B=3, T=90, V=29, K=10, seed=11),
    dict(B=2, T=250, V=29, K=64, seed=12),
    dict(B=2, T=200, V=29, K=50, seed=13, quant=0.5),
    dict(B=2, T=150, V=5, K=30, seed=14, quant=1.0, blank_id=2),
    dict(B=2, T=150, V=29, K=20, seed=15, blank_bias=5.0),
    dict(B=2, T=60, V=200, K=16, seed=16, top_n=12),
    dict(B=1, T=500, V=29, K=100, seed=17),
    # vocabulary pruning as implemented (decoder_utils.cpp:10-45): the cumulative cut really triggers below ln 2
    dict(B=2, T=120, V=29, K=24, seed=18, top_n=40, cutoff_prob=0.5),
    dict(B=2, T=120, V=29, K=24, seed=19, top_n=10, cutoff_prob=0.6),
    dict(B=2, T=100, V=40, K=16, seed=20, top_n=6, prob_input=True),
    dict(B=2, T=100, V=40, K=16, seed=21, top_n=40, cutoff_prob=0.55, prob_input=True),
    dict(B=2, T=60, V=1000, K=20, seed=22, top_n=40, cutoff_prob=0.99),
    dict(B=2, T=80, V=64, K=20, seed=23, top_n=8, quant=0.5),            # equal values at the cut: std::sort's order decides
    dict(B=2, T=80, V=29, K=20, seed=24, top_n=40, cutoff_prob=0.6, quant=0.25),
    dict(B=2, T=50, V=64, K=8, seed=25, top_n=1),
]


@pytest.mark.skipif(not ou.have_reference(), reason="oracle/_ref not built (needs the reference checkout)")
@pytest.mark.parametrize("c", CASES, ids=lambda c: "B%(B)d_T%(T)d_V%(V)d_K%(K)d_s%(seed)d" % c)
def test_live_differential(c):
    blank = c.get("blank_id", 0)
    lp = ou.synth_logprobs(c["B"], c["T"], c["V"], c["seed"], quant=c.get("quant"), blank_bias=c.get("blank_bias", 0.0), blank_id=blank)
    x = np.exp(lp) if c.get("prob_input") else lp
    kw = dict(beam=c["K"], cutoff_top_n=c.get("top_n", 40), cutoff_prob=c.get("cutoff_prob", 1.0), blank_id=blank, log_input=not c.get("prob_input"))
    ou.assert_same(ou.decode(x, which="restated", **kw), ou.decode(x, which="reference", **kw))


def test_prob_and_log_input_agree_on_labels():
    lp = ou.synth_logprobs(2, 50, 29, 21)
    a = ou.decode(lp, beam=8, log_input=True)
    b = ou.decode(np.exp(lp), beam=8, log_input=False)
    assert np.array_equal(a["tokens"][:, 0], b["tokens"][:, 0])
