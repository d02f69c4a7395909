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
]


@pytest.mark.skipif(not ou.have_reference(), reason="oracle/_ref not built (needs the reference checkout)")
@pytest.mark.parametrize("c", CASES, ids=lambda c: "B%(B)d_T%(T)d_V%(V)d_K%(K)d_s%(seed)d" % c)
def test_live_differential(c):
    blank = c.get("blank_id", 0)
    lp = ou.synth_logprobs(c["B"], c["T"], c["V"], c["seed"], quant=c.get("quant"), blank_bias=c.get("blank_bias", 0.0), blank_id=blank)
    kw = dict(beam=c["K"], cutoff_top_n=c.get("top_n", 40), blank_id=blank, log_input=True)
    ou.assert_same(ou.decode(lp, which="restated", **kw), ou.decode(lp, which="reference", **kw))


def test_prob_and_log_input_agree_on_labels():
    lp = ou.synth_logprobs(2, 50, 29, 21)
    a = ou.decode(lp, beam=8, log_input=True)
    b = ou.decode(np.exp(lp), beam=8, log_input=False)
    assert np.array_equal(a["tokens"][:, 0], b["tokens"][:, 0])
