"""The GPU decoder's per-utterance core (ctcdecode_amd/csrc/beam_core.h) compiled for the host with a sequential
execution policy, against the oracle and the committed reference fixtures.  This validates the ALGORITHM the HIP kernel
runs (DFS-ordered beam + LCP array, Euler-tour candidate slots, exact tie replay) without a GPU; the -m gpu tests then
check the same source running as a workgroup."""
import numpy as np
import pytest

import degenerate_util as du
import golden_util as gu
import oracle_util as ou


@pytest.mark.parametrize("name", [n for n in gu.names() if "prob" not in n])
def test_core_matches_reference_fixtures(name):
    args, want = gu.load(name)
    assert args.pop("log_input")
    ou.assert_same(ou.decode_core_host(**args), want, name)


def test_core_randomized_against_oracle():
    rng = np.random.default_rng(7)
    stats = np.zeros(5, np.int64)
    for it in range(120):
        V = int(rng.choice([3, 5, 9, 29, 29, 64]))
        K = int(rng.choice([1, 2, 5, 16, 50, 100, 128]))
        T = int(rng.integers(1, 200))
        quant = [None, None, 0.5, 1.0, 0.25, 2.0][int(rng.integers(0, 6))]
        bias = float(rng.choice([0, 0, 3, 6, -2]))
        blank = int(rng.integers(0, V))
        top_n = int(rng.choice([40, 40, 40, max(1, V // 2), 3]))
        lp = ou.synth_logprobs(2, T, V, 5000 + it, quant=quant, blank_bias=bias, blank_id=blank)
        sl = rng.integers(0, T + 5, size=2).astype(np.int32) if it % 4 == 0 else None
        kw = dict(beam=K, blank_id=blank, cutoff_top_n=top_n)
        a = ou.decode(lp, sl, which="restated", want_stats=True, **kw)
        ou.assert_same(a, ou.decode_core_host(lp, sl, **kw), "case %d V=%d K=%d T=%d q=%s" % (it, V, K, T, quant))
        stats += a["stats"].sum(0)
    # the sweep must actually exercise the hard paths: tie splits at the K boundary, revived dead-interior nodes
    assert stats[1] > 100 and stats[2] > 1000, stats


def test_core_north_star_shape():
    lp = ou.synth_logprobs(2, 1000, 29, 5)
    ou.assert_same(ou.decode(lp, beam=100), ou.decode_core_host(lp, beam=100))


def _edge_cases():
    yield "only_blank", ou.synth_logprobs(2, 20, 1, 1), dict(beam=5)
    yield "two_labels_k1", ou.synth_logprobs(2, 30, 2, 2), dict(beam=1, blank_id=1)
    yield "one_frame", ou.synth_logprobs(3, 1, 29, 3), dict(beam=100)
    lp = ou.synth_logprobs(2, 60, 9, 4)
    lp[:, ::3, 2] = -np.inf
    lp[:, 5, :] = -np.inf
    yield "minus_inf_inputs", lp, dict(beam=12)
    lp = ou.synth_logprobs(2, 40, 9, 5)
    lp[:, :, 0] = 0.0
    yield "certain_blank", lp, dict(beam=12)
    lp = np.full((2, 50, 6), np.float32(-1.7917595), np.float32)
    yield "all_equal_maximal_ties", lp, dict(beam=20)
    yield "coarse_quantised", ou.synth_logprobs(1, 300, 4, 7, quant=4.0), dict(beam=64, blank_id=3)
    lp = ou.synth_logprobs(2, 40, 9, 8)
    lp[0, 10:20, :] = -3.0e38
    yield "near_minus_flt_max", lp, dict(beam=12)
    # zeros of both signs (scores that are -0.0 / +0.0 compare equal), positive "log-probabilities", a denormal
    lp = np.zeros((2, 30, 5), np.float32)
    lp[:, :, 1] = -0.0
    lp[:, ::2, 2] = 0.25
    lp[:, 1::3, 3] = -1e-40
    lp[1, :, 0] = -0.0
    yield "signed_zeros_and_positive_values", lp, dict(beam=16)
    # every log-probability -0.0: sums of a prefix score and a log-probability are then never -0 (the score is not), which
    # is what lets the child-scoring loop skip the zero canonicalisation of its keys (beam_core.h ord_f32_raw)
    yield "all_negative_zero", np.full((2, 25, 7), np.float32(-0.0), np.float32), dict(beam=24)


EDGE = list(_edge_cases())


@pytest.mark.parametrize("name,lp,kw", EDGE, ids=[e[0] for e in EDGE])
def test_core_edge_cases(name, lp, kw):
    ou.assert_same(ou.decode(lp, which="restated", **kw), ou.decode_core_host(lp, **kw), name)


@pytest.mark.skipif(not ou.have_reference(), reason="oracle/_ref not built")
@pytest.mark.parametrize("name,lp,kw", EDGE, ids=[e[0] for e in EDGE])
def test_oracle_edge_cases_against_live_reference(name, lp, kw):
    ou.assert_same(ou.decode(lp, which="restated", **kw), ou.decode(lp, which="reference", **kw), name)


def _degenerate_sweep(n, seed, reference_every=0, prune=False, chunked=False):
    rng = np.random.default_rng(seed)
    kinds = set()
    for it in range(n):
        meta, lp = du.make_case(rng)
        kw = dict(beam=meta["K"], blank_id=meta["blank"])
        sl = None
        if prune:
            kw.update(cutoff_top_n=int(rng.choice([40, max(1, meta["V"] // 2), 3, 1])), cutoff_prob=float(rng.choice([1.0, 1.0, 0.9, 0.5])))
            sl = rng.integers(0, meta["T"] + 3, size=2).astype(np.int32) if it % 3 == 0 else None
        want = ou.decode(lp, sl, which="restated", **kw)
        if chunked:
            bounds = sorted(set(int(v) for v in rng.integers(0, meta["T"] + 1, size=int(rng.integers(0, 6)))))
            got = ou.decode_core_host_chunked(lp, bounds, **kw)
        else:
            got = ou.decode_core_host(lp, sl, **kw)
        ou.assert_same(want, got, "degenerate case %d %s %s" % (it, meta, kw))
        if reference_every and it % reference_every == 0 and ou.have_reference():
            ou.assert_same(want, ou.decode(lp, sl, which="reference", **kw), "oracle vs live reference, case %d %s" % (it, meta))
        kinds.add(meta["kind"])
    assert len(kinds) == len(du.KINDS)


def test_core_degenerate_inputs_follow_the_reference_argument_order():
    """Whole frames of -inf, sums that overflow, -inf tail padding without seq_lens: log_sum_exp(-FLT_MAX, -inf) depends
    on the order of its arguments (decoder_utils.h:47-54), i.e. on the permutation std::nth_element left in `prefixes`
    (ctc_beam_search_decoder.cpp:87-142,150-154).  Round 2 always added the repeat contribution first and differed from
    the reference on 20-90 % of such inputs; danger mode (beam_core.h enter_danger) replays the order."""
    _degenerate_sweep(300, 101, reference_every=3)


def test_core_degenerate_inputs_pruned_and_ragged():
    _degenerate_sweep(150, 102, reference_every=5, prune=True)


def test_core_degenerate_inputs_streamed():
    """Danger mode is part of the parked stream state: any chunking of a degenerate utterance equals the one-shot result."""
    _degenerate_sweep(150, 103, chunked=True)


@pytest.mark.parametrize("level", ["1", "2"])
def test_core_degenerate_inputs_hbm_scratch_layouts(monkeypatch, level):
    monkeypatch.setenv("CTC_HOST_BIG", level)
    _degenerate_sweep(60, 104 + int(level))


@pytest.mark.parametrize("level", ["1", "2"])
def test_core_hbm_scratch_layout(monkeypatch, level):
    """The wide-beam workspace layouts: level 1 keeps the rare-path per-slot arrays outside the workgroup's LDS, level 2
    (beam_width ~1000) also the slot keys and the rarely read per-entry arrays."""
    monkeypatch.setenv("CTC_HOST_BIG", level)
    for seed, kw in [(44, dict(quant=0.5)), (45, dict(blank_bias=3.0)), (46, {})]:
        lp = ou.synth_logprobs(2, 150, 29, seed, **kw)
        ou.assert_same(ou.decode(lp, beam=64), ou.decode_core_host(lp, beam=64))
    lp = ou.synth_logprobs(1, 40, 29, 47)
    ou.assert_same(ou.decode(lp, beam=1000), ou.decode_core_host(lp, beam=1000), "beam 1000")


def test_core_rank_table_tags_wrap():
    """The pruned default's compile-time class tags the rank table's entries with the frame (mod 1024) instead of taking a frame's candidates
    out of the table again; the table is wiped when the tags repeat: utterances longer than 1024 frames, against the oracle."""
    for seed, (V, top_n, K, T) in enumerate([(300, 20, 30, 2100), (64, 40, 50, 1100)]):
        lp = ou.synth_logprobs(1, T, V, 4400 + seed)
        want = ou.decode(lp, beam=K, cutoff_top_n=top_n)
        got = ou.decode_core_host(lp, None, beam=K, cutoff_top_n=top_n)
        ou.assert_same(got, want, "rank-table tags V=%d T=%d" % (V, T))


def test_core_streaming_equals_one_shot():
    """The stream state (beam parked between chunks) on the host build: arbitrary chunkings, empty chunks included."""
    rng = np.random.default_rng(3)
    for it in range(40):
        V = int(rng.choice([3, 9, 29]))
        K = int(rng.choice([2, 16, 50, 100]))
        T = int(rng.integers(2, 150))
        quant = [None, 0.5, 1.0][int(rng.integers(0, 3))]
        lp = ou.synth_logprobs(2, T, V, 900 + it, quant=quant, blank_bias=float(rng.choice([0, 3])))
        bounds = sorted(set(int(x) for x in rng.integers(0, T + 1, size=int(rng.integers(0, 6)))))
        ou.assert_same(ou.decode(lp, beam=K), ou.decode_core_host_chunked(lp, bounds, beam=K), "chunks %s" % bounds)


def test_core_time_steps_beyond_16_bits():
    """A pool node packs its label and the low 16 bits of its time step into one word (12-byte nodes); launches that can pass
    frame 65535 keep the high bits in a side array.  One-shot and streamed across the boundary."""
    T = 66500
    lp = ou.synth_logprobs(1, T, 3, 17, blank_bias=2.5)  # blank-dominated: the label sequences stay short, their time steps span all of T
    want = ou.decode(lp, beam=4)
    assert int(want["timesteps"][0, 0, : want["lens"][0, 0]].max()) > 65536
    ou.assert_same(want, ou.decode_core_host(lp, beam=4), "T > 65536")
    ou.assert_same(want, ou.decode_core_host_chunked(lp, [100, 65000, 65536, 65537, 66000], beam=4), "T > 65536, streamed")


def test_core_helpers():
    """ord_f32 (score -> sortable key), the shift forms of the divisions, log2 helpers, info-word packing."""
    import ctypes

    lib = ctypes.CDLL(ou.build_core_host())
    lib.ctccore_check_helpers.restype = ctypes.c_longlong
    lib.ctccore_check_helpers.argtypes = [ctypes.c_ulonglong, ctypes.c_longlong]
    assert lib.ctccore_check_helpers(12345, 2_000_000) == 0


def test_compact_results_expand_to_the_padded_tensors():
    """Compact delivery (beam_core.h OutRefs::c_*, compact_results.h): every entry hands over only the labels it does not
    share with its DFS predecessor; expanded, the result must be the padded tensors exactly -- zeros included."""
    for seed, (B, T, V, K, kw) in enumerate([(3, 200, 29, 50, {}), (2, 150, 9, 100, dict(quant=0.5)), (4, 60, 29, 16, dict(ragged=True)),
                                             (2, 2, 3, 50, dict(blank_id=1)), (2, 300, 29, 100, dict(blank_bias=4.0))]):
        blank = kw.get("blank_id", 0)
        lp = ou.synth_logprobs(B, T, V, 300 + seed, quant=kw.get("quant"), blank_bias=kw.get("blank_bias", 0.0), blank_id=blank)
        sl = np.array([T, 0, 1, T // 2][:B], np.int32) if kw.get("ragged") else None
        want = ou.decode(lp, sl, beam=K, blank_id=blank)
        got = ou.decode_core_host_compact(lp, sl, beam=K, blank_id=blank)
        ou.assert_same(got, want, "compact case %d" % seed)
        assert np.array_equal(got["tokens"], want["tokens"]) and np.array_equal(got["timesteps"], want["timesteps"])  # the zero fill too
        assert got["labels_used"] <= int(want["lens"].sum())
        if T >= 150 and not kw.get("blank_bias"):
            assert got["labels_used"] * 2 < int(want["lens"].sum()), "the beam's sequences overlap: far fewer labels travel than rows hold"


def test_log_softmax_host_twin_is_a_log_softmax():
    """The host twin of the logits pre-pass (exact definition in include/ctcdecode_amd.h) against float64 arithmetic."""
    rng = np.random.default_rng(11)
    for V in (1, 2, 29, 64, 65, 130, 1000):
        x = (rng.standard_normal((7, V)) * 4).astype(np.float32)
        x[1, :] -= 200.0
        if V > 2:
            x[2, 1] = -np.inf
            x[3, :] = -np.inf
        y = ou.log_softmax_rows(x)
        with np.errstate(invalid="ignore", divide="ignore"):
            x64 = x.astype(np.float64)
            m = x64.max(-1, keepdims=True)
            want = x64 - m - np.log(np.exp(x64 - m).sum(-1, keepdims=True))
        ok = np.isfinite(want)
        assert np.allclose(y[ok], want[ok], rtol=0, atol=2e-6 * (1 + np.log(V)))
        assert np.all(np.isneginf(y[~ok]))
