"""-m gpu parity tests proper: the HIP path, called through the product API / C ABI, against the oracle and the
committed reference fixtures.  Integer tensors and scores are compared BIT-EXACT on the region the reference defines."""
import os

import numpy as np
import pytest

import golden_util as gu
import oracle_util as ou

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    assert torch.cuda.is_available(), "these tests need an MI355X"
    return torch


def _decode(torch, probs, seq_lens=None, beam=100, cutoff_prob=1.0, cutoff_top_n=40, blank_id=0, log_input=True, threads=None,
            host_entry=False, fixed_layout=True):
    import ctcdecode_amd

    V = probs.shape[2]
    dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], cutoff_top_n=cutoff_top_n, cutoff_prob=cutoff_prob,
                                       beam_width=beam, blank_id=blank_id, log_probs_input=log_input, device="cuda:0")
    if threads:
        dec.set_threads(threads)
    if not fixed_layout:
        dec.set_fixed_layout(False)
    try:
        out, sc, ts, ln = dec.decode(torch.from_numpy(np.ascontiguousarray(probs)),
                                     torch.from_numpy(seq_lens) if seq_lens is not None else None)
    except NotImplementedError as e:
        pytest.skip(str(e))
    return dict(tokens=out.numpy(), timesteps=ts.numpy(), scores=sc.numpy(), lens=ln.numpy())


def _with_nres(got, want):
    got = dict(got)
    got["nres"] = want["nres"]
    # rows the reference does not define must be zero in our output
    for b in range(got["lens"].shape[0]):
        n = int(want["nres"][b])
        assert not got["lens"][b, n:].any() and not got["scores"][b, n:].any() and not got["tokens"][b, n:].any()
    return got


def test_library_is_the_hip_build(torch_mod):
    import ctcdecode_amd._native as n

    assert b"gfx950" in n.lib.ctcd_version()
    maps = open("/proc/self/maps").read()
    assert "libctcdecode_amd.so" in maps


def test_library_builds_on_this_box(torch_mod, tmp_path):
    """The HIP sources compile HERE (hipcc --offload-arch=gfx950 on the GPU box, not only in the build container) and
    the result decodes like the shipped library: the quick build (north-star class kernels only) goes to a scratch
    directory and is driven through the raw C ABI, next to the product library already loaded in this process."""
    import ctypes
    import shutil

    from ctcdecode_amd import _build

    if not os.environ.get("CTCD_TEST_BUILD_ON_BOX"):
        pytest.skip("opt-in (CTCD_TEST_BUILD_ON_BOX=1; run once per round: profiles/r04c_build_on_box.txt -- 8 s on the GPU box)")
    if not (shutil.which("hipcc") or os.path.exists(os.path.join(_build.ROCM, "bin", "hipcc"))):
        pytest.skip("no hipcc on this box")
    torch = torch_mod
    so = _build.build(defines=["CTC_QUICK_BUILD=1"], out=str(tmp_path / "libctcdecode_quick.so"))
    lib = ctypes.CDLL(so)
    h = ctypes.c_void_p()
    lib.ctcd_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]
    assert lib.ctcd_create(ctypes.byref(h), 0) == 0
    lib.ctcd_beam_decode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_double] + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 6
    B, T, V, K = 3, 60, 29, 100
    lp = ou.synth_logprobs(B, T, V, 77)
    d = torch.from_numpy(lp).cuda()
    tok = torch.empty((B, K, T), dtype=torch.int32, device="cuda")
    ts = torch.empty_like(tok)
    sc = torch.empty((B, K), dtype=torch.float32, device="cuda")
    ln = torch.empty((B, K), dtype=torch.int32, device="cuda")
    rc = lib.ctcd_beam_decode(h, d.data_ptr(), None, B, T, V, K, 4, 1.0, 40, 0, 1, tok.data_ptr(), ts.data_ptr(), sc.data_ptr(), ln.data_ptr(), None, None)
    assert rc == 0, rc
    torch.cuda.synchronize()
    got = dict(tokens=tok.cpu().numpy(), timesteps=ts.cpu().numpy(), scores=sc.cpu().numpy(), lens=ln.cpu().numpy())
    want = ou.decode(lp, beam=K)
    ou.assert_same(_with_nres(got, want), want, "library built on this box")
    # ... and exactly like the product library in this process
    mine = _decode(torch, lp, beam=K)
    for k in ("tokens", "timesteps", "scores", "lens"):
        assert np.array_equal(got[k], mine[k]), k


@pytest.mark.parametrize("name", gu.names())
def test_reference_fixtures(torch_mod, name):
    args, want = gu.load(name)
    got = _decode(torch_mod, **args)
    ou.assert_same(_with_nres(got, want), want, name)


def test_reference_golden_strings(torch_mod):
    """tests/test_decode.py:37-53,66-91 of the reference, through the drop-in class."""
    import ctcdecode_amd

    vocab = ["'", " ", "a", "b", "c", "d", "_"]
    args, _ = gu.load("ref_fixtures_prob")
    dec = ctcdecode_amd.CTCBeamDecoder(vocab, beam_width=20, blank_id=vocab.index("_"))
    out, sc, ts, ln = dec.decode(torch_mod.from_numpy(args["probs"]))
    strings = ["".join(vocab[x] for x in out[b][0][: ln[b][0]]) for b in range(2)]
    assert strings == ["acdc", "b'a"]
    dec = ctcdecode_amd.CTCBeamDecoder(vocab, beam_width=20, blank_id=vocab.index("_"), log_probs_input=True, num_processes=24)
    out, sc, ts, ln = dec.decode(torch_mod.from_numpy(args["probs"]).log())
    strings = ["".join(vocab[x] for x in out[b][0][: ln[b][0]]) for b in range(2)]
    assert strings == ["acdc", "b'a"]
    assert out.dtype == torch_mod.int32 and sc.dtype == torch_mod.float32 and tuple(sc.shape) == (2, 20) and tuple(ts.shape) == (2, 20, 6)


CASES = [
    dict(B=4, T=100, V=29, K=10, seed=31),
    dict(B=3, T=250, V=29, K=64, seed=32),
    dict(B=3, T=200, V=29, K=50, seed=33, quant=0.5),
    dict(B=2, T=150, V=5, K=30, seed=34, quant=1.0, blank_id=2),
    dict(B=2, T=150, V=29, K=20, seed=35, blank_bias=5.0),
    dict(B=2, T=120, V=3, K=128, seed=36, quant=0.5, blank_id=1),
    dict(B=2, T=200, V=9, K=100, seed=37, quant=0.25, blank_bias=3.0, blank_id=8),
    dict(B=1, T=60, V=29, K=1, seed=38),
    dict(B=5, T=64, V=29, K=16, seed=39, ragged=True),
]


@pytest.mark.parametrize("threads", [64, 256, 1024])
@pytest.mark.parametrize("c", CASES, ids=lambda c: "B%(B)d_T%(T)d_V%(V)d_K%(K)d_s%(seed)d" % c)
def test_randomized_against_oracle(torch_mod, c, threads):
    blank = c.get("blank_id", 0)
    lp = ou.synth_logprobs(c["B"], c["T"], c["V"], c["seed"], quant=c.get("quant"), blank_bias=c.get("blank_bias", 0.0), blank_id=blank)
    sl = np.array([c["T"], 0, 1, c["T"] // 2, c["T"] + 9][: c["B"]], np.int32) if c.get("ragged") else None
    want = ou.decode(lp, sl, beam=c["K"], blank_id=blank, which="restated")
    got = _decode(torch_mod, lp, sl, beam=c["K"], blank_id=blank, threads=threads)
    ou.assert_same(_with_nres(got, want), want)


PRUNED = [
    dict(B=3, T=80, V=64, K=20, seed=51, top_n=8),
    dict(B=2, T=60, V=200, K=16, seed=52, top_n=12),
    dict(B=2, T=90, V=64, K=24, seed=53, top_n=8, quant=0.5),             # equal values at / above the cut -> host-resolved frames
    dict(B=2, T=70, V=29, K=16, seed=54, top_n=40, cutoff_prob=0.5),       # the cumulative cut really triggers (exp(0.5)-1 = 0.65)
    dict(B=2, T=70, V=29, K=16, seed=55, top_n=10, cutoff_prob=0.6),
    dict(B=2, T=70, V=29, K=16, seed=56, top_n=40, cutoff_prob=0.99),      # BASELINE.json configs[3] setting: never triggers
    dict(B=2, T=60, V=40, K=12, seed=57, top_n=6, prob_input=True),
    dict(B=2, T=50, V=40, K=12, seed=58, top_n=40, cutoff_prob=0.55, prob_input=True),
    dict(B=2, T=40, V=1000, K=10, seed=59, top_n=40, cutoff_prob=0.99, blank_bias=2.0),
    dict(B=3, T=50, V=64, K=8, seed=60, top_n=1),                          # blank may be pruned away entirely
    dict(B=2, T=40, V=1000, K=10, seed=61, top_n=40, cutoff_prob=0.99, quant=0.25),  # configs[3] setting with tie-heavy frames
    dict(B=2, T=40, V=200, K=10, seed=62, top_n=12, quant=1.0),            # (the wave-per-frame prune variant)
    dict(B=2, T=40, V=300, K=10, seed=63, top_n=40, cutoff_prob=0.7, quant=0.5, prob_input=True),
    dict(B=2, T=30, V=12, K=10, seed=64, top_n=5, quant=1.0),              # rows shorter than the introsort threshold
]


@pytest.mark.parametrize("c", PRUNED, ids=lambda c: "V%(V)d_n%(top_n)d_s%(seed)d" % c)
def test_vocabulary_pruning_against_oracle(torch_mod, c):
    import ctcdecode_amd
    import ctcdecode_amd._native as n

    lp = ou.synth_logprobs(c["B"], c["T"], c["V"], c["seed"], quant=c.get("quant"), blank_bias=c.get("blank_bias", 0.0))
    x = np.exp(lp) if c.get("prob_input") else lp
    kw = dict(beam=c["K"], cutoff_top_n=c["top_n"], cutoff_prob=c.get("cutoff_prob", 1.0), log_input=not c.get("prob_input"))
    want = ou.decode(x, which="restated", **kw)
    dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(c["V"])], cutoff_top_n=c["top_n"], cutoff_prob=c.get("cutoff_prob", 1.0),
                                       beam_width=c["K"], log_probs_input=not c.get("prob_input"))
    # frames whose order is std::sort's business (equal values at / above the cut) or whose cumulative sum comes close to
    # cutoff_prob are flagged by the fast prune pass and settled by the device's replay of std::sort + the exact chain of
    # binary64 log / exp -- with its arrays in LDS, and (second round) in global memory, the path of very long rows.
    # Nothing goes to the host toolchain (round 4: the host path is gone).
    for in_lds in (True, False):
        n.check(n.lib.ctcd_debug_set_prune_resolve(dec._handle, 1 if in_lds else 0))
        out, sc, ts, ln = dec.decode(torch_mod.from_numpy(np.ascontiguousarray(x)))
        got = dict(tokens=out.numpy(), timesteps=ts.numpy(), scores=sc.numpy(), lens=ln.numpy())
        ou.assert_same(_with_nres(got, want), want, "replay in LDS %s" % in_lds)
        flagged, host_rows = n.lib.ctcd_last_prune_flagged_rows(dec._handle), n.lib.ctcd_last_prune_host_rows(dec._handle)
        assert host_rows == 0
        if c.get("quant"):
            assert flagged > 0, "quantised inputs must exercise the tie resolution"


@pytest.mark.parametrize("threads", [128, 256, 1024])
def test_pruned_wide_beam_regression(torch_mod, threads):
    """Found by tests/sweeps/gpu_stress.py: with vocabulary pruning and more surviving beams than one wave emits, the rank table
    was un-registered by the first wave while later waves were still emitting (wrong log_prob for beams >= 128)."""
    lp = ou.synth_logprobs(4, 61, 64, 7057, quant=2.0, blank_id=32)
    sl = np.array([11, 30, 32, 47], np.int32)
    want = ou.decode(lp, sl, beam=200, blank_id=32, cutoff_top_n=32, which="restated")
    got = _decode(torch_mod, lp, sl, beam=200, blank_id=32, cutoff_top_n=32, threads=threads)
    ou.assert_same(_with_nres(got, want), want)


def test_randomised_stress_subset(torch_mod):
    """A slice of tests/sweeps/gpu_stress.py (random shapes, pruning, ragged lengths, streaming, every workgroup size)."""
    import subprocess
    import sys as _sys

    r = subprocess.run([_sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "sweeps", "gpu_stress.py"),
                        "--n", "120", "--seed", "5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_north_star_shape_parity_and_properties(torch_mod):
    """BASELINE.json configs[1]: B=256, T=1000, V=29, beam=100.  ALL 256 items bit-exact against the real reference
    (oracle/_ref, one host thread per core: ~20 s on the GPU box; the restatement stands in where it is not built),
    size-independent properties on all of them, and run-to-run determinism."""
    B, T, V, K = 256, 1000, 29, 100
    lp = ou.synth_logprobs(B, T, V, 2024)
    got = _decode(torch_mod, lp, beam=K)
    again = _decode(torch_mod, lp, beam=K)
    for k in got:
        assert np.array_equal(got[k], again[k]), "non-deterministic " + k
    which = "reference" if ou.have_reference() else "restated"
    want = ou.decode(lp, beam=K, which=which, threads=os.cpu_count())
    ou.assert_same(_with_nres(got, want), want, "north-star, all items vs " + which)
    lens, sc, tok, ts = got["lens"], got["scores"], got["tokens"], got["timesteps"]
    assert (np.diff(sc, axis=1) >= 0).all(), "beam_scores must ascend (best first)"
    assert (lens >= 0).all() and (lens <= T).all()
    pos = np.arange(T)[None, None, :]
    valid = pos < lens[:, :, None]
    assert (tok[valid] > 0).all() and (tok[valid] < V).all(), "labels are non-blank vocabulary ids"
    assert (tok[~valid] == 0).all() and (ts[~valid] == 0).all(), "outside the valid region everything is zero"
    assert (ts[valid] >= 0).all() and (ts[valid] < T).all()
    # beams of one item are pairwise distinct prefixes
    for b in range(0, B, 16):
        seen = {tuple(tok[b, p, : lens[b, p]]) for p in range(K)}
        assert len(seen) == K


def test_wide_beam_hbm_scratch_layout(torch_mod):
    """beam_width=500 (BASELINE.json configs[2] per-utterance shape): the per-slot rare-path arrays move to HBM scratch."""
    for seed, T, quant in [(71, 150, None), (72, 120, 0.5)]:
        lp = ou.synth_logprobs(2, T, 29, seed, quant=quant)
        want = ou.decode(lp, beam=500, which="restated")
        got = _decode(torch_mod, lp, beam=500)
        ou.assert_same(_with_nres(got, want), want, "K=500 seed %d" % seed)


def test_rank_table_tags_wrap_offline_and_streamed(torch_mod):
    """Round 6: kernel <0,0,2,true,1024> tags its rank-table entries with the frame (mod 1024) and wipes the table when the tags repeat.
    Utterances longer than 1024 frames -- in one launch, and as streams whose chunks end before, at and behind the wrap -- equal the
    oracle and the run-time layout's kernel (which takes the candidates out of the table every frame)."""
    import ctcdecode_amd

    torch = torch_mod
    V, top_n, K, T = 300, 20, 30, 1150
    lp = ou.synth_logprobs(2, T, V, 4411)
    want = ou.decode(lp, beam=K, cutoff_top_n=top_n)
    got = _decode(torch, lp, beam=K, cutoff_top_n=top_n)
    ou.assert_same(_with_nres(got, want), want, "rank-table tags, one launch")
    rt = _decode(torch, lp, beam=K, cutoff_top_n=top_n, fixed_layout=False)
    for key in ("tokens", "timesteps", "lens"):
        assert np.array_equal(got[key], rt[key]), key
    dec = ctcdecode_amd.OnlineCTCBeamDecoder([str(i) for i in range(V)], cutoff_top_n=top_n, beam_width=K, log_probs_input=True, device="cuda:0")
    states = [ctcdecode_amd.DecoderState(dec) for _ in range(2)]
    x = torch.from_numpy(lp)
    bounds = [0, 400, 1023, 1024, 1025, T]
    for a, b in zip(bounds[:-1], bounds[1:]):
        out, sc, ts, ln = dec.decode(x[:, a:b], states, [b == T] * 2)
    L = out.shape[2]
    chunked = dict(tokens=np.zeros((2, K, T), np.int32), timesteps=np.zeros((2, K, T), np.int32), scores=sc.numpy(), lens=ln.numpy(), nres=want["nres"])
    chunked["tokens"][:, : out.shape[1], :L] = out.numpy()
    chunked["timesteps"][:, : out.shape[1], :L] = ts.numpy()
    ou.assert_same(chunked, want, "rank-table tags, streamed across the wrap")


def test_wide_beam_compile_time_layout_equals_run_time_layout(torch_mod):
    """Round 6: beams of up to 500 over up to 29 labels (configs[2]'s decoder) run the first wide-beam layout at a compile-time size
    (decode_kernel.h LAYOUT 3) and read one packed record per parent in phase B (beam_core.h kParentRec).  Same tensors as the run-time
    layout's kernel bit for bit -- ragged lengths, quantised rows (ties), a small vocabulary -- and both equal the oracle; a vocabulary
    of 30 labels stays on the run-time layout and is checked against the oracle as well."""
    for K, V, T, seed, quant in [(200, 29, 90, 81, None), (333, 29, 70, 82, 0.5), (500, 29, 60, 83, 0.25), (500, 5, 80, 84, None), (160, 29, 50, 85, None)]:
        lp = ou.synth_logprobs(3, T, V, seed, quant=quant)
        sl = np.array([T, T // 2, 0], dtype=np.int32)
        want = ou.decode(lp, sl, beam=K, which="restated")
        got = _decode(torch_mod, lp, sl, beam=K)
        ou.assert_same(_with_nres(got, want), want, "compile-time wide layout K=%d V=%d" % (K, V))
        rt = _decode(torch_mod, lp, sl, beam=K, fixed_layout=False)
        for key in ("tokens", "timesteps", "lens"):
            assert np.array_equal(got[key], rt[key]), (K, V, key)
        assert np.array_equal(got["scores"].view(np.uint32), rt["scores"].view(np.uint32)), (K, V)
    lp = ou.synth_logprobs(2, 60, 30, 86)
    want = ou.decode(lp, beam=400, which="restated")
    ou.assert_same(_with_nres(_decode(torch_mod, lp, beam=400), want), want, "run-time wide layout K=400 V=30")


def test_beam_width_1000(torch_mod):
    """beam_width=1000 at V=29: 31 000 candidate slots; the slot keys and the rarely read per-entry arrays live in HBM scratch
    (workspace level 2), only the beam itself and the per-frame temporaries stay in LDS."""
    for seed, T, quant in [(73, 120, None), (74, 90, 0.5)]:
        lp = ou.synth_logprobs(2, T, 29, seed, quant=quant)
        want = ou.decode(lp, beam=1000, which="restated")
        got = _decode(torch_mod, lp, beam=1000)
        ou.assert_same(_with_nres(got, want), want, "K=1000 seed %d" % seed)


def test_config3_shape_long_wide(torch_mod):
    """T=2000, beam_width=500 (configs[2] is B=2048 over 8 GPUs = 256 such utterances per GPU): the whole per-GPU batch of
    256 utterances with CTCD_FULL_PARITY=1 (the reference needs ~100 core-seconds per utterance at this shape: ten minutes on
    the 256-thread GPU box, measured in round 3 -- 256/256 bit-exact, profiles/r03_parity_sweep.json); 48 utterances in the
    default suite.  All checked against the real reference (the restatement where oracle/_ref is not built)."""
    full = os.environ.get("CTCD_FULL_PARITY") == "1"  # all 256: ten minutes of host time (run once per round: profiles/r03_parity_sweep.json)
    lp = ou.synth_logprobs(256 if full else 48, 2000, 29, 81)
    got = _decode(torch_mod, lp, beam=500)
    which = "reference" if ou.have_reference() else "restated"
    want = ou.decode(lp, beam=500, which=which, threads=os.cpu_count())
    ou.assert_same(_with_nres(got, want), want, "configs[2] shape vs " + which)


def test_config4_shape_large_vocab(torch_mod):
    """BASELINE.json configs[3]: V=10000, beam_width=100, cutoff_top_n=40, cutoff_prob=0.99, B=64, T=500; all 64 items
    are checked bit-exact against the oracle (its per-frame std::sort of 10k values makes the CPU side slow: one thread each)."""
    import ctcdecode_amd
    import ctcdecode_amd._native as n

    B, T, V, K = 64, 500, 10000, 100
    lp = ou.synth_logprobs(B, T, V, 91)
    dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], cutoff_top_n=40, cutoff_prob=0.99, beam_width=K, log_probs_input=True)
    out, sc, ts, ln = dec.decode(torch_mod.from_numpy(lp))
    got = dict(tokens=out.numpy(), timesteps=ts.numpy(), scores=sc.numpy(), lens=ln.numpy())
    sample = list(range(0, B, 1 if (os.cpu_count() or 1) >= 64 else 2))  # every item (every other one on small hosts), against the real reference where it is built
    which = "reference" if ou.have_reference() else "restated"
    want = ou.decode(lp[sample], beam=K, cutoff_top_n=40, cutoff_prob=0.99, which=which, threads=os.cpu_count())
    ou.assert_same(_with_nres({k: v[sample] for k, v in got.items()}, want), want, "configs[3] sample vs " + which)
    assert (np.diff(got["scores"], axis=1) >= 0).all()
    print("configs[3]: frames flagged / resolved on the host:", n.lib.ctcd_last_prune_flagged_rows(dec._handle),
          n.lib.ctcd_last_prune_host_rows(dec._handle), "of", B * T)
    assert n.lib.ctcd_last_prune_host_rows(dec._handle) == 0


def test_edge_cases_on_gpu(torch_mod):
    from test_core_host import EDGE

    for name, lp, kw in EDGE:
        want = ou.decode(lp, which="restated", **kw)
        got = _decode(torch_mod, lp, **kw)
        ou.assert_same(_with_nres(got, want), want, name)


@pytest.mark.parametrize("threads", [128, 1024])
def test_degenerate_inputs_follow_the_reference_argument_order(torch_mod, threads):
    """Whole frames of -inf, sums that overflow, -inf tail padding without seq_lens: the kernel's danger
    mode replays std::nth_element every frame and adds a prefix's two contributions in the order of the reference's
    `prefixes` array (decoder_utils.h:47-54, ctc_beam_search_decoder.cpp:87-142,150-154).  Offline, both workspace layouts,
    and streamed with random chunking; every fourth case also against the live reference when it is built."""
    import ctcdecode_amd
    import degenerate_util as du

    rng = np.random.default_rng(2024 + threads)
    kinds = set()
    for it in range(60):
        meta, lp = du.make_case(rng)
        K, blank, T, V = meta["K"], meta["blank"], meta["T"], meta["V"]
        kw = dict(beam=K, blank_id=blank)
        want = ou.decode(lp, which="restated", **kw)
        if it % 4 == 0 and ou.have_reference():
            ou.assert_same(want, ou.decode(lp, which="reference", **kw), "oracle vs live reference %s" % meta)
        if it % 3 != 2:
            got = _decode(torch_mod, lp, threads=threads, fixed_layout=it % 3 == 0, **kw)
        else:
            dec = ctcdecode_amd.OnlineCTCBeamDecoder([str(i) for i in range(V)], beam_width=K, blank_id=blank, log_probs_input=True)
            dec.set_threads(threads)
            states = [ctcdecode_amd.DecoderState(dec) for _ in range(2)]
            bounds = [0] + sorted(set(int(v) for v in rng.integers(0, T + 1, size=int(rng.integers(0, 4))))) + [T]
            x = torch_mod.from_numpy(lp)
            for i in range(len(bounds) - 1):
                out, sc, ts, ln = dec.decode(x[:, bounds[i]:bounds[i + 1]], states, [i == len(bounds) - 2] * 2)
            got = dict(tokens=np.zeros((2, K, T), np.int32), timesteps=np.zeros((2, K, T), np.int32), scores=sc.numpy(), lens=ln.numpy())
            got["tokens"][:, : out.shape[1], : out.shape[2]] = out.numpy()
            got["timesteps"][:, : out.shape[1], : out.shape[2]] = ts.numpy()
        ou.assert_same(_with_nres(got, want), want, "degenerate case %d %s" % (it, meta))
        kinds.add(meta["kind"])
    assert len(kinds) == len(du.KINDS)


def test_degenerate_frames_inside_the_north_star_shape(torch_mod):
    """-inf tail padding without seq_lens and a -inf frame in the middle at full size (T=1000, V=29, beam 100)."""
    lp = ou.synth_logprobs(4, 1000, 29, 77)
    lp[0, 700:, :] = -np.inf
    lp[1, 500, :] = -np.inf
    lp[2, 100:104, :] = -3.0e38
    want = ou.decode(lp, beam=100, which="reference" if ou.have_reference() else "restated")
    ou.assert_same(_with_nres(_decode(torch_mod, lp, beam=100), want), want)


def test_two_workgroups_per_cu_build(torch_mod):
    """The second build of the fixed-layout kernel (<= 64 VGPRs, exact-replay scratch in HBM with LDS staging: 67 KB of LDS,
    two utterances per CU): forced with set_cu_sharing(1), and picked automatically for a batch that outnumbers the CUs.
    Same results as the default build: random cases, tie-heavy cases (many exact replays), degenerate inputs, pruning."""
    import ctcdecode_amd
    import degenerate_util as du

    def run(lp, sl=None, mode=1, **kw):
        V = lp.shape[2]
        d = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], cutoff_top_n=kw.get("cutoff_top_n", 40), beam_width=kw["beam"],
                                         blank_id=kw.get("blank_id", 0), log_probs_input=True, device="cuda:0")
        d.set_cu_sharing(mode)
        out, sc, ts, ln = d.decode(torch_mod.from_numpy(lp), torch_mod.from_numpy(sl) if sl is not None else None)
        return dict(tokens=out.numpy(), timesteps=ts.numpy(), scores=sc.numpy(), lens=ln.numpy())

    for c in CASES:
        if c["K"] > 128 or c["V"] > 32:
            continue
        blank = c.get("blank_id", 0)
        lp = ou.synth_logprobs(c["B"], c["T"], c["V"], c["seed"], quant=c.get("quant"), blank_bias=c.get("blank_bias", 0.0), blank_id=blank)
        sl = np.array([c["T"], 0, 1, c["T"] // 2, c["T"] + 9][: c["B"]], np.int32) if c.get("ragged") else None
        want = ou.decode(lp, sl, beam=c["K"], blank_id=blank)
        ou.assert_same(_with_nres(run(lp, sl, beam=c["K"], blank_id=blank), want), want, "OCC2 %s" % c)
    lp = ou.synth_logprobs(2, 400, 29, 321, quant=0.5)  # coarse values: ties at the K boundary in most frames -> exact replays
    want = ou.decode(lp, beam=100)
    ou.assert_same(_with_nres(run(lp, beam=100), want), want, "OCC2 tie-heavy")
    lp = ou.synth_logprobs(2, 200, 29, 322)
    want = ou.decode(lp, beam=64, cutoff_top_n=12)
    ou.assert_same(_with_nres(run(lp, beam=64, cutoff_top_n=12), want), want, "OCC2 pruned")
    rng = np.random.default_rng(77)
    for it in range(24):
        meta, lp = du.make_case(rng)
        kw = dict(beam=meta["K"], blank_id=meta["blank"])
        want = ou.decode(lp, **kw)
        ou.assert_same(_with_nres(run(lp, **kw), want), want, "OCC2 degenerate %s" % meta)
    # more utterances than CUs: the library switches to this build by itself
    ncu = torch_mod.cuda.get_device_properties(0).multi_processor_count
    lp = ou.synth_logprobs(ncu + 44, 40, 29, 323, quant=1.0)
    want = ou.decode(lp, beam=100)
    ou.assert_same(_with_nres(run(lp, mode=-1, beam=100), want), want, "B > #CUs")
    ou.assert_same(_with_nres(run(lp, mode=0, beam=100), want), want, "B > #CUs, default build")


def test_time_steps_beyond_16_bits(torch_mod):
    """12-byte pool nodes keep 16 bits of the time step; past frame 65535 the high bits live in a side array (one-shot and
    streamed across the boundary)."""
    import ctcdecode_amd

    T = 66500
    lp = ou.synth_logprobs(2, T, 3, 17, blank_bias=2.5)
    want = ou.decode(lp, beam=4)
    assert int(want["timesteps"][0, 0, : want["lens"][0, 0]].max()) > 65536
    ou.assert_same(_with_nres(_decode(torch_mod, lp, beam=4), want), want, "T > 65536")
    dec = ctcdecode_amd.OnlineCTCBeamDecoder(["0", "1", "2"], beam_width=4, blank_id=0, log_probs_input=True)
    states = [ctcdecode_amd.DecoderState(dec) for _ in range(2)]
    bounds = [0, 100, 65000, 65536, 65537, 66000, T]
    x = torch_mod.from_numpy(lp)
    for i in range(len(bounds) - 1):
        out, sc, ts, ln = dec.decode(x[:, bounds[i]:bounds[i + 1]], states, [i == len(bounds) - 2] * 2)
    got = dict(tokens=np.zeros((2, 4, T), np.int32), timesteps=np.zeros((2, 4, T), np.int32), scores=sc.numpy(), lens=ln.numpy())
    got["tokens"][:, : out.shape[1], : out.shape[2]] = out.numpy()
    got["timesteps"][:, : out.shape[1], : out.shape[2]] = ts.numpy()
    ou.assert_same(_with_nres(got, want), want, "T > 65536, streamed")


@pytest.mark.parametrize("V,K,T,top_n,quant", [(700, 100, 20, 700, None), (1200, 60, 15, 1200, 0.5), (900, 120, 25, 900, 1.0), (2000, 40, 30, 5000, None)])
def test_more_than_65535_candidate_slots(torch_mod, V, K, T, top_n, quant):
    """cutoff_top_n >= V with hundreds to thousands of labels: beam * (V + 2) candidate slots exceed 16 bits (70 k - 108 k
    here).  Workspace level 3: 32-bit slot indices in the exact replay's arrays, every per-slot array in HBM scratch.  Coarse
    values (quant) make most frames tie at the beam boundary -> exact std::nth_element replays over 100 k elements."""
    import ctcdecode_amd

    assert K * (min(V, top_n) + 2) > 65535
    lp = ou.synth_logprobs(2, T, V, 1000 + V, quant=quant)
    sl = np.array([T, T // 2], np.int32)
    want = ou.decode(lp, sl, beam=K, cutoff_top_n=top_n)
    got = _decode(torch_mod, lp, sl, beam=K, cutoff_top_n=top_n)
    ou.assert_same(_with_nres(got, want), want, "V=%d K=%d" % (V, K))
    # the same, streamed in three chunks
    dec = ctcdecode_amd.OnlineCTCBeamDecoder([str(i) for i in range(V)], cutoff_top_n=top_n, beam_width=K, blank_id=0, log_probs_input=True)
    states = [ctcdecode_amd.DecoderState(dec) for _ in range(2)]
    x = torch_mod.from_numpy(lp)
    bounds = [0, 3, T // 2, T]
    for i in range(3):
        lo, hi = bounds[i], bounds[i + 1]
        out, sc, ts, ln = dec.decode(x[:, lo:hi], states, [i == 2] * 2, seq_lens=torch_mod.from_numpy(np.clip(sl - lo, 0, hi - lo).astype(np.int32)))
    g2 = dict(tokens=np.zeros((2, K, T), np.int32), timesteps=np.zeros((2, K, T), np.int32), scores=sc.numpy(), lens=ln.numpy())
    g2["tokens"][:, : out.shape[1], : out.shape[2]] = out.numpy()
    g2["timesteps"][:, : out.shape[1], : out.shape[2]] = ts.numpy()
    ou.assert_same(_with_nres(g2, want), want, "V=%d K=%d streamed" % (V, K))


def test_host_path_of_decode(torch_mod):
    """decode() = CPU tensor in, four CPU tensors out.  Round 3: (a) the kernel is launched before its input has crossed PCIe and
    its row fetch waits for the frame blocks that follow on a second stream; (b) a finished utterance mirrors its compact
    results into page-locked host memory itself and host threads expand utterance by utterance while the kernel still runs.
    Every combination of the two, ragged lengths, the mirror too small for some / all utterances, with a scorer -- all
    against the oracle, and identical to each other."""
    import ctcdecode_amd
    from test_lm import DATA, LABELS29

    B, T, V, K = 48, 300, 29, 60
    lp = ou.synth_logprobs(B, T, V, 4711, quant=0.5)
    sl = np.random.default_rng(3).integers(0, T + 1, size=B).astype(np.int32)
    sl[:4] = [T, 0, 1, 127]
    for seq in (None, sl):
        want = ou.decode(lp, seq, beam=K)
        for streaming in (True, False):
            for mcap in (None, 0, 20000):
                d = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], beam_width=K, log_probs_input=True, device="cuda:0")
                d.set_host_path(input_streaming=streaming, mirror_cap_labels=mcap)
                for rep in range(2):  # (the second call reuses the staging buffers of the first)
                    out, sc, ts, ln = d.decode(torch_mod.from_numpy(lp), torch_mod.from_numpy(seq) if seq is not None else None)
                    got = dict(tokens=out.numpy(), timesteps=ts.numpy(), scores=sc.numpy(), lens=ln.numpy())
                    ou.assert_same(_with_nres(got, want), want, "streaming=%s mirror=%s ragged=%s rep %d" % (streaming, mcap, seq is not None, rep))
    path = os.path.join(DATA, "test.arpa")
    lp = ou.synth_logprobs(40, 260, 29, 4712)
    lp[:, :, LABELS29.index(" ")] += np.float32(1.0)
    scr = ou.Scorer(0.5, 1.0, path, LABELS29, "restated")
    want = ou.decode(lp, scorer=scr, beam=40)
    for streaming in (True, False):
        d = ctcdecode_amd.CTCBeamDecoder(LABELS29, model_path=path, alpha=0.5, beta=1.0, beam_width=40, log_probs_input=True, device="cuda:0")
        d.set_host_path(input_streaming=streaming)
        out, sc, ts, ln = d.decode(torch_mod.from_numpy(lp))
        got = dict(tokens=out.numpy(), timesteps=ts.numpy(), scores=sc.numpy(), lens=ln.numpy())
        ou.assert_same(_with_nres(got, want), want, "with scorer, streaming=%s" % streaming)


def test_streamed_input_that_never_arrives_gives_up_quickly(torch_mod):
    """the kernel's wait for streamed rows gives up after about a second -- once, not once per frame -- and the call
    repeats itself with the rows copied up front.  Also: a workgroup narrower than the vocabulary (rows not prefetched) must
    not read rows that have not arrived (the streaming is simply not used there)."""
    import time

    import ctcdecode_amd
    import ctcdecode_amd._native as n

    B, T, V, K = 16, 1024, 300, 8   # 19 MB of rows: streamed (>= 1 MB, T >= 128, V <= 512)
    lp = ou.synth_logprobs(B, T, V, 99)
    want = ou.decode(lp, beam=K, cutoff_top_n=V, which="restated")
    dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], cutoff_top_n=V, beam_width=K, log_probs_input=True)
    n.check(n.lib.ctcd_debug_set_host_path(dec._handle, 2, -2))
    t0 = time.time()
    out, sc, ts, ln = dec.decode(torch_mod.from_numpy(lp))
    dt = time.time() - t0
    got = dict(tokens=out.numpy(), timesteps=ts.numpy(), scores=sc.numpy(), lens=ln.numpy())
    ou.assert_same(_with_nres(got, want), want, "after the give-up")
    assert dt < 20.0, "the give-up must not be paid once per frame (%.1f s)" % dt
    # a workgroup of 64 threads over 300 labels: rows are not prefetched
    dec2 = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], cutoff_top_n=V, beam_width=K, log_probs_input=True)
    dec2.set_threads(64)
    out, sc, ts, ln = dec2.decode(torch_mod.from_numpy(lp))
    got = dict(tokens=out.numpy(), timesteps=ts.numpy(), scores=sc.numpy(), lens=ln.numpy())
    ou.assert_same(_with_nres(got, want), want, "64 threads, 300 labels")


def test_capability_boundaries(torch_mod):
    """Every CTCD_EUNSUPPORTED edge of the no-LM path: on the supported side of a limit the call decodes
    and matches the oracle; one step beyond, it raises NotImplementedError -- cleanly: the same decoder object then decodes
    an ordinary batch correctly.  The limits: beam_width <= 16383, labels <= 65534 (<= 32767 when pruning), one workgroup's
    LDS (beam ~1100 at V=29), K * (candidates + 2) <= 16 777 215 slots.  Up to 65535 slots the ordinary layouts apply;
    beyond (cutoff_top_n >= V with thousands of labels: round 2 refused these) the layout with 32-bit slot indices and
    everything per slot in HBM scratch takes over -- slowly, as the reference is at such shapes (decoder_utils.cpp:33-35)."""
    import ctcdecode_amd

    def dec_for(V, K, top_n=40, cutoff_prob=1.0):
        return ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], cutoff_top_n=top_n, cutoff_prob=cutoff_prob, beam_width=K,
                                            log_probs_input=True, device="cuda:0")

    def check(V, K, top_n, T=12, B=2, seed=5):
        lp = ou.synth_logprobs(B, T, V, seed)
        d = dec_for(V, K, top_n)
        out, sc, ts, ln = d.decode(torch_mod.from_numpy(lp))
        want = ou.decode(lp, beam=K, cutoff_top_n=top_n)
        got = dict(tokens=out.numpy(), timesteps=ts.numpy(), scores=sc.numpy(), lens=ln.numpy())
        ou.assert_same(_with_nres(got, want), want, "V=%d K=%d top_n=%d" % (V, K, top_n))

    def refused(V, K, top_n, T=12):
        d = dec_for(V, K, top_n)
        lp = ou.synth_logprobs(2, T, V, 6)
        with pytest.raises(NotImplementedError):
            d.decode(torch_mod.from_numpy(lp))
        # the refusal left the decoder usable: a second, supported call on the same library state decodes correctly
        check(29, 10, 40)

    # candidate slots: K * (min(V, top_n) + 2); 65535 is where the 32-bit-slot layout takes over
    check(1000, 65, 1000)              # 65 * 1002 = 65130 slots, no pruning (cutoff_top_n >= V)
    check(1000, 66, 1000)              # 66 * 1002 = 66132: workspace level 3
    check(300, 200, 300, T=8)          # 200 * 302 = 60400
    check(300, 220, 300, T=8)          # 66440
    check(2000, 100, 600, T=8)         # pruned: 100 * 602 = 60200
    check(2000, 110, 600, T=8)         # 66220
    check(3000, 40, 5000, T=10, seed=8)   # 120 080 slots, every label a candidate
    refused(60000, 300, 60000, T=2)    # 18 000 600 slots
    # the beam's own arrays: beam 1000 fits one workgroup's LDS (test_beam_width_1000), 1400 does not
    refused(29, 1400, 40)
    refused(29, 20000, 40)             # beyond the 14-bit entry index as well
    # labels
    refused(40000, 4, 40)              # pruning with more than 32767 labels
    lp = ou.synth_logprobs(1, 3, 65535, 7)
    with pytest.raises(NotImplementedError):
        dec_for(65535, 1, 65535).decode(torch_mod.from_numpy(lp))
    # wide beams decode without a scorer but not (yet) with one: see tests/test_gpu_lm.py::test_lm_capability_boundaries


def test_host_pointer_entry_point(torch_mod):
    """ctcd_beam_decode_host: the entry point a maintainer of the reference would bind in place of paddle_beam_decode
     -- CPU buffers in, CPU buffers out, through the raw C ABI."""
    import ctypes

    import ctcdecode_amd._native as n

    args, want = gu.load("ragged_b5_t60_k16")
    probs = np.ascontiguousarray(args["probs"])
    sl = np.ascontiguousarray(args["seq_lens"], np.int32)
    B, T, V = probs.shape
    K = args["beam"]
    tok = np.full((B, K, T), -7, np.int32)
    ts = np.full((B, K, T), -7, np.int32)
    sc = np.full((B, K), -7, np.float32)
    ln = np.full((B, K), -7, np.int32)
    nres = np.zeros((B,), np.int32)
    h = ctypes.c_void_p()
    n.check(n.lib.ctcd_create(ctypes.byref(h), 0))
    try:
        n.check(n.lib.ctcd_beam_decode_host(h, probs.ctypes.data, sl.ctypes.data, B, T, V, K, 4, 1.0, args["cutoff_top_n"], args["blank_id"], 1,
                                            tok.ctypes.data, ts.ctypes.data, sc.ctypes.data, ln.ctypes.data, nres.ctypes.data))
    finally:
        n.lib.ctcd_destroy(h)
    got = dict(tokens=tok, timesteps=ts, scores=sc, lens=ln, nres=nres)
    ou.assert_same(got, want, "host entry point")
    assert np.array_equal(nres, want["nres"])
    # same entry point with vocabulary pruning and tied values (the flagged frames' replay in LDS, then in global memory)
    lp = ou.synth_logprobs(3, 90, 64, 53, quant=0.5)
    want = ou.decode(lp, beam=24, cutoff_top_n=8, which="restated")
    B, T, V, K = 3, 90, 64, 24
    tok = np.full((B, K, T), -7, np.int32); ts = np.full((B, K, T), -7, np.int32)
    sc = np.full((B, K), -7, np.float32); ln = np.full((B, K), -7, np.int32); nres = np.zeros((B,), np.int32)
    h = ctypes.c_void_p()
    n.check(n.lib.ctcd_create(ctypes.byref(h), 0))
    try:
        n.check(n.lib.ctcd_beam_decode_host(h, lp.ctypes.data, None, B, T, V, K, 4, 1.0, 8, 0, 1,
                                            tok.ctypes.data, ts.ctypes.data, sc.ctypes.data, ln.ctypes.data, nres.ctypes.data))
        assert n.lib.ctcd_last_prune_flagged_rows(h) > 0 and n.lib.ctcd_last_prune_host_rows(h) == 0
        n.check(n.lib.ctcd_debug_set_prune_resolve(h, 0))  # and with the replay's arrays in global memory (its own scratch buffer)
        n.check(n.lib.ctcd_beam_decode_host(h, lp.ctypes.data, None, B, T, V, K, 4, 1.0, 8, 0, 1,
                                            tok.ctypes.data, ts.ctypes.data, sc.ctypes.data, ln.ctypes.data, nres.ctypes.data))
        assert n.lib.ctcd_last_prune_flagged_rows(h) > 0 and n.lib.ctcd_last_prune_host_rows(h) == 0
    finally:
        n.lib.ctcd_destroy(h)
    ou.assert_same(dict(tokens=tok, timesteps=ts, scores=sc, lens=ln, nres=nres), want, "host entry point, pruned")
    for b in range(B):
        for p in range(K):
            assert not tok[b, p, ln[b, p]:].any() and not ts[b, p, ln[b, p]:].any()


def test_streaming_reference_cases(torch_mod):
    """tests/test_decode.py:117-139,161-187 of the reference (online decoder, no LM): whole utterance in one call, and
    split into two chunks."""
    import ctcdecode_amd

    vocab = ["'", " ", "a", "b", "c", "d", "_"]
    args, _ = gu.load("ref_fixtures_prob")
    probs = torch_mod.from_numpy(args["probs"])
    dec = ctcdecode_amd.OnlineCTCBeamDecoder(vocab, beam_width=20, blank_id=vocab.index("_"))
    s1, s2 = ctcdecode_amd.DecoderState(dec), ctcdecode_amd.DecoderState(dec)
    out, sc, ts, ln = dec.decode(probs, [s1, s2], [True, True])
    assert ["".join(vocab[x] for x in out[b][0][: ln[b][0]]) for b in range(2)] == ["acdc", "b'a"]
    s1, s2 = ctcdecode_amd.DecoderState(dec), ctcdecode_amd.DecoderState(dec)
    out, sc, ts, ln = dec.decode(probs[:, :2], [s1, s2], [False, False])
    assert tuple(out.shape) == (2, 0, 0) and (ln == 0).all()
    out, sc, ts, ln = dec.decode(probs[:, 2:], [s1, s2], [True, True])
    assert ["".join(vocab[x] for x in out[b][0][: ln[b][0]]) for b in range(2)] == ["acdc", "b'a"]
    assert out.shape[2] >= int(ln.max())  # test_online_decoder_decoding_with_a_lot_calls_no_lm_check_size


@pytest.mark.parametrize("case", [dict(T=240, V=29, K=50, seed=61, cuts=[1, 2, 100, 100, 239]), dict(T=150, V=9, K=100, seed=62, quant=0.5, cuts=[0, 75, 150]),
                                  dict(T=300, V=29, K=20, seed=63, blank_bias=4.0, cuts=[37, 38, 200]), dict(T=120, V=64, K=16, seed=64, top_n=8, cuts=[60])],
                         ids=lambda c: "T%(T)d_V%(V)d_K%(K)d" % c)
def test_streaming_equals_one_shot(torch_mod, case):
    """Feeding an utterance in arbitrary chunks (empty ones included, also an empty final chunk) must give exactly the
    one-shot result, which is checked against the oracle; timesteps keep counting across chunks."""
    import ctcdecode_amd

    c = case
    B = 3
    lp = ou.synth_logprobs(B, c["T"], c["V"], c["seed"], quant=c.get("quant"), blank_bias=c.get("blank_bias", 0.0))
    want = ou.decode(lp, beam=c["K"], cutoff_top_n=c.get("top_n", 40), which="restated")
    dec = ctcdecode_amd.OnlineCTCBeamDecoder([str(i) for i in range(c["V"])], beam_width=c["K"], cutoff_top_n=c.get("top_n", 40), log_probs_input=True)
    states = [ctcdecode_amd.DecoderState(dec) for _ in range(B)]
    x = torch_mod.from_numpy(lp)
    bounds = [0] + list(c["cuts"]) + [c["T"]]
    for i in range(len(bounds) - 1):
        last = i == len(bounds) - 2
        # (check=False: a chunk in which no stream ends is queued without waiting for its status words -- the serving-loop form)
        out, sc, ts, ln = dec.decode(x[:, bounds[i]:bounds[i + 1]], states, [last] * B, check=bool(i % 2))
    K, T = c["K"], c["T"]
    L = out.shape[2]
    got = dict(tokens=np.zeros((B, K, T), np.int32), timesteps=np.zeros((B, K, T), np.int32), scores=sc.numpy(), lens=ln.numpy(), nres=want["nres"])
    got["tokens"][:, : out.shape[1], :L] = out.numpy()
    got["timesteps"][:, : out.shape[1], :L] = ts.numpy()
    ou.assert_same(got, want, "chunked")


def test_streaming_two_groups_of_streams_on_two_hip_streams(torch_mod):
    """Two groups of streams, each behind its own decoder on its own HIP stream, fed alternately without host synchronisation
    (check=False) -- the serving form that hides a chunk's stragglers (bench.py: streaming.two_groups_in_flight) -- give exactly the
    one-shot results."""
    import ctcdecode_amd

    V, K, T, B, chunk = 29, 40, 400, 6, 50
    lps = [ou.synth_logprobs(B, T, V, 171), ou.synth_logprobs(B, T, V, 172)]
    wants = [ou.decode(lp, beam=K, which="restated") for lp in lps]
    decs = [ctcdecode_amd.OnlineCTCBeamDecoder([str(i) for i in range(V)], beam_width=K, cutoff_top_n=V, log_probs_input=True) for _ in range(2)]
    states = [[ctcdecode_amd.DecoderState(d) for _ in range(B)] for d in decs]
    xs = [torch_mod.from_numpy(lp).cuda() for lp in lps]
    hs = [torch_mod.cuda.Stream(), torch_mod.cuda.Stream()]
    torch_mod.cuda.synchronize()
    res = [None, None]
    for f0 in range(0, T, chunk):
        last = f0 + chunk >= T
        for g in range(2):
            with torch_mod.cuda.stream(hs[g]):
                res[g] = decs[g].decode(xs[g][:, f0:f0 + chunk].contiguous(), states[g], [last] * B, check=last)
    torch_mod.cuda.synchronize()
    for g in range(2):
        out, sc, ts, ln = res[g]
        L = out.shape[2]
        got = dict(tokens=np.zeros((B, K, T), np.int32), timesteps=np.zeros((B, K, T), np.int32), scores=sc.numpy(), lens=ln.numpy(), nres=wants[g]["nres"])
        got["tokens"][:, : out.shape[1], :L] = out.numpy()
        got["timesteps"][:, : out.shape[1], :L] = ts.numpy()
        ou.assert_same(got, wants[g], "group %d" % g)


def test_streaming_mixed_batch_and_growth(torch_mod):
    """Streams of different ages in one batch (one ends while the other continues), ragged chunk lengths, and a stream
    that outgrows its initial node pool (1024 frames)."""
    import ctcdecode_amd

    V, K = 29, 12
    lp = ou.synth_logprobs(2, 1500, V, 66)
    want = ou.decode(lp, np.array([1500, 700], np.int32), beam=K, which="restated")
    dec = ctcdecode_amd.OnlineCTCBeamDecoder([str(i) for i in range(V)], beam_width=K, log_probs_input=True)
    a, b = ctcdecode_amd.DecoderState(dec), ctcdecode_amd.DecoderState(dec)
    x = torch_mod.from_numpy(lp)
    dec.decode(x[:, :500], [a, b], [False, False])
    o1, s1, t1, l1 = dec.decode(x[:, 500:1000], [a, b], [False, True], seq_lens=torch_mod.tensor([500, 200]))  # b ends at frame 700
    assert int(l1[0].max()) == 0 and int(l1[1, 0]) == int(want["lens"][1, 0])
    assert np.array_equal(o1[1, 0, : l1[1, 0]].numpy(), want["tokens"][1, 0, : want["lens"][1, 0]])
    assert np.array_equal(s1[1].numpy().view(np.uint32), want["scores"][1].view(np.uint32))
    o2, s2, t2, l2 = dec.decode(x[:1, 1000:], [a], [True])
    assert np.array_equal(l2[0].numpy(), want["lens"][0]) and np.array_equal(s2[0].numpy().view(np.uint32), want["scores"][0].view(np.uint32))
    for p in range(K):
        n = int(l2[0, p])
        assert np.array_equal(o2[0, p, :n].numpy(), want["tokens"][0, p, :n]) and np.array_equal(t2[0, p, :n].numpy(), want["timesteps"][0, p, :n])


def test_streaming_final_call_compact_delivery(torch_mod):
    """The call that ends streams hands its results over as compact records expanded on the host, straight into [B, R, L] tensors
    (ctcd_stream_decode_to_host; R = the most results, L = the longest beam, binding.cpp:186-205): same tensors with the records
    coming from the kernel's page-locked mirror and -- mirror too small for any of them -- fetched from the device afterwards; a
    batch in which only some streams end; R and L equal to what the padded delivery (ctcd_stream_decode) returns."""
    import ctcdecode_amd
    import ctcdecode_amd._native as n

    V, K, T, B = 29, 30, 260, 5
    lp = ou.synth_logprobs(B, T, V, 181)
    x = torch_mod.from_numpy(lp).cuda()
    want = ou.decode(lp, beam=K, which="restated")
    res = []
    for mirror_cap in (-1, 0):
        dec = ctcdecode_amd.OnlineCTCBeamDecoder([str(i) for i in range(V)], beam_width=K, cutoff_top_n=V, log_probs_input=True)
        n.check(n.lib.ctcd_debug_set_host_path(dec._handle, 1, mirror_cap))
        st = [ctcdecode_amd.DecoderState(dec) for _ in range(B)]
        dec.decode(x[:, :100].contiguous(), st, [False] * B)
        out, sc, ts, ln = dec.decode(x[:, 100:].contiguous(), st, [True] * B)
        R, L = int(want["nres"].max()), int(want["lens"].max())
        assert tuple(out.shape) == (B, R, L) and tuple(ts.shape) == (B, R, L) and not out.is_cuda
        got = dict(tokens=np.zeros((B, K, T), np.int32), timesteps=np.zeros((B, K, T), np.int32), scores=sc.numpy(), lens=ln.numpy(), nres=want["nres"])
        got["tokens"][:, :R, :L] = out.numpy()
        got["timesteps"][:, :R, :L] = ts.numpy()
        ou.assert_same(got, want, "mirror cap %d" % mirror_cap)
        res.append((out.numpy().copy(), ts.numpy().copy()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    # only streams 1 and 3 end; the others' rows are zero and they go on afterwards
    dec = ctcdecode_amd.OnlineCTCBeamDecoder([str(i) for i in range(V)], beam_width=K, cutoff_top_n=V, log_probs_input=True)
    st = [ctcdecode_amd.DecoderState(dec) for _ in range(B)]
    ends = [False, True, False, True, False]
    w2 = ou.decode(lp[:, :150], beam=K, which="restated")
    out, sc, ts, ln = dec.decode(x[:, :150].contiguous(), st, ends)
    for b in range(B):
        if ends[b]:
            assert np.array_equal(ln[b].numpy(), w2["lens"][b]) and np.array_equal(sc[b].numpy().view(np.uint32), w2["scores"][b].view(np.uint32))
            for p in range(int(w2["nres"][b])):
                m = int(ln[b, p])
                assert np.array_equal(out[b, p, :m].numpy(), w2["tokens"][b, p, :m]) and np.array_equal(ts[b, p, :m].numpy(), w2["timesteps"][b, p, :m])
                assert not out[b, p, m:].any()
        else:
            assert not out[b].any() and not ln[b].any() and not sc[b].any()
    rest = [st[b] for b in range(B) if not ends[b]]
    keep = [b for b in range(B) if not ends[b]]
    out, sc, ts, ln = dec.decode(x[keep, 150:].contiguous(), rest, [True] * len(rest))
    for i, b in enumerate(keep):
        assert np.array_equal(ln[i].numpy(), want["lens"][b]) and np.array_equal(sc[i].numpy().view(np.uint32), want["scores"][b].view(np.uint32))
        m = int(ln[i, 0])
        assert np.array_equal(out[i, 0, :m].numpy(), want["tokens"][b, 0, :m])


def test_empty_and_degenerate_batches(torch_mod):
    import ctcdecode_amd

    dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(5)], beam_width=4, log_probs_input=True)
    out, sc, ts, ln = dec.decode(torch_mod.zeros((0, 7, 5)))
    assert tuple(out.shape) == (0, 4, 7)
    out, sc, ts, ln = dec.decode(torch_mod.zeros((2, 0, 5)))
    assert tuple(out.shape) == (2, 4, 0) and (ln == 0).all()
    with pytest.raises(ValueError):
        dec.decode(torch_mod.zeros((1, 3, 6)))
    with pytest.raises(ValueError):
        ctcdecode_amd.CTCBeamDecoder(["a", "b"], blank_id=5, log_probs_input=True).decode(torch_mod.zeros((1, 3, 2)))
    with pytest.raises(ValueError):  # scorer.cpp:57 aborts on a bad path; here it is an error
        ctcdecode_amd.CTCBeamDecoder(["a", "b"], model_path="/no/such/lm.arpa")


def test_device_math_bit_exact_vs_host_libm(torch_mod):
    """exact_math.h on the GPU against the GPU box's own glibc expf/logf (what the reference's log_sum_exp calls)."""
    import ctypes

    import ctcdecode_amd
    import ctcdecode_amd._native as n

    dec = ctcdecode_amd.CTCBeamDecoder(["a", "b"], log_probs_input=True)
    chk, bad = ctypes.c_longlong(), ctypes.c_longlong()
    # logf on every float in [1, 2]
    n.check(n.lib.ctcd_debug_math_check(dec._handle, 1, 0x3F800000, 0x40000000, 1, None, None, 0, ctypes.byref(chk), ctypes.byref(bad)))
    assert chk.value == 0x40000000 - 0x3F800000 + 1 and bad.value == 0, (chk.value, bad.value)
    # expf on [-88, -0], every 7th float, plus the far tail
    n.check(n.lib.ctcd_debug_math_check(dec._handle, 0, 0x80000000, 0xC2B00000, 7, None, None, 0, ctypes.byref(chk), ctypes.byref(bad)))
    assert bad.value == 0 and chk.value > 1.5e8
    n.check(n.lib.ctcd_debug_math_check(dec._handle, 0, 0xC2B00000, 0xFF7FFFFF, 100003, None, None, 0, ctypes.byref(chk), ctypes.byref(bad)))
    assert bad.value == 0
    rng = np.random.default_rng(5)
    N = 1 << 22
    x = rng.uniform(-4000, 0, N).astype(np.float32)
    y = (x - rng.uniform(0, 30, N).astype(np.float32) * np.where(rng.integers(0, 3, N) == 0, np.float32(0.05), np.float32(1))).astype(np.float32)
    y[::7] = x[::7]
    y[::1013] = -np.finfo(np.float32).max
    swap = rng.integers(0, 2, N).astype(bool)
    x, y = np.where(swap, y, x), np.where(swap, x, y)
    x, y = np.ascontiguousarray(x), np.ascontiguousarray(y)
    n.check(n.lib.ctcd_debug_math_check(dec._handle, 2, 0, 0, 1, x.ctypes.data, y.ctypes.data, N, ctypes.byref(chk), ctypes.byref(bad)))
    assert chk.value == N and bad.value == 0


def test_device_math_f64_bit_exact_vs_host_libm(torch_mod):
    """exact_math_f64.h on the GPU against the GPU box's own glibc log / exp in binary64 -- what the reference's pruning and its
    probability -> log conversion call (decoder_utils.cpp:16,29,42; decoder_utils.h:47-54 with T = double)."""
    import ctypes

    import ctcdecode_amd
    import ctcdecode_amd._native as n

    dec = ctcdecode_amd.CTCBeamDecoder(["a", "b"], log_probs_input=True)
    chk, bad = ctypes.c_longlong(), ctypes.c_longlong()
    # log(p) and log(p + FLT_MIN) on every 16th positive float (subnormal .. +inf), plus the zeros, a negative and a NaN pattern
    for mode in (3, 4):
        n.check(n.lib.ctcd_debug_math_check(dec._handle, mode, 0x00000001, 0x7F800000, 16, None, None, 0, ctypes.byref(chk), ctypes.byref(bad)))
        assert bad.value == 0 and chk.value > 1.3e8, (mode, chk.value, bad.value)
        for lo in (0x00000000, 0x80000000, 0xBF800000, 0x7FC00000):
            n.check(n.lib.ctcd_debug_math_check(dec._handle, mode, lo, lo, 1, None, None, 0, ctypes.byref(chk), ctypes.byref(bad)))
            assert bad.value == 0, (mode, hex(lo))
    # every float of [0.9, 1.1]: the branch around 1 of the binary64 log
    n.check(n.lib.ctcd_debug_math_check(dec._handle, 3, 0x3F666666, 0x3F8CCCCD, 1, None, None, 0, ctypes.byref(chk), ctypes.byref(bad)))
    assert bad.value == 0 and chk.value > 2.5e6
    # exp on every 16th negative float (-0 .. -FLT_MAX, -inf) and on the positive ones up to 128
    n.check(n.lib.ctcd_debug_math_check(dec._handle, 5, 0x80000000, 0xFF800000, 16, None, None, 0, ctypes.byref(chk), ctypes.byref(bad)))
    assert bad.value == 0 and chk.value > 1.3e8
    n.check(n.lib.ctcd_debug_math_check(dec._handle, 5, 0x00000000, 0x43000000, 16, None, None, 0, ctypes.byref(chk), ctypes.byref(bad)))
    assert bad.value == 0
    # log_sum_exp<double> on pairs as the cumulative cut sees them (a running value >= 0 against log-probabilities)
    rng = np.random.default_rng(6)
    N = 1 << 22
    x = rng.uniform(0, 0.7, N).astype(np.float32)
    y = (-rng.exponential(4.0, N)).astype(np.float32)
    y[::5] = (-rng.uniform(0, 120, N)[::5]).astype(np.float32)
    y[::1013] = -np.finfo(np.float32).max
    y[::2027] = -np.inf
    swap = rng.integers(0, 2, N).astype(bool)
    x, y = np.where(swap, y, x), np.where(swap, x, y)
    x, y = np.ascontiguousarray(x), np.ascontiguousarray(y)
    n.check(n.lib.ctcd_debug_math_check(dec._handle, 6, 0, 0, 1, x.ctypes.data, y.ctypes.data, N, ctypes.byref(chk), ctypes.byref(bad)))
    assert chk.value == N and bad.value == 0


def test_prune_nan_rows_are_defined(torch_mod):
    """A NaN among the values of a pruned frame: the reference's std::sort call is undefined behaviour there (its comparator is
    not a strict weak order); here a NaN ranks below every number.  The call must neither hang nor touch the host, and a row
    whose NaNs lie outside the kept top_n must decode exactly as the same row with -inf in their place."""
    import ctcdecode_amd
    import ctcdecode_amd._native as n

    lp = ou.synth_logprobs(2, 40, 64, 71)
    bad = lp.copy()
    bad[:, ::3, 50:] = np.nan                  # the lowest ranks of every third frame
    ref = lp.copy()
    ref[:, ::3, 50:] = -np.inf
    dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(64)], cutoff_top_n=8, beam_width=16, log_probs_input=True)
    a = dec.decode(torch_mod.from_numpy(bad))
    b = dec.decode(torch_mod.from_numpy(ref))
    for u, v in zip(a, b):
        assert torch_mod.equal(u, v)
    assert n.lib.ctcd_last_prune_host_rows(dec._handle) == 0


@pytest.mark.parametrize("threads", [256, 1024])
def test_both_workspace_layouts(torch_mod, threads):
    """Small shapes normally run the kernel variant with compile-time LDS addresses; the run-time layout (used by wide
    beams / large vocabularies) must give the same, oracle-identical, results on them."""
    for seed, (B, T, V, K, top_n) in enumerate([(3, 150, 29, 100, 40), (2, 90, 8, 128, 3), (2, 60, 32, 17, 40), (2, 200, 29, 64, 40)]):
        lp = ou.synth_logprobs(B, T, V, 900 + seed, quant=[None, 0.5, None, 1.0][seed])
        want = ou.decode(lp, beam=K, cutoff_top_n=top_n)
        for fixed in (True, False):
            got = _decode(torch_mod, lp, beam=K, cutoff_top_n=top_n, threads=threads, fixed_layout=fixed)
            ou.assert_same(_with_nres(got, want), want, "layout fixed=%s B%d T%d V%d K%d" % (fixed, B, T, V, K))


def test_workgroup_size_must_be_a_power_of_two(torch_mod):
    import ctcdecode_amd

    dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(5)], beam_width=4, log_probs_input=True, device="cuda:0")
    for ok in (0, 64, 128, 256, 512, 1024):
        dec.set_threads(ok)
    for bad in (32, 96, 768, 2048, -64):
        with pytest.raises(ValueError):
            dec.set_threads(bad)


def test_result_delivery_paths_agree(torch_mod):
    """decode() (compact over PCIe + host expansion), decode_padded() (padded tensors over PCIe), decode_device() (HBM) and
    decode_compact() + expand_compact() must give identical tensors, zero fill included; inputs on either side."""
    import ctcdecode_amd

    for seed, (B, T, V, K, ragged) in enumerate([(5, 300, 29, 100, True), (3, 120, 9, 40, False), (2, 1, 29, 7, False), (300, 50, 29, 8, False)]):
        lp = ou.synth_logprobs(B, T, V, 1200 + seed, quant=0.5 if seed == 1 else None)
        sl = np.array([(53 * i) % (T + 4) for i in range(B)], np.int32) if ragged else None
        dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], beam_width=K, log_probs_input=True, num_processes=7)
        x = torch_mod.from_numpy(lp)
        s = torch_mod.from_numpy(sl) if sl is not None else None
        a = dec.decode(x, s)
        b = dec.decode_padded(x, s)
        c = dec.decode(x.cuda(), s.cuda() if s is not None else None)
        d = tuple(t.cpu() for t in dec.decode_device(x, s))
        hdr, ent, labels, sc, ln = dec.decode_compact(x, s)
        out, ts = dec.expand_compact(hdr, ent, labels, T)
        e = (out.cpu(), sc.cpu(), ts.cpu(), ln.cpu())
        for other, name in ((b, "padded"), (c, "device input"), (d, "decode_device"), (e, "compact + device expansion")):
            for i in range(4):
                assert torch_mod.equal(a[i], other[i]), "%s differs in tensor %d (case %d)" % (name, i, seed)
        assert int(labels.numel()) <= int(ln.sum())
        want = ou.decode(lp, sl, beam=K)
        got = dict(tokens=a[0].numpy(), scores=a[1].numpy(), timesteps=a[2].numpy(), lens=a[3].numpy())
        ou.assert_same(_with_nres(got, want), want, "delivery case %d" % seed)


def test_logits_input_prepass(torch_mod):
    """Raw-logit input (logits_input=True / log_input == 2, an extension): the device's float32 log_softmax is defined to
    the bit (include/ctcdecode_amd.h ctcd_log_softmax) -- checked against its host twin -- and decoding logits equals the
    reference decoding those log-probabilities."""
    import ctcdecode_amd

    torch = torch_mod
    rng = np.random.default_rng(5)
    for V in (1, 2, 29, 64, 65, 130, 1000, 260, 1024, 2052, 4096, 10240, 16384, 16388):  # (> 256 and a multiple of 4: one workgroup per row)
        x = (rng.standard_normal((3, 17, V)) * 4).astype(np.float32)
        x[0, 1] -= 200.0
        x[1, 2, ::3] = -np.inf
        x[2, 3] = -np.inf
        x[2, 4] *= 40.0  # exp(x - max) underflows for most labels
        dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], device="cuda:0")
        got = dec.log_softmax(torch.from_numpy(x)).cpu().numpy()
        want = ou.log_softmax_rows(x)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "V=%d" % V
        sl = np.array([17, 5, 0], np.int32)
        got = dec.log_softmax(torch.from_numpy(x), torch.from_numpy(sl)).cpu().numpy()
        for b in range(3):
            assert np.array_equal(got[b, :sl[b]].view(np.uint32), want[b, :sl[b]].view(np.uint32)) and not got[b, sl[b]:].any()
    for (B, T, V, K, kw) in [(4, 150, 29, 100, {}), (3, 60, 9, 16, dict(cutoff_top_n=4)), (2, 40, 300, 25, dict(cutoff_top_n=40, cutoff_prob=0.9))]:
        logits = (rng.standard_normal((B, T, V)) * 2.5).astype(np.float32)
        sl = np.array([T, T // 2, 0, 7][:B], np.int32)
        dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], beam_width=K, logits_input=True, device="cuda:0", **kw)
        out, sc, ts, ln = dec.decode(torch.from_numpy(logits), torch.from_numpy(sl))
        got = dict(tokens=out.numpy(), timesteps=ts.numpy(), scores=sc.numpy(), lens=ln.numpy())
        want = ou.decode(ou.log_softmax_rows(logits), sl, beam=K, **kw)
        ou.assert_same(_with_nres(got, want), want, "logits B%d T%d V%d" % (B, T, V))
        # device-resident logits through the HBM-to-HBM entry as well
        d2 = dec.decode_device(torch.from_numpy(logits).cuda(), torch.from_numpy(sl).cuda())
        assert np.array_equal(d2[1].cpu().numpy().view(np.uint32), sc.numpy().view(np.uint32))


def _logit_rows_for_prune(rng, T, V):
    """Frames that exercise every branch of the logits -> candidates pass: plain, quantised (ties in x), wide spread (distinct
    logits that round to one normalised value), plateaus at the cut, -inf entries, a frame without a finite logit, a far outlier."""
    x = (rng.standard_normal((T, V)) * 3).astype(np.float32)
    for t in range(T):
        kind = t % 9
        if kind == 1:
            x[t] = np.round(x[t] * 2) / 2                      # few hundred distinct values: ties everywhere
        elif kind == 2:
            x[t] = np.round(x[t])                                # coarser still
        elif kind == 3:
            x[t] += 60000.0                                      # a large offset: the logits are quantised to 2^-8, m is huge
        elif kind == 4:
            x[t] = x[t] * 1e-4 + 9.0                             # all within 1e-3: y ~ -log V, many x share a y
        elif kind == 5:
            x[t, ::3] = -np.inf
        elif kind == 6:
            x[t] = -np.inf if t % 2 == 0 else -7.25              # no finite logit / every value equal
        elif kind == 7:
            x[t, rng.integers(0, V)] += 200.0                    # exp(x - m) underflows everywhere else
        elif kind == 8:
            x[t] = -np.abs(x[t]) * 30                            # most terms below the -88 cutoff of expf
    return x


@pytest.mark.parametrize("V,top_n,cp", [(260, 40, 1.0), (1000, 40, 0.9), (2048, 64, 1.0), (5000, 40, 0.6), (10240, 40, 1.0), (10240, 40, 0.95),
                                        (12000, 7, 1.0), (16384, 40, 0.99)])
def test_fused_logits_prune_equals_log_softmax_then_prune(torch_mod, V, top_n, cp):
    """log_input == 2 in front of a vocabulary prune: ONE kernel reads the logits and emits the kept candidates (the normalised
    rows are never written).  Its per-frame output -- count, labels, values -- must equal, bit for bit, what ctcd_log_softmax
    followed by the separate prune produces (debug switch), tie frames and the cumulative cut included; the normalised rows'
    definition is checked against its host twin, and the decode against the oracle fed the host twin's rows."""
    import ctcdecode_amd
    import ctcdecode_amd._native as n

    torch = torch_mod
    rng = np.random.default_rng(V + top_n)
    B, T, K = 3, 27, 20
    logits = np.stack([_logit_rows_for_prune(rng, T, V) for _ in range(B)])
    sl = np.array([T, T - 5, 0], np.int32)
    kw = dict(beam_width=K, cutoff_top_n=top_n, cutoff_prob=cp, logits_input=True, device="cuda:0")
    stride = min(top_n, V)
    res = {}
    for fused in (True, False):
        dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], **kw)
        dec.set_fused_logits(fused)
        out, sc, ts, ln = dec.decode(torch.from_numpy(logits), torch.from_numpy(sl))
        cnt, lab, val = dec.last_prune_rows(B * T, stride)
        res[fused] = (out.numpy(), sc.numpy(), ts.numpy(), ln.numpy(), cnt, lab, val, n.lib.ctcd_last_prune_flagged_rows(dec._handle))
    a, b = res[True], res[False]
    assert np.array_equal(a[4], b[4]), "counts differ in frames %s" % np.nonzero(a[4] != b[4])[0][:10]
    for r in range(B * T):
        c = int(a[4][r])
        assert np.array_equal(a[5][r, :c], b[5][r, :c]), "labels of frame %d" % r
        assert np.array_equal(a[6][r, :c].view(np.uint32), b[6][r, :c].view(np.uint32)), "values of frame %d" % r
    for k in range(4):
        assert np.array_equal(a[k], b[k]), "decode output %d" % k
    assert a[7] > 0 and b[7] > 0  # tie frames went through the std::sort replay on both sides
    # (the fused pass may flag a few more frames than the two-pass form: its NaN / +inf / image-of-the-bound guards)
    want = ou.decode(ou.log_softmax_rows(logits), sl, beam=K, cutoff_top_n=top_n, cutoff_prob=cp,
                     which="reference" if ou.have_reference() else "restated")
    got = dict(tokens=a[0], scores=a[1], timesteps=a[2], lens=a[3])
    ou.assert_same(_with_nres(got, want), want, "fused logits V=%d" % V)


@pytest.mark.parametrize("V,top_n,cp,probs", [(260, 40, 1.0, False), (1000, 40, 0.9, True), (2048, 64, 1.0, False), (4096, 40, 0.6, False),
                                              (10000, 40, 0.99, False), (10240, 40, 0.95, True), (10240, 3, 1.0, False)])
def test_prune_row_in_registers_equals_two_sweeps(torch_mod, V, top_n, cp, probs):
    """Round 6: the workgroup prune pass keeps a row in registers between its two looks at it (rows of up to 10 240 labels) instead of
    reading it twice.  Per frame -- count, labels, values -- and the decode equal the two-sweep form of rounds 2-5 bit for bit: rows
    with ties (flagged, settled by the std::sort replay on both sides), NaN rows, probability and log-probability input, ragged
    lengths; the decode also equals the oracle."""
    import ctcdecode_amd
    import ctcdecode_amd._native as n

    torch = torch_mod
    rng = np.random.default_rng(7 * V + top_n)
    B, T, K = 3, 27, 20
    logits = np.stack([_logit_rows_for_prune(rng, T, V) for _ in range(B)])
    x = ou.log_softmax_rows(logits)
    x[1, 3, 5] = np.float32("nan")  # (defined here: a NaN sorts below every number)
    if probs:
        x = np.exp(x.astype(np.float64)).astype(np.float32)
    sl = np.array([T, T - 5, 0], np.int32)
    kw = dict(beam_width=K, cutoff_top_n=top_n, cutoff_prob=cp, log_probs_input=not probs, device="cuda:0")
    stride = min(top_n, V)
    res = {}
    for reg in (True, False):
        dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], **kw)
        dec.set_prune_registers(reg)
        out, sc, ts, ln = dec.decode(torch.from_numpy(x), torch.from_numpy(sl))
        cnt, lab, val = dec.last_prune_rows(B * T, stride)
        res[reg] = (out.numpy(), sc.numpy(), ts.numpy(), ln.numpy(), cnt, lab, val, n.lib.ctcd_last_prune_flagged_rows(dec._handle))
    a, b = res[True], res[False]
    assert np.array_equal(a[4], b[4]), "counts differ in frames %s" % np.nonzero(a[4] != b[4])[0][:10]
    for r in range(B * T):
        c = int(a[4][r])
        assert np.array_equal(a[5][r, :c], b[5][r, :c]), "labels of frame %d" % r
        assert np.array_equal(a[6][r, :c].view(np.uint32), b[6][r, :c].view(np.uint32)), "values of frame %d" % r
    for k in range(4):
        assert np.array_equal(a[k].view(np.uint32) if a[k].dtype == np.float32 else a[k], b[k].view(np.uint32) if b[k].dtype == np.float32 else b[k]), "decode output %d" % k
    assert a[7] == b[7] and a[7] > 0  # the same frames go through the std::sort replay
    xo = x.copy()
    xo[1, 3, 5] = 0.0 if probs else -np.inf  # (the oracle's comparators are undefined on a NaN; below every number = what the library defines)
    if not probs or top_n < V:
        want = ou.decode(xo, sl, beam=K, cutoff_top_n=top_n, cutoff_prob=cp, log_input=not probs, which="restated")
        got = dict(tokens=a[0], scores=a[1], timesteps=a[2], lens=a[3])
        ou.assert_same(_with_nres(got, want), want, "prune in registers V=%d" % V)


def test_fused_logits_nan_and_inf_rows(torch_mod):
    """Frames holding NaN or +inf: no order argument applies, the fused pass hands them to the replay -- same output as the
    two-pass form (whose treatment of NaN is the library's own definition: below every number)."""
    import ctcdecode_amd

    torch = torch_mod
    rng = np.random.default_rng(99)
    B, T, V, K = 2, 12, 1024, 10
    logits = (rng.standard_normal((B, T, V)) * 2).astype(np.float32)
    logits[0, 1, 5] = np.nan
    logits[0, 2, ::7] = np.nan
    logits[0, 3, 17] = np.inf
    logits[1, 4, 100:120] = np.inf
    logits[1, 5] = np.nan
    res = {}
    for fused in (True, False):
        dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], beam_width=K, cutoff_top_n=40, logits_input=True, device="cuda:0")
        dec.set_fused_logits(fused)
        out, sc, ts, ln = dec.decode(torch.from_numpy(logits))
        cnt, lab, val = dec.last_prune_rows(B * T, 40)
        res[fused] = (cnt, lab, out.numpy(), ln.numpy())
    assert np.array_equal(res[True][0], res[False][0])
    for r in range(B * T):
        c = int(res[True][0][r])
        assert np.array_equal(res[True][1][r, :c], res[False][1][r, :c]), r
    assert np.array_equal(res[True][2], res[False][2]) and np.array_equal(res[True][3], res[False][3])


def test_prune_tie_replay_patterns(torch_mod):
    """Frames whose kept labels depend entirely on what std::sort does with equal values (the device replays libstdc++'s
    introsort for them: workgroup-parallel Hoare partitions for the long ranges, stl_emul.h below that): constant rows,
    sorted rows, few distinct values, long rows -- against the reference's real std::sort."""
    import ctcdecode_amd
    import ctcdecode_amd._native as n

    which = "reference" if ou.have_reference() else "restated"
    rng = np.random.default_rng(17)
    for V, top_n in [(1000, 40), (5000, 40), (10000, 40), (260, 64), (40, 7)]:
        T = 24
        rows = np.empty((T, V), np.float32)
        for t in range(T):
            kind = t % 6
            if kind == 0:
                rows[t] = -3.0                                           # every value equal
            elif kind == 1:
                rows[t] = -np.arange(V, dtype=np.float32) // 7           # descending plateaus
            elif kind == 2:
                rows[t] = (np.arange(V, dtype=np.float32) // 5) - V      # ascending plateaus
            elif kind == 3:
                rows[t] = rng.integers(0, 3, V).astype(np.float32) - 4   # three distinct values
            elif kind == 4:
                rows[t] = np.where(np.arange(V) % 2 == 0, -1.0, -2.0)    # alternating
            else:
                rows[t] = np.round(rng.standard_normal(V) * 2) - 6       # quantised noise
        lp = np.stack([rows, rows[::-1].copy()])
        want = ou.decode(lp, beam=12, cutoff_top_n=top_n, which=which)
        dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], cutoff_top_n=top_n, beam_width=12, log_probs_input=True)
        out, sc, ts, ln = dec.decode(torch_mod.from_numpy(lp))
        got = dict(tokens=out.numpy(), timesteps=ts.numpy(), scores=sc.numpy(), lens=ln.numpy())
        ou.assert_same(_with_nres(got, want), want, "V=%d vs %s" % (V, which))
        assert n.lib.ctcd_last_prune_flagged_rows(dec._handle) > 0 and n.lib.ctcd_last_prune_host_rows(dec._handle) == 0


def test_streaming_with_logits_input(torch_mod):
    """The raw-logit pre-pass is per frame, so chunked feeding through the streaming API gives the one-shot result."""
    import ctcdecode_amd

    rng = np.random.default_rng(23)
    B, T, V, K = 3, 90, 29, 40
    logits = (rng.standard_normal((B, T, V)) * 2).astype(np.float32)
    want = ou.decode(ou.log_softmax_rows(logits), beam=K)
    dec = ctcdecode_amd.OnlineCTCBeamDecoder([str(i) for i in range(V)], beam_width=K, logits_input=True)
    states = [ctcdecode_amd.DecoderState(dec) for _ in range(B)]
    x = torch_mod.from_numpy(logits)
    bounds = [0, 1, 40, 40, 77, T]
    for i in range(len(bounds) - 1):
        out, sc, ts, ln = dec.decode(x[:, bounds[i]:bounds[i + 1]], states, [i == len(bounds) - 2] * B)
    L = out.shape[2]
    got = dict(tokens=np.zeros((B, K, T), np.int32), timesteps=np.zeros((B, K, T), np.int32), scores=sc.numpy(), lens=ln.numpy(), nres=want["nres"])
    got["tokens"][:, : out.shape[1], :L] = out.numpy()
    got["timesteps"][:, : out.shape[1], :L] = ts.numpy()
    ou.assert_same(got, want, "chunked logits")


def test_async_compact_decode_tickets(torch_mod):
    """decode_compact_async / finish_compact (the non-blocking form a pipelined caller uses: the next batch is queued on
    another decoder before the host looks at this one): same results as the blocking calls, in any completion order."""
    import ctcdecode_amd

    torch = torch_mod
    B, T, V, K = 5, 120, 29, 32
    decs = [ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], beam_width=K, log_probs_input=True, device="cuda:0") for _ in range(2)]
    lps = [ou.synth_logprobs(B, T, V, 7100 + i) for i in range(4)]
    sl = np.array([T, 3, 0, T - 1, 60], np.int32)
    tickets = []
    for i, lp in enumerate(lps):  # two batches in flight at any time
        if len(tickets) == 2:
            j, d, tk = tickets.pop(0)
            hdr, ent, labels, sc, ln = d.finish_compact(tk)
            out, ts = d.expand_compact(hdr, ent, labels, T)
            want = ou.decode(lps[j], sl, beam=K)
            got = dict(tokens=out.cpu().numpy(), timesteps=ts.cpu().numpy(), scores=sc.cpu().numpy(), lens=ln.cpu().numpy())
            ou.assert_same(_with_nres(got, want), want, "ticket %d" % j)
        d = decs[i % 2]
        tickets.append((i, d, d.decode_compact_async(torch.from_numpy(lp), torch.from_numpy(sl))))
    for j, d, tk in tickets:
        hdr, ent, labels, sc, ln = d.finish_compact(tk)
        out, ts = d.expand_compact(hdr, ent, labels, T)
        want = ou.decode(lps[j], sl, beam=K)
        got = dict(tokens=out.cpu().numpy(), timesteps=ts.cpu().numpy(), scores=sc.cpu().numpy(), lens=ln.cpu().numpy())
        ou.assert_same(_with_nres(got, want), want, "ticket %d" % j)


def test_subtree_search_build_is_chosen_by_beam_shape_and_changes_nothing(torch_mod):
    """The second build of the north-star class kernel (phase A1 settles four subtrees per wave: ctcd_set_subtree_search): same
    results bit for bit when forced on random, blank-dominated and tie-heavy rows; chosen automatically after a checked launch
    that saw chain-shaped beams (blank-dominated rows), dropped again after one that saw bushy ones (random rows)."""
    import ctcdecode_amd

    V, K = 29, 100
    labels = [str(i) for i in range(V)]
    rows = {"randn": ou.synth_logprobs(6, 300, V, 11), "blank": ou.synth_logprobs(6, 300, V, 12, blank_bias=6.0),
            "quant": ou.synth_logprobs(6, 200, V, 13, quant=0.5, blank_bias=2.0)}
    for name, lp in rows.items():
        x = torch_mod.from_numpy(lp).cuda()
        outs = []
        for mode in (0, 1):
            dec = ctcdecode_amd.CTCBeamDecoder(labels, beam_width=K, log_probs_input=True)
            dec.set_subtree_search(mode)
            outs.append([t.cpu().numpy() for t in dec.decode_device(x)])
            assert dec.last_subtree_search() == mode
        for a, b in zip(*outs):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), name
        want = ou.decode(lp, beam=K, which="restated")
        got = dict(tokens=outs[1][0], scores=outs[1][1], timesteps=outs[1][2], lens=outs[1][3])
        ou.assert_same(_with_nres(got, want), want, "subtree-search build, " + name)
    dec = ctcdecode_amd.CTCBeamDecoder(labels, beam_width=K, log_probs_input=True)
    xb, xr = torch_mod.from_numpy(rows["blank"]).cuda(), torch_mod.from_numpy(rows["randn"]).cuda()
    dec.decode_device(xb)
    assert dec.last_subtree_search() == 0          # nothing known yet
    dec.decode_device(xb)
    assert dec.last_subtree_search() == 1          # the first launch reported chains
    dec.decode_device(xr)
    assert dec.last_subtree_search() == 1
    dec.decode_device(xr)
    assert dec.last_subtree_search() == 0          # ... and that one bushes


def test_decode_pipeline_two_launches_in_flight(torch_mod):
    """ctcdecode_amd.DecodePipeline (a batch smaller than the CU count leaves CUs idle -- the next batch is
    launched beside it on a second stream, default build of the kernel): every batch's results equal the oracle's whatever the
    number of launches in flight, tickets may be collected late, a resubmitted slot's old ticket is refused, and the rule
    inflight_for() follows the CU count."""
    import ctcdecode_amd

    torch = torch_mod
    V, K, T = 29, 24, 90
    labels = [str(i) for i in range(V)]
    lps = [ou.synth_logprobs(3 + (i % 3), T, V, 9100 + i) for i in range(7)]
    wants = [ou.decode(lp, beam=K) for lp in lps]
    for inflight in (1, 2, 3):
        pipe = ctcdecode_amd.DecodePipeline(lambda: ctcdecode_amd.CTCBeamDecoder(labels, beam_width=K, log_probs_input=True, device="cuda:0"), inflight=inflight)
        tickets = [pipe.submit(torch.from_numpy(lp).cuda()) for lp in lps]
        # (tickets of a slot that has been resubmitted were finished -- their status checked -- when the slot was reused:
        #  their results stay valid and can still be fetched)
        for i, tk in enumerate(tickets):
            out, sc, ts, ln = pipe.result(tk)
            got = dict(tokens=out.cpu().numpy(), timesteps=ts.cpu().numpy(), scores=sc.cpu().numpy(), lens=ln.cpu().numpy())
            ou.assert_same(_with_nres(got, wants[i]), wants[i], "pipeline inflight=%d batch %d" % (inflight, i))
        pipe.drain()
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    assert ctcdecode_amd.DecodePipeline.inflight_for(ncu // 2) == 2 and ctcdecode_amd.DecodePipeline.inflight_for(ncu) == 1
    # a failed batch (a scorer callback that raises) fails ITS ticket -- every time it is asked -- and nothing else: the slot takes the
    # next batch, the neighbours' results are intact, drain() reports a failure nobody has collected
    if getattr(ctcdecode_amd, "HAVE_LM", False):
        lab4 = ["_", "a", "b", " "]
        state = {"boom": False}

        class Boom(Exception):
            pass

        def cb(words):
            if state["boom"]:
                raise Boom("no model today")
            return -1.0

        scs = [ctcdecode_amd.CallbackScorer(cb, ["a", "ab", "ba"], 2, lab4, alpha=1.0, beta=0.5) for _ in range(2)]
        it = iter(scs)
        pipe = ctcdecode_amd.DecodePipeline(lambda: ctcdecode_amd.CTCBeamDecoder(lab4, scorer=next(it), beam_width=8, log_probs_input=True, device="cuda:0"), inflight=2)
        xs = [torch.from_numpy(ou.synth_logprobs(2, 30, 4, 700 + i, blank_bias=0.5)).cuda() for i in range(4)]
        t0 = pipe.submit(xs[0])
        state["boom"] = True
        try:
            t1 = pipe.submit(xs[1])   # (a callback scorer decodes inside submit: the exception may surface here ...)
        except Boom:
            t1 = None
        state["boom"] = False
        t2 = pipe.submit(xs[2])
        t3 = pipe.submit(xs[3])       # the slot of the failed batch is usable
        if t1 is not None:            # (... or it is kept on the ticket)
            for _ in range(2):
                with pytest.raises(Boom):
                    pipe.result(t1)
        assert len(pipe.result(t0)) == 4 and len(pipe.result(t2)) == 4 and len(pipe.result(t3)) == 4
        pipe.drain()
