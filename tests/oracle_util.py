"""ctypes front-ends for the CPU checkers under oracle/ (TEST INFRASTRUCTURE ONLY).

  * ``restated``  -> oracle/_build/libctcoracle.so  (this repo's CPU restatement, oracle/ctc_oracle.cpp)
  * ``reference`` -> oracle/_ref/libctcref.so       (the reference's own sources, built by oracle/Makefile)

Both expose the marshalling contract of ctcdecode/src/binding.cpp:55-99 behind a C ABI.
"""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RESTATED_SO = os.path.join(ROOT, "oracle", "_build", "libctcoracle.so")
REFERENCE_SO = os.path.join(ROOT, "oracle", "_ref", "libctcref.so")

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)


def have_reference():
    return os.path.exists(REFERENCE_SO)


def _ptr(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def _pack(strings):
    return b"".join(x.encode("utf-8") + b"\0" for x in strings)


class Scorer(object):
    """The LM scorer of one of the CPU checkers: ``restated`` = oracle/ctc_oracle.cpp's own, ``reference`` = the
    reference's scorer.cpp over the kenlm / OpenFST stand-ins of oracle/shim (binding.cpp:143-150,263-287)."""

    def __init__(self, alpha, beta, lm_path, labels, which="restated"):
        self.which, self.labels = which, list(labels)
        self.lib = ctypes.CDLL(RESTATED_SO if which == "restated" else REFERENCE_SO)
        self.pfx = "ctcoracle_scorer_" if which == "restated" else "ctcref_scorer_"
        f = getattr(self.lib, self.pfx + "create")
        f.restype = ctypes.c_void_p
        f.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
        self.handle = f(alpha, beta, os.fsencode(lm_path), _pack(self.labels), len(self.labels))
        if not self.handle:
            raise RuntimeError("could not load the language model %r" % (lm_path,))

    def _call(self, name, restype, *args, argtypes=()):
        f = getattr(self.lib, self.pfx + name)
        f.restype = restype
        f.argtypes = [ctypes.c_void_p] + list(argtypes)
        return f(self.handle, *args)

    def is_character_based(self):
        return bool(self._call("is_character_based", ctypes.c_int))

    def max_order(self):
        return self._call("max_order", ctypes.c_int)

    def dict_size(self):
        return self._call("dict_size", ctypes.c_int)

    def reset_params(self, alpha, beta):
        self._call("reset_params", None, alpha, beta, argtypes=[ctypes.c_double, ctypes.c_double])

    def cond_logprob(self, words):
        return self._call("cond_logprob", ctypes.c_double, _pack(words), len(words), argtypes=[ctypes.c_char_p, ctypes.c_int])

    def sent_logprob(self, words):
        return self._call("sent_logprob", ctypes.c_double, _pack(words), len(words), argtypes=[ctypes.c_char_p, ctypes.c_int])

    def __del__(self):
        if getattr(self, "handle", None):
            self._call("release", None)
            self.handle = None


def decode(probs, seq_lens=None, beam=100, cutoff_prob=1.0, cutoff_top_n=40, blank_id=0, log_input=True,
           threads=None, which="restated", want_stats=False, scorer=None):
    """Returns dict(tokens[B,K,T], timesteps[B,K,T], scores[B,K], lens[B,K], nres[B]) as numpy arrays.

    Unwritten positions are zero (the reference leaves them uninitialised, ctcdecode/__init__.py:83-86).
    """
    probs = np.ascontiguousarray(probs, dtype=np.float32)
    B, T, V = probs.shape
    if seq_lens is not None:
        seq_lens = np.ascontiguousarray(seq_lens, dtype=np.int32)
    threads = threads or os.cpu_count() or 1
    tok = np.zeros((B, beam, T), np.int32)
    ts = np.zeros((B, beam, T), np.int32)
    sc = np.zeros((B, beam), np.float32)
    ln = np.zeros((B, beam), np.int32)
    nres = np.zeros((B,), np.int32)
    if scorer is not None:  # paddle_beam_decode_lm (binding.cpp:122-140)
        assert scorer.which == which and len(scorer.labels) == V
        fn = scorer.lib.ctcoracle_decode_lm_f32 if which == "restated" else scorer.lib.ctcref_decode_lm_f32
        fn.argtypes = [_f32p, _i32p] + [ctypes.c_int] * 5 + [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p,
                                                             ctypes.c_void_p, _i32p, _i32p, _f32p, _i32p, _i32p]
        rc = fn(_ptr(probs, _f32p), _ptr(seq_lens, _i32p), B, T, V, beam, threads, cutoff_prob, cutoff_top_n, blank_id,
                int(bool(log_input)), _pack(scorer.labels), scorer.handle, _ptr(tok, _i32p), _ptr(ts, _i32p), _ptr(sc, _f32p),
                _ptr(ln, _i32p), _ptr(nres, _i32p))
        if rc != 1:
            raise RuntimeError("checker returned %d" % rc)
        return dict(tokens=tok, timesteps=ts, scores=sc, lens=ln, nres=nres)
    if which == "restated":
        lib = ctypes.CDLL(RESTATED_SO)
        stats = np.zeros((B, 5), np.int64) if want_stats else None
        rc = lib.ctcoracle_decode_f32(_ptr(probs, _f32p), _ptr(seq_lens, _i32p), B, T, V, beam, threads,
                                      ctypes.c_double(cutoff_prob), cutoff_top_n, blank_id, int(bool(log_input)),
                                      _ptr(tok, _i32p), _ptr(ts, _i32p), _ptr(sc, _f32p), _ptr(ln, _i32p),
                                      _ptr(nres, _i32p), _ptr(stats, _i64p))
    elif which == "reference":
        lib = ctypes.CDLL(REFERENCE_SO)
        stats = None
        rc = lib.ctcref_decode_f32(_ptr(probs, _f32p), _ptr(seq_lens, _i32p), B, T, V, beam, threads,
                                   ctypes.c_double(cutoff_prob), cutoff_top_n, blank_id, int(bool(log_input)),
                                   _ptr(tok, _i32p), _ptr(ts, _i32p), _ptr(sc, _f32p), _ptr(ln, _i32p),
                                   _ptr(nres, _i32p))
    else:
        raise ValueError(which)
    if rc != 1:
        raise RuntimeError("checker returned %d" % rc)
    out = dict(tokens=tok, timesteps=ts, scores=sc, lens=ln, nres=nres)
    if want_stats:
        out["stats"] = stats
    return out


def assert_same(a, b, what=""):
    """Bit-exact comparison of two result dicts on the region the reference defines."""
    assert np.array_equal(a["nres"], b["nres"]), what + " n_results differ"
    B, K = a["lens"].shape
    for bi in range(B):
        n = int(a["nres"][bi])
        assert np.array_equal(a["lens"][bi, :n], b["lens"][bi, :n]), "%s lens differ (item %d)" % (what, bi)
        sa = a["scores"][bi, :n].view(np.uint32)
        sb = b["scores"][bi, :n].view(np.uint32)
        assert np.array_equal(sa, sb), "%s scores differ bitwise (item %d): %s vs %s" % (
            what, bi, a["scores"][bi, :n][sa != sb][:4], b["scores"][bi, :n][sa != sb][:4])
        for p in range(n):
            L = int(a["lens"][bi, p])
            assert np.array_equal(a["tokens"][bi, p, :L], b["tokens"][bi, p, :L]), "%s tokens differ (item %d beam %d)" % (what, bi, p)
            assert np.array_equal(a["timesteps"][bi, p, :L], b["timesteps"][bi, p, :L]), "%s timesteps differ (item %d beam %d)" % (what, bi, p)


def synth_logprobs(B, T, V, seed, kind="randn", quant=None, blank_bias=0.0, blank_id=0):
    """Synthetic log-softmax inputs (numpy only, so fixtures are reproducible without torch).

    kind="randn": log_softmax of N(0,1) logits.  ``quant`` rounds the
    log-probs to multiples of ``quant`` -- this manufactures exact score ties, the hard case
    for parity.  ``blank_bias`` adds to the blank logit (blank-dominated posteriors).
    """
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, T, V)).astype(np.float32)
    x[:, :, blank_id] += np.float32(blank_bias)
    m = x.max(axis=-1, keepdims=True)
    lse = m + np.log(np.exp(x - m).sum(axis=-1, keepdims=True, dtype=np.float32), dtype=np.float32)
    lp = (x - lse).astype(np.float32)
    if quant:
        lp = (np.round(lp / np.float32(quant)) * np.float32(quant)).astype(np.float32)
    return lp


CORE_HOST_SO = os.path.join(ROOT, "oracle", "_build", "libctccore_host.so")


def build_core_host():
    """Compile the GPU decoder's per-utterance core for the host (sequential policy) -- test infrastructure only."""
    import subprocess

    src = os.path.join(ROOT, "tests", "native", "core_host.cpp")
    deps = [src] + [os.path.join(ROOT, "ctcdecode_amd", "csrc", f) for f in ("beam_core.h", "stl_emul.h", "exact_math.h", "exact_math_f64.h", "exact_math_f64_tables.h", "lm_tables.h", "lm_build.h", "compact_results.h")]
    if os.path.exists(CORE_HOST_SO) and all(os.path.getmtime(CORE_HOST_SO) >= os.path.getmtime(p) for p in deps):
        return CORE_HOST_SO
    os.makedirs(os.path.dirname(CORE_HOST_SO), exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mfma", "-DCTC_ASSUME_CHECKED"] + os.environ.get("CTC_HOST_EXTRA_FLAGS", "").split()
                   + [src, "-o", CORE_HOST_SO, "-lpthread"], check=True)
    return CORE_HOST_SO


def log_softmax_rows(x):
    """float32 log_softmax over the last axis exactly as the product's pre-pass defines it (host twin, C library expf/logf)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    lib = ctypes.CDLL(build_core_host())
    lib.ctccore_log_softmax_rows.restype = None
    lib.ctccore_log_softmax_rows.argtypes = [_f32p, ctypes.c_longlong, ctypes.c_int, _f32p]
    lib.ctccore_log_softmax_rows(_ptr(x, _f32p), x.size // x.shape[-1], x.shape[-1], _ptr(out, _f32p))
    return out


def decode_core_host(probs, seq_lens=None, beam=100, cutoff_prob=1.0, cutoff_top_n=40, blank_id=0, threads=None):
    probs = np.ascontiguousarray(probs, dtype=np.float32)
    B, T, V = probs.shape
    if seq_lens is not None:
        seq_lens = np.ascontiguousarray(seq_lens, dtype=np.int32)
    threads = threads or os.cpu_count() or 1
    tok = np.zeros((B, beam, T), np.int32)
    ts = np.zeros((B, beam, T), np.int32)
    sc = np.zeros((B, beam), np.float32)
    ln = np.zeros((B, beam), np.int32)
    nres = np.zeros((B,), np.int32)
    lib = ctypes.CDLL(build_core_host())
    rc = lib.ctccore_decode_f32(_ptr(probs, _f32p), _ptr(seq_lens, _i32p), B, T, V, beam, threads, ctypes.c_double(cutoff_prob),
                                cutoff_top_n, blank_id, _ptr(tok, _i32p), _ptr(ts, _i32p), _ptr(sc, _f32p), _ptr(ln, _i32p), _ptr(nres, _i32p))
    if rc != 1:
        raise RuntimeError("core returned %d" % rc)
    return dict(tokens=tok, timesteps=ts, scores=sc, lens=ln, nres=nres)


def decode_core_host_lm(probs, alpha, beta, lm_path, labels, seq_lens=None, beam=100, cutoff_prob=1.0, cutoff_top_n=40, blank_id=0,
                        log_input=True, threads=None):
    """The host build of the product's core with the LM tier (scorer built by ctcdecode_amd/csrc/lm_build.h)."""
    probs = np.ascontiguousarray(probs, dtype=np.float32)
    B, T, V = probs.shape
    if seq_lens is not None:
        seq_lens = np.ascontiguousarray(seq_lens, dtype=np.int32)
    threads = threads or os.cpu_count() or 1
    tok = np.zeros((B, beam, T), np.int32)
    ts = np.zeros((B, beam, T), np.int32)
    sc = np.zeros((B, beam), np.float32)
    ln = np.zeros((B, beam), np.int32)
    nres = np.zeros((B,), np.int32)
    meta = np.zeros((3,), np.int32)
    lib = ctypes.CDLL(build_core_host())
    fn = lib.ctccore_decode_lm_f32
    fn.argtypes = [_f32p, _i32p] + [ctypes.c_int] * 5 + [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                                         ctypes.c_char_p, ctypes.c_char_p, _i32p, _i32p, _f32p, _i32p, _i32p, _i32p]
    rc = fn(_ptr(probs, _f32p), _ptr(seq_lens, _i32p), B, T, V, beam, threads, cutoff_prob, cutoff_top_n, blank_id, int(bool(log_input)),
            alpha, beta, os.fsencode(lm_path), _pack(labels), _ptr(tok, _i32p), _ptr(ts, _i32p), _ptr(sc, _f32p), _ptr(ln, _i32p),
            _ptr(nres, _i32p), _ptr(meta, _i32p))
    if rc != 1:
        raise RuntimeError("core returned %d" % rc)
    return dict(tokens=tok, timesteps=ts, scores=sc, lens=ln, nres=nres, meta=tuple(int(v) for v in meta))


def decode_core_host_lm_cb(probs, alpha, beta, lm_path, labels, seq_lens=None, beam=100, blank_id=0, log_input=True):
    """The host-side scorer hook on the host build of the core: the decode asks a CACHE of a callback's answers, parks an
    utterance when the cache misses and resumes it once the callback (here: the built-in ARPA tables) has answered."""
    probs = np.ascontiguousarray(probs, dtype=np.float32)
    B, T, V = probs.shape
    if seq_lens is not None:
        seq_lens = np.ascontiguousarray(seq_lens, dtype=np.int32)
    tok = np.zeros((B, beam, T), np.int32)
    ts = np.zeros((B, beam, T), np.int32)
    sc = np.zeros((B, beam), np.float32)
    ln = np.zeros((B, beam), np.int32)
    nres = np.zeros((B,), np.int32)
    stats = np.zeros((2,), np.int64)
    lib = ctypes.CDLL(build_core_host())
    rc = lib.ctccore_decode_lm_cb_f32(_ptr(probs, _f32p), _ptr(seq_lens, _i32p) if seq_lens is not None else None, B, T, V, beam, blank_id,
                                      1 if log_input else 0, ctypes.c_double(alpha), ctypes.c_double(beta), lm_path.encode(), _pack(labels),
                                      _ptr(tok, _i32p), _ptr(ts, _i32p), _ptr(sc, _f32p), _ptr(ln, _i32p), _ptr(nres, _i32p),
                                      stats.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)))
    if rc != 1:
        raise RuntimeError("ctccore_decode_lm_cb_f32 failed: %d" % rc)
    return dict(tokens=tok, timesteps=ts, scores=sc, lens=ln, nres=nres, callback_calls=int(stats[0]), resumptions=int(stats[1]))


def decode_core_host_compact(probs, seq_lens=None, beam=100, blank_id=0):
    """The host build of the core writing COMPACT results, expanded by the product's host expansion (compact_results.h)."""
    probs = np.ascontiguousarray(probs, dtype=np.float32)
    B, T, V = probs.shape
    if seq_lens is not None:
        seq_lens = np.ascontiguousarray(seq_lens, dtype=np.int32)
    tok = np.full((B, beam, T), -9, np.int32)
    ts = np.full((B, beam, T), -9, np.int32)
    sc = np.zeros((B, beam), np.float32)
    ln = np.zeros((B, beam), np.int32)
    nres = np.zeros((B,), np.int32)
    used = ctypes.c_longlong(0)
    lib = ctypes.CDLL(build_core_host())
    rc = lib.ctccore_decode_compact_f32(_ptr(probs, _f32p), _ptr(seq_lens, _i32p), B, T, V, beam, blank_id, _ptr(tok, _i32p), _ptr(ts, _i32p),
                                        _ptr(sc, _f32p), _ptr(ln, _i32p), _ptr(nres, _i32p), ctypes.byref(used))
    if rc != 1:
        raise RuntimeError("core returned %d" % rc)
    return dict(tokens=tok, timesteps=ts, scores=sc, lens=ln, nres=nres, labels_used=int(used.value))


def core_host_lm_cond(lm_path, labels, words):
    """Scorer::get_log_cond_prob evaluated by the product's own tables (host copy, ctcdecode_amd/csrc/lm_build.h)."""
    lib = ctypes.CDLL(build_core_host())
    fn = lib.ctccore_lm_cond
    fn.restype = ctypes.c_double
    fn.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, _i32p]
    meta = np.zeros((3,), np.int32)
    v = fn(os.fsencode(lm_path), _pack(labels), len(labels), _pack(words), len(words), _ptr(meta, _i32p))
    return v, tuple(int(x) for x in meta)


def decode_core_host_chunked(probs, bounds, beam=100, blank_id=0):
    """The host build of the core fed chunk by chunk (bounds = frame boundaries 0 < ... < T) through its stream state."""
    probs = np.ascontiguousarray(probs, dtype=np.float32)
    B, T, V = probs.shape
    bounds = np.ascontiguousarray([0] + [int(x) for x in bounds] + [T], dtype=np.int32)
    tok = np.zeros((B, beam, T), np.int32)
    ts = np.zeros((B, beam, T), np.int32)
    sc = np.zeros((B, beam), np.float32)
    ln = np.zeros((B, beam), np.int32)
    nres = np.zeros((B,), np.int32)
    lib = ctypes.CDLL(build_core_host())
    rc = lib.ctccore_decode_chunked_f32(_ptr(probs, _f32p), B, T, V, beam, blank_id, _ptr(bounds, _i32p), len(bounds) - 1,
                                        _ptr(tok, _i32p), _ptr(ts, _i32p), _ptr(sc, _f32p), _ptr(ln, _i32p), _ptr(nres, _i32p))
    if rc != 1:
        raise RuntimeError("core returned %d" % rc)
    return dict(tokens=tok, timesteps=ts, scores=sc, lens=ln, nres=nres)
