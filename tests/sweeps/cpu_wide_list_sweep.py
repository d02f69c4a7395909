"""CPU sweep aimed at the select's wide-list path (round 6: a crowded bucket -- more than 128 keys -- is listed into the next beam's block
and ranked there): beams of 30 ... 600 entries over 29 / 64 / 100 labels, unquantised and quantised rows (quantised rows crowd every
bucket), ragged lengths, against the reference build (oracle/_ref) where present.  The host build's event counter says how many frames
took the path.  The path is an experiment switch (beam_core.h CTC_EXP_WIDE_LIST: exact, measured slower on the GPU, off by default):
build the host core with it -- CTC_HOST_EXTRA_FLAGS=-DCTC_EXP_WIDE_LIST -- or the sweep is an ordinary wide-beam sweep.
    python tests/sweeps/cpu_wide_list_sweep.py <seed> <seconds>      (CTC_HOST_BIG=1|2|3 / CTC_HOST_FARREP=1: the wide-beam layouts)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle_util as ou
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 300
which = "reference" if ou.have_reference() else "restated"
lib = ctypes.CDLL(ou.build_core_host())
cnt = (ctypes.c_longlong * 32)()
lib.ctccore_event_counts(cnt, 1)
t0 = time.time(); n = bad = 0
while time.time() - t0 < budget:
    V = int(rng.choice([29, 29, 64, 100]))
    K = int(rng.choice([30, 60, 129, 200, 300, 500, 600]))
    T = int(rng.integers(5, 70))
    quant = [None, None, None, 0.5, 0.25, 1.0, 0.0625][int(rng.integers(0, 7))]
    bias = float(rng.choice([0, 0, 3, -2]))
    seed = int(rng.integers(0, 1 << 30))
    lp = ou.synth_logprobs(2, T, V, seed, quant=quant, blank_bias=bias)
    if n % 4 == 1:  # scores near -2000: float32 keys 1.2e-4 apart, the collisions of long utterances
        lp[:, 0, :] -= np.float32(1900.0)
    sl = rng.integers(0, T + 3, size=2).astype(np.int32) if n % 3 == 0 else None
    kw = dict(beam=K, cutoff_top_n=V)
    a = ou.decode(lp, sl, which=which, **kw)
    b = ou.decode_core_host(lp, sl, threads=1, **kw)
    try:
        ou.assert_same(a, b, "x")
    except AssertionError:
        bad += 1
        print("MISMATCH", dict(V=V, K=K, T=T, quant=quant, bias=bias, seed=seed, n=n, sl=None if sl is None else sl.tolist()), flush=True)
    n += 1
k = lib.ctccore_event_counts(cnt, 0)
print("done: %d configurations against the %s oracle, %d mismatches; frames %d, selects on the fast path %d, of them through the wide list %d, slow-path rounds %d, exact replays %d"
      % (n, which, bad, cnt[0], cnt[11], cnt[26], cnt[22], cnt[2]))
sys.exit(1 if bad else 0)
