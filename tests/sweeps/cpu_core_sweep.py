"""CPU sweep: the host build of the product core (tests/native/core_host.cpp over beam_core.h) against the restated oracle on random
configurations for a time budget.  python tests/sweeps/cpu_core_sweep.py <seed> <seconds> [--degenerate [--reference]]
--degenerate: inputs with whole frames of -inf / overflowing sums (tests/degenerate_util.py), one-shot, pruned + ragged and
chunked in turn; --reference: compare against oracle/_ref (the reference's own sources) instead of the restatement."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle_util as ou
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 123)
t0 = time.time(); bad = 0; n = 0
stats = np.zeros(5, np.int64)
if "--degenerate" in sys.argv:
    import degenerate_util as du
    which = "reference" if "--reference" in sys.argv else "restated"
    while time.time() - t0 < float(sys.argv[2]):
        meta, lp = du.make_case(rng)
        kw = dict(beam=meta["K"], blank_id=meta["blank"])
        sl = None
        mode = n % 3
        if mode == 1:
            kw.update(cutoff_top_n=int(rng.choice([40, max(1, meta["V"] // 2), 3, 1])), cutoff_prob=float(rng.choice([1.0, 1.0, 0.9, 0.5])))
            sl = rng.integers(0, meta["T"] + 3, size=2).astype(np.int32) if n % 2 == 0 else None
        want = ou.decode(lp, sl, which=which, **kw)
        if mode == 2:
            bounds = sorted(set(int(v) for v in rng.integers(0, meta["T"] + 1, size=int(rng.integers(0, 6)))))
            got = ou.decode_core_host_chunked(lp, bounds, **kw)
        else:
            got = ou.decode_core_host(lp, sl, **kw)
        try:
            ou.assert_same(want, got, "x")
        except AssertionError:
            bad += 1
            print("MISMATCH", meta, kw, flush=True)
        n += 1
    print("done: %d degenerate configurations against the %s oracle, %d mismatches" % (n, which, bad))
    sys.exit(1 if bad else 0)
while time.time() - t0 < float(sys.argv[2]) if len(sys.argv) > 2 else 600:
    V = int(rng.choice([2, 3, 5, 9, 29, 29, 64, 100]))
    K = int(rng.choice([1, 2, 5, 16, 50, 100, 128, 200]))
    T = int(rng.integers(1, 260))
    quant = [None, None, 0.5, 1.0, 0.25, 2.0, 4.0][int(rng.integers(0, 7))]
    bias = float(rng.choice([0, 0, 3, 6, -2]))
    blank = int(rng.integers(0, V))
    top_n = int(rng.choice([40, 40, 40, max(1, V // 2), 3, 1]))
    cp = float(rng.choice([1.0, 1.0, 1.0, 0.9, 0.5]))
    seed = int(rng.integers(0, 1 << 30))
    lp = ou.synth_logprobs(2, T, V, seed, quant=quant, blank_bias=bias, blank_id=blank)
    sl = rng.integers(0, T + 5, size=2).astype(np.int32) if n % 3 == 0 else None
    kw = dict(beam=K, blank_id=blank, cutoff_top_n=top_n, cutoff_prob=cp)
    try:
        a = ou.decode(lp, sl, which="restated", want_stats=True, **kw)
        b = ou.decode_core_host(lp, sl, **kw)
        ou.assert_same(a, b, "x")
        stats += a["stats"].sum(0)
    except AssertionError as e:
        bad += 1
        print("MISMATCH", dict(V=V, K=K, T=T, quant=quant, bias=bias, blank=blank, top_n=top_n, cp=cp, seed=seed, sl=None if sl is None else sl.tolist()), flush=True)
    n += 1
print("done: %d configurations, %d mismatches, stats %s" % (n, bad, stats.tolist()))
