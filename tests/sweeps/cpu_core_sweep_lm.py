"""CPU sweep of the LM tier: host build of the product core against the restated oracle.
    python tests/sweeps/cpu_core_sweep_lm.py <seed> <seconds> [--degenerate] [--hook]
--hook: the same core behind the host-side scorer hook (a callback asking the built-in tables; unpruned rows) must equal the built-in path,
degenerate rows included (round 6); CTC_HOST_BIG=1|2|3 runs the hook on the wide-beam layouts."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle_util as ou
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'data')
LABELS29 = ["_", "'", " "] + [chr(ord("a") + i) for i in range(26)]
rng = np.random.default_rng(int(sys.argv[1]))
budget = float(sys.argv[2])
models = [("abcd_words.arpa", ["_", "a", "b", "c", "d", "'", " "]), ("chars.arpa", ["_", "a", "b", "c", "d", "'", "é", " "]), ("test.arpa", LABELS29)]
t0 = time.time(); bad = 0; n = 0
while time.time() - t0 < budget:
    arpa, labels = models[n % 3]
    V = len(labels)
    K = int(rng.choice([1, 2, 5, 16, 50, 100, 128]))
    T = int(rng.integers(1, 200))
    alpha, beta = float(rng.choice([0.0, 0.3, 1.0, 2.5])), float(rng.choice([-1.0, 0.0, 0.5, 1.5]))
    quant = [None, None, 0.5, 1.0][int(rng.integers(0, 4))]
    seed = int(rng.integers(0, 1 << 30))
    lp = ou.synth_logprobs(2, T, V, seed, quant=quant)
    lp[:, :, labels.index(" ")] += np.float32(rng.choice([0.0, 1.0, 2.0]))
    if "--degenerate" in sys.argv:  # whole frames of -inf, overflowing sums ... (tests/degenerate_util.py)
        import degenerate_util as du
        meta, lp = du.make_case(rng, V=V, labels_space=labels.index(" "))
        K, T, seed = meta["K"], meta["T"], (meta["seed"], meta["kind"])
    top_n = int(rng.choice([40, 40, 5]))
    sl = rng.integers(0, T + 3, size=2).astype(np.int32) if n % 5 == 0 else None
    kw = dict(seq_lens=sl, beam=K, cutoff_top_n=top_n, blank_id=0)
    path = os.path.join(DATA, arpa)
    try:
        if "--hook" in sys.argv:
            kw["cutoff_top_n"] = V
            a = ou.decode_core_host_lm(lp, alpha, beta, path, labels, threads=1, **kw)
            b = ou.decode_core_host_lm_cb(lp, alpha, beta, path, labels, seq_lens=sl, beam=K, blank_id=0)
            ou.assert_same(b, a, "x")
            n += 1
            continue
        sc = ou.Scorer(alpha, beta, path, labels, "restated")
        ou.assert_same(ou.decode_core_host_lm(lp, alpha, beta, path, labels, **kw), ou.decode(lp, scorer=sc, **kw), "x")
    except AssertionError:
        bad += 1
        print("MISMATCH", dict(arpa=arpa, K=K, T=T, alpha=alpha, beta=beta, quant=quant, seed=seed, top_n=top_n, sl=None if sl is None else sl.tolist()), flush=True)
    n += 1
print("done: %d LM configurations%s%s, %d mismatches" % (n, " behind the scorer hook" if "--hook" in sys.argv else "", " (degenerate rows)" if "--degenerate" in sys.argv else "", bad))
