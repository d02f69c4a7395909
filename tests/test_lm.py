"""LM tier on the CPU: pins the restated scorer + LM-mode decoder of oracle/ctc_oracle.cpp.

(a) kenlm's own published expectations for tests/test.arpa (lm/model_test.cc of github.com/kpu/kenlm: Starters,
    Continuation, Blanks) -- the only pin of the LM ARITHMETIC that does not go through a restatement of kenlm;
(b) the reference's golden "a a" (tests/test_decode.py:55-64 of the reference: test.arpa, alpha = beta = 0);
(c) live differential against oracle/_ref = the reference's own scorer.cpp / decoder LM hooks (compiled unmodified) over
    the kenlm / OpenFST stand-ins of oracle/shim, on word and character models with non-zero alpha and beta;
(d) committed fixtures generated from (c) (tests/golden_lm, tests/golden/make_golden_lm.py) so that the pin travels.
"""
import os

import numpy as np
import pytest

import golden_util as gu
import oracle_util as ou

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
TEST_ARPA = os.path.join(DATA, "test.arpa")
VOCAB7 = ["'", " ", "a", "b", "c", "d", "_"]
LOG10E = float(np.float32(0.4342944819))
WHICH = ["restated"] + (["reference"] if ou.have_reference() else [])

# (words fed after the null context, log10 p(last word | the ones before)); "<s>" first = kenlm's BeginSentenceState
KENLM_KATS = [
    (["<s>", "looking"], -0.4846522),
    (["<s>", ","], -1.383514 + -0.4149733),
    (["<s>", "this_is_not_found"], None),  # OOV: the Scorer returns OOV_SCORE before kenlm is asked (scorer.cpp:83-85)
    (["<s>", "looking", "on"], -0.348837),
    (["<s>", "looking", "on", "a"], -0.0155266),
    (["<s>", "looking", "on", "a", "little"], -0.00306122),
    (["<s>", "looking", "on", "a", "little", "the"], -4.04005),
    (["<s>", "looking", "on", "a", "little", "the", "biarritz"], -1.9889),
    (["<s>", "looking", "on", "a", "little", "more"], -0.00181395),
    (["<s>", "looking", "on", "a", "little", "more", "loin"], -0.0432557),
    (["also"], -1.687872),
    (["also", "would"], -2.0),
    (["also", "would", "consider"], -3.0),          # listed although its suffix "would consider" is not
    (["also", "would", "consider", "higher"], -4.0),
    (["also", "would", "consider", "higher", "looking"], -5.0),
    (["higher"], -1.509559),
    (["higher", "looking"], -1.285941 - 0.30103),
    (["higher", "looking", "consider"], -1.687872 - 0.4771212),
    (["would"], -1.687872),
    (["would", "consider"], -1.687872 - 0.30103),
    (["would", "consider", "higher"], -1.509559 - 0.30103),
    (["would", "consider", "higher", "looking"], -1.285941 - 0.30103),
    (["more", "."], -0.51363),
    (["more", ".", "</s>"], -0.0191651),
]


@pytest.mark.parametrize("which", WHICH)
def test_kenlm_published_values(which):
    sc = ou.Scorer(0.0, 0.0, TEST_ARPA, VOCAB7, which)
    assert not sc.is_character_based() and sc.max_order() == 5
    for words, want in KENLM_KATS:
        got = sc.cond_logprob(words)
        if want is None:
            assert got == -1000.0
        else:
            assert abs(got * LOG10E - want) <= 2e-5 * max(1.0, abs(want)), (words, got * LOG10E, want)
    # get_sent_log_prob (scorer.cpp:95-120): <s> padding to the model order, </s> appended, one window per word
    s = sc.sent_logprob(["looking", "on", "a", "little"])
    want = (-0.4846522 - 0.3488368 - 0.01552657 - 0.003061223) + (-1.029493 - 0.4771212 - 0.69897 - 0.69897 - 0.4771212)  # </s> backs off four times
    assert abs(s * LOG10E - want) < 1e-4
    assert sc.sent_logprob(["not_a_word"]) <= -1000.0


def test_kenlm_published_values_product_tables():
    """The same expectations against the PRODUCT's scorer tables (back-off automaton of lm_tables.h, built by lm_build.h)."""
    for words, want in KENLM_KATS:
        got, meta = ou.core_host_lm_cond(TEST_ARPA, VOCAB7, words)
        assert meta == (0, 5, 1)
        if want is None:
            assert got == -1000.0
        else:
            assert abs(got * LOG10E - want) <= 2e-5 * max(1.0, abs(want)), (words, got * LOG10E, want)
    ref = ou.Scorer(0.0, 0.0, TEST_ARPA, VOCAB7, "restated")
    rng = np.random.default_rng(3)
    vocab = ["also", "would", "consider", "higher", "looking", "on", "a", "little", "more", "loin", ".", ",", "<s>", "</s>", "the", "foo", "bar", "baz", "nope"]
    for _ in range(400):  # bitwise agreement with the restated oracle on random word sequences (incl. unknown words)
        words = [vocab[i] for i in rng.integers(0, len(vocab), size=int(rng.integers(1, 8)))]
        got, _ = ou.core_host_lm_cond(TEST_ARPA, VOCAB7, words)
        assert got == ref.cond_logprob(words), words


def _strings(r, vocab, b, n=3):
    return ["".join(vocab[x] for x in r["tokens"][b, p, : r["lens"][b, p]]) for p in range(n)]


@pytest.mark.parametrize("which", WHICH)
def test_reference_golden_a_a(which):
    """tests/test_decode.py:55-64 of the reference: probs_seq2 with test.arpa -> "a a" (dictionary gate; alpha = beta = 0)."""
    args, _ = gu.load("ref_fixtures_prob")
    sc = ou.Scorer(0.0, 0.0, TEST_ARPA, VOCAB7, which)
    assert sc.dict_size() == 1  # of the model's words only "a" can be spelled with these labels
    r = ou.decode(scorer=sc, which=which, **args)
    assert _strings(r, VOCAB7, 1, 1) == ["a a"] and _strings(r, VOCAB7, 0, 1) == ["a a"]


LABELS29 = ["_", "'", " "] + [chr(ord("a") + i) for i in range(26)]  # blank first
LM_CASES = [
    dict(name="testarpa_cfg4", arpa="test.arpa", labels=LABELS29, alpha=0.5, beta=1.0, B=3, T=120, K=32, seed=101, bias={" ": 1.5, "a": 1.0}),
    dict(name="testarpa_bigbeam", arpa="test.arpa", labels=LABELS29, alpha=1.25, beta=0.3, B=2, T=200, K=100, seed=102, bias={" ": 2.0}),
    dict(name="abcd_words", arpa="abcd_words.arpa", labels=["_", "a", "b", "c", "d", "'", " "], alpha=0.7, beta=0.9, B=3, T=150, K=40, seed=103, bias={" ": 0.5}),
    dict(name="abcd_words_negbeta", arpa="abcd_words.arpa", labels=["a", "b", " ", "c", "d", "'", "_"], blank=6, alpha=2.0, beta=-0.5, B=2, T=120, K=16, seed=104),
    dict(name="abcd_words_full", arpa="abcd_words.arpa", labels=["_", "a", "b", "c", "d", "'", " "], alpha=0.4, beta=2.0, B=2, T=100, K=3, seed=105, quant=0.5),
    dict(name="chars", arpa="chars.arpa", labels=["_", "a", "b", "c", "d", "'", "é", " "], alpha=0.6, beta=0.2, B=3, T=100, K=24, seed=106),
    dict(name="chars_prob_input", arpa="chars.arpa", labels=["_", "a", "b", "c", "d", "'", "é"], alpha=1.0, beta=0.0, B=2, T=80, K=50, seed=107, prob_input=True),
    dict(name="abcd_topn", arpa="abcd_words.arpa", labels=["_", "a", "b", "c", "d", "'", " "], alpha=0.7, beta=0.9, B=2, T=100, K=20, seed=108, top_n=4),
    dict(name="testarpa_ragged", arpa="test.arpa", labels=LABELS29, alpha=0.5, beta=1.0, B=4, T=60, K=12, seed=109, ragged=True, bias={" ": 1.5}),
]


def lm_case_inputs(c):
    V = len(c["labels"])
    blank = c.get("blank", 0)
    lp = ou.synth_logprobs(c["B"], c["T"], V, c["seed"], quant=c.get("quant"), blank_id=blank)
    if c.get("bias"):  # favour some labels so that words of the model actually complete
        x = lp.copy()
        for ch, v in c["bias"].items():
            x[:, :, c["labels"].index(ch)] += np.float32(v)
        m = x.max(-1, keepdims=True)
        lp = (x - (m + np.log(np.exp(x - m).sum(-1, keepdims=True)))).astype(np.float32)
    x = np.exp(lp) if c.get("prob_input") else lp
    sl = np.array([c["T"], 0, 1, c["T"] // 2][: c["B"]], np.int32) if c.get("ragged") else None
    kw = dict(seq_lens=sl, beam=c["K"], cutoff_top_n=c.get("top_n", 40), blank_id=blank, log_input=not c.get("prob_input"))
    return np.ascontiguousarray(x), kw


@pytest.mark.skipif(not ou.have_reference(), reason="oracle/_ref not built (needs the reference checkout)")
@pytest.mark.parametrize("c", LM_CASES, ids=lambda c: c["name"])
def test_lm_live_differential(c):
    x, kw = lm_case_inputs(c)
    res = {}
    for which in ("restated", "reference"):
        sc = ou.Scorer(c["alpha"], c["beta"], os.path.join(DATA, c["arpa"]), c["labels"], which)
        res[which] = ou.decode(x, scorer=sc, which=which, **kw)
        res[which + "_meta"] = (sc.is_character_based(), sc.max_order(), sc.dict_size())
    assert res["restated_meta"] == res["reference_meta"]
    ou.assert_same(res["restated"], res["reference"], c["name"])
    # the scorer must matter: without it the result differs
    plain = ou.decode(x, which="restated", **kw)
    assert not np.array_equal(plain["tokens"], res["restated"]["tokens"])


@pytest.mark.parametrize("c", LM_CASES, ids=lambda c: c["name"])
def test_product_core_with_lm_matches_oracle(c):
    """ctcdecode_amd/csrc/beam_core.h (LM instantiation) + lm_tables.h, host build, against the restated oracle."""
    x, kw = lm_case_inputs(c)
    path = os.path.join(DATA, c["arpa"])
    sc = ou.Scorer(c["alpha"], c["beta"], path, c["labels"], "restated")
    want = ou.decode(x, scorer=sc, **kw)
    got = ou.decode_core_host_lm(x, c["alpha"], c["beta"], path, c["labels"], **kw)
    assert got["meta"] == (int(sc.is_character_based()), sc.max_order(), sc.dict_size())
    ou.assert_same(got, want, c["name"])


def test_product_core_with_lm_random_sweep():
    rng = np.random.default_rng(11)
    models = [("abcd_words.arpa", ["_", "a", "b", "c", "d", "'", " "]), ("chars.arpa", ["_", "a", "b", "c", "d", "'", "é", " "]),
              ("test.arpa", LABELS29)]
    for it in range(60):
        arpa, labels = models[it % 3]
        V = len(labels)
        K = int(rng.choice([1, 2, 5, 16, 50, 100]))
        T = int(rng.integers(1, 120))
        alpha, beta = float(rng.choice([0.0, 0.3, 1.0, 2.5])), float(rng.choice([-1.0, 0.0, 0.5, 1.5]))
        quant = [None, None, 0.5, 1.0][int(rng.integers(0, 4))]
        lp = ou.synth_logprobs(2, T, V, 7000 + it, quant=quant)
        if " " in labels:
            lp[:, :, labels.index(" ")] += np.float32(rng.choice([0.0, 1.0, 2.0]))
        top_n = int(rng.choice([40, 40, 5]))
        sl = rng.integers(0, T + 3, size=2).astype(np.int32) if it % 5 == 0 else None
        kw = dict(seq_lens=sl, beam=K, cutoff_top_n=top_n, blank_id=0)
        path = os.path.join(DATA, arpa)
        sc = ou.Scorer(alpha, beta, path, labels, "restated")
        ou.assert_same(ou.decode_core_host_lm(lp, alpha, beta, path, labels, **kw), ou.decode(lp, scorer=sc, **kw),
                       "case %d %s K=%d T=%d alpha=%g beta=%g q=%s" % (it, arpa, K, T, alpha, beta, quant))


def test_product_core_with_lm_degenerate_inputs():
    """Whole frames of -inf / overflowing sums with a scorer: the order of a prefix's two contributions then follows the
    frame's std::sort (ctc_beam_search_decoder.cpp:75-76) on top of std::nth_element's permutation.  The restated oracle
    is checked against the live reference on every third case."""
    import degenerate_util as du

    rng = np.random.default_rng(12)
    models = [("abcd_words.arpa", ["_", "a", "b", "c", "d", "'", " "]), ("chars.arpa", ["_", "a", "b", "c", "d", "'", "é", " "]),
              ("test.arpa", LABELS29)]
    for it in range(150):
        arpa, labels = models[it % 3]
        meta, lp = du.make_case(rng, V=len(labels), labels_space=labels.index(" "))
        alpha, beta = float(rng.choice([0.0, 0.3, 1.0, 2.5])), float(rng.choice([-1.0, 0.0, 0.5, 1.5]))
        kw = dict(beam=meta["K"], cutoff_top_n=int(rng.choice([40, 40, 5])), blank_id=0)
        path = os.path.join(DATA, arpa)
        sc = ou.Scorer(alpha, beta, path, labels, "restated")
        want = ou.decode(lp, scorer=sc, **kw)
        what = "case %d %s %s alpha=%g beta=%g" % (it, arpa, meta, alpha, beta)
        ou.assert_same(ou.decode_core_host_lm(lp, alpha, beta, path, labels, **kw), want, what)
        if it % 3 == 0 and ou.have_reference():
            ref = ou.Scorer(alpha, beta, path, labels, "reference")
            ou.assert_same(ou.decode(lp, scorer=ref, which="reference", **kw), want, what + " (oracle vs live reference)")


def make_mid_lm(tmp_path_factory_or_dir, words=3000, grams=20000):
    """A generated 3-gram word model of a few thousand words (tools/make_big_lm.py): far more dictionary nodes and n-gram
    table slots than the committed models, so that table collisions, long probe sequences and deep dictionary paths occur."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_big_lm", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "make_big_lm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    path = os.path.join(str(tmp_path_factory_or_dir), "mid_words_%d.arpa" % words)
    if not os.path.exists(path):
        mod.make(path, words, grams, grams, seed=3)
    return path


def test_generated_mid_size_model(tmp_path):
    path = make_mid_lm(tmp_path)
    checkers = [ou.Scorer(0.6, 0.8, path, LABELS29, w) for w in (("restated", "reference") if ou.have_reference() else ("restated",))]
    assert checkers[0].dict_size() == 3000 and checkers[0].max_order() == 3
    rng = np.random.default_rng(5)
    vocab = [ln.split("\t")[1] for ln in open(path, encoding="utf-8").read().split("\\2-grams:")[0].splitlines() if ln.count("\t") == 2][3:]
    for _ in range(200):
        words = [vocab[int(i)] if rng.random() > 0.1 else "zzzq" for i in rng.integers(0, min(len(vocab), 400), size=int(rng.integers(1, 4)))]
        got, meta = ou.core_host_lm_cond(path, LABELS29, words)
        for sc in checkers:
            assert got == sc.cond_logprob(words), (words, sc.which)
    for seed, K in ((61, 32), (62, 100)):
        lp = ou.synth_logprobs(2, 150, 29, seed)
        lp[:, :, LABELS29.index(" ")] += np.float32(1.5)
        for ch in "etao":
            lp[:, :, LABELS29.index(ch)] += np.float32(0.7)
        want = ou.decode(lp, scorer=checkers[0], beam=K)
        ou.assert_same(ou.decode_core_host_lm(lp, 0.6, 0.8, path, LABELS29, beam=K), want, "mid-size model, product core")
        if len(checkers) > 1:
            ou.assert_same(ou.decode(lp, scorer=checkers[1], which="reference", beam=K), want, "mid-size model, live reference")
        assert int(want["lens"][:, 0].min()) > 5


def make_wide_label_lm(directory):
    """A word model over 99 labels (Latin, Greek, Cyrillic and Hebrew letters + blank + space): more labels than the 64-bit
    arc mask of a dictionary node holds -- the product then keeps sorted arc lists (lm_tables.h dict_find_wide).
    -> (path, labels)"""
    syms = [chr(c) for c in list(range(ord("a"), ord("z") + 1)) + list(range(0x3B1, 0x3B1 + 25)) + list(range(0x430, 0x430 + 32)) + list(range(0x5D0, 0x5D0 + 14))]
    labels = ["_", " "] + syms
    path = os.path.join(str(directory), "wide_words.arpa")
    if not os.path.exists(path):
        rng = np.random.default_rng(7)
        words = set()
        while len(words) < 400:
            words.add("".join(syms[int(i)] for i in rng.integers(0, len(syms) if rng.random() < 0.5 else 12, size=int(rng.integers(1, 5)))))
        words = sorted(words)
        with open(path, "w", encoding="utf-8") as f:
            bi = sorted({(words[int(x)], words[int(y)]) for x, y in rng.integers(0, len(words), size=(600, 2))})
            f.write("\\data\\\nngram 1=%d\nngram 2=%d\n\n\\1-grams:\n" % (len(words) + 3, len(bi)))
            f.write("-2.5\t<unk>\t-0.2\n-99\t<s>\t-0.5\n-1.2\t</s>\n")
            for w in words:
                f.write("%.3f\t%s\t%.3f\n" % (-rng.random() * 3 - 0.1, w, -rng.random()))
            f.write("\n\\2-grams:\n")
            for x, y in bi:
                f.write("%.3f\t%s %s\n" % (-rng.random() * 2 - 0.1, x, y))
            f.write("\n\\end\\\n")
    return path, labels


def wide_label_cases():
    rng = np.random.default_rng(8)
    for it in range(12):
        yield it, int(rng.choice([5, 20, 60])), int(rng.integers(5, 50)), int(rng.choice([40, 40, 99]))


def wide_label_inputs(it, T, V):
    lp = ou.synth_logprobs(2, T, V, 100 + it)
    lp[:, :, 1] += np.float32(2.0)
    lp[:, :, 2:14] += np.float32(2.0)
    return lp


def test_word_model_over_more_than_64_labels(tmp_path):
    """path_trie.cpp:59-96 / scorer.cpp:196-230 put no limit on the number of labels of a word model; round 2 refused > 64."""
    path, labels = make_wide_label_lm(tmp_path)
    sc = ou.Scorer(0.7, 0.5, path, labels, "restated")
    for it, K, T, top_n in wide_label_cases():
        lp = wide_label_inputs(it, T, len(labels))
        kw = dict(beam=K, cutoff_top_n=top_n, blank_id=0)
        want = ou.decode(lp, scorer=sc, **kw)
        got = ou.decode_core_host_lm(lp, 0.7, 0.5, path, labels, **kw)
        assert got["meta"] == (0, 2, 400)
        ou.assert_same(got, want, "99 labels, case %d" % it)
        if it % 4 == 0 and ou.have_reference():
            ref = ou.Scorer(0.7, 0.5, path, labels, "reference")
            ou.assert_same(ou.decode(lp, scorer=ref, which="reference", **kw), want, "99 labels, oracle vs live reference")
    assert int(want["lens"][:, 0].max()) > 5


def test_unigram_model_keeps_no_context():
    """An order-1 ARPA model that lists back-off weights: kenlm's state for it has length 0, so no back-off is
    ever added -- the product's tables, the restated oracle and the reference's scorer.cpp over the kenlm stand-in agree,
    on single queries and on a decode."""
    path = os.path.join(DATA, "unigram_bo.arpa")
    labels = ["_", "a", "b", " "]
    checkers = [ou.Scorer(0.8, 0.4, path, labels, w) for w in (("restated", "reference") if ou.have_reference() else ("restated",))]
    for words in (["ab"], ["ab", "ba"], ["a", "b", "abba"], ["zz", "ab"], ["ab", "</s>"]):
        got, meta = ou.core_host_lm_cond(path, labels, words)
        assert meta[1] == 1
        for sc in checkers:
            assert got == sc.cond_logprob(words), (words, sc.which)
    assert got == -0.9 / np.log10(np.e) or abs(got - (-0.9 / 0.4342944819)) < 1e-5  # "</s>" alone: no back-off of "ab" added
    lp = ou.synth_logprobs(2, 80, 4, 41)
    lp[:, :, 3] += np.float32(1.0)
    kw = dict(beam=20, blank_id=0)
    want = ou.decode(lp, scorer=checkers[0], **kw)
    ou.assert_same(ou.decode_core_host_lm(lp, 0.8, 0.4, path, labels, **kw), want, "unigram model, product core")
    if len(checkers) > 1:
        ou.assert_same(ou.decode(lp, scorer=checkers[1], which="reference", **kw), want, "unigram model, live reference")


@pytest.mark.parametrize("name", gu.lm_names())
def test_restatement_and_product_core_match_committed_lm_fixtures(name):
    args, lm, want = gu.load_lm(name)
    sc = ou.Scorer(lm["alpha"], lm["beta"], lm["lm_path"], lm["labels"], "restated")
    assert (int(sc.is_character_based()), sc.max_order(), sc.dict_size()) == lm["meta"]
    ou.assert_same(ou.decode(scorer=sc, **args), want, name + " (restated)")
    got = ou.decode_core_host_lm(args["probs"], lm["alpha"], lm["beta"], lm["lm_path"], lm["labels"],
                                 **{k: v for k, v in args.items() if k != "probs"})
    ou.assert_same(got, want, name + " (product core, host build)")


def test_reset_params_and_accessors():
    sc = ou.Scorer(0.5, 1.0, os.path.join(DATA, "abcd_words.arpa"), ["_", "a", "b", "c", "d", "'", " "], "restated")
    assert not sc.is_character_based() and sc.max_order() == 3 and sc.dict_size() == 17
    c = LM_CASES[2]
    x, kw = lm_case_inputs(c)
    a = ou.decode(x, scorer=sc, **kw)
    sc.reset_params(0.0, 0.0)  # binding.cpp:283-287
    b = ou.decode(x, scorer=sc, **kw)
    assert not np.array_equal(a["scores"], b["scores"])
    ch = ou.Scorer(0.5, 1.0, os.path.join(DATA, "chars.arpa"), ["_", "a", "b", "c", "d", "'", "é"], "restated")
    assert ch.is_character_based() and ch.dict_size() == 0


# ---- the host-side scorer hook (ctcdecode_amd/csrc/lm_callback.h): the decode asks a CACHE of a callback's answers, parks an
# utterance when the cache misses (ST_NEED_HOST) and resumes it once the callback has answered.  The callback used here asks the
# built-in ARPA tables -- one implementation of the interface -- so the results must equal the fixtures bit for bit.
@pytest.mark.parametrize("name", [n for n in gu.lm_names() if n not in ("abcd_topn",)])  # (the hook's host driver takes unpruned rows)
def test_scorer_hook_matches_committed_lm_fixtures(name):
    args, lm, want = gu.load_lm(name)
    if args["cutoff_top_n"] < args["probs"].shape[2]:
        pytest.skip("pruned fixture")
    got = ou.decode_core_host_lm_cb(args["probs"], lm["alpha"], lm["beta"], lm["lm_path"], lm["labels"], seq_lens=args["seq_lens"], beam=args["beam"],
                                    blank_id=args["blank_id"], log_input=args["log_input"])
    ou.assert_same(got, want, name)
    assert got["callback_calls"] > 0 and got["resumptions"] > 0


def test_scorer_hook_random_sweep():
    """Random word / character model configurations: callback path == built-in path of the same core (bit for bit), with far fewer
    callback calls than queries (the cache) and every utterance resumed at least once."""
    rng = np.random.default_rng(77)
    labels = ["_", "'", " "] + [chr(ord("a") + i) for i in range(26)]
    for it in range(24):
        arpa = os.path.join(gu.DATA_DIR, ["test.arpa", "chars.arpa", "abcd_words.arpa"][it % 3])
        labs = labels if it % 3 != 2 else ["_", " ", "a", "b", "c", "d"]
        V = len(labs)
        T = int(rng.integers(1, 70))
        K = int(rng.choice([1, 3, 8, 20, 40]))
        lp = ou.synth_logprobs(2, T, V, int(rng.integers(0, 1 << 30)), blank_bias=float(rng.choice([0.0, 2.0, 4.0])), quant=[None, 0.5, 1.0][it % 3])
        sl = rng.integers(0, T + 2, size=2).astype(np.int32) if it % 4 == 0 else None
        alpha, beta = float(rng.choice([0.0, 0.5, 1.3])), float(rng.choice([0.0, 1.0, -0.7]))
        a = ou.decode_core_host_lm(lp, alpha, beta, arpa, labs, seq_lens=sl, beam=K, cutoff_top_n=V, threads=1)
        b = ou.decode_core_host_lm_cb(lp, alpha, beta, arpa, labs, seq_lens=sl, beam=K)
        ou.assert_same(b, a, "it %d %s" % (it, os.path.basename(arpa)))


def test_scorer_hook_degenerate_rows():
    """Round 6: rows with whole frames of -inf / overflowing sums behind the hook (rounds 4-5 refused them: ST_CB_DANGER).  In danger mode
    a frame reads the order std::nth_element left the previous beam in; a frame abandoned at a cache miss has already written the NEXT
    beam's order -- into the copy of the other parity (beam_core.h fin_cur / fin_nxt), so running it again reads what the last committed
    frame left.  Callback path == built-in path of the same core, bit for bit; the built-in path against the oracle is
    test_product_core_with_lm_degenerate_inputs."""
    import degenerate_util as du

    rng = np.random.default_rng(2026)
    models = [("abcd_words.arpa", ["_", "a", "b", "c", "d", "'", " "]), ("chars.arpa", ["_", "a", "b", "c", "d", "'", "é", " "]), ("test.arpa", LABELS29)]
    resumed = 0
    for it in range(90):
        arpa, labels = models[it % 3]
        meta, lp = du.make_case(rng, V=len(labels), labels_space=labels.index(" "))
        alpha, beta = float(rng.choice([0.0, 0.3, 1.0, 2.5])), float(rng.choice([-1.0, 0.0, 0.5, 1.5]))
        path = os.path.join(DATA, arpa)
        a = ou.decode_core_host_lm(lp, alpha, beta, path, labels, beam=meta["K"], cutoff_top_n=len(labels), blank_id=0, threads=1)
        b = ou.decode_core_host_lm_cb(lp, alpha, beta, path, labels, beam=meta["K"], blank_id=0)
        ou.assert_same(b, a, "case %d %s %s alpha=%g beta=%g" % (it, arpa, meta, alpha, beta))
        resumed += b["resumptions"]
    assert resumed > 100


def test_scorer_hook_order_one_word_model():
    """a callback scorer of order 1 (windows of one word, no history) over a WORD model -- a prefix that ends in a
    partial non-word scores its "</s>" window from the callback's empty history, not from the built-in automaton's state 0
    (which is never a key of the callback's cache: the decode used to fail with "a query that cannot exist")."""
    arpa = os.path.join(gu.DATA_DIR, "unigram_bo.arpa")
    labs = ["_", " ", "a", "b"]
    rng = np.random.default_rng(5)
    for it in range(12):
        T, K = int(rng.integers(1, 50)), int(rng.choice([1, 4, 12, 30]))
        lp = ou.synth_logprobs(2, T, len(labs), 300 + it, blank_bias=float(rng.choice([0.0, 2.0])))
        alpha, beta = float(rng.choice([0.5, 1.3])), float(rng.choice([0.0, 1.0, -0.7]))
        a = ou.decode_core_host_lm(lp, alpha, beta, arpa, labs, beam=K, cutoff_top_n=len(labs), threads=1)
        b = ou.decode_core_host_lm_cb(lp, alpha, beta, arpa, labs, beam=K)
        ou.assert_same(b, a, "order-1 hook, case %d" % it)
