#!/usr/bin/env python3
"""Writes a synthetic word-level ARPA language model of realistic SIZE (default: 50 000 words spelled with a..z and ', order 3,
400 000 bigrams, 400 000 trigrams -- tables of tens of MB, far beyond L2-resident test.arpa's 37 unigrams) for the LM tier's
performance lines (bench.py other_configs) and its large-model parity test.  Deterministic; numpy only.

    python tools/make_big_lm.py /tmp/big_words.arpa [--words 50000 --bigrams 400000 --trigrams 400000 --seed 1]
"""
import argparse
import os

import numpy as np

LETTERS = list("abcdefghijklmnopqrstuvwxyz'")


def make(path, n_words=50000, n_bi=400000, n_tri=400000, seed=1):
    rng = np.random.default_rng(seed)
    words = set()
    # short words are frequent (they come first): spelled from a skewed letter distribution, so that prefixes are shared
    pl = np.array([8.2, 1.5, 2.8, 4.3, 12.7, 2.2, 2.0, 6.1, 7.0, 0.2, 0.8, 4.0, 2.4, 6.7, 7.5, 1.9, 0.1, 6.0, 6.3, 9.1, 2.8, 1.0, 2.4, 0.2, 2.0, 0.1, 0.3])
    pl = pl / pl.sum()
    order = []
    while len(order) < n_words:
        L = int(rng.integers(1, 4)) if len(order) < 300 else int(rng.integers(2, 10))
        w = "".join(LETTERS[i] for i in rng.choice(len(LETTERS), size=L, p=pl))
        if w in words or w.strip("'") == "":
            continue
        words.add(w)
        order.append(w)
    ranks = np.arange(1, n_words + 1, dtype=np.float64)
    p = 1.0 / ranks ** 1.05
    p = p / p.sum() * 0.85
    uni_lp = np.log10(p)
    W = n_words
    ids = np.arange(W)

    def draw(k):  # word ids, Zipf-distributed
        return rng.choice(ids, size=k, p=p / p.sum())

    bi = set()
    while len(bi) < n_bi:
        a, b = draw(n_bi), draw(n_bi)
        for x, y in zip(a.tolist(), b.tolist()):
            bi.add((x, y))
            if len(bi) >= n_bi:
                break
    bi = sorted(bi)
    bi_arr = np.array(bi, dtype=np.int64)
    tri = set()
    while len(tri) < n_tri:  # every trigram extends a listed bigram (its context exists, as an ARPA file requires)
        pick = bi_arr[rng.integers(0, len(bi_arr), size=n_tri)]
        c = draw(n_tri)
        for (x, y), z in zip(pick.tolist(), c.tolist()):
            tri.add((x, y, z))
            if len(tri) >= n_tri:
                break
    tri = sorted(tri)
    q3 = lambda n: np.round(-rng.random(n) * 3.0 - 0.05, 4)  # noqa: E731
    bo = lambda n: np.round(-rng.random(n) * 0.9, 4)  # noqa: E731
    with open(path, "w", encoding="utf-8") as f:
        f.write("\\data\\\nngram 1=%d\nngram 2=%d\nngram 3=%d\n\n\\1-grams:\n" % (W + 3, len(bi), len(tri)))
        f.write("-2.8\t<unk>\t-0.2\n-99\t<s>\t-0.6\n-1.3\t</s>\n")
        ub = bo(W)
        for i, w in enumerate(order):
            f.write("%.4f\t%s\t%.4f\n" % (uni_lp[i], w, ub[i]))
        f.write("\n\\2-grams:\n")
        lp2, b2 = q3(len(bi)), bo(len(bi))
        for i, (x, y) in enumerate(bi):
            f.write("%.4f\t%s %s\t%.4f\n" % (lp2[i], order[x], order[y], b2[i]))
        f.write("\n\\3-grams:\n")
        lp3 = q3(len(tri))
        for i, (x, y, z) in enumerate(tri):
            f.write("%.4f\t%s %s %s\n" % (lp3[i], order[x], order[y], order[z]))
        f.write("\n\\end\\\n")
    return path


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--words", type=int, default=50000)
    ap.add_argument("--bigrams", type=int, default=400000)
    ap.add_argument("--trigrams", type=int, default=400000)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    print(make(a.path, a.words, a.bigrams, a.trigrams, a.seed), os.path.getsize(a.path) >> 20, "MiB")
