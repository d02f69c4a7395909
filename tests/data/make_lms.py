#!/usr/bin/env python3
"""Writes the two synthetic ARPA language models the LM-tier tests use besides kenlm's public tests/data/test.arpa
(= tests/test.arpa of the reference, a data fixture):

  abcd_words.arpa  a WORD model (order 3) over words spelled with the labels a b c d ' -- short words that share
                   prefixes, so the dictionary gate and the word-boundary scoring fire on most frames of a random input;
  chars.arpa       a CHARACTER model (order 3; every entry is one UTF-8 character, one of them two bytes long), which
                   makes the scorer "character based" (scorer.cpp:65-71): every new label is scored.

Weights are round decimal numbers drawn from a seeded generator; back-off weights may be positive, zero or absent."""
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))


def write(path, order, grams):
    with open(path, "w", encoding="utf-8") as f:
        f.write("\\data\\\n")
        for n in range(1, order + 1):
            f.write("ngram %d=%d\n" % (n, len(grams[n])))
        for n in range(1, order + 1):
            f.write("\n\\%d-grams:\n" % n)
            for words, (p, b) in grams[n]:
                f.write("%s\t%s%s\n" % (p, " ".join(words), "" if b is None or n == order else "\t%s" % b))
        f.write("\n\\end\\\n")


def model(rng, vocab, order, n_higher):
    grams = {1: []}
    grams[1].append((("<unk>",), ("-2.5", "-0.25")))
    grams[1].append((("<s>",), ("-99", "-0.5")))
    grams[1].append((("</s>",), ("-1.25", None)))
    for w in vocab:
        grams[1].append(((w,), ("-%d.%03d" % (rng.randint(0, 2), rng.randint(1, 999)), rng.choice(["-0.125", "-0.5", "-0.75", "0", "0.25", None]))))
    have = {1: {g[0] for g in grams[1]}}
    for n in range(2, order + 1):
        grams[n], have[n] = [], set()
        tries = 0
        while len(grams[n]) < n_higher[n] and tries < 100000:
            tries += 1
            ctx = rng.choice(sorted(have[n - 1]))
            if ctx[-1] == "</s>" or "<unk>" in ctx:
                continue
            g = ctx + (rng.choice(vocab + ["</s>"]),)
            if g in have[n]:
                continue
            have[n].add(g)
            grams[n].append((g, ("-%d.%03d" % (rng.randint(0, 1), rng.randint(1, 999)), rng.choice(["-0.25", "-0.5", "0", "0.125", None]))))
    return grams


def main():
    rng = random.Random(20260924)
    words = ["a", "b", "ab", "ba", "abc", "ad", "da", "dab", "cab", "c", "cc", "d'", "a'b", "bad", "add", "dad", "abba"]
    write(os.path.join(HERE, "abcd_words.arpa"), 3, model(rng, words, 3, {2: 60, 3: 50}))
    chars = ["a", "b", "c", "d", "'", "é"]
    write(os.path.join(HERE, "chars.arpa"), 3, model(rng, chars, 3, {2: 25, 3: 40}))


if __name__ == "__main__":
    main()
