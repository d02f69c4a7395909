// Stand-in for kenlm's lm/config.hh: scorer.cpp:60-61 sets Config::enumerate_vocab only.  TEST INFRASTRUCTURE ONLY.
#pragma once
#include "lm/enumerate_vocab.hh"
namespace lm {
namespace ngram {
struct Config {
  EnumerateVocab *enumerate_vocab = nullptr;
};
}  // namespace ngram
}  // namespace lm
