// Stand-in for kenlm's lm/state.hh: the n-gram context a query carries (most recent word first) and the back-off
// weights of its suffixes, as in kenlm's lm::ngram::State (KENLM_MAX_ORDER = 6, setup.py:57).  TEST INFRASTRUCTURE ONLY.
#pragma once
#include "lm/word_index.hh"
#ifndef KENLM_MAX_ORDER
#define KENLM_MAX_ORDER 6
#endif
namespace lm {
namespace ngram {
struct State {
  WordIndex words[KENLM_MAX_ORDER - 1];
  float backoff[KENLM_MAX_ORDER - 1];
  unsigned char length;
};
}  // namespace ngram
}  // namespace lm
