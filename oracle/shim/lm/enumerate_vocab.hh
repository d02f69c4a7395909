// Stand-in for kenlm's lm/enumerate_vocab.hh: scorer.h:23-33 derives from it.
#pragma once
#include "lm/word_index.hh"
#include "util/string_piece.hh"
namespace lm {
class EnumerateVocab {
 public:
  virtual ~EnumerateVocab() {}
  virtual void Add(WordIndex index, const StringPiece &str) = 0;
};
}  // namespace lm
