// Stand-in for kenlm's lm/virtual_interface.hh: the four members of lm::base::Model / Vocabulary the reference calls
// (scorer.cpp:63,79,81,86).  TEST INFRASTRUCTURE ONLY; the implementation is lm/model.hh in this directory.
#pragma once
#include "lm/word_index.hh"
#include "util/string_piece.hh"
namespace lm {
namespace base {
class Vocabulary {
 public:
  virtual ~Vocabulary() {}
  virtual WordIndex Index(const StringPiece &str) const = 0;  // 0 (<unk>) for a word the model does not know
};
class Model {
 public:
  virtual ~Model() {}
  virtual unsigned char Order() const = 0;
  virtual void NullContextWrite(void *to) const = 0;
  virtual float BaseScore(const void *in_state, const WordIndex new_word, void *out_state) const = 0;
  virtual const Vocabulary &BaseVocabulary() const = 0;
};
}  // namespace base
}  // namespace lm
