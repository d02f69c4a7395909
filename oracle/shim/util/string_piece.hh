// Stand-in for kenlm's util/string_piece.hh: data()/length() (scorer.h:27-29) and the implicit conversion from
// std::string that BaseVocabulary().Index(words[i]) relies on (scorer.cpp:81).  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <cstddef>
#include <string>
class StringPiece {
 public:
  StringPiece(const char *p = nullptr, std::size_t n = 0) : p_(p), n_(n) {}
  StringPiece(const std::string &s) : p_(s.data()), n_(s.size()) {}
  const char *data() const { return p_; }
  std::size_t length() const { return n_; }
 private:
  const char *p_;
  std::size_t n_;
};
