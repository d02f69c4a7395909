// Stand-in for kenlm's util/string_piece.hh: scorer.h:27-29 only needs data()/length().
#pragma once
#include <cstddef>
class StringPiece {
 public:
  StringPiece(const char *p = nullptr, std::size_t n = 0) : p_(p), n_(n) {}
  const char *data() const { return p_; }
  std::size_t length() const { return n_; }
 private:
  const char *p_;
  std::size_t n_;
};
