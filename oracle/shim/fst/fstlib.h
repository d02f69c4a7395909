// Project-owned stand-in for <fst/fstlib.h> (OpenFST 1.6.7 is not vendored in
// the reference tree: setup.py:26-33 downloads it at build time).
//
// TEST INFRASTRUCTURE ONLY.  It exists so that the reference's *unmodified*
// no-LM decoder sources (ctc_beam_search_decoder.cpp, path_trie.cpp,
// decoder_utils.cpp) compile into oracle/_ref/.  Only type names are needed:
// none of this executes when ext_scorer == nullptr (the dictionary branch,
// path_trie.cpp:59-96, is guarded by has_dictionary_ == false).  Every method
// therefore traps if it is ever reached.
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <memory>
#include <string>
#include <unordered_map>

namespace fst {

[[noreturn]] inline void shim_unreachable() { std::abort(); }

struct TropicalWeight {
  float v;
  TropicalWeight(float x = 0.f) : v(x) {}
  static TropicalWeight Zero() { return TropicalWeight(std::numeric_limits<float>::infinity()); }
  static TropicalWeight One() { return TropicalWeight(0.f); }
  bool operator!=(const TropicalWeight &o) const { return v != o.v; }
};

struct StdArc {
  using Weight = TropicalWeight;
  using StateId = int;
  int ilabel, olabel;
  Weight weight;
  StateId nextstate;
  StdArc(int i, int o, Weight w, StateId n) : ilabel(i), olabel(o), weight(w), nextstate(n) {}
};

class StdVectorFst {
 public:
  using StateId = int;
  StateId AddState() { shim_unreachable(); }
  void SetStart(StateId) { shim_unreachable(); }
  StateId Start() const { shim_unreachable(); }
  void AddArc(StateId, const StdArc &) { shim_unreachable(); }
  void SetFinal(StateId, TropicalWeight) { shim_unreachable(); }
  TropicalWeight Final(StateId) const { shim_unreachable(); }
  int NumStates() const { shim_unreachable(); }
  StdVectorFst *Copy(bool = false) const { shim_unreachable(); }
};

enum MatchType { MATCH_INPUT = 1, MATCH_OUTPUT = 2 };

template <class F>
class SortedMatcher {
 public:
  SortedMatcher(const F &, MatchType) {}
  void SetState(typename F::StateId) { shim_unreachable(); }
  bool Find(int) { shim_unreachable(); }
  const StdArc &Value() const { shim_unreachable(); }
};

}  // namespace fst
