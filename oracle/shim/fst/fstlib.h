// Project-owned stand-in for <fst/fstlib.h> (OpenFST 1.6.7 is not vendored in the reference tree: setup.py:26-33
// downloads it at build time).
//
// TEST INFRASTRUCTURE ONLY.  It exists so that the reference's *unmodified* decoder sources
// (ctc_beam_search_decoder.cpp, path_trie.cpp, decoder_utils.cpp, scorer.cpp) compile into oracle/_ref/.
//
// What the reference uses of OpenFST, and what this stand-in restates of its published behaviour:
//   * a mutable vector FST (AddState/SetStart/AddArc/SetFinal/Final/Start/NumStates/Copy) -- a container, restated
//     as vectors of arcs per state                                  (decoder_utils.cpp:147-162, scorer.cpp:196-203)
//   * RmEpsilon          -- the dictionary has no epsilon arcs (add_word_to_fst only adds labelled arcs): no-op
//   * Determinize(a, b)  -- subset construction; the input is a bundle of word chains from one start state, so the
//                           result is the prefix trie of the words    (scorer.cpp:223)
//   * Minimize           -- merges equivalent states (shared suffixes).  It changes state NUMBERS only: Find / Final
//                           answers for a given label sequence are those of the un-minimised trie, and state numbers
//                           are never observable through the decoder.  Restated as a no-op.        (scorer.cpp:228)
//   * SortedMatcher      -- SetState / Find(label) / Value().nextstate on input labels >= 1       (path_trie.cpp:60-91)
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

namespace fst {

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
  struct State {
    std::vector<StdArc> arcs;
    TropicalWeight final_w = TropicalWeight::Zero();
  };
  StateId AddState() {
    states_.emplace_back();
    return (StateId)states_.size() - 1;
  }
  void SetStart(StateId s) { start_ = s; }
  StateId Start() const { return start_; }
  void AddArc(StateId s, const StdArc &a) { states_[s].arcs.push_back(a); }
  void SetFinal(StateId s, TropicalWeight w) { states_[s].final_w = w; }
  TropicalWeight Final(StateId s) const { return states_[s].final_w; }
  int NumStates() const { return (int)states_.size(); }
  StdVectorFst *Copy(bool = false) const { return new StdVectorFst(*this); }
  const std::vector<StdArc> &Arcs(StateId s) const { return states_[s].arcs; }

 private:
  std::vector<State> states_;
  StateId start_ = -1;
};

inline void RmEpsilon(StdVectorFst *) {}
inline void Minimize(StdVectorFst *) {}

// Subset construction (acceptor, unweighted: every arc and final weight of the dictionary is One()).
inline void Determinize(const StdVectorFst &in, StdVectorFst *out) {
  if (in.NumStates() == 0) return;
  std::map<std::set<int>, int> ids;
  std::vector<std::set<int>> todo;
  auto id_of = [&](const std::set<int> &s) {
    auto it = ids.find(s);
    if (it != ids.end()) return it->second;
    const int id = out->AddState();
    ids.emplace(s, id);
    todo.push_back(s);
    return id;
  };
  out->SetStart(id_of(std::set<int>{in.Start()}));
  for (size_t k = 0; k < todo.size(); ++k) {
    const std::set<int> cur = todo[k];
    const int src = ids[cur];
    std::map<int, std::set<int>> by_label;
    bool is_final = false;
    for (int s : cur) {
      if (in.Final(s) != TropicalWeight::Zero()) is_final = true;
      for (const StdArc &a : in.Arcs(s)) by_label[a.ilabel].insert(a.nextstate);
    }
    if (is_final) out->SetFinal(src, TropicalWeight::One());
    for (auto &kv : by_label) {
      const int dst = id_of(kv.second);
      out->AddArc(src, StdArc(kv.first, kv.first, TropicalWeight::One(), dst));
    }
  }
}

enum MatchType { MATCH_INPUT = 1, MATCH_OUTPUT = 2 };

template <class F>
class SortedMatcher {
 public:
  SortedMatcher(const F &f, MatchType) : fst_(&f) {}
  void SetState(typename F::StateId s) { state_ = s; }
  bool Find(int label) {
    for (const StdArc &a : fst_->Arcs(state_))
      if (a.ilabel == label) {
        cur_ = &a;
        return true;
      }
    return false;
  }
  const StdArc &Value() const { return *cur_; }

 private:
  const F *fst_;
  typename F::StateId state_ = 0;
  const StdArc *cur_ = nullptr;
};

}  // namespace fst
