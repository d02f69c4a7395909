// lm_callback.h -- the host-side swappable scorer hook (north_star: "the KenLM Scorer hook stays on host behind
// the existing swappable-scorer interface").  Host-only; no HIP types.
//
// The reference hands its decoder an opaque `void *scorer` (ctcdecode/src/binding.cpp:122-140) whose
// Scorer::get_log_cond_prob(words) (scorer.h:41-78, scorer.cpp:74-93) any implementation can back; it is called inside the
// per-frame loop (ctc_beam_search_decoder.cpp:120-137).  A host call per query is not an option inside a kernel that spends a
// few microseconds per frame, so the hook is served through a CACHE in the very tables the built-in scorer uses
// (lm_tables.h):
//
//   * a state is a word HISTORY (the N-1 words a window starts with; state 1 = N-1 x "<s>", what make_ngram pads a short
//     prefix with, scorer.cpp:184-189), (state, word) -> {log10 probability as float32 -- what kenlm's BaseScore returns and
//     the reference divides by NUM_FLT_LOGE --, next state} in the same open-addressed table of 16-byte slots;
//   * every state "backs off" to the empty context with weight 0 and every unigram probability is NaN: a pair the cache does
//     not hold reads as NaN through the unchanged n-gram query of the kernel.  The kernel then queues the pair, abandons the
//     frame it is in and parks the utterance in front of it (the streaming machinery: beam_core.h ST_NEED_HOST); the host
//     asks the callback, inserts the answers and resumes the launch.  A window the callback calls out-of-vocabulary is cached
//     as -inf and becomes the reference's OOV_SCORE (scorer.h:16) in the kernel;
//   * the dictionary of a word model (scorer.cpp:196-230) and the label map are built from the vocabulary the caller supplies,
//     exactly as for an ARPA file (lm_build.h build_labels_and_dictionary).
//
// The built-in ARPA tables are one implementation of the same interface (HostScorer::cond_log10): tests run the decoder with a
// callback that asks them and require bit-identical results.  A callback backed by the `kenlm` Python module (binary models
// included) is ctcdecode_amd.KenlmScorer.
#pragma once
#include <cmath>
#include <limits>
#include <chrono>
#include <string>
#include <cstring>
#include <vector>

#include "lm_build.h"

namespace ctclm {

// log10 p(words[n-1] | words[0 .. n-2]) as kenlm's BaseScore gives it (float32).  Returns 0 = ok, 1 = the window holds an
// out-of-vocabulary word (the reference returns OOV_SCORE), < 0 = error (the decode fails).
typedef int (*CondLog10Fn)(void *user, const char *const *words, int n, float *log10_prob);

struct CallbackLm {
  HostScorer hs;  // labels, dictionary, and the cache tables (ng, st_bo, st_fail, uni_prob, uni_state)
  CondLog10Fn fn = nullptr;
  void *user = nullptr;
  // Per state its word history: order - 1 word ids, flat (state 0 is never used); found through an open-addressed index of state ids
  // keyed by the history's hash.  (Rounds 4-5 kept a vector per state behind an unordered_map: two allocations and three dependent cache
  // misses per new window -- a third of the calling thread's time at a 50 000-word model.)
  size_t hl = 0;                                           // order - 1
  std::vector<uint32_t> hist_flat;                         // n_states() * hl words
  std::vector<uint32_t> st_index;                          // state id or 0 = empty; a power of two, at most half full
  size_t n_st = 0;
  std::vector<uint32_t> win_, nh_;                         // scratch of resolve()
  std::vector<const char *> ptr_;
  double cb_seconds = 0.0;                                 // time inside the callback (every 16th call is timed, x 16; whole batches in ask_many)
  size_t used = 0;                                         // cache slots in use
  std::vector<uint32_t> dirty;                             // slots written since the device copy was last brought up to date
  bool rehashed = true;                                    // the whole table must travel
  unsigned long long queries = 0;                          // callback calls so far

  size_t n_states() const { return n_st; }
  const uint32_t *history(uint32_t state) const { return hist_flat.data() + (size_t)state * hl; }
  static uint64_t hist_hash(const uint32_t *h, size_t n) {
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < n; ++i) { x ^= h[i] + 0x7F4A7C15u; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 32; }
    return x;
  }
  void prefetch_state(const uint32_t *h) const { __builtin_prefetch(&st_index[(size_t)hist_hash(h, hl) & (st_index.size() - 1)]); }
  uint32_t state_id(const uint32_t *h) {
    size_t mask = st_index.size() - 1;
    size_t at = (size_t)hist_hash(h, hl) & mask;
    for (; st_index[at] != 0u; at = (at + 1) & mask)
      if (hl == 0 || std::memcmp(history(st_index[at]), h, hl * 4) == 0) return st_index[at];
    const uint32_t id = (uint32_t)n_st++;
    hist_flat.insert(hist_flat.end(), h, h + hl);
    st_index[at] = id;
    if (n_st * 2 > st_index.size()) {  // (everything moves: ids stay)
      std::vector<uint32_t> bigger(st_index.size() * 2, 0u);
      mask = bigger.size() - 1;
      for (uint32_t q = 1; q < (uint32_t)n_st; ++q) {
        size_t a = (size_t)hist_hash(history(q), hl) & mask;
        while (bigger[a] != 0u) a = (a + 1) & mask;
        bigger[a] = q;
      }
      st_index.swap(bigger);
    }
    if (hs.st_bo.size() <= id) { hs.st_bo.resize((size_t)id * 2 + 64, 0.0f); hs.st_fail.resize(hs.st_bo.size(), 0u); }
    return id;
  }

  bool build(double alpha, double beta, int order, const std::vector<std::string> &vocabulary, const std::vector<std::string> &labels,
             CondLog10Fn fn_, void *user_) {
    if (!fn_) return hs.fail("no scorer callback");
    if (order < 1 || order > kMaxOrder) return hs.fail("max_order must be in [1, 6] (KENLM_MAX_ORDER of the reference's build)");
    fn = fn_; user = user_;
    hs.alpha = alpha; hs.beta = beta; hs.order = order; hs.labels = labels;
    hs.vocab.assign(1, "<unk>");
    auto add = [&](const std::string &w) {
      if (w == "<unk>" || hs.word_id.count(w)) return;
      hs.word_id[w] = (uint32_t)hs.vocab.size();
      hs.vocab.push_back(w);
    };
    for (const std::string &w : vocabulary) add(w);
    add("<s>");
    add("</s>");
    hs.char_based = true;  // Scorer::load_lm's test (scorer.cpp:65-71)
    for (const std::string &w : hs.vocab)
      if (w != "<unk>" && w != "<s>" && w != "</s>" && HostScorer::utf8_len(w) > 1) hs.char_based = false;
    // a character model is asked about every label: labels the vocabulary does not list get ids of their own, so that the
    // CALLBACK decides what is out of vocabulary (the reference passes the strings through, scorer.cpp:163-194)
    if (hs.char_based)
      for (const std::string &l : labels) add(l);
    hs.w_bos = hs.id_of("<s>");
    hs.w_eos = hs.id_of("</s>");
    hs.uni_prob.assign(hs.vocab.size(), std::numeric_limits<float>::quiet_NaN());
    hs.uni_state.assign(hs.vocab.size(), 0u);
    hs.st_bo.assign(64, 0.0f);
    hs.st_fail.assign(64, 0u);
    hs.ng.assign(1024, NgSlot{kEmptySlot, 0, 0, 0});
    hl = (size_t)order - 1;
    hist_flat.assign(hl, 0u);  // state 0: the kernel's "empty context", never a key
    n_st = 1;
    st_index.assign(1024, 0u);
    {
      const std::vector<uint32_t> h0(hl, hs.w_bos);
      hs.s0 = state_id(h0.data());
    }
    hs.clean0 = order - 1;  // no word is ever "unknown" to the tables: the callback decides
    used = 0; dirty.clear(); rehashed = true;
    return hs.build_labels_and_dictionary();
  }

  void grow(size_t slots) {  // a table of at least `slots` slots (a power of two): everything moves
    size_t n = hs.ng.size();
    while (n < slots) n *= 2;
    if (n == hs.ng.size()) return;
    std::vector<NgSlot> old;
    old.swap(hs.ng);
    hs.ng.assign(n, NgSlot{kEmptySlot, 0, 0, 0});
    for (const NgSlot &o : old)
      if (o.state != kEmptySlot) place(o);
    rehashed = true;
  }
  // room for `more` insertions without the table moving (a launch that waits for its answers -- ctcdecode_amd.hip cb_rounds --
  // applies single slots to the device copy while it runs: a rehash or a longer state array would pull the tables from under it)
  bool room_for(size_t more, size_t state_cap) const { return (used + more) * 2 <= hs.ng.size() && n_st + more < state_cap; }
  uint32_t insert(const NgSlot &s) {
    if ((used + 1) * 2 > hs.ng.size()) grow(hs.ng.size() * 2);  // at most half full
    const uint32_t h = place(s);
    ++used;
    if (!rehashed) dirty.push_back(h);
    return h;
  }
  long long find_slot(uint32_t state, uint32_t word) const {
    const uint32_t mask = (uint32_t)hs.ng.size() - 1;
    for (uint32_t h = ng_hash(state, word) & mask; hs.ng[h].state != kEmptySlot; h = (h + 1) & mask)
      if (hs.ng[h].state == state && hs.ng[h].word == word) return (long long)h;
    return -1;
  }
  uint32_t place(const NgSlot &s) {
    const uint32_t mask = (uint32_t)hs.ng.size() - 1;
    uint32_t h = ng_hash(s.state, s.word) & mask;
    while (hs.ng[h].state != kEmptySlot) h = (h + 1) & mask;
    hs.ng[h] = s;
    return h;
  }
  bool cached(uint32_t state, uint32_t word) const {
    const uint32_t mask = (uint32_t)hs.ng.size() - 1;
    for (uint32_t h = ng_hash(state, word) & mask; hs.ng[h].state != kEmptySlot; h = (h + 1) & mask)
      if (hs.ng[h].state == state && hs.ng[h].word == word) return true;
    return false;
  }

  // One queued pair in three steps -- prepare (the window's words), ask (the callback: nothing of this object is written, so several
  // windows may be asked at once from several threads when the callback allows it), commit (check the answer, cache it).
  struct Ask {
    uint32_t state = 0, word = 0;
    int n = 0, rc = 0;
    float p10 = 0.f;
    const char *ptr[kMaxOrder + 1];
  };
  bool prepare(uint32_t state, uint32_t word, Ask &a) {
    if (state == 0 || state >= n_st || word == 0 || word >= hs.vocab.size()) return hs.fail("scorer hook: the kernel queued a query that cannot exist (state " + std::to_string(state) + " of " + std::to_string(n_st) + ", word " + std::to_string(word) + " of " + std::to_string(hs.vocab.size()) + ")");
    a.state = state; a.word = word; a.n = (int)hl + 1; a.rc = 0; a.p10 = 0.f;
    const uint32_t *h = history(state);
    for (size_t i = 0; i < hl; ++i) a.ptr[i] = hs.vocab[h[i]].c_str();
    a.ptr[hl] = hs.vocab[word].c_str();
    return true;
  }
  void ask(Ask &a) const { a.rc = fn(user, a.ptr, a.n, &a.p10); }
  bool commit(const Ask &a, uint32_t *slot_out = nullptr) {
    ++queries;
    float p10 = a.p10;
    if (a.rc < 0) return hs.fail("scorer hook: the callback reported an error");
    if (a.rc == 0 && !(p10 == p10)) return hs.fail("scorer hook: the callback returned NaN");
    // (an out-of-vocabulary answer is cached as -inf: a callback that reports a probability of zero that way must say
    //  "out of vocabulary" -- return 1 -- or a finite floor instead; silently turning its -inf into OOV_SCORE would not be what
    //  the reference computes from it -- ADVICE r4)
    if (a.rc == 0 && (p10 > std::numeric_limits<float>::max() || p10 < -std::numeric_limits<float>::max()))
      return hs.fail("scorer hook: the callback returned an infinite log-probability (return 1 for windows with unknown words, a finite value otherwise)");
    if (a.rc != 0) p10 = -std::numeric_limits<float>::infinity();
    NgSlot s;
    s.state = a.state; s.word = a.word;
    std::memcpy(&s.prob_bits, &p10, 4);
    // the window's last N-1 words
    nh_.assign(history(a.state) + (hl ? 1 : 0), history(a.state) + hl);
    if (hl) nh_.push_back(a.word);
    s.next = state_id(nh_.data());
    const uint32_t at = insert(s);
    if (slot_out) *slot_out = at;
    return true;
  }
  // where the pair's next state will be looked up (commit): requested ahead of time
  void prefetch_next_state(uint32_t state, uint32_t word) {
    if (!hl || state >= n_st) return;
    nh_.assign(history(state) + 1, history(state) + hl);
    nh_.push_back(word);
    prefetch_state(nh_.data());
  }

  // one queued pair: ask the callback, cache the answer.  false: the callback failed (hs.error says how)
  // slot_out (optional): where the pair's slot sits in the table afterwards
  bool resolve(uint32_t state, uint32_t word, uint32_t *slot_out = nullptr) {
    if (state != 0 && state < n_st && word != 0 && word < hs.vocab.size()) {
      const long long at = find_slot(state, word);  // (queued by several prefixes / utterances in the same round)
      if (at >= 0) { if (slot_out) *slot_out = (uint32_t)at; return true; }
    }
    Ask a;
    if (!prepare(state, word, a)) return false;
    if ((queries & 15) == 0) {  // (what share of a cold decode is the callback's own time: bench.py reports it)
      const auto t0 = std::chrono::steady_clock::now();
      ask(a);
      cb_seconds += 16.0 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } else {
      ask(a);
    }
    return commit(a, slot_out);
  }
};

}  // namespace ctclm
