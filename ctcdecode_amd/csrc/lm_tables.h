// lm_tables.h -- the external scorer of the LM tier as flat tables a CDNA4 workgroup can query.
//
// Replaces, for the decode path, the reference's Scorer object (ctcdecode/src/scorer.{h,cpp}) and its two absent
// third-party engines: kenlm (n-gram query) and OpenFST (dictionary).  The scorer is BUILT on the host
// (lm_build.h: ARPA text -> tables, the counterpart of Scorer::setup, scorer.cpp:43-72,148-161,196-230) and its tables are
// mirrored into HBM once; the per-frame queries of DecoderState::next() (ctc_beam_search_decoder.cpp:120-137) then run
// inside the decode kernel -- a host round trip per frame is not an option at a few microseconds per frame.
//
//   * n-gram model = back-off automaton.  A state is a LISTED n-gram of order < N used as context (state 0 = empty
//     context); (state, word) -> {log10 prob, next state} for every listed n-gram, in an open-addressed table of
//     16-byte slots (one 128-bit load per probe); a miss adds the state's back-off weight and follows its failure
//     link (longest listed proper suffix).  The value returned is kenlm's GenericModel::FullScore: the log10 prob of
//     the longest listed n-gram plus the back-off weights of the longer contexts, added in float32 from the shorter
//     context to the longer (lm/model.cc); n-grams whose own suffix is not listed are found all the same (kenlm fills
//     such gaps at load time; here the failure link simply skips them -- their back-off weight is zero).
//   * Scorer::get_log_cond_prob (scorer.cpp:74-93) feeds the N words of make_ngram's window starting from the empty
//     context and returns OOV_SCORE as soon as one of them is unknown.  Fed incrementally, the automaton state after a
//     prefix equals the state that window would build (the longest listed suffix, at most N-1 words), so every beam
//     entry carries ONE state word plus a counter of how many in-vocabulary tokens it has seen since the last unknown
//     one ("clean"): a window holds an unknown word iff clean < N-1 or the new word is unknown.
//   * dictionary (word models) = prefix trie of (word + " ") over label ids with the children of a node stored
//     contiguously in label order: child(node, c) = first_child + popcount(mask & below(c)).  One 16-byte record per
//     node {label mask (64 bit), first child, word id of the word that ends here}.  Behaviour-equivalent to the
//     determinised + minimised FST of scorer.cpp:196-230 for Find / Final; a completed word restarts
//     the speller at the root (path_trie.cpp:83-92).  Models over MORE than 64 labels ("wide" dictionaries) keep the
//     same node numbering but, instead of a bit mask, the sorted labels of a node's arcs in a side array:
//     {first arc's index in dict_lab, #arcs, first child, word}; an arc is found by binary search (a capability for
//     large label sets, not a fast path: log2(#arcs) dependent reads per gate test).
#pragma once
#include <stdint.h>

#ifndef CTC_HD
#if defined(__HIPCC__)
#define CTC_HD __host__ __device__ __forceinline__
#else
#define CTC_HD inline
#endif
#endif

namespace ctclm {

constexpr uint32_t kEmptySlot = 0xFFFFFFFFu;
constexpr uint32_t kNoWord = 0xFFFFFFFFu;  // dictionary node at which no word ends
constexpr int kMaxOrder = 6;               // KENLM_MAX_ORDER of the reference's build (setup.py:57)
constexpr double kOovScore = -1000.0;      // scorer.h:16

struct NgSlot { uint32_t state, word, prob_bits, next; };         // 16 bytes
struct alignas(16) MissEntry { uint32_t state, word, item, flag; };  // host-side scorer hook: a queued query (one 16-byte store: it arrives whole)
struct DictNode { uint32_t mask_lo, mask_hi, first_child, word; };  // 16 bytes (wide dictionaries: mask_lo = first arc in dict_lab, mask_hi = #arcs)

struct LmView {
  const float *uni_prob;       // [W] log10 p(w), w = word id (0 = <unk>)
  const uint32_t *uni_state;   // [W] state of the unigram w
  const float *st_bo;          // [S] back-off weight of a state (st_bo[0] = 0)
  const uint32_t *st_fail;     // [S] longest listed proper suffix of the state (0 = empty context)
  const NgSlot *ng;            // [ng_mask + 1]
  const DictNode *dict;        // [D] (word models)
  const uint32_t *label_word;  // [V] character models: word id of each label's string (0 = unknown)
  const uint32_t *dict_lab;    // wide dictionaries (more than 64 labels): the nodes' arc labels, ascending per node
  int dict_wide;               // 1: DictNode = {first arc in dict_lab, #arcs, first child, word}
  uint32_t ng_mask;
  int order;                   // N
  int char_based;              // scorer.cpp:65-71
  int space_id;                // label index of " " (-1: none)
  uint32_t s0;                 // state after the (N-1) "<s>" tokens make_ngram pads a short history with
  int clean0;                  // N-1 if "<s>" is a word of the model, else 0
  uint32_t w_bos, w_eos;       // word ids of "<s>" and "</s>" (0 = unknown)
  double alpha, beta;
  // Host-side scorer hook (lm_callback.h): the tables are a CACHE of a host callback's answers -- an explicit automaton over
  // word histories, filled on demand.  A (state, word) pair that is not cached yet reads as NaN (every uni_prob is NaN,
  // every state backs off to the empty context with weight 0); the kernel then appends the pair to cb_miss (cb_count is its
  // bump counter, cb_cap pairs fit) and ends the launch for that utterance at the frame boundary; the host asks the
  // callback, inserts, and the launch resumes.  A cached "out of vocabulary" answer is log10 prob = -inf.
  int cb;                      // 1: callback scorer
  MissEntry *cb_miss;          // [cb_cap] (state, word) asked for by batch item `item`; flag = 1
  unsigned *cb_count;
  uint32_t cb_cap;
  int cb_ring;                 // 1: cb_miss is a ring of cb_cap (a power of two) pairs the host consumes while the launch runs
};

CTC_HD uint32_t ng_hash(uint32_t state, uint32_t word) {
  uint32_t h = state * 0x9E3779B1u ^ (word + 0x7F4A7C15u) * 0x85EBCA77u;
  h ^= h >> 15;
  h *= 0x2C1B3C6Du;
  h ^= h >> 13;
  return h;
}

// The table words the FIRST level of an n-gram query reads -- all independent of one another, so a caller that knows
// (state, word) early can request them long before it needs the result (the decode kernel does: lm_probe_issue() at the top
// of a frame, lm_score_with() two barriers later).
struct LmProbe {
  float uni_p; uint32_t uni_s;  // the unigram fall-back (depends on the word alone)
  float bo; uint32_t fail;      // back-off weight and failure link of the state
  NgSlot slot;                  // first probe of (state, word)
};
CTC_HD LmProbe lm_probe_issue(const LmView &L, uint32_t state, uint32_t word) {
  LmProbe p;
  p.uni_p = L.uni_prob[word];
  p.uni_s = L.uni_state[word];
  p.bo = L.st_bo[state];      // (state 0: weight 0, no failure link, and no n-gram is keyed by it -- the probe misses)
  p.fail = L.st_fail[state];
  p.slot = L.ng[ng_hash(state, word) & L.ng_mask];
  return p;
}

// log10 p(word | state) as kenlm computes it, and the state after the word.  word must be a known word (id != 0).
// `first` = lm_probe_issue(L, state, word).
CTC_HD float lm_score_with(const LmView &L, uint32_t state, uint32_t word, const LmProbe &first, uint32_t *next) {
  // back-off weights met so far, the most recent (= shortest context) first: a register "stack" (an indexed array would
  // live in scratch memory on the GPU)
  float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f, b4 = 0.f, b5 = 0.f;
  int nb = 0;
  uint32_t q = state;
  float prob = 0.f;
  uint32_t nx = 0;
  bool hit = false;
  // Every table word that MAY be needed is requested before anything is waited for: the unigram fall-back (it depends on the
  // word alone) with the first level, a level's back-off weight and failure link together with its first probe -- on the GPU
  // each dependent global access is a round trip of more than a thousand clocks.
  float bo_q = first.bo;
  uint32_t fail_q = first.fail;
  NgSlot s = first.slot;
  while (q != 0 && !hit) {
    uint32_t h = ng_hash(q, word) & L.ng_mask;
    for (;;) {
      if (s.state == q && s.word == word) {
        union { uint32_t u; float f; } cv;
        cv.u = s.prob_bits;
        prob = cv.f;
        nx = s.next;
        hit = true;
        break;
      }
      if (s.state == kEmptySlot) break;
      h = (h + 1) & L.ng_mask;
      s = L.ng[h];
    }
    if (!hit) {
      if (nb < kMaxOrder) { b5 = b4; b4 = b3; b3 = b2; b2 = b1; b1 = b0; b0 = bo_q; ++nb; }
      q = fail_q;
      if (q != 0) {  // the next level
        bo_q = L.st_bo[q];
        fail_q = L.st_fail[q];
        s = L.ng[ng_hash(q, word) & L.ng_mask];
      }
    }
  }
  if (!hit) {
    prob = first.uni_p;
    nx = first.uni_s;
  }
  float r = prob;  // float32, from the shorter context to the longer (lm/model.cc)
  if (nb > 0) r += b0;
  if (nb > 1) r += b1;
  if (nb > 2) r += b2;
  if (nb > 3) r += b3;
  if (nb > 4) r += b4;
  if (nb > 5) r += b5;
  *next = nx;
  return r;
}
CTC_HD float lm_score(const LmView &L, uint32_t state, uint32_t word, uint32_t *next) {
  return lm_score_with(L, state, word, lm_probe_issue(L, state, word), next);
}

// Scorer::get_log_cond_prob of the window that ends with `word` (scorer.cpp:74-93), given the entry's automaton state
// and clean counter: natural-log probability (double), or OOV_SCORE when the window holds an unknown word.  Also
// advances (state, clean) past the word.
CTC_HD double lm_cond_with(const LmView &L, uint32_t *state, int *clean, uint32_t word, const LmProbe &first) {  // word != 0, first = lm_probe_issue(L, *state, word)
  uint32_t nx;
  const float p10 = lm_score_with(L, *state, word, first, &nx);
  const bool oov = *clean < L.order - 1;
  *state = nx;
  *clean = *clean + 1 < L.order - 1 ? *clean + 1 : L.order - 1;
  if (oov) return kOovScore;
  return (double)p10 / (double)0.4342944819f;  // decoder_utils.h:14 NUM_FLT_LOGE is a float constant
}
CTC_HD double lm_cond(const LmView &L, uint32_t *state, int *clean, uint32_t word) {
  if (word == 0) {  // unknown: this window and the next N-1 are OOV; the history restarts after it
    *state = 0;
    *clean = 0;
    return kOovScore;
  }
  return lm_cond_with(L, state, clean, word, lm_probe_issue(L, *state, word));
}

// position of `label` among the arcs of a wide-dictionary node (lo = first arc, cnt = #arcs), -1 if it has no such arc
CTC_HD int dict_find_wide(const LmView &L, uint32_t lo, uint32_t cnt, int label) {
  uint32_t a = 0, b = cnt;
  while (a < b) {
    const uint32_t m = (a + b) >> 1;
    const uint32_t v = L.dict_lab[lo + m];
    if (v == (uint32_t)label) return (int)m;
    if (v < (uint32_t)label) a = m + 1; else b = m;
  }
  return -1;
}
CTC_HD bool dict_has(const DictNode &n, int label) {
  return label < 32 ? (n.mask_lo >> label) & 1u : (n.mask_hi >> (label - 32)) & 1u;
}
CTC_HD uint32_t dict_child(const DictNode &n, int label) {  // label must be an arc of n
  uint32_t below;
  if (label < 32) {
    below = (uint32_t)__builtin_popcount(n.mask_lo & ((1u << label) - 1u));
  } else {
    below = (uint32_t)__builtin_popcount(n.mask_lo) + (uint32_t)__builtin_popcount(n.mask_hi & ((1u << (label - 32)) - 1u));
  }
  return n.first_child + below;
}

}  // namespace ctclm
