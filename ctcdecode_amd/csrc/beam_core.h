// beam_core.h -- CTC prefix beam search for ONE utterance, written for a CDNA4 workgroup.
//
// This is the whole recurrence of the reference's DecoderState::next()/decode()
// (ctcdecode/src/ctc_beam_search_decoder.cpp:56-211) re-designed so that no trie is ever walked:
//
//   * The beam (<= K prefixes) lives in LDS as struct-of-arrays, kept in the trie's DFS PRE-ORDER, together
//     with lcp[i] = depth of the lowest common ancestor of entries i-1 and i.  (A sorted list of strings plus
//     its LCP array IS the compacted trie: subtree ranges, "is my parent in the beam", nearest in-beam ancestor
//     are all range-min questions on lcp[].)
//   * One time step lays the candidate prefixes out in an "Euler tour" slot array that is, by construction,
//     the order in which the reference's PathTrie::iterate_to_vec (path_trie.cpp:128-142) would emit them:
//       open(j)  : [revived dead-interior child | hole] [beam entry j itself]
//       close(i) : the brand-new children of entry i, in candidate-character order (children are appended at
//                  the END of PathTrie::children_, path_trie.cpp:94,103, i.e. after i's whole subtree)
//     with open(j) = 2j + Vnb*(j - a_j), close(i) = 2e_i + Vnb*(e_i - 1 - a_i), where e_i is the end of i's
//     subtree range, a_i the number of in-beam proper ancestors and Vnb the number of non-blank candidates.
//     Children that already exist (in the beam: "hit"; alive but not in the beam: "revive") leave a hole.
//   * Pruning = exact K-th largest of the 48-bit keys (score desc, character asc) = prefix_compare
//     (decoder_utils.cpp:122-132), found by a histogram select (1024 + 64 buckets over a window below the best
//     score, then an exact rank inside the one bucket that holds the K-th key).  If the K boundary cuts through a group of
//     EQUAL keys -- structural at long T, SURVEY.md 7.3-H2 -- or on the last step (whose permutation feeds the final
//     sorts), the workgroup replays libstdc++'s std::nth_element on the DFS-ordered candidate list (Hoare partitions
//     done in parallel, the small tail by one lane with stl_emul.h), so the same prefixes survive as in the
//     reference.  The <= K survivors are gathered, ranked by slot index (= DFS order) and emitted one per thread;
//     the new LCP array comes from range-minima over the old one (the LCA depth of two candidates is determined by
//     their parents' entries), which keeps the DFS-order invariant without ever scanning the dropped candidates.
//   * Trie nodes that survive a step are appended to a per-utterance pool in HBM {parent, char, timestep,
//     log_prob_c}; nothing transient is ever materialised (the reference news/deletes ~2.8k nodes per step).
//     The pool is read back only for (rare) dead-interior lookups and for the final back-trace, which reads every
//     label once (entry j only what it does not share with its DFS predecessor, in 32-label segments reached through
//     per-node express pointers) and copies the shared parts row to row.  The two final std::sorts are replayed
//     exactly too (sort_like_std), one pending introsort range per lane.
//   * Scores use the bit-exact float32 log_sum_exp of exact_math.h.
//   * Work inside a frame is split by WAVE ROLE where it is narrow (entries || new children in the scoring phase;
//     LCP || structure || probabilities || per-frame resets in the emission), and everything every wave executes is
//     kept scalar-cheap: with 16 waves on 4 SIMDs one instruction executed by all of them costs four issue slots.
//     For the usual class of shapes (beam <= 128, <= 32 labels: SMALLV) the bounds are given to the optimiser.
//
// The code is written against an execution policy X (thread id, thread count, barrier, block reductions):
// ctcdecode_amd.hip instantiates it with one workgroup per utterance; tests/native/core_host.cpp instantiates it with a
// single sequential "thread" so the very same source is differential-tested against the oracle on the CPU
// (test infrastructure only -- the product has no CPU path).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "exact_math.h"
#include "exact_math_f64.h"
#include "lm_tables.h"
#include "stl_emul.h"
#if defined(CTC_ASSUME_CHECKED)
#include <cstdio>
#include <cstdlib>
#endif

// Branch-probability hints for the rare paths of a frame (tie replay, revived nodes, pool overflow, last frame): the
// compiler then lays the common path out as fall-through code (measured: +0.7 % at the north-star shape).
#define CTC_RARE(x) __builtin_expect(!!(x), 0)
// (the speculative select's margin control: Decoder::spec_learn)
#if !defined(CTC_SPEC_UP)
#define CTC_SPEC_UP 1.5f
#endif
#if !defined(CTC_SPEC_DN)
#define CTC_SPEC_DN 0.7f
#endif
#if !defined(CTC_SPEC_OVER_SHIFT)
#define CTC_SPEC_OVER_SHIFT 2
#endif
#define CTC_USUAL(x) __builtin_expect(!!(x), 1)

namespace ctcbeam {

struct alignas(16) Int4v { int x, y, z, w; };  // four consecutive, 16-byte aligned words fetched with one access (X::load4)

struct PoolNode {   // one alive-or-retired trie node in HBM: 12 bytes (round 3; 16 before: a quarter of the kernel's HBM writes)
  int32_t parent;   // pool index, -1 for the root
  float lpc;        // the best log_prob_c seen while the node lived (path_trie.cpp:42-45)
  uint32_t cht;     // label (16 bits, 0xFFFF for the root) | low 16 bits of the time step of that log_prob_c << 16; the
                    // step's high bits live in a side array that is only touched by launches that can pass frame 65535
  CTC_HD int ch() const { const uint32_t c = cht & 0xFFFFu; return c == 0xFFFFu ? -1 : (int)c; }
  CTC_HD static uint32_t pack(int ch, int tstep) { return ((uint32_t)ch & 0xFFFFu) | ((uint32_t)tstep << 16); }
};

// Element of the exact replay's candidate list.  Up to 65535 slots the slot index rides in the low 16 bits of one word;
// the widest layout (more slots: cutoff_top_n >= V with thousands of labels) keeps it beside the key.
struct EkWide { uint64_t k; uint32_t slot, pad; };
template <bool HUGE> struct EkOps;
template <> struct EkOps<false> {
  using E = uint64_t; using Pos = uint16_t;
  CTC_HD static E make(uint64_t key48v, int slot) { return (key48v << 16) | (uint64_t)slot; }
  CTC_HD static uint64_t key(const E &e) { return e >> 16; }
  CTC_HD static int slot(const E &e) { return (int)(e & 0xFFFFu); }
};
template <> struct EkOps<true> {
  using E = EkWide; using Pos = uint32_t;
  CTC_HD static E make(uint64_t key48v, int slot) { E e; e.k = key48v; e.slot = (uint32_t)slot; e.pad = 0; return e; }
  CTC_HD static uint64_t key(const E &e) { return e.k; }
  CTC_HD static int slot(const E &e) { return (int)e.slot; }
};

enum : uint32_t { T_SELF = 0, T_CHILD = 1, T_REVIVED = 2, T_HOLE = 3 };
enum : int { ST_OK = 0, ST_POOL_OVERFLOW = 1, ST_BAD_CONFIG = 2 };

constexpr int kMaxBeam = 16383;   // 14-bit entry index inside a slot's info word
constexpr int kMaxVocab = 65534;  // 16-bit (character + 1)
constexpr int kIntMax = 0x7fffffff;

// Order-preserving map float -> uint32 (larger float = larger integer); -0.0 and +0.0 coincide, as they do
// under the reference's operator== / operator> on float scores.
CTC_HD uint32_t ord_f32(float f) {
  const uint32_t u = ctcmath::f32_to_bits(f + 0.0f);  // -0 + +0 = +0: both zeros get the same key; nothing else changes
  return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
}
// ... without the zero canonicalisation: for sums that cannot be -0.  (x + y is -0 only when both are -0; without a scorer a
// prefix score starts at +0 and is from then on a sum with at least one term that is not -0, or logf(.) + max(.), which
// gives -0 only from two -0 terms as well -- so score + log-probability is never -0, whatever the rows hold.)
CTC_HD uint32_t ord_f32_raw(float f) {
  const uint32_t u = ctcmath::f32_to_bits(f);
  return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
}
CTC_HD float unord_f32(uint32_t k) {  // inverse of ord_f32 (the key of either zero gives +0)
  return ctcmath::bits_to_f32((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}
// info word: [31:16] 0xFFFF-(ch+1) (larger = earlier under "character asc"), [15:14] type, [13:0] beam entry
CTC_HD uint32_t mk_info(int ch, uint32_t type, int entry) {
  return ((uint32_t)(0xFFFF - (ch + 1)) << 16) | (type << 14) | (uint32_t)entry;
}
CTC_HD uint32_t info_type(uint32_t info) { return (info >> 14) & 3u; }
CTC_HD int info_entry(uint32_t info) { return (int)(info & 0x3FFFu); }
CTC_HD int info_ch(uint32_t info) { return 0xFFFF - (int)(info >> 16) - 1; }
CTC_HD uint64_t key48(uint32_t skey, uint32_t info) { return ((uint64_t)skey << 16) | (info >> 16); }
constexpr uint32_t kHoleInfo = T_HOLE << 14;

struct Beam {  // struct-of-arrays, capacity K each
  int *node, *par, *ch, *dep, *lcp, *via, *viaanc, *viach;
  int *up;  // express pointer of the entry's node (see kExpress)
  float *bprev, *nbprev, *score, *lpc;
  // LM tier only (Dims::lm): the scorer state every prefix carries (lm_tables.h)
  int *lmst, *lmcl;           // n-gram automaton state, count of in-vocabulary tokens since the last unknown one
  int *acc_lo, *acc_hi;       // double: sum of the windows' conditional log-probs so far (Scorer::get_log_prob, scorer.cpp:111-120)
  int *dn, *dmlo, *dmhi, *dfc;  // word models: dictionary node, its label mask and first child
  int *spc_lo, *spc_hi;       // double: get_log_cond_prob of the window that ends with the word spelled so far (valid when a word ends here)
  int *spst, *spcl;           // automaton state / clean counter after that word
};
constexpr int kBeamArrays = 13, kBeamArraysLm = 12;

struct Dims {
  int K;       // beam width
  int V;       // vocabulary size
  int Vc_max;  // most candidate characters a step can have (V, or cutoff_top_n when pruning)
  int use_rank_table;  // 1 when candidate lists are pruned (rank_of[] needed)
  int lm;              // 1: decoding with the external scorer (the beam carries the LM arrays)
  CTC_HD int S_max() const { return K * (2 + Vc_max); }
};

// LDS scalars.  The per-step counters exist twice (index by step parity) so that a step can reset the other set for
// the step after next without racing with threads that still read this one.
enum {
  VAR_STATUS = 0, VAR_CUT, VAR_TAUC,
  VAR_INTO,  // the input rows did not arrive in time (streamed input, decode_utterance)
  VAR_FB0 = 4, VAR_FB1, VAR_FB2, VAR_FB3,   // 16-byte aligned groups: read back with one LDS access (X::uni4)
  VAR_TAU = 8, VAR_G, VAR_E,
  VAR_DANGER,  // sticky: this utterance has seen a log-probability that can make log_sum_exp depend on the order of its
               // arguments (decoder_utils.h:47-54): "danger mode" (Decoder::note_next_row).  Shares the 16-byte group of the
               // select's result so that phase C reads it for free.
  VAR_PAR0 = 12,  // first per-parity set
  P_NPIN = 0, P_LCOUNT, P_NMAXKEY, P_NMINKEY, P_NCAND, P_SIZE = 8,
  // speculative select (Decoder::kSpec), kept by ONE thread (X::spec_thread) -- every instruction all sixteen waves execute
  // costs sixteen clocks of the frame: the frame's threshold key, the estimate of the best candidate score it was derived
  // from, the safety margin, the observed distance best -> K-th (float bits), the largest log-probability of the staged row,
  // "the prediction is valid"
  // "the prediction is valid", and the histogram window of the frames that fall back (its log2 width, the key it is anchored at)
  // (the first four are what every thread reads before phase B: one 16-byte group)
  VAR_SPEC = VAR_PAR0 + 2 * P_SIZE, SP_THR = 0, SP_WLOG, SP_ANCHOR, SP_PRED, SP_BEST, SP_MARGIN, SP_GAP, SP_ROWMAX,
  VAR_ROWMAX = VAR_SPEC + SP_ROWMAX,
  VAR_QSTAT = VAR_SPEC + 9,   // entries of the launch's final beam with descendants in it (the shape statistic the host reads: chains or bushes)
  VAR_LMMISS = VAR_SPEC + 8,  // host-side scorer hook: this frame asked for something the cache does not hold (sticky within a launch)
  VAR_LMQ = VAR_SPEC + 10,    // ... and how many pairs the utterance has queued since init() / load_state()
  VAR_HOTN = VAR_SPEC + 4,    // wide-beam layouts (kHotPre; they have no speculative select: the slot is theirs): keys pre-listed by phase B
  VAR_COUNT = VAR_SPEC + 12
};
constexpr int kBins = 1024;     // histogram buckets of the select
constexpr int kBinsLog = 10;
constexpr int kListCap = 128;   // exact-rank list (one bucket's keys)
constexpr int kSmallK = 128, kSmallV = 32;  // the class of shapes of the fixed workspace layout (SMALLV == 1 below)
// A second class with a compile-time layout (round 6; SMALLV == 2): beam <= kMidK with at most kMidVc candidate labels per frame out
// of a vocabulary of at most kMidV labels, pruned -- the reference's DEFAULT decoder (beam_width 100, cutoff_top_n 40:
// ctcdecode/__init__.py:26-38) on any vocabulary too large for class 1, i.e. BASELINE configs[3] (V = 10 000).  Same role geometry
// as class 1 (at most 128 entries: two waves per role); 112 x 42 slots leave room for the 20 KB rank table.
constexpr int kMidK = 112, kMidVc = 40, kMidV = 10240;
constexpr int kHotCap = 256;    // speculative select (Decoder::kSpec): capacity of the frame's hot list (keys at or above the predicted threshold)
// (round 5: 24 -> 12.  The tail of a range on ONE lane costs an LDS round trip per access -- 8-18 k clocks for 24 elements --
//  against ~4 k for one more partition round by the whole workgroup; tie frames decide how long a launch lasts
//  (profiles/r05a_utt_spread.txt), and a constant is the one kind of change to this path that cannot disturb the register
//  allocation of the frame loop's common path: configs[1] -1.1 %, inputs without ties unchanged; 8 measures the same, 48 +2.5 %.)
#if !defined(CTC_SERIAL_CUT)
#define CTC_SERIAL_CUT 12
#endif
constexpr int kSerialCut = CTC_SERIAL_CUT;  // introselect ranges at most this long are finished by one lane
// Express pointers for the final back-trace: every pool node X (depth d >= 1) also records up(X) = its ancestor at
// depth ((d - 1) / kExpress) * kExpress, so a label sequence of length d is read back as d / kExpress + 1 independent
// segments of at most kExpress parent hops each instead of one chain of d dependent loads.
constexpr int kExpress = 32;
// Event statistics (X::count; the host build of the core adds them up, tools/beam_stats.py prints them; nothing on the GPU):
// how often a frame takes the paths that are rare on random input and common on beams shaped by a dictionary or by
// peaky acoustic posteriors.
enum Event { EV_FRAMES, EV_CANDIDATES, EV_EXACT, EV_INTERNAL, EV_PINNED, EV_LPC_UPDATE, EV_DEAD_PARENT, EV_REVIVE_CAND, EV_REVIVED, EV_WALK,
             EV_WALK_HOPS, EV_FAST_SELECT, EV_SINGLE_KEY, EV_BUCKET_KEYS, EV_SPEC_OK, EV_SPEC_UNDER, EV_SPEC_OVER, EV_SPEC_OTHER, EV_SPEC_HOT,
             EV_SLOW_BELOW, EV_SLOW_CROWDED, EV_SLOW_SINGLE, EV_SLOW_ROUNDS, EV_TIE_FRAMES, EV_TIE_KEYS, EV_TIE_BIG, EV_WIDE_LIST, EV_HOT_LIST, EV_COUNT };

struct Work {
  // The beam is double-buffered: step t reads the copy of parity p and writes the other one.  cur / nxt are re-derived
  // from (beam0, beam_blk, p) at the top of every step instead of being swapped, so that they are not values carried
  // around the frame loop (26 pointers that would otherwise stay live in scalar registers).
  Beam cur, nxt, beam0;
  size_t beam_blk;  // bytes from an array of copy 0 to the same array of copy 1
  size_t beam_blk_far;  // ... for the beam arrays that live in HBM scratch (wide-beam layouts)
  size_t blk_via, blk_lm;  // which of the two applies to via / viaanc / viach and to the LM arrays
  int *e, *anc, *ostart, *cstart, *pinr, *revr;  // per beam entry, this step
  int *ancbuf, *acntbuf;  // 2K each: nearest in-beam ancestor / number of in-beam ancestors, painted; by step parity
  uint32_t *hit;   // 2 words per entry: ranks (non-blank numbering) of the children that already exist
  float *b_new, *nb_new, *sc_new, *rev_lpc;
  int *cch;        // candidate characters of this step (unused in identity mode)
  float *clp;      // their log-probs (points into clpbuf: the buffer of the current frame)
  float *clpbuf;   // two frames' worth: the next frame's row is staged while the current one is decoded
  int16_t *rank_of;  // V entries, -1 = not a candidate (only when Dims::use_rank_table; V <= 32767 then)
  uint32_t *skey, *sinfo, *pos;  // S_max (+1 for pos): score key, info word, scratch
  int *surv;       // 3K: slots of the survivors | their rank in slot (= DFS) order | inverse of that ranking
  uint64_t *ek;    // S_max: (key48 << 16 | slot) in DFS order, for the exact replay (widest layout: 16-byte elements, EkWide)
  uint16_t *lr;    // 2 * S_max: stop positions of the parallel Hoare partition (widest layout: 32 bit each)
  uint32_t *wpre;  // build_ek_lazy's prefix over the bitmap's words (the histogram's block; widest layout: HBM scratch)
  int *bins;       // kBins buckets of the select histogram (+ kBins/16 more words: with them, the task lists of the final sorts)
  int *hotge;      // kHotCap counters of the speculative select's ranking (fixed-layout class only; zero between frames)
  int *lcpst;      // 7 x kSmallK: range minima of the current beam's LCP array (Decoder::kLcpTable; fixed-layout class without a scorer)
  uint32_t *list;  // kListCap: (key - bucket base + 1) of the keys in the K-th key's bucket
  int *lslot;      // kListCap: their slots
  uint32_t *bitmap;  // one bit per slot: survives (select fast path)
  int *fin, *sstack;
  int stage_skey;  // 1: the exact replay may stage its ranges in the LDS block of the slot keys (they live in LDS: not BIG == 2)
  int *apos;       // K: position of every beam entry in the reference's `prefixes` array (maintained in danger mode only)
  int *vars;
};

template <class P>
CTC_HD P *carve_ptr(char *&p, size_t count) {
  P *r = reinterpret_cast<P *>(p);
  p += ((count * sizeof(P) + 15) / 16) * 16;
  return r;
}

// Lay the workspace out in `base` (LDS on the GPU).  Returns bytes used; call with base == nullptr to size it.
// FARREP: the scratch of the exact std::nth_element replay (candidate list ek, stop positions lr: 12 bytes per slot,
// touched in ~1 % of the frames) lives in `far` (HBM scratch, per utterance; L2-resident in practice) instead of LDS,
// where it is half of the fixed-layout workgroup's 132 KB and keeps a second workgroup off the CU.  The replay's first
// round then runs in `far`; the range that remains is staged in LDS (Decoder::replay_nth_element).  Costs ~5 % of a lone
// launch (measured), buys 1.4-1.5x when two workgroups share a CU: the library picks per call (ctcdecode_amd.hip).
// BIG != 0 (implies FARREP): the per-slot info words live in `far` as well (and are only written on the rare paths:
// LAZY), so that wide beams still fit the 160 KiB of LDS; *far_bytes gets the size of everything in `far`.
// BIG == 1: the rare-path per-slot arrays in HBM; BIG == 2: also the slot keys and the rarely read per-entry arrays
// (dead-interior bookkeeping, existing-child ranks): the widest beams, slowly.  BIG == 3: as 2, for more than 65535
// candidate slots (cutoff_top_n >= V with thousands of labels): 32-bit slot indices in the replay's arrays, the select's
// bitmap and its word prefix in HBM as well -- everything per slot lives there; a capability, far from a fast path.
template <int BIG, bool FARREP = false>
CTC_HD size_t carve(Work &w, char *base, char *far, const Dims &d, size_t *far_bytes) {
  char *p = base;
  char *q = far;
  constexpr bool deep = BIG >= 2;  // (a compile-time choice: every array keeps a static address space, LDS or global)
  constexpr bool huge = BIG >= 3;
  const size_t K = (size_t)d.K, S = (size_t)d.S_max();
  const size_t Kr = ((K * 4 + 15) / 16) * 16;
  Beam *bs[2] = {&w.cur, &w.nxt};
  // the distance between the two copies of a beam array is the same for every array of one memory (Decoder::beam_at).
  // Wide-beam layouts keep the scorer's per-entry state (LM tier) in HBM scratch: a capability, not a fast path.
  constexpr bool lm_far = BIG != 0;
  const size_t n_lm = d.lm ? (size_t)kBeamArraysLm : 0;
  w.beam_blk = (size_t)(kBeamArrays - (deep ? 3 : 0) + (lm_far ? 0 : n_lm)) * Kr;
  w.beam_blk_far = ((deep ? 3 : 0) + (lm_far ? n_lm : 0)) * Kr;
  w.blk_via = deep ? w.beam_blk_far : w.beam_blk;
  w.blk_lm = lm_far ? w.beam_blk_far : w.beam_blk;
  for (int i = 0; i < 2; ++i) {
    Beam &b = *bs[i];
    char *f = q + (size_t)i * w.beam_blk_far;
    b.node = carve_ptr<int>(p, K); b.par = carve_ptr<int>(p, K); b.ch = carve_ptr<int>(p, K);
    b.dep = carve_ptr<int>(p, K); b.lcp = carve_ptr<int>(p, K);
    char *&pv3 = deep ? f : p;
    b.via = carve_ptr<int>(pv3, K); b.viaanc = carve_ptr<int>(pv3, K); b.viach = carve_ptr<int>(pv3, K);
    b.up = carve_ptr<int>(p, K);
    b.bprev = carve_ptr<float>(p, K); b.nbprev = carve_ptr<float>(p, K); b.score = carve_ptr<float>(p, K);
    b.lpc = carve_ptr<float>(p, K);
    const size_t Kl = d.lm ? K : 0;
    char *&pl = lm_far ? f : p;
    b.lmst = carve_ptr<int>(pl, Kl); b.lmcl = carve_ptr<int>(pl, Kl); b.acc_lo = carve_ptr<int>(pl, Kl); b.acc_hi = carve_ptr<int>(pl, Kl);
    b.dn = carve_ptr<int>(pl, Kl); b.dmlo = carve_ptr<int>(pl, Kl); b.dmhi = carve_ptr<int>(pl, Kl); b.dfc = carve_ptr<int>(pl, Kl);
    b.spc_lo = carve_ptr<int>(pl, Kl); b.spc_hi = carve_ptr<int>(pl, Kl); b.spst = carve_ptr<int>(pl, Kl); b.spcl = carve_ptr<int>(pl, Kl);
  }
  q += 2 * w.beam_blk_far;
  w.beam0 = w.cur;
  w.e = carve_ptr<int>(p, K); w.ancbuf = carve_ptr<int>(p, 2 * K); w.acntbuf = carve_ptr<int>(p, 2 * K); w.anc = w.ancbuf;
  w.ostart = carve_ptr<int>(p, K);
  w.cstart = carve_ptr<int>(p, K);
  w.pinr = carve_ptr<int>(deep ? q : p, K);
  w.revr = carve_ptr<int>(deep ? q : p, K); w.hit = carve_ptr<uint32_t>(p, 2 * K);
  w.b_new = carve_ptr<float>(p, K); w.nb_new = carve_ptr<float>(p, K); w.sc_new = carve_ptr<float>(p, K);
  w.rev_lpc = carve_ptr<float>(deep ? q : p, K);
  w.cch = carve_ptr<int>(p, (size_t)d.Vc_max);
  w.clpbuf = carve_ptr<float>(p, 2 * (size_t)d.Vc_max); w.clp = w.clpbuf;
  w.rank_of = carve_ptr<int16_t>(p, d.use_rank_table ? (size_t)d.V : 0);
  w.skey = carve_ptr<uint32_t>(deep ? q : p, S);
  w.surv = carve_ptr<int>(p, 3 * K + 4);
  // (the fixed-layout class also uses the two lists as the hot list of the speculative select: kHotCap entries + padding)
  const size_t lcap = (d.K <= kSmallK && d.Vc_max <= kMidVc) ? (size_t)kHotCap + 64 : (size_t)kListCap + 4;
  w.bins = carve_ptr<int>(p, kBins + kBins / 16 + 4); w.list = carve_ptr<uint32_t>(p, lcap);
  w.lslot = carve_ptr<int>(p, lcap);
  w.hotge = carve_ptr<int>(p, lcap > (size_t)kListCap + 4 ? (size_t)kHotCap : 0);
#if defined(CTC_EXP_LCP_TABLE)
  w.lcpst = carve_ptr<int>(p, (lcap > (size_t)kListCap + 4 && !d.lm) ? (size_t)7 * kSmallK : 0);
#else
  w.lcpst = nullptr;
#endif
  w.bitmap = carve_ptr<uint32_t>(huge ? q : p, 2 * ((S + 63) / 64 + 17));
  w.wpre = huge ? carve_ptr<uint32_t>(q, (S + 63) / 64 + 2) : reinterpret_cast<uint32_t *>(w.bins);
  // (last / exact / danger frames and finish() only: HBM scratch in the wide-beam layouts.  With a scorer: two copies -- behind the
  //  host-side hook a frame can be abandoned after it has written the NEXT beam's order, Decoder::fin_cur / fin_nxt)
  w.fin = carve_ptr<int>(BIG ? q : p, d.lm ? 2 * K : K);
  w.apos = carve_ptr<int>(BIG ? q : p, K);  // (read in danger mode only: HBM scratch in the wide-beam layouts)
  w.sstack = carve_ptr<int>(p, 3 * (2 * 32 + 2));
  w.vars = carve_ptr<int>(p, VAR_COUNT);
  // info words: one per slot -- except in the wide-beam layouts, which never store them (LAZY): there the array holds the
  // K survivors' words between the select and the emission
  w.sinfo = carve_ptr<uint32_t>(p, BIG ? K : S);
  w.pos = carve_ptr<uint32_t>(BIG ? q : p, K + 2);  // (finish(): label offsets of the compact results)
  char *&rr = (BIG || FARREP) ? q : p;
  w.ek = carve_ptr<uint64_t>(rr, huge ? 2 * S : S); w.lr = carve_ptr<uint16_t>(rr, (2 * S + 2) * (huge ? 2 : 1));
  w.stage_skey = deep ? 0 : 1;
  if (far_bytes) *far_bytes = (size_t)(q - far);
  return (size_t)(p - base);
}

// Persistent decoder state of one audio stream (the reference's DecoderState object kept alive between decode()
// calls, ctc_beam_search_decoder.h:73-124 / ctcdecode/__init__.py:253-272), stored in HBM between launches:
// hdr[0..11] = {frames fed so far (abs_time_step, ctc_beam_search_decoder.cpp:69), beam size, pool count, window log,
// best key, danger mode, worst key, the speculative select's prediction (valid, distance, margin), reserved}; arrays = the 13 beam arrays then fin, K entries each (kStateArrays).  The node pool lives in
// the same allocation and is passed separately.
struct StreamState {
  int *hdr;
  int *arrays;
  int finish;  // this call ends the stream: run DecoderState::decode()
};
enum { SH_FRAMES = 0, SH_N, SH_POOL, SH_WLOG, SH_MAXKEY, SH_DANGER, SH_MINKEY,
       SH_SP_PRED, SH_SP_GAP, SH_SP_MARGIN,  // the speculative select's prediction (Decoder::kSpec): a chunk starts where the last one stopped
       SH_WORDS = 12 };
constexpr int kStateArrays = 14;                               // without the LM tier
constexpr int kStateArraysLm = kStateArrays + kBeamArraysLm;   // with it: the LM arrays follow fin
CTC_HD int state_arrays(int lm) { return lm ? kStateArraysLm : kStateArrays; }

struct StepIn {
  int t;           // absolute time step
  int Vc;          // number of candidate characters
  int blank_rank;  // rank of the blank among the candidates, -1 if it was pruned away
  int identity;    // 1: candidate r is character r (no pruning)
  float blank_prob;  // LM tier: log-probability of the blank in this frame, as ctc_beam_search_decoder.cpp:78 takes it
};


// v / d and ceil(v / d) for a divisor that IS a power of two (wave counts, lanes per group; the workgroup size is
// restricted to powers of two for this): the GPU has no integer divide, a run-time division costs ~25 instructions on
// every wave, and a "shift if power of two, else divide" form makes the compiler evaluate both.
CTC_HD int div_p2(int v, int d) { return v >> __builtin_ctz((unsigned)d); }
CTC_HD int ceil_div_p2(int v, int d) { return div_p2(v + d - 1, d); }

CTC_HD double ctc_log_f64(double v) {  // std::log on a double as the reference's C library evaluates it, host or device (exact_math_f64.h)
  return ctcmath::log_f64(v, ctcmath::tables64());
}

CTC_HD int ceil_log2_u32(uint32_t v) {  // smallest s with (1 << s) >= v, v >= 1
  return v <= 1u ? 0 : 32 - __builtin_clz(v - 1u);
}
CTC_HD int ceil_log2_u64(uint64_t v) {  // smallest s with (1 << s) >= v, v >= 1
  return v <= 1 ? 0 : 64 - __builtin_clzll(v - 1);
}

// Where the results of a batch go (binding.cpp:79-99): read only once per utterance, at the very end -- handed over by
// address so that the six values need not stay in registers through the frame loop (X::fresh re-reads them there).
struct OutRefs {
  int32_t *tok, *ts;     // [B][K][T_stride]
  float *score;          // [B][K]
  int32_t *len;          // [B][K]
  int32_t *n_results;    // [B] or null
  int K, T_stride;
  // Compact delivery (tok == ts == nullptr): instead of K rows of T labels, every beam entry hands over only the labels it
  // does not share with its DFS predecessor -- the beam is a trie, its K label sequences overlap almost entirely.
  //   c_hdr[item]    = {#results, #labels of this item, first label's index in c_rag, 0}
  //   c_ent[item][j] = {result row of DFS entry j, #labels shared with entry j-1, length, index of its own labels in c_rag}
  //   c_rag[...]     = label | timestep << 16   (all items share one buffer; *c_count is its bump allocator)
  int32_t *c_hdr, *c_ent;
  uint32_t *c_rag;
  unsigned *c_count;
  unsigned c_cap;
  // Mirrors in page-locked HOST memory (device-visible addresses; null: none): an utterance that has finished copies its
  // compact results there itself -- coalesced stores over PCIe while other utterances are still being decoded -- and then
  // raises m_done[item] (1: mirrored; 2: its labels did not fit the mirror, fetch them from c_rag), so that the host can
  // expand utterance by utterance without waiting for the kernel to end (ctcd_beam_decode_to_host).
  int32_t *m_hdr, *m_ent, *m_done;
  uint32_t *m_rag;
  unsigned m_cap;
};
enum : int { ST_COMPACT_OVERFLOW = 3, ST_INPUT_TIMEOUT = 4,
             ST_NEED_HOST = 5,   // host-side scorer hook: a query the table cache does not hold yet -- the utterance is parked at a frame boundary
             ST_CB_DANGER = 6 }; // ... and degenerate inputs (danger mode) are not supported with a callback scorer

// What it takes to name a slot of a frame: the beam the frame started from, its #entries, #non-blank candidates (and the
// multiplier that divides by it), the blank's rank; the layout itself is in w.ostart / w.cstart / w.anc.
struct SlotCtx {
  const Beam *pb;
  int n, Vnb, brank;
  uint64_t vmagic;  // 2^32 / Vnb rounded up: (q * vmagic) >> 32 == q / Vnb for q < 2^16
};

// An execution policy seen through "every barrier is a full fence": for the stretches of the exact replay that work on
// arrays in HBM scratch while the kernel's ordinary barriers only wait for LDS traffic.
template <class X>
struct FullSyncView {
  X &x;
  CTC_HD int tid() const { return x.tid(); }
  CTC_HD int nt() const { return x.nt(); }
  CTC_HD void sync() { x.sync_full(); }
  CTC_HD int uni(int v) const { return x.uni(v); }
  CTC_HD void block_scan_u32(uint32_t mine, uint32_t *base_out, uint32_t *total_out) { x.block_scan_u32(mine, base_out, total_out); }
  CTC_HD int lanes() const { return x.lanes(); }
  CTC_HD auto ballot(bool p) const { return x.ballot(p); }
  template <class M> CTC_HD int count(M m) const { return x.count(m); }
  template <class M> CTC_HD int count_below(M m) const { return x.count_below(m); }
  CTC_HD uint32_t first_lane(uint32_t v) const { return x.first_lane(v); }
};

// IDENT: the utterance is decoded without vocabulary pruning (candidate r of every frame is label r).  A compile-time
// switch: the two modes keep different things in flight across a frame (next row vs. next candidate list), and mixing
// them in one instantiation makes the compiler wait for the prefetch where the other mode's registers are written.
// SMALLV: the caller guarantees beam <= kSmallK and at most kSmallV labels (the class of shapes of the fixed workspace
// layout): the paths for larger candidate sets are compiled out and the bounds are told to the optimiser.
#if defined(CTC_ASSUME_CHECKED)  // the host build of the tests verifies every assumption instead of exploiting it
#define CTC_ASSUME(c) do { if (!(c)) { std::fprintf(stderr, "beam_core.h:%d: assumption violated: %s\n", __LINE__, #c); std::abort(); } } while (0)
#elif defined(__clang__)
#define CTC_ASSUME(c) __builtin_assume(c)
#else
#define CTC_ASSUME(c) do { if (!(c)) __builtin_unreachable(); } while (0)
#endif

// LM: decode with the external scorer (ctc_beam_search_decoder.cpp:74-82,93-95,120-137,173-206; tables: lm_tables.h).
// LAZY (the wide-beam layouts, whose per-slot info words live in HBM scratch): the info word of a slot -- which label,
// which kind of candidate, which beam entry -- is a pure function of the slot layout, so it is not stored while the
// candidates are scored.  The <= K survivors find theirs with a binary search over the entries' slot offsets
// (info_of_slot); the rare paths that look at every slot (ties at the K boundary, exact replay) first rebuild all of them
// (fill_info).  Round 2 wrote S info words per frame to HBM: 30 GB per configs[2] launch, 18x the algorithmic bytes.
// FARREP: the exact replay's scratch lives in HBM (carve).  HUGE: more than 65535 candidate slots (carve BIG == 3).
template <class X, bool IDENT, int SMALLV = 0, bool LM = false, bool LAZY = false, bool FARREP = LAZY, bool HUGE = false, bool WORDLM = false, bool CB = false>
struct Decoder {
  using EO = EkOps<HUGE>;
  using Ek = typename EO::E;
  using LrT = typename EO::Pos;
  // the bounds of the shape class (SMALLV: 0 none, 1 = kSmallK x kSmallV labels, 2 = kMidK x kMidVc candidates of a pruned vocabulary)
  static constexpr int kClsK = SMALLV == 2 ? kMidK : kSmallK, kClsVc = SMALLV == 2 ? kMidVc : kSmallV, kClsSlots = kClsK * (2 + kClsVc);
  CTC_HD Ek *ekp() const { return reinterpret_cast<Ek *>(w.ek); }
  CTC_HD LrT *lrp() const { return reinterpret_cast<LrT *>(w.lr); }
  X &x;
  Work &w;
  const Dims d;
  const int blank;
  PoolNode *pool;
  int *pool_up;  // up(X), stored for the nodes whose depth is a multiple of kExpress only
  int *pool_thi; // time step >> 16 per node (behind pool_up); read / written only when long_t
  bool long_t = false;  // this launch may see absolute frame numbers beyond 65535
  const int pool_cap;
  const uint64_t *tbl;  // exact_math tables
  const ctclm::LmView *lm;  // LM tier only

  CTC_HD Decoder(X &x_, Work &w_, const Dims &d_, int blank_, PoolNode *pool_, int *pool_up_, int pool_cap_, const uint64_t *tbl_,
                 const ctclm::LmView *lm_ = nullptr)
      : x(x_), w(w_), d(d_), blank(blank_), pool(pool_), pool_up(pool_up_), pool_thi(pool_up_ + pool_cap_), pool_cap(pool_cap_), tbl(tbl_), lm(lm_) {
    if (LM) {
      const ctclm::DictNode root = lm->dict[0];
      root_lo = (int)root.mask_lo; root_hi = (int)root.mask_hi; root_fc = (int)root.first_child;
      lm_char_v = lm->char_based != 0; lm_wide_v = lm->dict_wide != 0; lm_space = lm->space_id;
      lm_alpha = lm->alpha; lm_beta = lm->beta;
      lm_dict = lm->dict;
    }
  }
  CTC_HD int node_tstep(const PoolNode &pn, int id) const { return (int)(pn.cht >> 16) | (CTC_RARE(long_t) ? pool_thi[id] << 16 : 0); }
  // a node's (label, time step) word and, where frame numbers need more than 16 bits, the step's high part
  CTC_HD void set_node_time(int id, int ch, int tstep, float lpc) const {
    pool[id].lpc = lpc;
    pool[id].cht = PoolNode::pack(ch, tstep);
    if (CTC_RARE(long_t)) pool_thi[id] = tstep >> 16;
  }

  CTC_HD float lse(float a, float b) const { return ctcmath::lse(a, b, tbl); }

  // "The tail is zero": in identity mode the blank is always a candidate, so the number of slots is S = n (1 + V), and
  // without a scorer n only grows while an utterance is decoded; a frame writes every slot below S.  The fixed-layout
  // kernels therefore keep the slot keys from S to the end of their block at zero -- cleared once (init / load_state), and
  // again after an exact replay that used the block as its staging area -- and the select's listing pass reads its keys
  // without bounds tests (decode_kernel.h list_bucket).  Only where the execution policy asks for it (the GPU).
  // With a scorer S can shrink (candidates are cut, the beam may not fill): the keys between the next frame's S and this
  // frame's are then cleared after the frame's closing barrier (step()).
  // The pruned default's compile-time class (SMALLV == 2, at most 40 candidates per frame): the rank table's entries carry the frame they
  // were written in -- (t mod 1024) << 6 | rank, 0xFFFF = never -- so that a frame's candidates need not be taken out of the table again
  // behind the emission (a barrier and a loop per frame); the table is wiped every 1024 frames, when the tags would repeat.
  static constexpr bool kRankEpoch = SMALLV == 2 && !IDENT && X::kRankEpoch;
  CTC_HD static int16_t rank_tag(int t, int r) { return (int16_t)(uint16_t)((((uint32_t)t & 1023u) << 6) | (uint32_t)r); }
  static constexpr bool kTailZero = IDENT && SMALLV && !LAZY && X::kZeroKeyTail;
  CTC_HD void zero_key_tail(int from) {
    if (!kTailZero) return;
    for (int i = from + x.tid(); i < kClsSlots; i += x.nt()) w.skey[i] = 0u;  // (the fixed layout's block: carve)
  }

  // ---- speculative select (round 4) -----------------------------------------------------------------------------------
  // The K-th best key moves with the best key from frame to frame, so the frame PREDICTS it: threshold = (best score of the
  // beam + the row's largest log-probability) - (that same quantity minus the K-th best score, as observed in the previous
  // frame) - margin.  While the candidates are scored (phase B) every key at or above the threshold is appended to a short
  // "hot list" (one wave-aggregated LDS atomic per pass) INSTEAD of being counted into the 1024-bucket histogram.  If the
  // list then holds H keys with K <= H <= kHotCap, the K best candidates are all in it -- whatever the threshold was: a key
  // outside the list is below every key inside -- and one all-pairs ranking of the list by the whole workgroup gives the
  // exact K-th key, says whether equal keys straddle the boundary, and marks the survivors; a prefix count over the
  // survivor bitmap puts them in slot (= DFS) order.  The histogram, the bucket search, the re-scan of all S slots, the
  // single-wave ranking and the single-wave bitmap expansion of the histogram select (five barriers, three single-wave
  // stages) become two all-wave stages and two barriers.  Everything else -- too few or too many hot keys, ties at the
  // boundary, the last frame, danger mode -- falls back to the histogram select, which first has to build its histogram
  // (rehistogram()): same survivors either way, by construction.
  // (round 6: pruned candidate lists as well -- the row's largest log-probability is then simply its first candidate's)
  // (round 6, measured and left off -- X::kSpecLm, -DCTC_EXP_SPEC_LM: the scorer's fixed-layout kernels as well.  Nothing in the scheme depends
  //  on where a candidate's score comes from, and the outputs are identical (GPU: three input kinds at the configs[4] shape; host sweeps) --
  //  but a dictionary leaves ~ 2 K live candidates per frame (206 at beam 100 with test.arpa), the K-th key lies in the bulk of them, the
  //  band "between K and 128 hot keys" is hit by 61 % of the frames only and their lists average 146 keys (the slow ranking form): kernel
  //  10.65 -> 10.93 ms on random rows, 10.61 -> 10.83 peaky, 9.72 -> 10.01 blank-dominated.  Listing EVERY live candidate whenever the last
  //  frame had at most 248 (84 % of the frames then settle, from lists of 156): 10.32 -> 10.53 / 10.31 -> 10.50 / 9.63 -> 9.57.  Not behind the scorer hook in any case: a frame
  //  that is abandoned and run again would have to restore the list.)
  static constexpr bool kSpec = SMALLV && (!LM || (X::kSpecLm && !CB)) && !LAZY && X::kSpecSelect;
  // Wide beams without a scorer (round 6; measured and left off: CTC_EXP_HOT_PRELIST): phase B pre-lists the candidates at or above a
  // threshold predicted from the previous frame (the speculative select's anchor -- previous best score + this row's best label
  // log-probability -- minus 1.5 x the last distance anchor -> K-th key) in the next beam's block; if the K-th key's bucket turns out to lie
  // at or above that threshold, the bucket is listed and the keys above it are marked FROM THAT LIST (1 000-1 700 entries) instead of by the
  // pass over all S slot keys (15 500 at beam 500).  The prediction holds in 88 % of the frames that stay on the select's fast path (66 % of
  // all; host build) and the outputs are identical -- and the kernel is SLOWER: configs[2] per-GPU shape 39.9 -> 41.65 ms, beam 300 14.44 ->
  // 15.07 (profiles/r06x_cfg2_prelist.txt).  Phase B is the critical, issue-bound phase of this kernel (its fourteen child waves: 8 k of a
  // frame's 31 k clocks at beam 400), and 1 000-1 700 appends per frame -- returning atomics on ONE LDS word -- queue behind each other there;
  // the wave-aggregated form costs ~12 instructions in nearly every pass of a 40-instruction loop (one candidate in nine is hot: every
  // wave-pass has one).  The listing pass it saves (3.9 k clocks on sixteen waves) is cheaper than anything that touches phase B.
#if defined(CTC_EXP_HOT_PRELIST)
  static constexpr bool kHotPre = LAZY && !LM && !SMALLV && !HUGE;
#else
  static constexpr bool kHotPre = false;
#endif
  struct HotPre { uint32_t thr; int cap; uint32_t *key; int *slot; };
  uint32_t st_gap = 0;  // distance anchor -> K-th key of the last frame that had more candidates than places (0: no prediction)
  uint32_t hot_anchor = 0;
  // Round 6: phase A1 of frame t + 1 reads nothing but the NEW beam's depth / LCP arrays, and the emission of frame t keeps six
  // of the sixteen waves busy.  The emission therefore writes those two arrays first, a barrier follows, and eight of the idle
  // waves run phase A1 of the next frame while the role waves finish the emission (pool appends, probabilities, best key); the
  // frame's closing fence is then also the barrier between A1 and A2 of the next frame, which starts at A2.  The first frame
  // of a launch is primed before the loop (prime_a1).  Speculative-select class at 1024 threads only (step(): emission).
  static constexpr bool kA1Overlap = kSpec && X::kA1Overlap;
  // Round 6: the emission's longest chain was the new LCP array -- per survivor a range minimum over the old one, a loop of LDS
  // round trips whose length differs from lane to lane.  One wave (the last: the children's part of phase B loses nothing by
  // it -- 26 instead of 28 parents per pass are four passes at beam 100 either way) builds a table of range minima of the
  // CURRENT beam's LCP array while the candidates are scored (build_lcp_table), and a survivor's LCP is two reads (lca_depth_tbl).
  static constexpr bool kLcpTable = kSpec && X::kLcpTable;
  // word models, fixed-layout class at 1024 threads, built-in tables: the n-gram query of a new entry's word runs beside phase B
  // (step(): phase A2 / phase B)
  // The run-time layouts without a scorer (round 6; wide beams first): phase B's children loop asks FIVE per-entry arrays for every parent -- first slot, parent label,
  // score, blank part, existing-children mask -- and in the run-time layouts each array costs an address of its own (ten VALU adds and five
  // LDS reads of the 49 instructions one candidate takes; the loop is bound by issue: 14 000 candidates per frame at beam 500).  Phase A2,
  // where each entry's thread computes the first slot anyway, packs {score, blank part, first slot, label} into ONE 16-byte record per entry
  // (in the next beam's block: nothing lives there between two emissions); the children loop then makes two reads (record, mask) from two
  // addresses.
  static constexpr bool kParentRec = !SMALLV && !LM && X::kParentRec && !kHotPre;  // (the pre-list experiment borrows the same block)
  static constexpr bool kLmOverlap = LM && WORDLM && SMALLV && !CB && !LAZY && X::kLmOverlap;
  static constexpr uint32_t kLmSpaceDeferred = 0x80000000u;  // high gate word of an entry whose space child the settling wave scores
  // ONE thread (X::spec_thread) turns what a frame observed -- its K-th key, the size of its hot list -- into the next frame's
  // threshold SP_THR, which everyone reads behind phase A2's barrier.  The best key of the previous frame's survivors is
  // still in that frame's per-parity counters (P_NMAXKEY: they are reset during THIS frame's emission).
  // ... in two parts, each where its wave has time to spare.  spec_learn: during the emission (the thread's wave is one of
  // those without a role there), from this frame's K-th key and hot-list length.  spec_predict: between the barriers of
  // phases A1 and A2 of the next frame, once that frame's anchor (the best key the emission found) is final.
  CTC_HD void spec_learn(bool selected, uint32_t tau, int hot, int K) const {
    int *sp = w.vars + VAR_SPEC;
    const int pred = sp[SP_PRED];
    const float best = ctcmath::bits_to_f32((uint32_t)sp[SP_BEST]);
    float margin = ctcmath::bits_to_f32((uint32_t)sp[SP_MARGIN]);
    const uint32_t anchor = (uint32_t)sp[SP_ANCHOR];
    int wl = 32;
    float cut = 0.f;
    if (selected) {
      // too few hot keys: widen the margin; more than a quarter above K: narrow it (beyond 128 keys the ranking costs two and
      // a half times as much).  Factors and trigger from a cost model over recorded key sets (tools/select_stats.py): falling
      // back 3.7 k clocks, a long list 1.5 k -- random rows 970 -> 710 clocks per frame against (x2, x0.81, K + K/2 + 10).
      if (pred) {
        if (hot < K) margin = margin < 32.f ? margin * CTC_SPEC_UP : margin;
        else if (hot > K + (K >> CTC_SPEC_OVER_SHIFT) + 3) margin = margin > 0.001f ? margin * CTC_SPEC_DN : margin;
      }
      cut = (best - unord_f32(tau)) + margin;  // the next threshold lies this far below the next frame's best estimate
      // the window of a frame that falls back to the histogram select: anchored at the best key, reaching twice as far down
      // as this frame's K-th key lay below ITS anchor, rounded up to a power of two (as the histogram select keeps it)
      const uint32_t kgap = anchor > tau ? anchor - tau : 0u;
      wl = (kgap ? 32 - __builtin_clz(kgap) : 0) + 1;
      wl = wl < kBinsLog ? kBinsLog : (wl > 32 ? 32 : wl);
    }
    sp[SP_MARGIN] = (int)ctcmath::f32_to_bits(margin); sp[SP_GAP] = (int)ctcmath::f32_to_bits(cut);
    sp[SP_PRED] = selected ? 1 : 0; sp[SP_WLOG] = wl;
  }
  CTC_HD void spec_predict(int t) const {
    int *sp = w.vars + VAR_SPEC;
    const int pred = sp[SP_PRED];
    const float rowmax = ctcmath::bits_to_f32((uint32_t)sp[SP_ROWMAX]), cut = ctcmath::bits_to_f32((uint32_t)sp[SP_GAP]);
    const uint32_t mk = (uint32_t)pvars(t - 1)[P_NMAXKEY];
    const float best = unord_f32(mk) + rowmax;
    uint32_t k = ord_f32(best - cut);
    k = k ? k : 1u;               // (holes have key 0 and are never hot)
    k = pred ? k : 0xFFFFFFFFu;   // no prediction: nothing is hot, the frame goes the histogram way
    sp[SP_THR] = (int)k; sp[SP_BEST] = (int)ctcmath::f32_to_bits(best); sp[SP_ANCHOR] = (int)mk;
  }

  // Step-to-step state, identical in every thread (kept in registers, not LDS)
  int st_n = 1, st_pool = 1, st_wlog = 32;
  uint32_t st_maxkey = 0;
  uint32_t st_minkey = 0;  // LM tier: key of the worst score in the beam (min_cutoff, ctc_beam_search_decoder.cpp:79)
  int root_lo = 0, root_hi = 0, root_fc = 0;  // LM tier: the dictionary's root record (where every word starts)
  // the scorer's parameters the per-candidate code needs, read once (the tables stay behind `lm`)
  bool lm_char_v = false, lm_wide_v = false;
  // WORDLM: this instantiation serves word models over at most 64 labels only (the caller picks it: ctcdecode_amd.hip) --
  // the character-model branches, which put an n-gram query into every candidate's path, and the wide-dictionary
  // branches are not compiled in (the kernel's frame loop is a third shorter: -3.6 % per frame)
  CTC_HD bool lm_char_() const { return !WORDLM && lm_char_v; }
  CTC_HD bool lm_wide_() const { return !WORDLM && lm_wide_v; }
  int lm_space = -1;
  double lm_alpha = 0.0, lm_beta = 0.0;
  const ctclm::DictNode *lm_dict = nullptr;
  // Host-side scorer hook: the tables are a cache of a host callback's answers (lm_tables.h LmView::cb).  A compile-time switch
  // (CB): as a run-time flag the hook's checks and the state they keep alive cost the built-in scorer's kernels 11 % per frame
  // (12.36 against 11.16 ms at the configs[4] shape); callback scorers get instantiations of their own.
  static constexpr bool lm_cb = LM && CB;
  // fin[]: the order std::nth_element left the beam in (read by danger mode and by finish()).  Behind the scorer hook the frame that
  // writes the next beam's order may be abandoned and run again: it writes the copy of the other parity, which becomes current only
  // when the frame commits (st_par flips).
  CTC_HD int *fin_cur() const { return w.fin + (lm_cb ? st_par * d.K : 0); }
  CTC_HD int *fin_nxt() const { return w.fin + (lm_cb ? (st_par ^ 1) * d.K : 0); }
  // get_log_cond_prob through the tables.  With a callback scorer a value that is not cached yet comes back as NaN: the pair
  // is queued for the host, the frame is marked (it will not be committed: step() / finish() return ST_NEED_HOST) and the
  // caller's state is left where it was; a cached "out of vocabulary" answer (-inf) becomes the reference's OOV_SCORE.
  CTC_HD double lm_cond_(uint32_t *st, int *cl, uint32_t word, bool *missed = nullptr) const {
    const uint32_t st0 = *st;
    const int cl0 = *cl;
    const double v = ctclm::lm_cond(*lm, st, cl, word);
    if (CTC_RARE(lm_cb)) {
      if (v != v) {
        w.vars[VAR_LMMISS] = 1;
        unsigned i = x.global_add(lm->cb_count, 1u);
        if (lm->cb_ring) i &= lm->cb_cap - 1;  // (a launch that waits for its answers: the host empties the list while it fills)
        if (i < lm->cb_cap) lm->cb_miss[i] = ctclm::MissEntry{st0, word, (uint32_t)x.item(), 1u};
        x.atomic_add(&w.vars[VAR_LMQ], 1);  // (pairs this utterance has queued since it was taken up: what its workgroup waits to see answered)
        *st = st0; *cl = cl0;
        if (missed) *missed = true;
        return 0.0;
      }
      if (v == -__builtin_huge_val()) return ctclm::kOovScore;
    }
    return v;
  }
  static constexpr int kLmPending = -1;       // dfc of an entry whose dictionary record / cached window are not fetched yet
  int st_par = 0;  // which copy of the beam is current

  template <class P>
  CTC_HD static P *shifted(P *q, size_t bytes) { return reinterpret_cast<P *>(reinterpret_cast<char *>(q) + bytes); }
  CTC_HD Beam beam_at(int par) const {
    const Beam &o = w.beam0;
    const size_t off = par ? w.beam_blk : 0, offv = par ? w.blk_via : 0, offl = par ? w.blk_lm : 0;
    Beam r;
    r.node = shifted(o.node, off); r.par = shifted(o.par, off); r.ch = shifted(o.ch, off); r.dep = shifted(o.dep, off);
    r.lcp = shifted(o.lcp, off); r.via = shifted(o.via, offv); r.viaanc = shifted(o.viaanc, offv); r.viach = shifted(o.viach, offv);
    r.up = shifted(o.up, off); r.bprev = shifted(o.bprev, off); r.nbprev = shifted(o.nbprev, off);
    r.score = shifted(o.score, off); r.lpc = shifted(o.lpc, off);
    if (LM) {
      r.lmst = shifted(o.lmst, offl); r.lmcl = shifted(o.lmcl, offl); r.acc_lo = shifted(o.acc_lo, offl); r.acc_hi = shifted(o.acc_hi, offl);
      r.dn = shifted(o.dn, offl); r.dmlo = shifted(o.dmlo, offl); r.dmhi = shifted(o.dmhi, offl); r.dfc = shifted(o.dfc, offl);
      r.spc_lo = shifted(o.spc_lo, offl); r.spc_hi = shifted(o.spc_hi, offl); r.spst = shifted(o.spst, offl); r.spcl = shifted(o.spcl, offl);
    }
    return r;
  }
  CTC_HD void select_beams() { w.cur = beam_at(st_par); w.nxt = beam_at(st_par ^ 1); }

  CTC_HD int *pvars(int t) const { return w.vars + VAR_PAR0 + (t & 1) * P_SIZE; }
  CTC_HD void reset_pvars(int *pv) const {
    pv[P_NPIN] = 0; pv[P_LCOUNT] = 0; pv[P_NMAXKEY] = 0;
    if (LM) { pv[P_NMINKEY] = -1; pv[P_NCAND] = 0; }
  }

  // ---- LM tier helpers -------------------------------------------------------------------------------------------
  CTC_HD static double mk_f64(int lo, int hi) {
    union { uint64_t u; double f; } cv;
    cv.u = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
    return cv.f;
  }
  CTC_HD static void put_f64(double v, int *lo, int *hi) {
    union { uint64_t u; double f; } cv;
    cv.f = v;
    *lo = (int)(uint32_t)cv.u;
    *hi = (int)(uint32_t)(cv.u >> 32);
  }
  // log_p += score * alpha; log_p += beta, with the reference's types (float score = double * double; float += double)
  CTC_HD float lm_apply(float log_p, double cond) const {  // ctc_beam_search_decoder.cpp:131-136
    float score = 0.0f;
    score = (float)(cond * lm_alpha);
    log_p += score;
    log_p = (float)((double)log_p + lm_beta);
    return log_p;
  }
  // does extending an entry with label c call the scorer?  (:121-122)
  CTC_HD bool lm_scores(int c) const { return lm_char_() || c == lm_space; }
  // get_log_cond_prob(make_ngram(.)) for "entry P of beam b extended by c" (:123-134).  Word model: the window ends with
  // the word P spells (cached when P was created).  Character model: the window ends with c itself.
  CTC_HD double lm_window(const Beam &b, int P, int c) const {
    if (!lm_char_()) return mk_f64(b.spc_lo[P], b.spc_hi[P]);
    uint32_t st = (uint32_t)b.lmst[P];
    int cl = b.lmcl[P];
    return lm_cond_(&st, &cl, lm->label_word[c]);
  }
  // may entry P be extended by c at all?  path_trie.cpp:59-70: only along the dictionary (word models)
  CTC_HD bool lm_allows(const Beam &b, int P, int c) const {
    if (lm_char_()) return true;
    if (CTC_RARE(lm_wide_())) return ctclm::dict_find_wide(*lm, (uint32_t)b.dmlo[P], (uint32_t)b.dmhi[P], c) >= 0;
    return c < 32 ? (((uint32_t)b.dmlo[P] >> c) & 1u) != 0u : (((uint32_t)b.dmhi[P] >> (c - 32)) & 1u) != 0u;
  }
  // The LM fields of a prefix: `from` = the entry it copies them from (self) or hangs off (child via label c >= 0).
  CTC_HD void lm_emit(const Beam &src, int from, int c, const Beam &dst, int k) const {
    uint32_t st = (uint32_t)src.lmst[from];
    int cl = src.lmcl[from];
    double acc = mk_f64(src.acc_lo[from], src.acc_hi[from]);
    if (c < 0) {  // the entry stays: everything carries over
      dst.lmst[k] = (int)st; dst.lmcl[k] = cl; dst.acc_lo[k] = src.acc_lo[from]; dst.acc_hi[k] = src.acc_hi[from];
      dst.dn[k] = src.dn[from]; dst.dmlo[k] = src.dmlo[from]; dst.dmhi[k] = src.dmhi[from]; dst.dfc[k] = src.dfc[from];
      dst.spc_lo[k] = src.spc_lo[from]; dst.spc_hi[k] = src.spc_hi[from]; dst.spst[k] = src.spst[from]; dst.spcl[k] = src.spcl[from];
      return;
    }
    if (lm_char_()) {
      acc += lm_cond_(&st, &cl, lm->label_word[c]);  // Scorer::get_log_prob sums the same windows (scorer.cpp:111-120)
      dst.dn[k] = 0; dst.dmlo[k] = 0; dst.dmhi[k] = 0; dst.dfc[k] = 0; dst.spc_lo[k] = 0; dst.spc_hi[k] = 0; dst.spst[k] = 0; dst.spcl[k] = 0;
    } else if (c == lm_space) {  // a word is complete: its window joins the sum, the speller restarts at the root (path_trie.cpp:83-92)
      acc += mk_f64(src.spc_lo[from], src.spc_hi[from]);
      st = (uint32_t)src.spst[from];
      cl = src.spcl[from];
      dst.dn[k] = 0; dst.dmlo[k] = root_lo; dst.dmhi[k] = root_hi; dst.dfc[k] = root_fc;  // (no word ends at the root)
      put_f64(ctclm::kOovScore, &dst.spc_lo[k], &dst.spc_hi[k]);
      dst.spst[k] = 0; dst.spcl[k] = 0;
    } else {
      // The new node's record and the window "..., word spelled so far" need table look-ups (dependent global accesses):
      // the entry is left PENDING (dfc = -1) and resolved by lm_resolve_entry() while the next frame's phase A runs --
      // nothing reads these fields before that frame's phase B.
      ctclm::DictNode pin;
      pin.mask_lo = (uint32_t)src.dmlo[from]; pin.mask_hi = (uint32_t)src.dmhi[from]; pin.first_child = (uint32_t)src.dfc[from]; pin.word = 0;
      const uint32_t node = CTC_RARE(lm_wide_()) ? pin.first_child + (uint32_t)ctclm::dict_find_wide(*lm, pin.mask_lo, pin.mask_hi, c) : ctclm::dict_child(pin, c);
      dst.dn[k] = (int)node; dst.dfc[k] = kLmPending;
    }
    dst.lmst[k] = (int)st; dst.lmcl[k] = cl;
    put_f64(acc, &dst.acc_lo[k], &dst.acc_hi[k]);
  }

  // Second half of lm_emit for a pending entry k of beam b, given its dictionary node's record: the node's arcs (the gate
  // of path_trie.cpp:59-70) and the window "..., word spelled so far" -- used when a space follows
  // (ctc_beam_search_decoder.cpp:128) and for the last word in decode() (:173-185).
  CTC_HD void lm_resolve_entry(const Beam &b, int k, const ctclm::DictNode &info) const {
    double cond = ctclm::kOovScore;
    uint32_t st2 = 0;
    int cl2 = 0;
    if (info.word != ctclm::kNoWord) {
      st2 = (uint32_t)b.lmst[k];
      cl2 = b.lmcl[k];
      bool missed = false;
      cond = lm_cond_(&st2, &cl2, info.word, &missed);
      if (CTC_RARE(missed)) return;  // (callback scorer: not cached yet -- the entry stays pending, the frame will not be committed)
    }
    b.dmlo[k] = (int)info.mask_lo; b.dmhi[k] = (int)info.mask_hi; b.dfc[k] = (int)info.first_child;
    put_f64(cond, &b.spc_lo[k], &b.spc_hi[k]);
    b.spst[k] = (int)st2; b.spcl[k] = cl2;
  }
  // Every pending entry of the current beam, with a barrier: before the beam is parked (streams) or read by finish().
  CTC_HD void lm_resolve_pending() {
    if (!LM) return;
    select_beams();
    const Beam b = w.cur;
    for (int k = x.tid(); k < st_n; k += x.nt())
      if (b.dfc[k] == kLmPending) lm_resolve_entry(b, k, lm->dict[b.dn[k]]);
    x.sync();
  }

  // ---- danger mode ---------------------------------------------------------------------------------------------------
  // log_sum_exp (decoder_utils.h:47-54) returns "the other argument" when one is <= -FLT_MAX, so for the pair
  // {-FLT_MAX, -inf} its value depends on the ORDER of the arguments.  A prefix whose parent is in the beam receives two
  // contributions to log_prob_nb_cur per frame -- the repeat of its own last label (ctc_beam_search_decoder.cpp:103-106)
  // and the extension of its parent (:108-139) -- in the order in which the two sit in the reference's `prefixes` array:
  // the permutation std::nth_element (:150-154) left there (with a scorer: the frame's std::sort, :75-76, on top of it).
  // That order matters only once a -inf has come into play, which needs a log-probability that is -inf / NaN or so large
  // that a sum can overflow: lp_bad().  The row of frame t+1 is in registers while frame t is decoded (it is prefetched):
  // the threads that hold it look at it after phase B (step(): note_next_row) and raise the sticky flag VAR_DANGER, which
  // phase C of the SAME frame already sees: from that frame on ("danger mode", sticky for the utterance / stream) every
  // frame replays std::nth_element exactly and records the array order -- fin[p] = beam entry at position p, apos[j] =
  // position of entry j -- so the frame that first meets a bad row finds the order its predecessor left.  (The first row
  // of a launch is checked where it is staged; the order it needs is the one init() / load_state() provide.)  Ordinary
  // inputs (log-softmax outputs, probabilities) never get here and pay one compare per prefetched value.
  CTC_HD static bool lp_bad(float v) { return !((ctcmath::f32_to_bits(v) & 0x7fffffffu) <= 0x60ad78ecu); }  // !(|v| <= 1e20f), NaN included
  CTC_HD bool lm_params_extreme() const {  // (alpha, beta so large that scores can overflow without any bad row)
    if (!LM) return false;
    const double a = lm_alpha, bt = lm_beta;
    return !(a > -1e15 && a < 1e15 && bt > -1e15 && bt < 1e15);
  }
  CTC_HD void note_lp(float v) const {
    if (CTC_RARE(lp_bad(v))) w.vars[VAR_DANGER] = 1;
  }
  // The order std::nth_element leaves N candidates (w.skey / w.sinfo over S slots) in: ord[] = slots of the K survivors by
  // array position (w.surv + 2K).  Returns true when the replay used the block of the slot keys as its staging area
  // (w.skey[] is then garbage).
  CTC_HD bool nth_element_order(int S, int N, int K, const SlotCtx *lz = nullptr) {
    const int tid = x.tid(), nt = x.nt();
    int *rk = w.surv + K, *ord = w.surv + 2 * K;
    if (LAZY) {
      build_ek_lazy(*lz, S);
    } else {  // the candidates in DFS (= slot) order: (48-bit key, slot) of every slot that is not a hole
      const uint32_t *sinfo = w.sinfo;
      const uint32_t *skey = w.skey;
      Ek *ek = ekp();
      x.compact_slots_to(S, [=](int s) -> bool { return info_type(sinfo[s]) != T_HOLE; },
                         [=](int r, int s) { ek[r] = EO::make(key48(skey[s], sinfo[s]), s); });
    }
    FullSyncView<X> fx{x};
    if (FARREP) fx.sync();  // the list is in HBM scratch
    const bool keys_gone = replay_nth_element(fx, N, K);
    for (int k = tid; k < K; k += nt) { rk[k] = 0; ord[k] = EO::slot(ekp()[k]); }  // ord: nth_element order
    x.sync();
    return keys_gone;
  }
  // K packed (key, entry) words for the std::sort replays: in LDS (the block of the slot keys is idle between frames and
  // after the last one) wherever the slot keys are
  CTC_HD uint64_t *sort_scratch() const { return FARREP && w.stage_skey ? reinterpret_cast<uint64_t *>(w.skey) : w.ek; }

  // With a scorer the frame starts by sorting `prefixes` (ctc_beam_search_decoder.cpp:75-76): the contributions of this
  // frame then come in THAT order.  apos[] := position after the sort (fin[] keeps the order std::nth_element left).
  CTC_HD void lm_sorted_order(const Beam &b, int n) {
    const int tid = x.tid(), nt = x.nt();
    uint64_t *pk = sort_scratch();
    const int *fin = fin_cur();
    for (int p = tid; p < n; p += nt) {
      const int a = fin[p];
      pk[p] = (key48(ord_f32(b.score[a]), mk_info(b.ch[a], 0, 0)) << 16) | (uint64_t)a;
    }
    x.sync();
    sort_like_std(pk, n, [](uint64_t a, uint64_t c) { return (a >> 16) > (c >> 16); });
    for (int p = tid; p < n; p += nt) w.apos[(int)(pk[p] & 0xFFFFu)] = p;
    for (int i = tid; i < kBins + kBins / 16; i += nt) w.bins[i] = 0;  // (the sort's task lists live in the histogram)
    if (tid == 0) w.vars[VAR_DANGER] = 1;                              // (... and its counters in the flag's 16-byte group)
    x.sync();
  }

  // ctc_beam_search_decoder.cpp:43-44 : root prefix, score = log_prob_b_prev = 0
  CTC_HD void init() {
    st_par = 0;
    select_beams();
    if (x.tid() == 0) {
      Beam &b = w.cur;
      b.node[0] = 0; b.par[0] = -1; b.ch[0] = -1; b.dep[0] = 0; b.lcp[0] = -1;
      b.via[0] = -1; b.viaanc[0] = -1; b.viach[0] = -1; b.up[0] = 0;
      b.bprev[0] = 0.f; b.nbprev[0] = CTC_NEG_MAX; b.score[0] = 0.f; b.lpc[0] = CTC_NEG_MAX;
      if (LM) {  // ctc_beam_search_decoder.cpp:46-52: the root starts at the dictionary's start state
        const ctclm::DictNode info = lm->dict[0];
        b.lmst[0] = (int)lm->s0; b.lmcl[0] = lm->clean0; b.acc_lo[0] = 0; b.acc_hi[0] = 0;
        b.dn[0] = 0; b.dmlo[0] = (int)info.mask_lo; b.dmhi[0] = (int)info.mask_hi; b.dfc[0] = (int)info.first_child;
        put_f64(ctclm::kOovScore, &b.spc_lo[0], &b.spc_hi[0]); b.spst[0] = 0; b.spcl[0] = 0;
      }
      PoolNode r; r.parent = -1; r.cht = PoolNode::pack(-1, 0); r.lpc = CTC_NEG_MAX;
      pool[0] = r;
      pool_up[0] = 0;
      pool_thi[0] = 0;
      w.vars[VAR_STATUS] = ST_OK;
      reset_pvars(pvars(0));
      reset_pvars(pvars(1));
      w.vars[VAR_DANGER] = lm_params_extreme() ? 1 : 0;
      w.vars[VAR_INTO] = 0;
      w.vars[VAR_QSTAT] = 0;
      w.vars[VAR_LMMISS] = 0;
      w.vars[VAR_LMQ] = 0;
      w.apos[0] = 0; w.fin[0] = 0;
    }
    // (behind the scorer hook an utterance is parked -- fin[] saved -- at any frame, long before a frame has had a reason to write it:
    //  both copies start as the identity instead of whatever the memory held)
    if (lm_cb)
      for (int i = x.tid(); i < 2 * d.K; i += x.nt()) w.fin[i] = i < d.K ? i : i - d.K;
    st_n = 1; st_pool = 1; st_wlog = 32;  // first select looks at the whole key range
    st_maxkey = ord_f32(0.f);
    st_minkey = ord_f32(0.f);
    for (int i = x.tid(); i < kBins + kBins / 16; i += x.nt()) w.bins[i] = 0;
    for (int i = x.tid(); i < 2 * d.K; i += x.nt()) w.hit[i] = 0;
    for (int i = x.tid(); i < 2 * d.K; i += x.nt()) { w.ancbuf[i] = -1; w.acntbuf[i] = 0; }
    if (d.use_rank_table)
      for (int c = x.tid(); c < d.V; c += x.nt()) w.rank_of[c] = -1;
    zero_key_tail(0);
    if (kSpec) x.sync();  // (thread 0's resets above come before the ones spec_reset adds)
    spec_reset(0);
    x.sync_full();
  }

  // Restore / park the beam of a stream.  Transient per-frame scratch is re-initialised exactly as init() does.
  CTC_HD void load_state(const StreamState &ss) {
    const int tid = x.tid(), nt = x.nt(), K = d.K;
    st_n = x.uni(ss.hdr[SH_N]); st_pool = x.uni(ss.hdr[SH_POOL]); st_wlog = x.uni(ss.hdr[SH_WLOG]);
    st_maxkey = (uint32_t)x.uni(ss.hdr[SH_MAXKEY]);
    st_minkey = (uint32_t)x.uni(ss.hdr[SH_MINKEY]);
    st_par = 0;
    select_beams();
    Beam &b = w.cur;
    int *ia[9] = {b.node, b.par, b.ch, b.dep, b.lcp, b.via, b.viaanc, b.viach, b.up};
    float *fa[4] = {b.bprev, b.nbprev, b.score, b.lpc};
    if (lm_cb)  // (the part of the order array the parked beam does not reach: defined, as after init())
      for (int i = st_n + tid; i < K; i += nt) fin_cur()[i] = i;
    for (int i = tid; i < st_n; i += nt) {
      for (int a = 0; a < 9; ++a) ia[a][i] = ss.arrays[a * K + i];
      for (int a = 0; a < 4; ++a) fa[a][i] = ctcmath::bits_to_f32((uint32_t)ss.arrays[(9 + a) * K + i]);
      const int f = ss.arrays[13 * K + i];  // the last frame of every chunk records the order std::nth_element left:
      fin_cur()[i] = f;                      // array position i holds beam entry f
      if (!lm_cb || (unsigned)f < (unsigned)K) w.apos[f] = i;  // (hook: a state parked before any frame wrote the order -- see init())
      if (LM) {
        int *la[kBeamArraysLm] = {b.lmst, b.lmcl, b.acc_lo, b.acc_hi, b.dn, b.dmlo, b.dmhi, b.dfc, b.spc_lo, b.spc_hi, b.spst, b.spcl};
        for (int a = 0; a < kBeamArraysLm; ++a) la[a][i] = ss.arrays[(kStateArrays + a) * K + i];
      }
    }
    if (tid == 0) {
      w.vars[VAR_STATUS] = ST_OK;
      reset_pvars(pvars(0));
      reset_pvars(pvars(1));
      w.vars[VAR_DANGER] = ss.hdr[SH_DANGER];
      w.vars[VAR_INTO] = 0;
      w.vars[VAR_QSTAT] = 0;
      w.vars[VAR_LMMISS] = 0;
      w.vars[VAR_LMQ] = 0;
    }
    for (int i = tid; i < kBins + kBins / 16; i += nt) w.bins[i] = 0;
    for (int i = tid; i < 2 * K; i += nt) { w.hit[i] = 0; w.ancbuf[i] = -1; w.acntbuf[i] = 0; }
    if (d.use_rank_table)
      for (int c = tid; c < d.V; c += nt) w.rank_of[c] = -1;
    zero_key_tail(0);
    if (kSpec) x.sync();
    spec_reset(x.uni(ss.hdr[SH_FRAMES]), ss.hdr);
    x.sync_full();
  }
  // The shape statistic the host chooses phase A1's build by (X::kQuarters): how many entries of the beam the launch ends with
  // have descendants in it -- one or two on random rows, about twenty on blank-dominated ones.  Once per launch, behind the last
  // frame.  (Counted per frame -- in phase A1, where the number falls out of the search, or on an idle wave of the emission -- it
  // cost random rows 1-2 % of the frame.)
  CTC_HD void shape_stat() {
    if (!(SMALLV && !LM)) return;
    select_beams();
    const Beam &b = w.cur;
    int cnt = 0;
    for (int j = x.tid(); j + 1 < st_n; j += x.nt()) cnt += b.lcp[j + 1] >= b.dep[j] ? 1 : 0;
    if (cnt) x.atomic_add(&w.vars[VAR_QSTAT], cnt);
    x.sync();
  }
  CTC_HD void save_state(const StreamState &ss, int frames) {
    const int tid = x.tid(), nt = x.nt(), K = d.K;
    select_beams();
    const Beam &b = w.cur;
    const int *ia[9] = {b.node, b.par, b.ch, b.dep, b.lcp, b.via, b.viaanc, b.viach, b.up};
    const float *fa[4] = {b.bprev, b.nbprev, b.score, b.lpc};
    for (int i = tid; i < st_n; i += nt) {
      for (int a = 0; a < 9; ++a) ss.arrays[a * K + i] = ia[a][i];
      for (int a = 0; a < 4; ++a) ss.arrays[(9 + a) * K + i] = (int)ctcmath::f32_to_bits(fa[a][i]);
      ss.arrays[13 * K + i] = fin_cur()[i];
      if (LM) {
        const int *la[kBeamArraysLm] = {b.lmst, b.lmcl, b.acc_lo, b.acc_hi, b.dn, b.dmlo, b.dmhi, b.dfc, b.spc_lo, b.spc_hi, b.spst, b.spcl};
        for (int a = 0; a < kBeamArraysLm; ++a) ss.arrays[(kStateArrays + a) * K + i] = la[a][i];
      }
    }
    if (tid == 0) {
      if (kSpec) {  // (the window state is kept in LDS by one thread: spec_learn; one frame stale at most, which
                    //  only ever costs time -- the select is exact for any window)
        st_wlog = w.vars[VAR_SPEC + SP_WLOG];
        st_maxkey = (uint32_t)pvars(frames - 1)[P_NMAXKEY];
      }
      ss.hdr[SH_FRAMES] = frames; ss.hdr[SH_N] = st_n; ss.hdr[SH_POOL] = st_pool; ss.hdr[SH_WLOG] = st_wlog;
      ss.hdr[SH_MAXKEY] = (int)st_maxkey; ss.hdr[SH_MINKEY] = (int)st_minkey;
      ss.hdr[SH_DANGER] = w.vars[VAR_DANGER];
      ss.hdr[SH_SP_PRED] = kSpec ? w.vars[VAR_SPEC + SP_PRED] : 0;  // (kernels without the speculative select write 0: a later
      ss.hdr[SH_SP_GAP] = kSpec ? w.vars[VAR_SPEC + SP_GAP] : 0;    //  chunk on one that has it starts without a prediction)
      ss.hdr[SH_SP_MARGIN] = kSpec ? w.vars[VAR_SPEC + SP_MARGIN] : 0;
    }
  }

  CTC_HD int rank_of_char(const StepIn &in, int c) const {
    if (c < 0) return -1;
    if (IDENT) return c < in.Vc ? c : -1;
    if (kRankEpoch) {
      const uint32_t v = (uint16_t)w.rank_of[c];
      return (v >> 6) == ((uint32_t)in.t & 1023u) ? (int)(v & 63u) : -1;
    }
    return w.rank_of[c];
  }

  // ---- LAZY info words ----------------------------------------------------------------------------------------------
  CTC_HD int cand_char(int rn, int brank) const {  // label of the rn-th non-blank candidate
    const int r = rn + ((brank >= 0 && rn >= brank) ? 1 : 0);
    return IDENT ? r : w.cch[r];
  }
  CTC_HD SlotCtx slot_ctx(const Beam &pb, int n, int Vnb, int brank) const {
    SlotCtx c;
    c.pb = &pb; c.n = n; c.Vnb = Vnb; c.brank = brank;
    c.vmagic = Vnb > 0 ? 0xFFFFFFFFull / (uint32_t)Vnb + 1ull : 0ull;
    return c;
  }
  // The info word of slot sl (not a hole), from the layout alone: open(j) = [revived | self]; between open(j) and
  // open(j + 1) lie the groups of brand-new children of the entries whose subtree ends at j + 1 -- j itself, then its
  // nearest in-beam ancestor, and so on up (close(i) = 2 e_i + Vnb (e_i - 1 - a_i), beam_core.h header).
  CTC_HD uint32_t info_of_slot(const SlotCtx &c, int sl) const {
    int lo = 0, hi = c.n - 1;
    while (lo < hi) {  // largest j with ostart[j] <= sl
      const int mid = (lo + hi + 1) >> 1;
      if (w.ostart[mid] <= sl) lo = mid; else hi = mid - 1;
    }
    const int j = lo, off = sl - w.ostart[j];
    if (off == 0) return mk_info(c.pb->viach[j], T_REVIVED, j);
    if (off == 1) return mk_info(c.pb->ch[j], T_SELF, j);
    const uint32_t q = (uint32_t)(off - 2);
    const int g = HUGE ? (int)(q / (uint32_t)c.Vnb) : (int)(((uint64_t)q * c.vmagic) >> 32);
    const int rn = (int)q - g * c.Vnb;
    int i = j;
    for (int h = 0; h < g; ++h) i = w.anc[i];
    return mk_info(cand_char(rn, c.brank), T_CHILD, i);
  }
  // LAZY: the DFS-ordered candidate list of the exact replay, w.ek[r] = (48-bit key, slot) of the r-th slot that is not a
  // hole (a hole has key 0), straight from the layout: one bit per slot says "candidate", a prefix over the bitmap's words
  // gives every candidate its rank, and each (entry, label) pair writes its own element -- the info words never exist.
  // Uses the select's bitmap and histogram as scratch (the caller zeroes the histogram again if a frame follows directly).
  CTC_HD void build_ek_lazy(const SlotCtx &c, int S) {
    const int tid = x.tid(), nt = x.nt();
    const uint32_t *skey = w.skey;
    uint32_t *bm = w.bitmap, *wpre = w.wpre;
    Ek *ek = ekp();
    x.mark_slots(S, bm, [=](int sl) -> bool { return skey[sl] != 0u; });
    x.sync();
    const int nw = (S + 63) / 64;  // <= 1024 for S < 65536: fits the histogram's 1024 + 64 words (more slots: wpre is in HBM)
    for (int i = tid; i <= nw; i += nt) wpre[i] = i < nw ? (uint32_t)(__builtin_popcount(bm[2 * i]) + __builtin_popcount(bm[2 * i + 1])) : 0u;
    x.sync();
    x.scan_excl(wpre, nw + 1);
    auto put = [&](int sl, int ch) {
      const uint32_t lo = bm[2 * (sl >> 6)], hi = bm[2 * (sl >> 6) + 1];
      const int bit = sl & 63;
      const uint32_t mlo = bit >= 32 ? 0xFFFFFFFFu : ((1u << bit) - 1u), mhi = bit > 32 ? ((1u << (bit - 32)) - 1u) : 0u;
      const int r = (int)wpre[sl >> 6] + __builtin_popcount(lo & mlo) + __builtin_popcount(hi & mhi);
      ek[r] = EO::make(key48(skey[sl], mk_info(ch, 0, 0)), sl);
    };
    for (int j = tid; j < c.n; j += nt) {
      const int s0 = w.ostart[j];
      if (skey[s0] != 0u) put(s0, c.pb->viach[j]);
      put(s0 + 1, c.pb->ch[j]);
    }
    for (int idx = tid; idx < c.n * c.Vnb; idx += nt) {
      const int i = HUGE ? idx / c.Vnb : (int)(((uint64_t)(uint32_t)idx * c.vmagic) >> 32), rn = idx - i * c.Vnb;
      const int sl = w.cstart[i] + rn;
      if (skey[sl] != 0u) put(sl, cand_char(rn, c.brank));
    }
    x.sync();
  }

  // which word of w.hit holds bit `bit` (non-blank candidate number) of entry i's existing-children mask: two words per
  // entry; the fixed-layout class (at most 31 non-blank candidates) uses ONE word per entry, indexed like every other
  // per-entry array (the scoring loop then advances one address for all of them)
  CTC_HD static int hit_word(int i, int bit) { return SMALLV == 1 ? i : 2 * i + (bit >> 5); }

  // log_p of extending beam entry P with character c (ctc_beam_search_decoder.cpp:110-118)
  CTC_HD float child_logp(int P, int c, float lp) const {
    const Beam &b = w.cur;
    if (c == b.ch[P]) return b.bprev[P] > CTC_NEG_MAX ? lp + b.bprev[P] : CTC_NEG_MAX;
    return lp + b.score[P];
  }

  CTC_HD uint64_t slot_key48(int s) const { return key48(w.skey[s], w.sinfo[s]); }

  // ------------------------------------------------------------------------------------------------------ select
  // Window of the first histogram round: [lo, 2^32) in score-key units, buckets of 2^shift keys, top bucket open.
  struct Window { uint32_t lo; int shift; };
  CTC_HD Window first_window() const {
    // width = 2^st_wlog keys ending at the previous best key (st_wlog is kept in [kBinsLog, 32]); all of it fits 32 bits
    CTC_ASSUME(st_wlog >= kBinsLog && st_wlog <= 32);
    Window wd;
    wd.lo = 1u;
    wd.shift = st_wlog - kBinsLog;
    if (st_wlog < 32) {
      const uint32_t width = 1u << st_wlog;
      if (width <= st_maxkey) wd.lo = st_maxkey - width + 1u;
    }
    return wd;
  }
  // ... and, in the wide-beam layouts, its place in the pre-list when it reaches the predicted threshold (thr >= wd.lo)
  CTC_HD void hist_add(const Window &wd, uint32_t key, const HotPre &hp, int slot) const {
    hist_add(wd, key);
    if (kHotPre && key >= hp.thr) {
      const int p = x.atomic_add(&w.vars[VAR_HOTN], 1);
      if (p < hp.cap) { hp.key[p] = key; hp.slot[p] = slot; }
    }
  }
  CTC_HD void hist_add(const Window &wd, uint32_t key) const {  // first-round histogram contribution of one candidate
    if (key >= wd.lo) {
      const uint32_t bk = (key - wd.lo) >> wd.shift;
      const int bb = bk < (uint32_t)(kBins - 1) ? (int)bk : kBins - 1;
      x.atomic_add(&w.bins[bb], 1);  // (one LDS atomic per candidate: a second, coarse level of counters kept by atomics as
                                     //  well cost 2.5 % of the frame -- many lanes hit the same few coarse words)
    }
  }

  // start of an utterance / of a stream's chunk: no prediction yet, empty hot list, clear survivor bitmap
  CTC_HD void spec_reset(int t0, const int *hdr = nullptr) {
    if (!kSpec) return;
    if (x.tid() == 0) {
      int *sp = w.vars + VAR_SPEC;
      sp[SP_THR] = -1; sp[SP_BEST] = 0; sp[SP_MARGIN] = (int)ctcmath::f32_to_bits(0.125f); sp[SP_GAP] = 0; sp[SP_PRED] = 0;
      if (hdr && hdr[SH_SP_PRED] == 1) {  // a stream's chunk continues with the prediction the previous chunk ended on
        sp[SP_PRED] = 1; sp[SP_GAP] = hdr[SH_SP_GAP]; sp[SP_MARGIN] = hdr[SH_SP_MARGIN];
      }
      sp[SP_WLOG] = st_wlog; sp[SP_ANCHOR] = (int)st_maxkey;
      w.vars[VAR_G] = 0;  // (the hot list's length is counted in the select's result group: one read gives it and the danger flag)
      w.vars[VAR_E] = 0;
      // (the first frame finds "the previous frame's best key" where every later one does: in the counters of the other parity)
      pvars(t0 - 1)[P_NMAXKEY] = (int)st_maxkey;
    }
    for (int i = x.tid(); i < kHotCap + 64; i += x.nt()) w.list[i] = 0u;
    for (int i = x.tid(); i < kHotCap; i += x.nt()) w.hotge[i] = 0;
    for (int i = x.tid(); i < 2 * ((kClsSlots + 63) / 64); i += x.nt()) w.bitmap[i] = 0u;
  }
  // The histogram of the frame's keys, for the frames the speculative select hands back: what phase B would have counted
  // (the window is kept in LDS by the thread that keeps the prediction: spec_learn).
  CTC_HD void rehistogram(int S, Window &wd) {
    const int tid = x.tid(), nt = x.nt();
    int sv[4];
    x.uni4(&w.vars[VAR_SPEC], sv);
    st_wlog = sv[SP_WLOG]; st_maxkey = (uint32_t)sv[SP_ANCHOR];
    wd = first_window();
    for (int i = tid; i < kBins; i += nt) w.bins[i] = 0;
    x.sync();
    for (int s = tid; s < S; s += nt) hist_add(wd, w.skey[s]);
    x.sync();
  }

  // K-th largest SCORE key (32 bit) among the S slots (holes have key 0).  The first-round histogram (window
  // first_window()) has already been accumulated in bins[].  Leaves tau (VAR_TAU), G = #keys > tau, E = #keys == tau.
  // Fast path (the first histogram round already isolates a small bucket): the pass that lists the bucket's keys also
  // records, one bit per slot, every key ABOVE the bucket; the ranking threads add the bucket's own survivors; the
  // caller then only expands that bitmap (in slot = DFS order).  Returns true when the bitmap is complete (the caller
  // must still discard it if several candidates tie on the K-th SCORE).  Precondition: pv[P_LCOUNT] == 0.
  // The keys of one histogram bucket [b32, b32 + bspan] are listed and ranked exactly: leaves tau = the `want`-th largest
  // of them (VAR_TAU), G = gsum + #bucket keys above tau, E = #keys equal to tau.  `direct` (first round only): the
  // listing pass also records, one bit per slot, every key ABOVE the bucket, and the ranking threads add the bucket's
  // own survivors, so the caller only has to expand the bitmap.  inb = #keys in the bucket (<= kListCap).
  template <bool COMPACT = false>
  CTC_HD void rank_bucket(int S, int *pv, uint32_t b32, uint32_t bspan, bool direct, int want, int gsum, int inb) {
    const int tid = x.tid(), nt = x.nt();
    CTC_ASSUME(inb >= 1 && inb <= kListCap);
    // one pass over the slots: bucket members are listed (key offset + slot); bit s of the bitmap = key above the bucket
    x.template list_bucket<kTailZero, COMPACT>(S, w.skey, b32, bspan, direct, w.bitmap, w.list, w.lslot, &pv[P_LCOUNT]);
    for (int q = tid; q < 4; q += nt) w.list[inb + q] = 0;  // pad to a multiple of four, below every real entry
    x.sync();
    x.mark(14);
    // a long list (wide beams: up to kListCap keys share the bucket) is ranked by eight lanes per key, each comparing an
    // eighth of the list; a short one by one lane per key
    const bool wide_rank = !SMALLV && inb > 32 && nt >= 8 * kListCap;
    // (fixed-layout class at 1024 threads: four lanes per key, each comparing a quarter of the list -- some twenty keys share
    //  the bucket on ordinary input, and the ranking is a single-wave stage: five rounds of loads and compares become two)
    const bool quad_rank = SMALLV && !COMPACT && x.nt_is(1024);
    const int psh = wide_rank ? 3 : quad_rank ? 2 : 0;
    for (int q0 = tid; q0 < (wide_rank ? 8 * kListCap : inb << psh); q0 += nt) {
      const int q = q0 >> psh, part = q0 & ((1 << psh) - 1), stride = 4 << psh;
      const uint32_t mine = q < inb ? w.list[q] : 0u;
      int g = 0, e = 0;
      for (int r = 4 * part; r < inb; r += stride) {
        const uint32_t o0 = w.list[r], o1 = w.list[r + 1], o2 = w.list[r + 2], o3 = w.list[r + 3];
        g += (o0 > mine) + (o1 > mine) + (o2 > mine) + (o3 > mine);
        e += (o0 == mine) + (o1 == mine) + (o2 == mine) + (o3 == mine);
      }
      if (wide_rank) {
        g = x.sum8(g); e = x.sum8(e);
        if (part != 0 || q >= inb) continue;
      } else if (quad_rank) {
        g = x.sum4(g); e = x.sum4(e);
        if (part != 0) continue;
      }
      if (g < want && want <= g + e) {  // every holder of the K-th key writes the same values
        w.vars[VAR_TAU] = (int)(b32 + (mine - 1u)); w.vars[VAR_G] = gsum + g; w.vars[VAR_E] = e;
      }
      if (direct && g < want) {  // key >= tau: survives (unless equal scores straddle the boundary -- caller's business)
        const int sl = w.lslot[q];
        x.atomic_or(&w.bitmap[sl >> 5], 1u << (sl & 31));
      }
    }
    x.sync();
  }

  // The same with the lists' places as arguments (the wide-beam layouts' pre-list, kHotPre; the wide-list experiment) -- a function of
  // its own: the north-star kernels keep rank_bucket's text, and with it their machine code (tools/kernel_hashes.sh; a reordered
  // condition in select_kth and two pointer arguments here had cost configs[1] 2.8 %).
  // lst / lsl: where the bucket's keys and slots are listed -- w.list / w.lslot (kListCap entries), or, for a crowded bucket of a beam
  // without a scorer, the block of the NEXT beam (wide_list_cap() entries: nothing lives there between two emissions).
  template <bool COMPACT = false>
  CTC_HD void rank_bucket_ext(int S, int *pv, uint32_t b32, uint32_t bspan, bool direct, int want, int gsum, int inb, uint32_t *lst, int *lsl,
                          const HotPre *hot, int hot_n) {
    const int tid = x.tid(), nt = x.nt();
    CTC_ASSUME(inb >= 1);
    if (kHotPre && hot != nullptr) {
      // the same listing from phase B's pre-list (every key >= hot->thr <= b32 is in it): the bitmap starts empty, a key above the
      // bucket sets its slot's bit, a member of the bucket takes a place in the list
      const int nw = 2 * ((S + 63) / 64);
      for (int i = tid; i < nw; i += nt) w.bitmap[i] = 0u;
      x.sync();
      for (int q = tid; q < hot_n; q += nt) {
        const uint32_t k = hot->key[q], dk = k - b32;
        const int sl = hot->slot[q];
        if (k >= b32) {
          if (dk <= bspan) {
            const int p = x.atomic_add(&pv[P_LCOUNT], 1);
            lst[p] = dk + 1u;
            lsl[p] = sl;
          } else {
            x.atomic_or(&w.bitmap[sl >> 5], 1u << (sl & 31));
          }
        }
      }
    } else
    // one pass over the slots: bucket members are listed (key offset + slot); bit s of the bitmap = key above the bucket
    x.template list_bucket<kTailZero, COMPACT>(S, w.skey, b32, bspan, direct, w.bitmap, lst, lsl, &pv[P_LCOUNT]);
    for (int q = tid; q < 4; q += nt) lst[inb + q] = 0;  // pad to a multiple of four, below every real entry
    x.sync();
    x.mark(14);
    // a long list (wide beams: up to kListCap keys share the bucket, a crowded bucket's wide list more) is ranked by eight lanes per
    // key, each comparing an eighth of the list; a short one by one lane per key
    const bool wide_rank = !SMALLV && inb > 32 && nt >= 8 * kListCap;
    // (fixed-layout class at 1024 threads: four lanes per key, each comparing a quarter of the list -- some twenty keys share
    //  the bucket on ordinary input, and the ranking is a single-wave stage: five rounds of loads and compares become two)
    const bool quad_rank = SMALLV && !COMPACT && x.nt_is(1024);
    const int psh = wide_rank ? 3 : quad_rank ? 2 : 0;
    for (int q0 = tid; q0 < (wide_rank ? 8 * (inb > kListCap ? inb : kListCap) : inb << psh); q0 += nt) {
      const int q = q0 >> psh, part = q0 & ((1 << psh) - 1), stride = 4 << psh;
      const uint32_t mine = q < inb ? lst[q] : 0u;
      int g = 0, e = 0;
      for (int r = 4 * part; r < inb; r += stride) {
        const uint32_t o0 = lst[r], o1 = lst[r + 1], o2 = lst[r + 2], o3 = lst[r + 3];
        g += (o0 > mine) + (o1 > mine) + (o2 > mine) + (o3 > mine);
        e += (o0 == mine) + (o1 == mine) + (o2 == mine) + (o3 == mine);
      }
      if (wide_rank) {
        g = x.sum8(g); e = x.sum8(e);
        if (part != 0 || q >= inb) continue;
      } else if (quad_rank) {
        g = x.sum4(g); e = x.sum4(e);
        if (part != 0) continue;
      }
      if (g < want && want <= g + e) {  // every holder of the K-th key writes the same values
        w.vars[VAR_TAU] = (int)(b32 + (mine - 1u)); w.vars[VAR_G] = gsum + g; w.vars[VAR_E] = e;
      }
      if (direct && g < want) {  // key >= tau: survives (unless equal scores straddle the boundary -- caller's business)
        const int sl = lsl[q];
        x.atomic_or(&w.bitmap[sl >> 5], 1u << (sl & 31));
      }
    }
    x.sync();
  }

  // entries of the wide bucket list (keys, then slots, in the next beam's block: w.beam_blk bytes from w.nxt.node)
  CTC_HD int wide_list_cap() const {
    const long long c = (long long)(w.beam_blk / 8) - 8;
#ifndef CTC_HOT_CAP
#define CTC_HOT_CAP 3072
#endif
    return c >= CTC_HOT_CAP ? CTC_HOT_CAP : c <= kListCap ? 0 : (int)(c & ~3LL);
  }
  CTC_HD bool select_kth(int S, int K, int *pv, const Window &wd, const HotPre &hp) {
    const int tid = x.tid(), nt = x.nt();
    // -> [0] bucket b* holding the need-th largest key (-1: below the window), [1] #keys in buckets above b*,
    //    [2] #keys in the window, [3] #keys in b*.  Two-level: sums of 16 buckets locate the group, then the bucket.
    //    Ends with a barrier.  The histogram is cleared before another round; after the last one step() clears it
    //    (with the other per-frame resets, on waves that are idle while the next beam is emitted).
    x.find_bucket(w.bins, K, &w.vars[VAR_FB0]);
    int fb[4];
    x.uni4(&w.vars[VAR_FB0], fb);
    x.mark(13);
    // The usual outcome: the first histogram isolates a bucket with a handful of keys, several values wide.  Everything
    // about it fits 32-bit arithmetic (the window's top is the previous best key, below 2^32).
#if defined(CTC_EXP_WIDE_LIST)
    // Round 6, measured and left off (CTC_EXP_WIDE_LIST): a CROWDED bucket (more than kListCap keys: at beam 500 one frame in three --
    // float32 scores near 10^3 are 10^-4 apart, and the children of equal prefixes inherit equal scores; 256 keys on average) listed
    // once, as the usual bucket is, into a list that borrows the next beam's block, and ranked there by all sixteen waves, so that the
    // frame stays on this path.  Exact (tests/sweeps/cpu_wide_list_sweep.py: 9 912 configurations / 96 k such frames against the
    // reference, 0 mismatches) and SLOWER: configs[2] per-GPU shape 39.85 -> 41.2 ms.  The phase timers say why
    // (profiles/r06r / r06t_phase_beam500*.json): listing 256 members costs the gather pass +0.9 us per frame, while the path it
    // replaces -- a second histogram over the S slots, which ends on a single key value, and one marking pass -- was worth 0.1 us.
    // The select's share of this kernel (25 %) is the two passes over 15 500 slot keys every frame makes, not the crowded frames.
    const int wcap = (!SMALLV && !LM) ? wide_list_cap() : 0;
    const bool bucket_ok = fb[0] >= 0 && (wd.shift != 0 || fb[0] == kBins - 1);
    const bool wide = bucket_ok && fb[3] > kListCap && fb[3] <= wcap;
    if (CTC_USUAL(bucket_ok && (fb[3] <= kListCap || wide))) {
      const uint32_t b32 = wd.lo + ((uint32_t)fb[0] << wd.shift);
      const uint32_t bspan = fb[0] == kBins - 1 ? 0xFFFFFFFFu - b32 : (1u << wd.shift) - 1u;
      if (tid == 0) { x.count(EV_FAST_SELECT, 1); x.count(EV_BUCKET_KEYS, fb[3]); if (fb[3] == 1) x.count(EV_SINGLE_KEY, 1); if (wide) x.count(EV_WIDE_LIST, 1); }
      uint32_t *wl = reinterpret_cast<uint32_t *>(w.nxt.node);
      // (the pre-list holds every key at or above its threshold unless it overflowed: usable when the bucket starts at or above it)
      const int hot_n = kHotPre ? x.uni(w.vars[VAR_HOTN]) : 0;
      const bool from_hot = kHotPre && !wide && hp.thr != 0xFFFFFFFFu && hp.thr <= b32 && hot_n <= hp.cap;
      if (tid == 0 && from_hot) x.count(EV_HOT_LIST, 1);
      rank_bucket_ext(S, pv, b32, bspan, true, K - fb[1], fb[1], fb[3], wide ? wl : w.list, wide ? reinterpret_cast<int *>(wl + wcap + 4) : w.lslot,
                  from_hot ? &hp : nullptr, hot_n);
      return true;
    }
#else
    (void)hp;
    if (CTC_USUAL(fb[0] >= 0 && fb[3] <= kListCap && (wd.shift != 0 || fb[0] == kBins - 1))) {
      const uint32_t b32 = wd.lo + ((uint32_t)fb[0] << wd.shift);
      const uint32_t bspan = fb[0] == kBins - 1 ? 0xFFFFFFFFu - b32 : (1u << wd.shift) - 1u;
      if (tid == 0) { x.count(EV_FAST_SELECT, 1); x.count(EV_BUCKET_KEYS, fb[3]); if (fb[3] == 1) x.count(EV_SINGLE_KEY, 1); }
      if (kHotPre) {
        // (the pre-list holds every key at or above its threshold unless it overflowed: usable when the bucket starts at or above it)
        const int hot_n = x.uni(w.vars[VAR_HOTN]);
        const bool from_hot = hp.thr != 0xFFFFFFFFu && hp.thr <= b32 && hot_n <= hp.cap;
        if (tid == 0 && from_hot) x.count(EV_HOT_LIST, 1);
        rank_bucket_ext(S, pv, b32, bspan, true, K - fb[1], fb[1], fb[3], w.list, w.lslot, from_hot ? &hp : nullptr, hot_n);
      } else {
        rank_bucket(S, pv, b32, bspan, true, K - fb[1], fb[1], fb[3]);
      }
      return true;
    }
#endif
    // Rare: the K-th key lies below the window, the bucket is crowded, or it is a single key value.
    uint64_t lo = wd.lo, hi = (uint64_t)1 << 32;  // current key range [lo, hi)
    int shift = wd.shift;
    int need = K, gbase = 0;
    bool first = true;
    for (;;) {
      const int bstar = fb[0], above = fb[1], total = fb[2], inb = fb[3];
      uint64_t blo = 0, bhi = 0;
      bool again = true;
      if (tid == 0) x.count(EV_SLOW_ROUNDS, 1);
      if (bstar < 0) {  // the K-th key lies below the window: look at everything under it
        if (tid == 0) x.count(EV_SLOW_BELOW, 1);
        gbase += total; need -= total; hi = lo; lo = 1;
      } else {
        blo = lo + ((uint64_t)bstar << shift);
        bhi = (bstar == kBins - 1) ? hi : blo + ((uint64_t)1 << shift);
        if (bhi > hi) bhi = hi;  // keys at or above hi are already counted in gbase
        if (shift == 0 && bstar < kBins - 1) {  // the bucket is a single key value
          if (tid == 0) x.count(EV_SLOW_SINGLE, 1);
          if (tid == 0) { w.vars[VAR_TAU] = (int)(uint32_t)blo; w.vars[VAR_G] = gbase + above; w.vars[VAR_E] = inb; }
          x.sync();
          return false;
        }
        if (inb <= kListCap) again = false;
        else { if (tid == 0) x.count(EV_SLOW_CROWDED, 1); gbase += above; need -= above; lo = blo; hi = bhi; }  // too crowded: histogram the bucket itself
      }
      if (!again) {  // exact rank inside the bucket, on offsets from its base
        rank_bucket<true>(S, pv, (uint32_t)blo, (uint32_t)(bhi - blo - 1), first, need - above, gbase + above, inb);
        return first;
      }
      // another histogram round over [lo, hi)
      first = false;
      for (int i = tid; i < kBins; i += nt) w.bins[i] = 0;
      x.sync();  // the histogram has been cleared by every thread
      const uint64_t width = hi - lo;
      shift = width <= (uint64_t)kBins ? 0 : ceil_log2_u64(width) - kBinsLog;
      const uint32_t lo32 = (uint32_t)lo, span = (uint32_t)(hi - lo - 1);  // key in range <=> key - lo32 <= span
      for (int s0 = tid; s0 < S; s0 += 4 * nt) {  // (four keys per trip, requested together)
        uint32_t kk[4];
        for (int u = 0; u < 4; ++u) kk[u] = s0 + u * nt < S ? w.skey[s0 + u * nt] : 0u;
        for (int u = 0; u < 4; ++u) {
          const uint32_t k = kk[u], dk = k - lo32;
          if (k >= lo32 && dk <= span) {
            const uint32_t bk = dk >> shift;
            const int bb = bk < (uint32_t)(kBins - 1) ? (int)bk : kBins - 1;
            x.atomic_add(&w.bins[bb], 1);
          }
        }
      }
      x.sync();
      x.find_bucket(w.bins, need, &w.vars[VAR_FB0]);
      x.uni4(&w.vars[VAR_FB0], fb);
    }
  }

  // Several candidates share the K-th score: order them by character (prefix_compare, decoder_utils.cpp:126-131).
  // m of the E candidates with score key tau must survive.  Sets VAR_TAUC (smallest surviving inverted-character
  // code) and returns true when that cut is unambiguous; false when it would split a group of equivalent prefixes
  // (or the group is too large to rank here): the caller then replays std::nth_element.
  CTC_HD bool resolve_by_character(int S, uint32_t tau, int m, int E, int *pv, const SlotCtx *lz = nullptr) {
    const int tid = x.tid(), nt = x.nt();
    if (E > kListCap) return false;
    if (tid == 0) { pv[P_LCOUNT] = 0; w.vars[VAR_CUT] = 0; }
    x.sync();
    for (int s = tid; s < S; s += nt)
      if (w.skey[s] == tau) w.list[x.atomic_add(&pv[P_LCOUNT], 1)] = ((LAZY ? info_of_slot(*lz, s) : w.sinfo[s]) >> 16) + 1u;
    x.sync();
    for (int q = tid; q < E; q += nt) {
      const uint32_t mine = w.list[q];
      int g = 0, e = 0;
      for (int r = 0; r < E; ++r) {
        const uint32_t o = w.list[r];
        g += o > mine;
        e += o == mine;
      }
      if (g < m && m <= g + e) { w.vars[VAR_TAUC] = (int)(mine - 1u); w.vars[VAR_CUT] = (g + e == m) ? 1 : 2; }
    }
    x.sync();
    return x.uni(w.vars[VAR_CUT]) == 1;
  }

  // kLcpTable: st[l][i] = min lcp[i .. i + 2^l - 1] of the current beam (entries at or beyond n count as +inf), l = 0 .. 6, by ONE
  // wave (`lane` of 64; two entries per lane).  Level l reads level l - 1 at a distance of 2^(l-1): a wave's LDS accesses
  // execute in order, so no barrier is needed between the levels.
  CTC_HD void build_lcp_table(const Beam &b, int n, int lane) const {
    int *st = w.lcpst;
    const int i0 = lane, i1 = lane + 64;
    int v0 = i0 < n ? b.lcp[i0] : kIntMax, v1 = i1 < n ? b.lcp[i1] : kIntMax;
    st[i0] = v0; st[i1] = v1;
    x.wave_lds_fence();
#if defined(__clang__)
#pragma unroll
#endif
    for (int l = 1; l <= 6; ++l) {
      const int h = 1 << (l - 1);
      const int *p = st + (l - 1) * kSmallK;
      const int a = p[i0 + h];                                   // (i0 + h < 128)
      const int c = i1 + h < kSmallK ? p[i1 + h] : kIntMax;
      v0 = a < v0 ? a : v0; v1 = c < v1 ? c : v1;
      st[l * kSmallK + i0] = v0; st[l * kSmallK + i1] = v1;
      x.wave_lds_fence();
    }
  }
  // ... and the depth of the lowest common ancestor of entries ja and jb through it: one LDS round trip, no loop
  CTC_HD int lca_depth_tbl(int ja, int jb) const {
    const int lo = ja < jb ? ja : jb, hi = ja < jb ? jb : ja;
    const int len = hi - lo;
    const int l = 31 - __builtin_clz((unsigned)(len > 0 ? len : 1));
    const int *p = w.lcpst + l * kSmallK;
    const int same = w.cur.dep[ja];
    const int a = p[lo + 1 < kSmallK ? lo + 1 : kSmallK - 1], c = p[hi - (1 << l) + 1];
    const int m = a < c ? a : c;
    return len == 0 ? same : m;
  }
  // min of lcp[] over (lo, hi] of the CURRENT beam (depth of the lowest common ancestor of entries lo and hi)
  CTC_HD int lca_depth(int ja, int jb) const {
    const Beam &b = w.cur;
    if (ja == jb) return b.dep[ja];
    const int lo = ja < jb ? ja : jb, hi = ja < jb ? jb : ja;
    // eight independent loads per trip (indices past the range are clamped onto its last element: harmless for a
    // minimum); consecutive survivors are rarely further apart, so this is one LDS round trip
    int m = kIntMax;
    for (int i = lo + 1; i <= hi; i += 8) {
      const int i1 = i + 1 < hi ? i + 1 : hi, i2 = i + 2 < hi ? i + 2 : hi, i3 = i + 3 < hi ? i + 3 : hi;
      const int i4 = i + 4 < hi ? i + 4 : hi, i5 = i + 5 < hi ? i + 5 : hi, i6 = i + 6 < hi ? i + 6 : hi, i7 = i + 7 < hi ? i + 7 : hi;
      const int l0 = b.lcp[i], l1 = b.lcp[i1], l2 = b.lcp[i2], l3 = b.lcp[i3];
      const int l4 = b.lcp[i4], l5 = b.lcp[i5], l6 = b.lcp[i6], l7 = b.lcp[i7];
      const int m01 = l0 < l1 ? l0 : l1, m23 = l2 < l3 ? l2 : l3, m45 = l4 < l5 ? l4 : l5, m67 = l6 < l7 ? l6 : l7;
      const int ma = m01 < m23 ? m01 : m23, mb = m45 < m67 ? m45 : m67;
      const int mm = ma < mb ? ma : mb;
      m = mm < m ? mm : m;
    }
    return m;
  }

  // ------------------------------------------------------------------------------------------------ exact replay
  // One partition step of std::nth_element on v[first, last) (median of three to the front, unguarded Hoare partition of
  // the rest: stl_emul.h split_with_median_pivot) by the whole workgroup.  Lp, Rp: scratch for last - first + 1 positions
  // each.  Returns the cut.
  template <class XX>
  CTC_HD int hoare_round(XX &xx, Ek *v, int first, int last, LrT *Lp, LrT *Rp) {
    if (FARREP) return stlemu::hoare_round_parallel<true>(xx, v, first, last, [](const Ek &e) { return EO::key(e); }, Lp, Rp, &w.vars[VAR_CUT]);
    return stlemu::hoare_round_parallel_chunks(xx, v, first, last, [](const Ek &e) { return EO::key(e); }, Lp, Rp, &w.vars[VAR_CUT]);
  }

  // std::nth_element(begin, begin+K, end, prefix_compare) on the DFS-ordered candidate list (w.ek[0, N)).
  // The list lives in HBM scratch: `fx` = the execution policy with every barrier a full fence (it also waits for global
  // memory), used for the rounds that run there.  Once the range is short enough it moves into LDS -- the elements into
  // the block of the slot keys (dead from here on: the emission takes the survivors' keys from the list itself), the stop
  // positions into the block of the NEXT beam (nothing lives there until the emission); widest-beam layout, whose slot
  // keys are not in LDS: both into the next beam's block -- positions rebased to the range's start, and moves back when
  // the selection is done.  Returns true when the slot keys were overwritten.
  template <class XX>
  CTC_HD bool replay_nth_element(XX &fx, int N, int K) {
    const int tid = x.tid(), nt = x.nt();
    Ek *v = ekp();
    LrT *lr = lrp();
    auto before = [](const Ek &a, const Ek &c) { return EO::key(a) > EO::key(c); };
    int first = 0, last = N, depth = 2 * stlemu::floor_lg(N);
    if (!FARREP) {  // everything in LDS
      while (last - first > kSerialCut && depth > 0) {
        --depth;
        const int cut = hoare_round(x, v, first, last, lr, lr + N + 1);
        if (cut <= K) first = cut; else last = cut;
      }
      if (tid == 0) stlemu::introselect(v, first, K, last, depth, before);
      x.sync();
      return false;
    }
    Ek *sv = reinterpret_cast<Ek *>(w.nxt.node);
    int stage_cap = (int)((w.beam_blk - 16) / (sizeof(Ek) + 2 * sizeof(LrT)));  // an element + its place in the two position lists
    if (w.stage_skey) {
      sv = reinterpret_cast<Ek *>(w.skey);
      const int cap_e = d.S_max() / 2, cap_p = (int)(w.beam_blk / 4) - 2;
      stage_cap = cap_e < cap_p ? cap_e : cap_p;
    }
    while (last - first > kSerialCut && depth > 0 && last - first > stage_cap) {
      --depth;
      const int cut = hoare_round(fx, v, first, last, lr, lr + N + 1);
      if (cut <= K) first = cut; else last = cut;
    }
    if (last - first > kSerialCut && depth > 0) {
      const int m0 = last - first, base = first;
      LrT *sLp = w.stage_skey ? reinterpret_cast<LrT *>(w.nxt.node) : reinterpret_cast<LrT *>(sv + m0), *sRp = sLp + m0 + 1;
      for (int i = tid; i < m0; i += nt) sv[i] = v[base + i];
      x.sync();
      int f2 = 0, l2 = m0;
      const int K2 = K - base;
      while (l2 - f2 > kSerialCut && depth > 0) {
        --depth;
        const int cut = hoare_round(x, sv, f2, l2, sLp, sRp);
        if (cut <= K2) f2 = cut; else l2 = cut;
      }
      if (tid == 0) stlemu::introselect(sv, f2, K2, l2, depth, before);
      x.sync();
      for (int i = tid; i < m0; i += nt) v[base + i] = sv[i];
      fx.sync();
      return w.stage_skey != 0;
    }
    if (tid == 0) stlemu::introselect(v, first, K, last, depth, before);
    fx.sync();
    return false;
  }

  // kA1Overlap: phase A1 for the FIRST frame of a launch (every later one runs beside its predecessor's emission).
  CTC_HD void prime_a1(int t) {
    if (!kA1Overlap) return;
    select_beams();
    phase_a1(w.cur, st_n, w.ancbuf + (t & 1) * d.K, w.acntbuf + (t & 1) * d.K, x.group(), x.ngroups());
    x.sync();
  }
  // ---- A1 (a method of its own since round 6: see kA1Overlap): per beam entry, from the LCP array alone: the end of its
  // subtree range (first later entry whose LCP with its predecessor is shallower than the entry), found by a wave-wide
  // search; every entry then "paints" its proper descendants with itself, which leaves in anc[] the nearest in-beam
  // ancestor (max = innermost enclosing range) and in acnt[] the number of in-beam ancestors.  Cost is bounded even for
  // deeply nested beams.  `grp` of `ngr` groups (waves; ngr a power of two) take part; anc[] / acnt[] are -1 / 0 on entry.
  CTC_HD void phase_a1(const Beam &b, int n, int *anc, int *acnt, int grp, int ngr) {
    // Entry j = row * ngr + column; in row r this group takes column (grp - r) mod ngr.  (A plain "column = grp"
    // split is badly unbalanced: interior entries tend to come with a fixed number of leaf children each, so they
    // sit at a fixed residue of j and would all land on the same few groups.)
    const int rows = ceil_div_p2(n, ngr);
    auto entry_of = [=](int r) { return r * ngr + ((grp - r) & (ngr - 1)); };  // ngr is a power of two
    for (int k0 = 0; k0 < rows; k0 += x.lanes()) {
      // leaves (the next entry is not a descendant) are settled one per lane; only entries with in-beam
      // descendants need the wave-wide search and the painting
      const int k = k0 + x.lane();
      const int j = entry_of(k);
      bool internal = false;
      int dj = 0;
      if (k < rows && j < n) {
        dj = b.dep[j];
        internal = j + 1 < n && b.lcp[j + 1] >= dj;
        if (!internal) w.e[j] = j + 1;
      }
      unsigned long long todo = x.ballot(internal);
      if (internal) x.count(EV_INTERNAL, 1);
      // (X::kQuarters: the build for chain-shaped beams without a scorer -- three or more interior entries in a wave, as blank-dominated rows
      //  have them in most frames; by its mere presence the search costs random rows 3 % of the frame, so it is a build of its own)
      if ((LM || (X::kQuarters && __builtin_popcountll(todo) >= 3)) && x.subtrees_by_quarters(todo, k0, grp, ngr, b.dep, b.lcp, n, w.e, anc, acnt)) todo = 0;
      while (todo) {
        const int kk = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int jj = entry_of(k0 + kk);
        const int q = x.first_below(b.lcp, jj + 2, n, x.pick(dj, kk));  // the entry's depth sits in lane kk
        if (x.lane() == 0) w.e[jj] = q;
        for (int c = jj + 1 + x.lane(); c < q; c += x.lanes()) {
          x.atomic_max(&anc[c], jj);
          x.atomic_add(&acnt[c], 1);
        }
      }
    }
  }

  // One time step.  w.clp/w.cch (and rank_of in pruned mode) hold this step's candidates; `last` selects the
  // bookkeeping that DecoderState::decode() needs (the permutation std::nth_element leaves behind).
  // `stage`/`stage_val`: in identity mode the caller hands over its prefetched value of the NEXT frame's row; it is
  // parked in the other half of clpbuf before the closing fence, so the next frame starts without a load phase.
  // Returns ST_OK or the failure (identical in every thread; also left in VAR_STATUS).
  // `next_cnt` / `next_val`: threads below next_cnt hold a log-probability of the NEXT frame's candidates (the caller's
  // prefetch; both may still be in flight when the step starts): looked at after phase B for danger mode (see note_lp).
  CTC_HD int step(const StepIn &in, bool last, bool stage = false, float stage_val = 0.f, int next_cnt = 0, float next_val = 0.f) {
    select_beams();
    const Beam b = w.cur;
    const Beam nb = w.nxt;
    const int tid = x.tid_fresh(), nt = x.nt();
    const int n = st_n, pool_count = st_pool;
    const int K = d.K;
    const int Vc = in.Vc, brank = in.blank_rank;
    if (SMALLV) { CTC_ASSUME(n >= 1 && n <= kClsK); CTC_ASSUME(Vc >= 1 && Vc <= kClsVc); CTC_ASSUME(d.K <= kClsK); }
    const int Vnb = Vc - (brank >= 0 ? 1 : 0);
    const int S = n * (2 + Vnb);
    const bool small_vocab = SMALLV || Vnb <= 64;  // existing children fit a 64-bit mask per parent
    int *pv = pvars(in.t);
    int *surv = w.surv, *rk = w.surv + K, *ord = w.surv + 2 * K;
    Window wd{1u, 0};
    if (!kSpec) wd = first_window();
    HotPre hp{0xFFFFFFFFu, 0, nullptr, nullptr};
    if (kHotPre) {
      hp.cap = wide_list_cap();
      hp.key = reinterpret_cast<uint32_t *>(w.nxt.node);
      hp.slot = reinterpret_cast<int *>(hp.key + hp.cap + 4);
      if (x.tid() == 0) w.vars[VAR_HOTN] = 0;  // (phase B is two barriers away)
    }
    // LM tier: candidates whose (prefix score + label log-prob) falls below the worst prefix's score plus the blank's
    // log-prob (minus beta) are skipped once the beam is full (ctc_beam_search_decoder.cpp:74-82,93-95).  The prefixes
    // are visited best first there and the loop breaks at the first miss; scores only fall from there on and float
    // addition is monotone, so the break is the same as this per-candidate test.
    float min_cutoff = CTC_NEG_MAX;
    bool full_beam = false;
    if (LM) {
      const double bpos = lm_beta > 0.0 ? lm_beta : 0.0;  // std::max(0.0, beta)
      min_cutoff = (float)((double)(unord_f32(st_minkey) + in.blank_prob) - bpos);
      full_beam = n == K;
    }
    auto cut = [&](float lp, float prefix_score) { return LM && full_beam && lp + prefix_score < min_cutoff; };
    if (LM && CTC_RARE(x.uni(w.vars[VAR_DANGER]) != 0)) {
      // (behind the scorer hook a frame can be abandoned after it has written the next beam's order: that goes to the copy of the
      //  other parity -- fin_nxt() --, so the order this frame reads is the one the last COMMITTED frame left.  Rounds 4-5 refused
      //  the combination: ST_CB_DANGER)
      lm_sorted_order(b, n);
    }

    // ---- A1 (phase_a1): subtree ranges, nearest in-beam ancestors.  kA1Overlap: it already ran, beside the previous frame's emission.
    w.anc = w.ancbuf + (in.t & 1) * K;
    int *acnt = w.acntbuf + (in.t & 1) * K;
    // LM tier (word models): the entries the previous frame created are pending (lm_emit) -- their dictionary record is
    // requested here and everything that hangs on it is settled between the barriers of phase A2, on threads that have
    // no part in A2 when the workgroup has them; phase B is the first reader.
    // (Measured and dropped: requesting the record from the emitting thread one frame earlier and the first level of the
    //  n-gram look-up here -- the look-up's cost in phase A2 is its arithmetic and its later levels, not the first round
    //  trip; and spreading the pending entries over all idle waves -- every wave then pays the set-up.)
    const bool lm_job = LM && !lm_char_();
    const int lm_joff = (SMALLV && x.nt_is(1024)) ? kSmallK : nt >= 2 * ((n + 63) & ~63) ? ((n + 63) & ~63) : 0;
    int lm_jk = -1;
    bool lm_defer = false;  // (kLmOverlap: the entry's n-gram query may run beside phase B -- see phase A2)
    ctclm::DictNode lm_jinfo;
    if (lm_job) {
      const int k = tid - lm_joff;
      if (k >= 0 && k < n) {
        const int fc = b.dfc[k], dnk = b.dn[k];
        if (fc == kLmPending) {
          lm_jk = k; lm_jinfo = lm_dict[dnk];
          // ... unless the entry has descendants in the beam (a revived node: its children's scores in phase B's entry part read
          // the window of the word it spells -- path_trie.cpp:40-57 -- while the query would still be in flight)
          if (kLmOverlap) lm_defer = !(k + 1 < n && b.lcp[k + 1] >= b.dep[k]);
        }
      }
    }
    x.tick();
    if (!kA1Overlap) {
      phase_a1(b, n, w.anc, acnt, x.group(), x.ngroups());
      x.tick();
      x.sync();
    }
    if (kSpec && tid == x.spec_thread()) spec_predict(in.t);
    // ---- A2: Euler-tour slot offsets; which children of in-beam parents already exist
    int npin = 0;
    Int4v *prec = reinterpret_cast<Int4v *>(w.nxt.node);  // (kParentRec)
    for (int j = tid; j < n; j += nt) {
      const int dj = b.dep[j], q = w.e[j], P = w.anc[j], a = acnt[j];
      w.ostart[j] = 2 * j + Vnb * (j - a);
      const int cs_j = 2 * q + Vnb * (q - 1 - a);
      w.cstart[j] = cs_j;
      if (kParentRec) prec[j] = Int4v{(int)ctcmath::f32_to_bits(b.score[j]), (int)ctcmath::f32_to_bits(b.bprev[j]), cs_j, b.ch[j]};
      int pr = -1, rr = -1;
      if (P >= 0) {
        if (b.dep[P] == dj - 1) {                            // parent in the beam: "hit" (path_trie.cpp:40-48)
          pr = rank_of_char(in, b.ch[j]);
        } else {
          // dead-interior child X of the nearest in-beam ancestor on the way down to j (alive because j is below it)
          if (CTC_RARE(b.viaanc[j] != b.node[P])) {
            int hops = dj - b.dep[P] - 1, xn = b.node[j];
            x.count(EV_WALK, 1); x.count(EV_WALK_HOPS, hops);
#if !defined(CTC_NO_WALK_FROM_PARENT)
            // (scorer instantiations -- more than one walk per frame there, each hop a dependent HBM read on the two waves the others wait
            //  for: the first level up is the entry's own parent field, no read)
            int h0 = 0;
            if (LM && hops >= 1) { xn = b.par[j]; h0 = 1; }
            for (int h = h0; h < hops; ++h) xn = pool[xn].parent;
#else
            for (int h = 0; h < hops; ++h) xn = pool[xn].parent;
#endif
            b.via[j] = xn;
            b.viaanc[j] = b.node[P];
            b.viach[j] = pool[xn].ch();
          }
          // j is the first beam entry below X iff its predecessor is outside X's subtree: then j revives X
          if (b.lcp[j] <= b.dep[P]) rr = rank_of_char(in, b.viach[j]);
          x.count(EV_DEAD_PARENT, 1); if (rr >= 0) x.count(EV_REVIVE_CAND, 1);
        }
        const int r = pr >= 0 ? pr : rr;
        if (r >= 0 && small_vocab) {
          const int bit = r - ((brank >= 0 && r > brank) ? 1 : 0);
          x.atomic_or(&w.hit[hit_word(P, bit)], 1u << (bit & 31));
        }
      }
      w.pinr[j] = pr;
      w.revr[j] = rr;
      npin += pr >= 0;
    }
    // (the other waves had no entry: skip the reduction; with at most one entry per thread the count is a ballot's population)
    if (x.group() * x.lanes() < n) {
      if (SMALLV && x.nt_is(1024)) x.wave_add_flag(&pv[P_NPIN], npin != 0);
      else x.wave_add(&pv[P_NPIN], npin);
    }
    // Word models in the fixed-layout class (kLmOverlap): only HALF of a new entry's settling happens here -- its dictionary
    // record, i.e. the gate of its children (path_trie.cpp:59-70), which phase B needs for every candidate.  The n-gram query
    // of the word it spells (dependent global loads: two to three round trips, 2.6 k clocks on two waves while fourteen waited
    // at this barrier) is needed by ONE candidate of phase B -- the entry's space child -- and runs beside phase B on the same two
    // waves, which then score that child themselves (below).  The entry is marked in the high gate word (free: <= 32 labels),
    // so that the child waves leave the slot alone; the mark comes off behind phase B's barrier.
    const bool lm_ovl = kLmOverlap && lm_job;
    if (lm_job && tid >= lm_joff) {
      if (lm_jk >= 0) {
        if (lm_ovl && lm_defer) { b.dmlo[lm_jk] = (int)lm_jinfo.mask_lo; b.dmhi[lm_jk] = (int)kLmSpaceDeferred; b.dfc[lm_jk] = (int)lm_jinfo.first_child; }
        else lm_resolve_entry(b, lm_jk, lm_jinfo);
      }
      for (int k = tid - lm_joff + (nt - lm_joff); k < n; k += nt - lm_joff)  // (workgroups with fewer threads than entries)
        if (b.dfc[k] == kLmPending) lm_resolve_entry(b, k, lm->dict[b.dn[k]]);
    }
    x.sync();
    x.mark(0);
    const int npin_total = pv[P_NPIN];  // final since the barrier above; requested here so that phase C does not wait for it
    const uint32_t thr = kSpec ? (uint32_t)x.uni(w.vars[VAR_SPEC + SP_THR]) : 0xFFFFFFFFu;  // (written between the barriers of phase A: spec_predict)
    if (kHotPre) {
      // the pre-list's threshold: the speculative select's anchor -- the previous best score plus this row's best label log-probability,
      // which takes the row's own swing out of the distance -- minus 1.25 x the distance anchor -> K-th key the last frame showed
      hot_anchor = ord_f32(unord_f32(st_maxkey) + ctcmath::bits_to_f32((uint32_t)x.uni(w.vars[VAR_ROWMAX])));
      if (st_gap != 0 && hp.cap > 0 && small_vocab && st_wlog < 32) {
#ifndef CTC_HOT_MARGIN_SHIFT
#define CTC_HOT_MARGIN_SHIFT 1
#endif
        const uint64_t reach = (uint64_t)st_gap + (st_gap >> CTC_HOT_MARGIN_SHIFT) + 2;
        const uint32_t t = (uint64_t)hot_anchor > reach ? (uint32_t)(hot_anchor - reach) : 1u;
        hp.thr = t < wd.lo ? wd.lo : t;
      }
    }

    // ---- B: score every candidate, lay it out in DFS (Euler-tour) slot order and count it into the select histogram.
    // B1 (beam entries themselves + revived children) and B2 (brand-new children) are independent: with enough
    // waves they run side by side on disjoint threads.
    // (fixed-layout class: two entry waves, a compile-time split.  Wider beams: two entry waves as well, taking the entries in
    //  several rounds -- the children's part is the longer one, an entry round costs 0.7 us, a round of children 0.25 us on
    //  fourteen waves: at beam 500 eight entry waves left the children 32 rounds on the other eight, 44.4 -> 42.8 ms per batch;
    //  four entry waves 43.3, one 44.8)
    const int n1 = (SMALLV && x.nt_is(1024)) ? kSmallK : ((n + 63) & ~63) > 128 ? 128 : (n + 63) & ~63;
    const bool split = nt - n1 >= 128;
    if (!split || tid < n1) {
      // (Measured and dropped: a raised wave priority for the entry part -- neutral.)
      const float lp_blank = brank >= 0 ? w.clp[brank] : CTC_NEG_MAX;
      int ncand = 0;
      for (int j = tid; j < n; j += (split ? n1 : nt)) {
        const int c = b.ch[j];
        const int danger = w.vars[VAR_DANGER];  // (set by an earlier frame at the latest: the flag for THIS frame's row was raised one frame ahead)
        const int r = rank_of_char(in, c);
        const float sc = b.score[j], nbp = b.nbprev[j];
        float bcur = (brank >= 0 && !cut(lp_blank, sc)) ? lp_blank + sc : CTC_NEG_MAX;             // :97-101
        float nbcur = CTC_NEG_MAX;
        const bool has_rep = r >= 0 && !cut(w.clp[r], sc);
        if (has_rep) nbcur = w.clp[r] + nbp;  // :103-106 -- log_sum_exp(-FLT_MAX, y) returns y (decoder_utils.h:50)
        const int P = w.anc[j];
        const int pr = w.pinr[j];
#if !defined(CTC_NO_LM_REVIVE_PREFETCH)
        // (scorer instantiations: a dictionary leaves three revival candidates per frame -- tools/beam_stats.py --, so one lane of the entry
        //  waves takes the revive path below in nearly every frame, and its read of the dead node's label probability from the pool
        //  (HBM: ~1 k clocks) sat behind the two log_sum_exp chains.  Requested here, it arrives while they run.)
        const int rx_pre = LM ? w.revr[j] : -1;
        float xl_pre = 0.f;
        if (LM && rx_pre >= 0) xl_pre = pool[b.via[j]].lpc;
#endif
        // Pool updates (global stores) are issued after everything else of the iteration: the memory waits the compiler
        // places in the arithmetic below would otherwise also wait for their acknowledgement.  (Scorer instantiations only:
        // dozens of updates per frame there, one in ten frames on random rows.)
        int upd_node = -1, upd_xn = -1, upd_xc = 0;
        float upd_lp = 0.f, upd_xlp = 0.f;
        if (pr >= 0) {
          const float lp = w.clp[pr];
          if (!cut(lp, b.score[P])) {
            x.count(EV_PINNED, 1);
            if (b.lpc[j] < lp) {                                             // path_trie.cpp:42-45
              x.count(EV_LPC_UPDATE, 1);
              b.lpc[j] = lp;
              if (LM) { upd_node = b.node[j]; upd_lp = lp; }  // (the pool is updated at the end of the iteration: see below)
              else set_node_time(b.node[j], c, in.t, lp);
            }
            float logp = child_logp(P, c, lp);
            if (LM && lm_scores(c)) logp = lm_apply(logp, lm_window(b, P, c));  // :120-137
            // :138-139.  (danger mode: the parent sits before the entry in `prefixes` -> its extension was added first)
            // (one call: the arguments trade places -- two inlined copies of the exact log_sum_exp sat here)
            const bool parent_first = CTC_RARE(danger != 0) && has_rep && w.apos[P] < w.apos[j];
            nbcur = lse(parent_first ? logp : nbcur, parent_first ? nbcur : logp);
          }
        }
        w.b_new[j] = bcur;
        w.nb_new[j] = nbcur;
        const float ns = lse(bcur, nbcur);                                  // path_trie.cpp:131-136
        w.sc_new[j] = ns;
        const int s0 = w.ostart[j];
        uint32_t k0 = 0, i0 = kHoleInfo;
        const int rx = w.revr[j];
        if (CTC_RARE(rx >= 0 && !cut(w.clp[rx], b.score[P]))) {                       // path_trie.cpp:40-57: hit + revive
          const int cx = b.viach[j];
          const float lp = w.clp[rx];
          const int xn = b.via[j];
#if !defined(CTC_NO_LM_REVIVE_PREFETCH)
          float xl = LM ? xl_pre : pool[xn].lpc;
#else
          float xl = pool[xn].lpc;
#endif
          if (xl < lp) {
            xl = lp;
            if (LM) { upd_xn = xn; upd_xc = cx; upd_xlp = lp; }
            else set_node_time(xn, cx, in.t, lp);
          }
          w.rev_lpc[j] = xl;  // read back by whoever compacts the revived node (same step, other thread)
          float logp = child_logp(P, cx, lp);
          if (LM && lm_scores(cx)) logp = lm_apply(logp, lm_window(b, P, cx));
          k0 = ord_f32(logp);
          i0 = mk_info(cx, T_REVIVED, j);
          ++ncand;
        }
        ++ncand;
        const uint32_t k1 = ord_f32(ns);
        w.skey[s0] = k0;
        w.skey[s0 + 1] = k1;
        if (!LAZY) { w.sinfo[s0] = i0; w.sinfo[s0 + 1] = mk_info(c, T_SELF, j); }
        if (kSpec) {
          x.hot_append(k0 >= thr, k0, s0, w.list, w.lslot, &w.vars[VAR_G]);  // (a revived node: rare)
          x.hot_append_wave(k1 >= thr, k1, s0 + 1, w.list, w.lslot, &w.vars[VAR_G]);
        } else if (small_vocab) { hist_add(wd, k0, hp, s0); hist_add(wd, k1, hp, s0 + 1); }
        if (LM && upd_node >= 0) set_node_time(upd_node, c, in.t, upd_lp);
        if (LM && CTC_RARE(upd_xn >= 0)) set_node_time(upd_xn, upd_xc, in.t, upd_xlp);
      }
      if (LM) x.wave_add(&pv[P_NCAND], ncand);
    }
    x.mark(1);
    if (lm_ovl && tid >= n1 && tid < 2 * n1) {  // the two waves that settle the new entries: second half, and the space children
      int ncand = 0;
      if (lm_jk >= 0 && lm_defer) {
        const int k = lm_jk;
        double cond = ctclm::kOovScore;
        uint32_t st2 = 0;
        int cl2 = 0;
        if (lm_jinfo.word != ctclm::kNoWord) {
          st2 = (uint32_t)b.lmst[k];
          cl2 = b.lmcl[k];
          cond = lm_cond_(&st2, &cl2, lm_jinfo.word);
        }
        put_f64(cond, &b.spc_lo[k], &b.spc_hi[k]);
        b.spst[k] = (int)st2; b.spcl[k] = cl2;
        // the candidate (entry k, space): what a lane of the children's part computes for it (:108-139), once the window is known
        const int r = rank_of_char(in, lm_space);
        if (r >= 0) {
          const int rn = r - ((brank >= 0 && r > brank) ? 1 : 0);
          const float lp = w.clp[r], psc = b.score[k];
          const uint32_t hw = w.hit[hit_word(k, rn)];
          uint32_t live = 0u - (((hw >> (rn & 31)) & 1u) ^ 1u);
          live &= 0u - (((lm_jinfo.mask_lo >> lm_space) & 1u) & (uint32_t)!cut(lp, psc));
          float logp = lp + psc;  // (a new entry was made by a label that is not the space: never the repeat rule)
          if (live) { logp = lm_apply(logp, cond); ++ncand; }
          const uint32_t key = ord_f32(logp) & live;
          const int sl = w.cstart[k] + rn;
          w.skey[sl] = key;
          if (!LAZY) w.sinfo[sl] = live ? mk_info(lm_space, T_CHILD, k) : kHoleInfo;
          if (kSpec) x.hot_append(key >= thr, key, sl, w.list, w.lslot, &w.vars[VAR_G]);
          else hist_add(wd, key);
        }
      }
      x.wave_add(&pv[P_NCAND], ncand);
    }
    if (kLcpTable && tid >= nt - 64) {
      build_lcp_table(b, n, tid - (nt - 64));
    } else if ((!split || tid >= n1) && !(lm_ovl && tid < 2 * n1)) {
      const int t2 = lm_ovl ? tid - 2 * n1 : split ? tid - n1 : tid, nt2 = lm_ovl ? nt - 2 * n1 : (split ? nt - n1 : nt) - (kLcpTable ? 64 : 0);
      const int sh = ceil_log2_u32((uint32_t)(Vnb > 1 ? Vnb : 1));
      // lanes per parent: a power of two >= Vnb -- or, in the class of the pruned default (SMALLV == 2: up to 40 candidates), 40:
      // 22 parents per pass on fourteen waves instead of 14 (beam 100: five passes instead of eight)
      constexpr bool fixed_lp = SMALLV == 2;
      const int lp2 = fixed_lp ? kMidVc : 1 << sh;
      int ncand = 0;
      if (kParentRec && small_vocab && nt2 >= lp2) {
        // Wide beams without a scorer: the same groups of lp2 lanes per parent, reading the packed records of phase A2 -- and a loop that
        // every lane of a wave leaves together (the wave's first lane has its smallest parent: a scalar counter closes the loop, the lanes
        // past the last parent are masked inside).  The general form below ends each lane's loop on its own: ten of its instructions
        // per candidate were the bookkeeping of a divergent loop.
        const int g0 = t2 >> sh, rn = t2 & (lp2 - 1), ng = nt2 >> sh;
        const bool lane = rn < Vnb && g0 < ng;
        const int r = lane ? rn + ((brank >= 0 && rn >= brank) ? 1 : 0) : 0;
        const int c = IDENT ? r : w.cch[r];
        const float lp = w.clp[r];
        const uint32_t childinfo = mk_info(c, T_CHILD, 0);
        const int g_first = x.uni(g0);
        const int g_last = g_first + ((x.lanes() >> sh) > 1 ? (x.lanes() >> sh) - 1 : 0);  // the wave's largest first parent
        auto one = [&](int i) {
          const Int4v rc = x.load4(reinterpret_cast<const int *>(prec + i));
          const uint32_t hw = w.hit[hit_word(i, rn)];
          const float psc = ctcmath::bits_to_f32((uint32_t)rc.x), pbp = ctcmath::bits_to_f32((uint32_t)rc.y);
          const uint32_t live = 0u - (((hw >> (rn & 31)) & 1u) ^ 1u);  // all ones unless the child already exists
          const float ext = lp + psc, rep = pbp > CTC_NEG_MAX ? lp + pbp : CTC_NEG_MAX;  // :110-118
          const float logp = c == rc.w ? rep : ext;
          const uint32_t k = ord_f32_raw(logp) & live;
          const int sl = rc.z + rn;
          w.skey[sl] = k;
          if (!LAZY) w.sinfo[sl] = x.bitsel(live, childinfo + (uint32_t)i, kHoleInfo);
          hist_add(wd, k, hp, sl);
        };
        if (lane) {
          int i = g0;
          // (measured: two parents per trip, both records requested before either is used -- no gain, 38.1 against 38.0 ms)
          for (int i0 = g_last; i0 < n; i0 += ng, i += ng) one(i);  // (every parent of the wave's lanes exists: no test per lane)
          if (i < n) one(i);                                        // (the last pass: the wave's first groups only)
        }
      } else if (small_vocab && nt2 >= lp2) {        // a group of lp2 lanes per parent, one lane per character
        const int g0 = fixed_lp ? t2 / kMidVc : t2 >> sh;   // this lane's first parent
        const int rn = fixed_lp ? t2 - g0 * kMidVc : t2 & (lp2 - 1);
        const int ng = fixed_lp ? nt2 / kMidVc : nt2 >> sh;
        if (rn < Vnb && g0 < ng) {
          const int r = rn + ((brank >= 0 && rn >= brank) ? 1 : 0);
          const int c = IDENT ? r : w.cch[r];
          const float lp = w.clp[r];
          const uint32_t childinfo = mk_info(c, T_CHILD, 0);
          // (word models: the lane's label picks ONE of the two gate words of a parent, once -- and the gate is folded into
          //  `live` with plain bit operations, so that the compiler requests the word together with the parent's other
          //  fields instead of behind the cutoff test: one LDS round trip per parent instead of two)
          const int *gate_w = LM && WORDLM ? (c < 32 ? b.dmlo : b.dmhi) : nullptr;
          const int gate_sh = c & 31;
          // (the info word of (label, parent i) is childinfo + i: it is the loop's induction variable)
          uint32_t ci = childinfo + (uint32_t)g0;
          const uint32_t ci_end = childinfo + (uint32_t)n;
          // (speculative select: the list space of a pass is reserved with a returning LDS atomic -- a full round trip.  The
          //  loop is pipelined by hand: the atomic is issued, then the NEXT pass's parent fields are requested, and only then
          //  is the atomic's result used -- LDS answers in order, so the append never waits on its own)
          // (Measured and dropped, round 6: this loop under a scalar trip counter with the last, partial pass peeled off -- what gave the
          //  wide-beam layouts 1.2 % (kParentRec) -- leaves the north-star kernel where it is: 43 -> 41 instructions, 5.129 / 5.137 ms.)
          // (Measured and dropped, round 4: two parents per trip -- both parents' fields requested together, the second's
          //  arithmetic in the first's waits: +5 % on the north-star kernel, +2 % at beam 500.  A trip is not waiting for its
          //  parent's fields.)
          struct Par { int cs; uint32_t hw; int pch; float psc, pbp; uint32_t gw, gh; };
          auto fetch = [&](int i) {
            Par p;
            // everything this candidate needs from its parent, requested in one go (one LDS round trip), no branches
            if (kParentRec) {
              const Int4v r = x.load4(reinterpret_cast<const int *>(prec + i));
              p.psc = ctcmath::bits_to_f32((uint32_t)r.x); p.pbp = ctcmath::bits_to_f32((uint32_t)r.y); p.cs = r.z; p.pch = r.w;
              p.hw = w.hit[hit_word(i, rn)];
            } else {
              p.cs = w.cstart[i]; p.hw = w.hit[hit_word(i, rn)]; p.pch = b.ch[i]; p.psc = b.score[i]; p.pbp = b.bprev[i];
            }
            p.gw = (LM && WORDLM) ? (uint32_t)gate_w[i] : 0u;
#if !defined(CTC_NO_LM_FETCH_DEFER)
            // (kLmOverlap: the "space child is being settled" mark travels with the parent's other fields -- asked for behind the cutoff
            //  test it cost the space lanes' waves an LDS round trip of their own in every trip)
            p.gh = kLmOverlap ? (uint32_t)b.dmhi[i] : 0u;
#else
            p.gh = 0u;
#endif
            return p;
          };
          int i = g0;
          bool act = ci < ci_end;
          Par cur{};
          if (act) cur = fetch(i);
          while (act) {
            const int cs = cur.cs;
            const uint32_t hw = cur.hw;
            const int pch = cur.pch;
            const float psc = cur.psc, pbp = cur.pbp;
            uint32_t live = 0u - (((hw >> (rn & 31)) & 1u) ^ 1u);  // all ones unless the child already exists
            bool deferred = false;
            const float ext = lp + psc, rep = pbp > CTC_NEG_MAX ? lp + pbp : CTC_NEG_MAX;  // :110-118
            float logp = c == pch ? rep : ext;
            if (LM) {
              if (WORDLM) {
                const uint32_t gate = (cur.gw >> gate_sh) & 1u;
                live &= 0u - (gate & (uint32_t)!cut(lp, psc));                    // :93-95, path_trie.cpp:59-70
              } else if (cut(lp, psc) || !lm_allows(b, i, c)) {
                live = 0u;
              }
              if (kLmOverlap) {
#if !defined(CTC_NO_LM_FETCH_DEFER)
                deferred = lm_ovl && c == lm_space && cur.gh == kLmSpaceDeferred;
#else
                deferred = lm_ovl && c == lm_space && (uint32_t)b.dmhi[i] == kLmSpaceDeferred;
#endif
                if (deferred) live = 0u;
              }
              if (live && lm_scores(c)) logp = lm_apply(logp, lm_window(b, i, c));  // :120-137
              ncand += live ? 1 : 0;
            }
            const uint32_t k = (LM ? ord_f32(logp) : ord_f32_raw(logp)) & live;
            const int s = cs + rn;
            if (!(kLmOverlap && deferred)) {  // (the space child of an entry whose word is still being looked up: scored by the settling wave)
              w.skey[s] = k;
              if (!LAZY) w.sinfo[s] = x.bitsel(live, ci, kHoleInfo);
            }
            const bool hotk = kSpec && k >= thr;
            const auto tk = x.hot_issue(hotk, &w.vars[VAR_G]);
            if (!kSpec && !(kLmOverlap && deferred)) hist_add(wd, k, hp, s);
            i += ng; ci += (uint32_t)ng;
            act = ci < ci_end;
            if (act) cur = fetch(i);
            if (kSpec) x.hot_commit(tk, hotk, k, s, w.list, w.lslot);
          }
        }
      } else {
        for (int idx = t2; idx < n * Vnb; idx += nt2) {
          const int i = idx / Vnb, rn = idx - i * Vnb;
          const int r = rn + ((brank >= 0 && rn >= brank) ? 1 : 0);
          const int c = IDENT ? r : w.cch[r];
          const int s = w.cstart[i] + rn;
          bool exists = small_vocab && ((w.hit[hit_word(i, rn)] >> (rn & 31)) & 1u);
          float logp = CTC_NEG_MAX;
          if (LM) {
            // (with more than 64 candidate labels the children that already exist are punched out afterwards; they are
            // counted here and subtracted there)
            if (cut(w.clp[r], b.score[i]) || !lm_allows(b, i, c)) exists = true;
            if (!exists) {
              logp = child_logp(i, c, w.clp[r]);
              if (lm_scores(c)) logp = lm_apply(logp, lm_window(b, i, c));
              ++ncand;
            }
          } else if (!exists) {
            logp = child_logp(i, c, w.clp[r]);
          }
          const uint32_t k = exists ? 0u : ord_f32(logp);
          w.skey[s] = k;
          if (!LAZY) w.sinfo[s] = exists ? kHoleInfo : mk_info(c, T_CHILD, i);
          if (kSpec) x.hot_append(k >= thr, k, s, w.list, w.lslot, &w.vars[VAR_G]);
          else if (small_vocab) hist_add(wd, k, hp, s);
        }
      }
      if (LM) x.wave_add(&pv[P_NCAND], ncand);
    }
    // (speculative select: the look at the next frame's row -- see below -- comes before the phase's barrier, because the
    //  select reads the flag right behind it)
    if (kSpec && tid < next_cnt) note_lp(next_val);
    x.sync();
    if (lm_ovl && tid >= n1 && tid < 2 * n1 && lm_jk >= 0 && lm_defer) b.dmhi[lm_jk] = (int)lm_jinfo.mask_hi;  // (the mark comes off: the next reader is the emission, two barriers on)
    if (!small_vocab) {  // children that already exist leave a hole in their parent's group; then the histogram
      for (int j = tid; j < n; j += nt) {
        const int r = w.pinr[j] >= 0 ? w.pinr[j] : w.revr[j];
        if (r >= 0) {
          const int s = w.cstart[w.anc[j]] + r - ((brank >= 0 && r > brank) ? 1 : 0);
          if (LM && w.skey[s] != 0u) x.atomic_add(&pv[P_NCAND], -1);
          w.skey[s] = 0;
          if (!LAZY) w.sinfo[s] = kHoleInfo;
        }
      }
      x.sync();
      for (int s = tid; s < S; s += nt) hist_add(wd, w.skey[s]);
      x.sync();
    }
    // the next frame's row (in the caller's prefetch registers since before this frame started: long arrived): a bad value
    // puts the utterance into danger mode from THIS frame's selection on (the barriers of phase C order the store)
    if (!kSpec && tid < next_cnt) note_lp(next_val);
    x.mark(2);
    x.probe_keys(in.t, S, w.skey, K, kSpec ? (uint32_t)w.vars[VAR_SPEC + SP_ANCHOR] : st_maxkey, w.clp, Vc);  // (host build: statistics of a frame's keys; nothing on the GPU)

    // ---- C: the K-th best key.  #candidates = beam entries + new children - children that already exist as entries
    const int N = LM ? x.uni(pv[P_NCAND]) : n * (1 + Vnb) - x.uni(npin_total);
    uint32_t tau = 0, tauc = 0;
    bool exact = false, have_bitmap = false;
    SlotCtx lz;  // LAZY: no info words are stored; whoever needs one derives it from the layout
    if (LAZY) lz = slot_ctx(b, n, Vnb, brank);
    bool keys_in_ord = false;  // the exact replay overwrote w.skey[]: the survivors' keys are in ord[] (by beam position)
    bool spec_done = false;    // the speculative select settled the frame: surv[] holds the K survivors in slot order
    int hot = 0;
    if (kSpec && CTC_USUAL(N > K)) {
      // [1]: the hot list's length, [2]: zero (reset with it), [3]: the danger flag -- requested FIRST (LDS answers in order),
      // the ranking's first reads behind it: they are in flight while the length is looked at
      const Int4v tvv = x.load4(&w.vars[VAR_TAU]);
      const auto pre = x.spec_pre(w.list, w.lslot);
      int tv0[4];
      x.uni4v(tvv, tv0);
      hot = tv0[1];
      // (Measured and dropped: a second attempt with another threshold when the list comes up short or overflows -- a pass
      //  over the slot keys that extends or rebuilds the list.  It settles 11 of the 17 % of frames that fall back on random
      //  rows, but the pass and the longer list it leaves -- more than 128 keys: four times the compares -- cost 3 k clocks
      //  against the 4.4 k of falling back, and the attempts that fail pay both: +2 % kernel time.)
      if (CTC_USUAL(!last && hot >= K && hot <= kHotCap && x.spec_fits(hot) && tv0[3] == 0)) {
        // (w.hotge[]: the ranking's scratch, zero between frames; the VAR_TAU group takes the report of the K-th key's lane)
        const auto r = x.spec_select(pre, hot, K, w.list, w.lslot, w.bitmap, w.hotge, S, surv, &w.vars[VAR_TAU]);
        spec_done = r.ok != 0;
        if (CTC_USUAL(spec_done)) tau = r.tau;
      }
      if (CTC_RARE(!spec_done)) rehistogram(S, wd);
      if (spec_done) x.count(EV_SPEC_HOT, hot);
      x.count(spec_done ? EV_SPEC_OK : hot < K ? EV_SPEC_UNDER : hot > kHotCap ? EV_SPEC_OVER : EV_SPEC_OTHER, 1);
    }
    if (CTC_USUAL(N > K) && !spec_done) {  // ctc_beam_search_decoder.cpp:150
      have_bitmap = select_kth(S, K, pv, wd, hp);
      int tv[4];
      x.uni4(&w.vars[VAR_TAU], tv);
      tau = (uint32_t)tv[0];
      const int E = tv[2], m = K - tv[1];
      if (CTC_RARE(E > m)) {
        if (tid == 0) { x.count(EV_TIE_FRAMES, 1); x.count(EV_TIE_KEYS, E); if (E > kListCap) x.count(EV_TIE_BIG, 1); }
        have_bitmap = false;
        if (resolve_by_character(S, tau, m, E, pv, &lz)) tauc = (uint32_t)x.uni(w.vars[VAR_TAUC]);
        else exact = true;  // the boundary splits a group of equivalent prefixes
      } else if (!SMALLV && !have_bitmap) {
        // Wide beams: the select left its fast path (at beam 500 on random rows one frame in four: a key value near the K-th is
        // shared by more candidates than the bucket list holds -- equal scores are inherited by the children of equal prefixes)
        // and knows tau without having marked the survivors.  One pass that only writes the survivor bitmap, then the same
        // parallel expansion as the fast path, instead of an ordered compaction of all S slots.
        x.mark_ge(S, w.skey, tau, w.bitmap);
        x.sync();
        have_bitmap = true;
      }
#ifdef CTC_EXP_ALWAYS_EXACT  // (measurement builds: the cost of one exact replay = the change of the frame time)
      exact = true;
#endif
      if (CTC_RARE(last)) exact = true;  // decode() sorts the array exactly as nth_element left it (:164-190)
      if (CTC_RARE(tv[3] != 0)) exact = true;  // danger mode (VAR_DANGER, read with the select's result): the next frame adds its contributions in that order
    }
    x.mark(5);

    // ---- D: who survives, in DFS (= slot) order.  Normally an ordered compaction of the slots that pass the
    // threshold; when the outcome depends on it, an exact replay of std::nth_element followed by a ranking by slot.
    const int n_new = N < K ? N : K;
    if (tid == 0) { x.count(EV_FRAMES, 1); x.count(EV_CANDIDATES, N); if (exact) x.count(EV_EXACT, 1); }
    if (CTC_USUAL(kSpec && spec_done)) {
      x.mark(3);
    } else if (CTC_RARE(exact)) {
      keys_in_ord = nth_element_order(S, N, K, &lz);
      if (SMALLV) {
        for (int q = tid; q < K; q += nt) {  // rank by slot
          const int mine = ord[q];
          int r = 0;
          for (int o = 0; o < K; ++o) r += ord[o] < mine;
          rk[q] = r;
          surv[r] = mine;
        }
        x.sync();
      } else {
        // Wide beams: K compares per survivor are K * K / nt LDS reads per thread (beam 500: 10 us of a 42 us replay frame).  The
        // survivors are marked in the slot bitmap and expanded in slot order -- the select's own expansion --, and every entry of
        // the std::nth_element order finds its rank by bisection.
        const int nw = (S + 63) / 64;
        for (int i = tid; i < 2 * nw; i += nt) w.bitmap[i] = 0u;
        x.sync();
        for (int q = tid; q < K; q += nt) { const int s = ord[q]; x.atomic_or(&w.bitmap[s >> 5], 1u << (s & 31)); }
        x.sync();
        x.template expand_bitmap<false>(w.bitmap, nw, surv);
        for (int q = tid; q < K; q += nt) {
          const int mine = ord[q];
          int lo = 0, hi = K - 1;
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (surv[mid] < mine) lo = mid + 1; else hi = mid;
          }
          rk[q] = lo;
        }
        x.sync();
      }
      if (keys_in_ord) {
        for (int q = tid; q < K; q += nt) ord[rk[q]] = (int)(uint32_t)(EO::key(ekp()[q]) >> 16);
        zero_key_tail(S);  // (the replay staged its ranges in the block of the slot keys; the frames to come rewrite [0, S))
        x.sync();
      }
      x.mark(6);
    } else if (have_bitmap) {
      x.template expand_bitmap<LM>(w.bitmap, (S + 63) / 64, surv);
      x.mark(3);
    } else {
      const uint32_t *skey = w.skey, *sinfo = w.sinfo;
      const bool all = N <= K;
      x.compact_slots(S, surv, [=](int s) -> bool {
        const uint32_t k = skey[s];
        return all ? (k != 0u) : (k > tau || (k == tau && (tauc == 0u || ((LAZY ? info_of_slot(lz, s) : sinfo[s]) >> 16) >= tauc)));
      });
      x.mark(3);
    }
    x.mark(4);
    if (CTC_RARE(pool_count + n_new > pool_cap)) {  // cannot happen when the pool is sized 1 + K*T
      if (tid == 0) w.vars[VAR_STATUS] = ST_POOL_OVERFLOW;
      x.sync_full();
      return ST_POOL_OVERFLOW;
    }
    // Three independent parts per survivor -- its LCP with the previous survivor, its structural fields (+ the pool
    // append), its probabilities -- go to three different sets of waves when the workgroup has them.
    uint32_t kloc = 0, kmin = 0xFFFFFFFFu;
    bool r_prob_any = true;
    constexpr bool lazy_info = LAZY;
    uint32_t *sinf = reinterpret_cast<uint32_t *>(w.list);  // LAZY: the survivors' info words
    if (lazy_info) {
      sinf = w.sinfo;  // (K words: carve)
      for (int k = tid; k < n_new; k += nt) sinf[k] = info_of_slot(lz, surv[k]);
      x.sync();
    }
    if (kA1Overlap) {
      // Threads [0, 128) LCP, [128, 256) structure, [256, 384) probabilities -- two waves each, whatever the number of survivors
      // (at most 128) --, waves 6..13 phase A1 of the NEXT frame, waves 14 and 15 the per-frame resets and the select's prediction.
      const int role = (tid >= kSmallK) + (tid >= 2 * kSmallK) + (tid >= 3 * kSmallK);
      const int k = tid - role * kSmallK;  // (role 3: the thread's number among the 640 without a part in the emission)
      const bool act = role < 3 && k < n_new;
      int *oa = w.ancbuf + ((in.t + 1) & 1) * K, *oc = w.acntbuf + ((in.t + 1) & 1) * K;  // the next frame's paint buffers
      int s = 0;
      uint32_t inf = kHoleInfo, pinf = kHoleInfo;
      if (act) {  // (the previous survivor's words -- the LCP role's -- ride in the same two round trips)
        s = surv[k];
        const int sp = surv[k > 0 ? k - 1 : 0];
        inf = w.sinfo[s]; pinf = w.sinfo[sp];
      }
      const uint32_t type = info_type(inf);
      const int j = info_entry(inf);
      const bool self = type == T_SELF, child = type == T_CHILD;
      const int c = self ? -1 : info_ch(inf);
      r_prob_any = role == 2;
      // ---- first half: everything phase A1 reads (nb.lcp, nb.dep) is written; the other roles request what they need
      int p_id = 0, p_node = 0, p_upv = 0, p_dep = 0;                     // structure: the pool append, issued behind the barrier
      float q_b = 0.f, q_nb = 0.f, q_sc = 0.f, q_lpc = 0.f, q_score = 0.f, q_bprev = 0.f;  // probabilities: requested here, used behind it
      int q_ch = 0;
      uint32_t q_key = 0;
      if (act && role == 0) {
        // LCP with the previous survivor: the LCA depth of two candidates is the LCA depth of the entries they hang
        // off (a brand-new child never lies on an existing path), capped by the depth of a revived interior node.
        int l = -1;
        if (k > 0) {
          const int pj = info_entry(pinf);
          l = kLcpTable ? lca_depth_tbl(pj, j) : lca_depth(pj, j);
          if (CTC_RARE(type == T_REVIVED)) { const int dx = b.dep[w.anc[j]] + 1; l = dx < l ? dx : l; }
          if (CTC_RARE(info_type(pinf) == T_REVIVED)) { const int dx = b.dep[w.anc[pj]] + 1; l = dx < l ? dx : l; }
        }
        nb.lcp[k] = l;
      }
      if (act && role == 1) {
        // An entry that stays (T_SELF) and a brand-new child (T_CHILD, of entry j) both draw everything from entry j:
        // one batch of loads, then selects.  A revived interior node (rare) hangs off the nearest in-beam ancestor of j.
        const int node_j = b.node[j], par_j = b.par[j], ch_j = b.ch[j], dep_j = b.dep[j];
        const int via_j = b.via[j], viaanc_j = b.viaanc[j], viach_j = b.viach[j], up_j = b.up[j];
        const int id = pool_count + k;                                              // ids by beam position (gaps are harmless)
        const int upv = (dep_j & (kExpress - 1)) == 0 ? node_j : up_j;
        int o_node = self ? node_j : id, o_par = self ? par_j : node_j, o_ch = self ? ch_j : c;
        int o_dep = self ? dep_j : dep_j + 1, o_viaanc = self ? viaanc_j : -1, o_up = self ? up_j : upv;
        if (CTC_RARE(!self && !child)) {                                            // path_trie.cpp:50-56 : revived
          const int P = w.anc[j];
          x.count(EV_REVIVED, 1);
          o_node = via_j; o_par = b.node[P]; o_dep = b.dep[P] + 1;
          o_up = (b.dep[P] & (kExpress - 1)) == 0 ? b.node[P] : b.up[P];
        }
        nb.dep[k] = o_dep;
        nb.node[k] = o_node; nb.par[k] = o_par; nb.ch[k] = o_ch;
        nb.via[k] = via_j; nb.viaanc[k] = o_viaanc; nb.viach[k] = viach_j; nb.up[k] = o_up;  // via/viach: only read when viaanc matches
        p_id = id; p_node = node_j; p_upv = upv; p_dep = dep_j;
      }
      if (act && role == 2) {
        q_b = w.b_new[j]; q_nb = w.nb_new[j]; q_sc = w.sc_new[j]; q_lpc = b.lpc[j];
        q_ch = b.ch[j]; q_score = b.score[j]; q_bprev = b.bprev[j];
        q_key = keys_in_ord ? (uint32_t)ord[k] : w.skey[s];
      }
      if (role == 3 && k < K) { oa[k] = -1; oc[k] = 0; }
      x.sync();
      x.tick();
      // ---- second half
      if (role == 3) {
        if (tid < 14 * 64) {
          phase_a1(nb, n_new, oa, oc, x.group() - 6, 8);
        } else {
          // Per-frame resets for the next step: the hot list's keys (its unused tail must read as zero), the survivor bitmap,
          // the existing-children masks (last read in phase B), the counters of the other parity.
          const int t0 = tid - 14 * 64;
          if (t0 < 64) {  // (wave 14: the arrays; wave 15: the counters and the select's prediction -- a long chain on one lane)
            for (int i = t0; i < kHotCap + 64; i += 64) w.list[i] = 0u;
            for (int i = t0; i < 2 * ((kClsSlots + 63) / 64); i += 64) w.bitmap[i] = 0u;
            for (int i = t0; i < (SMALLV == 1 ? n : 2 * n); i += 64) w.hit[i] = 0;
          } else if (tid == x.spec_thread()) {
            reset_pvars(pvars(in.t + 1));
            w.vars[VAR_G] = 0; w.vars[VAR_E] = 0;
            spec_learn(N > K, tau, hot, K);
          }
        }
      } else if (act && role == 1) {
        if (child) {                                                                // path_trie.cpp:97-105
          PoolNode pn; pn.parent = p_node; pn.cht = PoolNode::pack(c, in.t); pn.lpc = w.clp[rank_of_char(in, c)];
          pool[p_id] = pn;
          if (CTC_RARE(long_t)) pool_thi[p_id] = in.t >> 16;
          if (((p_dep + 1) & (kExpress - 1)) == 0) pool_up[p_id] = p_upv;
        }
      } else if (act && role == 2) {
        float o_b = q_b, o_nb = q_nb, o_sc = q_sc, o_lpc = q_lpc;
        if (!self) {              // a new or revived prefix starts from its first path only (path_trie.cpp:52-56, 99-104)
          const float lp = w.clp[rank_of_char(in, c)];
          float logp;
          if (CTC_USUAL(child)) {                                                   // ctc_beam_search_decoder.cpp:110-118
            const float rep = q_bprev > CTC_NEG_MAX ? lp + q_bprev : CTC_NEG_MAX;
            logp = c == q_ch ? rep : lp + q_score;
            o_lpc = lp;
          } else {
            logp = child_logp(w.anc[j], c, lp);
            o_lpc = w.rev_lpc[j];
          }
          o_b = CTC_NEG_MAX; o_nb = logp; o_sc = logp;
        }
        nb.bprev[k] = o_b; nb.nbprev[k] = o_nb; nb.score[k] = o_sc; nb.lpc[k] = o_lpc;
        kloc = q_key;
      }
    } else
    {
      // (fixed-layout class at its usual 1024 threads: at most 128 survivors -- two waves per role, whatever their number:
      //  the role of a thread, the number of roles and the threads left for the resets are then compile-time facts instead
      //  of two dozen scalar instructions every wave executes before its first load)
      const int ne = (SMALLV && x.nt_is(1024)) ? kSmallK : (n_new + 63) & ~63;
      // (wide beams: the workgroup has two sets of waves per survivor, not three: LCP + structure || probabilities [+ scorer state])
      const bool two = !SMALLV && nt < 3 * ne && nt >= 2 * ne;
      const bool roles = nt >= 3 * ne || two;
      // (LM tier: a fourth part -- the scorer state of the new entries, whose dependent table look-ups are the longest
      //  chain of the emission -- gets its own waves when there are enough)
      const int nroles = two ? 2 : (LM && nt >= 4 * ne) ? 4 : 3;
      const int role = roles ? (tid >= ne) + (tid >= 2 * ne) + (tid >= 3 * ne) + (tid >= 4 * ne) : -1;  // nroles and above = no part
      const bool r_lcp = role <= 0, r_struct = role < 0 || (two ? role == 0 : role == 1), r_prob = role < 0 || (two ? role == 1 : role == 2);
      const bool r_lm = LM && (nroles == 4 && roles ? role == 3 : r_prob);
      r_prob_any = r_prob;
      // Per-frame resets for the next step, on the threads that have no part in the emission (all of them otherwise):
      // the select histogram, the existing-children masks (last read in phase B), the paint buffers and counters of
      // the other parity.
      {
        const bool spare = roles && nt > nroles * ne;
        if (!spare || tid >= nroles * ne) {
          const int t0 = spare ? tid - nroles * ne : tid, tstep = spare ? nt - nroles * ne : nt;
          if (kSpec) {  // the hot list's keys (its unused tail must read as zero) and the survivor bitmap; the histogram is
                        // cleared by the frames that build one (rehistogram)
            for (int i = t0; i < kHotCap + 64; i += tstep) w.list[i] = 0u;
            for (int i = t0; i < 2 * ((kClsSlots + 63) / 64); i += tstep) w.bitmap[i] = 0u;
          } else {
            for (int i = t0; i < kBins; i += tstep) w.bins[i] = 0;
          }
          for (int i = t0; i < (SMALLV == 1 ? n : 2 * n); i += tstep) w.hit[i] = 0;
          int *oa = w.ancbuf + ((in.t + 1) & 1) * K, *oc = w.acntbuf + ((in.t + 1) & 1) * K;
          for (int i = t0; i < K; i += tstep) { oa[i] = -1; oc[i] = 0; }
          if (t0 == 0) {
            reset_pvars(pvars(in.t + 1));
            if (kSpec) { w.vars[VAR_G] = 0; w.vars[VAR_E] = 0; }
          }
          if (kSpec && tid == x.spec_thread()) spec_learn(N > K, tau, hot, K);
        }
      }
      for (int k = roles ? tid - role * ne : tid; k < n_new && role < nroles; k += roles ? ne : nt) {
        const int s = surv[k];
        const uint32_t inf = lazy_info ? sinf[k] : w.sinfo[s];
        const uint32_t type = info_type(inf);
        const int j = info_entry(inf);
        if (r_lcp) {
          // LCP with the previous survivor: the LCA depth of two candidates is the LCA depth of the entries they hang
          // off (a brand-new child never lies on an existing path), capped by the depth of a revived interior node.
          int l = -1;
          if (k > 0) {
            const uint32_t pinf = lazy_info ? sinf[k - 1] : w.sinfo[surv[k - 1]];
            const int pj = info_entry(pinf);
            l = kLcpTable ? lca_depth_tbl(pj, j) : lca_depth(pj, j);
            if (type == T_REVIVED) { const int dx = b.dep[w.anc[j]] + 1; l = dx < l ? dx : l; }
            if (info_type(pinf) == T_REVIVED) { const int dx = b.dep[w.anc[pj]] + 1; l = dx < l ? dx : l; }
          }
          nb.lcp[k] = l;
        }
        // An entry that stays (T_SELF) and a brand-new child (T_CHILD, of entry j) both draw everything from entry j:
        // one batch of loads, then selects -- no divergence between the two kinds, which share waves.  A revived
        // interior node (rare) hangs off the nearest in-beam ancestor of j instead and overrides afterwards.
        const bool self = type == T_SELF, child = type == T_CHILD;
        const int c = self ? -1 : info_ch(inf);
        if (r_struct) {
          const int node_j = b.node[j], par_j = b.par[j], ch_j = b.ch[j], dep_j = b.dep[j];
          const int via_j = b.via[j], viaanc_j = b.viaanc[j], viach_j = b.viach[j], up_j = b.up[j];
          const int id = pool_count + k;                                              // ids by beam position (gaps are harmless)
          const int upv = (dep_j & (kExpress - 1)) == 0 ? node_j : up_j;
          int o_node = self ? node_j : id, o_par = self ? par_j : node_j, o_ch = self ? ch_j : c;
          int o_dep = self ? dep_j : dep_j + 1, o_viaanc = self ? viaanc_j : -1, o_up = self ? up_j : upv;
          // A new child presets the dead-interior cache (phase A2) for the commonest case: its parent entry j leaves the
          // beam while j's own parent stays -- then the nearest in-beam ancestor is node par_j and the child of that node
          // on the way down is node_j, reached by label ch_j.  (The cache is keyed by the ancestor's node: a preset that
          // does not apply is simply never matched.)
          int o_via = via_j, o_viach = viach_j;
          // (when j's own parent is not in the beam, j's cache entry is the one that will apply to the child as well)
          // (kept to the scorer's instantiations: beams on random rows have next to no dead interior nodes -- tools/beam_stats.py --
          //  and the north-star kernel is not to pay an LDS read for them)
          if (LM) {
            if (child && (w.pinr[j] >= 0 || par_j < 0)) { o_viaanc = par_j; o_via = node_j; o_viach = ch_j; }
            else if (child) o_viaanc = viaanc_j;
          }
          if (child) {                                                                // path_trie.cpp:97-105
            PoolNode pn; pn.parent = node_j; pn.cht = PoolNode::pack(c, in.t); pn.lpc = w.clp[rank_of_char(in, c)];
            pool[id] = pn;
            if (CTC_RARE(long_t)) pool_thi[id] = in.t >> 16;
            // (up(X) is read back only from express nodes -- the back-trace hops from one multiple of kExpress levels to
            //  the next -- so only those store it: 1 node in kExpress)
            if (((dep_j + 1) & (kExpress - 1)) == 0) pool_up[id] = upv;
          } else if (CTC_RARE(!self)) {                                                         // path_trie.cpp:50-56 : revived
            const int P = w.anc[j];
            x.count(EV_REVIVED, 1);
            o_node = via_j; o_par = b.node[P]; o_dep = b.dep[P] + 1;
            // up(revived node), as for a new child of P: P's node when P sits on an express level, else P's own express pointer
            o_up = (b.dep[P] & (kExpress - 1)) == 0 ? b.node[P] : b.up[P];
          }
          nb.node[k] = o_node; nb.par[k] = o_par; nb.ch[k] = o_ch; nb.dep[k] = o_dep;
          nb.via[k] = o_via; nb.viaanc[k] = o_viaanc; nb.viach[k] = o_viach; nb.up[k] = o_up;  // via/viach: only read when viaanc matches
        }
        if (r_prob) {
          const float b_n = w.b_new[j], nb_n = w.nb_new[j], sc_n = w.sc_new[j], lpc_j = b.lpc[j];
          const int ch_j = b.ch[j];
          const float score_j = b.score[j], bprev_j = b.bprev[j];
          float o_b = b_n, o_nb = nb_n, o_sc = sc_n, o_lpc = lpc_j;
          if (!self) {              // a new or revived prefix starts from its first path only (path_trie.cpp:52-56, 99-104)
            const float lp = w.clp[rank_of_char(in, c)];
            float logp;
            if (child) {                                                              // ctc_beam_search_decoder.cpp:110-118
              const float rep = bprev_j > CTC_NEG_MAX ? lp + bprev_j : CTC_NEG_MAX;
              logp = c == ch_j ? rep : lp + score_j;
              o_lpc = lp;
            } else {
              logp = child_logp(w.anc[j], c, lp);
              o_lpc = w.rev_lpc[j];
            }
            if (LM && lm_scores(c)) logp = lm_apply(logp, lm_window(b, child ? j : w.anc[j], c));  // as scored in phase B
            o_b = CTC_NEG_MAX; o_nb = logp; o_sc = logp;
          }
          nb.bprev[k] = o_b; nb.nbprev[k] = o_nb; nb.score[k] = o_sc; nb.lpc[k] = o_lpc;
        }
        if (r_lm) lm_emit(b, (self || child) ? j : w.anc[j], self ? -1 : c, nb, k);
        if (r_prob) {
          const uint32_t ks = keys_in_ord ? (uint32_t)ord[k] : w.skey[s];
          kloc = ks > kloc ? ks : kloc;
          kmin = ks < kmin ? ks : kmin;
        }
      }
    }
    if (x.uni((int)r_prob_any)) {  // only the waves that handled probabilities
      x.wave_max_to(&pv[P_NMAXKEY], kloc);
      if (LM) x.wave_min_to(&pv[P_NMINKEY], kmin);
    }
    // the order std::nth_element left the survivors in (identity when it was not called: then the flag is looked up here)
    if (CTC_RARE(last || exact || (N <= K && x.uni(w.vars[VAR_DANGER]) != 0))) {
      int *fin = fin_nxt();  // (current once this frame commits)
      for (int q = tid; q < n_new; q += nt) {
        const int r = exact ? rk[q] : q;
        fin[q] = r;
        w.apos[r] = q;
      }
    } else if (lm_cb) {  // (two copies: a frame that leaves the order alone carries it over)
      const int *fc = fin_cur();
      int *fn = fin_nxt();
      for (int q = tid; q < d.K; q += nt) fn[q] = fc[q];
    }
    // un-register this step's candidates from the rank table -- only once every wave has finished emitting (the emit
    // loop above still looks characters up in it)
    if (!IDENT && !kRankEpoch) {
      x.sync();
      for (int r = tid; r < Vc; r += nt) w.rank_of[w.cch[r]] = -1;
    }
    if (stage && tid < d.V) w.clpbuf[((in.t + 1) & 1) * d.Vc_max + tid] = stage_val;
    if ((kSpec || kHotPre) && stage) x.row_max_store(&w.vars[VAR_ROWMAX], stage_val, d.V);
    x.mark(7);
    x.sync_full();  // pool writes of this step (global memory) are visible to every wave from here on
    x.mark(9);
    // host-side scorer hook: the frame asked for something the cache does not hold -- nothing of it is committed (the next
    // beam lives in the other copy, its nodes beyond the pool's count; the few things a frame changes in place -- a prefix's
    // best label probability and its node's time stamp -- are rewritten identically when the frame runs again)
    if (LM && CTC_RARE(lm_cb) && x.uni(w.vars[VAR_LMMISS]) != 0) return ST_NEED_HOST;
    if (kTailZero && LM) {  // (the emission has read its survivors' keys: nothing looks at this frame's slots any more)
      const int S_next = n_new * (2 + Vnb);
      if (CTC_RARE(S_next < S))
        for (int i = S_next + tid; i < S; i += nt) w.skey[i] = 0u;
    }
    // every thread advances its copy of the step state
    {
      // next select window.  It is anchored at this step's best key (an upper bound for the next step's keys when
      // log-probabilities are <= 0) and must reach down to the next K-th key: twice the distance from THIS step's
      // anchor (the previous best key) to this step's K-th key, rounded up to a power of two.
      if (!kSpec) {  // (speculative select: one thread keeps the window and the prediction in LDS -- spec_learn / spec_predict)
        int wl = 32;
        st_gap = 0;
        if (N > K) {
          const uint32_t gap = st_maxkey > tau ? st_maxkey - tau : 0;
          wl = (gap ? 32 - __builtin_clz(gap) : 0) + 1;  // = ceil(log2(gap + 1)) + 1
          wl = wl < 10 ? 10 : (wl > 32 ? 32 : wl);
          st_gap = kHotPre ? (hot_anchor > tau ? hot_anchor - tau : 0u) : gap;
        }
        st_wlog = wl;
        st_maxkey = (uint32_t)x.uni(pv[P_NMAXKEY]);
      }
      if (LM) st_minkey = (uint32_t)x.uni(pv[P_NMINKEY]);
      st_n = n_new;
      st_pool = pool_count + n_new;
    }
    x.tick();
    x.dump(in.t, n_new, nb.node, nb.dep, nb.lcp, nb.score);
    x.mark(8);
    st_par ^= 1;
    return ST_OK;
  }

  // == std::sort(v, v + n, before) of libstdc++ (stl_emul.h), element for element -- also where `before` ties.
  // Introsort's sub-ranges are disjoint once split, so the order in which they are finished cannot change the
  // result: every lane takes one pending range per round (median-of-three Hoare split, or the heap sort once the depth
  // budget is spent), and the closing insertion sort -- which never moves an element across a split point, because
  // the comparison is strict -- is done per final range (at most 16 elements) by one lane each.
  template <class T, class C>
  CTC_HD void sort_like_std(T *v, int n, C before) {
    const int tid = x.tid(), nt = x.nt();
    constexpr int kTaskCap = (kBins + kBins / 16) / 6;
    if (n <= 16 || n / 17 + 1 > kTaskCap) {
      if (tid == 0) stlemu::sort(v, 0, n, before, w.sstack);
      x.sync();
      return;
    }
    stlemu::sort_parallel(x, v, n, before, w.bins, w.bins + 3 * kTaskCap, w.surv, w.vars + VAR_TAU);
  }

  // DecoderState::decode() + get_beam_search_result + binding.cpp:85-99 for one utterance.
  // `had_steps`: false when the utterance has no frames (fin is then just the root); `max_depth`: bound on the length
  // of a label sequence (the number of frames fed).
  // Returns ST_OK, or ST_COMPACT_OVERFLOW when the compact label buffer is too small (identical in every thread).
  CTC_HD int finish(bool had_steps, int max_depth, const OutRefs *outs, int item) {
    const OutRefs o = *x.fresh(outs);
    const int T_stride = o.T_stride;
    const size_t ko = (size_t)o.K * (size_t)o.T_stride;
    int32_t *out_tok = o.tok + (size_t)item * ko, *out_ts = o.ts + (size_t)item * ko;
    float *out_score = o.score + (size_t)item * o.K;
    int32_t *out_len = o.len + (size_t)item * o.K;
    int32_t *n_results = o.n_results ? o.n_results + item : nullptr;
    select_beams();
    const Beam &b = w.cur;
    const int tid = x.tid(), nt = x.nt();
    const int n = st_n;
    if (SMALLV) { CTC_ASSUME(n >= 1 && n <= kClsK); CTC_ASSUME(d.K <= kClsK); }
    const int nres = n < d.K ? n : d.K;
    if (!had_steps)
      for (int k = tid; k < nres; k += nt) fin_cur()[k] = k;
    x.sync();
    // The two std::sorts (ctc_beam_search_decoder.cpp:188-190, decoder_utils.cpp:59) order by (score desc, character
    // asc).  Equal float32 scores are common in a beam (it spans a few hundred ulps), so the order libstdc++ leaves
    // equivalent prefixes in matters: both sorts are replayed exactly, on (key, entry) pairs packed into one word.
    float *ext = w.sc_new;       // LM tier: score + last word's LM score (the map `scores` of decode(), :168-185)
    float *approx = w.b_new;     //          PathTrie::approx_ctc (:194-208)
    {
      const float *sc = b.score;
      const int *ch = b.ch;
      uint64_t *pk = sort_scratch();
      if (LM) {
        for (int a = tid; a < n; a += nt) {
          // (callback scorer: an entry whose window is not cached yet has no LM fields -- the launch ends below and finish()
          //  runs again once the host has answered; asking with what the fields happen to hold would queue nonsense)
          if (CTC_RARE(lm_cb) && b.dfc[a] == kLmPending) continue;
          // the word the prefix ends in, when it does not end in a space (:173-185; word models only)
          const bool partial = !lm_char_() && b.dep[a] > 0 && ch[a] != lm_space;
          const bool word_here = partial && lm_allows(b, a, lm_space);  // a word of the model ends exactly here
          const double wcond = word_here ? mk_f64(b.spc_lo[a], b.spc_hi[a]) : ctclm::kOovScore;
          float e = sc[a];
          if (partial) {
            float score = 0.0f;
            score = (float)(wcond * lm_alpha);
            score = (float)((double)score + lm_beta);
            e += score;
          }
          ext[a] = e;
          // Scorer::get_sent_log_prob of the prefix's words (scorer.cpp:95-120): the windows of the completed words are
          // already summed in acc; then the word it ends in, then "</s>"
          double total = mk_f64(b.acc_lo[a], b.acc_hi[a]);
          uint32_t st = (uint32_t)b.lmst[a];
          int cl = b.lmcl[a];
          if (b.dep[a] == 0) {               // empty prefix: the sentence is N x "<s>" then "</s>" (:97-100)
            total += lm_cond_(&st, &cl, lm->w_bos);
          } else if (partial) {
            total += wcond;
            if (word_here) { st = (uint32_t)b.spst[a]; cl = b.spcl[a]; }
            else {
              // the word is not in the vocabulary: the window of "</s>" starts behind it.  (Orders above 1: an unknown word
              // inside the window makes it OOV without a query.  A callback scorer of order 1 -- windows of one word, no
              // history -- does query: from ITS empty history, lm->s0; state 0 is the built-in automaton's and never a key of
              // the callback's cache: ADVICE r4.)
              st = (lm_cb && lm->clean0 == 0) ? lm->s0 : 0u;
              cl = 0;
            }
          }
          total += lm_cond_(&st, &cl, lm->w_eos);
          double ap = (double)e;
          ap = ap - (double)(size_t)b.dep[a] * lm_beta;   // "remove word insert": per label (:203)
          ap -= total * lm_alpha;                          // :205
          approx[a] = (float)ap;
        }
        x.sync();
        if (CTC_RARE(lm_cb) && x.uni(w.vars[VAR_LMMISS]) != 0) return ST_NEED_HOST;  // (nothing has been written to the results yet)
      }
      for (int p = tid; p < nres; p += nt) {
        const int a = fin_cur()[p];
        pk[p] = (key48(ord_f32(LM ? ext[a] : sc[a]), mk_info(ch[a], 0, 0)) << 16) | (uint64_t)a;
      }
      x.sync();
      auto before = [](uint64_t a, uint64_t c) { return (a >> 16) > (c >> 16); };
      sort_like_std(pk, nres, before);   // :188-190, by the scores map
      if (LM) {                          // decoder_utils.cpp:59 sorts by the RAW score
        for (int p = tid; p < nres; p += nt) {
          const int a = (int)(pk[p] & 0xFFFFu);
          pk[p] = (key48(ord_f32(sc[a]), mk_info(ch[a], 0, 0)) << 16) | (uint64_t)a;
        }
        x.sync();
      }
      sort_like_std(pk, nres, before);
      for (int p = tid; p < nres; p += nt) fin_cur()[p] = (int)(pk[p] & 0xFFFFu);
    }
    if (tid == 0 && n_results) *n_results = nres;
    x.sync();
    for (int p = tid; p < nres; p += nt) {
      const int j = fin_cur()[p];
      out_score[p] = LM ? -approx[j] : -b.score[j];  // decoder_utils.cpp:68 (approx_ctc = score without a scorer)
      out_len[p] = b.dep[j];
    }
    // path_trie.cpp:113-126 (get_path_vec).  Neighbours in the (DFS-ordered) beam share their first lcp labels, so
    // every entry j reads back only the labels below that shared part -- depths (lcp[j], dep[j]] -- and the shared
    // part of its row is then copied from the rows that did read it.  The read-back itself goes segment by segment
    // (kExpress): segment 0 is the tail above the node's express ancestor, segment i >= 1 the kExpress labels below
    // the i-th express ancestor; segment-major order keeps the walks of one wave equally long.
    int *row_of = w.pinr, *owner = w.e;
    for (int p = tid; p < nres; p += nt) row_of[fin_cur()[p]] = p;
    for (int j = tid; j < nres; j += nt) {  // owner[j]: the nearest earlier entry that shares less with its own predecessor
      int o = j - 1;
      const int l = b.lcp[j];
      while (o >= 0 && b.lcp[o] >= l) --o;
      owner[j] = o;
    }
    x.sync();
    x.mark(4);
    const bool compact = o.c_hdr != nullptr;
    unsigned cbase = 0;
    uint32_t ctotal = 0;
    if (compact) {
      for (int j = tid; j < nres; j += nt) {
        const int lo = b.lcp[j] > 0 ? b.lcp[j] : 0;
        w.pos[j] = (uint32_t)(b.dep[j] > lo ? b.dep[j] - lo : 0);
      }
      if (tid == 0) w.pos[nres] = 0;
      x.sync();
      const uint32_t total = x.scan_excl(w.pos, nres + 1);
      if (tid == 0) w.vars[VAR_CUT] = (int)x.global_add(o.c_count, total);
      x.sync();
      cbase = (unsigned)x.uni(w.vars[VAR_CUT]);
      ctotal = total;
      const bool fits = cbase + total <= o.c_cap && cbase + total >= cbase;
      if (tid == 0) {
        int32_t *h = o.c_hdr + (size_t)item * 4;
        h[0] = fits ? nres : 0; h[1] = fits ? (int32_t)total : 0; h[2] = (int32_t)cbase; h[3] = 0;
        if (!fits) w.vars[VAR_STATUS] = ST_COMPACT_OVERFLOW;
      }
      if (!fits) return ST_COMPACT_OVERFLOW;
      for (int j = tid; j < nres; j += nt) {
        int32_t *e = o.c_ent + ((size_t)item * o.K + j) * 4;
        e[0] = row_of[j]; e[1] = b.lcp[j] > 0 ? b.lcp[j] : 0; e[2] = b.dep[j]; e[3] = (int32_t)(cbase + w.pos[j]);
      }
    }
    const int maxseg = max_depth / kExpress + 1;
    for (int idx = tid; idx < nres * maxseg; idx += nt) {
      const int i = idx / nres, j = idx - i * nres;
      const int dj = b.dep[j];
      const int lo = b.lcp[j] > 0 ? b.lcp[j] : 0;
      const int base = ((dj - 1) / kExpress) * kExpress;  // depth of the first express ancestor
      if (dj <= 0 || i > base / kExpress) continue;
      int dd = i == 0 ? dj : base - (i - 1) * kExpress;
      int stop = i == 0 ? base : dd - kExpress;
      if (dd <= lo) continue;
      stop = stop < lo ? lo : stop;
      int xn;
      if (i == 0) {
        xn = b.node[j];
      } else {
        xn = b.up[j];
        for (int h = 1; h < i; ++h) xn = pool_up[xn];
      }
      if (compact) {
        uint32_t *seg = o.c_rag + (size_t)cbase + w.pos[j] - lo;  // label at depth q (lo < q <= dj) sits at seg[q - 1]
        while (dd > stop) {
          const PoolNode pn = pool[xn];
          seg[dd - 1] = pn.cht;  // label | time step << 16: the compact format's own packing (T <= 65536 there)
          xn = pn.parent;
          --dd;
        }
        continue;
      }
      const size_t row = (size_t)row_of[j] * T_stride;
      int32_t *tk = out_tok + row, *ts = out_ts + row;
      while (dd > stop) {
        const PoolNode pn = pool[xn];
        tk[dd - 1] = pn.ch();
        ts[dd - 1] = node_tstep(pn, xn);
        xn = pn.parent;
        --dd;
      }
    }
    if (compact) {
      if (o.m_done) {  // hand the finished utterance to the host right away
        x.sync_full();  // (its labels are complete in c_rag)
        const bool mirror = cbase + ctotal <= o.m_cap;
        if (mirror) {
          for (uint32_t i = (uint32_t)tid; i < ctotal; i += (uint32_t)nt) o.m_rag[cbase + i] = o.c_rag[cbase + i];
          const int32_t *es = o.c_ent + (size_t)item * o.K * 4;
          int32_t *ed = o.m_ent + (size_t)item * o.K * 4;
          for (int i = tid; i < nres * 4; i += nt) ed[i] = es[i];
          if (tid < 4) o.m_hdr[(size_t)item * 4 + tid] = o.c_hdr[(size_t)item * 4 + tid];
        }
        x.fence_system();
        x.sync_full();
        if (tid == 0) x.store_system(&o.m_done[item], mirror ? 1 : 2);
      }
      return ST_OK;
    }
    x.sync_full();  // rows are read back below by other waves
    x.mark(15);
    const int grp = x.group(), ngr = x.ngroups(), lane = x.lane(), lanes = x.lanes();
    constexpr int kU = 8;  // labels per lane and round: that many loads in flight
    for (int j = 1 + grp; j < nres; j += ngr) {
      const size_t row = (size_t)row_of[j] * T_stride;
      const int hi = b.lcp[j];
      const int o0 = owner[j];
      for (int q0 = lane; q0 < hi; q0 += lanes * kU) {
        int32_t tk[kU], ts[kU];
        for (int u = 0; u < kU; ++u) {
          const int q = q0 + u * lanes;
          if (q < hi) {
            int o = o0;
            while (b.lcp[o] > q) o = owner[o];  // label q of row j was read back by entry o (lcp[0] = -1 ends the walk)
            const size_t src = (size_t)row_of[o] * T_stride + q;
            tk[u] = out_tok[src];
            ts[u] = out_ts[src];
          }
        }
        for (int u = 0; u < kU; ++u) {
          const int q = q0 + u * lanes;
          if (q < hi) { out_tok[row + q] = tk[u]; out_ts[row + q] = ts[u]; }
        }
      }
    }
    return ST_OK;
  }
};

// Candidate lists of a pruned utterance (produced by the vocabulary-prune pass, decoder_utils.cpp:10-45):
// cnt[t] candidates at step t, stored at ch/lp[t * stride + r] in the reference's order (descending probability).
struct PrunedRows {
  const int *cnt;
  const int *ch;
  const float *lp;
  int stride;
};

// Whole utterance: `rows` = [len, V] float32 log-probabilities (identity mode) or nullptr with `pr` set.
// LM tier: `lm` = the scorer's tables, `raw` = the caller's own [len, V] rows (log-probabilities or probabilities,
// `raw_log` says which): ctc_beam_search_decoder.cpp:78 takes the blank's log-probability from them directly.
template <bool IDENT, int SMALLV = 0, bool LM = false, bool LAZY = false, bool FARREP = LAZY, bool HUGE = false, bool WORDLM = false, bool CB = false, class X>
CTC_HD int decode_utterance(X &x, Work &w, const Dims &d, int blank, const float *rows, const PrunedRows *pr, int len,
                            PoolNode *pool, int *pool_up, int pool_cap, const uint64_t *tbl, const OutRefs *outs, int item,
                            const StreamState *ss = nullptr, const ctclm::LmView *lm = nullptr, const float *raw = nullptr,
                            int raw_log = 1, const int *frames_ready = nullptr) {
  if (SMALLV == 1) { CTC_ASSUME(d.K >= 1 && d.K <= kSmallK); CTC_ASSUME(d.V >= 1 && d.V <= kSmallV); CTC_ASSUME(d.Vc_max >= 1 && d.Vc_max <= kSmallV); CTC_ASSUME(blank >= 0 && blank < kSmallV); }
  if (SMALLV == 2) { CTC_ASSUME(d.K >= 1 && d.K <= kMidK); CTC_ASSUME(d.V >= 1 && d.V <= kMidV); CTC_ASSUME(d.Vc_max >= 1 && d.Vc_max <= kMidVc); }
  using Dec0 = Decoder<X, IDENT, SMALLV, LM, LAZY, FARREP, HUGE, WORDLM, CB>;
  Dec0 dec(x, w, d, blank, pool, pool_up, pool_cap, tbl, lm);
  // a stream continues where its previous chunk stopped: frame numbers (the `timesteps` output) keep counting
  const int t0 = ss ? x.uni(ss->hdr[SH_FRAMES]) : 0;
  dec.long_t = (long long)t0 + len > 65536;  // (frame numbers 0 .. 65535 fit the node's 16 bits)
  if (t0 > 0) dec.load_state(*ss); else dec.init();
  dec.prime_a1(t0);
  const int tid = x.tid(), nt = x.nt();
  // Prefetch: the candidates of step t+1 are requested from HBM before step t runs, so the latency hides behind it.
  const int width = IDENT ? d.V : pr->stride;
  const bool prefetch = width <= nt;
  float pre_lp = 0.f;
  int pre_ch = 0, pre_cnt = 0;
  // Streamed input (frames_ready != null; identity mode with prefetch only): the rows are still crossing PCIe, frame block
  // by frame block, while this kernel runs; *frames_ready (uncached memory, written by the copy stream behind every
  // block) says how many frames of every utterance have arrived.  The threads that fetch a row wait for it -- once per
  // block; a wait that lasts absurdly long gives up (status ST_INPUT_TIMEOUT) rather than hang the GPU.
  // Giving up is sticky: the thread that gave up never spins again, and the workgroup leaves the frame loop at
  // the next check (every sixteenth frame) -- a launch whose input never arrives costs about a second, not a second per
  // frame.
  int ready_cached = 0;
  bool gave_up = false;
  auto wait_frames = [&](int need) {
    if (frames_ready != nullptr && need > ready_cached && !gave_up) {
      for (int spins = 0;; ++spins) {
        ready_cached = x.load_system(frames_ready);
        if (ready_cached >= need) break;
        if (spins > (1 << 20)) { w.vars[VAR_INTO] = 1; gave_up = true; break; }  // (~1 s)
        x.nap();
      }
    }
  };
  if (prefetch && len > 0) {
    if (!IDENT) {
      pre_cnt = pr->cnt[0];
      if (tid < width) { pre_ch = pr->ch[tid]; pre_lp = pr->lp[tid]; }
    } else if (tid < width) {
      wait_frames(1);
      pre_lp = rows[tid];
    }
  }
  if (IDENT && prefetch && len > 0) {  // frame 0 goes straight to LDS; from then on step() stages frame t+1
    if (tid < d.V) { w.clpbuf[(t0 & 1) * d.Vc_max + tid] = pre_lp; dec.note_lp(pre_lp); }  // (the launch's first row: checked here)
    if (Dec0::kSpec || Dec0::kHotPre) x.row_max_store(&w.vars[VAR_ROWMAX], pre_lp, d.V);
    x.sync();
  }
  for (int t = 0; t < len; ++t) {
    StepIn in;
    in.t = t0 + t;
    in.blank_prob = 0.f;
    if (LM) {
      const float rb = raw[(size_t)t * d.V + blank];
      in.blank_prob = raw_log ? rb : (float)ctc_log_f64((double)rb);  // float blank_prob = log_input ? p : std::log(p)
    }
    x.trace_frame(in.t);
    bool stage = false;
    // danger mode looks one row ahead: row t + 1 is examined while frame t is decoded (row 0 where it is loaded)
    int next_cnt = 0;  // threads below it hold a value of row t + 1 in next_val
    float next_val = 0.f;
    using Dec = Dec0;
    if (IDENT) {
      in.Vc = d.V;
      in.identity = 1;
      in.blank_rank = blank;
      if (prefetch) {
        w.clp = w.clpbuf + ((t0 + t) & 1) * d.Vc_max;
        stage = t + 1 < len;
        if (stage && tid < d.V) {
          wait_frames(t + 2);
          pre_lp = rows[(size_t)(t + 1) * d.V + tid];  // consumed at the end of this frame
        }
        next_cnt = stage ? d.V : 0;
        next_val = pre_lp;
      } else {
        wait_frames(t + 2 < len ? t + 2 : len);  // (rows that are not prefetched are read here, this frame's and a look at the next)
        for (int r = tid; r < d.V; r += nt) {
          const float v = rows[(size_t)t * d.V + r];
          w.clp[r] = v;
          if (t == 0) dec.note_lp(v);
          if (t + 1 < len && Dec::lp_bad(rows[(size_t)(t + 1) * d.V + r])) { next_cnt = tid + 1; next_val = -__builtin_huge_valf(); }
        }
        x.sync();
        if (Dec0::kSpec || Dec0::kHotPre) {  // (rows that are not prefetched -- workgroups narrower than the vocabulary: the host build of the tests)
          if (tid == 0) {
            float mx = w.clp[0];
            for (int r = 1; r < d.V; ++r) mx = w.clp[r] > mx ? w.clp[r] : mx;
            w.vars[VAR_ROWMAX] = (int)ctcmath::f32_to_bits(mx);
          }
          x.sync();
        }
      }
    } else {
      in.identity = 0;
      if (prefetch) {
        in.Vc = x.uni(pre_cnt);
        if (Dec0::kRankEpoch && CTC_RARE((in.t & 1023) == 0 && t > 0)) {  // (the tags are about to repeat: the last frame's barrier is behind us)
          for (int c = tid; c < d.V; c += nt) w.rank_of[c] = -1;
          x.sync();
        }
        if (tid < in.Vc) {
          w.cch[tid] = pre_ch; w.clp[tid] = pre_lp; w.rank_of[pre_ch] = Dec0::kRankEpoch ? Dec0::rank_tag(in.t, tid) : (int16_t)tid;
          if (t == 0) dec.note_lp(pre_lp);
          // (speculative select: the anchor of this frame's prediction -- the list is in descending order, decoder_utils.cpp:23-24)
          if ((Dec0::kSpec || Dec0::kHotPre) && tid == 0) w.vars[VAR_ROWMAX] = (int)ctcmath::f32_to_bits(pre_lp);
        }
        if (t + 1 < len) {
          pre_cnt = pr->cnt[t + 1];
          if (tid < width) { pre_ch = pr->ch[(size_t)(t + 1) * width + tid]; pre_lp = pr->lp[(size_t)(t + 1) * width + tid]; }
          next_cnt = pre_cnt;
          next_val = pre_lp;
        }
      } else {
        in.Vc = x.uni(pr->cnt[t]);
        for (int r = tid; r < in.Vc; r += nt) {
          const int c = pr->ch[(size_t)t * width + r];
          const float v = pr->lp[(size_t)t * width + r];
          w.cch[r] = c;
          w.clp[r] = v;
          w.rank_of[c] = Dec0::kRankEpoch ? Dec0::rank_tag(in.t, r) : (int16_t)r;
          if (t == 0) dec.note_lp(v);
          if ((Dec0::kSpec || Dec0::kHotPre) && r == 0) w.vars[VAR_ROWMAX] = (int)ctcmath::f32_to_bits(v);
        }
        if (t + 1 < len) {
          const int cn = x.uni(pr->cnt[t + 1]);
          for (int r = tid; r < cn; r += nt)
            if (Dec::lp_bad(pr->lp[(size_t)(t + 1) * width + r])) { next_cnt = tid + 1; next_val = -__builtin_huge_valf(); }
        }
      }
      x.sync();
      in.blank_rank = x.uni(dec.rank_of_char(in, blank));
    }
    x.tick();
    x.mark(10);
    const int st = dec.step(in, t == len - 1, stage, pre_lp, next_cnt, next_val);
    x.mark(12);
    if (LM && CTC_RARE(st == ST_NEED_HOST)) {  // park the utterance in front of this frame; the host fills the cache and resumes it
      dec.lm_resolve_pending();                 // (queues what the pending entries need as well: fewer round trips)
      if (ss) dec.save_state(*ss, t0 + t);
      return st;
    }
    if (st != ST_OK) return st;
    // (streamed input that stopped arriving: step() ended with a full fence, the flag is visible to every wave)
    if (frames_ready != nullptr && (t & 15) == 15 && x.uni(w.vars[VAR_INTO]) != 0) return ST_INPUT_TIMEOUT;
  }
  if (LM) dec.lm_resolve_pending();
  if (ss) dec.save_state(*ss, t0 + len);
  int fs = ST_OK;
  if (!ss || ss->finish) fs = dec.finish(t0 + len > 0, t0 + len, outs, item);
  dec.shape_stat();
  x.sync();
  x.mark(11);
  if (frames_ready != nullptr && x.uni(w.vars[VAR_INTO]) != 0) return ST_INPUT_TIMEOUT;
  return fs;
}

}  // namespace ctcbeam
