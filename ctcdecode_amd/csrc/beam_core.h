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
//     (decoder_utils.cpp:122-132), found by a histogram select (256 buckets over a window below the best score,
//     then an exact rank inside the one bucket that holds the K-th key).  If the K boundary cuts through a group of
//     EQUAL keys -- structural at long T, SURVEY.md 7.3-H2 -- or on the last step (whose permutation feeds the final
//     sorts), the workgroup replays libstdc++'s std::nth_element on the DFS-ordered candidate list (Hoare partitions
//     done in parallel, the small tail by one lane with stl_emul.h), so the same prefixes survive as in the
//     reference.  Survivors are compacted in slot order by one fused scan that also produces the new LCP array,
//     which keeps the DFS-order invariant.
//   * Trie nodes that survive a step are appended to a per-utterance pool in HBM {parent, char, timestep,
//     log_prob_c}; nothing transient is ever materialised (the reference news/deletes ~2.8k nodes per step).
//     The pool is read back only for (rare) dead-interior lookups and for the final back-trace.
//   * Scores use the bit-exact float32 log_sum_exp of exact_math.h.
//
// The code is written against an execution policy X (thread id, thread count, barrier, block reductions):
// kernels.hip instantiates it with one workgroup per utterance; tests/native/core_host.cpp instantiates it with a
// single sequential "thread" so the very same source is differential-tested against the oracle on the CPU
// (test infrastructure only -- the product has no CPU path).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "exact_math.h"
#include "stl_emul.h"

namespace ctcbeam {

struct PoolNode {   // one alive-or-retired trie node in HBM (16 bytes, one dwordx4 access)
  int32_t parent;   // pool index, -1 for the root
  int32_t ch;       // label, -1 for the root
  int32_t tstep;    // time step of the best log_prob_c seen while the node lived (path_trie.cpp:42-45)
  float lpc;        // that log_prob_c
};

enum : uint32_t { T_SELF = 0, T_CHILD = 1, T_REVIVED = 2, T_HOLE = 3 };
enum : int { ST_OK = 0, ST_POOL_OVERFLOW = 1, ST_BAD_CONFIG = 2 };

constexpr int kMaxBeam = 16383;   // 14-bit entry index inside a slot's info word
constexpr int kMaxVocab = 65534;  // 16-bit (character + 1)
constexpr int kIntMax = 0x7fffffff;

// Order-preserving map float -> uint32 (larger float = larger integer); -0.0 and +0.0 coincide, as they do
// under the reference's operator== / operator> on float scores.
CTC_HD uint32_t ord_f32(float f) {
  uint32_t u = ctcmath::f32_to_bits(f);
  if (u == 0x80000000u) u = 0;
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
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
  float *bprev, *nbprev, *score, *lpc;
};

struct Dims {
  int K;       // beam width
  int V;       // vocabulary size
  int Vc_max;  // most candidate characters a step can have (V, or cutoff_top_n when pruning)
  int use_rank_table;  // 1 when candidate lists are pruned (rank_of[] needed)
  CTC_HD int S_max() const { return K * (2 + Vc_max); }
};

enum {
  VAR_N = 0, VAR_POOL, VAR_DMIN, VAR_STATUS, VAR_MAXKEY, VAR_WLOG,          // persistent across steps
  VAR_VALID, VAR_LCOUNT, VAR_FB0, VAR_FB1, VAR_FB2, VAR_FB3, VAR_CUT,      // scratch within a step
  VAR_TAU_LO, VAR_TAU_HI, VAR_G, VAR_E, VAR_COUNT = 24
};
constexpr int kBins = 256;      // histogram buckets of the select
constexpr int kListCap = 128;   // exact-rank list (one bucket's keys)
constexpr int kSerialCut = 96;  // introselect ranges at most this long are finished by one lane

// What the fused compaction scan returns to every thread (see X::seg_scan).
struct SegOut {
  int excl, exclc;    // survivors / surviving new children in the chunks before this thread's
  int total, totalc;  // workgroup totals
  int carry;          // min LCP value since the last survivor before this thread's chunk
  int dmin;           // min depth over all survivors
  uint32_t maxkey;    // max score key over all survivors
};

struct Work {
  Beam cur, nxt;
  int *e, *anc, *ostart, *cstart, *hasvia, *pinr, *revr;  // per beam entry, this step
  uint32_t *hit;   // 2 words per entry: ranks (non-blank numbering) of the children that already exist
  float *b_new, *nb_new, *sc_new;
  int *cch;        // candidate characters of this step (unused in identity mode)
  float *clp;      // their log-probs
  int *rank_of;    // V entries, -1 = not a candidate (only when Dims::use_rank_table)
  uint32_t *skey, *sinfo, *pos;  // S_max (+1 for pos): score key, info word, scratch
  int *slcp;       // S_max: LCP depth of the slot with its predecessor slot
  uint64_t *ek;    // S_max: (key48 << 16 | slot) in DFS order, for the exact replay
  uint16_t *lr;    // 2 * S_max: stop positions of the parallel Hoare partition
  int *bins;       // kBins
  uint64_t *list;  // kListCap
  int *fin, *sstack;
  int *vars;
};

template <class P>
CTC_HD P *carve_ptr(char *&p, size_t count) {
  P *r = reinterpret_cast<P *>(p);
  p += ((count * sizeof(P) + 15) / 16) * 16;
  return r;
}

// Lay the workspace out in `base` (LDS on the GPU).  Returns bytes used; call with base == nullptr to size it.
CTC_HD size_t carve(Work &w, char *base, const Dims &d) {
  char *p = base;
  const size_t K = (size_t)d.K, S = (size_t)d.S_max();
  Beam *bs[2] = {&w.cur, &w.nxt};
  for (int i = 0; i < 2; ++i) {
    Beam &b = *bs[i];
    b.node = carve_ptr<int>(p, K); b.par = carve_ptr<int>(p, K); b.ch = carve_ptr<int>(p, K);
    b.dep = carve_ptr<int>(p, K); b.lcp = carve_ptr<int>(p, K); b.via = carve_ptr<int>(p, K);
    b.viaanc = carve_ptr<int>(p, K); b.viach = carve_ptr<int>(p, K);
    b.bprev = carve_ptr<float>(p, K); b.nbprev = carve_ptr<float>(p, K); b.score = carve_ptr<float>(p, K);
    b.lpc = carve_ptr<float>(p, K);
  }
  w.e = carve_ptr<int>(p, K); w.anc = carve_ptr<int>(p, K); w.ostart = carve_ptr<int>(p, K);
  w.cstart = carve_ptr<int>(p, K); w.hasvia = carve_ptr<int>(p, K); w.pinr = carve_ptr<int>(p, K);
  w.revr = carve_ptr<int>(p, K); w.hit = carve_ptr<uint32_t>(p, 2 * K);
  w.b_new = carve_ptr<float>(p, K); w.nb_new = carve_ptr<float>(p, K); w.sc_new = carve_ptr<float>(p, K);
  w.cch = carve_ptr<int>(p, (size_t)d.Vc_max);
  w.clp = carve_ptr<float>(p, (size_t)d.Vc_max);
  w.rank_of = carve_ptr<int>(p, d.use_rank_table ? (size_t)d.V : 0);
  w.skey = carve_ptr<uint32_t>(p, S); w.sinfo = carve_ptr<uint32_t>(p, S);
  w.pos = carve_ptr<uint32_t>(p, S + 2); w.slcp = carve_ptr<int>(p, S);
  w.ek = carve_ptr<uint64_t>(p, S); w.lr = carve_ptr<uint16_t>(p, 2 * S + 2);
  w.bins = carve_ptr<int>(p, kBins); w.list = carve_ptr<uint64_t>(p, kListCap);
  w.fin = carve_ptr<int>(p, K);
  w.sstack = carve_ptr<int>(p, 3 * (2 * 32 + 2));
  w.vars = carve_ptr<int>(p, VAR_COUNT);
  return (size_t)(p - base);
}

struct StepIn {
  int t;           // absolute time step
  int Vc;          // number of candidate characters
  int blank_rank;  // rank of the blank among the candidates, -1 if it was pruned away
  int identity;    // 1: candidate r is character r (no pruning)
};

CTC_HD int ceil_log2_u64(uint64_t v) {  // smallest s with (1 << s) >= v, v >= 1
  int s = 0;
  while (s < 63 && ((uint64_t)1 << s) < v) ++s;
  return s;
}

template <class X>
struct Decoder {
  X &x;
  Work &w;
  const Dims d;
  const int blank;
  PoolNode *pool;
  const int pool_cap;
  const uint64_t *tbl;  // exact_math tables

  CTC_HD Decoder(X &x_, Work &w_, const Dims &d_, int blank_, PoolNode *pool_, int pool_cap_, const uint64_t *tbl_)
      : x(x_), w(w_), d(d_), blank(blank_), pool(pool_), pool_cap(pool_cap_), tbl(tbl_) {}

  CTC_HD float lse(float a, float b) const { return ctcmath::lse(a, b, tbl); }

  // ctc_beam_search_decoder.cpp:43-44 : root prefix, score = log_prob_b_prev = 0
  CTC_HD void init() {
    if (x.tid() == 0) {
      Beam &b = w.cur;
      b.node[0] = 0; b.par[0] = -1; b.ch[0] = -1; b.dep[0] = 0; b.lcp[0] = -1;
      b.via[0] = -1; b.viaanc[0] = -1; b.viach[0] = -1;
      b.bprev[0] = 0.f; b.nbprev[0] = CTC_NEG_MAX; b.score[0] = 0.f; b.lpc[0] = CTC_NEG_MAX;
      PoolNode r; r.parent = -1; r.ch = -1; r.tstep = 0; r.lpc = CTC_NEG_MAX;
      pool[0] = r;
      w.vars[VAR_N] = 1; w.vars[VAR_POOL] = 1; w.vars[VAR_DMIN] = 0; w.vars[VAR_STATUS] = ST_OK;
      w.vars[VAR_MAXKEY] = (int)ord_f32(0.f);
      w.vars[VAR_WLOG] = 32;  // first select looks at the whole key range
    }
    if (d.use_rank_table)
      for (int c = x.tid(); c < d.V; c += x.nt()) w.rank_of[c] = -1;
    x.sync();
  }

  CTC_HD int rank_of_char(const StepIn &in, int c) const {
    if (c < 0) return -1;
    if (in.identity) return c < in.Vc ? c : -1;
    return w.rank_of[c];
  }

  // log_p of extending beam entry P with character c (ctc_beam_search_decoder.cpp:110-118)
  CTC_HD float child_logp(int P, int c, float lp) const {
    const Beam &b = w.cur;
    if (c == b.ch[P]) return b.bprev[P] > CTC_NEG_MAX ? lp + b.bprev[P] : CTC_NEG_MAX;
    return lp + b.score[P];
  }

  CTC_HD uint64_t slot_key48(int s) const { return key48(w.skey[s], w.sinfo[s]); }

  // ------------------------------------------------------------------------------------------------------ select
  // K-th largest 48-bit key among the S slots (holes have key 0).  Leaves tau (VAR_TAU_*), G = #keys > tau,
  // E = #keys == tau and VAR_VALID = #candidates in vars.  Precondition: bins[] and VAR_VALID/VAR_LCOUNT zeroed.
  CTC_HD void select_kth(int S, int K) {
    const int tid = x.tid(), nt = x.nt();
    // first window: [maxkey - 2^wlog, +inf) in score-key units, 256 buckets
    const uint32_t maxkey = (uint32_t)w.vars[VAR_MAXKEY];
    const int wlog = w.vars[VAR_WLOG];
    uint64_t lo, hi = (uint64_t)1 << 48;
    if (wlog >= 32 || ((uint64_t)1 << wlog) > (uint64_t)maxkey) lo = 1;
    else lo = ((uint64_t)maxkey - ((uint64_t)1 << wlog) + 1) << 16;
    if (lo < 1) lo = 1;
    int need = K, gbase = 0;
    bool first = true;
    for (;;) {
      uint64_t width = hi - lo;
      if (first && wlog < 32) width = (uint64_t)1 << (wlog + 16);  // buckets sized for the window, top bucket open-ended
      const int shift = width <= (uint64_t)kBins ? 0 : ceil_log2_u64(width) - 8;
      int valid = 0;
      for (int s = tid; s < S; s += nt) {
        const uint32_t inf = w.sinfo[s];
        valid += info_type(inf) != T_HOLE;
        const uint64_t k = key48(w.skey[s], inf);
        if (k >= lo && k < hi) {
          uint64_t bk = (k - lo) >> shift;
          x.atomic_add(&w.bins[bk < (uint64_t)(kBins - 1) ? (int)bk : kBins - 1], 1);
        }
      }
      if (first) x.wave_add(&w.vars[VAR_VALID], valid);
      x.sync();
      x.mark(12);
      // -> [0] bucket b* holding the need-th largest key (-1: below the window), [1] #keys in buckets above b*,
      //    [2] #keys in the window, [3] #keys in b*.  Also re-zeroes bins[] and ends with a barrier.
      x.find_bucket(w.bins, kBins, need, &w.vars[VAR_FB0]);
      const int bstar = w.vars[VAR_FB0], above = w.vars[VAR_FB1], total = w.vars[VAR_FB2], inb = w.vars[VAR_FB3];
      const int N = w.vars[VAR_VALID];
      x.mark(13);
      if (first && N <= K) return;  // nothing to prune (ctc_beam_search_decoder.cpp:150)
      first = false;
      if (bstar < 0) {  // the K-th key lies below the window: look at everything under it
        gbase += total; need -= total; hi = lo; lo = 1;
        continue;
      }
      const uint64_t blo = lo + ((uint64_t)bstar << shift);
      uint64_t bhi = (bstar == kBins - 1) ? hi : blo + ((uint64_t)1 << shift);
      if (bhi > hi) bhi = hi;  // keys at or above hi are already counted in gbase
      if (shift == 0 && bstar < kBins - 1) {  // the bucket is a single key value
        if (tid == 0) {
          w.vars[VAR_TAU_LO] = (int)(uint32_t)blo; w.vars[VAR_TAU_HI] = (int)(uint32_t)(blo >> 32);
          w.vars[VAR_G] = gbase + above; w.vars[VAR_E] = inb;
        }
        x.sync();
        return;
      }
      if (inb <= kListCap) {  // exact rank inside the bucket
        for (int s = tid; s < S; s += nt) {
          const uint64_t k = slot_key48(s);
          if (k >= blo && k < bhi) w.list[x.atomic_add(&w.vars[VAR_LCOUNT], 1)] = k;
        }
        x.sync();
        x.mark(14);
        const int want = need - above;  // rank (1-based, descending) of tau inside the bucket
        for (int q = tid; q < inb; q += nt) {
          const uint64_t mine = w.list[q];
          int g = 0, e = 0;
          for (int r = 0; r < inb; ++r) {
            const uint64_t o = w.list[r];
            g += o > mine;
            e += o == mine;
          }
          if (g < want && want <= g + e) {  // every holder of the K-th key writes the same values
            w.vars[VAR_TAU_LO] = (int)(uint32_t)mine; w.vars[VAR_TAU_HI] = (int)(uint32_t)(mine >> 32);
            w.vars[VAR_G] = gbase + above + g; w.vars[VAR_E] = e;
          }
        }
        x.sync();
        return;
      }
      gbase += above; need -= above; lo = blo; hi = bhi;  // too crowded: histogram the bucket itself
    }
  }

  // ------------------------------------------------------------------------------------------------ exact replay
  // std::nth_element(begin, begin+K, end, prefix_compare) on the DFS-ordered candidate list (w.ek[0, N)).
  CTC_HD void replay_nth_element(int N, int K) {
    const int tid = x.tid(), nt = x.nt();
    uint64_t *v = w.ek;
    auto before = [](uint64_t a, uint64_t c) { return (a >> 16) > (c >> 16); };
    int first = 0, last = N, depth = 2 * stlemu::floor_lg(N);
    uint16_t *Lp = w.lr, *Rp = w.lr + N + 1;
    while (last - first > kSerialCut && depth > 0) {
      --depth;
      if (tid == 0) stlemu::median_to(v, first, first + 1, first + (last - first) / 2, last - 1, before);
      x.sync();
      // Hoare partition of [first+1, last) around v[first]: the t-th element from the left that is not better than
      // the pivot is exchanged with the t-th from the right that is not worse, until the two scans cross.
      const int lo = first + 1, m = last - lo;
      const uint64_t kp = v[first] >> 16;
      for (int i = tid; i <= m; i += nt) {
        uint32_t f = 0;
        if (i < m) {
          const uint64_t k = v[lo + i] >> 16;
          f = (k <= kp ? 1u : 0u) | (k >= kp ? 0x10000u : 0u);
        }
        w.pos[i] = f;
      }
      x.sync();
      const uint32_t tot = x.scan_excl(w.pos, m + 1);
      const int nL = (int)(tot & 0xFFFFu), nR = (int)(tot >> 16);
      for (int i = tid; i < m; i += nt) {
        const uint32_t p0 = w.pos[i], p1 = w.pos[i + 1];
        if ((p1 ^ p0) & 0xFFFFu) Lp[p0 & 0xFFFFu] = (uint16_t)(lo + i);
        if ((p1 ^ p0) >> 16) Rp[nR - 1 - (int)(p0 >> 16)] = (uint16_t)(lo + i);
      }
      if (tid == 0) Rp[nR] = (uint16_t)first;  // the pivot itself stops the right-to-left scan
      x.sync();
      // Iteration t of the serial loop stops its left scan at min(Lp[t], Rp[t-1]) (the element swapped into Rp[t-1]
      // is itself a stop) and ends, returning that position, as soon as it is not left of the right scan's stop.
      const int tmax = nL < nR + 1 ? nL : nR + 1;
      auto crossed = [&](int t) { return t >= nL || t > nR || Lp[t] >= Rp[t]; };
      for (int t = tid; t <= tmax; t += nt) {
        if (!crossed(t)) {
          stlemu::exch(v, (int)Lp[t], (int)Rp[t]);
        } else if (t == 0 || !crossed(t - 1)) {
          int c = t < nL ? (int)Lp[t] : kIntMax;
          if (t > 0 && (int)Rp[t - 1] < c) c = Rp[t - 1];
          w.vars[VAR_CUT] = c;
        }
      }
      x.sync();
      const int cut = w.vars[VAR_CUT];
      if (cut <= K) first = cut; else last = cut;
    }
    if (tid == 0) stlemu::introselect(v, first, K, last, depth, before);
    x.sync();
  }

  // One time step.  w.clp/w.cch (and rank_of in pruned mode) hold this step's candidates; `last` selects the
  // bookkeeping that DecoderState::decode() needs (the permutation std::nth_element leaves behind).
  CTC_HD void step(const StepIn &in, bool last) {
    Beam &b = w.cur;
    Beam &nb = w.nxt;
    const int tid = x.tid(), nt = x.nt();
    const int n = w.vars[VAR_N];
    const int pool_count = w.vars[VAR_POOL];
    const int dmin = w.vars[VAR_DMIN];
    const int K = d.K;
    const int Vc = in.Vc, brank = in.blank_rank;
    const int Vnb = Vc - (brank >= 0 ? 1 : 0);
    const int S = n * (2 + Vnb);
    const float lp_blank = brank >= 0 ? w.clp[brank] : CTC_NEG_MAX;
    const bool small_vocab = Vnb <= 64;  // existing children fit a 64-bit mask per parent

    // ---- A1: subtree ends and nearest in-beam ancestors from the LCP array
    for (int j = tid; j < n; j += nt) {
      const int dj = b.dep[j];
      int q = j + 1;
      while (q < n && b.lcp[q] >= dj) ++q;
      w.e[j] = q;
      int a = -1, m = kIntMax;
      for (int i = j - 1; i >= 0; --i) {
        const int l = b.lcp[i + 1];
        m = l < m ? l : m;
        if (m < dmin) break;
        if (b.dep[i] <= m) { a = i; break; }
      }
      w.anc[j] = a;
      w.hit[2 * j] = 0;
      w.hit[2 * j + 1] = 0;
    }
    for (int i = tid; i < kBins; i += nt) w.bins[i] = 0;
    if (tid == 0) { w.vars[VAR_VALID] = 0; w.vars[VAR_LCOUNT] = 0; }
    x.sync();
    x.mark(0);

    // ---- A2: slot offsets; which children of in-beam parents already exist (in the beam, or dead-interior)
    for (int j = tid; j < n; j += nt) {
      int a = 0;
      for (int i = w.anc[j]; i >= 0; i = w.anc[i]) ++a;
      const int ej = w.e[j];
      w.ostart[j] = 2 * j + Vnb * (j - a);
      w.cstart[j] = 2 * ej + Vnb * (ej - 1 - a);
      const int P = w.anc[j];
      int hv = 0, pr = -1, rr = -1;
      if (P >= 0) {
        if (b.dep[P] == b.dep[j] - 1) {                      // parent in the beam: "hit" (path_trie.cpp:40-48)
          pr = rank_of_char(in, b.ch[j]);
        } else {
          // dead-interior child X of the nearest in-beam ancestor on the way down to j (alive because j is below it)
          hv = 1;
          if (b.viaanc[j] != b.node[P]) {
            int hops = b.dep[j] - b.dep[P] - 1, xn = b.node[j];
            for (int h = 0; h < hops; ++h) xn = pool[xn].parent;
            b.via[j] = xn;
            b.viaanc[j] = b.node[P];
            b.viach[j] = pool[xn].ch;
          }
          // j is the first beam entry below X iff its predecessor is outside X's subtree: then j revives X
          if (b.lcp[j] <= b.dep[P]) rr = rank_of_char(in, b.viach[j]);
        }
        const int r = pr >= 0 ? pr : rr;
        if (r >= 0 && small_vocab) {
          const int bit = r - ((brank >= 0 && r > brank) ? 1 : 0);
          x.atomic_or(&w.hit[2 * P + (bit >> 5)], 1u << (bit & 31));
        }
      }
      w.hasvia[j] = hv;
      w.pinr[j] = pr;
      w.revr[j] = rr;
    }
    x.sync();
    x.mark(1);

    // ---- B: score every candidate and lay it out in DFS (Euler-tour) slot order.
    // B1 (beam entries themselves + revived children) and B2 (brand-new children) are independent: with enough
    // waves they run side by side on disjoint threads.
    const int n1 = (n + 63) & ~63;
    const bool split = nt - n1 >= 128;
    if (!split || tid < n1) {
      for (int j = tid; j < n; j += (split ? n1 : nt)) {
        const int c = b.ch[j];
        const int r = rank_of_char(in, c);
        const float sc = b.score[j], nbp = b.nbprev[j];
        float bcur = brank >= 0 ? lp_blank + sc : CTC_NEG_MAX;             // :97-101
        float nbcur = CTC_NEG_MAX;
        if (r >= 0) nbcur = lse(nbcur, w.clp[r] + nbp);                     // :103-106
        const int P = w.anc[j];
        const int pr = w.pinr[j];
        if (pr >= 0) {
          const float lp = w.clp[pr];
          if (b.lpc[j] < lp) {                                             // path_trie.cpp:42-45
            b.lpc[j] = lp;
            pool[b.node[j]].tstep = in.t;
            pool[b.node[j]].lpc = lp;
          }
          nbcur = lse(nbcur, child_logp(P, c, lp));                         // :138-139
        }
        w.b_new[j] = bcur;
        w.nb_new[j] = nbcur;
        const float ns = lse(bcur, nbcur);                                  // path_trie.cpp:131-136
        w.sc_new[j] = ns;
        const int s0 = w.ostart[j];
        uint32_t k0 = 0, i0 = kHoleInfo;
        int l0 = kIntMax, l1 = b.lcp[j];
        const int rx = w.revr[j];
        if (rx >= 0) {                                                      // path_trie.cpp:40-57: hit + revive
          const int cx = b.viach[j];
          const float lp = w.clp[rx];
          const int xn = b.via[j];
          if (pool[xn].lpc < lp) {
            pool[xn].tstep = in.t;
            pool[xn].lpc = lp;
          }
          k0 = ord_f32(child_logp(P, cx, lp));
          i0 = mk_info(cx, T_REVIVED, j);
          l0 = b.lcp[j];
          l1 = b.dep[P] + 1;
        }
        w.skey[s0] = k0; w.sinfo[s0] = i0; w.slcp[s0] = l0;
        w.skey[s0 + 1] = ord_f32(ns); w.sinfo[s0 + 1] = mk_info(c, T_SELF, j); w.slcp[s0 + 1] = l1;
      }
    }
    if (!split || tid >= n1) {
      const int t2 = split ? tid - n1 : tid, nt2 = split ? nt - n1 : nt;
      int lp2 = 1;
      while (lp2 < Vnb) lp2 <<= 1;                   // lanes per parent (power of two >= Vnb)
      if (small_vocab && nt2 >= lp2) {               // a group of lp2 lanes per parent, one lane per character
        const int rn = t2 & (lp2 - 1);
        const int ng = nt2 / lp2;
        const int sh = ceil_log2_u64((uint64_t)lp2);
        if (rn < Vnb && (t2 >> sh) < ng) {
          const int r = rn + ((brank >= 0 && rn >= brank) ? 1 : 0);
          const int c = in.identity ? r : w.cch[r];
          const float lp = w.clp[r];
          for (int i = t2 >> sh; i < n; i += ng) {
            const int s = w.cstart[i] + rn;
            const bool exists = (w.hit[2 * i + (rn >> 5)] >> (rn & 31)) & 1u;
            w.skey[s] = exists ? 0u : ord_f32(child_logp(i, c, lp));
            w.sinfo[s] = exists ? kHoleInfo : mk_info(c, T_CHILD, i);
            w.slcp[s] = exists ? kIntMax : b.dep[i];
          }
        }
      } else {
        for (int idx = t2; idx < n * Vnb; idx += nt2) {
          const int i = idx / Vnb, rn = idx - i * Vnb;
          const int r = rn + ((brank >= 0 && rn >= brank) ? 1 : 0);
          const int c = in.identity ? r : w.cch[r];
          const int s = w.cstart[i] + rn;
          const bool exists = small_vocab && ((w.hit[2 * i + (rn >> 5)] >> (rn & 31)) & 1u);
          w.skey[s] = exists ? 0u : ord_f32(child_logp(i, c, w.clp[r]));
          w.sinfo[s] = exists ? kHoleInfo : mk_info(c, T_CHILD, i);
          w.slcp[s] = exists ? kIntMax : b.dep[i];
        }
      }
    }
    x.sync();
    if (!small_vocab) {  // children that already exist leave a hole in their parent's group
      for (int j = tid; j < n; j += nt) {
        const int r = w.pinr[j] >= 0 ? w.pinr[j] : w.revr[j];
        if (r >= 0) {
          const int s = w.cstart[w.anc[j]] + r - ((brank >= 0 && r > brank) ? 1 : 0);
          w.skey[s] = 0; w.sinfo[s] = kHoleInfo; w.slcp[s] = kIntMax;
        }
      }
      x.sync();
    }
    x.mark(2);

    // ---- C: the K-th best key
    select_kth(S, K);
    const int N = w.vars[VAR_VALID];
    uint64_t tau = 0;
    bool exact = false, tie = false;
    if (N > K) {
      tau = ((uint64_t)(uint32_t)w.vars[VAR_TAU_HI] << 32) | (uint32_t)w.vars[VAR_TAU_LO];
      tie = w.vars[VAR_E] > K - w.vars[VAR_G];  // the boundary splits a group of equivalent prefixes
      exact = tie || last;                      // decode() sorts the array exactly as nth_element left it (:164-190)
    }
    x.mark(5);

    // ---- D: exact replay of std::nth_element when the outcome depends on it
    if (exact) {
      for (int s = tid; s <= S; s += nt) w.pos[s] = (s < S && info_type(w.sinfo[s]) != T_HOLE) ? 1u : 0u;
      x.sync();
      x.scan_excl(w.pos, S + 1);
      for (int s = tid; s < S; s += nt)
        if (w.pos[s + 1] != w.pos[s]) w.ek[w.pos[s]] = (slot_key48(s) << 16) | (uint64_t)s;
      x.sync();
      replay_nth_element(N, K);
      for (int s = tid; s < S; s += nt) w.pos[s] = 0;
      x.sync();
      for (int k = tid; k < K; k += nt) w.pos[(int)(w.ek[k] & 0xFFFFu)] = 1u;
      x.sync();
      x.mark(6);
    }

    // ---- E: fused compaction.  Each thread owns a contiguous chunk of slots; one workgroup scan yields the new beam
    // index of every survivor, the id of every surviving new node, and the LCP carried across chunk boundaries.
    const int chunk = ((S + nt - 1) / nt) | 1;  // odd stride: conflict-free LDS access across lanes
    const int c_lo = tid * chunk < S ? tid * chunk : S, c_hi = c_lo + chunk < S ? c_lo + chunk : S;
    auto survives = [&](int s, uint32_t inf) -> bool {
      if (exact) return w.pos[s] != 0;
      if (info_type(inf) == T_HOLE) return false;
      if (N <= K) return true;
      const uint64_t k = key48(w.skey[s], inf);
      return k > tau || (k == tau && !tie);
    };
    int cnt = 0, cntc = 0, tailmin = kIntMax, dloc = kIntMax;
    uint32_t kmax = 0;
    bool has = false;
    for (int s = c_lo; s < c_hi; ++s) {
      const uint32_t inf = w.sinfo[s];
      const int l = w.slcp[s];
      tailmin = l < tailmin ? l : tailmin;
      if (survives(s, inf)) {
        ++cnt;
        cntc += info_type(inf) == T_CHILD;
        has = true;
        tailmin = kIntMax;
        const int j = info_entry(inf);
        const int dd = info_type(inf) == T_SELF ? b.dep[j] : (info_type(inf) == T_CHILD ? b.dep[j] + 1 : b.dep[w.anc[j]] + 1);
        dloc = dd < dloc ? dd : dloc;
        const uint32_t sk = w.skey[s];
        kmax = sk > kmax ? sk : kmax;
      }
    }
    x.mark(3);
    SegOut so;
    x.seg_scan(cnt, cntc, has, tailmin, dloc, kmax, so);
    x.mark(4);
    const int n_new = so.total, n_child = so.totalc;
    if (pool_count + n_child > pool_cap) {  // cannot happen when the pool is sized 1 + K*T
      if (tid == 0) w.vars[VAR_STATUS] = ST_POOL_OVERFLOW;
      x.sync();
      return;
    }
    {
      int k = so.excl, cid = so.exclc, run = so.carry;
      for (int s = c_lo; s < c_hi; ++s) {
        const uint32_t inf = w.sinfo[s];
        const int l = w.slcp[s];
        run = l < run ? l : run;
        if (!survives(s, inf)) continue;
        const uint32_t type = info_type(inf);
        const int j = info_entry(inf);
        nb.lcp[k] = k == 0 ? -1 : run;
        run = kIntMax;
        if (last && exact) w.pos[s] = 0x80000000u | (uint32_t)k;
        if (type == T_SELF) {
          nb.node[k] = b.node[j]; nb.par[k] = b.par[j]; nb.ch[k] = b.ch[j]; nb.dep[k] = b.dep[j];
          nb.via[k] = b.via[j]; nb.viaanc[k] = b.viaanc[j]; nb.viach[k] = b.viach[j];
          nb.bprev[k] = w.b_new[j]; nb.nbprev[k] = w.nb_new[j]; nb.score[k] = w.sc_new[j]; nb.lpc[k] = b.lpc[j];
        } else {
          const int c = info_ch(inf);
          const int P = (type == T_CHILD) ? j : w.anc[j];
          const float lp = w.clp[rank_of_char(in, c)];
          const float logp = child_logp(P, c, lp);
          int id;
          float lpc;
          if (type == T_CHILD) {  // path_trie.cpp:97-105
            id = pool_count + cid;
            ++cid;
            PoolNode pn; pn.parent = b.node[P]; pn.ch = c; pn.tstep = in.t; pn.lpc = lp;
            pool[id] = pn;
            lpc = lp;
          } else {                // path_trie.cpp:50-56 : revived, probabilities reset
            id = b.via[j];
            lpc = pool[id].lpc;
          }
          nb.node[k] = id; nb.par[k] = b.node[P]; nb.ch[k] = c; nb.dep[k] = b.dep[P] + 1;
          nb.via[k] = -1; nb.viaanc[k] = -1; nb.viach[k] = -1;
          nb.bprev[k] = CTC_NEG_MAX; nb.nbprev[k] = logp; nb.score[k] = logp; nb.lpc[k] = lpc;
        }
        ++k;
      }
    }
    if (tid == 0) {
      w.vars[VAR_N] = n_new;
      w.vars[VAR_POOL] = pool_count + n_child;
      w.vars[VAR_DMIN] = so.dmin;
      // next select window.  It is anchored at this step's best key (an upper bound for the next step's keys when
      // log-probabilities are <= 0) and must reach down to the next K-th key: twice the distance from THIS step's
      // anchor (the previous best key) to this step's K-th key, rounded up to a power of two.
      int wl = 32;
      if (N > K) {
        const uint32_t anchor = (uint32_t)w.vars[VAR_MAXKEY];
        const uint32_t t32 = (uint32_t)(tau >> 16);
        const uint32_t gap = anchor > t32 ? anchor - t32 : 0;
        wl = ceil_log2_u64((uint64_t)gap + 1) + 1;
        wl = wl < 10 ? 10 : (wl > 32 ? 32 : wl);
      }
      w.vars[VAR_WLOG] = wl;
      w.vars[VAR_MAXKEY] = (int)so.maxkey;
    }
    // un-register this step's candidates from the rank table
    if (!in.identity)
      for (int r = tid; r < Vc; r += nt) w.rank_of[w.cch[r]] = -1;
    x.sync();
    if (last) {
      if (exact)
        for (int k = tid; k < K; k += nt) w.fin[k] = (int)(w.pos[(int)(w.ek[k] & 0xFFFFu)] & 0x7FFFFFFFu);
      else
        for (int k = tid; k < n_new; k += nt) w.fin[k] = k;
      x.sync();
    }
    x.mark(8);
    Beam t = w.cur; w.cur = w.nxt; w.nxt = t;
  }

  // DecoderState::decode() + get_beam_search_result + binding.cpp:85-99 for one utterance.
  // `had_steps`: false when the utterance has no frames (fin is then just the root).
  CTC_HD void finish(bool had_steps, int T_stride, int32_t *out_tok, int32_t *out_ts, float *out_score, int32_t *out_len,
                     int32_t *n_results) {
    const Beam &b = w.cur;
    const int tid = x.tid(), nt = x.nt();
    const int n = w.vars[VAR_N];
    const int nres = n < d.K ? n : d.K;
    if (!had_steps)
      for (int k = tid; k < nres; k += nt) w.fin[k] = k;
    x.sync();
    if (tid == 0) {
      const float *sc = b.score;
      const int *ch = b.ch;
      auto before = [sc, ch](int a, int c) {
        return key48(ord_f32(sc[a]), mk_info(ch[a], 0, 0)) > key48(ord_f32(sc[c]), mk_info(ch[c], 0, 0));
      };
      stlemu::sort(w.fin, 0, nres, before, w.sstack);  // ctc_beam_search_decoder.cpp:188-190
      stlemu::sort(w.fin, 0, nres, before, w.sstack);  // decoder_utils.cpp:59
      if (n_results) *n_results = nres;
    }
    x.sync();
    for (int p = tid; p < nres; p += nt) {
      const int j = w.fin[p];
      out_score[p] = -b.score[j];           // decoder_utils.cpp:68 (approx_ctc = score without a scorer)
      int dd = b.dep[j], xn = b.node[j];
      out_len[p] = dd;
      int32_t *tk = out_tok + (size_t)p * T_stride, *ts = out_ts + (size_t)p * T_stride;
      while (dd > 0) {                      // path_trie.cpp:113-126
        const PoolNode pn = pool[xn];
        tk[dd - 1] = pn.ch;
        ts[dd - 1] = pn.tstep;
        xn = pn.parent;
        --dd;
      }
    }
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
template <class X>
CTC_HD int decode_utterance(X &x, Work &w, const Dims &d, int blank, const float *rows, const PrunedRows *pr, int len,
                            PoolNode *pool, int pool_cap, const uint64_t *tbl, int T_stride, int32_t *out_tok,
                            int32_t *out_ts, float *out_score, int32_t *out_len, int32_t *n_results) {
  Decoder<X> dec(x, w, d, blank, pool, pool_cap, tbl);
  dec.init();
  const int tid = x.tid(), nt = x.nt();
  // Row prefetch: the row of step t+1 is requested from HBM before step t runs, so its latency hides behind the step.
  const bool prefetch = pr == nullptr && d.V <= nt;
  float pre = 0.f;
  if (prefetch && len > 0 && tid < d.V) pre = rows[tid];
  for (int t = 0; t < len; ++t) {
    StepIn in;
    in.t = t;
    if (pr == nullptr) {
      in.Vc = d.V;
      in.identity = 1;
      in.blank_rank = blank;
      if (prefetch) {
        if (tid < d.V) w.clp[tid] = pre;
        if (t + 1 < len && tid < d.V) pre = rows[(size_t)(t + 1) * d.V + tid];
      } else {
        for (int r = tid; r < d.V; r += nt) w.clp[r] = rows[(size_t)t * d.V + r];
      }
      x.sync();
    } else {
      in.Vc = pr->cnt[t];
      in.identity = 0;
      for (int r = tid; r < in.Vc; r += nt) {
        const int c = pr->ch[(size_t)t * pr->stride + r];
        w.cch[r] = c;
        w.clp[r] = pr->lp[(size_t)t * pr->stride + r];
        w.rank_of[c] = r;
      }
      x.sync();
      in.blank_rank = w.rank_of[blank];
    }
    x.mark(10);
    dec.step(in, t == len - 1);
    if (w.vars[VAR_STATUS] != ST_OK) return w.vars[VAR_STATUS];
  }
  dec.finish(len > 0, T_stride, out_tok, out_ts, out_score, out_len, n_results);
  x.sync();
  x.mark(11);
  return ST_OK;
}

}  // namespace ctcbeam
