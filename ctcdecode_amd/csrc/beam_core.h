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
//     (decoder_utils.cpp:122-132).  If the K boundary cuts through a group of EQUAL keys -- structural at long T,
//     SURVEY.md 7.3-H2 -- or on the last step (whose permutation feeds the final sorts), one lane replays
//     libstdc++'s std::nth_element (stl_emul.h) on the DFS-ordered candidate list, so the same prefixes survive
//     as in the reference.  Survivors are compacted in slot order, which keeps the DFS order invariant.
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

enum { VAR_N = 0, VAR_POOL = 1, VAR_DMIN = 2, VAR_STATUS = 3, VAR_NEXT_DMIN = 4, VAR_COUNT = 8 };

struct Work {
  Beam cur, nxt;
  int *e, *anc, *ostart, *cstart, *hasvia;  // per beam entry, this step
  float *b_new, *nb_new, *sc_new;
  int *cch;        // candidate characters of this step (unused in identity mode)
  float *clp;      // their log-probs
  int *rank_of;    // V entries, -1 = not a candidate (only when Dims::use_rank_table)
  uint32_t *skey, *sinfo, *pos, *perm;  // S_max (+1 for pos)
  int *slot_of, *fin, *sstack;
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
  w.cstart = carve_ptr<int>(p, K); w.hasvia = carve_ptr<int>(p, K);
  w.b_new = carve_ptr<float>(p, K); w.nb_new = carve_ptr<float>(p, K); w.sc_new = carve_ptr<float>(p, K);
  w.cch = carve_ptr<int>(p, (size_t)d.Vc_max);
  w.clp = carve_ptr<float>(p, (size_t)d.Vc_max);
  w.rank_of = carve_ptr<int>(p, d.use_rank_table ? (size_t)d.V : 0);
  w.skey = carve_ptr<uint32_t>(p, S); w.sinfo = carve_ptr<uint32_t>(p, S);
  w.pos = carve_ptr<uint32_t>(p, S + 1); w.perm = carve_ptr<uint32_t>(p, S);
  w.slot_of = carve_ptr<int>(p, K); w.fin = carve_ptr<int>(p, K);
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
      w.vars[VAR_NEXT_DMIN] = kIntMax;
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

    // ---- A: subtree ends and nearest in-beam ancestors from the LCP array
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
    }
    x.sync();
    for (int j = tid; j < n; j += nt) {
      int a = 0;
      for (int i = w.anc[j]; i >= 0; i = w.anc[i]) ++a;
      const int ej = w.e[j];
      w.ostart[j] = 2 * j + Vnb * (j - a);
      w.cstart[j] = 2 * ej + Vnb * (ej - 1 - a);
      // dead-interior child of the nearest in-beam ancestor on the way down to j (alive because j is below it)
      const int P = w.anc[j];
      int hv = 0;
      if (P >= 0 && b.dep[P] < b.dep[j] - 1) {
        hv = 1;
        if (b.viaanc[j] != b.node[P]) {
          int hops = b.dep[j] - b.dep[P] - 1, xn = b.node[j];
          for (int h = 0; h < hops; ++h) xn = pool[xn].parent;
          b.via[j] = xn;
          b.viaanc[j] = b.node[P];
          b.viach[j] = pool[xn].ch;
        }
      }
      w.hasvia[j] = hv;
    }
    if (tid == 0) w.vars[VAR_NEXT_DMIN] = kIntMax;
    x.sync();

    // ---- B1: beam entries themselves (blank / repeat / parent-extension mass), revived dead-interior children
    for (int j = tid; j < n; j += nt) {
      const int c = b.ch[j];
      const int r = rank_of_char(in, c);
      const float sc = b.score[j], nbp = b.nbprev[j];
      float bcur = brank >= 0 ? lp_blank + sc : CTC_NEG_MAX;               // :97-101
      float nbcur = CTC_NEG_MAX;
      if (r >= 0) nbcur = lse(nbcur, w.clp[r] + nbp);                       // :103-106
      const int P = w.anc[j];
      const bool pin = P >= 0 && b.dep[P] == b.dep[j] - 1;                  // parent is in the beam
      if (pin && r >= 0) {
        const float lp = w.clp[r];
        if (b.lpc[j] < lp) {                                               // path_trie.cpp:42-45
          b.lpc[j] = lp;
          pool[b.node[j]].tstep = in.t;
          pool[b.node[j]].lpc = lp;
        }
        nbcur = lse(nbcur, child_logp(P, c, lp));                           // :138-139
      }
      w.b_new[j] = bcur;
      w.nb_new[j] = nbcur;
      const float ns = lse(bcur, nbcur);                                    // path_trie.cpp:131-136
      w.sc_new[j] = ns;
      const int s0 = w.ostart[j];
      w.skey[s0 + 1] = ord_f32(ns);
      w.sinfo[s0 + 1] = mk_info(c, T_SELF, j);
      uint32_t k0 = 0, i0 = kHoleInfo;
      // j is the first beam entry below that dead child X iff its predecessor is outside X's subtree
      if (w.hasvia[j] && b.lcp[j] <= b.dep[P]) {
        const int cx = b.viach[j];
        const int rx = rank_of_char(in, cx);
        if (rx >= 0) {                                                      // path_trie.cpp:40-57: hit + revive
          const float lp = w.clp[rx];
          const int xn = b.via[j];
          if (pool[xn].lpc < lp) {
            pool[xn].tstep = in.t;
            pool[xn].lpc = lp;
          }
          k0 = ord_f32(child_logp(P, cx, lp));
          i0 = mk_info(cx, T_REVIVED, j);
        }
      }
      w.skey[s0] = k0;
      w.sinfo[s0] = i0;
    }
    // ---- B2: brand-new children of every entry (never materialised unless they survive)
    for (int idx = tid; idx < n * Vnb; idx += nt) {
      const int i = idx / Vnb, rn = idx - i * Vnb;
      const int r = rn + ((brank >= 0 && rn >= brank) ? 1 : 0);
      const int c = in.identity ? r : w.cch[r];
      const int s = w.cstart[i] + rn;
      w.skey[s] = ord_f32(child_logp(i, c, w.clp[r]));
      w.sinfo[s] = mk_info(c, T_CHILD, i);
    }
    x.sync();
    // ---- B3: children that already exist leave a hole in their parent's group
    for (int j = tid; j < n; j += nt) {
      const int P = w.anc[j];
      if (P < 0) continue;
      int r = -1;
      if (b.dep[P] == b.dep[j] - 1) {
        r = rank_of_char(in, b.ch[j]);
      } else if (info_type(w.sinfo[w.ostart[j]]) == T_REVIVED) {
        r = rank_of_char(in, b.viach[j]);
      }
      if (r >= 0) {
        const int s = w.cstart[P] + r - ((brank >= 0 && r > brank) ? 1 : 0);
        w.skey[s] = 0;
        w.sinfo[s] = kHoleInfo;
      }
    }
    x.sync();

    // ---- C: how many candidates, and the K-th best key
    int local = 0;
    for (int s = tid; s < S; s += nt) local += info_type(w.sinfo[s]) != T_HOLE;
    const int N = x.reduce_add(local);
    bool exact = false;
    uint32_t tau_s = 0, tau_c = 0;
    if (N > K) {
      for (int bit = 31; bit >= 0; --bit) {
        const uint32_t trial = tau_s | (1u << bit);
        local = 0;
        for (int s = tid; s < S; s += nt) local += w.skey[s] >= trial;
        if (x.reduce_add(local) >= K) tau_s = trial;
      }
      int lg = 0, le = 0;
      for (int s = tid; s < S; s += nt) {
        const uint32_t k = w.skey[s];
        lg += k > tau_s;
        le += k == tau_s;
      }
      const int G = x.reduce_add(lg), E = x.reduce_add(le);
      const int m = K - G;
      if (E > m) {  // several candidates share the boundary score: order them by character (prefix_compare)
        for (int bit = 15; bit >= 0; --bit) {
          const uint32_t trial = tau_c | (1u << bit);
          local = 0;
          for (int s = tid; s < S; s += nt) local += (w.skey[s] == tau_s) && ((w.sinfo[s] >> 16) >= trial);
          if (x.reduce_add(local) >= m) tau_c = trial;
        }
        local = 0;
        for (int s = tid; s < S; s += nt) local += (w.skey[s] == tau_s) && ((w.sinfo[s] >> 16) >= tau_c);
        if (x.reduce_add(local) > m) exact = true;  // the boundary splits a group of equivalent prefixes
      }
      if (last) exact = true;  // decode() sorts the array exactly as nth_element left it (:164-190)
    }

    // ---- D: survivors
    if (exact) {
      for (int s = tid; s <= S; s += nt) w.pos[s] = (s < S && info_type(w.sinfo[s]) != T_HOLE) ? 1u : 0u;
      x.sync();
      x.scan_excl(w.pos, S + 1);
      for (int s = tid; s < S; s += nt)
        if (w.pos[s + 1] != w.pos[s]) w.perm[w.pos[s]] = (uint32_t)s;
      x.sync();
      if (tid == 0) {
        const uint32_t *sk = w.skey, *si = w.sinfo;
        stlemu::nth_element(w.perm, 0, K, N,
                            [sk, si](uint32_t a, uint32_t c) { return key48(sk[a], si[a]) > key48(sk[c], si[c]); });
      }
      x.sync();
      for (int s = tid; s <= S; s += nt) w.pos[s] = 0;
      x.sync();
      for (int k = tid; k < K; k += nt) {
        const uint32_t s = w.perm[k];
        w.pos[s] = 1u | (info_type(w.sinfo[s]) == T_CHILD ? 0x10000u : 0u);
      }
      x.sync();
    } else {
      for (int s = tid; s <= S; s += nt) {
        uint32_t f = 0;
        if (s < S) {
          const uint32_t k = w.skey[s], inf = w.sinfo[s];
          const bool keep = (N <= K) ? (info_type(inf) != T_HOLE) : (k > tau_s || (k == tau_s && (inf >> 16) >= tau_c));
          if (keep) f = 1u | (info_type(inf) == T_CHILD ? 0x10000u : 0u);
        }
        w.pos[s] = f;
      }
      x.sync();
    }
    const uint32_t total = x.scan_excl(w.pos, S + 1);
    const int n_new = (int)(total & 0xFFFFu), n_child = (int)(total >> 16);
    if (pool_count + n_child > pool_cap) {  // cannot happen when the pool is sized 1 + K*T
      if (tid == 0) w.vars[VAR_STATUS] = ST_POOL_OVERFLOW;
      x.sync();
      return;
    }

    // ---- E: compact the survivors (slot order = DFS order) into the next beam; append surviving new nodes
    for (int s = tid; s < S; s += nt) {
      const uint32_t p0 = w.pos[s];
      if (((w.pos[s + 1] ^ p0) & 0xFFFFu) == 0) continue;
      const int k = (int)(p0 & 0xFFFFu);
      const uint32_t inf = w.sinfo[s];
      const uint32_t type = info_type(inf);
      const int j = info_entry(inf);
      w.slot_of[k] = s;
      if (type == T_SELF) {
        nb.node[k] = b.node[j]; nb.par[k] = b.par[j]; nb.ch[k] = b.ch[j]; nb.dep[k] = b.dep[j];
        nb.via[k] = b.via[j]; nb.viaanc[k] = b.viaanc[j]; nb.viach[k] = b.viach[j];
        nb.bprev[k] = w.b_new[j]; nb.nbprev[k] = w.nb_new[j]; nb.score[k] = w.sc_new[j]; nb.lpc[k] = b.lpc[j];
        x.atomic_min(&w.vars[VAR_NEXT_DMIN], b.dep[j]);
      } else {
        const int c = info_ch(inf);
        const int P = (type == T_CHILD) ? j : w.anc[j];
        const float lp = w.clp[rank_of_char(in, c)];
        const float logp = child_logp(P, c, lp);
        int id;
        float lpc;
        if (type == T_CHILD) {  // path_trie.cpp:97-105
          id = pool_count + (int)(p0 >> 16);
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
        x.atomic_min(&w.vars[VAR_NEXT_DMIN], b.dep[P] + 1);
      }
    }
    if (last && exact)
      for (int k = tid; k < K; k += nt) w.fin[k] = (int)(w.pos[w.perm[k]] & 0xFFFFu);
    else if (last)
      for (int k = tid; k < n_new; k += nt) w.fin[k] = k;
    x.sync();

    // ---- F: LCP of consecutive survivors = min over the slots between them
    for (int k = tid; k < n_new; k += nt) {
      int m = kIntMax;
      if (k == 0) {
        m = -1;
      } else {
        for (int s = w.slot_of[k - 1] + 1; s <= w.slot_of[k]; ++s) {
          const uint32_t inf = w.sinfo[s];
          const uint32_t type = info_type(inf);
          if (type == T_HOLE) continue;
          const int j = info_entry(inf);
          int l;
          if (type == T_CHILD) l = b.dep[j];
          else if (type == T_REVIVED) l = b.lcp[j];
          else l = (info_type(w.sinfo[s - 1]) == T_REVIVED) ? b.dep[w.anc[j]] + 1 : b.lcp[j];
          m = l < m ? l : m;
        }
      }
      nb.lcp[k] = m;
    }
    if (tid == 0) {
      w.vars[VAR_N] = n_new;
      w.vars[VAR_POOL] = pool_count + n_child;
      w.vars[VAR_DMIN] = w.vars[VAR_NEXT_DMIN];
    }
    // un-register this step's candidates from the rank table
    if (!in.identity)
      for (int r = tid; r < Vc; r += nt) w.rank_of[w.cch[r]] = -1;
    x.sync();
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
    dec.step(in, t == len - 1);
    if (w.vars[VAR_STATUS] != ST_OK) return w.vars[VAR_STATUS];
  }
  dec.finish(len > 0, T_stride, out_tok, out_ts, out_score, out_len, n_results);
  return ST_OK;
}

}  // namespace ctcbeam
