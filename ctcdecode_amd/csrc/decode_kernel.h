// decode_kernel.h -- the workgroup execution policy (barriers, wave-level scans and compactions in DPP) and the decode
// kernel template of the CTC prefix beam search (one workgroup per utterance, beam_core.h inside).  Shared by
// ctcdecode_amd.hip (host side: picks an instantiation, launches it) and decode_kernels.hip (the instantiations, compiled
// as several translation units in parallel: ctcdecode_amd/_build.py).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "beam_core.h"

namespace ctcdk {

using namespace ctcbeam;

// ------------------------------------------------------------------------------------------------ device policy
// Cross-lane data movement uses DPP (row shifts + row broadcasts, a few cycles each) instead of ds_bpermute-based
// shuffles (~100 cycles each on the critical path): every block primitive below is a wave-level scan.
// lane mask of a predicate.  (HIP's __ballot(int) compares an integer with zero: the compiler then materialises the
// predicate as 0 / 1 in a vector register and compares it again -- two vector instructions per ballot; the builtin takes
// the boolean and folds into the compare that produced it.)
#define CTC_BALLOT(p) __builtin_amdgcn_ballot_w64(p)
#define CTC_DPP(old, v, ctrl, rowmask) __builtin_amdgcn_update_dpp((int)(old), (int)(v), (ctrl), (rowmask), 0xf, false)

// Inclusive scan over the 64 lanes of a wave; op(left, mine); `ident` is op's left identity.
template <class Op>
__device__ __forceinline__ int wave_scan(int v, const int ident, Op op) {
  v = op(CTC_DPP(ident, v, 0x111, 0xf), v);  // row_shr:1
  v = op(CTC_DPP(ident, v, 0x112, 0xf), v);  // row_shr:2
  v = op(CTC_DPP(ident, v, 0x114, 0xf), v);  // row_shr:4
  v = op(CTC_DPP(ident, v, 0x118, 0xf), v);  // row_shr:8
  v = op(CTC_DPP(ident, v, 0x142, 0xa), v);  // row_bcast:15 -> rows 1, 3
  v = op(CTC_DPP(ident, v, 0x143, 0xc), v);  // row_bcast:31 -> rows 2, 3
  return v;
}
__device__ __forceinline__ int wave_sum(int v) {
  return __builtin_amdgcn_readlane(wave_scan(v, 0, [](int a, int b) { return a + b; }), 63);
}
__device__ __forceinline__ int wave_min(int v) {
  return __builtin_amdgcn_readlane(wave_scan(v, ctcbeam::kIntMax, [](int a, int b) { return a < b ? a : b; }), 63);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readlane(
      wave_scan((int)v, 0, [](int a, int b) { return (uint32_t)a > (uint32_t)b ? a : b; }), 63);
}

// FAR: part of the workspace lives in HBM (BIG layout): every barrier must then also drain global memory traffic.
// PROF: 3 = product build with the four-entries-per-wave subtree search of phase A1 (kQuarters: chain-shaped beams);
// PROF: 0 = product build; 1 = per-phase timers (mark) and the beam dump; 2 = barrier timeline only (no timers: the
// timers' mutable state would put this object into scratch memory and distort the timeline).
// NT: the workgroup size when it is known at compile time (0 = read blockDim): wave counts, the role split of a frame
// and the slot-to-wave assignment then fold to constants.
template <int PROF, bool FAR, int NT = 0>
struct DevX {
  static constexpr bool kZeroKeyTail = true;  // beam_core.h Decoder::kTailZero
  int *red;  // 2 x 16 ints of LDS
  int parity;
  long long *prof;   // PROF: per-phase cycle accumulators (LDS), written by thread 0
  long long last;
  __device__ __forceinline__ void mark(int id) {
    if (PROF == 1 && threadIdx.x == 0) {
      const long long now = (long long)wall_clock64();
      prof[id] += now - last;
      last = now;
    }
  }
  // debugging aid (profiling build only): the beam after every frame -> dbg[t][0] = n, then (node, dep, lcp, score bits) per entry
  int *dbg; int dbg_stride;
  __device__ __forceinline__ void dump(int t, int n, const int *node, const int *dep, const int *lcp, const float *score) {
    if (PROF == 1 && dbg) {
      int *o = dbg + (size_t)t * dbg_stride;
      if (threadIdx.x == 0) o[0] = n;
      for (int i = threadIdx.x; i < n; i += nt()) {
        o[1 + 4 * i] = node[i]; o[2 + 4 * i] = dep[i]; o[3 + 4 * i] = lcp[i]; o[4 + 4 * i] = __float_as_int(score[i]);
      }
    }
  }
  __device__ __forceinline__ int tid() const { return (int)threadIdx.x; }
  // the thread id as a value the optimiser must treat as new at this point: predicates on it ("tid < 128", "lane 0 of the last
  // wave") are then recomputed where they are used -- one compare -- instead of being hoisted out of the frame loop into pairs of
  // scalar registers that live across it, spill, and come back as two v_readlane each (round 6: beam_core.h step())
  __device__ __forceinline__ int tid_fresh() const {
    int t = (int)threadIdx.x;
#if defined(CTC_EXP_FRESH_TID)
    asm volatile("" : "+v"(t));
#endif
    return t;
  }
  __device__ __forceinline__ int nt() const { return NT ? NT : (int)blockDim.x; }
  __device__ __forceinline__ constexpr bool nt_is(int v) const { return NT == v; }  // the workgroup size is this compile-time value
  __device__ __forceinline__ constexpr bool far() const { return FAR; }  // part of the workspace lives in HBM
  // LDS-only barrier: waits for this wave's LDS traffic, not for outstanding global loads/stores (the row prefetch
  // and the pool appends stay in flight across phases).  sync_full() is the fence that also drains global memory.
  __device__ __forceinline__ void sync() {
    if (PROF == 2 && tl) tl_rec();
    if (FAR) __syncthreads();
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (PROF == 2 && tl) tl_rec();
  }
  // a wave's own LDS writes, made visible to its other lanes' later reads (no barrier: one wave working alone)
  __device__ __forceinline__ void wave_lds_fence() const { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
  __device__ __forceinline__ void sync_full() {
    if (PROF == 2 && tl) tl_rec();
    __syncthreads();
    if (PROF == 2 && tl) tl_rec();
  }
  // Barrier timeline (profiling build, tools/barrier_timeline.py): during a few chosen frames of batch item 0 every
  // wave stores the shader clock when it arrives at, and when it leaves, each barrier.
  // (stamps go to LDS -- a global store per stamp would make the compiler's memory waits part of the measurement --
  // and are copied out when the kernel ends)
  // The only mutable state, the per-wave record counters, lives in LDS (tlcnt): mutable members would push this whole
  // object into scratch memory.  A counter at or beyond tl_cap means "not recording".
  long long *tl; int *tlcnt; int tl_cap, tl_f0, tl_nf;
  __device__ __forceinline__ void tl_rec() {
    if ((threadIdx.x & 63) == 0) {
      const int i = atomicAdd(&tlcnt[threadIdx.x >> 6], 1);
      if (i < tl_cap) tl[(threadIdx.x >> 6) * tl_cap + i] = (long long)clock64();
    }
  }
  __device__ __forceinline__ void tick() { if (PROF == 2 && tl) tl_rec(); }  // extra stamp between barriers
  __device__ __forceinline__ void trace_frame(int t) {
    if (PROF == 2 && tl && (threadIdx.x & 63) == 0) {
      if (t == tl_f0) tlcnt[threadIdx.x >> 6] = 0;
      if (t == tl_f0 + tl_nf) tlcnt[threadIdx.x >> 6] = tl_cap;
      // a frame marker (bit 62 set): frames take different paths with different numbers of stamps (speculative select /
      // histogram select / exact replay); tools/barrier_timeline.py groups the recorded frames by their stamp count
      const int i = atomicAdd(&tlcnt[threadIdx.x >> 6], 1);
      if (i < tl_cap) tl[(threadIdx.x >> 6) * tl_cap + i] = (long long)clock64() | (1ll << 62);
    }
  }
  // a value every thread of the workgroup holds identically -> scalar register (branches/loops on it become scalar)
  __device__ __forceinline__ int uni(int v) const { return __builtin_amdgcn_readfirstlane(v); }
  // four consecutive, 16-byte aligned LDS words every thread reads identically: one ds_read_b128
  __device__ __forceinline__ void uni4(const int *p, int *out) const {
    const int4 v = *reinterpret_cast<const int4 *>(p);
    out[0] = __builtin_amdgcn_readfirstlane(v.x); out[1] = __builtin_amdgcn_readfirstlane(v.y);
    out[2] = __builtin_amdgcn_readfirstlane(v.z); out[3] = __builtin_amdgcn_readfirstlane(v.w);
  }
  // ... in two halves: the request, and the move to scalar registers (whatever is requested in between rides behind it)
  __device__ __forceinline__ ctcbeam::Int4v load4(const int *p) const { return *reinterpret_cast<const ctcbeam::Int4v *>(p); }
  __device__ __forceinline__ void uni4v(const ctcbeam::Int4v &v, int *out) const {
    out[0] = __builtin_amdgcn_readfirstlane(v.x); out[1] = __builtin_amdgcn_readfirstlane(v.y);
    out[2] = __builtin_amdgcn_readfirstlane(v.z); out[3] = __builtin_amdgcn_readfirstlane(v.w);
  }
  __device__ __forceinline__ float unif(float v) const { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
  // issue priority of this wave among the waves of its SIMD (0 = default .. 3).  (Raising it for the wave that works
  // alone in a phase, or for the child-scoring waves of phase B, was measured: neutral to 0.6 % slower -- not used.)
  template <int P>
  __device__ __forceinline__ void prio() const { __builtin_amdgcn_s_setprio(P); }
  __device__ __forceinline__ void count(int, int) const {}  // (event statistics of the host build)
  __device__ __forceinline__ void probe_keys(int, int, const uint32_t *, int, uint32_t, const float *, int) const {}
  // a pointer the compiler must treat as new: what it points to is (re)loaded after this point, not kept live before it
  template <class P>
  __device__ __forceinline__ const P *fresh(const P *p) const {
    asm volatile("" : "+s"(p));
    return p;
  }
  // the value lane `idx` holds (idx uniform)
  __device__ __forceinline__ int pick(int v, int idx) const { return __builtin_amdgcn_readlane(v, idx); }
  __device__ __forceinline__ int atomic_add(int *p, int v) { return atomicAdd(p, v); }
  __device__ __forceinline__ void atomic_max(int *p, int v) { atomicMax(p, v); }
  // a "group" = one wave: work items that the 64 lanes search / paint together
  __device__ __forceinline__ int group() const { return __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6); }
  __device__ __forceinline__ int ngroups() const { return (nt() + 63) >> 6; }
  __device__ __forceinline__ int lane() const { return (int)threadIdx.x & 63; }
  __device__ __forceinline__ int lanes() const { return 64; }
  __device__ __forceinline__ unsigned long long ballot(bool p) const { return CTC_BALLOT(p); }
  // first q in [from, n) with arr[q] < bound (n if none); arguments uniform across the wave
  __device__ __forceinline__ int first_below(const int *arr, int from, int n, int bound) const {
    for (int base = from; base < n; base += 64) {
      const int q = base + ((int)threadIdx.x & 63);
      const unsigned long long m = CTC_BALLOT(q < n && arr[q] < bound);
      if (m) return base + __ffsll((long long)m) - 1;
    }
    return n;
  }
  // Phase A1 for a wave that holds several entries with in-beam descendants (`todo`: their lanes; the entry of lane l is
  // row k0 + l of this wave's column walk, beam_core.h step()): four of them at a time, one per quarter of the wave --
  // the end of each one's subtree range (first later entry whose LCP with its predecessor is shallower than the entry),
  // then the entry painted onto its descendants.  Returns false (nothing done) when a single entry is pending: the
  // whole wave then searches for that one.  Beams decoded under a dictionary are long chains: most entries are interior.
  __device__ __forceinline__ bool subtrees_by_quarters(unsigned long long todo, int k0, int grp, int ngr, const int *dep, const int *lcp, int n,
                                                       int *e, int *anc, int *acnt) {
    if ((todo & (todo - 1)) == 0) return false;
    constexpr int L = 16;  // lanes per entry (measured: 8 is slower, 64 = the single-entry search below)
    const int lane = (int)threadIdx.x & 63, sub = lane / L, sl = lane & (L - 1);
    while (todo) {
      int kk = -1;
      for (int s4 = 0; s4 < 64 / L; ++s4) {  // this quarter's entry: the sub-th lowest pending lane
        const int bit = todo ? __builtin_ctzll(todo) : -1;
        if (s4 == sub) kk = bit;
        todo &= todo - 1;  // (0 stays 0)
      }
      const bool act = kk >= 0;
      const int r = k0 + (act ? kk : 0);
      const int jj = r * ngr + ((grp - r) & (ngr - 1));
      const int dj = act ? dep[jj] : 0;
      int q = n;
      bool done = !act;
      for (int base = jj + 2;; base += L) {
        const int p = base + sl;
        const unsigned long long m = CTC_BALLOT(!done && p < n && lcp[p] < dj);
        const unsigned mine = (unsigned)(m >> (L * sub)) & ((1u << L) - 1u);
        if (!done) {
          if (mine) { q = base + __builtin_ctz(mine); done = true; }
          else if (base + L >= n) done = true;
        }
        if (CTC_BALLOT(!done) == 0) break;
      }
      if (act) {
        if (sl == 0) e[jj] = q;
        for (int c = jj + 1 + sl; c < q; c += L) {
          atomicMax(&anc[c], jj);
          atomicAdd(&acnt[c], 1);
        }
      }
    }
    return true;
  }
  __device__ __forceinline__ void atomic_or(uint32_t *p, uint32_t v) { atomicOr(p, v); }

  // ---- speculative select (beam_core.h Decoder::kSpec) -------------------------------------------------------------------
#if defined(CTC_NO_SPEC_SELECT)
  static constexpr bool kSpecSelect = false;
#else
  static constexpr bool kSpecSelect = true;
#endif
#if defined(CTC_NO_RANK_EPOCH)
  static constexpr bool kRankEpoch = false;
#else
  static constexpr bool kRankEpoch = true;  // (beam_core.h kRankEpoch)
#endif
#if defined(CTC_NO_PARENT_REC)
  static constexpr bool kParentRec = false;
#else
  static constexpr bool kParentRec = true;  // (beam_core.h kParentRec: wide beams, one packed record per parent for phase B)
#endif
#if defined(CTC_EXP_SPEC_LM)  // (measured slower, round 6: beam_core.h kSpec)
  static constexpr bool kSpecLm = true;
#else
  static constexpr bool kSpecLm = false;
#endif
  static constexpr bool kQuarters = PROF == 3;  // (beam_core.h phase A1)
  // Round 6 experiments, all measured SLOWER than the kernel without them and therefore off (-DCTC_EXP_... builds them):
#if defined(CTC_EXP_LCP_TABLE)
  static constexpr bool kLcpTable = NT == 1024;  // (beam_core.h kLcpTable: the last wave builds range minima of the LCP array during phase B)
#else
  static constexpr bool kLcpTable = false;
#endif
#if defined(CTC_EXP_A1_OVERLAP)
  static constexpr bool kA1Overlap = NT == 1024;  // (beam_core.h kA1Overlap: phase A1 of the next frame beside the emission; the geometry it assumes)
#else
  static constexpr bool kA1Overlap = false;
#endif
#if defined(CTC_NO_LM_OVERLAP)
  static constexpr bool kLmOverlap = false;
#else
  static constexpr bool kLmOverlap = NT == 1024;  // (beam_core.h kLmOverlap: the geometry it assumes)
#endif
  // (the ranking gives every hot key one lane of the first 128 or 256 threads)
  __device__ __forceinline__ bool spec_fits(int hot) const { return (hot <= 128 ? 128 : 256) <= nt(); }
  __device__ __forceinline__ int spec_thread() const { return nt() - 64; }  // lane 0 of the last wave keeps the prediction
  // Append (key, slot) of every lane whose candidate is hot: each hot lane takes its own place with a returning LDS atomic.
  // (A few percent of the candidates are hot -- two or three lanes of a wave per pass -- so the same-address atomics barely
  //  serialise, and the wave-aggregated form (ballot, lane count, leader election, one atomic, readlane) measured 18
  //  instructions longer per pass on every scoring wave.)  Keys beyond the list's capacity are counted but land in a dummy
  //  place behind it (the select then falls back).
  __device__ __forceinline__ void hot_append(bool hot, uint32_t key, int slot, uint32_t *hk, int *hs, int *cnt) {
    if (hot) {
      int p = atomicAdd(cnt, 1);
      p = p < ctcbeam::kHotCap ? p : ctcbeam::kHotCap;
      hk[p] = key; hs[p] = slot;
    }
  }
  // ... in two halves, for loops that have something to request between them (beam_core.h phase B): the place is requested,
  // and only waited for when the append is committed.
  using HotTicket = int;
  __device__ __forceinline__ HotTicket hot_issue(bool hot, int *cnt) {
    int p = 0;
    if (hot) p = atomicAdd(cnt, 1);
    return p;
  }
  __device__ __forceinline__ void hot_commit(HotTicket p, bool hot, uint32_t key, int slot, uint32_t *hk, int *hs) {
    asm volatile("" : "+v"(p));  // (the place is first LOOKED AT here: the compiler otherwise waits for it right behind the atomic)
    if (hot) {
      p = p < ctcbeam::kHotCap ? p : ctcbeam::kHotCap;
      hk[p] = key; hs[p] = slot;
    }
  }
  // the largest of the V (<= 64) values the first lanes of wave 0 hold -> *dst (float bits), by wave 0
  __device__ __forceinline__ void row_max_store(int *dst, float v, int V) const {
    if (threadIdx.x < 64) {
      const float mine = (int)threadIdx.x < V ? v : -__builtin_huge_valf();
      // (an order-preserving integer image: the DPP scan works on ints)
      const int key = (int)ctcbeam::ord_f32(mine);
      const uint32_t mx = wave_max_u32((uint32_t)key);
      if (threadIdx.x == 0) *dst = (int)ctcmath::f32_to_bits(ctcbeam::unord_f32(mx));
    }
  }
  // acc + the number of the four keys that are >= mine.  Hand-scheduled: a compare into a scalar register pair, then an
  // add-with-carry that takes that pair as its carry-in -- two instructions per key.  (The compiler's form is compare,
  // wait states for the VCC hazard, select / add: three to four issue slots per key, and this runs on all sixteen waves.)
  __device__ __forceinline__ int count_ge4(const uint4 o, const uint32_t mine, int acc) const {
    int r;
    unsigned long long m0, m1, m2;
    asm volatile(
        "v_cmp_ge_u32_e64 %1, %4, %8\n\t"
        "v_cmp_ge_u32_e64 %2, %5, %8\n\t"
        "v_cmp_ge_u32_e64 %3, %6, %8\n\t"
        "v_cmp_ge_u32_e64 vcc, %7, %8\n\t"
        "v_addc_co_u32_e64 %0, %1, %9, 0, %1\n\t"
        "v_addc_co_u32_e64 %0, %2, %0, 0, %2\n\t"
        "v_addc_co_u32_e64 %0, %3, %0, 0, %3\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, 0, vcc"
        : "=&v"(r), "=&s"(m0), "=&s"(m1), "=&s"(m2)
        : "v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w), "v"(mine), "v"(acc)
        : "vcc");
    return r;
  }
  // ... the same through the wave: every lane of a B1 wave may be hot (an entry's own new score usually is), and sixty-four
  // same-address atomics would serialise: ballot, one atomic by the first hot lane, places by lane count.
  __device__ __forceinline__ void hot_append_wave(bool hot, uint32_t key, int slot, uint32_t *hk, int *hs, int *cnt) {
    const unsigned long long m = CTC_BALLOT(hot);
    if (m) {  // (uniform)
      const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
      int base = 0;
      if (hot && below == 0) base = atomicAdd(cnt, __popcll(m));
      base = __builtin_amdgcn_readlane(base, __builtin_ctzll(m));
      int p = base + below;
      p = p < ctcbeam::kHotCap ? p : ctcbeam::kHotCap;
      if (hot) { hk[p] = key; hs[p] = slot; }
    }
  }
  // The hot list hk[0, H) / hs[0, H) (K <= H <= kHotCap; hk zero beyond H up to kHotCap) holds every candidate key at or
  // above the frame's threshold.
  //   Stage 1, all waves: thread t stands for key q = t mod L (L = 128 or 256 lanes of keys) and compares it with its
  //   share of the list (part t / L: 16 keys at 1024 threads and H <= 128 -- four 128-bit reads, the same addresses in every
  //   lane of a wave); the partial counts of "keys >= mine" meet in gearr[q] (LDS atomics; zero on entry, and left zero).
  //   Stage 2, the first L threads only (two or four waves; the others go straight to the barrier -- an instruction that all
  //   sixteen waves execute costs sixteen clocks, one that two execute costs four or five): a key survives iff its count is
  //   <= K, the K-th key is the one whose count is K -- it exists iff no group of equal keys straddles the boundary, and its
  //   lane reports it (res[0] = the key, res[2] = K); survivors set their bit in the (zeroed) bitmap.
  //   Stage 3, the same threads: a prefix count over the bitmap's 128-bit groups (one per lane, redundantly per wave) gives
  //   each survivor its rank in slot (= DFS) order: surv[rank] = slot.  The other waves meanwhile fetch the report, so that
  //   nobody reads LDS behind the closing barrier.
  // res[2] must be zero on entry.  Returns {the K-th key, 1} or {-, 0}: then equal keys straddle the boundary and nothing
  // was written to surv[].  Three barriers; everything is visible when it returns.
  // (Measured and dropped: one key per lane + sixteen readlanes into scalar compare operands instead of four broadcast
  //  128-bit reads -- sixteen times fewer LDS bytes, 1.5 % slower: the stage is bound by the instructions every wave issues
  //  behind the barrier, not by LDS bandwidth.  Stages 1 and 2 as one stage without LDS traffic -- every wave holds the list in two registers and
  //  ranks eight keys of its own through scalar registers: readlane, two compares, two population counts, writelane; one
  //  barrier less, no atomics -- 3 % SLOWER: ~85 instructions on all sixteen waves against ~45 + a two-wave stage.)
  struct SpecPre { uint32_t mine; uint4 o0, o1, o2, o3; };
  // (the list reads of stage 1 in the usual configuration -- 1024 threads, at most 128 hot keys -- requested before the
  //  list's length is known: they ride with the read of that length instead of behind it)
  __device__ __forceinline__ SpecPre spec_pre(const uint32_t *hk, const int *) const {
    SpecPre pre;
    pre.mine = 0u; pre.o0 = pre.o1 = pre.o2 = pre.o3 = make_uint4(0u, 0u, 0u, 0u);
    if (NT == 1024) {
      const int t = (int)threadIdx.x;
      pre.mine = hk[t & 127];
      const uint4 *src = reinterpret_cast<const uint4 *>(hk + (t >> 7) * 16);
      pre.o0 = src[0]; pre.o1 = src[1]; pre.o2 = src[2]; pre.o3 = src[3];
    }
    return pre;
  }
  struct SpecResult { uint32_t tau; int ok; };
  __device__ __forceinline__ SpecResult spec_select(const SpecPre &pre, int H, int K, const uint32_t *hk, const int *hs, uint32_t *bitmap, int *gearr, int S, int *surv,
                                                    int *res) {
    const int t = (int)threadIdx.x, n = nt();
    const int lsh = H <= 128 ? 7 : 8, L = 1 << lsh;
    const int q = t & (L - 1), part = t >> lsh;
    uint32_t mine;
    int ge = 0;
    if (NT == 1024 && H <= 128) {
      mine = pre.mine;
#if defined(CTC_EXP_TICKS)
      tick();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      tick();
#endif
      ge = count_ge4(pre.o3, mine, count_ge4(pre.o2, mine, count_ge4(pre.o1, mine, count_ge4(pre.o0, mine, 0))));
    } else {
      const int per = ctcbeam::div_p2(L << lsh, n);  // keys per part: a multiple of 8 when n <= 1024
      mine = hk[q];
      const uint4 *src = reinterpret_cast<const uint4 *>(hk + part * per);
      for (int i = 0; i < (per >> 2); i += 2) {
        const uint4 o0 = src[i], o1 = src[i + 1];
        ge = count_ge4(o1, mine, count_ge4(o0, mine, ge));
      }
    }
#if defined(CTC_EXP_TICKS)
    asm volatile("" : "+v"(ge));
    tick();
#endif
    atomicAdd(&gearr[q], ge);
    sync();
#if defined(CTC_EXP_RANK_ALLPAIRS)
    // Round 6, VERDICT r5 item 1(b), measurement build only: the survivors' ranks in slot order by the all-pairs machinery on ALL
    // sixteen waves -- a survivor's rank = number of survivors with a smaller slot -- instead of the bitmap + prefix chain on the two
    // lead waves.  Thread (q, part) looks at the sixteen hot keys of its part: their counts (complete since the barrier above) say
    // which of them survive, their slots which of those lie before key q's.  Costs ~75 instructions on every wave against ~45 + ~60
    // on two: measured slower.
    if (NT == 1024 && H <= 128) {
      const int myslot = q < H ? hs[q] : 0x7fffffff;
      const int4 *gp = reinterpret_cast<const int4 *>(gearr + part * 16);
      const int4 *sp = reinterpret_cast<const int4 *>(hs + part * 16);
      int cnt = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int4 g4 = gp[i], s4 = sp[i];
        cnt += ((unsigned)(g4.x - 1) < (unsigned)K && s4.x < myslot) + ((unsigned)(g4.y - 1) < (unsigned)K && s4.y < myslot) +
               ((unsigned)(g4.z - 1) < (unsigned)K && s4.z < myslot) + ((unsigned)(g4.w - 1) < (unsigned)K && s4.w < myslot);
      }
      atomicAdd(&gearr[128 + q], cnt);
      sync();
      if (t < 128) {
        const int g = gearr[t], rk = gearr[128 + t];
        gearr[t] = 0; gearr[128 + t] = 0;
        const bool valid = t < H;
        if (valid && g == K) { res[0] = (int)mine; res[2] = K; }
        if (valid && g <= K) surv[rk] = hs[t];
      }
      sync();
      const int4 rv2 = *reinterpret_cast<const int4 *>(res);
      SpecResult r2;
      r2.tau = (uint32_t)__builtin_amdgcn_readfirstlane(rv2.x);
      r2.ok = __builtin_amdgcn_readfirstlane(rv2.z) == K ? 1 : 0;
      sync();  // (nobody reads the report behind this point: the caller resets it during the emission)
      return r2;
    }
#endif
    const bool lead = t < L;  // (whole waves)
    bool keep = false;
    int slot = 0;
    if (lead) {
      const int g = gearr[t];
      gearr[t] = 0;  // (for the next frame)
      const bool valid = t < H;
      slot = valid ? hs[t] : 0;
      keep = valid && g <= K;
      if (valid && g == K) { res[0] = (int)mine; res[2] = K; }
      if (keep) atomicOr(&bitmap[slot >> 5], 1u << (slot & 31));
    }
    sync();
    const int4 rv = *reinterpret_cast<const int4 *>(res);  // (consumed behind the closing barrier)
    if (lead) {
      const int lane = t & 63;
      const int ngr = (S + 127) >> 7;  // 128-bit groups of the bitmap (at most 64: S <= 8192)
      uint4 g4 = make_uint4(0u, 0u, 0u, 0u);
      if (lane < ngr) g4 = *reinterpret_cast<const uint4 *>(bitmap + 4 * lane);
      const uint4 m4 = *reinterpret_cast<const uint4 *>(bitmap + 4 * (slot >> 7));  // my own group (group 0 for lanes without a key)
      const int cnt = __popc(g4.x) + __popc(g4.y) + __popc(g4.z) + __popc(g4.w);
      const int incl = wave_scan(cnt, 0, [](int a, int b) { return a + b; });
      const int pre_g = __shfl(incl - cnt, slot >> 7, 64);  // survivors in the groups below mine
      const int wi = (slot >> 5) & 3;
      const uint32_t wsel = wi == 0 ? m4.x : wi == 1 ? m4.y : wi == 2 ? m4.z : m4.w;
      const int inner = (wi > 0 ? __popc(m4.x) : 0) + (wi > 1 ? __popc(m4.y) : 0) + (wi > 2 ? __popc(m4.z) : 0) + __popc(wsel & ((1u << (slot & 31)) - 1u));
      if (__builtin_amdgcn_readfirstlane(rv.z) == K && keep) surv[pre_g + inner] = slot;
    }
    sync();
    SpecResult r;
    r.tau = (uint32_t)__builtin_amdgcn_readfirstlane(rv.x);
    r.ok = __builtin_amdgcn_readfirstlane(rv.z) == K ? 1 : 0;
    return r;
  }

  // (a & mask) | (b & ~mask): one v_bitop3_b32 (the compiler builds it from three instructions)
  __device__ __forceinline__ uint32_t bitsel(uint32_t mask, uint32_t a, uint32_t b) const { return __builtin_amdgcn_bitop3_b32(mask, a, b, 0xCA); }
  // results mirrored into host memory: make this thread's stores visible system-wide / publish a flag there
  __device__ __forceinline__ void fence_system() const { __threadfence_system(); }
  // streamed input: a counter another agent (the copy stream) advances in uncached memory; a short pause between two looks
  __device__ __forceinline__ int load_system(const int *p) const { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }
  __device__ __forceinline__ void nap() const { __builtin_amdgcn_s_sleep(8); }
  __device__ __forceinline__ void store_system(int32_t *p, int v) const { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
  // sum over the aligned group of eight lanes this lane belongs to (every lane of the wave must call it)
  __device__ __forceinline__ int sum8(int v) const {
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    return v;
  }
  // sum over the aligned group of four lanes this lane belongs to (two quad permutes; the whole quad must be active)
  __device__ __forceinline__ int sum4(int v) const {
    v += CTC_DPP(0, v, 0xB1, 0xf);  // quad_perm:[1,0,3,2]
    v += CTC_DPP(0, v, 0x4E, 0xf);  // quad_perm:[2,3,0,1]
    return v;
  }
  // *p += number of lanes of this wave whose flag is set: a ballot's population, one LDS atomic
  __device__ __forceinline__ void wave_add_flag(int *p, bool f) {
    const int c = __popcll(CTC_BALLOT(f));
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(p, c);
  }
  // one LDS atomic per wave
  __device__ __forceinline__ void wave_add(int *p, int v) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(p, v);
  }

  __device__ __forceinline__ void wave_max_to(int *p, uint32_t v) {
    v = wave_max_u32(v);
    if ((threadIdx.x & 63) == 0 && v) atomicMax((unsigned *)p, v);
  }
  __device__ __forceinline__ unsigned global_add(unsigned *p, unsigned v) { return atomicAdd(p, v); }
  __device__ __forceinline__ int item() const { return (int)blockIdx.x; }  // batch item of this workgroup
  __device__ __forceinline__ void wave_min_to(int *p, uint32_t v) {
    v = ~wave_max_u32(~v);
    if ((threadIdx.x & 63) == 0) atomicMin((unsigned *)p, v);
  }

  // bit s of bitmap = pred(s), for every slot s in [0, S): each wave owns a contiguous range of slots (the same mapping
  // as compact_slots), so one ballot is one 64-bit word of the bitmap.  pred may have side effects (list appends).
  template <class Pred>
  __device__ __forceinline__ void mark_slots(int S, uint32_t *bitmap, Pred pred) {
    const int lane = (int)threadIdx.x & 63, nw = (nt() + 63) >> 6;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int rounds = ctcbeam::ceil_div_p2(S, 64 * nw);
    const int first = wave * rounds * 64;
    if (rounds <= 4) {  // common case: straight-line, the four predicates (and their LDS reads) issued together
      const int s0 = first + lane, s1 = s0 + 64, s2 = s0 + 128, s3 = s0 + 192;
      const bool f0 = s0 < S && pred(s0);
      const bool f1 = rounds > 1 && s1 < S && pred(s1);
      const bool f2 = rounds > 2 && s2 < S && pred(s2);
      const bool f3 = rounds > 3 && s3 < S && pred(s3);
      const unsigned long long m0 = CTC_BALLOT(f0), m1 = CTC_BALLOT(f1), m2 = CTC_BALLOT(f2), m3 = CTC_BALLOT(f3);
      if (lane < rounds) {
        const unsigned long long m = lane == 0 ? m0 : lane == 1 ? m1 : lane == 2 ? m2 : m3;
        bitmap[2 * (wave * rounds + lane)] = (uint32_t)m;
        bitmap[2 * (wave * rounds + lane) + 1] = (uint32_t)(m >> 32);
      }
      return;
    }
    for (int it = 0; it < rounds; ++it) {
      const int s = first + it * 64 + lane;
      const unsigned long long m = CTC_BALLOT(s < S && pred(s));
      if (lane == 0) {
        bitmap[2 * (wave * rounds + it)] = (uint32_t)m;
        bitmap[2 * (wave * rounds + it) + 1] = (uint32_t)(m >> 32);
      }
    }
  }
  // One pass over the slot keys for the select: the keys inside the bucket [b32, b32 + bspan] are appended to list[]
  // (key offset + 1) / lslot[] (slot), and -- when `direct` -- bit s of the bitmap says "key above the bucket".
  // A wave reserves the list space of ALL its slots with ONE returning LDS atomic: a returning atomic costs a full LDS
  // round trip, and one per bucket member (in divergent code, once per round) was most of this pass.
  // TZ ("tail is zero"): the caller keeps every key from slot S up to the next multiple of the workgroup's slots-per-pass
  // at zero (beam_core.h: identity mode without a scorer -- S never shrinks there), so up to four rounds per wave read
  // their keys without a bounds test: a guarded load is seven instructions on every wave, a plain one is one.
  template <int R, bool TZ>
  __device__ __forceinline__ void list_group(int S, const uint32_t *skey, int s0, uint32_t b32, uint32_t bspan, bool direct, uint32_t *bmw,
                                             uint32_t *list, int *lslot, int *lcount) {
    const int lane = (int)threadIdx.x & 63;
#define CTC_BELOW(m) ((int)__builtin_amdgcn_mbcnt_hi((unsigned)((m) >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)(m), 0u)))
    // One compare per question (the lane mask of a single compare IS the ballot; a conjunction of two is first
    // materialised as 0 / 1 per lane and compared again): b32 + bspan never exceeds 2^32 - 1 (the bucket lies inside the
    // key range), so a key below the bucket wraps to an offset beyond bspan, and "above the bucket" is one compare with
    // its top.  (b32 >= 1: holes, key 0, never pass.)
    const uint32_t top = b32 + bspan;
    const unsigned long long dm = direct ? ~0ull : 0ull;  // (uniform)
    uint32_t k[R], d[R];
    unsigned long long m[R], a[R];
    int c[R], tot = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) k[r] = (TZ || s0 + 64 * r < S) ? skey[s0 + 64 * r] : 0u;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      d[r] = k[r] - b32;
      m[r] = CTC_BALLOT(d[r] <= bspan);
      a[r] = CTC_BALLOT(k[r] > top) & dm;
      c[r] = __popcll(m[r]);
      tot += c[r];
    }
    if (tot) {  // (uniform)
      int base = 0;
      if (lane == 0) base = atomicAdd(lcount, tot);
      base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (d[r] <= bspan) { const int p = base + CTC_BELOW(m[r]); list[p] = d[r] + 1u; lslot[p] = s0 + 64 * r; }
        base += c[r];
      }
    }
    // the R words of the bitmap, by one lane (the masks are scalars: no per-lane selection among them)
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) *reinterpret_cast<uint2 *>(bmw + 2 * r) = make_uint2((uint32_t)a[r], (uint32_t)(a[r] >> 32));
    }
#undef CTC_BELOW
  }
  // COMPACT: the rare callers (the select's fallback rounds) take one round at a time -- one small copy of the code
  // instead of another set of straight-line instantiations in a kernel that lives in a 64 KB instruction cache.
  template <bool TZ = false, bool COMPACT = false>
  __device__ __forceinline__ void list_bucket(int S, const uint32_t *skey, uint32_t b32, uint32_t bspan, bool direct, uint32_t *bitmap,
                                              uint32_t *list, int *lslot, int *lcount) {
    const int lane = (int)threadIdx.x & 63, nw = (nt() + 63) >> 6;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int rounds = ctcbeam::ceil_div_p2(S, 64 * nw);
    const int first = wave * rounds * 64;
    // four rounds at a time, straight-line: their keys are requested together (one LDS round trip), and one returning
    // atomic reserves the list space of all four.  (Wide beams take several such groups: 15 rounds at beam 500.)
    if (!COMPACT && TZ && (rounds == 3 || rounds == 4)) {  // the usual shapes: one group, no bounds tests
      uint32_t *bmw = bitmap + 2 * (wave * rounds);
      if (rounds == 3) list_group<3, true>(S, skey, first + lane, b32, bspan, direct, bmw, list, lslot, lcount);
      else list_group<4, true>(S, skey, first + lane, b32, bspan, direct, bmw, list, lslot, lcount);
      return;
    }
    if (COMPACT || TZ) {
      for (int r0 = 0; r0 < rounds; ++r0)
        list_group<1, false>(S, skey, first + r0 * 64 + lane, b32, bspan, direct, bitmap + 2 * (wave * rounds + r0), list, lslot, lcount);
      return;
    }
    for (int r0 = 0; r0 < rounds; r0 += 4) {
      const int nr = rounds - r0;  // rounds left (uniform)
      uint32_t *bmw = bitmap + 2 * (wave * rounds + r0);
      const int s0 = first + r0 * 64 + lane;
      if (nr >= 4) list_group<4, false>(S, skey, s0, b32, bspan, direct, bmw, list, lslot, lcount);
      else if (nr == 3) list_group<3, false>(S, skey, s0, b32, bspan, direct, bmw, list, lslot, lcount);
      else if (nr == 2) list_group<2, false>(S, skey, s0, b32, bspan, direct, bmw, list, lslot, lcount);
      else list_group<1, false>(S, skey, s0, b32, bspan, direct, bmw, list, lslot, lcount);
    }
  }
  // bit s of the bitmap = (skey[s] >= tau), for every word that covers [0, S): one ballot per 64 slots
  __device__ __forceinline__ void mark_ge(int S, const uint32_t *skey, uint32_t tau, uint32_t *bitmap) {
    const int lane = (int)threadIdx.x & 63, nw = (nt() + 63) >> 6;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    // (four groups of 64 slots per trip: their keys are requested together -- one LDS round trip instead of four)
    const int step = nw * 64;
    for (int s0 = wave * 64; s0 < S; s0 += 4 * step) {
      uint32_t k[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int s = s0 + u * step + lane;
        k[u] = s < S ? skey[s] : 0u;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int b0 = s0 + u * step;
        if (b0 >= S) break;
        const unsigned long long m = CTC_BALLOT(k[u] >= tau);
        if (lane == 0) { bitmap[2 * (b0 >> 6)] = (uint32_t)m; bitmap[2 * (b0 >> 6) + 1] = (uint32_t)(m >> 32); }
      }
    }
  }
  // out[r] = s for the r-th set bit s of the bitmap (ascending); one wave, the others wait at the closing barrier.
  // CLUSTERED: the set bits come in long runs (the LM tier: a dictionary-constrained beam keeps the children of few
  // parents) -- one lane per BYTE of the bitmap on as many waves as that takes, instead of one lane per 64-bit word on
  // one wave whose lanes would loop over dozens of bits while their neighbours idle.  (Measured: −1 % per LM frame; no
  // change for the plain kernel, which keeps the single-wave form.)
  template <bool CLUSTERED = false>
  __device__ __forceinline__ void expand_bitmap(const uint32_t *bitmap, int nwords64, int *out) {
    if (CLUSTERED && nwords64 <= 64 && nwords64 * 8 <= nt()) {
      const int lane = (int)threadIdx.x & 63;
      const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
      if (wave * 64 < nwords64 * 8) {
        // every participating wave scans the words' populations for itself; a lane's first rank = (bits in lower words)
        // + (bits in the lower bytes of its word)
        const int g = wave * 64 + lane, wi = g >> 3, by = g & 7;
        const unsigned long long *b64 = reinterpret_cast<const unsigned long long *>(bitmap);
        const unsigned long long mineword = lane < nwords64 ? b64[lane] : 0ull;
        const unsigned long long w = wi < nwords64 ? b64[wi] : 0ull;
        const int cnt = __popcll(mineword);
        const int incl = wave_scan(cnt, 0, [](int a, int b) { return a + b; });
        const int pfx = __shfl(incl - cnt, wi & 63, 64);
        unsigned bits = (unsigned)(w >> (8 * by)) & 0xFFu;
        int base = pfx + __popcll(w & ((1ull << (8 * by)) - 1ull));
        const int s0 = wi * 64 + 8 * by;
        while (bits) {
          out[base++] = s0 + __builtin_ctz(bits);
          bits &= bits - 1u;
        }
      }
      sync();
      return;
    }
    if (nwords64 > 64 && ((nwords64 + 63) & ~63) <= nt()) {
      // wide beams (beam 500: 243 words): one wave per 64 words, all at once; a wave first adds up the populations of the
      // segments before its own (round 2 walked the segments one after the other on wave 0)
      const int lane = (int)threadIdx.x & 63;
      const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
      if (wv * 64 < nwords64) {
        int running = 0;
        for (int seg = 0; seg < wv; ++seg) {
          const int wj = seg * 64 + lane;
          running += wave_sum(__popc(bitmap[2 * wj]) + __popc(bitmap[2 * wj + 1]));
        }
        const int wi = wv * 64 + lane;
        unsigned long long word = 0ull;
        if (wi < nwords64) word = (unsigned long long)bitmap[2 * wi] | ((unsigned long long)bitmap[2 * wi + 1] << 32);
        const int cnt = __popcll(word);
        const int incl = wave_scan(cnt, 0, [](int a, int b) { return a + b; });
        int base = running + incl - cnt;
        while (word) {
          out[base++] = wi * 64 + __builtin_ctzll(word);
          word &= word - 1ull;
        }
      }
      sync();
      return;
    }
    if (threadIdx.x < 64) {
      const int lane = (int)threadIdx.x;
      int running = 0;
      for (int w0 = 0; w0 < nwords64; w0 += 64) {
        const int wi = w0 + lane;
        unsigned long long word = 0ull;
        if (wi < nwords64) word = (unsigned long long)bitmap[2 * wi] | ((unsigned long long)bitmap[2 * wi + 1] << 32);
        const int cnt = __popcll(word);
        const int incl = wave_scan(cnt, 0, [](int a, int b) { return a + b; });
        int base = running + incl - cnt;
        while (word) {
          out[base++] = wi * 64 + __builtin_ctzll(word);
          word &= word - 1ull;
        }
        running += __builtin_amdgcn_readlane(incl, 63);
      }
    }
    sync();
  }

  // Ordered compaction: out[r] = s for every slot s in [0, S) with pred(s), r = number of such slots below s.
  // Each wave owns a contiguous range of slots (consecutive lanes = consecutive slots), so the rank of a slot is
  // (survivors in lower waves) + (survivors in this wave's earlier rounds) + (set ballot bits below the lane):
  // no atomics, no sorting.  Starts and ends with a barrier-consistent state (caller synced before; syncs after).
  template <class Pred>
  __device__ __forceinline__ void compact_slots(int S, int *out, Pred pred) {
    compact_slots_to(S, pred, [=](int r, int s) { out[r] = s; });
  }
  // ... the general form: emit(r, s) is called once for every slot s that passes, r = its rank among them
  template <class Pred, class Emit>
  __device__ __forceinline__ void compact_slots_to(int S, Pred pred, Emit emit) {
    const int lane = (int)threadIdx.x & 63, nw = (nt() + 63) >> 6;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int rounds = ctcbeam::ceil_div_p2(S, 64 * nw);
    const int first = wave * rounds * 64;
    int *row = red + parity * 16;
    parity ^= 1;
    // (one general form: since the select hands over a bitmap this compaction only runs on the rare paths -- ties resolved by
    //  character, the exact replay's candidate list -- and a straight-line special case for few rounds was 2 KB of code
    //  in the frame loop of every kernel)
#define CTC_BELOW(m) ((int)__builtin_amdgcn_mbcnt_hi((unsigned)((m) >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)(m), 0u)))
    unsigned long long flags = 0ull;
    int cnt = 0;
    for (int it = 0; it < rounds; ++it) {
      const int s = first + it * 64 + lane;
      const bool f = s < S && pred(s);
      cnt += __popcll(CTC_BALLOT(f));
      if (f && it < 64) flags |= 1ull << it;
    }
    if (lane == 0) row[wave] = cnt;
    sync();
    // exclusive prefix over the <= 16 wave totals: one LDS read per lane + a row-level DPP scan, then one readlane
    int tot = lane < nw ? row[lane] : 0;
    tot += CTC_DPP(0, tot, 0x111, 0xf); tot += CTC_DPP(0, tot, 0x112, 0xf);
    tot += CTC_DPP(0, tot, 0x114, 0xf); tot += CTC_DPP(0, tot, 0x118, 0xf);
    int base = wave > 0 ? __builtin_amdgcn_readlane(tot, wave - 1) : 0;
    for (int it = 0; it < rounds; ++it) {
      const int s = first + it * 64 + lane;
      const bool f = it < 64 ? ((flags >> it) & 1ull) != 0ull : (s < S && pred(s));
      const unsigned long long m = CTC_BALLOT(f);
      if (f) emit(base + CTC_BELOW(m), s);
      base += __popcll(m);
    }
#undef CTC_BELOW
    sync();
  }

  // Histogram complete (caller synced): bins[0, 1024).  Wave 0 finds the bucket holding the need-th largest key: every
  // lane adds up 16 consecutive buckets, a suffix scan over those 64 sums picks the group, a 16-lane suffix scan inside
  // it picks the bucket.  Everyone gets out[0..3] = {bucket or -1, #keys above it, #keys
  // total, #keys in it} after the closing barrier.
  __device__ __forceinline__ void find_bucket(const int *bins, int need, int *out) {
    if (threadIdx.x < 64) {
      const int lane = (int)threadIdx.x;
      const int c = 63 - lane;  // lane 0 owns the TOP coarse bucket: a prefix scan over lanes is a suffix sum over buckets
      // its 16 fine buckets: four 128-bit reads, one round trip.  (Measured in round 3: reading them one dword per step in a
      // lane-rotated, bank-conflict-free order is 1 % SLOWER -- 16 dependent address computations cost more than the
      // conflicts of the four wide reads.)
      const int4 *f4 = reinterpret_cast<const int4 *>(bins + 16 * c);
      const int4 q0 = f4[0], q1 = f4[1], q2 = f4[2], q3 = f4[3];
      const int cv = ((q0.x + q0.y) + (q0.z + q0.w)) + ((q1.x + q1.y) + (q1.z + q1.w)) + ((q2.x + q2.y) + (q2.z + q2.w)) + ((q3.x + q3.y) + (q3.z + q3.w));
      const int incl = wave_scan(cv, 0, [](int a, int b) { return a + b; });
      const int total = __builtin_amdgcn_readlane(incl, 63);
      const unsigned long long m = CTC_BALLOT(incl >= need);
      if (m == 0ull) {
        if (lane == 0) { out[0] = -1; out[1] = 0; out[2] = total; out[3] = 0; }
      } else {
        const int l1 = __ffsll((long long)m) - 1;            // lane of the coarse bucket holding the key
        const int cstar = 63 - l1;
        const int above_c = __builtin_amdgcn_readlane(incl, l1) - __builtin_amdgcn_readlane(cv, l1);
        const int fv = lane < 16 ? bins[cstar * 16 + (15 - lane)] : 0;  // its 16 fine buckets, top one in lane 0
        const int fincl = wave_scan(fv, 0, [](int a, int b) { return a + b; }) + above_c;
        const unsigned long long mf = CTC_BALLOT(lane < 16 && fincl >= need);
        const int l2 = __ffsll((long long)mf) - 1;
        if (lane == l2) { out[0] = cstar * 16 + (15 - lane); out[1] = fincl - fv; out[2] = total; out[3] = fv; }
      }
    }
    sync();
  }

  // Exclusive prefix (in thread order) and total of one 32-bit value per thread; contains one barrier.  (Used with two
  // 16-bit counters packed into the word.)
  // group operations of stl_emul.h hoare_round_parallel: a group is a wavefront
  __device__ __forceinline__ int count(uint64_t m) const { return __builtin_popcountll(m); }
  __device__ __forceinline__ int count_below(uint64_t m) const { return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
  __device__ __forceinline__ uint32_t first_lane(uint32_t v) const { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
  __device__ __forceinline__ void block_scan_u32(uint32_t mine, uint32_t *base_out, uint32_t *total_out) {
    const int t = (int)threadIdx.x, wave = t >> 6, nw = (nt() + 63) >> 6;
    const uint32_t incl = (uint32_t)wave_scan((int)mine, 0, [](int a, int b) { return a + b; });
    int *row = red + parity * 16;
    parity ^= 1;
    if ((t & 63) == 63) row[wave] = (int)incl;
    sync();
    uint32_t tot = (t & 63) < nw ? (uint32_t)row[t & 63] : 0u;
    tot += (uint32_t)CTC_DPP(0, tot, 0x111, 0xf); tot += (uint32_t)CTC_DPP(0, tot, 0x112, 0xf);
    tot += (uint32_t)CTC_DPP(0, tot, 0x114, 0xf); tot += (uint32_t)CTC_DPP(0, tot, 0x118, 0xf);
    const uint32_t below = wave > 0 ? (uint32_t)__builtin_amdgcn_readlane((int)tot, wave - 1) : 0u;
    *base_out = below + incl - mine;
    *total_out = (uint32_t)__builtin_amdgcn_readlane((int)tot, nw - 1);
  }

  // In-place exclusive prefix sum of a[0, n) in LDS; returns the total.  Each thread owns a contiguous chunk.
  __device__ __forceinline__ uint32_t scan_excl(uint32_t *a, int n) {
    const int nthreads = nt(), t = (int)threadIdx.x;
    const int chunk = ctcbeam::ceil_div_p2(n, nthreads) | 1;  // odd stride: no LDS bank conflicts across lanes
    const int lo = min(t * chunk, n), hi = min(lo + chunk, n);
    uint32_t sum = 0;
    for (int i = lo; i < hi; ++i) sum += a[i];
    const uint32_t incl = (uint32_t)wave_scan((int)sum, 0, [](int x, int y) { return x + y; });
    const int wave = t >> 6, nw = (nthreads + 63) >> 6;
    int *row = red + parity * 16;
    parity ^= 1;
    if ((t & 63) == 63) row[wave] = (int)incl;
    sync();
    uint32_t base = 0, total = 0;
    for (int i = 0; i < nw; ++i) {
      const uint32_t v = (uint32_t)row[i];
      if (i < wave) base += v;
      total += v;
    }
    uint32_t run = base + incl - sum;
    for (int i = lo; i < hi; ++i) {
      const uint32_t v = a[i];
      a[i] = run;
      run += v;
    }
    sync();
    return total;
  }
};

constexpr int kTimelineCap = 128;  // barrier timeline entries per wave (16 KB of LDS in the profiling build)

struct KernelArgs {
  const float *probs;       // [B, T, V] log-probabilities
  const int32_t *seq_lens;  // [B] or null
  int B, T, V, K, blank;
  Dims dims;
  PoolNode *pool;           // [B, pool_stride]
  int *pool_up;             // [B, 2, pool_stride] express pointers (beam_core.h kExpress), then the time steps' high parts
  long long pool_stride;
  const uint64_t *tables;   // 64 words (exact_math.h)
  OutRefs outs;             // result tensors; read through the kernel-argument segment at the end of an utterance
  int32_t *status;          // [B]
  int32_t *shape;           // [B] or null: 16 x (entries of the item's final beam with descendants in it) -- chains or bushes
  long long *prof;          // [B, 16] phase timers (profiling build of the kernel only)
  int *dbg;                 // profiling build: beam of item 0 after every frame, [T][1 + 4K] (or null)
  long long *tl;            // profiling build: barrier timeline of item 0, [waves][tl_cap] (or null)
  int tl_cap, tl_f0, tl_nf;
  char *far;                // per-utterance HBM scratch (exact-replay arrays; wide-beam layouts: more), far_stride bytes each
  long long far_stride;
  // streaming (ctcd_stream_decode): per item, the HBM block that holds its parked beam + node pool
  char **st_base;           // [B] or null
  const int *st_poolcap;    // [B] nodes the pool of each stream can hold
  const unsigned char *st_eos;  // [B] 1: this call ends the stream (run decode())
  long long st_pool_off;    // byte offset of the node pool inside a stream block
  // LM tier (ctcd_beam_decode_lm): the scorer's tables, and the caller's own rows (the blank's log-probability is taken
  // from them, ctc_beam_search_decoder.cpp:78)
  ctclm::LmView lm;
  const float *raw;         // [B, T, V] as given by the caller
  int raw_log;              // 1: they are log-probabilities
  const int *frames_ready;  // streamed input (host-tensor entry point): frames of every utterance that have arrived; null: all
  // host-side scorer hook (resumed launches): per item, the frame of its [T, V] rows this launch starts at (null: 0), and
  // where the kernel reports the frames the item's parked state has consumed when the launch ends (null: nowhere)
  const int *frame_off;
  int *frames_done;
  const int *pr_cnt;        // pruned mode: [B, T] candidates per frame (null in identity mode)
  const int *pr_ch;         //              [B, T, pr_stride] their labels, reference order
  const float *pr_lp;       //              [B, T, pr_stride] their log-probabilities
  int pr_stride;
  // host-side scorer hook, the launch that WAITS for its answers (LM == 3 builds; null: a miss ends the utterance's launch as before).
  // All in page-locked host memory.  An utterance that misses parks, and its workgroup stays: the host -- polling the miss list -- asks
  // the callback and appends the answered cache slots to a log; the workgroup copies the log's new entries into the device table
  // (every workgroup applies every entry itself: its own stores are what its later loads are guaranteed to see) and takes the utterance
  // up again from its parked state, as a new launch would.
  const unsigned *cb_log_len;        // entries published so far (only grows); 0xFFFFFFFF: give up waiting (the host relaunches)
  const uint32_t *cb_log_idx;        // [cap] slot index in lm.ng
  const ctclm::NgSlot *cb_log_slot;  // [cap] its contents
  int32_t *cb_done;                  // [B] 1 + status once the workgroup has left
  const unsigned *cb_ans;            // [B] queued pairs of the item the host has dealt with (with their slots in the log by then)
};

// LAYOUT: 0 = the workspace is laid out for the call's own beam width / vocabulary (array bases are run-time values);
// 1 = fixed layout for beam <= kFixedK, vocabulary <= kFixedV: every LDS array sits at a compile-time address, which
// frees the scalar registers the bases would occupy and folds them into the instructions' offset fields.
constexpr int kFixedK = ctcbeam::kSmallK, kFixedV = ctcbeam::kSmallV;
__host__ __device__ constexpr Dims fixed_layout_dims(bool lm = false) { return Dims{kFixedK, kFixedV, kFixedV, 1, lm ? 1 : 0}; }
__host__ __device__ inline bool fits_fixed_layout(const Dims &d) { return d.K <= kFixedK && d.V <= kFixedV && d.Vc_max <= kFixedV; }
// LAYOUT 2 (round 6): the second class with a compile-time layout -- beam <= kMidK, <= kMidVc candidates per frame of a pruned
// vocabulary of <= kMidV labels (beam_core.h kMidK: the reference's default decoder on a large vocabulary, BASELINE configs[3]).
__host__ __device__ constexpr Dims mid_layout_dims() { return Dims{ctcbeam::kMidK, ctcbeam::kMidV, ctcbeam::kMidVc, 1, 0}; }
__host__ __device__ inline bool fits_mid_layout(const Dims &d) {
  return d.K <= ctcbeam::kMidK && d.V <= ctcbeam::kMidV && d.Vc_max <= ctcbeam::kMidVc && d.use_rank_table && !d.lm;
}

// LAYOUT 3 (round 6): the first wide-beam layout (BIG == 1) at a compile-time size -- beam <= kWideK over <= kWideV labels, no pruning, no
// scorer: BASELINE configs[2]'s decoder (beam 500 over the 29 labels of English characters; the largest beam whose slot keys still fit one
// workgroup's LDS).  The algorithm is the run-time layout's (beam_core.h SMALLV = 0); what changes is that every LDS array sits at an
// address the instructions can hold, as in LAYOUT 1.
constexpr int kWideK = 500, kWideV = 29;
__host__ __device__ constexpr Dims wide_layout_dims() { return Dims{kWideK, kWideV, kWideV, 0, 0}; }
__host__ __device__ inline bool fits_wide_layout(const Dims &d) { return d.K <= kWideK && d.V <= kWideV && d.Vc_max <= kWideV && !d.use_rank_table && !d.lm; }

// PRUNED: the candidates of every frame come from the vocabulary-prune pass (a.pr_*), otherwise they are the rows of a.probs.
// OCC2 (fixed layout only): the build for two workgroups per CU -- at most 64 VGPRs (8 waves per SIMD) and the exact
// replay's scratch in HBM (67 KB of LDS instead of 132 KB).  A lone workgroup runs ~7 % slower than in the default build
// (fewer registers, replay rounds in HBM), two per CU together 1.4-1.5x faster: the library launches this build for
// batches that outnumber the CUs and when the caller keeps several launches in flight (ctcd_set_cu_sharing).
// LM: 0 = no scorer, 1 = any scorer, 2 = word model over at most 64 labels (the character-model and wide-dictionary
// branches are not compiled in: beam_core.h WORDLM), 3 = any scorer behind the host-side hook (beam_core.h CB: the tables are a
// cache of a callback's answers; a miss parks the utterance).
template <int PROF, int BIG, int LAYOUT, bool PRUNED, int NT = 0, int LM = 0, bool OCC2 = false>
__global__ void __launch_bounds__(1024, OCC2 ? 8 : 1) ctc_beam_decode_kernel(KernelArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ uint64_t tbl[64];
  __shared__ int red[32];
  const int b = (int)blockIdx.x;
  if (threadIdx.x < 64) tbl[threadIdx.x] = a.tables[threadIdx.x];
  Work w;
  // (every layout keeps the exact replay's scratch in per-utterance HBM scratch; the wide-beam layouts more: beam_core.h carve)
  if (LAYOUT == 1) carve<0, OCC2>(w, smem, a.far + (size_t)b * a.far_stride, fixed_layout_dims(LM != 0), nullptr);
  else if (LAYOUT == 2) carve<0, OCC2>(w, smem, a.far + (size_t)b * a.far_stride, mid_layout_dims(), nullptr);
  else if (LAYOUT == 3) carve<BIG>(w, smem, a.far + (size_t)b * a.far_stride, wide_layout_dims(), nullptr);
  else carve<BIG>(w, smem, a.far + (size_t)b * a.far_stride, a.dims, nullptr);
  __shared__ long long prof[16];
  constexpr int kTlCap = (LM || BIG) ? kTimelineCap / 2 : kTimelineCap;  // (the LM tier's and the wide-beam layout's workspaces leave 8 KB for the stamps)
  __shared__ long long tlbuf[PROF == 2 ? 16 * kTlCap : 1];
  __shared__ int tlcnt[16];
  if (PROF == 2 && threadIdx.x < 16) tlcnt[threadIdx.x] = kTlCap;
  if (PROF == 2 && a.tl && b == 0)
    for (int i = threadIdx.x; i < 16 * kTlCap; i += blockDim.x) tlbuf[i] = 0;
  if (PROF == 1 && threadIdx.x < 16) prof[threadIdx.x] = 0;
  // PROF 4 / 5: the twins of builds 0 / 3 that take STREAMED input (a.frames_ready: the host-tensor entry point feeds the rows while the
  // kernel runs).  The north-star class's default builds do not: the polling code at the top of the frame loop and the registers it
  // holds across the loop cost the HBM-resident case 2.9 % (round 5: 5.24 -> 5.09 ms in tools/raw_multi.py; the word-model LM kernel
  // 10.94 -> 10.70 ms; the kernels sit at the scalar-register limit), so that case gets a build without them and the
  // streamed case keeps the build it had.
  constexpr int XP = PROF == 4 ? 0 : PROF == 5 ? 3 : PROF;
  // (every instantiation for which this holds is listed, with its twin, in ctcdecode_amd.hip streamed_input_twin: the launch code swaps them)
  constexpr bool kNoStreamedInput = (PROF == 0 || PROF == 3) && LAYOUT == 1 && NT == 1024 && (LM == 0 || LM == 2) && !PRUNED && BIG == 0 && (!OCC2 || LM == 0);
  DevX<XP, BIG != 0, NT> x{red, 0, prof, 0, (PROF == 1 && a.dbg && b == 0) ? a.dbg : nullptr, 1 + 4 * a.K,
                    (PROF == 2 && b == 0 && a.tl) ? tlbuf : nullptr, tlcnt, kTlCap, a.tl_f0, a.tl_nf};
  int len = a.seq_lens ? __builtin_amdgcn_readfirstlane(a.seq_lens[b]) : a.T;
  len = len < 0 ? 0 : (len > a.T ? a.T : len);  // binding.cpp:64-65
  __syncthreads();
  if (PROF == 1) x.last = (long long)wall_clock64();
  const size_t f0 = a.frame_off ? (size_t)__builtin_amdgcn_readfirstlane(a.frame_off[b]) : 0;  // (resumed launches of the scorer hook)
  PrunedRows prow;
  if (PRUNED) {
    prow.cnt = a.pr_cnt + (size_t)b * a.T + f0;
    prow.ch = a.pr_ch + ((size_t)b * a.T + f0) * a.pr_stride;
    prow.lp = a.pr_lp + ((size_t)b * a.T + f0) * a.pr_stride;
    prow.stride = a.pr_stride;
  }
  PoolNode *pool = a.pool + (size_t)b * a.pool_stride;
  int *pool_up = a.pool_up + (size_t)b * 2 * a.pool_stride;  // [express pointers | time steps' high parts] of this utterance
  int pool_cap = (int)a.pool_stride;
  StreamState ss;
  if (a.st_base) {
    char *base = a.st_base[b];
    ss.hdr = (int *)base;
    ss.arrays = ss.hdr + SH_WORDS;
    ss.finish = a.st_eos[b];
    pool = (PoolNode *)(base + a.st_pool_off);
    pool_cap = a.st_poolcap[b];
    pool_up = (int *)(pool + pool_cap);
  }
  // the result pointers stay in the kernel-argument segment until the utterance ends (finish() reads them from there)
#if defined(__HIP_DEVICE_COMPILE__)
  const OutRefs *outs = (const OutRefs *)((const char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(KernelArgs, outs));
#else
  const OutRefs *outs = &a.outs;
#endif
  const ctclm::LmView *lmv = nullptr;
#if defined(__HIP_DEVICE_COMPILE__)
  if (LM) lmv = (const ctclm::LmView *)((const char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(KernelArgs, lm));
#else
  if (LM) lmv = &a.lm;
#endif
  int st;
  if (LM == 3) {
    // (the scorer hook's builds: the call sits in a loop -- see KernelArgs::cb_log_len)
    __shared__ unsigned cb_n;
    size_t fo = f0;
    unsigned log_pos = 0, q_total = 0;
    for (;;) {
      const int before = a.st_base ? __builtin_amdgcn_readfirstlane(__hip_atomic_load(&ss.hdr[SH_FRAMES], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0;
      if (PRUNED) {
        prow.cnt = a.pr_cnt + (size_t)b * a.T + fo;
        prow.ch = a.pr_ch + ((size_t)b * a.T + fo) * a.pr_stride;
        prow.lp = a.pr_lp + ((size_t)b * a.T + fo) * a.pr_stride;
      }
      st = decode_utterance<!PRUNED, (LAYOUT == 3 ? 0 : LAYOUT), true, BIG != 0, BIG != 0 || OCC2, BIG == 3, false, true>(x, w, a.dims, a.blank, PRUNED ? nullptr : a.probs + ((size_t)b * a.T + fo) * a.V,
                                  PRUNED ? &prow : (const PrunedRows *)nullptr, len, pool, pool_up, pool_cap, tbl, outs, b,
                                  a.st_base ? &ss : (const StreamState *)nullptr, lmv, a.raw + ((size_t)b * a.T + fo) * a.V, a.raw_log, (const int *)nullptr);
      if (st != ctcbeam::ST_NEED_HOST || a.cb_log_len == nullptr || !a.st_base) break;
      __threadfence_system();  // every lane's queued pairs are on their way to the host before the workgroup says it waits
      __syncthreads();         // (and the parked state is complete)
      const unsigned asked = (unsigned)w.vars[ctcbeam::VAR_LMQ];
      if (asked == 0) break;  // (parked without a question: the host reports it)
      q_total += asked;
      if (threadIdx.x == 0) {
        // wait until the host has dealt with every pair this utterance queued (their slots are in the log by then)
        unsigned n = 0;
        for (int spins = 0;; ++spins) {
          const unsigned ans = __hip_atomic_load(&a.cb_ans[b], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
          n = __hip_atomic_load(a.cb_log_len, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
          if (n == 0xFFFFFFFFu || ans >= q_total) break;
          if (spins > (1 << 21)) { n = 0xFFFFFFFFu; break; }  // (~2 s without an answer: leave, the host will relaunch)
          __builtin_amdgcn_s_sleep(8);
        }
        cb_n = n;
      }
      __syncthreads();
      const unsigned n = cb_n;
      if (n == 0xFFFFFFFFu || n < log_pos) break;
      ctclm::NgSlot *ng = const_cast<ctclm::NgSlot *>(a.lm.ng);
      for (unsigned i = log_pos + threadIdx.x; i < n; i += blockDim.x) ng[a.cb_log_idx[i]] = a.cb_log_slot[i];
      log_pos = n;
      __threadfence();  // (release + acquire at device scope: the table entries are written, and nothing read from here on -- the table, the
                        //  parked state -- comes from a line this CU cached before)
      const int consumed = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&ss.hdr[SH_FRAMES], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) - before;
      __syncthreads();
      fo += (size_t)consumed;
      len -= consumed;
    }
  } else {
  st = decode_utterance<!PRUNED, (LAYOUT == 3 ? 0 : LAYOUT), LM != 0, BIG != 0, BIG != 0 || OCC2, BIG == 3, LM == 2, LM == 3>(x, w, a.dims, a.blank, PRUNED ? nullptr : a.probs + ((size_t)b * a.T + f0) * a.V,
                                  PRUNED ? &prow : (const PrunedRows *)nullptr, len, pool, pool_up, pool_cap, tbl, outs, b,
                                  a.st_base ? &ss : (const StreamState *)nullptr, lmv, LM ? a.raw + ((size_t)b * a.T + f0) * a.V : nullptr, a.raw_log,
                                  (PRUNED || kNoStreamedInput) ? (const int *)nullptr : a.frames_ready);
  }
  if (threadIdx.x == 0) {
    if (a.shape) a.shape[b] = 16 * w.vars[ctcbeam::VAR_QSTAT];
    a.status[b] = st;
    if (a.frames_done && a.st_base) a.frames_done[b] = ss.hdr[SH_FRAMES];  // (written by this thread in save_state)
    if (LM == 3 && a.cb_done) {  // (the host's service loop ends when every workgroup has reported)
      __threadfence_system();
      __hip_atomic_store(&a.cb_done[b], 1 + st, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (PROF == 2 && a.tl && b == 0) {
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * kTlCap; i += blockDim.x) a.tl[i] = tlbuf[i];  // (LM build: [16][kTimelineCap / 2])
  }
  if (PROF == 1 && threadIdx.x < 16) a.prof[(size_t)b * 16 + threadIdx.x] = prof[threadIdx.x];
}

// Every instantiation the library launches: X(PROF, BIG, LAYOUT, PRUNED, NT, LM, OCC2, translation-unit group).
// (Groups 0-3 hold the headline families; new instantiations go elsewhere so that those translation units stay as they are.)
// decode_kernels.hip instantiates the ones of its group (-DCTC_KERNEL_GROUP=g); ctcdecode_amd.hip declares them all extern.
#define CTC_KERNEL_GROUPS 12
#if defined(CTC_QUICK_BUILD) && CTC_QUICK_BUILD == 2  // experiment builds of the LM tier: its north-star class kernel and the timeline twin
#define CTC_KERNEL_LIST(X) X(0, 0, 1, false, 1024, 2, false, 0) X(2, 0, 1, false, 1024, 2, false, 1) X(0, 0, 1, false, 1024, true, false, 1) X(2, 0, 1, false, 1024, true, false, 0) X(4, 0, 1, false, 1024, 2, false, 1)
#elif defined(CTC_QUICK_BUILD) && CTC_QUICK_BUILD == 3  // workgroup-size sweep: the north-star class kernel with CTC_QUICK_NT threads folded in
#define CTC_KERNEL_LIST(X) X(0, 0, 1, false, CTC_QUICK_NT, false, false, 0)
#elif defined(CTC_QUICK_BUILD) && CTC_QUICK_BUILD == 4  // experiment builds of the pruned classes: compile-time layouts 1 and 2, and the run-time layout they replace
#define CTC_KERNEL_LIST(X) X(0, 0, 2, true, 1024, false, false, 0) X(0, 0, 1, true, 1024, false, false, 1) X(0, 0, 0, true, 0, false, false, 1) X(0, 0, 0, false, 0, false, false, 0)
#elif defined(CTC_QUICK_BUILD)  // experiment builds: the north-star class kernels and the barrier-timeline twin only
#define CTC_KERNEL_LIST(X) \
  X(0, 0, 1, false, 1024, false, false, 0) X(2, 0, 1, false, 1024, false, false, 1) X(0, 0, 1, false, 1024, false, true, 1) X(4, 0, 1, false, 1024, false, false, 1) X(4, 0, 1, false, 1024, false, true, 0)
#else
#define CTC_KERNEL_LIST(X)                                                                                                          \
  X(0, 0, 1, false, 1024, false, false, 0) X(0, 0, 1, true, 1024, false, false, 1) X(2, 0, 1, false, 1024, false, false, 2)               \
  X(0, 1, 0, false, 0, false, false, 3) X(0, 1, 0, true, 0, false, false, 4) X(0, 2, 0, false, 0, false, false, 5) X(0, 2, 0, true, 0, false, false, 5) \
  X(0, 0, 0, false, 0, false, false, 6) X(0, 0, 0, true, 0, false, false, 6) X(0, 0, 1, false, 0, false, false, 7) X(0, 0, 1, true, 0, false, false, 7) \
  X(1, 1, 0, false, 0, false, false, 8) X(1, 1, 0, true, 0, false, false, 8) X(1, 0, 1, false, 0, false, false, 9) X(1, 0, 1, true, 0, false, false, 9) \
  X(1, 0, 0, false, 0, false, false, 10) X(1, 0, 0, true, 0, false, false, 10) X(0, 1, 3, false, 1024, false, false, 10)                                                       \
  X(0, 0, 0, false, 0, true, false, 11) X(0, 0, 0, true, 0, true, false, 11) X(0, 0, 1, false, 1024, true, false, 2)                     \
  X(0, 0, 1, true, 1024, true, false, 3) X(2, 0, 1, false, 1024, true, false, 4)                                                       \
  X(0, 0, 1, false, 1024, 2, false, 0) X(0, 0, 1, true, 1024, 2, false, 1) X(0, 0, 1, false, 1024, 2, true, 2) X(0, 0, 1, true, 1024, 2, true, 3) X(2, 0, 1, false, 1024, 2, false, 5) \
  X(0, 0, 1, false, 1024, false, true, 0) X(0, 0, 1, true, 1024, false, true, 1) X(0, 0, 1, false, 1024, true, true, 10) X(0, 0, 1, true, 1024, true, true, 11) \
  X(0, 1, 0, false, 0, true, false, 6) X(0, 1, 0, true, 0, true, false, 7) X(0, 2, 0, false, 0, true, false, 8) X(0, 2, 0, true, 0, true, false, 9) \
  X(0, 3, 0, false, 0, false, false, 5) X(0, 3, 0, true, 0, false, false, 4) X(0, 3, 0, false, 0, true, false, 6) X(0, 3, 0, true, 0, true, false, 7) \
  X(0, 1, 0, false, 1024, false, false, 9) X(0, 1, 0, true, 1024, false, false, 10) X(2, 1, 0, false, 1024, false, false, 11) \
  X(4, 0, 1, false, 1024, false, false, 3) X(5, 0, 1, false, 1024, false, false, 5) X(4, 0, 1, false, 1024, 2, false, 7) X(4, 0, 1, false, 1024, false, true, 9) \
  X(0, 0, 2, true, 1024, false, false, 0) \
  X(3, 0, 1, false, 1024, false, false, 1) X(3, 0, 1, true, 1024, false, false, 2) X(0, 0, 1, false, 1024, 3, false, 8) X(0, 0, 1, true, 1024, 3, false, 9) X(0, 0, 0, false, 0, 3, false, 10) X(0, 0, 0, true, 0, 3, false, 4) \
  X(0, 1, 0, false, 0, 3, false, 6) X(0, 1, 0, true, 0, 3, false, 7) X(0, 2, 0, false, 0, 3, false, 8) X(0, 2, 0, true, 0, 3, false, 9) X(0, 3, 0, false, 0, 3, false, 11) X(0, 3, 0, true, 0, 3, false, 5)
#endif

}  // namespace ctcdk
