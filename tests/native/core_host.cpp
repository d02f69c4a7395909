// TEST INFRASTRUCTURE ONLY: the GPU decoder's per-utterance core (ctcdecode_amd/csrc/beam_core.h) instantiated with a
// single sequential "thread", behind the same C signature as the oracle, so that the algorithm (DFS-ordered beam +
// LCP array, Euler-tour slots, exact tie handling) is differential-tested on the CPU before it ever runs on a GPU.
// The product never links this file.
#define CTC_EXACT_MATH_HOST_TABLES
#include "../../ctcdecode_amd/csrc/beam_core.h"
#include "../../ctcdecode_amd/csrc/lm_build.h"
#include "../../ctcdecode_amd/csrc/lm_callback.h"
#include "../../ctcdecode_amd/csrc/compact_results.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <limits>
#include <thread>
#include <vector>

namespace {

struct HostX {
  static constexpr bool kZeroKeyTail = false;  // (the listing pass below tests its bounds)
  bool far_ = false;  // the workspace layout under test keeps the exact-replay arrays "far" (CTC_HOST_BIG)
  bool far() const { return far_; }
  int tid() const { return 0; }
  int tid_fresh() const { return 0; }
  void wave_lds_fence() const {}
  int nt() const { return 1; }
  constexpr bool nt_is(int) const { return false; }
  void sync() {}
  void sync_full() {}
  int uni(int v) const { return v; }
  void uni4(const int *p, int *out) const { out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; out[3] = p[3]; }
  ctcbeam::Int4v load4(const int *p) const { return ctcbeam::Int4v{p[0], p[1], p[2], p[3]}; }
  void uni4v(const ctcbeam::Int4v &v, int *out) const { out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w; }
  int group() const { return 0; }
  int ngroups() const { return 1; }
  int lane() const { return 0; }
  int lanes() const { return 1; }
  unsigned long long ballot(bool p) const { return p ? 1ull : 0ull; }
  int first_below(const int *arr, int from, int n, int bound) const {
    int q = from;
    while (q < n && arr[q] >= bound) ++q;
    return q;
  }
  bool subtrees_by_quarters(unsigned long long, int, int, int, const int *, const int *, int, int *, int *, int *) { return false; }  // (one lane: nothing to split)
  void atomic_max(int *p, int v) { *p = std::max(*p, v); }
  float unif(float v) const { return v; }
  int pick(int v, int) const { return v; }  // the value lane `idx` holds (one lane here)
  void mark(int) {}
  void trace_frame(int) {}
  template <int P> void prio() const {}
  template <class P> const P *fresh(const P *p) const { return p; }
  // event statistics (beam_core.h Event): summed over every decode of the process, read by ctccore_event_counts()
  static long long *counters() { static long long c[ctcbeam::EV_COUNT]; return c; }
  static int &cur_frame() { static int t = 0; return t; }
  void count(int k, int v) const {
    counters()[k] += v;
    // CTC_HOST_TRACE_EXACT=1: print the frames that replay std::nth_element (finding a tie frame for tools/barrier_timeline.py)
    static const bool trace = getenv("CTC_HOST_TRACE_EXACT") != nullptr;
    if (trace && k == ctcbeam::EV_EXACT) fprintf(stderr, "exact replay at frame %d\n", cur_frame());
  }
  void tick() {}
  // speculative select (beam_core.h Decoder::kSpec), sequentially: the same contract as the device policy's
  static constexpr bool kSpecSelect = true;
#if defined(CTC_NO_RANK_EPOCH)
  static constexpr bool kRankEpoch = false;
#else
  static constexpr bool kRankEpoch = true;
#endif
#if defined(CTC_NO_PARENT_REC)
  static constexpr bool kParentRec = false;
#else
  static constexpr bool kParentRec = true;
#endif
#if defined(CTC_EXP_SPEC_LM)  // (measured slower, round 6: beam_core.h kSpec)
  static constexpr bool kSpecLm = true;
#else
  static constexpr bool kSpecLm = false;
#endif
  static constexpr bool kQuarters = false;
  static constexpr bool kLcpTable = false;   // (a wave of its own builds it beside phase B)
  static constexpr bool kA1Overlap = false;  // (an overlap of wave roles: nothing to overlap with one thread)
  static constexpr bool kLmOverlap = false;  // (a split of the workgroup's waves: nothing to overlap with one thread)
  bool spec_fits(int) const { return true; }
  void hot_append(bool hot, uint32_t key, int slot, uint32_t *hk, int *hs, int *cnt) {
    if (!hot) return;
    const int p = (*cnt)++;
    if (p < ctcbeam::kHotCap) { hk[p] = key; hs[p] = slot; }
  }
  using HotTicket = int *;
  HotTicket hot_issue(bool, int *cnt) { return cnt; }
  void hot_commit(HotTicket cnt, bool hot, uint32_t key, int slot, uint32_t *hk, int *hs) { hot_append(hot, key, slot, hk, hs, cnt); }
  void row_max_store(int *dst, float v, int V) const { if (V >= 1) memcpy(dst, &v, 4); }  // (one lane: a one-label row)
  int spec_thread() const { return 0; }
  void hot_append_wave(bool hot, uint32_t key, int slot, uint32_t *hk, int *hs, int *cnt) { hot_append(hot, key, slot, hk, hs, cnt); }
  struct SpecPre {};
  SpecPre spec_pre(const uint32_t *, const int *) const { return SpecPre{}; }
  struct SpecResult { uint32_t tau; int ok; };
  SpecResult spec_select(const SpecPre &, int H, int K, const uint32_t *hk, const int *hs, uint32_t *bitmap, int *scratch, int S, int *surv, int *res) {
    (void)scratch; (void)S; (void)res;
    SpecResult r{0u, 0};
    if (getenv("CTC_HOST_NO_SPEC")) return r;
    std::vector<int> keep;
    for (int q = 0; q < H; ++q) {
      int ge = 0;
      for (int k = 0; k < H; ++k) ge += hk[k] >= hk[q];
      if (ge <= K) { keep.push_back(hs[q]); bitmap[hs[q] >> 5] |= 1u << (hs[q] & 31); }
      if (ge == K) { r.tau = hk[q]; r.ok = 1; }
    }
    if (!r.ok) return r;  // equal keys straddle the boundary
    std::sort(keep.begin(), keep.end());
    for (int k = 0; k < K; ++k) surv[k] = keep[k];
    return r;
  }
  // CTC_DUMP_KEYS=<file>: every frame's slot keys, for offline studies of the select
  void probe_keys(int t, int S, const uint32_t *skey, int K, uint32_t maxkey, const float *clp, int Vc) const {
    cur_frame() = t;
    static FILE *f = getenv("CTC_DUMP_KEYS") ? fopen(getenv("CTC_DUMP_KEYS"), "wb") : nullptr;
    if (!f) return;
    float mx = clp[0];
    for (int i = 1; i < Vc; ++i) mx = std::max(mx, clp[i]);
    uint32_t hdr[5] = {(uint32_t)t, (uint32_t)S, (uint32_t)K, maxkey, 0u};
    memcpy(&hdr[4], &mx, 4);
    fwrite(hdr, 4, 5, f);
    fwrite(skey, 4, (size_t)S, f);
    fflush(f);
  }
  void dump(int, int, const int *, const int *, const int *, const float *) {}
  uint32_t scan_excl(uint32_t *a, int n) {
    uint32_t run = 0;
    for (int i = 0; i < n; ++i) {
      uint32_t v = a[i];
      a[i] = run;
      run += v;
    }
    return run;
  }
  int atomic_add(int *p, int v) { int o = *p; *p += v; return o; }
  void atomic_or(uint32_t *p, uint32_t v) { *p |= v; }
  void mark_ge(int S, const uint32_t *skey, uint32_t tau, uint32_t *bitmap) {
    for (int wd = 0; wd < 2 * ((S + 63) / 64); ++wd) bitmap[wd] = 0u;
    for (int s = 0; s < S; ++s)
      if (skey[s] >= tau) bitmap[s >> 5] |= 1u << (s & 31);
  }
  uint32_t bitsel(uint32_t mask, uint32_t a, uint32_t b) const { return (a & mask) | (b & ~mask); }
  int sum8(int v) const { return v; }
  int sum4(int v) const { return v; }
  void wave_add_flag(int *p, bool f) { *p += f ? 1 : 0; }
  void fence_system() const {}
  int load_system(const int *p) const { return *p; }
  void nap() const {}
  void store_system(int32_t *p, int v) const { *p = v; }
  void wave_add(int *p, int v) { *p += v; }
  void find_bucket(int *bins, int need, int *out) {
    using ctcbeam::kBins;
    int run = 0, bstar = -1, above = 0, inb = 0, total = 0;
    for (int b = 0; b < kBins; ++b) total += bins[b];
    for (int b = kBins - 1; b >= 0; --b) {
      const int v = bins[b];
      if (run + v >= need) { bstar = b; above = run; inb = v; break; }
      run += v;
    }
    out[0] = bstar; out[1] = above; out[2] = total; out[3] = inb;
  }
  template <class Pred>
  void mark_slots(int S, uint32_t *bitmap, Pred pred) {
    for (int wd = 0; wd < 2 * ((S + 63) / 64); ++wd) bitmap[wd] = 0;
    for (int s = 0; s < S; ++s)
      if (pred(s)) bitmap[s >> 5] |= 1u << (s & 31);
  }
  template <bool TZ = false, bool COMPACT = false>
  void list_bucket(int S, const uint32_t *skey, uint32_t b32, uint32_t bspan, bool direct, uint32_t *bitmap, uint32_t *list, int *lslot,
                   int *lcount) {
    for (int wd = 0; wd < 2 * ((S + 63) / 64); ++wd) bitmap[wd] = 0;
    for (int s = 0; s < S; ++s) {
      const uint32_t k = skey[s];
      if (k < b32) continue;
      const uint32_t dk = k - b32;
      if (dk > bspan) {
        if (direct) bitmap[s >> 5] |= 1u << (s & 31);
        continue;
      }
      const int li = (*lcount)++;
      list[li] = dk + 1u;
      lslot[li] = s;
    }
  }
  template <bool CLUSTERED = false>
  void expand_bitmap(const uint32_t *bitmap, int nwords64, int *out) {
    int k = 0;
    for (int s = 0; s < nwords64 * 64; ++s)
      if ((bitmap[s >> 5] >> (s & 31)) & 1u) out[k++] = s;
  }
  template <class Pred>
  void compact_slots(int S, int *out, Pred pred) {
    int k = 0;
    for (int s = 0; s < S; ++s)
      if (pred(s)) out[k++] = s;
  }
  template <class Pred, class Emit>
  void compact_slots_to(int S, Pred pred, Emit emit) {
    int k = 0;
    for (int s = 0; s < S; ++s)
      if (pred(s)) emit(k++, s);
  }
  void block_scan_u32(uint32_t mine, uint32_t *base_out, uint32_t *total_out) { *base_out = 0; *total_out = mine; }
  int count(unsigned long long m) const { return (int)m; }
  int count_below(unsigned long long) const { return 0; }
  uint32_t first_lane(uint32_t v) const { return v; }
  void wave_max_to(int *p, uint32_t v) { *p = (int)std::max((uint32_t)*p, v); }
  void wave_min_to(int *p, uint32_t v) { *p = (int)std::min((uint32_t)*p, v); }
  unsigned global_add(unsigned *p, unsigned v) { unsigned o = *p; *p += v; return o; }
  int item_ = 0;
  int item() const { return item_; }
};

// Vocabulary pruning exactly as the reference does it (decoder_utils.cpp:10-45) -- host stand-in for the GPU prune pass.
template <class T>
T log_add(T a, T b) {
  const T floor_v = -std::numeric_limits<T>::max();
  if (a <= floor_v) return b;
  if (b <= floor_v) return a;
  T top = std::max(a, b);
  return std::log(std::exp(a - top) + std::exp(b - top)) + top;
}

void prune_row(const float *row, int V, double cutoff_prob, int top_n, int *cnt, int *ch, float *lp) {
  std::vector<std::pair<int, double>> pv;
  for (int i = 0; i < V; ++i) pv.emplace_back(i, (double)row[i]);
  size_t keep = (size_t)V;
  std::sort(pv.begin(), pv.end(), [](const std::pair<int, double> &a, const std::pair<int, double> &b) { return a.second > b.second; });
  if (std::log(cutoff_prob) < 0.0) {
    double cum = 0.0;
    keep = 0;
    for (size_t i = 0; i < pv.size(); ++i) {
      cum = log_add(cum, pv[i].second);
      ++keep;
      if (cum >= cutoff_prob || keep >= (size_t)top_n) break;
    }
  } else {
    keep = (size_t)top_n;
  }
  *cnt = (int)keep;
  for (size_t i = 0; i < keep; ++i) {
    ch[i] = pv[i].first;
    lp[i] = (float)pv[i].second;
  }
}

}  // namespace

static int decode_impl(const float *probs, const int32_t *seq_lens, int B, int T, int V, int beam, int num_threads,
                       double cutoff_prob, int cutoff_top_n, int blank_id, int32_t *out_tokens,
                       int32_t *out_timesteps, float *out_scores, int32_t *out_lens, int32_t *n_results,
                       const ctclm::LmView *lm, const float *raw, int raw_log);

// Host twin of the product's log_softmax pre-pass (ctcdecode_amd.hip log_softmax_rows_kernel; input mode 2 / ctcd_log_softmax):
// the same float32 operations in the same order, with the C library's expf / logf.
// event statistics since the process started (or since the last call with reset != 0): out[ctcbeam::EV_COUNT].  Decodes that run on
// several host threads add up racily -- use num_threads = 1 for exact counts.
extern "C" int ctccore_event_counts(long long *out, int reset) {
  for (int i = 0; i < ctcbeam::EV_COUNT; ++i) { out[i] = HostX::counters()[i]; if (reset) HostX::counters()[i] = 0; }
  return ctcbeam::EV_COUNT;
}

extern "C" void ctccore_log_softmax_rows(const float *x, long long rows, int V, float *out) {
  for (long long r = 0; r < rows; ++r) {
    const float *xr = x + (size_t)r * V;
    float *yr = out + (size_t)r * V;
    float m = -INFINITY;
    for (int j = 0; j < V; ++j) m = xr[j] > m ? xr[j] : m;
    m += 0.0f;  // (a zero maximum is +0)
    if (!(m > -INFINITY)) {
      for (int j = 0; j < V; ++j) yr[j] = -INFINITY;
      continue;
    }
    float part[64];
    for (int l = 0; l < 64; ++l) {
      float p = 0.0f;
      for (int j = l; j < V; j += 64) {
        const float d = xr[j] - m;
        p += d < -88.0f ? 0.0f : std::exp(d);
      }
      part[l] = p;
    }
    for (int off = 1; off < 64; off <<= 1) {
      float nxt[64];
      for (int l = 0; l < 64; ++l) nxt[l] = part[l] + part[l ^ off];
      std::memcpy(part, nxt, sizeof(part));
    }
    const float ls = std::log(part[0]);
    for (int j = 0; j < V; ++j) yr[j] = (xr[j] - m) - ls;
  }
}

// log-probability input only (the prob->log conversion is a separate, elementwise stage of the product).
extern "C" int ctccore_decode_f32(const float *probs, const int32_t *seq_lens, int B, int T, int V, int beam, int num_threads,
                                  double cutoff_prob, int cutoff_top_n, int blank_id, int32_t *out_tokens,
                                  int32_t *out_timesteps, float *out_scores, int32_t *out_lens, int32_t *n_results) {
  return decode_impl(probs, seq_lens, B, T, V, beam, num_threads, cutoff_prob, cutoff_top_n, blank_id, out_tokens, out_timesteps,
                     out_scores, out_lens, n_results, nullptr, nullptr, 1);
}

// LM tier: the scorer is built by the product's own host code (lm_build.h); labels = V NUL-terminated strings.
// log_input == 0: `probs` are probabilities (converted here the way the product's pre-pass does, decoder_utils.cpp:42).
extern "C" int ctccore_decode_lm_f32(const float *probs, const int32_t *seq_lens, int B, int T, int V, int beam, int num_threads,
                                     double cutoff_prob, int cutoff_top_n, int blank_id, int log_input, double alpha, double beta,
                                     const char *lm_path, const char *labels, int32_t *out_tokens, int32_t *out_timesteps,
                                     float *out_scores, int32_t *out_lens, int32_t *n_results, int32_t *meta3) {
  std::vector<std::string> lab(V);
  for (int i = 0; i < V; ++i) {
    lab[i] = labels;
    labels += lab[i].size() + 1;
  }
  ctclm::HostScorer hs;
  if (!hs.build(alpha, beta, lm_path, lab)) return -100;
  if (meta3) { meta3[0] = hs.char_based; meta3[1] = hs.order; meta3[2] = hs.dict_size; }
  const ctclm::LmView view = hs.view();
  std::vector<float> logp;
  const float *in = probs;
  if (!log_input) {
    logp.resize((size_t)B * T * V);
    for (size_t i = 0; i < logp.size(); ++i) logp[i] = (float)std::log((double)probs[i] + (double)std::numeric_limits<float>::min());
    in = logp.data();
  }
  return decode_impl(in, seq_lens, B, T, V, beam, num_threads, cutoff_prob, cutoff_top_n, blank_id, out_tokens, out_timesteps,
                     out_scores, out_lens, n_results, &view, probs, log_input);
}

// The host-side scorer hook (lm_callback.h) driven the way the product drives it: the decode runs against a CACHE of a
// callback's answers; an utterance that asks for something the cache does not hold is parked in front of the frame it was in
// (ST_NEED_HOST), the queued pairs are answered by the callback, and the utterance resumes from its parked state.  The
// callback here asks the built-in ARPA tables (HostScorer::cond_log10) -- one implementation of the interface -- so the
// results must equal ctccore_decode_lm_f32's bit for bit.  stats2: {callback calls, resumptions}.
extern "C" int ctccore_decode_lm_cb_f32(const float *probs, const int32_t *seq_lens, int B, int T, int V, int beam, int blank_id, int log_input,
                                        double alpha, double beta, const char *lm_path, const char *labels, int32_t *out_tokens,
                                        int32_t *out_timesteps, float *out_scores, int32_t *out_lens, int32_t *n_results, long long *stats2) {
  using namespace ctcbeam;
  std::vector<std::string> lab(V);
  for (int i = 0; i < V; ++i) { lab[i] = labels; labels += lab[i].size() + 1; }
  ctclm::HostScorer ref;
  if (!ref.build(alpha, beta, lm_path, lab)) return -100;
  std::vector<std::string> vocabulary;
  for (const std::string &w : ref.vocab)
    if (w != "<unk>") vocabulary.push_back(w);
  ctclm::CallbackLm cb;
  auto fn = [](void *user, const char *const *words, int n, float *p10) -> int {
    const ctclm::HostScorer *r = (const ctclm::HostScorer *)user;
    std::vector<std::string> ws(words, words + n);
    return r->cond_log10(ws, p10);
  };
  if (!cb.build(alpha, beta, ref.order, vocabulary, lab, fn, &ref)) return -101;
  if (cb.hs.char_based != ref.char_based || cb.hs.dict_size != ref.dict_size) return -102;
  std::vector<float> logp;
  const float *in = probs;
  if (!log_input) {
    logp.resize((size_t)B * T * V);
    for (size_t i = 0; i < logp.size(); ++i) logp[i] = (float)std::log((double)probs[i] + (double)std::numeric_limits<float>::min());
    in = logp.data();
  }
  Dims d;
  d.K = beam; d.V = V; d.Vc_max = V; d.use_rank_table = 0; d.lm = 1;
  Work w;
  size_t far_bytes = 0;
  // CTC_HOST_BIG=1|2|3: the wide-beam layouts behind the hook as well (round 6: the hook's kernels exist for them)
  const int flevel = getenv("CTC_HOST_BIG") ? getenv("CTC_HOST_BIG")[0] - '0' : 0;
  std::vector<char> mem((flevel == 3 ? carve<3>(w, nullptr, nullptr, d, &far_bytes) : flevel == 2 ? carve<2>(w, nullptr, nullptr, d, &far_bytes)
                         : flevel == 1 ? carve<1>(w, nullptr, nullptr, d, &far_bytes) : carve<0>(w, nullptr, nullptr, d, &far_bytes)) + 64);
  std::vector<char> far(far_bytes + 64);
  std::vector<ctclm::MissEntry> miss(65536);
  unsigned nmiss = 0;
  long long resumes = 0;
  const bool wordlm = !cb.hs.char_based && !cb.hs.dict_wide && !getenv("CTC_HOST_GENERAL_LM");
  for (int b = 0; b < B; ++b) {
    std::vector<int> hdr(SH_WORDS, 0), arrays((size_t)kStateArraysLm * beam, 0);
    std::vector<PoolNode> pool((size_t)1 + (size_t)beam * T);
    std::vector<int> pool_up(2 * pool.size());
    int len = seq_lens ? seq_lens[b] : T;
    len = std::max(0, std::min(len, T));
    for (int guard = 0;; ++guard) {
      if (guard > 4 * T + 64) return -103;  // (every resumption consumes a frame or answers a query: this cannot loop)
      ctclm::LmView view = cb.hs.view();
      view.cb = 1; view.cb_miss = miss.data(); view.cb_count = &nmiss; view.cb_cap = (uint32_t)miss.size(); view.cb_ring = 0;
      nmiss = 0;
      const int done = hdr[SH_FRAMES];
      if (flevel == 3) carve<3>(w, mem.data(), far.data(), d, nullptr);
      else if (flevel == 2) carve<2>(w, mem.data(), far.data(), d, nullptr);
      else if (flevel == 1) carve<1>(w, mem.data(), far.data(), d, nullptr);
      else carve<0>(w, mem.data(), far.data(), d, nullptr);
      HostX x;
      x.item_ = b;
      x.far_ = flevel != 0;
      StreamState ss{hdr.data(), arrays.data(), 1};
      const OutRefs outs{out_tokens, out_timesteps, out_scores, out_lens, n_results, beam, T, nullptr, nullptr, nullptr, nullptr, 0u, nullptr, nullptr, nullptr, nullptr, 0u};
      const float *rows = in + ((size_t)b * T + done) * V, *raw = probs + ((size_t)b * T + done) * V;
      int st;
      if (flevel == 3) st = decode_utterance<true, false, true, true, true, true, false, true>(x, w, d, blank_id, rows, (const PrunedRows *)nullptr, len - done, pool.data(), pool_up.data(),
                                     (int)pool.size(), ctcmath::host_tables().w, &outs, b, &ss, &view, raw, log_input);
      else if (flevel) st = decode_utterance<true, false, true, true, true, false, false, true>(x, w, d, blank_id, rows, (const PrunedRows *)nullptr, len - done, pool.data(), pool_up.data(),
                                     (int)pool.size(), ctcmath::host_tables().w, &outs, b, &ss, &view, raw, log_input);
      else if (wordlm) st = decode_utterance<true, false, true, false, false, false, true, true>(x, w, d, blank_id, rows, (const PrunedRows *)nullptr, len - done, pool.data(), pool_up.data(),
                                     (int)pool.size(), ctcmath::host_tables().w, &outs, b, &ss, &view, raw, log_input);
      else st = decode_utterance<true, false, true, false, false, false, false, true>(x, w, d, blank_id, rows, (const PrunedRows *)nullptr, len - done, pool.data(), pool_up.data(),
                                     (int)pool.size(), ctcmath::host_tables().w, &outs, b, &ss, &view, raw, log_input);
      if (st == ST_OK) break;
      if (st != ST_NEED_HOST) return -st;
      if (nmiss == 0 || nmiss > view.cb_cap) return -104;
      for (unsigned i = 0; i < nmiss; ++i) {
        if ((int)miss[i].item != b || miss[i].flag != 1u) return -106;
        if (!cb.resolve(miss[i].state, miss[i].word)) return -105;
      }
      ++resumes;
    }
  }
  if (stats2) { stats2[0] = (long long)cb.queries; stats2[1] = resumes; }
  return 1;
}

// Compact result delivery: decode every item into the compact form (one shared label buffer, bump-allocated), then expand
// it with the product's host expansion.  Must reproduce the padded results exactly; *labels_used reports the sharing.
extern "C" int ctccore_decode_compact_f32(const float *probs, const int32_t *seq_lens, int B, int T, int V, int beam, int blank_id,
                                          int32_t *out_tokens, int32_t *out_timesteps, float *out_scores, int32_t *out_lens,
                                          int32_t *n_results, long long *labels_used) {
  using namespace ctcbeam;
  Dims d;
  d.K = beam; d.V = V; d.Vc_max = V; d.use_rank_table = 0; d.lm = 0;
  Work w;
  size_t far_bytes = 0;
  std::vector<char> mem(carve<0>(w, nullptr, nullptr, d, &far_bytes) + 64);
  std::vector<char> far(far_bytes + 64);  // (the exact replay's scratch: HBM on the device)
  std::vector<int32_t> hdr((size_t)B * 4, 0), ent((size_t)B * beam * 4, 0);
  std::vector<uint32_t> rag((size_t)B * beam * T + 1);
  // the host mirrors a finished utterance copies its results into (OutRefs::m_*): the expansion below reads THEM
  std::vector<int32_t> m_hdr((size_t)B * 4, -7), m_ent((size_t)B * beam * 4, -7), m_done((size_t)B, 0);
  std::vector<uint32_t> m_rag(rag.size(), 0xDEADBEEFu);
  unsigned count = 0;
  for (int b = 0; b < B; ++b) {
    std::vector<PoolNode> pool((size_t)1 + (size_t)beam * T);
    std::vector<int> pool_up(2 * pool.size());  // express pointers | time steps' high parts
    int len = seq_lens ? seq_lens[b] : T;
    len = std::max(0, std::min(len, T));
    carve<0>(w, mem.data(), far.data(), d, nullptr);
    HostX x;
    const OutRefs outs{nullptr, nullptr, out_scores, out_lens, n_results, beam, T, hdr.data(), ent.data(), rag.data(), &count, (unsigned)(rag.size() - 1),
                       m_hdr.data(), m_ent.data(), m_done.data(), m_rag.data(), (unsigned)(rag.size() - 1)};
    int st = decode_utterance<true>(x, w, d, blank_id, probs + (size_t)b * T * V, (const PrunedRows *)nullptr, len, pool.data(), pool_up.data(),
                                    (int)pool.size(), ctcmath::host_tables().w, &outs, b);
    if (st != ST_OK) return -st;
  }
  for (int b = 0; b < B; ++b) {
    if (m_done[b] != 1) return -90;
    expand_item_host(m_hdr.data(), m_ent.data(), m_rag.data(), b, beam, T, out_tokens, out_timesteps);
  }
  if (labels_used) *labels_used = count;
  return 1;
}

// Scorer::get_log_cond_prob through the product's tables (host copy): words = n NUL-terminated strings
extern "C" double ctccore_lm_cond(const char *lm_path, const char *labels, int V, const char *words, int n, int32_t *meta3) {
  std::vector<std::string> lab(V), ws(n);
  for (int i = 0; i < V; ++i) { lab[i] = labels; labels += lab[i].size() + 1; }
  for (int i = 0; i < n; ++i) { ws[i] = words; words += ws[i].size() + 1; }
  ctclm::HostScorer hs;
  if (!hs.build(0.0, 0.0, lm_path, lab)) return 1e300;
  if (meta3) { meta3[0] = hs.char_based; meta3[1] = hs.order; meta3[2] = hs.dict_size; }
  return hs.cond_log_prob(ws);
}

static int decode_impl(const float *probs, const int32_t *seq_lens, int B, int T, int V, int beam, int num_threads,
                       double cutoff_prob, int cutoff_top_n, int blank_id, int32_t *out_tokens,
                       int32_t *out_timesteps, float *out_scores, int32_t *out_lens, int32_t *n_results,
                       const ctclm::LmView *lm, const float *raw, int raw_log) {
  using namespace ctcbeam;
  const bool pruned = std::log(cutoff_prob) < 0.0 || cutoff_top_n < V;
  Dims d;
  d.K = beam;
  d.V = V;
  d.Vc_max = pruned ? std::min(V, cutoff_top_n) : V;
  d.use_rank_table = pruned ? 1 : 0;
  d.lm = lm ? 1 : 0;
  std::atomic<int> next{0}, bad{0};
  auto work = [&] {
    Work w;
    size_t far_bytes = 0;
    const bool big = getenv("CTC_HOST_BIG") != nullptr;  // exercise the HBM-scratch layouts too (CTC_HOST_BIG=1 or 2)
    const bool farrep = !big && getenv("CTC_HOST_FARREP") != nullptr;  // ... and the layout whose replay scratch alone is "far"
    const int flevel = big ? getenv("CTC_HOST_BIG")[0] - '0' : 1;  // 1, 2, 3 (3: 32-bit slot indices)
    std::vector<char> mem((farrep ? carve<0, true>(w, nullptr, nullptr, d, &far_bytes) : !big ? carve<0>(w, nullptr, nullptr, d, &far_bytes) : flevel == 3 ? carve<3>(w, nullptr, nullptr, d, &far_bytes) : flevel == 2 ? carve<2>(w, nullptr, nullptr, d, &far_bytes)
                                                                                   : carve<1>(w, nullptr, nullptr, d, &far_bytes)) + 64);
    std::vector<char> far(far_bytes + 64);
    std::vector<PoolNode> pool((size_t)1 + (size_t)beam * T);
    std::vector<int> pool_up(2 * pool.size());  // express pointers | time steps' high parts
    std::vector<int> pcnt(T), pch((size_t)T * d.Vc_max);
    std::vector<float> plp((size_t)T * d.Vc_max);
    for (;;) {
      int b = next.fetch_add(1);
      if (b >= B) return;
      int len = seq_lens ? seq_lens[b] : T;
      len = std::max(0, std::min(len, T));
      if (farrep) carve<0, true>(w, mem.data(), far.data(), d, nullptr);
      else if (!big) carve<0>(w, mem.data(), far.data(), d, nullptr);
      else if (flevel == 3) carve<3>(w, mem.data(), far.data(), d, nullptr);
      else if (flevel == 2) carve<2>(w, mem.data(), far.data(), d, nullptr);
      else carve<1>(w, mem.data(), far.data(), d, nullptr);
      HostX x;
      x.far_ = big;
      const float *rows = probs + (size_t)b * T * V;
      PrunedRows pr{pcnt.data(), pch.data(), plp.data(), d.Vc_max};
      if (pruned)
        for (int t = 0; t < len; ++t) prune_row(rows + (size_t)t * V, V, cutoff_prob, cutoff_top_n, &pcnt[t], &pch[(size_t)t * d.Vc_max], &plp[(size_t)t * d.Vc_max]);
      int st;
      const OutRefs outs{out_tokens, out_timesteps, out_scores, out_lens, n_results, beam, T, nullptr, nullptr, nullptr, nullptr, 0u, nullptr, nullptr, nullptr, nullptr, 0u};
      if (lm && big && flevel == 3) {  // ... with 32-bit slot indices
        const float *rawb = raw + (size_t)b * T * V;
        if (pruned) st = decode_utterance<false, false, true, true, true, true>(x, w, d, blank_id, (const float *)nullptr, &pr, len, pool.data(), pool_up.data(), (int)pool.size(),
                                  ctcmath::host_tables().w, &outs, b, (const StreamState *)nullptr, lm, rawb, raw_log);
        else st = decode_utterance<true, false, true, true, true, true>(x, w, d, blank_id, rows, (const PrunedRows *)nullptr, len, pool.data(), pool_up.data(), (int)pool.size(),
                                  ctcmath::host_tables().w, &outs, b, (const StreamState *)nullptr, lm, rawb, raw_log);
      } else if (lm && big) {  // wide-beam layouts with the scorer: its per-entry state lives in the HBM scratch, no info words (LAZY)
        const float *rawb = raw + (size_t)b * T * V;
        if (pruned) st = decode_utterance<false, false, true, true>(x, w, d, blank_id, (const float *)nullptr, &pr, len, pool.data(), pool_up.data(), (int)pool.size(),
                                  ctcmath::host_tables().w, &outs, b, (const StreamState *)nullptr, lm, rawb, raw_log);
        else st = decode_utterance<true, false, true, true>(x, w, d, blank_id, rows, (const PrunedRows *)nullptr, len, pool.data(), pool_up.data(), (int)pool.size(),
                                  ctcmath::host_tables().w, &outs, b, (const StreamState *)nullptr, lm, rawb, raw_log);
      } else if (lm && !lm->char_based && !lm->dict_wide && !getenv("CTC_HOST_GENERAL_LM") && beam <= 128 && V <= 32 && !getenv("CTC_HOST_LM_RUNTIME_LAYOUT")) {
        // the word-model instantiation of the fixed-layout class (SMALLV = 1: what the device runs for these shapes -- with the speculative select)
        const float *rawb = raw + (size_t)b * T * V;
        if (pruned) st = decode_utterance<false, 1, true, false, false, false, true>(x, w, d, blank_id, (const float *)nullptr, &pr, len, pool.data(), pool_up.data(), (int)pool.size(),
                                  ctcmath::host_tables().w, &outs, b, (const StreamState *)nullptr, lm, rawb, raw_log);
        else st = decode_utterance<true, 1, true, false, false, false, true>(x, w, d, blank_id, rows, (const PrunedRows *)nullptr, len, pool.data(), pool_up.data(), (int)pool.size(),
                                  ctcmath::host_tables().w, &outs, b, (const StreamState *)nullptr, lm, rawb, raw_log);
      } else if (lm && !lm->char_based && !lm->dict_wide && !getenv("CTC_HOST_GENERAL_LM")) {  // the word-model instantiation, as the product picks it
        const float *rawb = raw + (size_t)b * T * V;
        if (pruned) st = decode_utterance<false, false, true, false, false, false, true>(x, w, d, blank_id, (const float *)nullptr, &pr, len, pool.data(), pool_up.data(), (int)pool.size(),
                                  ctcmath::host_tables().w, &outs, b, (const StreamState *)nullptr, lm, rawb, raw_log);
        else st = decode_utterance<true, false, true, false, false, false, true>(x, w, d, blank_id, rows, (const PrunedRows *)nullptr, len, pool.data(), pool_up.data(), (int)pool.size(),
                                  ctcmath::host_tables().w, &outs, b, (const StreamState *)nullptr, lm, rawb, raw_log);
      } else if (lm && beam <= 128 && V <= 32 && !getenv("CTC_HOST_LM_RUNTIME_LAYOUT")) {  // any scorer, fixed-layout class
        const float *rawb = raw + (size_t)b * T * V;
        if (pruned) st = decode_utterance<false, 1, true>(x, w, d, blank_id, (const float *)nullptr, &pr, len, pool.data(), pool_up.data(), (int)pool.size(),
                                  ctcmath::host_tables().w, &outs, b, (const StreamState *)nullptr, lm, rawb, raw_log);
        else st = decode_utterance<true, 1, true>(x, w, d, blank_id, rows, (const PrunedRows *)nullptr, len, pool.data(), pool_up.data(), (int)pool.size(),
                                  ctcmath::host_tables().w, &outs, b, (const StreamState *)nullptr, lm, rawb, raw_log);
      } else if (lm) {
        const float *rawb = raw + (size_t)b * T * V;
        if (pruned) st = decode_utterance<false, false, true>(x, w, d, blank_id, (const float *)nullptr, &pr, len, pool.data(), pool_up.data(), (int)pool.size(),
                                  ctcmath::host_tables().w, &outs, b, (const StreamState *)nullptr, lm, rawb, raw_log);
        else st = decode_utterance<true, false, true>(x, w, d, blank_id, rows, (const PrunedRows *)nullptr, len, pool.data(), pool_up.data(), (int)pool.size(),
                                  ctcmath::host_tables().w, &outs, b, (const StreamState *)nullptr, lm, rawb, raw_log);
      } else if (big && flevel == 3) {  // ... and the layout for more than 65535 candidate slots (32-bit slot indices)
        if (pruned) st = decode_utterance<false, false, false, true, true, true>(x, w, d, blank_id, (const float *)nullptr, &pr, len, pool.data(), pool_up.data(), (int)pool.size(),
                                  ctcmath::host_tables().w, &outs, b);
        else st = decode_utterance<true, false, false, true, true, true>(x, w, d, blank_id, rows, (const PrunedRows *)nullptr, len, pool.data(), pool_up.data(), (int)pool.size(),
                                  ctcmath::host_tables().w, &outs, b);
      } else if (big) {  // the wide-beam layouts do not store the per-slot info words (LAZY)
        if (pruned) st = decode_utterance<false, false, false, true>(x, w, d, blank_id, (const float *)nullptr, &pr, len, pool.data(), pool_up.data(), (int)pool.size(),
                                  ctcmath::host_tables().w, &outs, b);
        else st = decode_utterance<true, false, false, true>(x, w, d, blank_id, rows, (const PrunedRows *)nullptr, len, pool.data(), pool_up.data(), (int)pool.size(),
                                  ctcmath::host_tables().w, &outs, b);
      } else if (farrep) {
        if (pruned) st = decode_utterance<false, false, false, false, true>(x, w, d, blank_id, (const float *)nullptr, &pr, len, pool.data(), pool_up.data(), (int)pool.size(),
                                  ctcmath::host_tables().w, &outs, b);
        else st = decode_utterance<true, false, false, false, true>(x, w, d, blank_id, rows, (const PrunedRows *)nullptr, len, pool.data(), pool_up.data(), (int)pool.size(),
                                  ctcmath::host_tables().w, &outs, b);
      } else if (pruned && !(beam <= 128 && V <= 32) && beam <= kMidK && d.Vc_max <= kMidVc && V <= kMidV && !getenv("CTC_HOST_NO_CLASS2")) {
        // the second class with a compile-time layout on the device (the pruned default on a large vocabulary: beam_core.h kMidK)
        st = decode_utterance<false, 2>(x, w, d, blank_id, (const float *)nullptr, &pr, len, pool.data(), pool_up.data(), (int)pool.size(),
                                  ctcmath::host_tables().w, &outs, b);
      } else if (beam <= 128 && V <= 32) {  // the shapes the device runs with its fixed layout
        if (pruned) st = decode_utterance<false, true>(x, w, d, blank_id, (const float *)nullptr, &pr, len, pool.data(), pool_up.data(), (int)pool.size(),
                                  ctcmath::host_tables().w, &outs, b);
        else st = decode_utterance<true, true>(x, w, d, blank_id, rows, (const PrunedRows *)nullptr, len, pool.data(), pool_up.data(), (int)pool.size(),
                                  ctcmath::host_tables().w, &outs, b);
      } else {
        if (pruned) st = decode_utterance<false>(x, w, d, blank_id, (const float *)nullptr, &pr, len, pool.data(), pool_up.data(), (int)pool.size(),
                                  ctcmath::host_tables().w, &outs, b);
        else st = decode_utterance<true>(x, w, d, blank_id, rows, (const PrunedRows *)nullptr, len, pool.data(), pool_up.data(), (int)pool.size(),
                                  ctcmath::host_tables().w, &outs, b);
      }
      if (st != ST_OK) bad = st;
    }
  };
  std::vector<std::thread> pool_threads;
  for (int i = 1; i < std::min(num_threads, B); ++i) pool_threads.emplace_back(work);
  work();
  for (auto &t : pool_threads) t.join();
  return bad ? -bad : 1;
}

// Streaming form of the same core: every utterance is fed chunk by chunk (frame boundaries in `bounds`, nchunks + 1
// increasing values from 0 to T), state parked in a StreamState between chunks, results taken at the last chunk.
// Must reproduce the one-shot decode exactly.  No vocabulary pruning here (identity candidate lists).
extern "C" int ctccore_decode_chunked_f32(const float *probs, int B, int T, int V, int beam, int blank_id, const int32_t *bounds,
                                          int nchunks, int32_t *out_tokens, int32_t *out_timesteps, float *out_scores,
                                          int32_t *out_lens, int32_t *n_results) {
  using namespace ctcbeam;
  Dims d;
  d.K = beam; d.V = V; d.Vc_max = V; d.use_rank_table = 0; d.lm = 0;
  Work w;
  size_t far_bytes = 0;
  const bool big = getenv("CTC_HOST_BIG") != nullptr;  // the HBM-scratch layouts (CTC_HOST_BIG=1 or 2), as in decode_impl
  const int flevel = big ? getenv("CTC_HOST_BIG")[0] - '0' : 1;
  std::vector<char> mem((!big ? carve<0>(w, nullptr, nullptr, d, &far_bytes) : flevel == 3 ? carve<3>(w, nullptr, nullptr, d, &far_bytes) : flevel == 2 ? carve<2>(w, nullptr, nullptr, d, &far_bytes)
                                                                                 : carve<1>(w, nullptr, nullptr, d, &far_bytes)) + 64);
  std::vector<char> far(far_bytes + 64);
  for (int b = 0; b < B; ++b) {
    std::vector<PoolNode> pool((size_t)1 + (size_t)beam * T);
    std::vector<int> pool_up(2 * pool.size());  // express pointers | time steps' high parts
    std::vector<int> hdr(SH_WORDS, 0), arrays((size_t)kStateArrays * beam, 0);
    for (int c = 0; c < nchunks; ++c) {
      const int lo = bounds[c], hi = bounds[c + 1];
      // a fresh workspace every chunk, as a new kernel launch would have
      std::fill(mem.begin(), mem.end(), (char)0x5a);
      std::fill(far.begin(), far.end(), (char)0x5a);
      if (!big) carve<0>(w, mem.data(), far.data(), d, nullptr);
      else if (flevel == 3) carve<3>(w, mem.data(), far.data(), d, nullptr);
      else if (flevel == 2) carve<2>(w, mem.data(), far.data(), d, nullptr);
      else carve<1>(w, mem.data(), far.data(), d, nullptr);
      HostX x;
      x.far_ = big;
      StreamState ss{hdr.data(), arrays.data(), c == nchunks - 1 ? 1 : 0};
      const OutRefs outs{out_tokens, out_timesteps, out_scores, out_lens, n_results, beam, T, nullptr, nullptr, nullptr, nullptr, 0u, nullptr, nullptr, nullptr, nullptr, 0u};
      int st;
      if (big && flevel == 3) st = decode_utterance<true, false, false, true, true, true>(x, w, d, blank_id, probs + ((size_t)b * T + lo) * V, (const PrunedRows *)nullptr, hi - lo,
                                pool.data(), pool_up.data(), (int)pool.size(), ctcmath::host_tables().w, &outs, b, &ss);
      else if (big) st = decode_utterance<true, false, false, true>(x, w, d, blank_id, probs + ((size_t)b * T + lo) * V, (const PrunedRows *)nullptr, hi - lo,
                                pool.data(), pool_up.data(), (int)pool.size(), ctcmath::host_tables().w, &outs, b, &ss);
      else st = decode_utterance<true>(x, w, d, blank_id, probs + ((size_t)b * T + lo) * V, (const PrunedRows *)nullptr, hi - lo, pool.data(),
                                pool_up.data(), (int)pool.size(), ctcmath::host_tables().w, &outs, b, &ss);
      if (st != ST_OK) return -st;
    }
  }
  return 1;
}

// Small helpers of beam_core.h checked exhaustively / on random patterns (tests/test_core_host.py).  Returns the
// number of violations found.
extern "C" long long ctccore_check_helpers(unsigned long long seed, long long n_random) {
  using namespace ctcbeam;
  long long bad = 0;
  auto bits_f = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
  // ord_f32: order-preserving on every pair of non-NaN floats, both zeros coincide (operator== / operator> of the reference)
  auto check_pair = [&](uint32_t ua, uint32_t ub) {
    const float a = bits_f(ua), b = bits_f(ub);
    if (a != a || b != b) return;
    const uint32_t ka = ord_f32(a), kb = ord_f32(b);
    if ((a > b) != (ka > kb) || (a == b) != (ka == kb)) ++bad;
  };
  const uint32_t special[] = {0x00000000u, 0x80000000u, 0x00000001u, 0x80000001u, 0x007fffffu, 0x807fffffu, 0x00800000u, 0x80800000u,
                              0x3f800000u, 0xbf800000u, 0x7f7fffffu, 0xff7fffffu, 0x7f800000u, 0xff800000u, 0xc4610000u, 0xc4610001u};
  for (uint32_t a : special)
    for (uint32_t b : special) check_pair(a, b);
  unsigned long long s = seed * 6364136223846793005ull + 1442695040888963407ull;
  for (long long i = 0; i < n_random; ++i) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    const uint32_t ua = (uint32_t)(s >> 32);
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    uint32_t ub = (uint32_t)(s >> 32);
    if (i & 1) ub = ua + (uint32_t)((s >> 8) & 7) - 3u;  // neighbours in bit space
    check_pair(ua, ub);
  }
  // shifts that stand in for divisions
  for (int d = 1; d <= 4096; d <<= 1)
    for (int v = 0; v < 70000; v += (v < 300 ? 1 : 97)) {
      if (div_p2(v, d) != v / d) ++bad;
      if (ceil_div_p2(v, d) != (v + d - 1) / d) ++bad;
    }
  for (uint32_t v = 1; v < 100000; ++v) {
    int sft = 0;
    while ((1ull << sft) < v) ++sft;
    if (ceil_log2_u32(v) != sft || ceil_log2_u64(v) != sft) ++bad;
  }
  if (ceil_log2_u32(0x80000000u) != 31 || ceil_log2_u32(0x80000001u) != 32 || ceil_log2_u32(0xFFFFFFFFu) != 32) ++bad;
  // info words: round trip, and "character ascending" = larger top half first
  for (int ch = -1; ch < 66000; ch += (ch < 70 ? 1 : 331))
    for (uint32_t type = 0; type < 4; ++type)
      for (int e = 0; e <= kMaxBeam; e += 4095) {
        if (ch + 1 > 0xFFFF) continue;
        const uint32_t inf = mk_info(ch, type, e);
        if (info_ch(inf) != ch || info_type(inf) != type || info_entry(inf) != e) ++bad;
        if (ch >= 0 && (mk_info(ch, 0, 0) >> 16) >= (mk_info(ch - 1, 0, 0) >> 16)) ++bad;
      }
  return bad;
}
