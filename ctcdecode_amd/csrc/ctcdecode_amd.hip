// ctcdecode_amd.hip -- gfx950 kernels + C ABI (include/ctcdecode_amd.h) of the CTC prefix beam-search decoder.
//
// Grid mapping (replaces the reference's ThreadPool fan-out, ctc_beam_search_decoder.cpp:259-275): one workgroup per
// utterance; the whole T-step recurrence of that utterance runs inside ONE persistent kernel launch with the beam in
// LDS (a kernel boundary costs ~1.5 us on MI355X -- more than a whole time step should).  The per-utterance algorithm
// is beam_core.h; this file supplies the workgroup execution policy (barriers, wave-shuffle reductions/scans), the
// elementwise prob->log pre-pass, and the host-side marshalling of binding.cpp:35-101.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <limits>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/ctcdecode_amd.h"
#define CTC_EXACT_MATH_HOST_TABLES
#include "decode_kernel.h"
#include "exact_math_f64.h"
#include "lm_build.h"
#include "lm_callback.h"
#include "compact_results.h"

namespace {

using namespace ctcbeam;

// (execution policy + kernel template: decode_kernel.h; its instantiations are compiled in decode_kernels.hip)
using namespace ctcdk;
#define CTC_X_EXTERN(PROF_, BIG_, LAYOUT_, PRUNED_, NT_, LM_, OCC2_, G_) extern template __global__ void ctcdk::ctc_beam_decode_kernel<PROF_, BIG_, LAYOUT_, PRUNED_, NT_, LM_, OCC2_>(ctcdk::KernelArgs);
} namespace ctcdk { CTC_KERNEL_LIST(CTC_X_EXTERN) } namespace {
#undef CTC_X_EXTERN


// The data of the bit-exact binary64 log / exp (exact_math_f64.h): the vocabulary pruning and the probability -> log
// conversion evaluate exactly what the reference's C library evaluates (decoder_utils.cpp:16,29,42), on the device.
#define g_t64 (ctcmath::tables64())

// prob -> log-prob exactly as decoder_utils.cpp:42 : float(log(double(p) + FLT_MIN)), every element, bit for bit.
// (Rounds 1-3 used the device library's log() and sent the elements whose float rounding it could not guarantee through
//  the host's libm; the restated routine needs no second opinion.)
__global__ void prob_to_log_kernel(const float *in, float *out, size_t n, const int32_t *seq_lens, int T, int V) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t tv = (size_t)T * V;
  for (; i < n; i += stride) {
    if (seq_lens) {  // frames beyond the utterance's length are never read by the reference (binding.cpp:64-65)
      const size_t b = i / tv;
      const int t = (int)((i - b * tv) / (size_t)V);
      if (t >= seq_lens[b]) continue;
    }
    out[i] = (float)ctcmath::log_f64((double)in[i] + (double)FLT_MIN, g_t64);
  }
}

// Raw-logit input (log_input == 2, an extension: the reference's callers run log_softmax themselves): one wave per frame,
//   y_j = (x_j - m) - logf(s),  m = max_j x_j (a zero maximum taken as +0),  s = sum_j expf(x_j - m) in float32,
// where lane l first adds up the terms j = l, l + 64, ... in increasing j and the 64 partial sums are then combined by a
// butterfly (lane ^ 1, ^ 2, ... ^ 32); expf / logf are the bit-exact restatements of exact_math.h (expf below -88 is 0).
// The order of the additions is part of the definition: tests/native/core_host.cpp computes the same thing with the C
// library, bit for bit.  A frame without a finite logit yields -inf everywhere.
__global__ void __launch_bounds__(256) log_softmax_rows_kernel(const float *in, float *out, long long rows, const int32_t *seq_lens, int T, int V,
                                                               const uint64_t *tables) {
  __shared__ uint64_t tbl[64];
  if (threadIdx.x < 64) tbl[threadIdx.x] = tables[threadIdx.x];
  __syncthreads();
  const int lane = (int)threadIdx.x & 63;
  const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
  for (long long r = wave0; r < rows; r += nwaves) {
    if (seq_lens) {  // frames beyond the utterance's length are never read (binding.cpp:64-65)
      const long long b = r / T;
      if ((int)(r - b * T) >= seq_lens[b]) continue;
    }
    const float *x = in + (size_t)r * V;
    float *y = out + (size_t)r * V;
    float m = -INFINITY;
    for (int j = lane; j < V; j += 64) { const float v = x[j]; m = v > m ? v : m; }
    for (int off = 1; off < 64; off <<= 1) { const float o = __shfl_xor(m, off, 64); m = o > m ? o : m; }
    m += 0.0f;  // (a zero maximum is +0)
    if (!(m > -INFINITY)) {
      for (int j = lane; j < V; j += 64) y[j] = -INFINITY;
      continue;
    }
    float part = 0.0f;
    for (int j = lane; j < V; j += 64) part += ctcmath::expf_nonpos(x[j] - m, tbl);
    for (int off = 1; off < 64; off <<= 1) part += __shfl_xor(part, off, 64);
    const float ls = ctcmath::logf_normal(part, tbl);
    for (int j = lane; j < V; j += 64) y[j] = (x[j] - m) - ls;
  }
}

// The same normalisation for long rows, shaped for HBM bandwidth: one workgroup of four waves per frame, the row read ONCE
// with 128-bit loads and held in registers (thread t: the float4s t, t + 256, ...), written once.  The sum keeps the order
// the definition fixes -- 64 chains, chain l over the terms j = l, l + 64, ... in increasing j -- although no thread holds a
// chain's terms: the exponentials of 1024 consecutive labels at a time go to LDS (every thread computes four), and one wave
// per such chunk (the four take turns) extends the 64 chains by sixteen terms each, lane l reading its chain's terms at
// l + 64 k.  Two chunk buffers, one barrier per chunk; the chains' running values pass from wave to wave through LDS.
// Requires V % 4 == 0 and V <= 1024 * F4.
constexpr int kLsmChunk = 1024;  // labels per chunk of exponentials (four per thread)
struct LsmLds {
  float terms[2][kLsmChunk];
  float chain[64];
  float wmax[4];
};
template <int F4>
__device__ __forceinline__ float wg_exp_sum(const float4 (&v)[F4], int nv4, float m, LsmLds &s, const uint64_t *tbl, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int u = 0; u < F4; ++u) {
    if (256 * u < nv4) {  // (uniform; chunks past the row's end would add zeros)
      float4 e = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (tid + 256 * u < nv4) {
        // (two at a time: four interleaved binary64 evaluations on top of the row cost registers -- the other waves hide the latency)
        e.x = ctcmath::expf_nonpos(v[u].x - m, tbl); e.y = ctcmath::expf_nonpos(v[u].y - m, tbl);
        __builtin_amdgcn_sched_barrier(0);
        e.z = ctcmath::expf_nonpos(v[u].z - m, tbl); e.w = ctcmath::expf_nonpos(v[u].w - m, tbl);
        __builtin_amdgcn_sched_barrier(0);
      }
      *reinterpret_cast<float4 *>(&s.terms[u & 1][4 * tid]) = e;
      __syncthreads();
      if (wave == (u & 3)) {
        float p = u ? s.chain[lane] : 0.0f;
        const float *c = &s.terms[u & 1][lane];
#pragma unroll
        for (int k = 0; k < kLsmChunk / 64; ++k) p += c[64 * k];
        s.chain[lane] = p;
      }
    }
  }
  __syncthreads();
  float part = s.chain[lane];
  for (int off = 1; off < 64; off <<= 1) part += __shfl_xor(part, off, 64);
  return part;
}
// the row's maximum (NaN-ignoring, as the one-wave kernel's), known to every thread; contains one barrier
__device__ __forceinline__ float wg_row_max(float mine, LsmLds &s, int tid) {
  for (int off = 1; off < 64; off <<= 1) { const float o = __shfl_xor(mine, off, 64); mine = o > mine ? o : mine; }
  if ((tid & 63) == 0) s.wmax[tid >> 6] = mine;
  __syncthreads();
  float m = s.wmax[0];
  m = s.wmax[1] > m ? s.wmax[1] : m; m = s.wmax[2] > m ? s.wmax[2] : m; m = s.wmax[3] > m ? s.wmax[3] : m;
  return m + 0.0f;  // (a zero maximum is +0 whichever zero the reduction met first: part of the definition)
}
template <int F4>
__global__ void __launch_bounds__(256, F4 <= 10 ? 5 : 4) log_softmax_rows_wg_kernel(const float *in, float *out, long long rows, const int32_t *seq_lens, int T,
                                                                                     int V, const uint64_t *tables) {
  __shared__ uint64_t tbl[64];
  __shared__ __attribute__((aligned(16))) LsmLds s;
  const int tid = (int)threadIdx.x, nv4 = V >> 2;
  if (tid < 64) tbl[tid] = tables[tid];
  __syncthreads();
  for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
    if (seq_lens) {  // frames beyond the utterance's length are never read (binding.cpp:64-65)
      const long long b = r / T;
      if ((int)(r - b * T) >= seq_lens[b]) continue;
    }
    const float4 *x4 = reinterpret_cast<const float4 *>(in + (size_t)r * V);
    float4 *y4 = reinterpret_cast<float4 *>(out + (size_t)r * V);
    int tq = tid;  // (opaque per frame: the chunks' predicates and offsets are recomputed, not kept across the frame loop)
    asm volatile("" : "+v"(tq));
    float4 v[F4];
#pragma unroll
    for (int u = 0; u < F4; ++u) {
      const int i4 = tq + 256 * u;
      v[u] = i4 < nv4 ? x4[i4] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    }
    float mine = -INFINITY;
#pragma unroll
    for (int u = 0; u < F4; ++u) {
      mine = v[u].x > mine ? v[u].x : mine; mine = v[u].y > mine ? v[u].y : mine;
      mine = v[u].z > mine ? v[u].z : mine; mine = v[u].w > mine ? v[u].w : mine;
    }
    const float m = wg_row_max(mine, s, tid);
    if (!(m > -INFINITY)) {
#pragma unroll
      for (int u = 0; u < F4; ++u)
        if (tq + 256 * u < nv4) y4[tq + 256 * u] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    } else {
      const float ls = ctcmath::logf_normal(wg_exp_sum<F4>(v, nv4, m, s, tbl, tq), tbl);
#pragma unroll
      for (int u = 0; u < F4; ++u)
        if (tq + 256 * u < nv4) y4[tq + 256 * u] = make_float4((v[u].x - m) - ls, (v[u].y - m) - ls, (v[u].z - m) - ls, (v[u].w - m) - ls);
    }
    __syncthreads();  // (the chunk buffers and wmax are reused by the next frame)
  }
}

// ------------------------------------------------------------------------------------------------ vocabulary prune
// get_pruned_log_probs (decoder_utils.cpp:10-45) for every frame, one wave per frame, no barriers: the top
// min(cutoff_top_n, V) values in descending order (and, with cutoff_prob < 1, the reference's cumulative cut).
// The order std::sort gives EQUAL values is toolchain behaviour the reference inherits, and its cumulative cut is a
// sequential chain of binary64 log / exp: whenever a frame's result could depend on either (equal values at or above the
// cut, a cumulative sum within rounding distance of cutoff_prob) the frame is flagged and settled by prune_resolve_kernel
// -- on the device: a replay of libstdc++'s std::sort and the chain with the bit-exact log / exp of exact_math_f64.h.
// Everything else is decided here, exactly (the kept probabilities' logs with the same bit-exact log).
// NaN: the reference's comparator is not a strict weak order on rows that hold one (its std::sort call is undefined
// behaviour there); here a NaN ranks below every number, -inf included, and is otherwise carried through.
constexpr int kPruneCand = 256;  // capacity of the pre-filter candidate list per frame

struct PruneArgs {
  const float *in;          // [B, T, V]
  const int32_t *seq_lens;  // [B] or null
  int T, V, top_n, log_input, stride;
  long long rows;           // B * T
  double cutoff_prob;
  int *cnt, *ch;
  float *lp;
  unsigned *n_flag, *flag_rows;
  unsigned flag_cap;
  double cut_exp;            // exp(cutoff_prob)
  const uint64_t *tables;    // exact_math.h Tables, in global memory
  float *row_max, *row_lse;  // log_input == 2 (raw logits, prune_logits_wg_kernel): every frame's maximum and logf(sum of exponentials)
};

// log_input == 2: the value the prune ranks is the frame's log-softmax (x - m) - ls, never stored as a row; m = -inf (no finite
// logit): -inf everywhere, as log_softmax_rows_kernel defines it
__device__ __forceinline__ float prune_row_value(const PruneArgs &a, float x, float m, float ls) {
  if (a.log_input != 2) return x;
  return m > -INFINITY ? (x - m) - ls : -INFINITY;
}

__device__ __forceinline__ uint32_t prune_key(float v) {  // order of the doubles the reference compares; -0 == +0
  uint32_t u = __float_as_uint(v);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 1u;  // NaN: below every number (key 0 = no element)
  if (u == 0x80000000u) u = 0;
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Inclusive prefix sum of a double over the 64 lanes of a wave with DPP moves on the two halves of the value (a
// ds_bpermute-based shuffle costs an LDS round trip per step; this runs while the other waves of the workgroup wait).
__device__ __forceinline__ double f64_dpp(double v, const int ctrl_sel) {
  union { double d; int i[2]; } in, out;
  in.d = v;
  switch (ctrl_sel) {  // (the DPP control word is an immediate)
    case 0: out.i[0] = CTC_DPP(0, in.i[0], 0x111, 0xf); out.i[1] = CTC_DPP(0, in.i[1], 0x111, 0xf); break;  // row_shr:1
    case 1: out.i[0] = CTC_DPP(0, in.i[0], 0x112, 0xf); out.i[1] = CTC_DPP(0, in.i[1], 0x112, 0xf); break;  // row_shr:2
    case 2: out.i[0] = CTC_DPP(0, in.i[0], 0x114, 0xf); out.i[1] = CTC_DPP(0, in.i[1], 0x114, 0xf); break;  // row_shr:4
    case 3: out.i[0] = CTC_DPP(0, in.i[0], 0x118, 0xf); out.i[1] = CTC_DPP(0, in.i[1], 0x118, 0xf); break;  // row_shr:8
    case 4: out.i[0] = CTC_DPP(0, in.i[0], 0x142, 0xa); out.i[1] = CTC_DPP(0, in.i[1], 0x142, 0xa); break;  // row_bcast:15 -> rows 1, 3
    default: out.i[0] = CTC_DPP(0, in.i[0], 0x143, 0xc); out.i[1] = CTC_DPP(0, in.i[1], 0x143, 0xc); break; // row_bcast:31 -> rows 2, 3
  }
  return out.d;  // (lanes without a source receive +0.0: the identity)
}
__device__ __forceinline__ double wave_scan_f64_sum(double v) {
  v += f64_dpp(v, 0); v += f64_dpp(v, 1); v += f64_dpp(v, 2); v += f64_dpp(v, 3); v += f64_dpp(v, 4); v += f64_dpp(v, 5);
  return v;
}
__device__ __forceinline__ double f64_from_lane(double v, int lane_idx) {
  union { double d; int i[2]; } in, out;
  in.d = v;
  out.i[0] = __builtin_amdgcn_readlane(in.i[0], lane_idx); out.i[1] = __builtin_amdgcn_readlane(in.i[1], lane_idx);
  return out.d;
}

__device__ __forceinline__ double log_add_f64(double a, double b) {  // decoder_utils.h:47-54 with T = double
  const double neg = -1.7976931348623157e308;
  if (a <= neg) return b;
  if (b <= neg) return a;
  const double m = a > b ? a : b;
  return log(exp(a - m) + exp(b - m)) + m;
}

// R > 0: the frame's keys are held in registers (64*R >= V); R == 0: re-read from memory on every pass.
template <int R>
__global__ void __launch_bounds__(256) prune_rows_kernel(PruneArgs a) {
  extern __shared__ __attribute__((aligned(16))) char psm[];
  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6, wpb = (int)blockDim.x >> 6;
  const int n = a.top_n < a.V ? a.top_n : a.V;
  // per wave: kept keys, labels, labels in final order (stride each); pre-filter candidates (keys, labels)
  uint32_t *lkey = (uint32_t *)psm + (size_t)wave * (3 * a.stride + 2 * kPruneCand);
  int *lidx = (int *)lkey + a.stride;
  int *sidx = lidx + a.stride;
  uint32_t *ckey = (uint32_t *)(sidx + a.stride);
  int *cidx = (int *)ckey + kPruneCand;
  for (long long r = (long long)blockIdx.x * wpb + wave; r < a.rows; r += (long long)gridDim.x * wpb) {
    if (a.seq_lens) {  // frames beyond the utterance's length are never read (binding.cpp:64-65)
      const long long b = r / a.T;
      int len = a.seq_lens[b];
      len = len < 0 ? 0 : len;
      if ((int)(r - b * a.T) >= len) continue;
    }
    const float *x = a.in + (size_t)r * a.V;
    bool flag = false;
    uint32_t keys[R > 0 ? R : 1];
    if (R > 0) {
#pragma unroll
      for (int u = 0; u < R; ++u) keys[u] = lane + 64 * u < a.V ? prune_key(x[lane + 64 * u]) : 0u;  // 0 < every real key
    }
    uint32_t tau = 0;
    int g = 0, e = 0, base = 0;
    bool done = false;
    if (R > 0 && n <= 64) {
      // Pre-filter: tau is at least the n-th largest of the 64 per-lane maxima (those are n elements >= it), so only
      // keys >= that bound (a few dozen of V) can be among the top n.  They are listed in LDS and ranked exactly.
      uint32_t lmax = 0;
#pragma unroll
      for (int u = 0; u < R; ++u) lmax = keys[u] > lmax ? keys[u] : lmax;
      uint32_t bound = 0;
      for (int bit = 31; bit >= 0; --bit) {
        const uint32_t trial = bound | (1u << bit);
        if (__popcll(__ballot(lmax >= trial)) >= n) bound = trial;
      }
      int ns = 0;
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const bool in = keys[u] >= bound && keys[u] != 0u;
        const unsigned long long m = __ballot(in);
        if (m) {
          const int p = ns + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
          if (in && p < kPruneCand) { ckey[p] = keys[u]; cidx[p] = lane + 64 * u; }
          ns += __popcll(m);
        }
      }
      if (ns <= kPruneCand) {
        // exact n-th largest among the ns candidates (every lane ranks its own candidates against all of them)
        uint32_t found = 0;
        for (int q = lane; q < ns; q += 64) {
          const uint32_t mine = ckey[q];
          int gg = 0, ee = 0;
          for (int o = 0; o < ns; ++o) {
            const uint32_t k = ckey[o];
            gg += k > mine;
            ee += k == mine;
          }
          if (gg < n && n <= gg + ee) found = mine;
        }
        const unsigned long long mf = __ballot(found != 0u);
        tau = (uint32_t)__builtin_amdgcn_readlane((int)found, __ffsll((long long)mf) - 1);
        for (int q = lane; q < ns; q += 64) { g += ckey[q] > tau; e += ckey[q] == tau; }
        g = wave_sum(g);
        e = wave_sum(e);
        if (e > n - g) flag = true;  // equal values straddle the cut: std::sort decides which of them are kept
        for (int q0 = 0; q0 < ns; q0 += 64) {
          const int q = q0 + lane;
          const uint32_t k = q < ns ? ckey[q] : 0u;
          const bool keep = q < ns && (k > tau || (k == tau && !flag));
          const unsigned long long m = __ballot(keep);
          if (keep) {
            const int p = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            lkey[p] = k;
            lidx[p] = cidx[q];
          }
          base += __popcll(m);
        }
        done = true;
      }
    }
    if (!done) {
      // n-th largest key, bit by bit, over all V values
      for (int bit = 31; bit >= 0; --bit) {
        const uint32_t trial = tau | (1u << bit);
        int c = 0;
        if (R > 0) {
#pragma unroll
          for (int u = 0; u < R; ++u) c += keys[u] >= trial;
        } else {
          for (int i = lane; i < a.V; i += 64) c += prune_key(x[i]) >= trial;
        }
        if (wave_sum(c) >= n) tau = trial;
      }
      if (R > 0) {
#pragma unroll
        for (int u = 0; u < R; ++u) { g += keys[u] > tau; e += keys[u] == tau; }
      } else {
        for (int i = lane; i < a.V; i += 64) {
          const uint32_t k = prune_key(x[i]);
          g += k > tau;
          e += k == tau;
        }
      }
      g = wave_sum(g);
      e = wave_sum(e);
      if (e > n - g) flag = true;  // equal values straddle the cut: std::sort decides which of them are kept
      // gather the kept values
      auto take = [&](int i, uint32_t k, bool valid) {
        const bool keep = valid && (k > tau || (k == tau && !flag));
        const unsigned long long m = __ballot(keep);
        if (keep) {
          const int p = base + __popcll(m & ((1ull << lane) - 1ull));
          lkey[p] = k;
          lidx[p] = i;
        }
        base += __popcll(m);
      };
      if (R > 0) {
#pragma unroll
        for (int u = 0; u < R; ++u) take(lane + 64 * u, keys[u], lane + 64 * u < a.V);
      } else {
        for (int i0 = 0; i0 < a.V; i0 += 64) {
          const int i = i0 + lane;
          take(i, i < a.V ? prune_key(x[i]) : 0u, i < a.V);
        }
      }
    }
    const int kept = base;  // == n unless flagged
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes are visible to its other lanes
    // rank (descending); equal kept values -> their order is std::sort's business
    int *och = a.ch + (size_t)r * a.stride;
    float *olp = a.lp + (size_t)r * a.stride;
    for (int q = lane; q < kept; q += 64) {
      const uint32_t mine = lkey[q];
      int rank = 0, dup = 0;
      for (int o = 0; o < kept; ++o) {
        const uint32_t k = lkey[o];
        rank += k > mine;
        dup += k == mine;
      }
      if (dup > 1) flag = true;
      const int idx = lidx[q];
      float v = x[idx];
      if (!a.log_input) {  // decoder_utils.cpp:42
        v = (float)ctcmath::log_f64((double)v + (double)FLT_MIN, g_t64);
      }
      if (dup <= 1) { och[rank] = idx; olp[rank] = v; sidx[rank] = idx; }
    }
    flag = __ballot(flag) != 0ull;
    int len = kept;
    if (a.cutoff_prob < 1.0 && !flag) {
      // decoder_utils.cpp:25-32: cum = log_sum_exp(cum, log p_i) starting from cum = 0.0 (sic), i.e. after i+1 terms
      // cum = log(1 + p_0 + ... + p_i); keep going until cum >= cutoff_prob or cutoff_top_n entries.  Evaluated here
      // as a wave-parallel prefix sum (differs from the reference's sequential double chain by ~1e-14 relative); a
      // frame where any partial sum comes within 1e-9 of the threshold is left to prune_resolve_kernel's exact chain.
      int stop = kept;  // number of entries kept
      double carry = 0.0;
      for (int i0 = 0; i0 < kept && stop == kept; i0 += 64) {
        const int i = i0 + lane;
        double p = 0.0;
        if (i < kept) {
          const double v = (double)x[sidx[i]];
          p = a.log_input ? exp(v) : v;
        }
        double incl = p;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const double o = __shfl_up(incl, off, 64);
          if (lane >= off) incl += o;
        }
        const double cum = log(1.0 + carry + incl);
        const bool near = i < kept && (fabs(cum - a.cutoff_prob) <= 1e-9 * (1.0 + fabs(cum)) || !(cum == cum));
        const bool hit = i < kept && (cum >= a.cutoff_prob || i + 1 >= a.top_n);
        const unsigned long long mh = __ballot(hit), mn = __ballot(near);
        const int firsthit = mh ? __ffsll((long long)mh) - 1 : 64;
        if (mn && (__ffsll((long long)mn) - 1) <= firsthit) flag = true;  // ambiguous before (or at) the stopping point
        if (mh) stop = i0 + firsthit + 1;
        carry += f64_from_lane(incl, 63);
      }
      len = stop;
      flag = __ballot(flag) != 0ull;
    }
    if (lane == 0) {
      a.cnt[r] = len;
      if (flag) {
        const unsigned k = atomicAdd(a.n_flag, 1u);
        if (k < a.flag_cap) a.flag_rows[k] = (unsigned)r;
      }
    }
  }
}

// The same pass for large vocabularies, shaped for HBM bandwidth (the one genuinely HBM-bound kernel of this library:
// V * 4 bytes read per frame, 8 * top_n + 4 written).  One workgroup of four waves per frame, two sweeps over the row
// with 128-bit loads: the first keeps only every thread's maximum (a lower bound of the n-th largest value follows from
// the lanes' maxima: in each wave the ceil(n/4)-th largest of its 64 lane maxima, to 20 bits; the smallest of the four
// wave bounds has at least n values above it), the second -- served by L2, the row was just read -- lists the few
// dozen values at or above the bound in LDS, where wave 0 ranks them exactly.  Ties, the cumulative cut and the log
// conversion are decided as in prune_rows_kernel above.  Few registers per thread (nothing of the row is kept), so
// eight workgroups share a CU and hide each other's latencies.
// Requires V % 4 == 0 (16-byte aligned rows), V <= 1024 * F4, cutoff_top_n <= 64.
// decoder_utils.cpp:25-32 for one frame, by one wave: the kept candidates (sidx[0, kept): their labels, best first) are cut
// where the running sum of their probabilities reaches cutoff_prob (the reference accumulates log(1 + sum): its running
// value starts at 0.0 in log space).  This is the fast form -- a wave-parallel prefix sum and the device library's
// exp()/log(), not the reference's sequential chain: whenever the comparison with cutoff_prob could go either way before (or
// at) the stopping point, `flag` is raised and prune_resolve_kernel walks the chain exactly (prune_exact_cut).
__device__ __forceinline__ int prune_cumulative_cut(const PruneArgs &a, const float *x, const int *sidx, int kept, int lane, bool &flag, const uint64_t *tbl) {
  // cum = log(1 + sum) >= cutoff_prob  <=>  1 + sum >= exp(cutoff_prob) (a.cut_exp, from the host's libm): no logarithm here, and
  // the probabilities of log-probability rows from expf's own evaluation kept in binary64 (exact_math.h expf_core_f64, ~2e-10):
  // sums within 1e-8 of the threshold -- fifty times that error -- are not decided here.  (The device library's exp() / log(),
  // used until round 5, cost the kernels ~35 registers and their constants.)
  int stop = kept;
  double carry = 1.0;
  for (int i0 = 0; i0 < kept && stop == kept; i0 += 64) {
    const int i = i0 + lane;
    double p = 0.0;
    bool odd = false;
    if (i < kept) {
      const float v = sidx ? x[sidx[i]] : x[i];  // (sidx == nullptr: x[] already holds the kept values, best first)
      odd = !(v <= 80.0f) || (!a.log_input && v < 0.0f);  // (outside expf_core_f64's range, a NaN, a negative probability)
      p = a.log_input ? ctcmath::expf_core_f64(odd ? 0.0f : v, tbl) : (double)v;
    }
    const double sum = carry + wave_scan_f64_sum(p);
    const bool near = i < kept && (odd || fabs(sum - a.cut_exp) <= 1e-8 * a.cut_exp || !(sum == sum));
    const bool hit = i < kept && (sum >= a.cut_exp || i + 1 >= a.top_n);
    const unsigned long long mh = __ballot(hit), mn = __ballot(near);
    const int firsthit = mh ? __ffsll((long long)mh) - 1 : 64;
    if (mn && (__ffsll((long long)mn) - 1) <= firsthit) flag = true;
    if (mh) stop = i0 + firsthit + 1;
    carry = f64_from_lane(sum, 63);
  }
  flag = __ballot(flag) != 0ull;
  return stop;
}

// (workgroups per CU: the pass is bound by the latency chain of a frame inside a workgroup, so what counts is how many frames a CU
//  has in flight.  Rounds 2-5, with the device library's exp / log in the cumulative cut (88 VGPRs at the compiler's own choice):
//  0.418 ms at five, 0.349 at six, 0.39 / 0.43 at seven / eight, where the register budget cost more than the extra frames brought.
//  Round 6, without them (63 VGPRs): 0.354 ms at six, 0.336 at seven, 0.323 at eight -- profiles/r06i_prune_variants.txt)
#ifndef CTC_PRUNE_WG_OCC
#define CTC_PRUNE_WG_OCC 8
#endif
// REG (round 6, late): the row stays in registers between the two looks at it (4 * F4 VGPRs; fewer workgroups per CU) instead of
// being read again -- the second sweep fetched four lines in five once more (a line serves eight threads, one thread in five
// looks again): counter traffic 1.78x the row.
#ifndef CTC_PRUNE_REG_OCC
#define CTC_PRUNE_REG_OCC 5
#endif
template <int F4, bool REG = false>
__global__ void __launch_bounds__(256, REG ? (F4 <= 4 ? 8 : F4 <= 10 ? CTC_PRUNE_REG_OCC : 3) : CTC_PRUNE_WG_OCC) prune_rows_wg_kernel(PruneArgs a) {
  extern __shared__ __attribute__((aligned(16))) char psm[];
  __shared__ uint32_t s_bound[4];
  __shared__ int s_cnt;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = a.top_n < a.V ? a.top_n : a.V;
  const int nq = (n + 3) >> 2;  // per-wave share
  // LDS: the kept values in final order (for the cumulative cut) | spare | (unused) | candidate keys | their labels | their values
  float *sval = (float *)psm;
  uint32_t *ckey = (uint32_t *)(sval + ((3 * a.stride + 3) & ~3));  // (16-byte aligned: read four keys at a time)
  int *cidx = (int *)ckey + kPruneCand;
  float *cval = (float *)(cidx + kPruneCand);
  const int nv4 = a.V >> 2;
  constexpr int kChunk = F4 < 5 ? F4 : 5;  // 128-bit loads in flight per thread (measured: 2, 4 and 10 are slower)
  for (long long r = blockIdx.x; r < a.rows; r += gridDim.x) {
    if (a.seq_lens) {
      const long long b = r / a.T;
      int len = a.seq_lens[b];
      len = len < 0 ? 0 : len;
      if ((int)(r - b * a.T) >= len) continue;
    }
    const float *x = a.in + (size_t)r * a.V;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    if (tid == 0) s_cnt = 0;
    uint32_t lmax = 0;
    float4 vr[REG ? F4 : 1];
    int tq = tid;
    if (REG) {
      asm volatile("" : "+v"(tq));  // (opaque per frame: prune_logits_wg_kernel says why)
#pragma unroll
      for (int u = 0; u < F4; ++u) {
        const int i4 = tq + 256 * u;
        vr[u] = i4 < nv4 ? x4[i4] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      }
#pragma unroll
      for (int u = 0; u < F4; ++u) {
        const int i4 = tq + 256 * u;
        if (i4 < nv4) {
          const uint32_t k0 = prune_key(vr[u].x), k1 = prune_key(vr[u].y), k2 = prune_key(vr[u].z), k3 = prune_key(vr[u].w);
          const uint32_t m01 = k0 > k1 ? k0 : k1, m23 = k2 > k3 ? k2 : k3, m = m01 > m23 ? m01 : m23;
          lmax = m > lmax ? m : lmax;
        }
      }
    }
    for (int u0 = 0; !REG && u0 < F4; u0 += kChunk) {
      float4 v[kChunk];
#pragma unroll
      for (int u = 0; u < kChunk; ++u) {
        const int i4 = tid + 256 * (u0 + u);
        v[u] = (u0 + u < F4 && i4 < nv4) ? x4[i4] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      }
#pragma unroll
      for (int u = 0; u < kChunk; ++u) {
        const int i4 = tid + 256 * (u0 + u);
        if (u0 + u < F4 && i4 < nv4) {
          const uint32_t k0 = prune_key(v[u].x), k1 = prune_key(v[u].y), k2 = prune_key(v[u].z), k3 = prune_key(v[u].w);
          const uint32_t m01 = k0 > k1 ? k0 : k1, m23 = k2 > k3 ? k2 : k3, m = m01 > m23 ? m01 : m23;
          lmax = m > lmax ? m : lmax;
        }
      }
    }
    uint32_t bw = 0;
    for (int bit = 31; bit >= 12; --bit) {
      const uint32_t trial = bw | (1u << bit);
      if (__popcll(__ballot(lmax >= trial)) >= nq) bw = trial;
    }
    if (lane == 0) s_bound[wave] = bw;
    __syncthreads();
    uint32_t bound = s_bound[0];
    bound = s_bound[1] < bound ? s_bound[1] : bound;
    bound = s_bound[2] < bound ? s_bound[2] : bound;
    bound = s_bound[3] < bound ? s_bound[3] : bound;
    if (REG) {
      if (lmax >= bound && lmax != 0u) {
#pragma unroll
        for (int u = 0; u < F4; ++u) {
          const int i4 = tq + 256 * u;
          if (i4 < nv4) {
            const float4 v = vr[u];
            const uint32_t kk[4] = {prune_key(v.x), prune_key(v.y), prune_key(v.z), prune_key(v.w)};
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (kk[e] >= bound && kk[e] != 0u) {
                const int p = atomicAdd(&s_cnt, 1);
                if (p < kPruneCand) { ckey[p] = kk[e]; cidx[p] = 4 * i4 + e; cval[p] = e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }
              }
          }
        }
      }
    } else if (lmax >= bound && lmax != 0u) {  // (only the few threads that hold a value at or above the bound sweep again)
      for (int u = 0; u < F4; ++u) {
        const int i4 = tid + 256 * u;
        if (i4 >= nv4) break;
        const float4 v = x4[i4];
        const uint32_t kk[4] = {prune_key(v.x), prune_key(v.y), prune_key(v.z), prune_key(v.w)};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (kk[e] >= bound && kk[e] != 0u) {
            const int p = atomicAdd(&s_cnt, 1);
            if (p < kPruneCand) { ckey[p] = kk[e]; cidx[p] = 4 * i4 + e; cval[p] = e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }
          }
      }
    }
    __syncthreads();
    const int ns = s_cnt;
    if (wave == 0) {
      bool flag = ns > kPruneCand;  // more values above the bound than the list holds: prune_resolve_kernel decides this frame
      int kept = 0;
      int *och = a.ch + (size_t)r * a.stride;
      float *olp = a.lp + (size_t)r * a.stride;
      if (!flag) {
        // every lane ranks its own candidates among all of them: rank = #greater; the n-th largest has rank < n <= rank + #equal
        // (everything this wave needs from here on sits in LDS -- the other three waves of the workgroup wait for it, so a
        //  global access or a chain of dependent LDS reads here is paid by the whole row: the keys are read four at a
        //  time, the values come from the list the second sweep filled)
        for (int p = ns + lane; p < ((ns + 3) & ~3); p += 64) ckey[p] = 0u;  // pad to a multiple of four, below every real key
        __builtin_amdgcn_s_waitcnt(0xc07f);
        for (int q0 = 0; q0 < ns; q0 += 64) {
          const int q = q0 + lane;
          const uint32_t mine = q < ns ? ckey[q] : 0xFFFFFFFFu;
          int gg = 0, ee = 0;
          for (int o = 0; o < ns; o += 4) {
            const uint4 k4 = *reinterpret_cast<const uint4 *>(ckey + o);
            gg += (k4.x > mine) + (k4.y > mine) + (k4.z > mine) + (k4.w > mine);
            ee += (k4.x == mine) + (k4.y == mine) + (k4.z == mine) + (k4.w == mine);
          }
          const bool keep = q < ns && gg < n;
          if (keep && gg + ee > n) flag = true;   // equal values straddle the cut: std::sort decides which of them are kept
          if (keep && ee > 1) flag = true;        // equal kept values: their order is std::sort's business
          if (keep && ee == 1) {
            const int idx = cidx[q];
            float v = cval[q];
            if (!a.log_input) {  // decoder_utils.cpp:42
              v = (float)ctcmath::log_f64((double)v + (double)FLT_MIN, g_t64);
            }
            och[gg] = idx; olp[gg] = v; sval[gg] = cval[q];  // (sval: the row's own value, before any prob -> log conversion)
          }
          kept += __popcll(__ballot(keep));
        }
        if (kept > n) kept = n;
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes are visible to its other lanes
      flag = __ballot(flag) != 0ull;
      int len = kept;
      if (a.cutoff_prob < 1.0 && !flag) len = prune_cumulative_cut(a, sval, nullptr, kept, lane, flag, a.tables);
      if (lane == 0) {
        a.cnt[r] = len;
        if (flag) {
          const unsigned k = atomicAdd(a.n_flag, 1u);
          if (k < a.flag_cap) a.flag_rows[k] = (unsigned)r;
        }
      }
    }
    __syncthreads();  // the lists are reused by the next frame
  }
}

// Raw logits (log_input == 2) straight into the prune: the frame's log-softmax is never written.  One workgroup per frame as
// above, but the row -- read once, 128-bit loads -- stays in registers: its maximum and the lower bound of the n-th largest
// logit come from the first look at it, the sum of exponentials in the order ctcd_log_softmax defines from wg_exp_sum, and only
// the few dozen labels at or above the bound are ever normalised: y = (x - m) - ls, ranked on y (two logits can round to one y).
// x -> y is monotone, so a label below the bound cannot outrank one above it; it could TIE with the n-th largest y if that
// equals the bound's own image -- such a frame is flagged, as are frames that hold a NaN or +inf (no order argument there), and
// prune_resolve_kernel settles flagged frames from the logits with the frame's (m, ls) stored here.
// Requires V % 4 == 0, V <= 1024 * F4, cutoff_top_n <= 64.
__device__ __forceinline__ float prune_key_value(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
#ifndef CTC_PRUNE_LOGITS_OCC
#define CTC_PRUNE_LOGITS_OCC 5
#endif
template <int F4>
__global__ void __launch_bounds__(256, F4 <= 4 ? 6 : F4 <= 10 ? CTC_PRUNE_LOGITS_OCC : 3) prune_logits_wg_kernel(PruneArgs a, const uint64_t *tables) {
  extern __shared__ __attribute__((aligned(16))) char psm[];
  __shared__ uint64_t tbl[64];
  __shared__ __attribute__((aligned(16))) LsmLds s;
  __shared__ uint32_t s_bound[4];
  __shared__ int s_cnt;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = a.top_n < a.V ? a.top_n : a.V;
  const int nq = (n + 3) >> 2;  // per-wave share
  float *sval = (float *)psm;
  uint32_t *ckey = (uint32_t *)(sval + ((3 * a.stride + 3) & ~3));
  int *cidx = (int *)ckey + kPruneCand;
  float *cval = (float *)(cidx + kPruneCand);
  const int nv4 = a.V >> 2;
  if (tid < 64) tbl[tid] = tables[tid];
  __syncthreads();
  for (long long r = blockIdx.x; r < a.rows; r += gridDim.x) {
    if (a.seq_lens) {
      const long long b = r / a.T;
      int len = a.seq_lens[b];
      len = len < 0 ? 0 : len;
      if ((int)(r - b * a.T) >= len) continue;
    }
    const float4 *x4 = reinterpret_cast<const float4 *>(a.in + (size_t)r * a.V);
    // (the thread index, opaque per frame: derived from the hoisted one, every chunk's predicate and offsets would be kept in
    //  registers across the frame loop -- thirty-odd of them, spilled)
    int tq = tid;
    asm volatile("" : "+v"(tq));
    float4 v[F4];
#pragma unroll
    for (int u = 0; u < F4; ++u) {
      const int i4 = tq + 256 * u;
      v[u] = i4 < nv4 ? x4[i4] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    }
    if (tid == 0) s_cnt = 0;
    // (maxima on the values themselves -- a NaN is skipped, as by the one-wave kernel; the key of the thread's maximum is the
    //  maximum of its keys whenever the row holds no NaN, and rows that do are flagged below)
    float mine = -INFINITY;
#pragma unroll
    for (int u = 0; u < F4; ++u) mine = fmaxf(fmaxf(fmaxf(mine, v[u].x), fmaxf(v[u].y, v[u].z)), v[u].w);
    const uint32_t lmax = prune_key(mine);
    uint32_t bw = 0;
    for (int bit = 31; bit >= 12; --bit) {
      const uint32_t trial = bw | (1u << bit);
      if (__popcll(__ballot(lmax >= trial)) >= nq) bw = trial;
    }
    if (lane == 0) s_bound[wave] = bw;
    const float m = wg_row_max(mine, s, tid);  // (contains a barrier: s_bound is in place)
    uint32_t bound = s_bound[0];
    bound = s_bound[1] < bound ? s_bound[1] : bound;
    bound = s_bound[2] < bound ? s_bound[2] : bound;
    bound = s_bound[3] < bound ? s_bound[3] : bound;
    float ls = 0.0f;
    bool rowflag = !(m > -INFINITY);
    if (!rowflag) {
      const float sum = wg_exp_sum<F4>(v, nv4, m, s, tbl, tq);
      rowflag = !(sum == sum);  // a NaN or +inf among the logits: (x - m) is a NaN for it, and so is the sum
      ls = ctcmath::logf_normal(sum, tbl);
    }
    if (tid == 0) { a.row_max[r] = m; a.row_lse[r] = ls; }
    // the labels at or above the bound, compared as values: for numbers, x >= value(bound) <=> key(x) >= bound (the key is monotone
    // and value(bound)'s key is bound itself); a bound below every number's key decodes to a NaN pattern: everything is listed
    const float bf = prune_key_value(bound);
    const bool all = bound <= prune_key(-INFINITY);
    if (!rowflag && (all || mine >= bf)) {
#pragma unroll
      for (int u = 0; u < F4; ++u) {
        const int i4 = tq + 256 * u;
        if (i4 < nv4) {
          const float xs[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (all || xs[e] >= bf) {
              const int p = atomicAdd(&s_cnt, 1);
              if (p < kPruneCand) { const float y = (xs[e] - m) - ls; ckey[p] = prune_key(y); cidx[p] = 4 * i4 + e; cval[p] = y; }
            }
          }
        }
      }
    }
    __syncthreads();
    const int ns = s_cnt;
    if (wave == 0) {
      bool flag = rowflag || ns > kPruneCand;
      int kept = 0;
      int *och = a.ch + (size_t)r * a.stride;
      float *olp = a.lp + (size_t)r * a.stride;
      if (!flag) {
        // the image of the bound: no label outside the list has a larger y
        const uint32_t ybkey = bound ? prune_key((prune_key_value(bound) - m) - ls) : 0u;
        for (int p = ns + lane; p < ((ns + 3) & ~3); p += 64) ckey[p] = 0u;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        for (int q0 = 0; q0 < ns; q0 += 64) {
          const int q = q0 + lane;
          const uint32_t mine_k = q < ns ? ckey[q] : 0xFFFFFFFFu;
          int gg = 0, ee = 0;
          for (int o = 0; o < ns; o += 4) {
            const uint4 k4 = *reinterpret_cast<const uint4 *>(ckey + o);
            gg += (k4.x > mine_k) + (k4.y > mine_k) + (k4.z > mine_k) + (k4.w > mine_k);
            ee += (k4.x == mine_k) + (k4.y == mine_k) + (k4.z == mine_k) + (k4.w == mine_k);
          }
          const bool keep = q < ns && gg < n;
          if (keep && gg + ee > n) flag = true;   // equal values straddle the cut: std::sort decides which of them are kept
          if (keep && ee > 1) flag = true;        // equal kept values: their order is std::sort's business
          if (keep && mine_k <= ybkey) flag = true;  // a label outside the list could tie with this one
          if (keep && ee == 1) { och[gg] = cidx[q]; olp[gg] = cval[q]; sval[gg] = cval[q]; }
          kept += __popcll(__ballot(keep));
        }
        if (kept > n) kept = n;
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      flag = __ballot(flag) != 0ull;
      int len = kept;
      if (a.cutoff_prob < 1.0 && !flag) len = prune_cumulative_cut(a, sval, nullptr, kept, lane, flag, tbl);
      if (lane == 0) {
        a.cnt[r] = len;
        if (flag) {
          const unsigned k = atomicAdd(a.n_flag, 1u);
          if (k < a.flag_cap) a.flag_rows[k] = (unsigned)r;
        }
      }
    }
    __syncthreads();  // the lists are reused by the next frame
  }
}

// Flagged frames are settled here, on the device.  Most flags are ties: equal values at or above the cut, whose order (and,
// at the cut, which of them are kept) is whatever std::sort leaves behind -- a function of the whole row.  One workgroup
// per flagged frame replays that std::sort call (decoder_utils.cpp:19-20: (index, double) pairs in index order, compared
// on the value alone) with stl_emul.h's workgroup-parallel introsort, takes the first min(top_n, V) pairs, converts them
// with the bit-exact binary64 log and walks the cumulative cut as the reference does: a sequential chain of
// log_sum_exp<double> (prune_exact_cut).  Nothing is left for the host (rounds 1-3 sent borderline libm roundings, NaN rows
// and rows too long for the workgroup's LDS there: the first are exact now, the second defined -- prune_key --, the third
// sort in a per-workgroup block of global memory, slowly).
struct WgSortX {
  uint32_t *wsum;  // one word per wave (shared)
  // exclusive prefix (in thread order) and total of one word per thread; contains a barrier
  __device__ __forceinline__ int lanes() const { return 64; }
  __device__ __forceinline__ uint64_t ballot(bool p) const { return __builtin_amdgcn_ballot_w64(p); }
  __device__ __forceinline__ int count(uint64_t m) const { return __builtin_popcountll(m); }
  __device__ __forceinline__ int count_below(uint64_t m) const { return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
  __device__ __forceinline__ uint32_t first_lane(uint32_t v) const { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
  __device__ __forceinline__ void block_scan_u32(uint32_t mine, uint32_t *base_out, uint32_t *total_out) {
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6, nw = ((int)blockDim.x + 63) >> 6;
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int q = 0; q < nw; ++q) { const uint32_t t = wsum[q]; if (q < wave) base += t; tot += t; }
    *base_out = base + incl - mine;
    *total_out = tot;
    __syncthreads();  // (wsum is reused by the next round)
  }
  __device__ __forceinline__ int tid() const { return (int)threadIdx.x; }
  __device__ __forceinline__ int nt() const { return (int)blockDim.x; }
  __device__ __forceinline__ void sync() { __syncthreads(); }
  __device__ __forceinline__ int atomic_add(int *p, int v) { return atomicAdd(p, v); }
  __device__ __forceinline__ int uni(int v) const { return __builtin_amdgcn_readfirstlane(v); }
};
__host__ __device__ inline int prune_resolve_task_cap(int V) { return V / 17 + 2; }
constexpr int kResolveBigCut = 256;    // ranges longer than this are split by the whole workgroup
constexpr int kResolveThreads = 1024;  // (the row fills most of a CU's LDS: one workgroup per CU whatever its size)
constexpr int kResolveMaxV = 65535;    // 16-bit positions and counters
__host__ __device__ inline size_t prune_resolve_lds_bytes(int V, int n) {
  // pairs | Lp, Rp | two task lists | final ranges | stack of long ranges | counters | the kept labels
  return (size_t)V * 8 + (size_t)2 * (V + 2) * 2 + (size_t)6 * prune_resolve_task_cap(V) * 2 + (size_t)2 * (V / 2 + 1) * 2 + 3 * 64 * 4 + 64 +
         (size_t)n * 4 + 64;
}
// decoder_utils.cpp:25-32 to the letter, by one thread: cum = log_sum_exp<double>(cum, log p_i) from cum = 0.0 over the sorted
// candidates until cum >= cutoff_prob or cutoff_top_n of them are taken; every log / exp the bit-exact one.
__device__ int prune_exact_cut(const PruneArgs &a, const float *row, const int *sidx, int n, float m, float ls) {
  double cum = 0.0;
  int keep = 0;
  for (int i = 0; i < n; ++i) {
    const double v = (double)prune_row_value(a, row[sidx[i]], m, ls);
    cum = ctcmath::lse_f64(cum, a.log_input ? v : ctcmath::log_f64(v, g_t64), g_t64);
    ++keep;
    if (cum >= a.cutoff_prob || keep >= a.top_n) break;
  }
  return keep;
}

// far != nullptr: the sort's arrays do not fit the workgroup's LDS (vocabularies beyond ~11 000 labels): they live in a block
// of global memory per workgroup (far_stride bytes each; every barrier of the sort is a full fence already).
__global__ void __launch_bounds__(kResolveThreads) prune_resolve_kernel(PruneArgs a, char *far, size_t far_stride) {
  extern __shared__ __attribute__((aligned(16))) char rsm_lds[];
  __shared__ uint32_t s_wsum[kResolveThreads / 64];
  char *rsm = far ? far + (size_t)blockIdx.x * far_stride : rsm_lds;
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int V = a.V, n = a.top_n < V ? a.top_n : V;
  const int tcap = prune_resolve_task_cap(V);
  unsigned long long *v = (unsigned long long *)rsm;
  uint16_t *Lp = (uint16_t *)(v + V), *Rp = Lp + (V + 2), *cur = Rp + (V + 2), *nxt = cur + 3 * tcap, *small = nxt + 3 * tcap;
  int *bstack = (int *)(((uintptr_t)(small + 2 * (V / 2 + 1)) + 15) & ~(uintptr_t)15);
  int *cnt = bstack + 3 * 64, *sidx = cnt + 4;
  const unsigned raw = *a.n_flag, nf = raw < a.flag_cap ? raw : a.flag_cap;
  WgSortX x{s_wsum};
  for (unsigned k = blockIdx.x; k < nf; k += gridDim.x) {
    const long long r = (long long)a.flag_rows[k];
    const float *row = a.in + (size_t)r * V;
    const float m = a.log_input == 2 ? a.row_max[r] : 0.0f, ls = a.log_input == 2 ? a.row_lse[r] : 0.0f;  // (raw logits: prune_logits_wg_kernel)
    __syncthreads();
    for (int i = tid; i < V; i += kResolveThreads) v[i] = ((unsigned long long)prune_key(prune_row_value(a, row[i], m, ls)) << 32) | (unsigned)i;
    __syncthreads();
    // decoder_utils.cpp:19-20: (index, double) pairs in index order, std::sort on the value alone, descending
    // (stl_emul.h: long ranges split by the whole workgroup, the rest by one thread per range; only the ranges that reach
    //  the first n places -- the prune pass keeps the best top_n of a row)
    stlemu::sort_prefix_parallel(x, v, V, n, kResolveBigCut, [](unsigned long long e) { return (uint32_t)(e >> 32); }, Lp, Rp, cur, nxt, small,
                                 cnt, bstack);
    if (tid < 64) {
      int *och = a.ch + (size_t)r * a.stride;
      float *olp = a.lp + (size_t)r * a.stride;
      for (int q = lane; q < n; q += 64) {
        const int idx = (int)(uint32_t)v[q];
        float val = prune_row_value(a, row[idx], m, ls);
        if (!a.log_input) val = (float)ctcmath::log_f64((double)val + (double)FLT_MIN, g_t64);  // decoder_utils.cpp:42
        och[q] = idx; olp[q] = val; sidx[q] = idx;
      }
      __threadfence_block();  // the wave's own writes (LDS or global) are visible to its lane 0
      if (lane == 0) a.cnt[r] = a.cutoff_prob < 1.0 ? prune_exact_cut(a, row, sidx, n, m, ls) : n;
    }
  }
}

// Compact results (beam_core.h OutRefs::c_*) -> the reference's tensors: tokens / timesteps [B, K, T], zero outside the
// valid prefixes.  HBM-bound (it writes 8 B per label position of the padded pair).  One workgroup per item; every wave
// takes whole rows, its lanes consecutive label positions (coalesced 256-byte stores).  Label q of DFS entry j is held
// by the nearest entry o <= j with lcp[o] <= q; owner[] (the nearest earlier entry that shares LESS with its own
// predecessor) lets the search skip runs of entries, as in the decode kernel's own back-trace.
constexpr int kExpandMaxK = 4096;  // entries whose tables fit LDS; wider beams take the generic slow loop
__global__ void __launch_bounds__(1024) expand_compact_kernel(const int32_t *hdr, const int32_t *ent, const uint32_t *rag, int K, int T,
                                                              int32_t *tok, int32_t *ts) {
  extern __shared__ int esm[];
  int *lcp = esm, *dep = esm + K, *off = esm + 2 * K, *row = esm + 3 * K, *owner = esm + 4 * K;
  unsigned *used = (unsigned *)(esm + 5 * K);  // K bits
  const int b = (int)blockIdx.x;
  const int nres = hdr[(size_t)b * 4];
  int32_t *tk0 = tok + (size_t)b * K * T, *ts0 = ts + (size_t)b * K * T;
  for (int i = (int)threadIdx.x; i < (K + 31) / 32; i += (int)blockDim.x) used[i] = 0u;
  for (int j = (int)threadIdx.x; j < nres; j += (int)blockDim.x) {
    const int4 e = *reinterpret_cast<const int4 *>(ent + ((size_t)b * K + j) * 4);
    row[j] = e.x; lcp[j] = j == 0 ? 0 : e.y; dep[j] = e.z; off[j] = e.w;
  }
  __syncthreads();
  for (int j = (int)threadIdx.x; j < nres; j += (int)blockDim.x) {
    int o = j - 1;
    const int l = lcp[j];
    while (o >= 0 && lcp[o] >= l) --o;
    owner[j] = o < 0 ? 0 : o;
    atomicOr(&used[row[j] >> 5], 1u << (row[j] & 31));
  }
  __syncthreads();
  const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63, nw = (int)blockDim.x >> 6;
  for (int j = wave; j < K; j += nw) {
    if (j < nres) {
      int32_t *tk = tk0 + (size_t)row[j] * T, *tt = ts0 + (size_t)row[j] * T;
      const int dj = dep[j];
      for (int q = lane; q < T; q += 64) {
        int32_t c = 0, sv = 0;
        if (q < dj) {
          int o = j;
          while (lcp[o] > q) o = owner[o];  // (lcp[0] = 0 ends the walk)
          const uint32_t v = rag[(uint32_t)off[o] + (uint32_t)(q - lcp[o])];
          c = (int32_t)(v & 0xFFFFu);
          sv = (int32_t)(v >> 16);
        }
        tk[q] = c;
        tt[q] = sv;
      }
    }
    // rows without a result: row index j doubles as the row to check
    if (!((used[j >> 5] >> (j & 31)) & 1u)) {
      int32_t *tk = tk0 + (size_t)j * T, *tt = ts0 + (size_t)j * T;
      for (int q = lane; q < T; q += 64) { tk[q] = 0; tt[q] = 0; }
    }
  }
}

__global__ void debug_math_kernel(int mode, uint32_t start, uint32_t stride, const float *xs, const float *ys, float *out,
                                  size_t n, const uint64_t *tables) {
  __shared__ uint64_t tbl[64];
  if (threadIdx.x < 64) tbl[threadIdx.x] = tables[threadIdx.x];
  __syncthreads();
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (mode == 2) {
    out[i] = ctcmath::lse(xs[i], ys[i], tbl);
  } else {
    const float x = ctcmath::bits_to_f32(start + (uint32_t)i * stride);
    out[i] = mode == 0 ? ctcmath::expf_nonpos(x, tbl) : ctcmath::logf_normal(x, tbl);
  }
}

// binary64 log / exp / log_sum_exp of exact_math_f64.h on float images (modes 3..6 of ctcd_debug_math_check)
__global__ void debug_math64_kernel(int mode, uint32_t start, uint32_t stride, const float *xs, const float *ys, uint64_t *out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double r;
  if (mode == 6) {
    r = ctcmath::lse_f64((double)xs[i], (double)ys[i], g_t64);
  } else {
    const double x = (double)ctcmath::bits_to_f32(start + (uint32_t)i * stride);
    r = mode == 3 ? ctcmath::log_f64(x, g_t64) : mode == 4 ? ctcmath::log_f64(x + (double)FLT_MIN, g_t64) : ctcmath::exp_f64(x, g_t64);
  }
  out[i] = ctcmath::f64_to_bits(r);
}

// ------------------------------------------------------------------------------------------------ host side
thread_local std::string g_err;
int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}
#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) return fail(CTCD_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));    \
  } while (0)

// Every C ABI entry point works on the decoder's device and puts the caller's current device back on exit (a process
// that uses several GPUs must not find PyTorch's current device switched by a decode -- or by a destructor that the
// garbage collector runs at an arbitrary time).
struct DeviceGuard {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur != dev) prev = cur;
    if (cur != dev) err = hipSetDevice(dev);
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard &) = delete;
  DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define CTC_ON_DEVICE(dev)  \
  DeviceGuard guard_(dev);  \
  HIP_TRY(guard_.err)

struct Buf {
  void *p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return CTCD_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return fail(CTCD_EHIP, std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e));
    cap = bytes;
    return CTCD_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

}  // namespace

// A few host threads that stay parked between calls (expanding 2 x [B, K, T] on one thread would take longer than the decode)
struct HostPool {
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv, done_cv;
  std::function<void(int)> job;
  int n_items = 0, next = 0, running = 0;
  unsigned long long epoch = 0;
  bool stop = false;
  void start(int n) {  // (also grows a running pool: the new threads start behind the jobs already done)
    unsigned long long e;
    { std::lock_guard<std::mutex> g(mu); e = epoch; }
    for (int i = 0; i < n; ++i) th.emplace_back([this, e] { loop(e); });
  }
  void loop(unsigned long long seen) {
    for (;;) {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return stop || epoch != seen; });
      if (stop) return;
      seen = epoch;
      for (;;) {
        if (next >= n_items) break;
        const int i = next++;
        lk.unlock();
        job(i);
        lk.lock();
      }
      if (--running == 0) done_cv.notify_all();
    }
  }
  void run(int items, std::function<void(int)> f) {
    if (th.empty() || items <= 1) { for (int i = 0; i < items; ++i) f(i); return; }
    std::unique_lock<std::mutex> lk(mu);
    job = std::move(f); n_items = items; next = 0; running = (int)th.size(); ++epoch;
    cv.notify_all();
    done_cv.wait(lk, [&] { return running == 0; });
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> g(mu); stop = true; }
    cv.notify_all();
    for (auto &t : th) t.join();
  }
};

struct ctcd_decoder {
  int device = 0;
  Buf c_hdr, c_ent, c_rag, c_cnt, c_sc, c_ln;  // compact results of the host-tensor entry points
  void *h_stage = nullptr;  // page-locked staging for them
  size_t h_stage_cap = 0;
  HostPool *workers = nullptr;
  int threads = 0;  // 0 = choose per call from the number of candidate slots
  int max_lds = 0;
  int cu_count = 0;
  long long lds_floor = -1;  // CTCD_LDS_FLOOR (experiments): dynamic LDS bytes every decode launch asks for at least
  int cu_sharing = -1;       // ctcd_set_cu_sharing: 1 = always launch the two-workgroups-per-CU build, 0 = never, -1 = when B > #CUs
  Buf pool, status, tables, logp, lsm, flags, stage_in, stage_out, pr_cnt, pr_ch, pr_lp, pr_ml, far, st_args;
  Buf prune_in, prune_out, st_lens;  // own staging: the host-pointer entry points keep their tensors in stage_in/out
  Buf cb_blocks, cb_ctl;             // scorer hook: the temporary streams' blocks of a one-shot decode; per-item control words
  int32_t *h_cb = nullptr;           // ... and, page-locked, what a round reports: [status | frames done | #misses]
  size_t h_cb_items = 0;
  long long prune_flagged_rows = 0;  // frames of the last call the fast prune pass flagged (settled by prune_resolve_kernel)
  unsigned *h_flagged = nullptr;     // page-locked: that count, copied behind the kernels
  bool flagged_pending = false;
  bool resolve_in_global = false;    // tests: the std::sort replay's arrays in global memory whatever the row length
  bool tables_ready = false;
  bool timing = false;
  bool profile = false, dbg_on = false;
  bool no_fixed_layout = false;  // debugging: always use the run-time workspace layout
  bool no_prune_reg = false;     // debugging / tests: the two-sweep form of the workgroup prune kernel
  bool no_hook_wait = false;     // tests: a callback scorer's launches end at a miss (the form of rounds 4-5) instead of waiting for the answer
  int last_cb_waits = 0;         // answer batches the last call's launches were handed while they waited
  bool no_fused_logits = false;  // tests: raw logits always through the one-wave log_softmax pass and the separate prune
  Buf prof, dbg, tl;
  int tl_f0 = 0, tl_nf = 0;
  int status_items = 0;          // items of the launch whose status words (and shape statistic) are in d->status
  bool tl_armed = false;
  // phase A1's four-entries-per-wave subtree search (decode_kernel.h kQuarters): -1 = chosen per launch from the beam shape the
  // last checked launch reported (chains: on), 0 / 1 = forced
  int subtree_mode = -1;
  bool subtree_on = false;       // the automatic choice for the next launch
  int last_subtree_search = 0;   // what the last launch used
  int last_cb_rounds = 0;        // scorer hook: launches the last decode through a callback scorer took (ctcd_last_scorer_rounds)
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;  // decode kernel | vocabulary-prune pass
  bool prune_timed = false;
  // streamed input of the host-tensor entry point: rows cross PCIe frame block by frame block while the kernel runs
  void *fg_in = nullptr;          // fine-grained (uncached) device memory: [frames-arrived counter | rows | seq_lens]
  size_t fg_in_cap = 0;
  int *h_cnt = nullptr;           // page-locked: the counter values the copy stream writes behind each block
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_in = nullptr;
  bool no_input_streaming = false;
  bool stall_input_test = false;
  bool general_lm_kernel = false;  // CTCD_GENERAL_LM_KERNEL=1: word models, too, run the scorer instantiations that serve every model
  bool input_timed_out = false;   // set by ctcd_check_status when an utterance reports ST_INPUT_TIMEOUT
  long long mirror_cap_override = -1;  // tests: labels the host mirror of the compact results holds (-1: a third of the worst case)
  hipStream_t last_stream = nullptr;  // the stream of the last decode launch (ctcd_check_status reads the status words on it)
  // streaming calls: their per-item arguments (block pointers, pool capacities, end-of-stream flags, chunk lengths) travel
  // in ONE copy from page-locked staging -- two slots, an event each: a chunk call costs no stream synchronisation
  char *h_stargs[2] = {nullptr, nullptr};
  size_t h_stargs_cap = 0;
  hipEvent_t ev_stargs[2] = {nullptr, nullptr};
  unsigned long long stream_calls = 0;
  std::mutex mu;
  std::mutex mu_host;  // the host-tensor entry points: compact buffers, page-locked staging and the worker threads are per decoder
};

// One audio stream's parked decoder state (ctcd_stream_*): a single HBM block [header | beam arrays | node pool].
struct ctcd_stream {
  unsigned long long seen_in_call = 0;  // ctcd_stream_decode: "this state is already part of the current call" in O(1)
  int device = 0;
  ctcd_scorer *scorer = nullptr;  // DecoderState is created with its scorer (binding.cpp:243-261)
  char *block = nullptr;
  size_t bytes = 0;
  int V = 0, beam = 0;
  long long frames = 0;      // frames fed so far (host mirror of the header word)
  long long cap_frames = 0;  // frames the node pool can take
};

// The external scorer (ctcdecode/src/scorer.h:41-110, created by paddle_get_scorer, binding.cpp:143-150): built on the host
// from the ARPA file (lm_build.h), its tables mirrored into the HBM of one device (lm_tables.h).
// Helper threads of a callback scorer whose callback may be called from several threads at once (ctcd_scorer_set_callback_threads): while a
// waiting launch is being served they spin on a generation counter, take every P-th window of the batch the serving thread publishes and
// report once per batch; between launches they sleep.  The serving thread takes its own share, so P threads ask P windows at a time.
struct CbAskPool {
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv;
  bool active = false, stop = false;           // (under mu)
  std::atomic<unsigned long long> gen{0};      // batches published
  std::atomic<int> acks{0};                    // helpers done with the current batch
  ctclm::CallbackLm *cl = nullptr;
  ctclm::CallbackLm::Ask *batch = nullptr;
  int n = 0;
  int parts() const { return (int)th.size() + 1; }
  void worker(int w) {
    unsigned long long seen = gen.load(std::memory_order_acquire);
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return active || stop; });
        if (stop) return;
      }
      for (unsigned spin = 0;; ++spin) {
        const unsigned long long g = gen.load(std::memory_order_acquire);
        if (g != seen) {
          seen = g;
          const int P = parts();
          for (int i = w; i < n; i += P) cl->ask(batch[i]);
          acks.fetch_add(1, std::memory_order_release);
          spin = 0;
        } else if ((spin & 4095) == 4095) {
          std::lock_guard<std::mutex> lk(mu);
          if (!active || stop) break;
        }
        __builtin_ia32_pause();
      }
    }
  }
  void start(int threads, ctclm::CallbackLm *c) {
    shutdown();
    cl = c;
    stop = false;
    for (int w = 1; w < threads; ++w) th.emplace_back([this, w] { worker(w); });
  }
  void shutdown() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; active = false; }
    cv.notify_all();
    for (auto &t : th) t.join();
    th.clear();
  }
  void begin() { if (th.empty()) return; { std::lock_guard<std::mutex> lk(mu); active = true; } cv.notify_all(); }
  void end() { if (th.empty()) return; std::lock_guard<std::mutex> lk(mu); active = false; }
  // asks batch[0 .. n): this thread's share here, the helpers' shares beside it (all of them in this thread when the batch is short)
  void ask_all(ctclm::CallbackLm::Ask *b, int count) {
    if (th.empty() || count < 4) { for (int i = 0; i < count; ++i) cl->ask(b[i]); return; }
    batch = b; n = count;
    acks.store(0, std::memory_order_relaxed);
    gen.fetch_add(1, std::memory_order_release);
    const int P = parts();
    for (int i = 0; i < count; i += P) cl->ask(b[i]);
    while (acks.load(std::memory_order_acquire) != P - 1) __builtin_ia32_pause();
  }
};

struct ctcd_scorer {
  ctclm::HostScorer host;
  int device = 0;
  char *blob = nullptr;      // one HBM allocation holding every table
  ctclm::LmView dview;       // the tables as the kernel sees them (alpha / beta are refreshed at every launch)
  // Host-side scorer hook (ctcd_scorer_create_callback, lm_callback.h): the device tables are a cache of the callback's
  // answers -- grown between launches (cb_sync); `host` keeps the scalar facts the accessors report
  ctclm::CallbackLm *cbl = nullptr;
  char *cb_ng = nullptr, *cb_st = nullptr, *cb_uni = nullptr, *cb_miss = nullptr;  // cache slots | zeroed state arrays | NaN unigrams | miss list + counter
  size_t cb_ng_slots = 0, cb_st_cap = 0;
  char *cb_stage = nullptr;  // dirty cache slots of a round, sent with ONE copy and scattered by a kernel: [indices | slots]
  size_t cb_stage_cap = 0;   // ... slots it holds
  char *cb_stage_hp = nullptr;  // ... its page-locked host end
  hipEvent_t cb_stage_ev = nullptr;  // ... and the last copy out of it
  std::mutex cb_mu;          // one decode at a time mutates the cache
  // page-locked, device-visible block of the launches that wait for their answers (cb_rounds): [log length | workgroup reports |
  // pairs answered per item | miss list | log: slot indices | log: slots]
  char *h_live = nullptr, *d_live = nullptr;  // host address / the same memory as the device addresses it
  unsigned live_miss_hw = 0;                  // miss-list entries the last launch may have written (reset to the sentinel before the next)
  std::vector<uint32_t> live_stamp;           // per cache slot: the waiting launch whose log holds it
  std::vector<unsigned long long> live_memo;  // (state, word) pairs recently put into / found in that log (direct-mapped, 512 KB)
  uint32_t live_launch = 0;
  CbAskPool ask_pool;                         // helper threads of a thread-safe callback (none unless asked for)
};
constexpr uint32_t kCbMissCap = 1u << 18;  // queued (state, word) pairs per round (2 MB); more are dropped and asked again
constexpr uint32_t kCbLogCap = 1u << 21;   // cache slots one waiting launch can be handed (40 MB of page-locked memory)
constexpr size_t kCbLiveSlots0 = (size_t)1 << 22;   // a waiting launch cannot have its tables moved: they start with room for 2 M windows (64 MB of HBM) ...
constexpr size_t kCbLiveStates0 = (size_t)1 << 22;  // ... and 4 M states (two zeroed arrays, 32 MB)
constexpr uint32_t kCbLiveItems = 1u << 16;
constexpr size_t kLiveOffDone = 256, kLiveOffAns = kLiveOffDone + (size_t)kCbLiveItems * 4, kLiveOffMiss = kLiveOffAns + (size_t)kCbLiveItems * 4,
                 kLiveOffIdx = kLiveOffMiss + (size_t)kCbMissCap * sizeof(ctclm::MissEntry), kLiveOffSlot = kLiveOffIdx + (size_t)kCbLogCap * 4,
                 kLiveBytes = kLiveOffSlot + (size_t)kCbLogCap * sizeof(ctclm::NgSlot);

namespace {

struct CompactOut {          // compact result delivery (beam_core.h OutRefs::c_*); all device pointers
  int32_t *hdr, *ent;
  uint32_t *rag;
  unsigned *count;
  unsigned cap;
  // mirrors in page-locked host memory the kernel writes a finished utterance's results to (OutRefs::m_*); null: none
  int32_t *m_hdr = nullptr, *m_ent = nullptr, *m_done = nullptr;
  uint32_t *m_rag = nullptr;
  unsigned m_cap = 0;
};

struct StreamCall {          // extra arguments of a streaming decode (lens: the chunk lengths, host memory)
  ctcd_stream **states;
  const unsigned char *is_eos;  // host
  int out_T;
  const int32_t *lens;          // host: frames of this chunk per item
  bool any_eos;
  // resumed launches of the scorer hook (decode_lm_callback): the results of items that finished in an earlier launch stay
  // (nothing is zero-filled), every item starts at its own frame of the rows, and reports how far its parked state got
  bool no_clear = false;
  const int *frame_off = nullptr;   // device, [B]
  int *frames_done = nullptr;       // device, [B]
  const int32_t *row_lens = nullptr;  // device, [B]: frame_off + lens = the rows the pre-passes (log conversion, pruning) cover
  const char *live = nullptr;         // the scorer's page-locked block as the device sees it (ctcd_scorer::d_live): the launch waits for its answers
};

std::atomic<unsigned long long> g_stream_call_id{0};
// (the parked arrays are always laid out for the LM tier's larger set: a stream block is sized once, before its scorer matters)
size_t stream_pool_offset(int beam) { return ((size_t)(SH_WORDS + kStateArraysLm * (size_t)beam) * 4 + 255) / 256 * 256; }
// a stream block holds [header | beam arrays | node pool (nodes * 12 B) | express pointers (nodes * 4 B) | high parts of the
// nodes' time steps (nodes * 4 B, zero until the stream passes frame 65535)]
size_t stream_nodes(long long frames, int beam) { return (size_t)frames * beam + 1; }
size_t stream_block_bytes(long long cap_frames, int beam) {
  return stream_pool_offset(beam) + stream_nodes(cap_frames, beam) * (sizeof(PoolNode) + 2 * sizeof(int));
}
size_t stream_thi_offset(long long cap_frames, int beam) {
  return stream_pool_offset(beam) + stream_nodes(cap_frames, beam) * (sizeof(PoolNode) + sizeof(int));
}

Dims make_dims(int beam, int V, int cutoff_top_n, double cutoff_prob, bool lm = false) {
  const bool pruned = cutoff_prob < 1.0 || cutoff_top_n < V;
  Dims d;
  d.K = beam;
  d.V = V;
  d.Vc_max = pruned ? (cutoff_top_n < V ? cutoff_top_n : V) : V;
  d.use_rank_table = pruned ? 1 : 0;
  d.lm = lm ? 1 : 0;
  return d;
}

int check_args(int B, int T, int V, int beam, int cutoff_top_n, int blank_id, const void *probs, const void *tok,
               const void *ts, const void *sc, const void *ln) {
  if (B < 0 || T < 0 || V <= 0 || beam <= 0 || cutoff_top_n <= 0) return fail(CTCD_EINVAL, "B, T >= 0 and V, beam_width, cutoff_top_n > 0 required");
  if (blank_id < 0 || blank_id >= V) return fail(CTCD_EINVAL, "blank_id must index the vocabulary");
  if (beam > kMaxBeam) return fail(CTCD_EUNSUPPORTED, "beam_width > 16383");
  if (V > kMaxVocab) return fail(CTCD_EUNSUPPORTED, "vocabulary > 65534");
  if ((long long)beam * T + 1 > 0x7fffffffLL) return fail(CTCD_EUNSUPPORTED, "beam_width * T too large");
  if (B > 0 && T > 0 && (!probs || !tok || !ts)) return fail(CTCD_EINVAL, "null tensor");
  if (B > 0 && (!sc || !ln)) return fail(CTCD_EINVAL, "null tensor");
  return CTCD_OK;
}

}  // namespace

extern "C" {

const char *ctcd_last_error(void) { return g_err.c_str(); }
const char *ctcd_version(void) { return "ctcdecode_amd 0.1 (gfx950)"; }

int ctcd_workgroup_lds_bytes(int beam, int V, int cutoff_top_n, double cutoff_prob) {
  if (beam <= 0 || V <= 0 || cutoff_top_n <= 0) return CTCD_EINVAL;
  Work w;
  const size_t n = carve<0>(w, nullptr, nullptr, make_dims(beam, V, cutoff_top_n, cutoff_prob), nullptr);
  return n > 0x7fffffffu ? 0x7fffffff : (int)n;
}

int ctcd_create(ctcd_decoder **out, int device_id) {
  if (!out) return fail(CTCD_EINVAL, "out == NULL");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device_id < 0 || device_id >= ndev) return fail(CTCD_EINVAL, "no such HIP device");
  ctcd_decoder *d = new ctcd_decoder;
  d->device = device_id;
  int v = 0;
  HIP_TRY(hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device_id));
  d->max_lds = v;
  HIP_TRY(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device_id));
  d->cu_count = v;
  if (const char *e = getenv("CTCD_LDS_FLOOR")) d->lds_floor = atoll(e);
  if (const char *e = getenv("CTCD_NO_INPUT_STREAMING")) d->no_input_streaming = atoi(e) != 0;
  if (const char *e = getenv("CTCD_GENERAL_LM_KERNEL")) d->general_lm_kernel = atoi(e) != 0;
  *out = d;
  return CTCD_OK;
}

void ctcd_destroy(ctcd_decoder *d) {
  if (!d) return;
  DeviceGuard guard_(d->device);
  if (d->fg_in) (void)hipFree(d->fg_in);
  if (d->h_cnt) (void)hipHostFree(d->h_cnt);
  if (d->h_flagged) (void)hipHostFree(d->h_flagged);
  for (int i = 0; i < 2; ++i) {
    if (d->h_stargs[i]) (void)hipHostFree(d->h_stargs[i]);
    if (d->ev_stargs[i]) (void)hipEventDestroy(d->ev_stargs[i]);
  }
  if (d->copy_stream) (void)hipStreamDestroy(d->copy_stream);
  if (d->ev_in) (void)hipEventDestroy(d->ev_in);
  if (d->ev0) { (void)hipEventDestroy(d->ev0); (void)hipEventDestroy(d->ev1); (void)hipEventDestroy(d->ev2); (void)hipEventDestroy(d->ev3); }
  d->pool.release(); d->status.release(); d->prof.release(); d->tables.release(); d->logp.release(); d->lsm.release(); d->pr_ml.release(); d->flags.release();
  d->stage_in.release(); d->stage_out.release(); d->pr_cnt.release(); d->pr_ch.release(); d->pr_lp.release(); d->far.release(); d->st_args.release(); d->prune_in.release(); d->prune_out.release(); d->st_lens.release();
  d->dbg.release(); d->tl.release();
  d->cb_blocks.release(); d->cb_ctl.release();
  if (d->h_cb) (void)hipHostFree(d->h_cb);
  d->c_hdr.release(); d->c_ent.release(); d->c_rag.release(); d->c_cnt.release(); d->c_sc.release(); d->c_ln.release();
  if (d->h_stage) (void)hipHostFree(d->h_stage);
  delete d->workers;
  delete d;
}

int ctcd_debug_set_host_path(ctcd_decoder *d, int input_streaming, long long mirror_cap_labels) {
  if (!d) return fail(CTCD_EINVAL, "decoder == NULL");
  if (input_streaming >= 0) d->no_input_streaming = input_streaming == 0;
  d->stall_input_test = input_streaming == 2;  // tests: the rows are sent but "frames arrived" never advances -- the kernel must give up
  if (mirror_cap_labels >= -1) d->mirror_cap_override = mirror_cap_labels;
  return CTCD_OK;
}

int ctcd_set_cu_sharing(ctcd_decoder *d, int mode) {
  if (!d || mode < -1 || mode > 1) return fail(CTCD_EINVAL, "cu sharing mode must be -1 (automatic), 0 or 1");
  d->cu_sharing = mode;
  return CTCD_OK;
}

int ctcd_set_subtree_search(ctcd_decoder *d, int mode) {
  if (!d || mode < -1 || mode > 1) return fail(CTCD_EINVAL, "subtree search mode must be -1 (automatic), 0 or 1");
  d->subtree_mode = mode;
  return CTCD_OK;
}
int ctcd_last_subtree_search(const ctcd_decoder *d) { return d ? d->last_subtree_search : -1; }

int ctcd_set_threads(ctcd_decoder *d, int t) {
  if (!d || t < 0 || t > 1024 || (t && (t < 64 || (t & (t - 1))))) return fail(CTCD_EINVAL, "threads must be 0 (automatic) or a power of two in [64, 1024]");
  d->threads = t;
  return CTCD_OK;
}

// The builds whose kernels ignore KernelArgs::frames_ready (decode_kernel.h kNoStreamedInput) and the twins that poll it.  EVERY
// instantiation for which kNoStreamedInput holds must be listed here.
static const void *streamed_input_twin(const void *fn) {
  struct Pair { const void *plain, *twin; };
  static const Pair tab[] = {
#if defined(CTC_QUICK_BUILD) && CTC_QUICK_BUILD == 2
    {(const void *)ctc_beam_decode_kernel<0, 0, 1, false, 1024, 2>, (const void *)ctc_beam_decode_kernel<4, 0, 1, false, 1024, 2>},
#elif defined(CTC_QUICK_BUILD) && (CTC_QUICK_BUILD == 3 || CTC_QUICK_BUILD == 4)
    {nullptr, nullptr},
#elif defined(CTC_QUICK_BUILD)
    {(const void *)ctc_beam_decode_kernel<0, 0, 1, false, 1024>, (const void *)ctc_beam_decode_kernel<4, 0, 1, false, 1024>},
    {(const void *)ctc_beam_decode_kernel<0, 0, 1, false, 1024, false, true>, (const void *)ctc_beam_decode_kernel<4, 0, 1, false, 1024, false, true>},
#else
    {(const void *)ctc_beam_decode_kernel<0, 0, 1, false, 1024>, (const void *)ctc_beam_decode_kernel<4, 0, 1, false, 1024>},
    {(const void *)ctc_beam_decode_kernel<3, 0, 1, false, 1024>, (const void *)ctc_beam_decode_kernel<5, 0, 1, false, 1024>},
    {(const void *)ctc_beam_decode_kernel<0, 0, 1, false, 1024, false, true>, (const void *)ctc_beam_decode_kernel<4, 0, 1, false, 1024, false, true>},
    {(const void *)ctc_beam_decode_kernel<0, 0, 1, false, 1024, 2>, (const void *)ctc_beam_decode_kernel<4, 0, 1, false, 1024, 2>},
#endif
  };
  for (const Pair &p : tab)
    if (p.plain && p.plain == fn) return p.twin;
  return nullptr;
}

// log_softmax of every frame (ctcd_log_softmax's definition): long rows by a workgroup each, short ones by a wave each
static int launch_log_softmax(ctcd_decoder *d, const float *in, float *out, long long rows, const int32_t *lens, int T, int V, hipStream_t stream) {
  const uint64_t *tb = (const uint64_t *)d->tables.p;
  if (V % 4 == 0 && V > 256 && V <= 16384 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0 && !d->no_fused_logits) {
    const dim3 grid((unsigned)std::min<long long>(rows, 256 * 32));
    if (V <= 1024) hipLaunchKernelGGL(log_softmax_rows_wg_kernel<1>, grid, dim3(256), 0, stream, in, out, rows, lens, T, V, tb);
    else if (V <= 2048) hipLaunchKernelGGL(log_softmax_rows_wg_kernel<2>, grid, dim3(256), 0, stream, in, out, rows, lens, T, V, tb);
    else if (V <= 4096) hipLaunchKernelGGL(log_softmax_rows_wg_kernel<4>, grid, dim3(256), 0, stream, in, out, rows, lens, T, V, tb);
    else if (V <= 10240) hipLaunchKernelGGL(log_softmax_rows_wg_kernel<10>, grid, dim3(256), 0, stream, in, out, rows, lens, T, V, tb);
    else hipLaunchKernelGGL(log_softmax_rows_wg_kernel<16>, grid, dim3(256), 0, stream, in, out, rows, lens, T, V, tb);
  } else {
    hipLaunchKernelGGL(log_softmax_rows_kernel, dim3((unsigned)std::min<long long>((rows + 3) / 4, 256 * 64)), dim3(256), 0, stream, in, out, rows, lens, T, V, tb);
  }
  HIP_TRY(hipGetLastError());
  return CTCD_OK;
}

static int decode_common(ctcd_decoder *d, const float *probs, const int32_t *seq_lens, int B, int T, int V, int beam,
                         double cutoff_prob, int cutoff_top_n, int blank_id, int log_input, int32_t *out_tok, int32_t *out_ts,
                         float *out_sc, int32_t *out_len, int32_t *n_results, void *stream_, const StreamCall *sc,
                         ctcd_scorer *scorer = nullptr, const CompactOut *co = nullptr, const int *frames_ready = nullptr) {
  if (!d) return fail(CTCD_EINVAL, "decoder == NULL");
  if (scorer && scorer->device != d->device) return fail(CTCD_EINVAL, "the scorer's tables live on another device than the decoder");
  if (scorer && (int)scorer->host.labels.size() != V) return fail(CTCD_EINVAL, "the scorer was built for a different number of labels");
  const int out_T = sc ? sc->out_T : T;
  // (a streaming call in which no stream ends has out_T == 0 and may pass null token / timestep buffers)
  const bool no_rows = (sc && out_T == 0) || co;
  if (co && (out_T > 65536 || V > 65535)) return fail(CTCD_EUNSUPPORTED, "compact results pack label and frame into 16 bits each");
  int rc = check_args(B, T, V, beam, cutoff_top_n, blank_id, probs, no_rows ? (const void *)d : (const void *)out_tok,
                      no_rows ? (const void *)d : (const void *)out_ts, out_sc, out_len);
  if (rc) return rc;
  std::lock_guard<std::mutex> lock(d->mu);
  hipStream_t stream = (hipStream_t)stream_;
  CTC_ON_DEVICE(d->device);
  if (B == 0) return CTCD_OK;
  const Dims dims = make_dims(beam, V, cutoff_top_n, cutoff_prob, scorer != nullptr);
  // more than 65535 candidate slots (cutoff_top_n >= V with thousands of labels): the layout with 32-bit slot indices and
  // everything per slot in HBM scratch (workspace level 3) -- the reference has no such limit (decoder_utils.cpp:33-35)
  const bool huge = dims.S_max() > 65535;
  if ((long long)beam * (dims.Vc_max + 2) > (1LL << 24) - 1)
    return fail(CTCD_EUNSUPPORTED, "beam_width * (candidates + 2) exceeds 16777215 candidate slots");
  if (huge && d->profile) return fail(CTCD_EUNSUPPORTED, "the instrumented kernel builds do not include the layout for more than 65535 candidate slots");
  if (dims.use_rank_table && V > 32767) return fail(CTCD_EUNSUPPORTED, "vocabulary pruning with more than 32767 labels");
  Work wtmp;
  size_t far_bytes = 0;
  // workgroup size (measured): 1024 threads for the usual shapes; below ~1300 candidate slots 512 is marginally better
  // (fewer idle waves), fewer than that is always slower (the new-children phase wants its own waves)
  int threads = d->threads;
  if (threads == 0) threads = (dims.S_max() <= 1300 && !scorer) ? 512 : 1024;
  // (the LM tier has the fixed-layout kernel at 1024 threads only)
  const bool fixed = fits_fixed_layout(dims) && !d->no_fixed_layout && (!scorer || threads == 1024);
  // the second compile-time layout: the pruned default (beam <= 112, cutoff_top_n <= 40) on a vocabulary of up to 10 240 labels
  const bool fixed2 = !fixed && fits_mid_layout(dims) && !d->no_fixed_layout && !scorer && threads == 1024 && !d->profile;
  const Dims ldims = fixed ? fixed_layout_dims(scorer != nullptr) : fixed2 ? mid_layout_dims() : dims;
  // two workgroups per CU (OCC2 build of the fixed-layout kernel): batches that outnumber the CUs, or on request
  const bool hooked = scorer && scorer->cbl;  // a callback scorer: its own kernel instantiations (beam_core.h CB)
  const bool occ2 = fixed && threads == 1024 && !d->profile && !hooked && (d->cu_sharing == 1 || (d->cu_sharing < 0 && B > d->cu_count));
  size_t lds = occ2 ? carve<0, true>(wtmp, nullptr, nullptr, ldims, &far_bytes) : carve<0>(wtmp, nullptr, nullptr, ldims, &far_bytes);
  bool big = false;
  int far_level = 1;
  if (huge) {
    big = true;
    far_level = 3;
    lds = carve<3>(wtmp, nullptr, nullptr, dims, &far_bytes);
  } else if (lds + 2048 > (size_t)d->max_lds) {  // wide beam: rare-path arrays go to HBM scratch
    big = true;
    lds = carve<1>(wtmp, nullptr, nullptr, dims, &far_bytes);
    if (lds + 2048 > (size_t)d->max_lds) {  // wider still: the slot keys and the rarely read per-entry arrays follow them
      far_level = 2;
      lds = carve<2>(wtmp, nullptr, nullptr, dims, &far_bytes);
    }
  }
  // the first wide-beam layout at its compile-time size (decode_kernel.h LAYOUT 3: beam <= 500 over <= 29 labels, no pruning, no scorer)
  bool wide3 = false;
#if !defined(CTC_QUICK_BUILD)
  if (big && far_level == 1 && fits_wide_layout(dims) && !d->no_fixed_layout && !scorer && threads == 1024 && !d->profile) {
    size_t fb3 = 0;
    const size_t lds3 = carve<1>(wtmp, nullptr, nullptr, wide_layout_dims(), &fb3);
    if (lds3 + 2048 <= (size_t)d->max_lds) { wide3 = true; lds = lds3; far_bytes = fb3; }
  }
#endif
  far_bytes = (far_bytes + 255) / 256 * 256;
  if (lds + 2048 > (size_t)d->max_lds)
    return fail(CTCD_EUNSUPPORTED, "beam_width * (candidates + 2) needs " + std::to_string(lds) + " B of LDS, more than one workgroup has");
  if (scorer && d->profile && !d->tl_armed) return fail(CTCD_EUNSUPPORTED, "the phase-timer kernel builds do not include the LM tier (the barrier timeline does)");
  if ((rc = d->far.ensure((size_t)B * far_bytes))) return rc;

  // outputs: everything outside the valid region is defined as 0
  const size_t kt = co ? 0 : (size_t)B * beam * out_T;
  if (co && !(sc && sc->no_clear)) {  // (resumed launches of the scorer hook: the records of items that finished earlier stay)
    HIP_TRY(hipMemsetAsync(co->count, 0, 4, stream));
    HIP_TRY(hipMemsetAsync(co->hdr, 0, (size_t)B * 16, stream));
  }
  if (kt && !(sc && sc->no_clear)) {
    HIP_TRY(hipMemsetAsync(out_tok, 0, kt * 4, stream));
    HIP_TRY(hipMemsetAsync(out_ts, 0, kt * 4, stream));
  }
  if ((!sc || sc->any_eos) && !(sc && sc->no_clear)) {  // (a streaming call in which no stream ends writes no results: nothing to define)
    HIP_TRY(hipMemsetAsync(out_sc, 0, (size_t)B * beam * 4, stream));
    HIP_TRY(hipMemsetAsync(out_len, 0, (size_t)B * beam * 4, stream));
    if (sc && n_results) HIP_TRY(hipMemsetAsync(n_results, 0, (size_t)B * 4, stream));
  }

  if (!d->tables_ready) {
    if ((rc = d->tables.ensure(sizeof(ctcmath::Tables)))) return rc;
    HIP_TRY(hipMemcpy(d->tables.p, ctcmath::host_tables().w, sizeof(ctcmath::Tables), hipMemcpyHostToDevice));
    d->tables_ready = true;
  }
  // streaming: per-item block pointers, pool capacities and end-of-stream flags go to the device
  char **st_base = nullptr;
  int *st_cap = nullptr;
  unsigned char *st_eos = nullptr;
  if (sc) {
    // [block pointers | pool capacities | chunk lengths | end-of-stream flags], staged in page-locked memory (two slots,
    // an event each) and sent with ONE asynchronous copy: a chunk call synchronises nothing (round 3: two stream
    // synchronisations per call, one for this block and one for the lengths)
    const size_t off_cap = (size_t)B * 8, off_len = off_cap + (size_t)B * 4, off_eos = off_len + (size_t)B * 4, need = (off_eos + (size_t)B + 31) & ~(size_t)15;
    if ((rc = d->st_args.ensure(2 * need))) return rc;
    if (d->h_stargs_cap < need) {
      for (int i = 0; i < 2; ++i) {
        if (d->ev_stargs[i]) HIP_TRY(hipEventSynchronize(d->ev_stargs[i]));
        if (d->h_stargs[i]) (void)hipHostFree(d->h_stargs[i]);
        d->h_stargs[i] = nullptr;
        HIP_TRY(hipHostMalloc((void **)&d->h_stargs[i], need, hipHostMallocDefault));
        if (!d->ev_stargs[i]) HIP_TRY(hipEventCreateWithFlags(&d->ev_stargs[i], hipEventDisableTiming));
      }
      d->h_stargs_cap = need;
    }
    const int slot = (int)(d->stream_calls++ & 1);
    HIP_TRY(hipEventSynchronize(d->ev_stargs[slot]));  // (the copy of two calls ago: long done)
    char *hb = d->h_stargs[slot];
    for (int b = 0; b < B; ++b) {
      ctcd_stream *st = sc->states[b];
      ((char **)hb)[b] = st->block;
      ((int *)(hb + off_cap))[b] = (int)(st->cap_frames * beam + 1);
      ((int *)(hb + off_len))[b] = sc->lens[b];
      hb[off_eos + b] = sc->is_eos[b] ? 1 : 0;
    }
    char *db = (char *)d->st_args.p + (size_t)slot * need;
    HIP_TRY(hipMemcpyAsync(db, hb, off_eos + (size_t)B, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(d->ev_stargs[slot], stream));
    st_base = (char **)db;
    st_cap = (int *)(db + off_cap);
    st_eos = (unsigned char *)db + off_eos;
    seq_lens = (const int32_t *)(db + off_len);
  }
  // the rows the pre-passes cover (resumed launches of the scorer hook start inside the rows: StreamCall::row_lens)
  const int32_t *pre_lens = (sc && sc->row_lens) ? sc->row_lens : seq_lens;

  // raw logits in front of a vocabulary prune: one pass does both (prune_logits_wg_kernel -- the normalised rows are never written).
  // Not with a scorer: the LM tier reads the blank's log-probability from the rows themselves (ctc_beam_search_decoder.cpp:78).
  const bool fuse_logits = log_input == 2 && dims.use_rank_table && T > 0 && !scorer && !d->no_fused_logits && V % 4 == 0 && V > 256 &&
                           V <= 16384 && std::min(cutoff_top_n, V) <= 64 && ((uintptr_t)probs & 15) == 0;
  if (log_input == 2 && !fuse_logits) {  // raw logits: normalise once, in HBM; everything below sees log-probabilities
    if (T > 0) {
      if ((rc = d->lsm.ensure((size_t)B * T * V * 4))) return rc;
      if (pre_lens) HIP_TRY(hipMemsetAsync(d->lsm.p, 0, (size_t)B * T * V * 4, stream));  // frames past an utterance's end stay defined
      if ((rc = launch_log_softmax(d, probs, (float *)d->lsm.p, (long long)B * T, pre_lens, T, V, stream))) return rc;
      probs = (const float *)d->lsm.p;
    }
    log_input = 1;
  }
  const long long pool_stride = (long long)beam * T + 1;
  // per utterance: nodes (12 B each), then -- per utterance again -- express pointers and the time steps' high parts (4 B each)
  if (!sc && (rc = d->pool.ensure((size_t)B * pool_stride * (sizeof(PoolNode) + 2 * sizeof(int))))) return rc;
  if ((rc = d->status.ensure((size_t)B * 8))) return rc;  // [status words | beam-shape statistic]
  // every status word starts as -1 ("no result"): a workgroup that never ran cannot read back as ST_OK
  HIP_TRY(hipMemsetAsync(d->status.p, 0xff, (size_t)B * 4, stream));
  d->last_stream = stream;
  const float *logp = probs;
  d->prune_flagged_rows = 0;
  d->flagged_pending = false;
  if (dims.use_rank_table && T > 0) {
    // vocabulary prune pass (also converts the kept probabilities to log space when log_input == 0)
    const long long rows = (long long)B * T;
    const int stride = dims.Vc_max;

    if ((rc = d->pr_cnt.ensure((size_t)rows * 4))) return rc;
    if ((rc = d->pr_ch.ensure((size_t)rows * stride * 4))) return rc;
    if ((rc = d->pr_lp.ensure((size_t)rows * stride * 4))) return rc;
    const int wpb = 4;
    const size_t psm = (size_t)wpb * (3 * (size_t)stride + 2 * kPruneCand) * 4;
    if (psm > (size_t)d->max_lds) return fail(CTCD_EUNSUPPORTED, "cutoff_top_n too large for the prune pass");
    const int blocks = (int)std::min<long long>((rows + wpb - 1) / wpb, 256 * 16);
    const void *pfn = nullptr;
    pfn = V <= 64 ? (const void *)prune_rows_kernel<1> : V <= 256 ? (const void *)prune_rows_kernel<4>
                    : V <= 1024 ? (const void *)prune_rows_kernel<16> : V <= 4096 ? (const void *)prune_rows_kernel<64>
                    : V <= 10240 ? (const void *)prune_rows_kernel<160> : (const void *)prune_rows_kernel<0>;
    bool wg_kernel = false;
    if (V % 4 == 0 && V > 256 && V <= 16384 && std::min(cutoff_top_n, V) <= 64) {  // the bandwidth-shaped variant
      wg_kernel = true;
      pfn = V <= 1024 ? (const void *)prune_rows_wg_kernel<1> : V <= 2048 ? (const void *)prune_rows_wg_kernel<2>
          : V <= 4096 ? (const void *)prune_rows_wg_kernel<4> : V <= 10240 ? (const void *)prune_rows_wg_kernel<10>
          : (const void *)prune_rows_wg_kernel<16>;
      static const bool reg_rows = getenv("CTCD_PRUNE_REG") ? atoi(getenv("CTCD_PRUNE_REG")) != 0 : true;
      if (reg_rows && !d->no_prune_reg)  // the row held in registers between the two looks at it (V <= 10240: beyond, the registers run out)
        pfn = V <= 1024 ? (const void *)prune_rows_wg_kernel<1, true> : V <= 2048 ? (const void *)prune_rows_wg_kernel<2, true>
            : V <= 4096 ? (const void *)prune_rows_wg_kernel<4, true> : V <= 10240 ? (const void *)prune_rows_wg_kernel<10, true> : pfn;
    }
    if (fuse_logits) {  // (same shape conditions)
      pfn = V <= 1024 ? (const void *)prune_logits_wg_kernel<1> : V <= 2048 ? (const void *)prune_logits_wg_kernel<2>
          : V <= 4096 ? (const void *)prune_logits_wg_kernel<4> : V <= 10240 ? (const void *)prune_logits_wg_kernel<10>
          : (const void *)prune_logits_wg_kernel<16>;
      if ((rc = d->pr_ml.ensure((size_t)rows * 8))) return rc;
    }
    const size_t psm_launch = wg_kernel ? (3 * (size_t)stride + 3 * kPruneCand + 4) * 4 : psm;
    // (measured: a persistent grid of 1280 or 2560 workgroups is slower, 16 k / 32 k the same)
    const int blocks_launch = wg_kernel ? (int)std::min<long long>(rows, 256 * 32) : blocks;
    HIP_TRY(hipFuncSetAttribute(pfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)psm_launch));
    // Frames the fast pass cannot settle (ties at the cut, a cumulative sum next to cutoff_prob) are flagged and settled by
    // prune_resolve_kernel right behind it -- on the device, on the same stream: no read-back, no host arithmetic, no
    // synchronisation before the decode kernel.  The flag list holds every frame of the call (rows words).
    const unsigned cap = (unsigned)std::min<long long>(rows, 0x7fffffffLL);
    if ((rc = d->flags.ensure(8 + (size_t)cap * 4))) return rc;
    unsigned *n_flag = (unsigned *)d->flags.p, *flag_rows = (unsigned *)((char *)d->flags.p + 8);
    d->prune_timed = false;
    const size_t rbytes = prune_resolve_lds_bytes(V, std::min(cutoff_top_n, V));
    const bool in_lds = rbytes + 1024 <= (size_t)d->max_lds && !d->resolve_in_global;
    const int rblocks = in_lds ? 256 : 64;
    if (in_lds) HIP_TRY(hipFuncSetAttribute((const void *)prune_resolve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rbytes));
    const size_t rstride = (rbytes + 255) / 256 * 256;
    if (!in_lds && (rc = d->prune_in.ensure(rstride * rblocks))) return rc;  // (rows too long for LDS: the sort's arrays in global memory)
    HIP_TRY(hipMemsetAsync(n_flag, 0, 8, stream));
    HIP_TRY(hipMemsetAsync(d->pr_cnt.p, 0, (size_t)rows * 4, stream));
    PruneArgs pa;
    pa.in = probs; pa.seq_lens = pre_lens; pa.T = T; pa.V = V; pa.top_n = cutoff_top_n; pa.log_input = log_input;
    pa.stride = stride; pa.rows = rows; pa.cutoff_prob = cutoff_prob; pa.cnt = (int *)d->pr_cnt.p; pa.ch = (int *)d->pr_ch.p;
    pa.lp = (float *)d->pr_lp.p; pa.n_flag = n_flag; pa.flag_rows = flag_rows; pa.flag_cap = cap;
    pa.cut_exp = std::exp(cutoff_prob); pa.tables = (const uint64_t *)d->tables.p;
    pa.row_max = fuse_logits ? (float *)d->pr_ml.p : nullptr; pa.row_lse = fuse_logits ? (float *)d->pr_ml.p + rows : nullptr;
    const uint64_t *ptables = (const uint64_t *)d->tables.p;
    void *pargs[] = {&pa, (void *)&ptables};  // (the second argument: prune_logits_wg_kernel only)
    if (d->timing) HIP_TRY(hipEventRecord(d->ev2, stream));
    HIP_TRY(hipLaunchKernel(pfn, dim3(blocks_launch), dim3(wpb * 64), pargs, psm_launch, stream));
    HIP_TRY(hipGetLastError());
    if (d->timing) { HIP_TRY(hipEventRecord(d->ev3, stream)); d->prune_timed = true; }
    // (prune_resolve_kernel keeps positions and counters in 16 bits: the refusal of pruning with more than 32767 labels above is
    //  what guarantees they fit -- said here, next to the launch, so that the two cannot drift apart: ADVICE r4)
    static_assert(kResolveMaxV >= 32767, "prune_resolve_kernel's 16-bit positions must cover every vocabulary the prune pass accepts");
    if (V > kResolveMaxV) return fail(CTCD_EUNSUPPORTED, "vocabulary pruning: too many labels for the std::sort replay's 16-bit positions");
    hipLaunchKernelGGL(prune_resolve_kernel, dim3(rblocks), dim3(kResolveThreads), in_lds ? rbytes : 0, stream, pa, in_lds ? (char *)nullptr : (char *)d->prune_in.p, rstride);
    HIP_TRY(hipGetLastError());
    // (statistics only: the number of flagged frames travels to page-locked memory behind the kernels; whoever asks for it
    //  -- ctcd_last_prune_flagged_rows -- waits for the stream)
    if (!d->h_flagged) HIP_TRY(hipHostMalloc((void **)&d->h_flagged, 8, hipHostMallocDefault));
    HIP_TRY(hipMemcpyAsync(d->h_flagged, n_flag, 4, hipMemcpyDeviceToHost, stream));
    d->flagged_pending = true;
  } else if (!log_input && T > 0) {
    const size_t n = (size_t)B * T * V;
    if ((rc = d->logp.ensure(n * 4))) return rc;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
    if (pre_lens) HIP_TRY(hipMemsetAsync(d->logp.p, 0, n * 4, stream));  // frames past an utterance's end stay defined
    hipLaunchKernelGGL(prob_to_log_kernel, dim3(blocks), dim3(256), 0, stream, probs, (float *)d->logp.p, n, pre_lens, T, V);
    HIP_TRY(hipGetLastError());
    logp = (const float *)d->logp.p;
  }

  KernelArgs a;
  a.probs = logp; a.seq_lens = seq_lens; a.B = B; a.T = T; a.V = V; a.K = beam; a.blank = blank_id; a.dims = dims;
  a.pool = (PoolNode *)d->pool.p; a.pool_stride = pool_stride;
  a.pool_up = (int *)(a.pool + (size_t)B * pool_stride);
  a.tables = (const uint64_t *)d->tables.p;
  a.outs.tok = out_tok; a.outs.ts = out_ts; a.outs.len = out_len; a.outs.n_results = n_results; a.outs.score = out_sc;
  a.outs.K = beam; a.outs.T_stride = out_T;
  a.outs.c_hdr = nullptr; a.outs.c_ent = nullptr; a.outs.c_rag = nullptr; a.outs.c_count = nullptr; a.outs.c_cap = 0;
  a.outs.m_hdr = nullptr; a.outs.m_ent = nullptr; a.outs.m_done = nullptr; a.outs.m_rag = nullptr; a.outs.m_cap = 0;
  if (co) {
    a.outs.tok = nullptr; a.outs.ts = nullptr;
    a.outs.c_hdr = co->hdr; a.outs.c_ent = co->ent; a.outs.c_rag = co->rag; a.outs.c_count = co->count; a.outs.c_cap = co->cap;
    a.outs.m_hdr = co->m_hdr; a.outs.m_ent = co->m_ent; a.outs.m_done = co->m_done; a.outs.m_rag = co->m_rag; a.outs.m_cap = co->m_cap;
  }
  a.status = (int32_t *)d->status.p;
  a.shape = a.status + B;
  d->status_items = B;
  a.st_base = st_base; a.st_poolcap = st_cap; a.st_eos = st_eos; a.st_pool_off = (long long)stream_pool_offset(beam);
  a.raw = probs; a.raw_log = log_input;
  std::memset(&a.lm, 0, sizeof(a.lm));
  if (scorer) {
    a.lm = scorer->dview;
    a.lm.alpha = scorer->host.alpha;  // reset_params (binding.cpp:283-287) takes effect at the next decode
    a.lm.beta = scorer->host.beta;
  }
  a.frames_ready = frames_ready;  // (streamed input of the host-tensor entry point; null: every row is in place)
  a.cb_log_len = nullptr; a.cb_log_idx = nullptr; a.cb_log_slot = nullptr; a.cb_done = nullptr; a.cb_ans = nullptr;
  a.lm.cb_ring = 0;
  if (sc && sc->live) {
    a.lm.cb_ring = 1;
    a.cb_log_len = (const unsigned *)sc->live; a.cb_done = (int32_t *)(sc->live + kLiveOffDone); a.cb_ans = (const unsigned *)(sc->live + kLiveOffAns);
    a.cb_log_idx = (const uint32_t *)(sc->live + kLiveOffIdx); a.cb_log_slot = (const ctclm::NgSlot *)(sc->live + kLiveOffSlot);
  }
  a.frame_off = sc ? sc->frame_off : nullptr;
  a.frames_done = sc ? sc->frames_done : nullptr;
  a.pr_cnt = nullptr; a.pr_ch = nullptr; a.pr_lp = nullptr; a.pr_stride = 0;
  if (dims.use_rank_table) {
    a.pr_cnt = (const int *)d->pr_cnt.p; a.pr_ch = (const int *)d->pr_ch.p; a.pr_lp = (const float *)d->pr_lp.p;
    a.pr_stride = dims.Vc_max;
  }
  a.prof = nullptr;
  a.dbg = nullptr;
  a.tl = nullptr; a.tl_cap = kTimelineCap; a.tl_f0 = d->tl_f0; a.tl_nf = d->tl_nf;
  if (d->profile && d->tl_armed) {
    if ((rc = d->tl.ensure((size_t)16 * kTimelineCap * 8))) return rc;
    HIP_TRY(hipMemsetAsync(d->tl.p, 0, (size_t)16 * kTimelineCap * 8, stream));
    a.tl = (long long *)d->tl.p;
  }
  if (d->profile && d->dbg_on) {
    if ((rc = d->dbg.ensure((size_t)(T + 1) * (1 + 4 * (size_t)beam) * 4))) return rc;
    a.dbg = (int *)d->dbg.p;
  }
  if (d->profile) {
    if ((rc = d->prof.ensure((size_t)B * 16 * 8))) return rc;
    a.prof = (long long *)d->prof.p;
  }
  a.far = (char *)d->far.p; a.far_stride = (long long)far_bytes;
  const bool pruned_mode = a.pr_cnt != nullptr;
  const void *fn;
#if defined(CTC_QUICK_BUILD) && CTC_QUICK_BUILD == 4
  if (big || scorer || d->profile || occ2 || !(pruned_mode || (!fixed && !fixed2)))
    return fail(CTCD_EUNSUPPORTED, "CTC_QUICK_BUILD=4: only the pruned-mode kernels of the compile-time layouts and the run-time layout were compiled");
  fn = fixed2 ? (const void *)ctc_beam_decode_kernel<0, 0, 2, true, 1024> : fixed && threads == 1024 ? (const void *)ctc_beam_decode_kernel<0, 0, 1, true, 1024>
       : pruned_mode ? (const void *)ctc_beam_decode_kernel<0, 0, 0, true> : (const void *)ctc_beam_decode_kernel<0, 0, 0, false>;
  if (fixed && threads != 1024) return fail(CTCD_EUNSUPPORTED, "CTC_QUICK_BUILD=4: 1024 threads");
#elif defined(CTC_QUICK_BUILD) && CTC_QUICK_BUILD == 2
  if (big || !fixed || pruned_mode || !scorer || occ2 || threads != 1024 || (d->profile && !d->tl_armed))
    return fail(CTCD_EUNSUPPORTED, "CTC_QUICK_BUILD=2: only the fixed-layout, no-prune, 1024-thread kernel of the LM tier was compiled");
  fn = d->profile ? (const void *)ctc_beam_decode_kernel<2, 0, 1, false, 1024, true> : (const void *)ctc_beam_decode_kernel<0, 0, 1, false, 1024, true>;
  if (!scorer->host.char_based && !scorer->host.dict_wide && !d->general_lm_kernel)
    fn = d->profile ? (const void *)ctc_beam_decode_kernel<2, 0, 1, false, 1024, 2> : (const void *)ctc_beam_decode_kernel<0, 0, 1, false, 1024, 2>;
#elif defined(CTC_QUICK_BUILD) && CTC_QUICK_BUILD == 3
  // Workgroup-size sweep (tools/build_variants.sh nt512:CTC_QUICK_BUILD=3,CTC_QUICK_NT=512; raw_multi.py --threads 512)
  if (big || !fixed || pruned_mode || scorer || occ2 || threads != CTC_QUICK_NT || d->profile)
    return fail(CTCD_EUNSUPPORTED, "CTC_QUICK_BUILD=3: only the fixed-layout, no-prune, no-LM kernel with CTC_QUICK_NT threads was compiled (ctcd_set_threads)");
  fn = (const void *)ctc_beam_decode_kernel<0, 0, 1, false, CTC_QUICK_NT>;
#elif defined(CTC_QUICK_BUILD)
  // Experiment builds (tools/build_variants.sh, seconds instead of minutes): only the north-star class kernel and its
  // barrier-timeline twin exist; everything else is refused.
  if (big || !fixed || pruned_mode || scorer || threads != 1024 || (d->profile && !d->tl_armed))
    return fail(CTCD_EUNSUPPORTED, "CTC_QUICK_BUILD: only the fixed-layout, no-prune, no-LM, 1024-thread kernel was compiled");
  fn = d->profile ? (const void *)ctc_beam_decode_kernel<2, 0, 1, false, 1024> : (const void *)ctc_beam_decode_kernel<0, 0, 1, false, 1024>;
  if (occ2) fn = (const void *)ctc_beam_decode_kernel<0, 0, 1, false, 1024, false, true>;
#else
#define CTC_PICK(PROF_)                                                                                                  \
  (big ? (pruned_mode ? (const void *)ctc_beam_decode_kernel<PROF_, 1, 0, true> : (const void *)ctc_beam_decode_kernel<PROF_, 1, 0, false>)    \
       : fixed ? (pruned_mode ? (const void *)ctc_beam_decode_kernel<PROF_, 0, 1, true> : (const void *)ctc_beam_decode_kernel<PROF_, 0, 1, false>) \
               : (pruned_mode ? (const void *)ctc_beam_decode_kernel<PROF_, 0, 0, true> : (const void *)ctc_beam_decode_kernel<PROF_, 0, 0, false>))
  fn = d->profile ? CTC_PICK(1) : CTC_PICK(0);
  if (big && far_level == 3) fn = pruned_mode ? (const void *)ctc_beam_decode_kernel<0, 3, 0, true> : (const void *)ctc_beam_decode_kernel<0, 3, 0, false>;
  if (big && far_level == 2) {  // the widest beams: their own instantiation (every workspace array keeps a static address space)
    if (d->profile) return fail(CTCD_EUNSUPPORTED, "the instrumented kernel builds do not include the widest-beam layout");
    fn = pruned_mode ? (const void *)ctc_beam_decode_kernel<0, 2, 0, true> : (const void *)ctc_beam_decode_kernel<0, 2, 0, false>;
  }
  if (!d->profile && big && far_level == 1 && threads == 1024)  // wide beams at the usual workgroup size: folded into the code as well
    fn = pruned_mode ? (const void *)ctc_beam_decode_kernel<0, 1, 0, true, 1024> : (const void *)ctc_beam_decode_kernel<0, 1, 0, false, 1024>;
  if (!d->profile && fixed && !big && threads == 1024)  // the usual case: workgroup size folded into the code
    fn = pruned_mode ? (const void *)ctc_beam_decode_kernel<0, 0, 1, true, 1024> : (const void *)ctc_beam_decode_kernel<0, 0, 1, false, 1024>;
  d->last_subtree_search = 0;
  if (!d->profile && fixed && !big && threads == 1024 && !scorer && !occ2 && (d->subtree_mode == 1 || (d->subtree_mode < 0 && d->subtree_on))) {
    // chain-shaped beams (blank-dominated rows: what acoustic models emit): the build whose phase A1 searches four subtrees per wave
    fn = pruned_mode ? (const void *)ctc_beam_decode_kernel<3, 0, 1, true, 1024> : (const void *)ctc_beam_decode_kernel<3, 0, 1, false, 1024>;
    d->last_subtree_search = 1;
  }
  if (occ2) fn = pruned_mode ? (const void *)ctc_beam_decode_kernel<0, 0, 1, true, 1024, false, true> : (const void *)ctc_beam_decode_kernel<0, 0, 1, false, 1024, false, true>;
  if (fixed2 && !big) fn = (const void *)ctc_beam_decode_kernel<0, 0, 2, true, 1024>;
  if (wide3) fn = (const void *)ctc_beam_decode_kernel<0, 1, 3, false, 1024>;
  if (d->profile && d->tl_armed) {  // the timeline build exists for the north-star class of shapes and for the first wide-beam layout
    if (threads != 1024) return fail(CTCD_EUNSUPPORTED, "barrier timeline: 1024 threads per workgroup (the product configuration)");
    if (big && far_level == 1 && !pruned_mode) {
      fn = (const void *)ctc_beam_decode_kernel<2, 1, 0, false, 1024>;  // (a quarter of the stamps: its LDS is nearly full)
    } else {
      if (big || !fixed || pruned_mode) return fail(CTCD_EUNSUPPORTED, "barrier timeline: beam <= 128 and <= 32 labels, or the first wide-beam layout; no pruning");
      fn = (const void *)ctc_beam_decode_kernel<2, 0, 1, false, 1024>;
    }
  }
#undef CTC_PICK
  if (scorer) {
    fn = pruned_mode ? (const void *)ctc_beam_decode_kernel<0, 0, 0, true, 0, true> : (const void *)ctc_beam_decode_kernel<0, 0, 0, false, 0, true>;
    if (fixed)  // the usual class of shapes: compile-time workspace layout and workgroup size, as without a scorer
      fn = pruned_mode ? (const void *)ctc_beam_decode_kernel<0, 0, 1, true, 1024, true> : (const void *)ctc_beam_decode_kernel<0, 0, 1, false, 1024, true>;
    if (occ2) fn = pruned_mode ? (const void *)ctc_beam_decode_kernel<0, 0, 1, true, 1024, true, true> : (const void *)ctc_beam_decode_kernel<0, 0, 1, false, 1024, true, true>;
    // word models over at most 64 labels (the usual scorer): the instantiations without the character-model and
    // wide-dictionary branches (-3.6 % per frame; CTCD_GENERAL_LM_KERNEL=1 keeps the general ones: tests run both)
    if (fixed && !big && !scorer->host.char_based && !scorer->host.dict_wide && !d->general_lm_kernel) {
      fn = pruned_mode ? (const void *)ctc_beam_decode_kernel<0, 0, 1, true, 1024, 2> : (const void *)ctc_beam_decode_kernel<0, 0, 1, false, 1024, 2>;
      if (occ2) fn = pruned_mode ? (const void *)ctc_beam_decode_kernel<0, 0, 1, true, 1024, 2, true> : (const void *)ctc_beam_decode_kernel<0, 0, 1, false, 1024, 2, true>;
    }
    // wide beams: the scorer's per-entry state moves to the HBM scratch with the other rare-path arrays (a capability, not a fast path)
    if (big && far_level == 3) fn = pruned_mode ? (const void *)ctc_beam_decode_kernel<0, 3, 0, true, 0, true> : (const void *)ctc_beam_decode_kernel<0, 3, 0, false, 0, true>;
    else if (big) fn = far_level == 2 ? (pruned_mode ? (const void *)ctc_beam_decode_kernel<0, 2, 0, true, 0, true> : (const void *)ctc_beam_decode_kernel<0, 2, 0, false, 0, true>)
                                 : (pruned_mode ? (const void *)ctc_beam_decode_kernel<0, 1, 0, true, 0, true> : (const void *)ctc_beam_decode_kernel<0, 1, 0, false, 0, true>);
    if (hooked) {
      if (d->profile) return fail(CTCD_EUNSUPPORTED, "the instrumented kernel builds do not include the scorer hook");
      fn = fixed ? (pruned_mode ? (const void *)ctc_beam_decode_kernel<0, 0, 1, true, 1024, 3> : (const void *)ctc_beam_decode_kernel<0, 0, 1, false, 1024, 3>)
                 : (pruned_mode ? (const void *)ctc_beam_decode_kernel<0, 0, 0, true, 0, 3> : (const void *)ctc_beam_decode_kernel<0, 0, 0, false, 0, 3>);
      // wide beams (round 6: the reference's scorer pointer works for any beam, binding.cpp:122-140): the hook's builds of the three wide-beam layouts
      if (big && far_level == 3) fn = pruned_mode ? (const void *)ctc_beam_decode_kernel<0, 3, 0, true, 0, 3> : (const void *)ctc_beam_decode_kernel<0, 3, 0, false, 0, 3>;
      else if (big) fn = far_level == 2 ? (pruned_mode ? (const void *)ctc_beam_decode_kernel<0, 2, 0, true, 0, 3> : (const void *)ctc_beam_decode_kernel<0, 2, 0, false, 0, 3>)
                                   : (pruned_mode ? (const void *)ctc_beam_decode_kernel<0, 1, 0, true, 0, 3> : (const void *)ctc_beam_decode_kernel<0, 1, 0, false, 0, 3>);
    }
    if (d->profile && d->tl_armed) {  // (shape conditions checked above)
      if (!fixed) return fail(CTCD_EUNSUPPORTED, "barrier timeline: beam <= 128, <= 32 labels");
      fn = (const void *)ctc_beam_decode_kernel<2, 0, 1, false, 1024, true>;
      if (!scorer->host.char_based && !scorer->host.dict_wide && !d->general_lm_kernel) fn = (const void *)ctc_beam_decode_kernel<2, 0, 1, false, 1024, 2>;
    }
  }
#endif
  // streamed input (a.frames_ready: the host-tensor entry point feeds the rows while the kernel runs): the north-star class's default
  // builds do not poll for rows (decode_kernel.h kNoStreamedInput) -- their twins do.  The mapping lives HERE, behind every branch of the
  // selection above (it used to be repeated per branch; a branch without it would decode rows that have not crossed PCIe).
  if (a.frames_ready) {
    const void *twin = streamed_input_twin(fn);
    if (twin) fn = twin;
#if defined(CTC_QUICK_BUILD) && (CTC_QUICK_BUILD == 3 || CTC_QUICK_BUILD == 4)
    else return fail(CTCD_EUNSUPPORTED, "this experiment build has no kernels that take streamed input");
#endif
  }
  // (CTCD_LDS_FLOOR: experiments with the occupancy the LDS request allows)
  if (d->lds_floor >= 0) lds = std::max(lds, std::min((size_t)d->lds_floor, (size_t)d->max_lds - 2048));
  HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (d->timing) HIP_TRY(hipEventRecord(d->ev0, stream));
  void *kargs[] = {&a};
  HIP_TRY(hipLaunchKernel(fn, dim3(B), dim3(threads), kargs, lds, stream));
  HIP_TRY(hipGetLastError());
  if (d->timing) HIP_TRY(hipEventRecord(d->ev1, stream));
  return CTCD_OK;
}

int ctcd_beam_decode(ctcd_decoder *d, const float *probs, const int32_t *seq_lens, int B, int T, int V, int beam,
                     int /*num_processes*/, double cutoff_prob, int cutoff_top_n, int blank_id, int log_input,
                     int32_t *out_tok, int32_t *out_ts, float *out_sc, int32_t *out_len, int32_t *n_results, void *stream_) {
  return decode_common(d, probs, seq_lens, B, T, V, beam, cutoff_prob, cutoff_top_n, blank_id, log_input, out_tok, out_ts, out_sc,
                       out_len, n_results, stream_, nullptr);
}

// ---- LM tier: the external scorer (binding.cpp:122-150,263-287)
int ctcd_scorer_create(ctcd_scorer **out, double alpha, double beta, const char *lm_path, const char *const *labels, int V,
                       int device_id) {
  if (!out || !lm_path || !labels || V <= 0) return fail(CTCD_EINVAL, "bad scorer arguments");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device_id < 0 || device_id >= ndev) return fail(CTCD_EINVAL, "no such HIP device");
  std::vector<std::string> lab(V);
  for (int i = 0; i < V; ++i) {
    if (!labels[i]) return fail(CTCD_EINVAL, "null label");
    lab[i] = labels[i];
  }
  ctcd_scorer *s = new ctcd_scorer;
  s->device = device_id;
  if (!s->host.build(alpha, beta, lm_path, lab)) {
    const std::string msg = s->host.error;
    delete s;
    return fail(msg.find("not supported") != std::string::npos || msg.find("supported for") != std::string::npos ? CTCD_EUNSUPPORTED : CTCD_EINVAL, msg);
  }
  CTC_ON_DEVICE(device_id);
  const ctclm::HostScorer &h = s->host;
  size_t off = 0;
  auto place = [&off](size_t bytes) { const size_t o = off; off = (off + bytes + 255) / 256 * 256; return o; };
  const size_t o_up = place(h.uni_prob.size() * 4), o_us = place(h.uni_state.size() * 4), o_bo = place(h.st_bo.size() * 4),
               o_fl = place(h.st_fail.size() * 4), o_ng = place(h.ng.size() * sizeof(ctclm::NgSlot)),
               o_dc = place(h.dict.size() * sizeof(ctclm::DictNode)), o_lw = place(h.label_word.size() * 4), o_dl = place(h.dict_lab.size() * 4);
  hipError_t e = hipMalloc((void **)&s->blob, off ? off : 256);
  if (e != hipSuccess) { delete s; return fail(CTCD_EHIP, std::string("hipMalloc: ") + hipGetErrorString(e)); }
  auto up = [&](size_t o, const void *src, size_t bytes) { return bytes ? hipMemcpy(s->blob + o, src, bytes, hipMemcpyHostToDevice) : hipSuccess; };
  if ((e = up(o_up, h.uni_prob.data(), h.uni_prob.size() * 4)) != hipSuccess || (e = up(o_us, h.uni_state.data(), h.uni_state.size() * 4)) != hipSuccess ||
      (e = up(o_bo, h.st_bo.data(), h.st_bo.size() * 4)) != hipSuccess || (e = up(o_fl, h.st_fail.data(), h.st_fail.size() * 4)) != hipSuccess ||
      (e = up(o_ng, h.ng.data(), h.ng.size() * sizeof(ctclm::NgSlot))) != hipSuccess ||
      (e = up(o_dc, h.dict.data(), h.dict.size() * sizeof(ctclm::DictNode))) != hipSuccess ||
      (e = up(o_lw, h.label_word.data(), h.label_word.size() * 4)) != hipSuccess ||
      (e = up(o_dl, h.dict_lab.data(), h.dict_lab.size() * 4)) != hipSuccess) {
    (void)hipFree(s->blob);
    delete s;
    return fail(CTCD_EHIP, std::string("hipMemcpy: ") + hipGetErrorString(e));
  }
  s->dview = h.view();
  s->dview.uni_prob = (const float *)(s->blob + o_up); s->dview.uni_state = (const uint32_t *)(s->blob + o_us);
  s->dview.st_bo = (const float *)(s->blob + o_bo); s->dview.st_fail = (const uint32_t *)(s->blob + o_fl);
  s->dview.ng = (const ctclm::NgSlot *)(s->blob + o_ng); s->dview.dict = (const ctclm::DictNode *)(s->blob + o_dc);
  s->dview.label_word = (const uint32_t *)(s->blob + o_lw);
  s->dview.dict_lab = (const uint32_t *)(s->blob + o_dl);
  *out = s;
  return CTCD_OK;
}

void ctcd_scorer_destroy(ctcd_scorer *s) {
  if (!s) return;
  DeviceGuard guard_(s->device);
  if (s->blob) (void)hipFree(s->blob);
  for (char *p : {s->cb_ng, s->cb_st, s->cb_uni, s->cb_miss, s->cb_stage})
    if (p) (void)hipFree(p);
  if (s->h_live) (void)hipHostFree(s->h_live);
  if (s->cb_stage_hp) (void)hipHostFree(s->cb_stage_hp);
  if (s->cb_stage_ev) (void)hipEventDestroy(s->cb_stage_ev);
  s->ask_pool.shutdown();
  delete s->cbl;
  delete s;
}

// ---- the host-side scorer hook (binding.cpp:122-150 hands the decoder an opaque scorer; scorer.h:41-78 is its interface)
// Brings the device copy of the cache up to date with the host's: whole tables after a rehash / growth, single slots otherwise.
__global__ void cb_scatter_kernel(ctclm::NgSlot *ng, const uint32_t *idx, const ctclm::NgSlot *slots, unsigned n) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ng[idx[i]] = slots[i];
}
static int cb_sync(ctcd_scorer *s, hipStream_t stream = nullptr) {
  ctclm::CallbackLm &c = *s->cbl;
  ctclm::HostScorer &h = c.hs;
  if (c.rehashed || s->cb_ng_slots != h.ng.size()) {
    if (s->cb_ng_slots != h.ng.size()) {
      if (s->cb_ng) (void)hipFree(s->cb_ng);
      s->cb_ng = nullptr;
      HIP_TRY(hipMalloc((void **)&s->cb_ng, h.ng.size() * sizeof(ctclm::NgSlot)));
      s->cb_ng_slots = h.ng.size();
    }
    HIP_TRY(hipMemcpy(s->cb_ng, h.ng.data(), h.ng.size() * sizeof(ctclm::NgSlot), hipMemcpyHostToDevice));
  } else if (!c.dirty.empty()) {
    // (round 5) the slots written since the last launch travel in ONE copy -- [their indices | the slots] -- and a kernel puts
    // them in place.  (Rounds 1-4: one blocking 16-byte hipMemcpy per slot -- ~10 us each, 500 a round at the configs[4] shape:
    // two thirds of a round's 8 ms.)
    const size_t n = c.dirty.size();
    if (s->cb_stage_cap < n) {  // (both ends of the copy grow together; the host end is page-locked: the copy is queued on `stream`, in front
                                //  of the scatter kernel, and its ordering comes from the stream -- ADVICE r5)
      if (stream) HIP_TRY(hipStreamSynchronize(stream)); else HIP_TRY(hipDeviceSynchronize());
      if (s->cb_stage) (void)hipFree(s->cb_stage);
      if (s->cb_stage_hp) (void)hipHostFree(s->cb_stage_hp);
      s->cb_stage = nullptr; s->cb_stage_hp = nullptr; s->cb_stage_cap = 0;
      const size_t cap = n * 2 + 1024;
      HIP_TRY(hipMalloc((void **)&s->cb_stage, cap * (4 + sizeof(ctclm::NgSlot))));
      HIP_TRY(hipHostMalloc((void **)&s->cb_stage_hp, cap * (4 + sizeof(ctclm::NgSlot)), hipHostMallocDefault));
      s->cb_stage_cap = cap;
    }
    if (!s->cb_stage_ev) HIP_TRY(hipEventCreateWithFlags(&s->cb_stage_ev, hipEventDisableTiming));
    HIP_TRY(hipEventSynchronize(s->cb_stage_ev));  // (the previous copy out of this block -- possibly queued by a call that has returned)
    uint32_t *hi = (uint32_t *)s->cb_stage_hp;
    ctclm::NgSlot *hsl = (ctclm::NgSlot *)(s->cb_stage_hp + n * 4);
    for (size_t k = 0; k < n; ++k) { hi[k] = c.dirty[k]; hsl[k] = h.ng[c.dirty[k]]; }
    HIP_TRY(hipMemcpyAsync(s->cb_stage, s->cb_stage_hp, n * (4 + sizeof(ctclm::NgSlot)), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(s->cb_stage_ev, stream));
    hipLaunchKernelGGL(cb_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (ctclm::NgSlot *)s->cb_ng, (const uint32_t *)s->cb_stage,
                       (const ctclm::NgSlot *)(s->cb_stage + n * 4), (unsigned)n);
    HIP_TRY(hipGetLastError());
  }
  c.rehashed = false;
  c.dirty.clear();
  if (s->cb_st_cap < h.st_bo.size()) {  // every state backs off to the empty context with weight 0: two zeroed arrays
    if (s->cb_st) (void)hipFree(s->cb_st);
    s->cb_st = nullptr;
    const size_t cap = std::max<size_t>(h.st_bo.size() * 2, s->h_live ? kCbLiveStates0 : (size_t)1 << 20);  // (head room: a waiting launch cannot have them replaced)
    HIP_TRY(hipMalloc((void **)&s->cb_st, cap * 8));
    HIP_TRY(hipMemset(s->cb_st, 0, cap * 8));
    s->cb_st_cap = cap;
  }
  s->dview.ng = (const ctclm::NgSlot *)s->cb_ng;
  s->dview.ng_mask = (uint32_t)h.ng.size() - 1;
  s->dview.st_bo = (const float *)s->cb_st;
  s->dview.st_fail = (const uint32_t *)(s->cb_st + s->cb_st_cap * 4);
  return CTCD_OK;
}

int ctcd_scorer_create_callback(ctcd_scorer **out, double alpha, double beta, int max_order, const char *const *vocabulary, int n_vocabulary,
                                ctcd_cond_log10_fn fn, void *user, const char *const *labels, int V, int device_id) {
  if (!out || !fn || !labels || V <= 0 || n_vocabulary < 0 || (n_vocabulary && !vocabulary)) return fail(CTCD_EINVAL, "bad scorer arguments");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device_id < 0 || device_id >= ndev) return fail(CTCD_EINVAL, "no such HIP device");
  std::vector<std::string> lab(V), voc(n_vocabulary);
  for (int i = 0; i < V; ++i) { if (!labels[i]) return fail(CTCD_EINVAL, "null label"); lab[i] = labels[i]; }
  for (int i = 0; i < n_vocabulary; ++i) { if (!vocabulary[i]) return fail(CTCD_EINVAL, "null vocabulary word"); voc[i] = vocabulary[i]; }
  ctcd_scorer *s = new ctcd_scorer;
  s->device = device_id;
  s->cbl = new ctclm::CallbackLm;
  s->ask_pool.cl = s->cbl;
  if (!s->cbl->build(alpha, beta, max_order, voc, lab, (ctclm::CondLog10Fn)fn, user)) {
    const std::string msg = s->cbl->hs.error;
    ctcd_scorer_destroy(s);
    return fail(CTCD_EINVAL, msg);
  }
  const ctclm::HostScorer &h = s->cbl->hs;
  // the scalar facts the accessors and the launches read (the tables stay with the callback object)
  s->host.alpha = alpha; s->host.beta = beta; s->host.order = h.order; s->host.char_based = h.char_based; s->host.dict_size = h.dict_size;
  s->host.space_id = h.space_id; s->host.labels = h.labels; s->host.dict_wide = h.dict_wide;
  CTC_ON_DEVICE(device_id);
  size_t off = 0;
  auto place = [&off](size_t bytes) { const size_t o = off; off = (off + bytes + 255) / 256 * 256; return o; };
  const size_t o_up = place(h.uni_prob.size() * 4), o_us = place(h.uni_state.size() * 4), o_dc = place(h.dict.size() * sizeof(ctclm::DictNode)),
               o_lw = place(h.label_word.size() * 4), o_dl = place(h.dict_lab.size() * 4);
  hipError_t e = hipMalloc((void **)&s->cb_uni, off ? off : 256);
  if (e == hipSuccess) e = hipMalloc((void **)&s->cb_miss, (size_t)kCbMissCap * sizeof(ctclm::MissEntry) + 256);
  if (e != hipSuccess) { ctcd_scorer_destroy(s); return fail(CTCD_EHIP, std::string("hipMalloc: ") + hipGetErrorString(e)); }
  auto up = [&](size_t o, const void *src, size_t bytes) { return bytes ? hipMemcpy(s->cb_uni + o, src, bytes, hipMemcpyHostToDevice) : hipSuccess; };
  if ((e = up(o_up, h.uni_prob.data(), h.uni_prob.size() * 4)) != hipSuccess || (e = up(o_us, h.uni_state.data(), h.uni_state.size() * 4)) != hipSuccess ||
      (e = up(o_dc, h.dict.data(), h.dict.size() * sizeof(ctclm::DictNode))) != hipSuccess ||
      (e = up(o_lw, h.label_word.data(), h.label_word.size() * 4)) != hipSuccess || (e = up(o_dl, h.dict_lab.data(), h.dict_lab.size() * 4)) != hipSuccess) {
    ctcd_scorer_destroy(s);
    return fail(CTCD_EHIP, std::string("hipMemcpy: ") + hipGetErrorString(e));
  }
  s->dview = h.view();
  s->dview.uni_prob = (const float *)(s->cb_uni + o_up); s->dview.uni_state = (const uint32_t *)(s->cb_uni + o_us);
  s->dview.dict = (const ctclm::DictNode *)(s->cb_uni + o_dc); s->dview.label_word = (const uint32_t *)(s->cb_uni + o_lw);
  s->dview.dict_lab = (const uint32_t *)(s->cb_uni + o_dl);
  s->dview.cb = 1;
  s->dview.cb_count = (unsigned *)s->cb_miss;
  s->dview.cb_miss = (ctclm::MissEntry *)(s->cb_miss + 256);
  s->dview.cb_cap = kCbMissCap;
  // the miss list itself lives in page-locked memory the host reads while a launch runs (the counter stays on the device)
  if (hipHostMalloc((void **)&s->h_live, kLiveBytes, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
      hipHostGetDevicePointer((void **)&s->d_live, s->h_live, 0) == hipSuccess) {
    std::memset(s->h_live, 0, kLiveOffMiss);
    std::memset(s->h_live + kLiveOffMiss, 0xff, (size_t)kCbMissCap * sizeof(ctclm::MissEntry));
    s->dview.cb_miss = (ctclm::MissEntry *)(s->d_live + kLiveOffMiss);
  } else {
    if (s->h_live) (void)hipHostFree(s->h_live);
    s->h_live = nullptr; s->d_live = nullptr;
    (void)hipGetLastError();
  }
  if (s->h_live) {  // (a waiting launch cannot have its tables moved: they get their room here, not in the first decode)
    s->cbl->grow(kCbLiveSlots0);
    s->live_stamp.assign(s->cbl->hs.ng.size(), 0u);
  }
  const int rc = cb_sync(s);
  if (rc) { ctcd_scorer_destroy(s); return rc; }
  *out = s;
  return CTCD_OK;
}
int ctcd_scorer_is_character_based(const ctcd_scorer *s) { return s ? (s->host.char_based ? 1 : 0) : -1; }
int ctcd_scorer_max_order(const ctcd_scorer *s) { return s ? s->host.order : -1; }
int ctcd_scorer_dict_size(const ctcd_scorer *s) { return s ? s->host.dict_size : -1; }
int ctcd_scorer_reset_params(ctcd_scorer *s, double alpha, double beta) {
  if (!s) return fail(CTCD_EINVAL, "scorer == NULL");
  s->host.alpha = alpha;
  s->host.beta = beta;
  return CTCD_OK;
}
double ctcd_scorer_cond_log_prob(const ctcd_scorer *s, const char *const *words, int n) {
  if (!s || !words || n < 0) return 0.0;
  if (s->cbl) {  // the callback itself, with the reference's conversion (scorer.cpp:74-93)
    float p10 = 0.f;
    const int rc = n > 0 ? s->cbl->fn(s->cbl->user, words, n, &p10) : 1;
    return rc == 0 ? (double)p10 / (double)0.4342944819f : ctclm::kOovScore;
  }
  std::vector<std::string> w(n);
  for (int i = 0; i < n; ++i) w[i] = words[i] ? words[i] : "";
  return s->host.cond_log_prob(w);
}
// ... and in the form the scorer hook speaks (ctcd_cond_log10_fn): the float32 log10 probability before the reference's
// conversion; returns 1 for a window with an unknown word.  Lets the built-in tables serve as a callback (tests, adapters).
int ctcd_scorer_cond_log10(const ctcd_scorer *s, const char *const *words, int n, float *log10_prob) {
  if (!s || !words || n <= 0 || !log10_prob) return -1;
  if (s->cbl) return s->cbl->fn(s->cbl->user, words, n, log10_prob);
  // (HostScorer::cond_log10 without the vector of strings: this function is what bench.py puts behind the hook as a native callback)
  const ctclm::LmView v = s->host.view();
  uint32_t st = 0;
  float p = 0.f;
  std::string key;
  for (int i = 0; i < n; ++i) {
    key.assign(words[i] ? words[i] : "");
    const uint32_t id = s->host.id_of(key);
    if (id == 0) return 1;
    p = ctclm::lm_score(v, st, id, &st);
  }
  *log10_prob = p;
  return 0;
}
int ctcd_scorer_set_callback_threads(ctcd_scorer *s, int threads) {
  if (!s || !s->cbl) return fail(CTCD_EINVAL, "not a callback scorer");
  if (threads < 1 || threads > 64) return fail(CTCD_EINVAL, "callback threads must be in [1, 64]");
  std::lock_guard<std::mutex> lk(s->cb_mu);
  s->ask_pool.start(threads, s->cbl);
  return CTCD_OK;
}
long long ctcd_scorer_callback_calls(const ctcd_scorer *s) { return s && s->cbl ? (long long)s->cbl->queries : 0; }
double ctcd_scorer_callback_seconds(const ctcd_scorer *s) { return s && s->cbl ? s->cbl->cb_seconds : 0.0; }

// Decoding with a callback scorer (ctcd_scorer_create_callback).  A launch decodes until an utterance asks for a (history, word)
// pair the device cache does not hold, parks that utterance in front of the frame it was in (every utterance runs as a stream:
// the caller's own, or a temporary one) and queues the pair; the host asks the callback, inserts the answers and launches
// again -- the parked utterances resume where they stopped, the finished ones pass through untouched -- until every utterance
// has consumed its rows.  Results equal those of a scorer whose tables were complete from the start (tests: the built-in ARPA
// tables behind the callback, bit for bit).  A capability, not a fast path: a launch per round of misses plus the callback's
// own time; a warm cache needs one launch.
static int cb_rounds(ctcd_decoder *d, ctcd_stream **states, const unsigned char *is_eos, const int32_t *lens, const float *probs, int B, int T, int V,
                     int beam, double cutoff_prob, int cutoff_top_n, int blank_id, int log_input, ctcd_scorer *scorer, int32_t *out_tok,
                     int32_t *out_ts, float *out_sc, int32_t *out_len, int32_t *n_results, int out_T, void *stream_, const CompactOut *co = nullptr) {
  // (the callback runs under this lock: a callback that decodes with the same scorer deadlocks -- include/ctcdecode_amd.h)
  std::lock_guard<std::mutex> cache_lock(scorer->cb_mu);
  hipStream_t stream = (hipStream_t)stream_;
  int rc0;
  if ((rc0 = d->cb_ctl.ensure((size_t)B * 12 + 256))) return rc0;  // [frame offsets | frames done | row lengths] (kept: no hipMalloc / hipFree per call)
  int *d_off = (int *)d->cb_ctl.p, *d_done = d_off + B, *d_rows = d_done + B;
  if (d->h_cb_items < (size_t)B) {
    if (d->h_cb) (void)hipHostFree(d->h_cb);
    d->h_cb = nullptr;
    HIP_TRY(hipHostMalloc((void **)&d->h_cb, ((size_t)B * 2 + 4) * 4, hipHostMallocDefault));
    d->h_cb_items = (size_t)B;
  }
  int32_t *st_h = d->h_cb, *fd_h = d->h_cb + B;
  unsigned *nmiss_h = (unsigned *)(d->h_cb + 2 * (size_t)B);
  HIP_TRY(hipMemcpyAsync(d_rows, lens, (size_t)B * 4, hipMemcpyHostToDevice, stream));
  std::vector<int32_t> done(B, 0), rem(B);
  std::vector<unsigned char> eos(B), finished(B, 0);
  std::vector<uint32_t> miss;
  int rc;
  // The launch that waits (decode_kernel.h KernelArgs::cb_log_len): a workgroup whose utterance misses stays on its CU while this
  // thread -- polling the miss list in page-locked memory -- asks the callback and publishes the answered slots; a round of misses
  // costs two PCIe crossings instead of a launch.  What it cannot do while the kernel runs -- move the table (rehash), lengthen the
  // state arrays, hold more than its log -- ends the wait: the workgroups leave as they always did and the loop below relaunches.
  static const bool live_off = getenv("CTCD_HOOK_WAIT") && atoi(getenv("CTCD_HOOK_WAIT")) == 0;
  const bool live = scorer->h_live && !live_off && !d->no_hook_wait && (uint32_t)B <= kCbLiveItems;
  ctclm::CallbackLm &cl = *scorer->cbl;
  d->last_cb_waits = 0;
  for (int round = 0;; ++round) {
    d->last_cb_rounds = round;
    if (round > 4 * T + 64) return fail(CTCD_EINTERNAL, "scorer hook: the decode does not make progress");
    if (live) {  // head room for the answers of a waiting launch: 8192 more slots / states before anything has to move
      if ((cl.used + 8192) * 2 > cl.hs.ng.size()) cl.grow(std::max<size_t>(cl.hs.ng.size() * 4, kCbLiveSlots0));  // (64 MB to begin with)
      if (cl.n_states() + 8192 >= scorer->cb_st_cap / 2 && cl.hs.st_bo.size() <= scorer->cb_st_cap) {  // (cb_sync doubles the device arrays)
        cl.hs.st_bo.resize(scorer->cb_st_cap + 64, 0.0f);
        cl.hs.st_fail.resize(cl.hs.st_bo.size(), 0u);
      }
    }
    if ((rc = cb_sync(scorer, stream))) return rc;
    int left = 0;
    bool any_eos = false;
    for (int b = 0; b < B; ++b) {
      rem[b] = finished[b] ? 0 : lens[b] - done[b];
      eos[b] = finished[b] ? 0 : is_eos[b];
      any_eos |= eos[b] != 0;
      left += !finished[b];
    }
    if (!left) break;
    HIP_TRY(hipMemcpyAsync(d_off, done.data(), (size_t)B * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemsetAsync(scorer->cb_miss, 0, 4, stream));
    StreamCall sc{states, eos.data(), out_T, rem.data(), any_eos};
    sc.no_clear = round > 0;
    sc.frame_off = d_off;
    sc.frames_done = d_done;
    sc.row_lens = d_rows;
    volatile unsigned *h_len = (volatile unsigned *)scorer->h_live;
    volatile int32_t *h_done = (volatile int32_t *)(scorer->h_live + kLiveOffDone);
    volatile uint32_t *h_miss = (volatile uint32_t *)(scorer->h_live + kLiveOffMiss);  // (four words per pair: ctclm::MissEntry)
    volatile unsigned *h_ans = (volatile unsigned *)(scorer->h_live + kLiveOffAns);
    if (scorer->h_live) {  // (the miss list of the last launch: back to "not written")
      std::memset(scorer->h_live + kLiveOffMiss, 0xff, (size_t)scorer->live_miss_hw * sizeof(ctclm::MissEntry));
      scorer->live_miss_hw = 0;
    }
    if (live) {
      *h_len = 0;
      for (int b = 0; b < B; ++b) { h_done[b] = finished[b] ? 1 : 0; h_ans[b] = 0; }
      std::atomic_thread_fence(std::memory_order_seq_cst);
      sc.live = scorer->d_live;
    }
    if ((rc = decode_common(d, probs, nullptr, B, T, V, beam, cutoff_prob, cutoff_top_n, blank_id, log_input, out_tok, out_ts, out_sc, out_len,
                            n_results, stream_, &sc, scorer, co)))
      return rc;
    // (one page-locked block: the three reports arrive behind the launch without staging copies)
    HIP_TRY(hipMemcpyAsync(st_h, d->status.p, (size_t)B * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(fd_h, d_done, (size_t)B * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(nmiss_h, scorer->cb_miss, 4, hipMemcpyDeviceToHost, stream));
    unsigned long long seen = 0;  // miss-list entries answered while the launch ran (the list is a ring then: entry i sits at i mod capacity)
    if (live) {
      uint32_t *l_idx = (uint32_t *)(scorer->h_live + kLiveOffIdx);
      ctclm::NgSlot *l_slot = (ctclm::NgSlot *)(scorer->h_live + kLiveOffSlot);
      // a slot goes into the log once per launch: a workgroup that asks for a pair the log already holds finds it there (it applies
      // the log before it waits), so the duplicates -- every park queues what its pending entries need -- cost a table probe each
      if (scorer->live_stamp.size() != cl.hs.ng.size()) scorer->live_stamp.assign(cl.hs.ng.size(), 0u);
      const uint32_t launch_id = ++scorer->live_launch;
      scorer->live_memo.assign((size_t)1 << 16, 0ull);
      unsigned log_n = 0;
      bool give_up = false;
      std::vector<unsigned> ans_local((size_t)B, 0u);
      auto t_last = std::chrono::steady_clock::now();
      static const bool trace = getenv("CTCD_HOOK_TIMING") != nullptr;  // (stderr: how busy this thread is while a launch waits on it)
      const auto t_begin = t_last;
      double busy = 0.0;
      scorer->ask_pool.begin();
      for (unsigned spin = 0;; ++spin) {
        const unsigned long long s0 = seen;
        const auto t_in = trace ? std::chrono::steady_clock::now() : t_begin;
        unsigned fresh = 0;
        for (bool more = true; more && !give_up;) {
          // The pairs that have arrived, up to 128 at a time, in three passes.  (1) classify: answered moments ago (memo), cached, or new to
          // the cache -- the table lines were requested while the pairs were read (a cold probe of a table of tens of MB is a DRAM access),
          // the words of a new window are looked up and the line of its next state requested; (2) ask the callback about the new windows
          // (the scorer's helper threads take a share each when it has any); (3) in the order of the list: cache the answers, append to
          // the log, publish per item.
          enum { kMemo = 0, kHave = 1, kNew = 2, kDup = 3 };
          constexpr int kBatch = 128;
          struct Pend { uint32_t st, wd, item, at; size_t ring; int kind; } pend[kBatch];
          ctclm::CallbackLm::Ask asks[kBatch];
          int np = 0, na = 0;
          const uint32_t mask = (uint32_t)cl.hs.ng.size() - 1;
          while (np < kBatch) {
            const size_t at_ring = (size_t)((seen + (unsigned)np) & (kCbMissCap - 1));
            if (h_miss[4 * at_ring + 3] == 0xFFFFFFFFu) { more = false; break; }  // (the flag word: the pair's 16 bytes arrive in one piece)
            std::atomic_thread_fence(std::memory_order_acquire);
            pend[np] = Pend{h_miss[4 * at_ring], h_miss[4 * at_ring + 1], h_miss[4 * at_ring + 2], 0u, at_ring, kMemo};
            const uint32_t hh = ctclm::ng_hash(pend[np].st, pend[np].wd) & mask;
            __builtin_prefetch(&cl.hs.ng[hh]);
            __builtin_prefetch(&scorer->live_stamp[hh]);
            ++np;
          }
          bool bad = false;
          int usable = np;  // (the table runs out of room at this pair: the ones before it are served, then the wait ends)
          for (int q = 0; q < np; ++q) {
            Pend &e = pend[q];
            // most of the list repeats pairs asked moments ago (every park queues what its pending entries need, neighbouring prefixes
            // and utterances want the same windows): a small direct-mapped memo of the pairs already in this launch's log answers
            // those without touching the table
            const unsigned long long key = ((unsigned long long)e.st << 32) | e.wd;
            if (scorer->live_memo[(size_t)((key * 0x9E3779B97F4A7C15ull) >> 48)] == key) continue;  // kMemo
            const long long have = cl.find_slot(e.st, e.wd);
            if (have >= 0) { e.kind = kHave; e.at = (uint32_t)have; continue; }
            e.kind = kNew;
            for (int o = 0; o < q; ++o)  // (the same new pair twice in one batch: the callback is asked once per window)
              if (pend[o].kind == kNew && pend[o].st == e.st && pend[o].wd == e.wd) { e.kind = kDup; break; }
            if (e.kind == kDup) continue;
            if (!cl.room_for((size_t)na + 1, scorer->cb_st_cap)) { usable = q; give_up = true; break; }  // (the tables would have to move: between launches)
            if (!cl.prepare(e.st, e.wd, asks[na])) { bad = true; break; }
            cl.prefetch_next_state(e.st, e.wd);
            ++na;
          }
          if (!bad && na) {
            const auto t0 = std::chrono::steady_clock::now();
            scorer->ask_pool.ask_all(asks, na);
            cl.cb_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
          }
          int ia = 0;
          bool log_full = false;
          for (int q = 0; q < usable && !bad; ++q) {
            Pend &e = pend[q];
            const size_t at_ring = e.ring;
            if (log_full && e.kind == kMemo) continue;
            const unsigned long long key = ((unsigned long long)e.st << 32) | e.wd;
            if (e.kind != kMemo) {
              uint32_t at = e.at;
              if (e.kind == kNew) {
                if (!cl.commit(asks[ia++], &at)) { bad = true; break; }
              } else if (e.kind == kDup) {
                at = (uint32_t)cl.find_slot(e.st, e.wd);
              }
              if (log_full) continue;  // (the answers that have been paid for are cached; their pairs stay on the list for the next launch)
              if (scorer->live_stamp[at] != launch_id) {
                if (log_n >= kCbLogCap) { give_up = true; log_full = true; continue; }
                scorer->live_stamp[at] = launch_id;
                l_idx[log_n] = at;
                l_slot[log_n] = cl.hs.ng[at];
                ++log_n;
                ++fresh;
                std::atomic_thread_fence(std::memory_order_release);
                *h_len = log_n;  // (published entry by entry: the stores of this thread arrive in order)
              }
              scorer->live_memo[(size_t)((key * 0x9E3779B97F4A7C15ull) >> 48)] = key;  // (in the log from here on)
            }
            if (e.item < (uint32_t)B) {  // the item's workgroup goes on when every pair it queued has been dealt with
              std::atomic_thread_fence(std::memory_order_release);
              h_ans[e.item] = ++ans_local[e.item];
            }
            h_miss[4 * at_ring] = 0xFFFFFFFFu; h_miss[4 * at_ring + 1] = 0xFFFFFFFFu; h_miss[4 * at_ring + 2] = 0xFFFFFFFFu; h_miss[4 * at_ring + 3] = 0xFFFFFFFFu;
            ++seen;
          }
          if (bad) {
            scorer->ask_pool.end();
            *h_len = 0xFFFFFFFFu;
            (void)hipStreamSynchronize(stream);
            scorer->live_miss_hw = kCbMissCap;
            return fail(CTCD_EINVAL, cl.hs.error);
          }
        }
        if (fresh) ++d->last_cb_waits;
        if (seen > s0) {
          t_last = std::chrono::steady_clock::now();
          if (trace) busy += std::chrono::duration<double>(t_last - t_in).count();
        }
        bool all = true;
        for (int b = 0; b < B && all; ++b) all = h_done[b] != 0;
        if (all) break;
        if (!give_up && (spin & 1023) == 1023) {
          // (nothing new for a second with workgroups still out, or a launch that has ended some other way)
          if (std::chrono::steady_clock::now() - t_last > std::chrono::seconds(1) || hipStreamQuery(stream) != hipErrorNotReady) give_up = true;
          (void)hipGetLastError();
        }
        if (give_up) { *h_len = 0xFFFFFFFFu; break; }
        __builtin_ia32_pause();
      }
      scorer->ask_pool.end();
      if (trace) fprintf(stderr, "scorer hook, waiting launch %d: %.2f ms, this thread busy %.2f ms with %llu queued pairs (%u new to the launch's log)%s\n", round,
                         1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(), 1e3 * busy, seen, log_n, give_up ? "; gave up" : "");
    }
    HIP_TRY(hipStreamSynchronize(stream));
    const unsigned nmiss = *nmiss_h;
    scorer->live_miss_hw = live ? kCbMissCap : (nmiss < kCbMissCap ? nmiss : kCbMissCap);  // (a ring may hold unconsumed pairs anywhere)
    bool need = false;
    for (int b = 0; b < B; ++b) {
      if (finished[b]) continue;
      if (st_h[b] == ST_OK) { finished[b] = 1; continue; }
      if (st_h[b] == ST_CB_DANGER)
        return fail(CTCD_EUNSUPPORTED, "a callback scorer does not support rows with infinite / overflowing log-probabilities (item " + std::to_string(b) + ")");
      if (st_h[b] != ST_NEED_HOST) return fail(CTCD_EINTERNAL, "decoder status " + std::to_string(st_h[b]) + " for item " + std::to_string(b));
      done[b] = (int32_t)(fd_h[b] - states[b]->frames);  // (the parked state counts the stream's frames; the rows, this call's)
      need = true;
    }
    if (need) {
      if (nmiss == 0) return fail(CTCD_EINTERNAL, "scorer hook: an utterance waits for the host but queued nothing");
      if (live) {  // what the ring still holds (the launch has ended: every pair queued is written)
        for (size_t i = 0; i < (size_t)kCbMissCap; ++i)
          if (h_miss[4 * i + 3] != 0xFFFFFFFFu && !scorer->cbl->resolve(h_miss[4 * i], h_miss[4 * i + 1])) return fail(CTCD_EINVAL, scorer->cbl->hs.error);
      } else {
        const unsigned take = nmiss < kCbMissCap ? nmiss : kCbMissCap;  // (pairs beyond the list's capacity are asked for again next round)
        miss.resize((size_t)4 * take);
        if (scorer->h_live) {  // (the list is in page-locked memory)
          for (size_t i = 0; i < (size_t)4 * take; ++i) miss[i] = h_miss[i];
        } else {
          HIP_TRY(hipMemcpy(miss.data(), scorer->cb_miss + 256, (size_t)take * sizeof(ctclm::MissEntry), hipMemcpyDeviceToHost));
        }
        for (unsigned i = 0; i < take; ++i)
          if (!scorer->cbl->resolve(miss[4 * i], miss[4 * i + 1])) return fail(CTCD_EINVAL, scorer->cbl->hs.error);
      }
    }
  }
  return CTCD_OK;
}

// ... for a whole batch: temporary streams in one allocation, a block per utterance sized for all T frames
static int decode_lm_callback(ctcd_decoder *d, const float *probs, const int32_t *seq_lens, int B, int T, int V, int beam, double cutoff_prob,
                              int cutoff_top_n, int blank_id, int log_input, ctcd_scorer *scorer, int32_t *out_tok, int32_t *out_ts,
                              float *out_sc, int32_t *out_len, int32_t *n_results, void *stream_, const CompactOut *co = nullptr) {
  if (!d) return fail(CTCD_EINVAL, "decoder == NULL");
  if (B <= 0 || T < 0 || beam <= 0 || beam > kMaxBeam) return B == 0 ? CTCD_OK : fail(CTCD_EINVAL, "bad arguments");
  CTC_ON_DEVICE(d->device);
  hipStream_t stream = (hipStream_t)stream_;
  std::vector<int32_t> len(B, T);
  if (seq_lens) {
    HIP_TRY(hipMemcpyAsync(len.data(), seq_lens, (size_t)B * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    for (int b = 0; b < B; ++b) len[b] = len[b] < 0 ? 0 : (len[b] > T ? T : len[b]);  // binding.cpp:64-65
  }
  const long long capf = T > 0 ? T : 1;
  const size_t blk = (stream_block_bytes(capf, beam) + 255) / 256 * 256;
  // (round 5: the blocks stay with the decoder -- rounds 1-4 allocated and freed them in every call, 384 MB at the configs[4]
  //  shape, and zeroed them with two memsets per utterance: a warm cache cost three times the built-in tables' decode)
  int rcb;
  if ((rcb = d->cb_blocks.ensure(blk * (size_t)B))) return rcb;
  char *blocks = (char *)d->cb_blocks.p;
  std::vector<ctcd_stream> sts(B);
  std::vector<ctcd_stream *> states(B);
  for (int b = 0; b < B; ++b) {
    ctcd_stream &st = sts[b];
    st.device = d->device; st.scorer = scorer; st.block = blocks + (size_t)b * blk; st.bytes = blk; st.V = V; st.beam = beam; st.frames = 0; st.cap_frames = capf;
    states[b] = &st;
  }
  // frames == 0: the first launch initialises the beam; the high parts of the nodes' time steps start at zero
  HIP_TRY(hipMemset2DAsync(blocks, blk, 0, stream_pool_offset(beam), (size_t)B, stream));
  HIP_TRY(hipMemset2DAsync(blocks + stream_thi_offset(capf, beam), blk, 0, stream_nodes(capf, beam) * sizeof(int), (size_t)B, stream));
  const std::vector<unsigned char> eos(B, 1);
  return cb_rounds(d, states.data(), eos.data(), len.data(), probs, B, T, V, beam, cutoff_prob, cutoff_top_n, blank_id, log_input, scorer, out_tok,
                   out_ts, out_sc, out_len, n_results, T, stream_, co);
}

int ctcd_beam_decode_lm(ctcd_decoder *d, const float *probs, const int32_t *seq_lens, int B, int T, int V, int beam,
                        int /*num_processes*/, double cutoff_prob, int cutoff_top_n, int blank_id, int log_input, ctcd_scorer *scorer,
                        int32_t *out_tok, int32_t *out_ts, float *out_sc, int32_t *out_len, int32_t *n_results, void *stream_) {
  if (!scorer) return fail(CTCD_EINVAL, "scorer == NULL (use ctcd_beam_decode)");
  if (scorer->cbl)
    return decode_lm_callback(d, probs, seq_lens, B, T, V, beam, cutoff_prob, cutoff_top_n, blank_id, log_input, scorer, out_tok, out_ts, out_sc,
                              out_len, n_results, stream_);
  return decode_common(d, probs, seq_lens, B, T, V, beam, cutoff_prob, cutoff_top_n, blank_id, log_input, out_tok, out_ts, out_sc,
                       out_len, n_results, stream_, nullptr, scorer);
}

// ---- streaming: DecoderState kept between calls (ctcdecode/__init__.py:143-272, binding.cpp:153-265)
int ctcd_stream_create_lm(ctcd_decoder *d, ctcd_stream **out, int V, int beam, int frames_hint, ctcd_scorer *scorer) {
  int rc = ctcd_stream_create(d, out, V, beam, frames_hint);
  if (rc == CTCD_OK) (*out)->scorer = scorer;
  return rc;
}

int ctcd_stream_create(ctcd_decoder *d, ctcd_stream **out, int V, int beam, int frames_hint) {
  if (!d || !out || V <= 0 || beam <= 0 || beam > kMaxBeam) return fail(CTCD_EINVAL, "bad stream parameters");
  CTC_ON_DEVICE(d->device);
  ctcd_stream *st = new ctcd_stream;
  st->device = d->device;
  st->V = V;
  st->beam = beam;
  st->cap_frames = frames_hint > 0 ? frames_hint : 1024;
  st->bytes = stream_block_bytes(st->cap_frames, beam);
  hipError_t e = hipMalloc((void **)&st->block, st->bytes);
  if (e != hipSuccess) { delete st; return fail(CTCD_EHIP, std::string("hipMalloc: ") + hipGetErrorString(e)); }
  e = hipMemset(st->block, 0, stream_pool_offset(beam));  // frames == 0: the first call initialises the beam
  if (e == hipSuccess) e = hipMemset(st->block + stream_thi_offset(st->cap_frames, beam), 0, stream_nodes(st->cap_frames, beam) * sizeof(int));
  if (e != hipSuccess) { (void)hipFree(st->block); delete st; return fail(CTCD_EHIP, std::string("hipMemset: ") + hipGetErrorString(e)); }
  *out = st;
  return CTCD_OK;
}

void ctcd_stream_destroy(ctcd_decoder *d, ctcd_stream *st) {
  if (!st) return;
  DeviceGuard guard_(st->device);
  if (st->block) (void)hipFree(st->block);
  delete st;
}

long long ctcd_stream_frames(const ctcd_stream *st) { return st ? st->frames : -1; }

// what every streaming call does first: the states are checked against the call, the chunk lengths clamped (binding.cpp:171),
// node pools that the chunk would overflow grown
static int stream_prepare(ctcd_decoder *d, ctcd_stream **states, const unsigned char *is_eos, const int32_t *seq_lens_host, int B, int T, int V,
                          int beam, int out_T, hipStream_t stream, std::vector<int32_t> &lens, bool &any_eos) {
  const unsigned long long call_id = ++g_stream_call_id;
  any_eos = false;
  for (int b = 0; b < B; ++b) {
    ctcd_stream *st = states[b];
    if (!st || st->V != V || st->beam != beam) return fail(CTCD_EINVAL, "stream state does not match the decoder configuration");
    if (st->scorer != states[0]->scorer) return fail(CTCD_EINVAL, "the streams of one batch must share their scorer");
    if (st->seen_in_call == call_id) return fail(CTCD_EINVAL, "the same stream state appears twice in one batch");
    st->seen_in_call = call_id;
    any_eos |= is_eos[b] != 0;
    int len = seq_lens_host ? seq_lens_host[b] : T;
    len = len < 0 ? 0 : (len > T ? T : len);  // binding.cpp:171
    lens[b] = len;
    if (is_eos[b] && st->frames + len > out_T) return fail(CTCD_EINVAL, "out_T is smaller than the number of frames of a finishing stream");
    if ((st->frames + len) * (long long)beam + 1 > 0x7fffffffLL) return fail(CTCD_EUNSUPPORTED, "stream too long for this beam width");
    if (st->frames + len > st->cap_frames) {  // grow the node pool (device-to-device copy of the parked state)
      long long cap = st->cap_frames * 2;
      while (cap < st->frames + len) cap *= 2;
      const size_t bytes = stream_block_bytes(cap, beam);
      char *nb = nullptr;
      HIP_TRY(hipMalloc((void **)&nb, bytes));
      HIP_TRY(hipStreamSynchronize(stream));
      const size_t used = stream_nodes(st->frames, beam), off = stream_pool_offset(beam);
      HIP_TRY(hipMemcpy(nb, st->block, off + used * sizeof(PoolNode), hipMemcpyDeviceToDevice));
      HIP_TRY(hipMemcpy(nb + off + stream_nodes(cap, beam) * sizeof(PoolNode),
                        st->block + off + stream_nodes(st->cap_frames, beam) * sizeof(PoolNode), used * sizeof(int), hipMemcpyDeviceToDevice));
      HIP_TRY(hipMemset(nb + stream_thi_offset(cap, beam), 0, stream_nodes(cap, beam) * sizeof(int)));
      HIP_TRY(hipMemcpy(nb + stream_thi_offset(cap, beam), st->block + stream_thi_offset(st->cap_frames, beam), used * sizeof(int), hipMemcpyDeviceToDevice));
      (void)hipFree(st->block);
      st->block = nb;
      st->bytes = bytes;
      st->cap_frames = cap;
    }
  }
  return CTCD_OK;
}

int ctcd_stream_decode(ctcd_decoder *d, ctcd_stream **states, const unsigned char *is_eos, const float *probs,
                       const int32_t *seq_lens_host, int B, int T, int V, int beam, int /*num_processes*/, double cutoff_prob,
                       int cutoff_top_n, int blank_id, int log_input, int32_t *out_tok, int32_t *out_ts, float *out_sc,
                       int32_t *out_len, int32_t *n_results, int out_T, void *stream_) {
  if (!d || !states || !is_eos || B < 0 || T < 0 || out_T < 0) return fail(CTCD_EINVAL, "bad arguments");
  if (B == 0) return CTCD_OK;
  CTC_ON_DEVICE(d->device);
  hipStream_t stream = (hipStream_t)stream_;
  std::vector<int32_t> lens(B);
  bool any_eos = false;
  {
    const int prc = stream_prepare(d, states, is_eos, seq_lens_host, B, T, V, beam, out_T, stream, lens, any_eos);
    if (prc) return prc;
  }
  // (the chunk lengths travel with the other per-item arguments: decode_common)
  if (states[0]->scorer && states[0]->scorer->cbl) {  // a callback scorer: as many launches as its cache needs
    const int rc = cb_rounds(d, states, is_eos, lens.data(), probs, B, T, V, beam, cutoff_prob, cutoff_top_n, blank_id, log_input, states[0]->scorer, out_tok,
                             out_ts, out_sc, out_len, n_results, out_T, stream_);
    if (rc) return rc;
    for (int b = 0; b < B; ++b) states[b]->frames += lens[b];
    return CTCD_OK;
  }
  StreamCall sc{states, is_eos, out_T, lens.data(), any_eos};
  int rc = decode_common(d, probs, nullptr, B, T, V, beam, cutoff_prob, cutoff_top_n, blank_id, log_input, out_tok,
                         out_ts, out_sc, out_len, n_results, stream_, &sc, states[0]->scorer);
  if (rc) return rc;
  for (int b = 0; b < B; ++b) states[b]->frames += lens[b];
  return CTCD_OK;
}

// The streaming call with its results delivered to HOST memory, sized as the reference sizes them (binding.cpp:186-205: tokens /
// timesteps [B, R, L] with R = the most results of any item that ended, L = the longest of their label sequences).  R and L are
// known only when the kernel has run: the caller hands over an allocator, called once with (R, L), that returns the two buffers
// (the reference's binding resizes its tensors at that point).  The results leave the GPU in compact form -- the kernel mirrors
// every finished stream's records into page-locked memory -- and host threads expand them into the caller's buffers: a tenth of the
// padded [B, K, out_T] pair crosses PCIe.  out_scores / out_lens [B, beam] and n_results [B] are host pointers too.  Synchronous.
// A callback scorer decodes launch after launch into padded device tensors (cb_rounds): not through this entry (CTCD_EUNSUPPORTED).
int ctcd_stream_decode_to_host(ctcd_decoder *d, ctcd_stream **states, const unsigned char *is_eos, const float *probs,
                               const int32_t *seq_lens_host, int B, int T, int V, int beam, int num_processes, double cutoff_prob,
                               int cutoff_top_n, int blank_id, int log_input, ctcd_result_alloc_fn alloc, void *alloc_user,
                               float *out_sc, int32_t *out_len, int32_t *n_results, int out_T, int *out_R, int *out_L, void *stream_) {
  if (!d || !states || !is_eos || !alloc || !out_sc || !out_len || !n_results || B < 0 || T < 0 || out_T < 0) return fail(CTCD_EINVAL, "bad arguments");
  if (out_R) *out_R = 0;
  if (out_L) *out_L = 0;
  if (B == 0) return CTCD_OK;
  CTC_ON_DEVICE(d->device);
  hipStream_t stream = (hipStream_t)stream_;
  if (states[0] && states[0]->scorer && states[0]->scorer->cbl) return fail(CTCD_EUNSUPPORTED, "a callback scorer's streams end through ctcd_stream_decode");
  if (out_T > 65536 || V > 65535) return fail(CTCD_EUNSUPPORTED, "compact results pack label and frame into 16 bits each");
  std::lock_guard<std::mutex> host_lock(d->mu_host);
  static const bool trace = getenv("CTCD_STREAM_TIMING") != nullptr;  // (stderr: where the call's time goes)
  const auto t0 = std::chrono::steady_clock::now();
  auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  double t_launch = 0, t_sync = 0, t_alloc = 0;
  std::vector<int32_t> lens(B);
  bool any_eos = false;
  int rc = stream_prepare(d, states, is_eos, seq_lens_host, B, T, V, beam, out_T, stream, lens, any_eos);
  if (rc) return rc;
  const size_t kk = (size_t)B * beam;
  const long long cap = std::max<long long>(ctcd_compact_label_capacity(B, beam, out_T), 1);
  if (cap > 0xFFFFFFFFLL) return fail(CTCD_EUNSUPPORTED, "batch too large for one compact label buffer; split the batch");
  if ((rc = d->c_hdr.ensure((size_t)B * 16)) || (rc = d->c_ent.ensure(kk * 16)) || (rc = d->c_rag.ensure((size_t)cap * 4)) ||
      (rc = d->c_cnt.ensure(256)) || (rc = d->c_sc.ensure(kk * 4)) || (rc = d->c_ln.ensure(kk * 4 + (size_t)B * 4)))
    return rc;
  int32_t *d_nres = (int32_t *)((char *)d->c_ln.p + kk * 4);
  // the page-locked mirror the kernel writes a finished stream's records to (as in ctcd_beam_decode_to_host)
  const size_t o_done = 256, o_hdr = (o_done + (size_t)B * 4 + 255) / 256 * 256, o_ent = o_hdr + (size_t)B * 16,
               o_lab = (o_ent + kk * 16 + 255) / 256 * 256;
  const size_t mcap = d->mirror_cap_override >= 0 ? (size_t)d->mirror_cap_override + 1 : std::max<size_t>((size_t)1 << 20, (size_t)cap / 3);
  const size_t need = o_lab + mcap * 4;
  if (d->h_stage_cap < need) {
    if (d->h_stage) (void)hipHostFree(d->h_stage);
    d->h_stage = nullptr;
    d->h_stage_cap = 0;
    HIP_TRY(hipHostMalloc(&d->h_stage, need, hipHostMallocMapped | hipHostMallocCoherent));
    d->h_stage_cap = need;
  }
  char *hs = (char *)d->h_stage, *ds = nullptr;
  HIP_TRY(hipHostGetDevicePointer((void **)&ds, d->h_stage, 0));
  volatile int32_t *done = (volatile int32_t *)(hs + o_done);
  for (int b = 0; b < B; ++b) done[b] = 0;
  std::memset(hs + o_hdr, 0, (size_t)B * 16);  // (streams that do not end write nothing)
  CompactOut co{(int32_t *)d->c_hdr.p, (int32_t *)d->c_ent.p, (uint32_t *)d->c_rag.p, (unsigned *)d->c_cnt.p, (unsigned)cap};
  co.m_hdr = (int32_t *)(ds + o_hdr); co.m_ent = (int32_t *)(ds + o_ent); co.m_done = (int32_t *)(ds + o_done);
  co.m_rag = (uint32_t *)(ds + o_lab); co.m_cap = (unsigned)std::min<size_t>(mcap, 0xFFFFFFFFu);
  StreamCall sc{states, is_eos, out_T, lens.data(), any_eos};
  rc = decode_common(d, probs, nullptr, B, T, V, beam, cutoff_prob, cutoff_top_n, blank_id, log_input, nullptr, nullptr, (float *)d->c_sc.p,
                     (int32_t *)d->c_ln.p, d_nres, stream, &sc, states[0]->scorer, &co);
  if (rc) return rc;
  for (int b = 0; b < B; ++b) states[b]->frames += lens[b];
  struct Drain {
    hipStream_t a;
    ~Drain() { (void)hipStreamSynchronize(a); }
  } drain{stream};
  if (any_eos) {
    HIP_TRY(hipMemcpyAsync(out_sc, d->c_sc.p, kk * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(out_len, d->c_ln.p, kk * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(n_results, d_nres, (size_t)B * 4, hipMemcpyDeviceToHost, stream));
  } else {
    std::memset(out_sc, 0, kk * 4); std::memset(out_len, 0, kk * 4); std::memset(n_results, 0, (size_t)B * 4);
  }
  t_launch = since();
  HIP_TRY(hipStreamSynchronize(stream));
  if ((rc = ctcd_check_status(d, B))) return rc;
  t_sync = since();
  if (!any_eos) return CTCD_OK;
  // the records: from the mirror, or -- an item whose labels fell beyond it -- everything from the device buffers
  const int32_t *hh = (const int32_t *)(hs + o_hdr), *he = (const int32_t *)(hs + o_ent);
  const uint32_t *hl = (const uint32_t *)(hs + o_lab);
  std::vector<int32_t> fh, fe;
  std::vector<uint32_t> fl;
  bool late = false;
  for (int b = 0; b < B; ++b) late |= is_eos[b] && done[b] != 1;
  if (late) {
    unsigned nlab = 0;
    HIP_TRY(hipMemcpy(&nlab, d->c_cnt.p, 4, hipMemcpyDeviceToHost));
    fh.resize((size_t)B * 4); fe.resize(kk * 4); fl.resize(nlab ? nlab : 1);
    HIP_TRY(hipMemcpy(fh.data(), d->c_hdr.p, (size_t)B * 16, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(fe.data(), d->c_ent.p, kk * 16, hipMemcpyDeviceToHost));
    if (nlab) HIP_TRY(hipMemcpy(fl.data(), d->c_rag.p, (size_t)nlab * 4, hipMemcpyDeviceToHost));
    hh = fh.data(); he = fe.data(); hl = fl.data();
  }
  int R = 0, L = 0;
  for (int b = 0; b < B; ++b) {
    const int nres = hh[(size_t)b * 4];
    R = std::max(R, nres);
    for (int j = 0; j < nres; ++j) L = std::max(L, he[((size_t)b * beam + j) * 4 + 2]);
  }
  if (out_R) *out_R = R;
  if (out_L) *out_L = L;
  int32_t *out_tok = nullptr, *out_ts = nullptr;
  if (alloc(alloc_user, R, L, &out_tok, &out_ts) != 0) return fail(CTCD_EINVAL, "the result allocator failed");
  t_alloc = since();
  if (R == 0 || L == 0) return CTCD_OK;
  if (!out_tok || !out_ts) return fail(CTCD_EINVAL, "the result allocator returned no buffers");
  if (!d->workers) d->workers = new HostPool;
  {
    const long long out_mb = (long long)B * R * L * 8 >> 20;
    const int want = std::max<long long>(num_processes, std::min<long long>(160, std::max<long long>(16, out_mb)));
    const int target = std::max(1, std::min(want, (int)std::thread::hardware_concurrency()));
    if (target > (int)d->workers->th.size()) d->workers->start(target - (int)d->workers->th.size());
  }
  d->workers->run(B, [=](int b) { ctcbeam::expand_item_host(hh, he, hl, b, beam, L, out_tok, out_ts, R); });
  if (trace) fprintf(stderr, "ctcd_stream_decode_to_host: queued %.3f ms, kernel done %.3f, buffers (R=%d, L=%d) %.3f, expanded by %d threads %.3f\n", t_launch, t_sync, R, L,
                     t_alloc, (int)d->workers->th.size(), since());
  return CTCD_OK;
}

int ctcd_beam_decode_host(ctcd_decoder *d, const float *probs, const int32_t *seq_lens, int B, int T, int V, int beam,
                          int num_processes, double cutoff_prob, int cutoff_top_n, int blank_id, int log_input,
                          int32_t *out_tok, int32_t *out_ts, float *out_sc, int32_t *out_len, int32_t *n_results) {
  return ctcd_beam_decode_lm_host(d, probs, seq_lens, B, T, V, beam, num_processes, cutoff_prob, cutoff_top_n, blank_id, log_input,
                                  nullptr, out_tok, out_ts, out_sc, out_len, n_results);
}

// ---- compact result delivery.  The K label sequences of an utterance overlap almost entirely (the beam
// is a trie): a decode that hands over, per beam entry, only the labels it does not share with its DFS predecessor moves
// ~40x fewer bytes than the padded [B, K, T] pair -- over PCIe for the reference's CPU-tensor contract, over xGMI for the
// multi-GPU gather.  Layout: beam_core.h OutRefs::c_*.
long long ctcd_compact_label_capacity(int B, int beam, int T) { return (long long)B * beam * T; }  // worst case: nothing shared

int ctcd_beam_decode_compact(ctcd_decoder *d, const float *probs, const int32_t *seq_lens, int B, int T, int V, int beam,
                             int /*num_processes*/, double cutoff_prob, int cutoff_top_n, int blank_id, int log_input,
                             ctcd_scorer *scorer, int32_t *c_hdr, int32_t *c_ent, uint32_t *c_labels, uint32_t *c_count,
                             long long label_capacity, float *out_sc, int32_t *out_len, int32_t *n_results, void *stream_) {
  if (!c_hdr || !c_ent || !c_labels || !c_count || label_capacity <= 0) return fail(CTCD_EINVAL, "compact buffers missing");
  CompactOut co{c_hdr, c_ent, c_labels, c_count, (unsigned)std::min<long long>(label_capacity, 0xFFFFFFFFLL)};
  // (round 5) a callback scorer: launch after launch like the padded form; an utterance hands its compact records over in the
  // launch it finishes in, the buffers and their bump allocator carry over from launch to launch
  if (scorer && scorer->cbl)
    return decode_lm_callback(d, probs, seq_lens, B, T, V, beam, cutoff_prob, cutoff_top_n, blank_id, log_input, scorer, nullptr, nullptr, out_sc, out_len,
                              n_results, stream_, &co);
  return decode_common(d, probs, seq_lens, B, T, V, beam, cutoff_prob, cutoff_top_n, blank_id, log_input, nullptr, nullptr, out_sc,
                       out_len, n_results, stream_, nullptr, scorer, &co);
}

int ctcd_expand_compact(ctcd_decoder *d, const int32_t *c_hdr, const int32_t *c_ent, const uint32_t *c_labels, int B, int beam, int T,
                        int32_t *out_tok, int32_t *out_ts, void *stream_) {
  if (!d || B < 0 || beam <= 0 || T < 0) return fail(CTCD_EINVAL, "bad arguments");
  if (B == 0 || T == 0) return CTCD_OK;
  if (!c_hdr || !c_ent || !c_labels || !out_tok || !out_ts) return fail(CTCD_EINVAL, "null tensor");
  CTC_ON_DEVICE(d->device);
  if (beam > kExpandMaxK) return fail(CTCD_EUNSUPPORTED, "device expansion of compact results: beam_width > 4096");
  const size_t esm = ((size_t)5 * beam + (beam + 31) / 32 + 4) * 4;
  if (esm + 1024 > (size_t)d->max_lds) return fail(CTCD_EUNSUPPORTED, "device expansion of compact results: the entry tables of this beam width exceed one workgroup's LDS");
  HIP_TRY(hipFuncSetAttribute((const void *)expand_compact_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)esm));  // (beam > ~3270 needs more than the default 64 KB)
  hipLaunchKernelGGL(expand_compact_kernel, dim3(B), dim3(beam >= 16 ? 1024 : 256), esm, (hipStream_t)stream_, c_hdr, c_ent, c_labels, beam, T,
                     out_tok, out_ts);
  HIP_TRY(hipGetLastError());
  return CTCD_OK;
}

// The reference's call as its Python makes it (ctcdecode/__init__.py:77-123 -> paddle_beam_decode[_lm]): the four results
// arrive as HOST tensors.  probs / seq_lens may live on either side (probs_on_device != 0: both are device pointers).
// The kernel hands its results over in compact form, they cross PCIe compact, and host threads expand them.
static int decode_to_host_locked(ctcd_decoder *d, const float *probs, const int32_t *seq_lens, int probs_on_device, int B, int T, int V,
                                 int beam, int num_processes, double cutoff_prob, int cutoff_top_n, int blank_id, int log_input,
                                 ctcd_scorer *scorer, int32_t *out_tok, int32_t *out_ts, float *out_sc, int32_t *out_len,
                                 int32_t *n_results, void *stream_);

int ctcd_beam_decode_to_host(ctcd_decoder *d, const float *probs, const int32_t *seq_lens, int probs_on_device, int B, int T, int V,
                             int beam, int num_processes, double cutoff_prob, int cutoff_top_n, int blank_id, int log_input,
                             ctcd_scorer *scorer, int32_t *out_tok, int32_t *out_ts, float *out_sc, int32_t *out_len,
                             int32_t *n_results, void *stream_) {
  if (!d) return fail(CTCD_EINVAL, "decoder == NULL");
  int rc = check_args(B, T, V, beam, cutoff_top_n, blank_id, probs, out_tok, out_ts, out_sc, out_len);
  if (rc) return rc;
  if (B == 0) return CTCD_OK;
  std::lock_guard<std::mutex> host_lock(d->mu_host);
  rc = decode_to_host_locked(d, probs, seq_lens, probs_on_device, B, T, V, beam, num_processes, cutoff_prob, cutoff_top_n, blank_id, log_input,
                             scorer, out_tok, out_ts, out_sc, out_len, n_results, stream_);
  if (rc == CTCD_EINTERNAL && d->input_timed_out) {
    // The streamed rows did not reach the kernel in time (its row fetch gives up after about a second): on this system the
    // copy stream evidently does not run beside the kernel.  Decode again the plain way, and stay with it.
    d->input_timed_out = false;
    d->no_input_streaming = true;
    if (d->copy_stream) (void)hipStreamSynchronize(d->copy_stream);
    rc = decode_to_host_locked(d, probs, seq_lens, probs_on_device, B, T, V, beam, num_processes, cutoff_prob, cutoff_top_n, blank_id, log_input,
                               scorer, out_tok, out_ts, out_sc, out_len, n_results, stream_);
  }
  return rc;
}

static int decode_to_host_locked(ctcd_decoder *d, const float *probs, const int32_t *seq_lens, int probs_on_device, int B, int T, int V,
                                 int beam, int num_processes, double cutoff_prob, int cutoff_top_n, int blank_id, int log_input,
                                 ctcd_scorer *scorer, int32_t *out_tok, int32_t *out_ts, float *out_sc, int32_t *out_len,
                                 int32_t *n_results, void *stream_) {
  int rc;
  CTC_ON_DEVICE(d->device);
  hipStream_t stream = (hipStream_t)stream_;
  const size_t nin = (size_t)B * T * V * 4, kk = (size_t)B * beam;
  const float *dprobs = probs;
  const int32_t *dlens = seq_lens;
  // Streamed input: with log-probabilities in host memory and no vocabulary pruning the kernel is launched BEFORE its
  // input has crossed PCIe; the rows follow frame block by frame block on a second stream (the input is [B, T, V]: a
  // block is a strided copy), each block followed by an update of the "frames arrived" counter the kernel's row fetch
  // waits on (beam_core.h decode_utterance).  Hides the 0.5 ms the 30 MB of a configs[1] batch take.
  const bool pruned_in = cutoff_prob < 1.0 || cutoff_top_n < V;
  // (V <= the workgroup size the launch will use -- decode_common: the caller's, or at least 512 -- so that the kernel
  //  prefetches its rows: ADVICE r3; rows that are not prefetched wait as well since round 4, this keeps the fast form)
  const bool stream_in = !probs_on_device && log_input == 1 && !pruned_in && T >= 128 && T <= 65536 && V <= (d->threads ? d->threads : 512) &&
                         nin >= ((size_t)1 << 20) && !d->no_input_streaming && !d->profile && !(scorer && scorer->cbl);
  int nblk = 0;
  int blk[10];
  const int *frames_ready = nullptr;
  if (stream_in) {
    const size_t off_rows = 256, off_sl = off_rows + (nin + 15) / 16 * 16, fg_need = off_sl + (size_t)B * 4 + 16;
    if (d->fg_in_cap < fg_need) {
      if (d->fg_in) (void)hipFree(d->fg_in);
      d->fg_in = nullptr; d->fg_in_cap = 0;
      HIP_TRY(hipExtMallocWithFlags(&d->fg_in, fg_need, hipDeviceMallocFinegrained));
      d->fg_in_cap = fg_need;
    }
    if (!d->copy_stream) {
      HIP_TRY(hipStreamCreateWithFlags(&d->copy_stream, hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&d->ev_in, hipEventDisableTiming));
      HIP_TRY(hipHostMalloc((void **)&d->h_cnt, 256, hipHostMallocDefault));
    }
    char *fg = (char *)d->fg_in;
    HIP_TRY(hipMemsetAsync(fg, 0, 4, stream));  // no frame has arrived
    if (seq_lens) HIP_TRY(hipMemcpyAsync(fg + off_sl, seq_lens, (size_t)B * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(d->ev_in, stream));
    HIP_TRY(hipStreamWaitEvent(d->copy_stream, d->ev_in, 0));  // (the blocks must not overtake the reset)
    dprobs = (const float *)(fg + off_rows);
    dlens = seq_lens ? (const int32_t *)(fg + off_sl) : nullptr;
    frames_ready = (const int *)fg;
    // a small first block, so that the kernel starts at once; the rest in seven equal parts
    blk[0] = 0; blk[1] = 16; nblk = 1;
    for (int c = 1; c <= 7; ++c) blk[++nblk] = 16 + (int)((long long)(T - 16) * c / 7);
  } else if (!probs_on_device) {
    const size_t off_sl = (nin + 15) / 16 * 16;
    if ((rc = d->stage_in.ensure(off_sl + (size_t)B * 4 + 16))) return rc;
    char *din = (char *)d->stage_in.p;
    if (nin) HIP_TRY(hipMemcpyAsync(din, probs, nin, hipMemcpyHostToDevice, stream));
    if (seq_lens) HIP_TRY(hipMemcpyAsync(din + off_sl, seq_lens, (size_t)B * 4, hipMemcpyHostToDevice, stream));
    dprobs = (const float *)din;
    dlens = seq_lens ? (const int32_t *)(din + off_sl) : nullptr;
  }
  const bool hook = scorer && scorer->cbl;  // a callback scorer: launch after launch (cb_rounds) into the padded tensors
  if (T > 65536 || V > 65535 || T == 0 || hook) {  // outside the compact format's 16-bit fields: the padded tensors travel
    const size_t kt = kk * T * 4;
    const size_t o_ts = (kt + 15) / 16 * 16, o_sc = o_ts * 2, o_ln = o_sc + (kk * 4 + 15) / 16 * 16, o_nr = o_ln + (kk * 4 + 15) / 16 * 16;
    if ((rc = d->stage_out.ensure(o_nr + (size_t)B * 4 + 16))) return rc;
    char *dout = (char *)d->stage_out.p;
    if (hook)
      rc = decode_lm_callback(d, dprobs, dlens, B, T, V, beam, cutoff_prob, cutoff_top_n, blank_id, log_input, scorer, (int32_t *)dout,
                              (int32_t *)(dout + o_ts), (float *)(dout + o_sc), (int32_t *)(dout + o_ln), (int32_t *)(dout + o_nr), stream);
    else
      rc = decode_common(d, dprobs, dlens, B, T, V, beam, cutoff_prob, cutoff_top_n, blank_id, log_input, (int32_t *)dout, (int32_t *)(dout + o_ts),
                         (float *)(dout + o_sc), (int32_t *)(dout + o_ln), (int32_t *)(dout + o_nr), stream, nullptr, scorer);
    if (rc) return rc;
    if ((rc = ctcd_check_status(d, B))) return rc;
    if (kt) {
      HIP_TRY(hipMemcpy(out_tok, dout, kt, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(out_ts, dout + o_ts, kt, hipMemcpyDeviceToHost));
    }
    HIP_TRY(hipMemcpy(out_sc, dout + o_sc, kk * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_len, dout + o_ln, kk * 4, hipMemcpyDeviceToHost));
    if (n_results) HIP_TRY(hipMemcpy(n_results, dout + o_nr, (size_t)B * 4, hipMemcpyDeviceToHost));
    return CTCD_OK;
  }
  const long long cap = ctcd_compact_label_capacity(B, beam, T);
  if (cap > 0xFFFFFFFFLL) return fail(CTCD_EUNSUPPORTED, "batch too large for one compact label buffer; split the batch");
  if ((rc = d->c_hdr.ensure((size_t)B * 16)) || (rc = d->c_ent.ensure(kk * 16)) || (rc = d->c_rag.ensure((size_t)cap * 4)) ||
      (rc = d->c_cnt.ensure(256)) || (rc = d->c_sc.ensure(kk * 4)) || (rc = d->c_ln.ensure(kk * 4 + (size_t)B * 4)))
    return rc;
  int32_t *d_nres = (int32_t *)((char *)d->c_ln.p + kk * 4);
  // Page-locked, device-visible staging the KERNEL writes to: [done flags | hdr | ent | labels].  A workgroup that has
  // finished its utterance copies the compact results there itself and raises the utterance's flag (beam_core.h
  // OutRefs::m_*): the results cross PCIe while the stragglers of the launch are still being decoded, and host threads
  // expand utterance after utterance as the flags come up -- when the kernel ends only its last utterances are left.
  const size_t o_done = 256, o_hdr = (o_done + (size_t)B * 4 + 255) / 256 * 256, o_ent = o_hdr + (size_t)B * 16,
               o_lab = (o_ent + kk * 16 + 255) / 256 * 256;
  // the mirror holds a third of the worst case (typical sharing: a tenth); an utterance whose labels fall beyond it is
  // fetched from the device buffer afterwards
  const size_t mcap = d->mirror_cap_override >= 0 ? (size_t)d->mirror_cap_override + 1 : std::max<size_t>((size_t)1 << 20, (size_t)cap / 3);
  const size_t need = o_lab + mcap * 4;
  if (d->h_stage_cap < need) {
    if (d->h_stage) (void)hipHostFree(d->h_stage);
    d->h_stage = nullptr;
    d->h_stage_cap = 0;
    HIP_TRY(hipHostMalloc(&d->h_stage, need, hipHostMallocMapped | hipHostMallocCoherent));
    d->h_stage_cap = need;
  }
  char *hs = (char *)d->h_stage, *ds = nullptr;
  HIP_TRY(hipHostGetDevicePointer((void **)&ds, d->h_stage, 0));
  volatile int32_t *done = (volatile int32_t *)(hs + o_done);
  for (int b = 0; b < B; ++b) done[b] = 0;
  CompactOut co{(int32_t *)d->c_hdr.p, (int32_t *)d->c_ent.p, (uint32_t *)d->c_rag.p, (unsigned *)d->c_cnt.p, (unsigned)cap};
  co.m_hdr = (int32_t *)(ds + o_hdr); co.m_ent = (int32_t *)(ds + o_ent); co.m_done = (int32_t *)(ds + o_done);
  co.m_rag = (uint32_t *)(ds + o_lab); co.m_cap = (unsigned)std::min<size_t>(mcap, 0xFFFFFFFFu);
  rc = decode_common(d, dprobs, dlens, B, T, V, beam, cutoff_prob, cutoff_top_n, blank_id, log_input, nullptr, nullptr, (float *)d->c_sc.p,
                     (int32_t *)d->c_ln.p, d_nres, stream, nullptr, scorer, &co, frames_ready);
  if (rc) return rc;
  // From here on work is queued that reads the caller's input and writes the caller's outputs: no way out of this
  // function -- error or not -- without both streams drained.
  struct Drain {
    hipStream_t a, b;
    ~Drain() { if (b) (void)hipStreamSynchronize(b); (void)hipStreamSynchronize(a); }
  } drain{stream, stream_in ? d->copy_stream : nullptr};
  if (stream_in) {  // the kernel is queued and waits for its rows: send them, frame block by frame block
    char *fg = (char *)d->fg_in;
    const size_t pitch = (size_t)T * V * 4;
    for (int c = 0; c < nblk; ++c) {
      const int f0 = blk[c], f1 = blk[c + 1];
      if (f1 <= f0) continue;
      HIP_TRY(hipMemcpy2DAsync(fg + 256 + (size_t)f0 * V * 4, pitch, (const char *)probs + (size_t)f0 * V * 4, pitch, (size_t)(f1 - f0) * V * 4, (size_t)B,
                               hipMemcpyHostToDevice, d->copy_stream));
      d->h_cnt[c] = f1;
      if (!d->stall_input_test) HIP_TRY(hipMemcpyAsync(fg, &d->h_cnt[c], 4, hipMemcpyHostToDevice, d->copy_stream));
    }
  }
  HIP_TRY(hipMemcpyAsync(out_sc, d->c_sc.p, kk * 4, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipMemcpyAsync(out_len, d->c_ln.p, kk * 4, hipMemcpyDeviceToHost, stream));
  if (n_results) HIP_TRY(hipMemcpyAsync(n_results, d_nres, (size_t)B * 4, hipMemcpyDeviceToHost, stream));
  {
    if (!d->workers) d->workers = new HostPool;
    // host threads that expand the results (memory-bound work: 8 bytes written per label position of the padded tensors;
    // the reference's default num_processes = 4 was chosen for its CPU decode): one per 4 MB of output, between 16 and 64,
    // more if the caller asks, never more than the machine has.  The pool grows when a later call wants more.
    const long long out_mb = (long long)B * beam * T * 8 >> 20;
    const int want = std::max<long long>(num_processes, std::min<long long>(64, std::max<long long>(16, out_mb / 4)));
    const int target = std::max(1, std::min(want, (int)std::thread::hardware_concurrency()));
    if (target > (int)d->workers->th.size()) d->workers->start(target - (int)d->workers->th.size());
  }
  const int32_t *hh = (const int32_t *)(hs + o_hdr), *he = (const int32_t *)(hs + o_ent);
  const uint32_t *hl = (const uint32_t *)(hs + o_lab);
  std::atomic<int> finished{0}, sync_rc{0};
  std::mutex late_mu;
  std::vector<int> late;  // utterances the kernel did not mirror (labels beyond the mirror) -- or that never reported
  const int dev = d->device;
  // job 0 waits for the stream (kernel + the small copies above) and tells the others; jobs 1 .. B expand utterance i - 1
  // as soon as its flag is up
  d->workers->run(B + 1, [&, hh, he, hl, done, dev, stream](int i) {
    if (i == 0) {
      (void)hipSetDevice(dev);
      if (hipStreamSynchronize(stream) != hipSuccess) sync_rc = 1;
      finished = 1;
      return;
    }
    const int b = i - 1;
    int v;
    for (int spin = 0; (v = done[b]) == 0 && !finished.load(std::memory_order_acquire); ++spin)
      if (spin > 64) std::this_thread::sleep_for(std::chrono::microseconds(20));
    if (v == 0) v = done[b];  // (the kernel has ended: whatever it wrote is visible)
    std::atomic_thread_fence(std::memory_order_acquire);
    if (v == 1) {
      ctcbeam::expand_item_host(hh, he, hl, b, beam, T, out_tok, out_ts);
    } else {
      std::lock_guard<std::mutex> g(late_mu);
      late.push_back(b);
    }
  });
  if (sync_rc) return fail(CTCD_EHIP, "hipStreamSynchronize failed");
  if (stream_in) HIP_TRY(hipStreamSynchronize(d->copy_stream));  // (long done: the kernel has consumed every row)
  if ((rc = ctcd_check_status(d, B))) return rc;
  if (!late.empty()) {  // rare: fetch the whole compact form from the device and expand the utterances left
    std::vector<int32_t> fh((size_t)B * 4), fe(kk * 4);
    unsigned nlab = 0;
    HIP_TRY(hipMemcpy(&nlab, d->c_cnt.p, 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> fl(nlab ? nlab : 1);
    HIP_TRY(hipMemcpy(fh.data(), d->c_hdr.p, (size_t)B * 16, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(fe.data(), d->c_ent.p, kk * 16, hipMemcpyDeviceToHost));
    if (nlab) HIP_TRY(hipMemcpy(fl.data(), d->c_rag.p, (size_t)nlab * 4, hipMemcpyDeviceToHost));
    const int *lt = late.data();
    const int32_t *ph = fh.data(), *pe = fe.data();
    const uint32_t *pl = fl.data();
    d->workers->run((int)late.size(), [=](int i) { ctcbeam::expand_item_host(ph, pe, pl, lt[i], beam, T, out_tok, out_ts); });
  }
  return CTCD_OK;
}

// paddle_beam_decode_lm as the reference's Python calls it (binding.cpp:122-140): CPU tensors in, CPU tensors out.
// scorer == NULL decodes without a language model.
int ctcd_beam_decode_lm_host(ctcd_decoder *d, const float *probs, const int32_t *seq_lens, int B, int T, int V, int beam,
                             int num_processes, double cutoff_prob, int cutoff_top_n, int blank_id, int log_input,
                             ctcd_scorer *scorer, int32_t *out_tok, int32_t *out_ts, float *out_sc, int32_t *out_len,
                             int32_t *n_results) {
  return ctcd_beam_decode_to_host(d, probs, seq_lens, 0, B, T, V, beam, num_processes, cutoff_prob, cutoff_top_n, blank_id, log_input,
                                  scorer, out_tok, out_ts, out_sc, out_len, n_results, nullptr);
}

// The normalisation that log_input == 2 applies, on its own (device pointers; `out` may alias `logits`): float32
// log_softmax over the last axis with the summation order documented at log_softmax_rows_kernel.
int ctcd_log_softmax(ctcd_decoder *d, const float *logits, const int32_t *seq_lens, int B, int T, int V, float *out, void *stream_) {
  if (!d || !logits || !out || B < 0 || T < 0 || V < 1) return fail(CTCD_EINVAL, "bad arguments");
  if (B == 0 || T == 0) return CTCD_OK;
  CTC_ON_DEVICE(d->device);
  hipStream_t stream = (hipStream_t)stream_;
  int rc;
  {
    std::lock_guard<std::mutex> lock(d->mu);
    if (!d->tables_ready) {
      if ((rc = d->tables.ensure(sizeof(ctcmath::Tables)))) return rc;
      HIP_TRY(hipMemcpy(d->tables.p, ctcmath::host_tables().w, sizeof(ctcmath::Tables), hipMemcpyHostToDevice));
      d->tables_ready = true;
    }
  }
  return launch_log_softmax(d, logits, out, (long long)B * T, seq_lens, T, V, stream);
}

// HIP-event timing of the decode kernel alone, on the stream it is launched on (bench.py's roofline figure).
int ctcd_set_timing(ctcd_decoder *d, int on) {
  if (!d) return fail(CTCD_EINVAL, "decoder == NULL");
  CTC_ON_DEVICE(d->device);
  if (on && !d->ev0) {
    HIP_TRY(hipEventCreate(&d->ev0));
    HIP_TRY(hipEventCreate(&d->ev1));
    HIP_TRY(hipEventCreate(&d->ev2));
    HIP_TRY(hipEventCreate(&d->ev3));
  }
  d->timing = on != 0;
  return CTCD_OK;
}

int ctcd_last_kernel_ms(ctcd_decoder *d, float *ms) {
  if (!d || !ms || !d->ev0) return fail(CTCD_EINVAL, "timing not enabled");
  HIP_TRY(hipEventSynchronize(d->ev1));
  HIP_TRY(hipEventElapsedTime(ms, d->ev0, d->ev1));
  return CTCD_OK;
}

// Duration of the vocabulary-prune kernel of the last decode (pruned configurations only; timing must be on).
int ctcd_last_prune_ms(ctcd_decoder *d, float *ms) {
  if (!d || !ms || !d->ev2 || !d->prune_timed) return fail(CTCD_EINVAL, "no timed prune pass");
  CTC_ON_DEVICE(d->device);
  HIP_TRY(hipEventSynchronize(d->ev3));
  HIP_TRY(hipEventElapsedTime(ms, d->ev2, d->ev3));
  return CTCD_OK;
}

// Phase profile (tools/phase_profile.py): run the instrumented build of the kernel and read its per-utterance timers
// (wall_clock64 ticks of a 100 MHz clock, accumulated per phase by thread 0 of each workgroup).
int ctcd_debug_set_profile(ctcd_decoder *d, int on) {
  if (!d) return fail(CTCD_EINVAL, "decoder == NULL");
  d->profile = on != 0;
  return CTCD_OK;
}

// Workspace layout: 1 (default) = small shapes use the kernel variant whose LDS arrays sit at compile-time addresses,
// 0 = always the run-time layout.  Results are identical; the switch exists so tests can run both variants.
int ctcd_debug_set_fixed_layout(ctcd_decoder *d, int on) {
  if (!d) return fail(CTCD_EINVAL, "decoder == NULL");
  d->no_fixed_layout = on == 0;
  return CTCD_OK;
}

// Flagged prune frames: 1 (default) = the std::sort replay keeps its arrays in LDS when the row fits, 0 = always in global
// memory (the path of rows beyond ~11 000 labels).  Results are identical; the switch exists for the tests.  (Until round 3
// "0" sent the flagged frames to the host toolchain; nothing goes there any more.)
int ctcd_debug_set_prune_resolve(ctcd_decoder *d, int on) {
  if (!d) return fail(CTCD_EINVAL, "decoder == NULL");
  d->resolve_in_global = on == 0;
  return CTCD_OK;
}

// Raw-logit input (log_input == 2): 1 (default) = long rows take the workgroup kernels (log_softmax_rows_wg_kernel; in front of a
// vocabulary prune the fused prune_logits_wg_kernel), 0 = always the one-wave log_softmax pass followed by the separate prune.
// Results are identical; the switch exists so tests can compare the two.
int ctcd_debug_set_fused_logits(ctcd_decoder *d, int on) {
  if (!d) return fail(CTCD_EINVAL, "decoder == NULL");
  d->no_fused_logits = on == 0;
  return CTCD_OK;
}

// The workgroup prune kernel: 1 (default) = the row stays in registers between the two looks at it, 0 = it is read twice (the form of
// rounds 2-5; vocabularies beyond 10 240 labels always).  Identical results; tests compare the two.
int ctcd_debug_set_prune_registers(ctcd_decoder *d, int on) {
  if (!d) return fail(CTCD_EINVAL, "decoder == NULL");
  d->no_prune_reg = on == 0;
  return CTCD_OK;
}

// Barrier timeline of batch item 0 (profiling build): call with out == NULL to arm frames [frame0, frame0 + nframes) of
// the following decodes, with out != NULL (int64 [16][ctcd_debug_timeline_cap()]) to fetch: per wave, the shader clock
// at arrival at / departure from each barrier, in program order.
int ctcd_debug_timeline_cap(void) { return kTimelineCap; }
int ctcd_debug_timeline(ctcd_decoder *d, int frame0, int nframes, long long *out) {
  if (!d) return fail(CTCD_EINVAL, "decoder == NULL");
  if (!out) { d->tl_f0 = frame0; d->tl_nf = nframes; d->tl_armed = nframes > 0; return CTCD_OK; }
  if (!d->tl.p) return fail(CTCD_EINVAL, "no timeline recorded");
  CTC_ON_DEVICE(d->device);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, d->tl.p, (size_t)16 * kTimelineCap * 8, hipMemcpyDeviceToHost));
  return CTCD_OK;
}

// beam of batch item 0 after every frame (profiling build): out = int32 [T][1 + 4*beam]
int ctcd_debug_beam_dump(ctcd_decoder *d, int on, int *out, int T, int beam) {
  if (!d) return fail(CTCD_EINVAL, "decoder == NULL");
  d->dbg_on = on != 0;
  if (out && d->dbg.p) {
    CTC_ON_DEVICE(d->device);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, d->dbg.p, (size_t)T * (1 + 4 * (size_t)beam) * 4, hipMemcpyDeviceToHost));
  }
  return CTCD_OK;
}

int ctcd_debug_get_profile(ctcd_decoder *d, long long *out, int B) {
  if (!d || !out || B <= 0 || !d->prof.p) return fail(CTCD_EINVAL, "no profile recorded");
  CTC_ON_DEVICE(d->device);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, d->prof.p, (size_t)B * 16 * 8, hipMemcpyDeviceToHost));
  return CTCD_OK;
}

// Device expf/logf/log_sum_exp against the host C library over a range of float bit patterns (tests only):
// mode 0: expf_nonpos(x), 1: logf_normal(x), x = bits lo, lo+stride, ... <= hi;  mode 2: lse(x, y) on n pairs.
int ctcd_debug_math_check(ctcd_decoder *d, int mode, uint32_t lo, uint32_t hi, uint32_t stride, const float *xs,
                          const float *ys, long long n_pairs, long long *checked, long long *mismatches);

// Number of frames of the last ctcd_beam_decode whose vocabulary prune was resolved on the host: none, ever (kept for callers
// of earlier rounds; the host path is gone).
long long ctcd_last_prune_host_rows(ctcd_decoder *d) { return d ? 0 : -1; }
int ctcd_last_scorer_rounds(ctcd_decoder *d) { return d ? d->last_cb_rounds : -1; }
int ctcd_last_scorer_waits(ctcd_decoder *d) { return d ? d->last_cb_waits : -1; }
int ctcd_set_scorer_wait(ctcd_decoder *d, int on) {
  if (!d) return fail(CTCD_EINVAL, "decoder == NULL");
  d->no_hook_wait = on == 0;
  return CTCD_OK;
}
// ... and the number the fast prune pass flagged (settled by the device's std::sort replay + exact cumulative chain).  The
// count arrives behind the call's kernels: asking for it waits for the launch stream.
long long ctcd_last_prune_flagged_rows(ctcd_decoder *d) {
  if (!d) return -1;
  if (d->flagged_pending) {
    DeviceGuard g(d->device);
    if (hipStreamSynchronize(d->last_stream) != hipSuccess) return -1;
    d->prune_flagged_rows = d->h_flagged ? (long long)*d->h_flagged : 0;
    d->flagged_pending = false;
  }
  return d->prune_flagged_rows;
}

// The vocabulary-prune pass's output of the last call, copied to host memory (test hook: lets a test compare prune variants
// directly): cnt[rows], labels / values [rows][stride] with stride = min(cutoff_top_n, V); entries at or beyond a frame's count
// are unspecified.  Waits for the launch stream.
int ctcd_debug_prune_rows(ctcd_decoder *d, long long rows, int stride, int32_t *cnt, int32_t *labels, float *values) {
  if (!d || rows < 0 || stride < 1 || !cnt || !labels || !values) return fail(CTCD_EINVAL, "bad arguments");
  std::lock_guard<std::mutex> lock(d->mu);
  CTC_ON_DEVICE(d->device);
  if (d->pr_cnt.cap < (size_t)rows * 4 || d->pr_ch.cap < (size_t)rows * stride * 4 || d->pr_lp.cap < (size_t)rows * stride * 4)
    return fail(CTCD_EINVAL, "the last call's prune pass did not produce that many frames");
  HIP_TRY(hipStreamSynchronize(d->last_stream));
  HIP_TRY(hipMemcpy(cnt, d->pr_cnt.p, (size_t)rows * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(labels, d->pr_ch.p, (size_t)rows * stride * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(values, d->pr_lp.p, (size_t)rows * stride * 4, hipMemcpyDeviceToHost));
  return CTCD_OK;
}

// The same words, fetched without blocking: the copy into `host_status` (B words, page-locked memory for a truly
// asynchronous copy) is enqueued on `stream` -- pass the stream the decode was launched on, before enqueuing anything else
// there -- and the caller waits for its own event; 0 = ST_OK for an item, anything else is a failure code.  A pipelined
// caller (the next batch's kernel already queued behind) inspects a batch's outcome this way without draining the stream.
int ctcd_fetch_status_async(ctcd_decoder *d, int B, int32_t *host_status, void *stream_) {
  if (!d || B < 0 || (B > 0 && !host_status)) return fail(CTCD_EINVAL, "bad arguments");
  if (B == 0) return CTCD_OK;
  CTC_ON_DEVICE(d->device);
  if (!d->status.p || d->status.cap < (size_t)B * 4) return fail(CTCD_EINVAL, "no decode of that many items has been launched");
  HIP_TRY(hipMemcpyAsync(host_status, d->status.p, (size_t)B * 4, hipMemcpyDeviceToHost, (hipStream_t)stream_));
  return CTCD_OK;
}

// Status words of the last ctcd_beam_decode on this decoder (device -> host); for callers of the async entry point.
int ctcd_check_status(ctcd_decoder *d, int B) {
  if (!d || B < 0) return fail(CTCD_EINVAL, "bad arguments");
  if (B == 0) return CTCD_OK;
  CTC_ON_DEVICE(d->device);
  if (!d->status.p || d->status.cap < (size_t)B * 4) return fail(CTCD_EINVAL, "no decode of that many items has been launched");
  const bool with_shape = d->status.cap >= (size_t)B * 8 && d->status_items == B;
  std::vector<int32_t> st((size_t)B * (with_shape ? 2 : 1));
  // ordered after the kernel on the stream it was launched on (a null-stream copy would not wait for a non-blocking stream)
  HIP_TRY(hipMemcpyAsync(st.data(), d->status.p, st.size() * 4, hipMemcpyDeviceToHost, d->last_stream));
  HIP_TRY(hipStreamSynchronize(d->last_stream));
  if (with_shape) {
    // The beam shape the launch ended with: entries of an item's final beam that have descendants in it (x 16), averaged over the
    // items.  Random rows: 1-2; blank-dominated rows: ~20.  The next launch's phase A1 is chosen by it (hysteresis 4 .. 8).
    long long sum = 0, cnt = 0;
    for (int b = 0; b < B; ++b)
      if (st[B + b] > 0 || st[b] == ST_OK) { sum += st[B + b]; ++cnt; }
    if (cnt) {
      const double per_frame = (double)sum / (double)cnt / 16.0;
      if (per_frame >= 8.0) d->subtree_on = true;
      else if (per_frame <= 4.0) d->subtree_on = false;
    }
  }
  for (int b = 0; b < B; ++b)
    if (st[b] != ST_OK) {
      if (st[b] == ST_INPUT_TIMEOUT) d->input_timed_out = true;
      return fail(CTCD_EINTERNAL, "decoder status " + std::to_string(st[b]) + " for item " + std::to_string(b));
    }
  return CTCD_OK;
}

int ctcd_debug_math_check(ctcd_decoder *d, int mode, uint32_t lo, uint32_t hi, uint32_t stride, const float *xs,
                          const float *ys, long long n_pairs, long long *checked, long long *mismatches) {
  if (!d || !checked || !mismatches || stride == 0) return fail(CTCD_EINVAL, "bad arguments");
  CTC_ON_DEVICE(d->device);
  int rc;
  if (!d->tables_ready) {
    if ((rc = d->tables.ensure(sizeof(ctcmath::Tables)))) return rc;
    HIP_TRY(hipMemcpy(d->tables.p, ctcmath::host_tables().w, sizeof(ctcmath::Tables), hipMemcpyHostToDevice));
    d->tables_ready = true;
  }
  *checked = 0;
  *mismatches = 0;
  const size_t chunk = (size_t)1 << 24;
  Buf out, inx, iny;
  if (mode >= 3 && mode <= 6) {  // binary64 log (3: log(p), 4: log(p + FLT_MIN)) / exp (5) on float images, log_sum_exp<double> on pairs (6)
    if ((rc = out.ensure(chunk * 8))) return rc;
    std::vector<uint64_t> h64(chunk);
    auto same = [](double want, uint64_t got) { return (want != want && ctcmath::bits_to_f64(got) != ctcmath::bits_to_f64(got)) || ctcmath::f64_to_bits(want) == got; };
    auto ref_lse64 = [](double x, double y) {  // decoder_utils.h:47-54, T = double
      const double neg = -DBL_MAX;
      if (x <= neg) return y;
      if (y <= neg) return x;
      const double m = x > y ? x : y;
      return std::log(std::exp(x - m) + std::exp(y - m)) + m;
    };
    if (mode == 6) {
      if (!xs || !ys || n_pairs < 0) return fail(CTCD_EINVAL, "pairs missing");
      if ((rc = inx.ensure(chunk * 4)) || (rc = iny.ensure(chunk * 4))) return rc;
      for (long long off = 0; off < n_pairs; off += (long long)chunk) {
        const size_t cnt = (size_t)std::min<long long>((long long)chunk, n_pairs - off);
        HIP_TRY(hipMemcpy(inx.p, xs + off, cnt * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(iny.p, ys + off, cnt * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(debug_math64_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, 0, 6, 0u, 1u, (const float *)inx.p, (const float *)iny.p, (uint64_t *)out.p, cnt);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(h64.data(), out.p, cnt * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < cnt; ++i)
          if (!same(ref_lse64((double)xs[off + i], (double)ys[off + i]), h64[i])) ++*mismatches;
        *checked += (long long)cnt;
      }
      inx.release();
      iny.release();
    } else {
      for (uint64_t start = lo; start <= hi; start += (uint64_t)chunk * stride) {
        const uint64_t cnt64 = ((uint64_t)hi - start) / stride + 1;
        const size_t cnt = (size_t)std::min<uint64_t>(cnt64, chunk);
        hipLaunchKernelGGL(debug_math64_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, 0, mode, (uint32_t)start, stride, (const float *)nullptr,
                           (const float *)nullptr, (uint64_t *)out.p, cnt);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(h64.data(), out.p, cnt * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < cnt; ++i) {
          const double x = (double)ctcmath::bits_to_f32((uint32_t)(start + i * stride));
          const double want = mode == 3 ? std::log(x) : mode == 4 ? std::log(x + (double)FLT_MIN) : std::exp(x);
          if (!same(want, h64[i])) ++*mismatches;
        }
        *checked += (long long)cnt;
      }
    }
    out.release();
    return CTCD_OK;
  }
  if ((rc = out.ensure(chunk * 4))) return rc;
  std::vector<float> host(chunk);
  auto ref_lse = [](float x, float y) {  // decoder_utils.h:47-54
    const float neg = -FLT_MAX;
    if (x <= neg) return y;
    if (y <= neg) return x;
    const float m = x > y ? x : y;
    return std::log(std::exp(x - m) + std::exp(y - m)) + m;
  };
  if (mode == 0 || mode == 1) {
    for (uint64_t start = lo; start <= hi; start += (uint64_t)chunk * stride) {
      const uint64_t cnt64 = ((uint64_t)hi - start) / stride + 1;
      const size_t cnt = (size_t)std::min<uint64_t>(cnt64, chunk);
      hipLaunchKernelGGL(debug_math_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, 0, mode, (uint32_t)start, stride,
                         (const float *)nullptr, (const float *)nullptr, (float *)out.p, cnt, (const uint64_t *)d->tables.p);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpy(host.data(), out.p, cnt * 4, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < cnt; ++i) {
        const float x = ctcmath::bits_to_f32((uint32_t)(start + i * stride));
        const float want = mode == 0 ? std::exp(x) : std::log(x);
        const bool same = ctcmath::f32_to_bits(want) == ctcmath::f32_to_bits(host[i]);
        // below -88 the restatement returns 0 where libm returns < 2^-126: indistinguishable inside log_sum_exp
        if (!same && !(mode == 0 && x < -88.0f && host[i] == 0.0f && 1.0f + want == 1.0f)) ++*mismatches;
      }
      *checked += (long long)cnt;
    }
  } else {
    if (!xs || !ys || n_pairs < 0) return fail(CTCD_EINVAL, "pairs missing");
    if ((rc = inx.ensure(chunk * 4)) || (rc = iny.ensure(chunk * 4))) return rc;
    for (long long off = 0; off < n_pairs; off += (long long)chunk) {
      const size_t cnt = (size_t)std::min<long long>((long long)chunk, n_pairs - off);
      HIP_TRY(hipMemcpy(inx.p, xs + off, cnt * 4, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(iny.p, ys + off, cnt * 4, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(debug_math_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, 0, 2, 0u, 1u, (const float *)inx.p,
                         (const float *)iny.p, (float *)out.p, cnt, (const uint64_t *)d->tables.p);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpy(host.data(), out.p, cnt * 4, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < cnt; ++i)
        if (ctcmath::f32_to_bits(ref_lse(xs[off + i], ys[off + i])) != ctcmath::f32_to_bits(host[i])) ++*mismatches;
      *checked += (long long)cnt;
    }
    inx.release();
    iny.release();
  }
  out.release();
  return CTCD_OK;
}

}  // extern "C"
