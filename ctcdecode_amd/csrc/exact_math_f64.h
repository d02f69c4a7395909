// exact_math_f64.h -- binary64 log / exp that are BIT-IDENTICAL to the reference's.
//
// The reference's vocabulary pruning and its probability -> log conversion go through the host C library's DOUBLE log / exp
// (ctcdecode/src/decoder_utils.cpp:16 log(cutoff_prob), :29 log(p) inside log_sum_exp<double> (decoder_utils.h:47-54:
// log(exp(x - m) + exp(y - m)) + m), :42 log(p + FLT_MIN)); glibc 2.35 in this image, IFUNC-resolved to the FMA variants on
// any AVX2+FMA CPU.  Those routines are not correctly rounded (0.52 ULP), so a result that is then rounded to float32, or
// compared with cutoff_prob, can only be reproduced by reproducing them bit for bit.  Rounds 1-3 decided every frame whose
// outcome could not depend on the last bit on the GPU and sent the rest through a host copy of the reference's function
// (a transcribed CPU fallback inside the product path).  This header is what replaces it: the two routines
// restated operation by operation -- every fused multiply-add where the compiled x86-64 code has one, read from the
// disassembly of this image's libm.so.6 (recipe: tools/extract_libm_tables_f64.py; data: exact_math_f64_tables.h) -- in IEEE
// binary64 add / mul / fma, which gfx950 implements exactly (denormals on: the device's default for f64).
//
// Scope: every argument the reference can produce.  log: positive normal and subnormal doubles, +0 / -0 (-inf), negative
// (NaN), +inf, NaN.  exp: the whole real line incl. the results that underflow gradually (the reference evaluates
// exp(x - max) for x down to -FLT_MAX), +-inf, NaN; only errno / the exception flags are not modelled.
// tests/native/exact_math_f64_check.cpp compares the host build with the live libm; tests/test_gpu_decode.py
// ::test_device_math_f64_bit_exact_vs_host_libm does the same for the device build against the GPU box's own libm.
//
// Build note: compile with -ffp-contract=off.  Every fused operation below is an explicit __builtin_fma.
#pragma once
#include <stdint.h>

#include "exact_math.h"
#include "exact_math_f64_tables.h"

namespace ctcmath {

struct Tables64 {
  uint64_t log_head[18];   // ln2hi, ln2lo, A0..A4, B0..B10
  uint64_t log_tab[256];   // {invc, logc} x 128
  uint64_t exp_head[8];    // invln2N, shift, negln2hiN, negln2loN, C2..C5
  uint64_t exp_tab[256];   // {tail, sbits} x 128
};
#define CTC_TABLES64_INIT {CTC_LOG64_HEAD, CTC_LOG64_TAB, CTC_EXP64_HEAD, CTC_EXP64_TAB}

// The tables where the code runs: a copy per translation unit (4.3 KB; dropped from the ones that never use it).
#if defined(__HIPCC__)
static __device__ const Tables64 kTables64Device = CTC_TABLES64_INIT;
#endif
static const Tables64 kTables64Host = CTC_TABLES64_INIT;
CTC_HD const Tables64 &tables64() {
#if defined(__HIP_DEVICE_COMPILE__)
  return kTables64Device;
#else
  return kTables64Host;
#endif
}

// == glibc 2.35 __log_fma(x)
CTC_HD double log_f64(double x, const Tables64 &T) {
  uint64_t ix = f64_to_bits(x);
  const uint32_t top = (uint32_t)(ix >> 48);
  auto H = [&](int i) { return bits_to_f64(T.log_head[i]); };
  // |x - 1| small: log1p polynomial of degree 11 with the leading terms in double-double
  if (ix - 0x3fee000000000000ull < 0x0003090000000000ull) {  // 1 - 2^-4 <= x < 1 + 0x1.09p-4
    if (ix == 0x3ff0000000000000ull) return 0.0;
    const double r = x - 1.0;
    const double b12 = __builtin_fma(r, H(9), H(8));     // B1 + r B2
    const double b45 = __builtin_fma(r, H(12), H(11));   // B4 + r B5
    const double r2 = r * r;
    const double b78 = __builtin_fma(r, H(15), H(14));   // B7 + r B8
    const double b123 = __builtin_fma(r2, H(10), b12);   // + r2 B3
    const double b456 = __builtin_fma(r2, H(13), b45);   // + r2 B6
    const double r3 = r * r2;
    double p = __builtin_fma(r2, H(16), b78);            // B7 + r B8 + r2 B9
    p = __builtin_fma(r3, H(17), p);                     // + r3 B10
    p = __builtin_fma(p, r3, b456);
    p = __builtin_fma(p, r3, b123);                      // B1 + r B2 + r2 B3 + r3 (B4 + ... + r3 (B7 + ...))
    const double two27 = 134217728.0;
    const double rw = __builtin_fma(r, two27, r);        // r + r 2^27
    const double rhi = __builtin_fma(-two27, r, rw);     // ... - r 2^27
    const double B0 = H(7);                              // -0.5
    const double rhi2 = rhi * rhi;
    const double rlo = r - rhi;
    const double hi = __builtin_fma(rhi2, B0, r);        // r + rhi rhi B0
    const double d = r - hi;
    const double rs = r + rhi;
    double lo = __builtin_fma(rhi2, B0, d);              // r - hi + rhi rhi B0
    const double t = B0 * rlo;
    lo = __builtin_fma(t, rs, lo);                       // += B0 rlo (rhi + r)
    const double y = __builtin_fma(p, r3, lo);           // r3 (...) + lo
    return hi + y;
  }
  if (top - 0x0010u >= 0x7ff0u - 0x0010u) {  // zero, subnormal, negative, inf, nan
    if (ix * 2 == 0) return -__builtin_huge_val();                  // log(+-0) = -inf
    if (ix == 0x7ff0000000000000ull) return x;                      // log(inf) = inf
    if ((top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u) return bits_to_f64(0x7ff8000000000000ull);  // negative, nan: invalid
    ix = f64_to_bits(x * 4503599627370496.0);                       // subnormal: normalise
    ix -= 52ull << 52;
  }
  const uint64_t tmp = ix - 0x3fe6000000000000ull;
  const int i = (int)((tmp >> 45) & 127u);
  const int k = (int)((int64_t)tmp >> 52);
  const uint64_t iz = ix - (tmp & (0xfffull << 52));
  const double invc = bits_to_f64(T.log_tab[2 * i]), logc = bits_to_f64(T.log_tab[2 * i + 1]);
  const double z = bits_to_f64(iz);
  const double kd = (double)k;
  const double r = __builtin_fma(z, invc, -1.0);
  const double w = __builtin_fma(kd, H(0), logc);        // kd Ln2hi + logc
  const double a12 = __builtin_fma(r, H(4), H(3));       // A1 + r A2
  const double hi = r + w;
  const double r2 = r * r;
  double lo = w - hi;
  lo = lo + r;
  lo = __builtin_fma(kd, H(1), lo);                      // + kd Ln2lo
  const double r3 = r * r2;
  const double a34 = __builtin_fma(r, H(6), H(5));       // A3 + r A4
  lo = __builtin_fma(r2, H(2), lo);                      // + r2 A0
  const double q = __builtin_fma(a34, r2, a12);
  const double y = __builtin_fma(r3, q, lo);
  return y + hi;
}

// == glibc 2.35 __exp_fma(x)
CTC_HD double exp_f64(double x, const Tables64 &T) {
  const uint64_t ix = f64_to_bits(x);
  uint32_t abstop = (uint32_t)(ix >> 52) & 0x7ffu;
  auto H = [&](int i) { return bits_to_f64(T.exp_head[i]); };
  if (abstop - 0x3c9u >= 0x3fu) {
    if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + x;             // |x| < 2^-54
    if (abstop >= 0x409u) {                                         // |x| >= 1024, inf, nan
      if (ix == 0xfff0000000000000ull) return 0.0;                  // exp(-inf)
      if (abstop >= 0x7ffu) return 1.0 + x;                         // inf, nan
      return (ix >> 63) ? 0.0 : __builtin_huge_val();               // underflow to +0 / overflow
    }
    abstop = 0;                                                     // 512 <= |x| < 1024: the scale needs care
  }
  const double shift = H(1);
  const double kds = __builtin_fma(x, H(0), shift);                 // x InvLn2N + Shift
  const uint64_t ki = f64_to_bits(kds);
  const double kd = kds - shift;
  double r = __builtin_fma(kd, H(2), x);                            // x + kd NegLn2hiN
  r = __builtin_fma(kd, H(3), r);                                   // + kd NegLn2loN
  const int idx = 2 * (int)(ki & 127u);
  const uint64_t top = ki << 45;
  const double c23 = __builtin_fma(r, H(5), H(4));                  // C2 + r C3
  const double tail_r = r + bits_to_f64(T.exp_tab[idx]);            // tail + r
  uint64_t sbits = T.exp_tab[idx + 1] + top;
  const double r2 = r * r;
  const double c45 = __builtin_fma(r, H(7), H(6));                  // C4 + r C5
  const double t1 = __builtin_fma(c23, r2, tail_r);
  const double r4 = r2 * r2;
  const double tmp = __builtin_fma(r4, c45, t1);
  if (abstop == 0) {  // specialcase(): the exponent of the scale may have left the normal range
    if ((ki & 0x80000000ull) == 0) {  // k > 0
      sbits -= 1009ull << 52;
      const double scale = bits_to_f64(sbits);
      return 0x1p1009 * __builtin_fma(scale, tmp, scale);
    }
    sbits += 1022ull << 52;           // k < 0
    const double scale = bits_to_f64(sbits);
    const double sm = scale * tmp;
    double y = scale + sm;
    if (y < 1.0) {  // the result is subnormal: round once, at the right place
      double lo = scale - y;
      lo = lo + sm;
      const double hi = 1.0 + y;
      double lo2 = 1.0 - hi;
      lo2 = lo2 + y;
      lo2 = lo2 + lo;
      y = (hi + lo2) - 1.0;
      if (y == 0.0) y = 0.0;  // (the sign of zero)
    }
    return 0x1p-1022 * y;
  }
  const double scale = bits_to_f64(sbits);
  return __builtin_fma(scale, tmp, scale);
}

// log_sum_exp<double> exactly as decoder_utils.h:47-54 evaluates it
CTC_HD double lse_f64(double x, double y, const Tables64 &T) {
  const double neg = -1.7976931348623157e308;
  if (x <= neg) return y;
  if (y <= neg) return x;
  const double m = (x < y) ? y : x;  // std::max(x, y)
  return log_f64(exp_f64(x - m, T) + exp_f64(y - m, T), T) + m;
}

}  // namespace ctcmath
