// exact_math.h -- float32 log-sum-exp that is BIT-IDENTICAL to the reference's.
//
// The reference accumulates prefix probabilities with
//     log_sum_exp<float>(x, y) = logf(expf(x - m) + expf(y - m)) + m
// (ctcdecode/src/decoder_utils.h:47-54), where expf/logf bind to the host C
// library (glibc 2.35 in this image; IFUNC-resolved to the FMA variants on any
// AVX2+FMA CPU).  Those routines are NOT correctly rounded, and at |score| ~
// 1800 one float ulp (1.2e-4) exceeds the 1e-4 score tolerance, so the only way
// to reproduce the reference's scores -- and through them its top-K decisions
// -- is to reproduce expf/logf bit for bit.  This header restates the glibc
// 2.35 algorithms (sysdeps/ieee754/flt-32/e_expf.c, e_logf.c: table lookup +
// degree-3 polynomial evaluated in double, one final rounding to float) with
// the exact operation order and fused multiply-adds of the x86-64 `*_fma`
// variants (read from the disassembly of this image's libm.so.6; recipe in
// tools/extract_libm_tables.py).  All arithmetic is IEEE binary64 add/mul/fma,
// which gfx950 implements exactly, so host and device agree by construction.
// tests/test_exact_math.py checks the host build against libm exhaustively on
// the log-sum-exp domain; tests/test_gpu_decode.py::test_device_math_bit_exact_vs_host_libm does the same for the device
// build against the GPU box's own libm.
//
// Build note: compile with -ffp-contract=off.  Every fused operation below is
// an explicit __builtin_fma; nothing else may be contracted.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define CTC_HD __host__ __device__ __forceinline__
#else
#define CTC_HD inline
#endif

namespace ctcmath {

// "minus infinity" of the reference: -FLT_MAX, never IEEE -inf (decoder_utils.h:12).
#define CTC_NEG_MAX (-3.402823466e+38f)

// glibc __exp2f_data (N = 32): tab[i] = bits(2^(i/32)) - (i << 47).
#define CTC_EXP2F_TAB                                                                          \
  {0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, \
   0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, \
   0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull, \
   0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, \
   0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull, \
   0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, \
   0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, \
   0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull}

// glibc __logf_data (N = 16): {invc, logc} pairs, stored here as two arrays.
#define CTC_LOGF_INVC                                                                      \
  {0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010b0p+0, 0x1.3c995b0b80385p+0, \
   0x1.30d190c8864a5p+0, 0x1.25e227b0b8ea0p+0, 0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0, \
   0x1.0953f419900a7p+0, 0x1.0000000000000p+0, 0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aa0p-1, \
   0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1}
#define CTC_LOGF_LOGC                                                                          \
  {-0x1.57bf7808caadep-2, -0x1.2bef0a7c06ddbp-2, -0x1.01eae7f513a67p-2, -0x1.b31d8a68224e9p-3, \
   -0x1.6574f0ac07758p-3, -0x1.1aa2bc79c8100p-3, -0x1.a4e76ce8c0e5ep-4, -0x1.1973c5a611cccp-4, \
   -0x1.252f438e10c1ep-5, 0x0.0p+0,             0x1.aa5aa5df25984p-5,  0x1.c5e53aa362eb4p-4,  \
   0x1.526e57720db08p-3,  0x1.bc2860d224770p-3,  0x1.1058bc8a07ee1p-2,  0x1.4043057b6ee09p-2}

// The 64 table words the routines need, in one block so a kernel can stage
// them into LDS with a single coalesced copy: [0,32) exp2f tab (as bits),
// [32,48) logf invc (as bits of the double), [48,64) logf logc.
struct Tables {
  uint64_t w[64];
};

CTC_HD double bits_to_f64(uint64_t u) {
  union { uint64_t u; double d; } c;
  c.u = u;
  return c.d;
}
CTC_HD uint64_t f64_to_bits(double d) {
  union { uint64_t u; double d; } c;
  c.d = d;
  return c.u;
}
CTC_HD uint32_t f32_to_bits(float f) {
  union { uint32_t u; float f; } c;
  c.f = f;
  return c.u;
}
CTC_HD float bits_to_f32(uint32_t u) {
  union { uint32_t u; float f; } c;
  c.u = u;
  return c.f;
}

// expf for the log-sum-exp domain: x <= 0 and finite (x = a - max(a, b)).
// Main path = glibc e_expf.c, FMA build:
//   kd' = fma(InvLn2N, xd, SHIFT); ki = bits(kd'); kd = kd' - SHIFT;
//   r = fma(InvLn2N, xd, -kd); s = tab[ki % 32] + (ki << 47);
//   z = fma(r, C0, C1); r2 = r*r; y = fma(r, C2, 1); y = fma(z, r2, y); (float)(y*s)
// For x < -88 glibc takes its special-case branch and returns a value below
// 2^-126; inside log_sum_exp that value is added to exactly 1.0f and cannot
// change the float sum, so 0 is returned here (documented domain restriction).
CTC_HD float expf_nonpos(float x, const uint64_t *tbl) {
  if (x < -88.0f) return 0.0f;
  const double InvLn2N = 0x1.71547652b82fep+5;
  const double Shift = 0x1.8p+52;
  const double C0 = 0x1.c6af84b912394p-20, C1 = 0x1.ebfce50fac4f3p-13, C2 = 0x1.62e42ff0c52d6p-6;
  double xd = (double)x;
  double kds = __builtin_fma(InvLn2N, xd, Shift);
  uint64_t ki = f64_to_bits(kds);
  double kd = kds - Shift;
  double r = __builtin_fma(InvLn2N, xd, -kd);
  uint64_t t = tbl[ki & 31] + (ki << 47);
  double s = bits_to_f64(t);
  double z = __builtin_fma(r, C0, C1);
  double r2 = r * r;
  double y = __builtin_fma(r, C2, 1.0);
  y = __builtin_fma(z, r2, y);
  return (float)(y * s);
}

// The same evaluation before its final rounding to float: exp(x) as a double with a relative error of ~2e-10 (the degree-3
// polynomial's), for -88 <= x <= 80 (below: 0).  NOT part of any bit-exact result -- the vocabulary prune's fast cumulative cut
// uses it where a comparison has nine digits to spare, and leaves everything closer to the exact chain.
CTC_HD double expf_core_f64(float x, const uint64_t *tbl) {
  if (x < -88.0f) return 0.0;
  const double InvLn2N = 0x1.71547652b82fep+5;
  const double Shift = 0x1.8p+52;
  const double C0 = 0x1.c6af84b912394p-20, C1 = 0x1.ebfce50fac4f3p-13, C2 = 0x1.62e42ff0c52d6p-6;
  double xd = (double)x;
  double kds = __builtin_fma(InvLn2N, xd, Shift);
  uint64_t ki = f64_to_bits(kds);
  double kd = kds - Shift;
  double r = __builtin_fma(InvLn2N, xd, -kd);
  uint64_t t = tbl[ki & 31] + (ki << 47);
  double s = bits_to_f64(t);
  double z = __builtin_fma(r, C0, C1);
  double r2 = r * r;
  double y = __builtin_fma(r, C2, 1.0);
  y = __builtin_fma(z, r2, y);
  return y * s;
}

// logf for normal positive finite x (the log-sum-exp domain is [1, 2]).
// glibc e_logf.c, FMA build:
//   tmp = ix - 0x3f330000; i = (tmp >> 19) % 16; k = (int)tmp >> 23; iz = ix - (tmp & 0xff800000)
//   r = fma(z, invc, -1); y0 = fma(k, Ln2, logc); y = fma(r, A1, A2); r2 = r*r;
//   y = fma(A0, r2, y); (float) fma(r2, y, r + y0)          and logf(1) = +0 exactly.
CTC_HD float logf_normal(float x, const uint64_t *tbl) {
  uint32_t ix = f32_to_bits(x);
  if (ix == 0x3f800000u) return 0.0f;
  const double Ln2 = 0x1.62e42fefa39efp-1;
  const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
  uint32_t tmp = ix - 0x3f330000u;
  int i = (int)((tmp >> 19) & 15u);
  int k = (int32_t)tmp >> 23;
  uint32_t iz = ix - (tmp & 0xff800000u);
  double invc = bits_to_f64(tbl[32 + i]);
  double logc = bits_to_f64(tbl[48 + i]);
  double z = (double)bits_to_f32(iz);
  double r = __builtin_fma(z, invc, -1.0);
  double y0 = __builtin_fma((double)k, Ln2, logc);
  double y = __builtin_fma(r, A1, A2);
  double r2 = r * r;
  double t = r + y0;
  y = __builtin_fma(A0, r2, y);
  return (float)__builtin_fma(r2, y, t);
}

// log_sum_exp<float> exactly as decoder_utils.h:47-54 evaluates it:  logf(expf(x - m) + expf(y - m)) + m, m = max.
// One of the two exponentials is expf(+0) = 1.0f exactly (glibc returns exactly 1 for a zero argument), and float
// addition commutes, so only the other one is evaluated: the sum, and everything after it, is bit-identical.
CTC_HD float lse(float x, float y, const uint64_t *tbl) {
  if (x <= CTC_NEG_MAX) return y;
  if (y <= CTC_NEG_MAX) return x;
  const float m = (x < y) ? y : x;  // std::max(x, y)
  const float lo = (x < y) ? x : y;
  const float s = 1.0f + expf_nonpos(lo - m, tbl);
  return logf_normal(s, tbl) + m;
}

}  // namespace ctcmath

// Host-side copy of the tables (one definition per translation unit that asks for it).
#if defined(CTC_EXACT_MATH_HOST_TABLES)
namespace ctcmath {
inline const Tables &host_tables() {
  static const Tables t = [] {
    Tables r;
    const uint64_t e[32] = CTC_EXP2F_TAB;
    const double ic[16] = CTC_LOGF_INVC;
    const double lc[16] = CTC_LOGF_LOGC;
    for (int i = 0; i < 32; ++i) r.w[i] = e[i];
    for (int i = 0; i < 16; ++i) {
      r.w[32 + i] = f64_to_bits(ic[i]);
      r.w[48 + i] = f64_to_bits(lc[i]);
    }
    return r;
  }();
  return t;
}
}  // namespace ctcmath
#endif
