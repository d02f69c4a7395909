// Host check of ctcdecode_amd/csrc/exact_math_f64.h against the C library the reference binds to
// (decoder_utils.cpp:16,29,42 / decoder_utils.h:53 with T = double -> glibc log / exp, binary64).
//   mode "log"  : float images -- every <stride>-th positive float p, log(p) and log(p + FLT_MIN) (decoder_utils.cpp:29,42)
//                 -- plus the whole |x - 1| < 2^-4 branch on a grid, subnormals, powers of two, random doubles
//   mode "exp"  : random doubles over [-760, 0], the 512 <= |x| < 1024 branch, tiny and huge arguments, a grid near 0
//   mode "lse"  : log_sum_exp<double> on the values the cumulative cut sees (decoder_utils.cpp:26-31)
// argv: mode [stride]; prints "mismatches=<n> checked=<m>", exits non-zero on any mismatch.  NaN results compare as equal.
#include "../../ctcdecode_amd/csrc/exact_math_f64.h"

#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <random>
#include <thread>
#include <vector>

static const ctcmath::Tables64 T = CTC_TABLES64_INIT;

static bool same(double a, double b) {
  if (a != a && b != b) return true;
  return ctcmath::f64_to_bits(a) == ctcmath::f64_to_bits(b);
}
static double ref_lse(double x, double y) {  // decoder_utils.h:47-54, T = double
  static double num_min = -std::numeric_limits<double>::max();
  if (x <= num_min) return y;
  if (y <= num_min) return x;
  double xmax = std::max(x, y);
  return std::log(std::exp(x - xmax) + std::exp(y - xmax)) + xmax;
}

int main(int argc, char **argv) {
  const char *mode = argc > 1 ? argv[1] : "log";
  const uint32_t stride = argc > 2 ? (uint32_t)atoi(argv[2]) : 7;
  const unsigned nthr = std::max(1u, std::thread::hardware_concurrency());
  std::atomic<uint64_t> bad{0}, checked{0};
  std::vector<std::thread> th;
  auto chk_log = [&](double x, uint64_t &b, uint64_t &c) {
    const double got = ctcmath::log_f64(x, T), want = log(x);
    if (!same(got, want)) { if (b < 5) fprintf(stderr, "log(%a): got %a want %a\n", x, got, want); ++b; }
    ++c;
  };
  auto chk_exp = [&](double x, uint64_t &b, uint64_t &c) {
    const double got = ctcmath::exp_f64(x, T), want = exp(x);
    if (!same(got, want)) { if (b < 5) fprintf(stderr, "exp(%a): got %a want %a\n", x, got, want); ++b; }
    ++c;
  };
  for (unsigned w = 0; w < nthr; ++w)
    th.emplace_back([&, w] {
      uint64_t b = 0, c = 0;
      std::mt19937_64 g(99 + w);
      if (!strcmp(mode, "log")) {
        // float images: bit patterns 1 .. 0x7f800000 (subnormal floats .. +inf)
        for (uint64_t u = 1 + (uint64_t)w * stride; u <= 0x7f800000ull; u += (uint64_t)nthr * stride) {
          const float p = ctcmath::bits_to_f32((uint32_t)u);
          chk_log((double)p, b, c);
          chk_log((double)p + (double)FLT_MIN, b, c);  // decoder_utils.cpp:42 (NUM_FLT_MIN is a float constant)
        }
        // the |x - 1| < 2^-4 branch and its borders, on a grid of doubles
        const uint64_t lo = 0x3fed000000000000ull, hi = 0x3ff2000000000000ull;
        for (uint64_t u = lo + w * 1048573ull; u < hi; u += (uint64_t)nthr * 1048573ull) chk_log(ctcmath::bits_to_f64(u), b, c);
        // sums exp(a) + exp(b) of the cumulative cut lie in [1, 2]: a grid there
        for (uint64_t u = 0x3ff0000000000000ull + w * 2097143ull; u <= 0x4000000000000000ull; u += (uint64_t)nthr * 2097143ull) chk_log(ctcmath::bits_to_f64(u), b, c);
        std::uniform_int_distribution<uint64_t> any(1, 0x7fefffffffffffffull);
        for (int i = 0; i < 2000000; ++i) chk_log(ctcmath::bits_to_f64(any(g)), b, c);
        if (w == 0) {
          for (double x : {0.0, -0.0, 1.0, -1.0, 4.9e-324, 2.2250738585072014e-308, 1e-310, (double)INFINITY, -(double)INFINITY, (double)NAN, 0.99, 1.0 - 0x1p-4, 1.0 + 0x1.09p-4, 0.5, 2.0})
            chk_log(x, b, c);
        }
      } else if (!strcmp(mode, "exp")) {
        std::uniform_real_distribution<double> d1(-760.0, 0.0), d2(-1.0, 0.0), d3(-1100.0, -500.0), d4(-50.0, 50.0);
        for (int i = 0; i < 3000000; ++i) { chk_exp(d1(g), b, c); chk_exp(d2(g), b, c); chk_exp(d4(g), b, c); }
        for (int i = 0; i < 500000; ++i) chk_exp(d3(g), b, c);
        // differences of float images (x - max with both doubles that came from floats): a grid of float differences
        for (uint64_t u = 0x80000000ull + (uint64_t)w * stride * 3; u <= 0xc4800000ull; u += (uint64_t)nthr * stride * 3)
          chk_exp((double)ctcmath::bits_to_f32((uint32_t)u), b, c);
        if (w == 0) {
          for (double x : {0.0, -0.0, -0x1p-54, -0x1p-55, 0x1p-60, -1e-300, -708.0, -709.5, -745.0, -745.2, -746.0, -1023.9, -1024.0, -1e10, -3.4e38, -1.7e308,
                           709.0, 709.8, 710.0, 1e5, -(double)INFINITY, (double)INFINITY, (double)NAN})
            chk_exp(x, b, c);
        }
      } else {
        std::uniform_real_distribution<double> base(-30.0, 2.0), d(0.0, 60.0);
        for (int i = 0; i < 3000000; ++i) {
          double x = base(g), y = (i % 7 == 0) ? x : x - d(g) * ((i % 3) ? 1.0 : 0.01);
          if (i % 5 == 0) { x = (double)(float)x; y = (double)(float)y; }
          if (i % 1013 == 0) y = -std::numeric_limits<double>::max();
          if (i % 977 == 0) y = -(double)std::numeric_limits<float>::max();
          if (i & 1) std::swap(x, y);
          const double got = ctcmath::lse_f64(x, y, T), want = ref_lse(x, y);
          if (!same(got, want)) { if (b < 5) fprintf(stderr, "lse(%a,%a): got %a want %a\n", x, y, got, want); ++b; }
          ++c;
        }
      }
      bad += b; checked += c;
    });
  for (auto &t : th) t.join();
  printf("mismatches=%llu checked=%llu\n", (unsigned long long)bad.load(), (unsigned long long)checked.load());
  return bad.load() ? 1 : 0;
}
