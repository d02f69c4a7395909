// Exhaustive host check of ctcdecode_amd/csrc/exact_math.h against the C library
// the reference binds to (decoder_utils.h:53 -> glibc expf/logf).
//   mode "expf": every float in [-88, -0] and +0          (2^31-ish / stride)
//   mode "logf": every float in [1, 2]
//   mode "lse" : N random pairs in the decoder's score range
// argv: mode [stride]   (stride > 1 subsamples; tests use 1 for logf, 1 for expf)
// Prints "mismatches=<n> checked=<m>" and exits non-zero on any mismatch.
#define CTC_EXACT_MATH_HOST_TABLES
#include "../../ctcdecode_amd/csrc/exact_math.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <random>
#include <thread>
#include <vector>
#include <atomic>
#include <algorithm>

static float ref_lse(float x, float y) {  // decoder_utils.h:47-54, verbatim semantics
  static float num_min = -std::numeric_limits<float>::max();
  if (x <= num_min) return y;
  if (y <= num_min) return x;
  float xmax = std::max(x, y);
  return std::log(std::exp(x - xmax) + std::exp(y - xmax)) + xmax;
}

int main(int argc, char **argv) {
  const char *mode = argc > 1 ? argv[1] : "logf";
  uint32_t stride = argc > 2 ? (uint32_t)atoi(argv[2]) : 1;
  const uint64_t *tbl = ctcmath::host_tables().w;
  unsigned nthr = std::max(1u, std::thread::hardware_concurrency());
  std::atomic<uint64_t> bad{0}, checked{0};
  std::vector<std::thread> th;
  if (!strcmp(mode, "expf") || !strcmp(mode, "logf")) {
    bool is_exp = !strcmp(mode, "expf");
    // expf: bit patterns 0x80000000 (-0) .. bits(-88.0f) are x in [-88, -0]
    uint32_t lo = is_exp ? 0x80000000u : 0x3f800000u;
    uint32_t hi = is_exp ? ctcmath::f32_to_bits(-88.0f) : 0x40000000u;
    for (unsigned w = 0; w < nthr; ++w)
      th.emplace_back([&, w] {
        uint64_t b = 0, c = 0;
        for (uint64_t u = (uint64_t)lo + (uint64_t)w * stride; u <= hi; u += (uint64_t)nthr * stride) {
          float x = ctcmath::bits_to_f32((uint32_t)u);
          float got = is_exp ? ctcmath::expf_nonpos(x, tbl) : ctcmath::logf_normal(x, tbl);
          float want = is_exp ? expf(x) : logf(x);
          if (ctcmath::f32_to_bits(got) != ctcmath::f32_to_bits(want)) {
            if (b < 5) fprintf(stderr, "%s(%a): got %a want %a\n", mode, x, got, want);
            ++b;
          }
          ++c;
        }
        bad += b; checked += c;
      });
    for (auto &t : th) t.join();
    if (is_exp) {  // the documented restriction: below -88 libm returns < 2^-126, so 1.0f + it == 1.0f
      for (float x : {-88.00001f, -90.f, -100.f, -103.5f, -104.f, -1000.f, -3.0e38f}) {
        float e = expf(x);
        if (!(1.0f + e == 1.0f) || ctcmath::expf_nonpos(x, tbl) != 0.0f) ++bad;
        ++checked;
      }
      if (ctcmath::f32_to_bits(ctcmath::expf_nonpos(0.0f, tbl)) != 0x3f800000u) ++bad;
    }
  } else {
    uint64_t n = argc > 2 ? strtoull(argv[2], 0, 10) : 20000000ull;
    for (unsigned w = 0; w < nthr; ++w)
      th.emplace_back([&, w] {
        std::mt19937_64 g(1234 + w);
        std::uniform_real_distribution<float> base(-4000.f, 0.f), d(0.f, 30.f);
        uint64_t b = 0, c = 0;
        for (uint64_t i = w; i < n; i += nthr) {
          float x = base(g);
          float y = (i % 7 == 0) ? x : x - d(g) * ((i % 3) ? 1.f : 0.05f);
          if (i % 1013 == 0) y = -std::numeric_limits<float>::max();
          if (i & 1) std::swap(x, y);
          float got = ctcmath::lse(x, y, tbl), want = ref_lse(x, y);
          if (ctcmath::f32_to_bits(got) != ctcmath::f32_to_bits(want)) {
            if (b < 5) fprintf(stderr, "lse(%a,%a): got %a want %a\n", x, y, got, want);
            ++b;
          }
          ++c;
        }
        bad += b; checked += c;
      });
    for (auto &t : th) t.join();
  }
  printf("mismatches=%llu checked=%llu\n", (unsigned long long)bad.load(), (unsigned long long)checked.load());
  return bad.load() ? 1 : 0;
}
