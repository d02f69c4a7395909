// compact_results.h -- host-side expansion of the compact result form (beam_core.h OutRefs::c_*, include/ctcdecode_amd.h
// "Compact result delivery") into the reference's padded tensors (binding.cpp:85-99: tokens / timesteps [B, K, T]).
#pragma once
#include <stdint.h>
#include <cstring>
#include <vector>

namespace ctcbeam {

// One item: the entries come in trie (DFS) order; each row takes what it shares with its predecessor from the
// predecessor's finished row, then its own labels, then zeros.  Rows without a result are zeroed.
inline void expand_item_host(const int32_t *hdr, const int32_t *ent, const uint32_t *rag, int b, int K, int T, int32_t *tok, int32_t *ts) {
  const int nres = hdr[(size_t)b * 4];
  int32_t *tk0 = tok + (size_t)b * K * T, *ts0 = ts + (size_t)b * K * T;
  std::vector<unsigned long long> used((size_t)(K + 63) / 64, 0ull);
  int prow = 0;
  for (int j = 0; j < nres; ++j) {
    const int32_t *e = ent + ((size_t)b * K + j) * 4;
    const int row = e[0], lcp = e[1], dep = e[2];
    const uint32_t *seg = rag + (uint32_t)e[3];
    int32_t *tk = tk0 + (size_t)row * T, *tt = ts0 + (size_t)row * T;
    if (lcp > 0) {
      std::memcpy(tk, tk0 + (size_t)prow * T, (size_t)lcp * 4);
      std::memcpy(tt, ts0 + (size_t)prow * T, (size_t)lcp * 4);
    }
    for (int q = lcp; q < dep; ++q) {
      const uint32_t v = seg[q - lcp];
      tk[q] = (int32_t)(v & 0xFFFFu);
      tt[q] = (int32_t)(v >> 16);
    }
    if (dep < T) {
      std::memset(tk + dep, 0, (size_t)(T - dep) * 4);
      std::memset(tt + dep, 0, (size_t)(T - dep) * 4);
    }
    used[(size_t)row >> 6] |= 1ull << (row & 63);
    prow = row;
  }
  for (int p = 0; p < K; ++p)
    if (!((used[(size_t)p >> 6] >> (p & 63)) & 1ull)) {
      std::memset(tk0 + (size_t)p * T, 0, (size_t)T * 4);
      std::memset(ts0 + (size_t)p * T, 0, (size_t)T * 4);
    }
}

}  // namespace ctcbeam
