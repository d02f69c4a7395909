// compact_results.h -- host-side expansion of the compact result form (beam_core.h OutRefs::c_*, include/ctcdecode_amd.h
// "Compact result delivery") into the reference's padded tensors (binding.cpp:85-99: tokens / timesteps [B, K, T]).
#pragma once
#include <stdint.h>
#include <cstring>
#include <vector>
#if defined(__x86_64__)
#include <emmintrin.h>
#endif

namespace ctcbeam {

// dst[0, n) = src[0, n), dst[n, T) = 0, written with streaming (non-temporal) stores: the 2 x [B, K, T] result is far
// larger than the caches and is not read again here, so ordinary stores would first FETCH every line they overwrite.
inline void stream_row(int32_t *dst, const int32_t *src, int n, int T) {
#if defined(__x86_64__)
  int q = 0;
  while (q < T && ((uintptr_t)(dst + q) & 15u)) { _mm_stream_si32(dst + q, q < n ? src[q] : 0); ++q; }
  for (; q + 4 <= n; q += 4) _mm_stream_si128((__m128i *)(dst + q), _mm_loadu_si128((const __m128i *)(src + q)));
  if (q < n && q + 4 <= T) {  // the vector that straddles the end of the valid part
    int32_t tmp[4] = {0, 0, 0, 0};
    for (int i = 0; q + i < n; ++i) tmp[i] = src[q + i];
    _mm_stream_si128((__m128i *)(dst + q), _mm_loadu_si128((const __m128i *)tmp));
    q += 4;
  }
  const __m128i zero = _mm_setzero_si128();
  for (; q + 4 <= T; q += 4) _mm_stream_si128((__m128i *)(dst + q), zero);
  for (; q < T; ++q) _mm_stream_si32(dst + q, q < n ? src[q] : 0);
#else
  std::memcpy(dst, src, (size_t)n * 4);
  std::memset(dst + n, 0, (size_t)(T - n) * 4);
#endif
}

// One item: the entries come in trie (DFS) order.  The label sequence of entry j is built in a cache-resident row buffer
// -- it keeps what j shares with its predecessor, only the entry's own labels are written into it -- and streamed out to
// row `row` of the two tensors, zeros behind it.  Rows without a result are zeroed.
// (R rows of T labels per item in the output; K = the entries' stride in `ent` -- the beam width.  The one-shot decode has R == K;
//  a streaming call's results are sized to the most results / the longest beam of the batch, binding.cpp:186-205)
inline void expand_item_host(const int32_t *hdr, const int32_t *ent, const uint32_t *rag, int b, int K, int T, int32_t *tok, int32_t *ts, int R = -1) {
  if (R < 0) R = K;
  const int nres = hdr[(size_t)b * 4];
  int32_t *tk0 = tok + (size_t)b * R * T, *ts0 = ts + (size_t)b * R * T;
  std::vector<unsigned long long> used((size_t)(K + 63) / 64, 0ull);
  std::vector<int32_t> buf((size_t)2 * T + 8, 0);
  int32_t *bt = buf.data(), *bs = buf.data() + T + 4;
  for (int j = 0; j < nres; ++j) {
    const int32_t *e = ent + ((size_t)b * K + j) * 4;
    const int row = e[0], lcp = e[1], dep = e[2];
    const uint32_t *seg = rag + (uint32_t)e[3];
    for (int q = lcp; q < dep; ++q) {
      const uint32_t v = seg[q - lcp];
      bt[q] = (int32_t)(v & 0xFFFFu);
      bs[q] = (int32_t)(v >> 16);
    }
    stream_row(tk0 + (size_t)row * T, bt, dep, T);
    stream_row(ts0 + (size_t)row * T, bs, dep, T);
    used[(size_t)row >> 6] |= 1ull << (row & 63);
  }
  for (int p = 0; p < R; ++p)
    if (!((used[(size_t)p >> 6] >> (p & 63)) & 1ull)) {
      stream_row(tk0 + (size_t)p * T, bt, 0, T);
      stream_row(ts0 + (size_t)p * T, bs, 0, T);
    }
#if defined(__x86_64__)
  _mm_sfence();
#endif
}

}  // namespace ctcbeam
