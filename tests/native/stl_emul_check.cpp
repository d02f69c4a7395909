// Element-for-element check of ctcdecode_amd/csrc/stl_emul.h against the real libstdc++ algorithms the
// reference calls (std::nth_element: ctc_beam_search_decoder.cpp:151; std::sort: :76,:189, decoder_utils.cpp:23,59).
// Inputs: index arrays ordered by a key array with heavy ties (the comparator sees ties as equivalent, so the
// resulting permutation of the INDICES exposes every internal decision), plus adversarial inputs built with
// McIlroy's "killer adversary" against the live std::sort so that the depth-limit / heap fallbacks execute.
#include "../../ctcdecode_amd/csrc/stl_emul.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

static long long g_checked = 0, g_bad = 0;

struct ByKey {
  const int *key;
  bool operator()(uint32_t a, uint32_t b) const { return key[a] > key[b]; }  // "better first", like prefix_compare
};

static void check_all(const std::vector<int> &key, std::mt19937 &g) {
  const int n = (int)key.size();
  ByKey cmp{key.data()};
  std::vector<uint32_t> base(n);
  std::iota(base.begin(), base.end(), 0u);
  // sort
  {
    std::vector<uint32_t> a = base, b = base;
    std::vector<int> stack(3 * (2 * stlemu::floor_lg(n > 0 ? n : 1) + 4));
    std::sort(a.begin(), a.end(), cmp);
    stlemu::sort(b.data(), 0, n, cmp, stack.data());
    ++g_checked;
    if (a != b) { ++g_bad; fprintf(stderr, "sort mismatch n=%d\n", n); }
    // second sort of an already-sorted array (the reference sorts twice)
    std::sort(a.begin(), a.end(), cmp);
    stlemu::sort(b.data(), 0, n, cmp, stack.data());
    ++g_checked;
    if (a != b) { ++g_bad; fprintf(stderr, "re-sort mismatch n=%d\n", n); }
  }
  // sort_parallel (the workgroup form, run here by one "thread"): whole array, a prefix (`limit`: only ranges that reach
  // the first `limit` places are sorted -- what the prune pass's tie replay asks for), 16-bit task lists
  if (n > 16 && n < 65536) {
    struct SeqX {
      int tid() const { return 0; }
      int nt() const { return 1; }
      void sync() {}
      int atomic_add(int *p, int v) { int o = *p; *p += v; return o; }
      int uni(int v) const { return v; }
    } sx;
    std::vector<uint32_t> want = base;
    std::sort(want.begin(), want.end(), cmp);
    const int cap = n / 17 + 2;
    for (int rep = 0; rep < 3; ++rep) {
      const int limit = rep == 0 ? 0x7fffffff : rep == 1 ? std::min(n, 40) : 1 + (int)(g() % n);
      std::vector<uint32_t> b = base;
      int cnt[4] = {0, 0, 0, 0};
      if (rep == 2) {
        std::vector<uint16_t> cur(3 * cap), nxt(3 * cap), small(2 * (n / 2 + 1));
        stlemu::sort_parallel(sx, b.data(), n, cmp, cur.data(), nxt.data(), small.data(), cnt, -1, limit);
      } else {
        std::vector<int> cur(3 * cap), nxt(3 * cap), small(2 * (n / 2 + 1));
        stlemu::sort_parallel(sx, b.data(), n, cmp, cur.data(), nxt.data(), small.data(), cnt, -1, limit);
      }
      ++g_checked;
      const int upto = std::min(n, limit);
      if (!std::equal(b.begin(), b.begin() + upto, want.begin())) { ++g_bad; fprintf(stderr, "sort_parallel mismatch n=%d limit=%d\n", n, limit); }
    }
    // sort_prefix_parallel: long ranges through the workgroup form of the Hoare partition (two prefix counts instead of
    // two scans), the rest through sort_parallel -- the prune pass's tie replay, run here by one "thread"
    struct SeqX2 {
      int tid() const { return 0; }
      int nt() const { return 1; }
      void sync() {}
      int atomic_add(int *p, int v) { int o = *p; *p += v; return o; }
      int uni(int v) const { return v; }
      void block_scan_u32(uint32_t mine, uint32_t *base, uint32_t *total) { *base = 0; *total = mine; }
      int lanes() const { return 1; }
      uint32_t ballot(bool p) const { return p ? 1u : 0u; }
      int count(uint32_t m) const { return (int)m; }
      int count_below(uint32_t) const { return 0; }
      uint32_t first_lane(uint32_t v) const { return v; }
    } sx2;
    for (int rep = 0; rep < 3; ++rep) {
      const int limit = rep == 0 ? n : rep == 1 ? std::min(n, 40) : 1 + (int)(g() % n);
      const int big_cut = rep == 2 ? 17 + (int)(g() % 300) : 256;
      // (index, key) pairs in index order, as the prune pass builds them: key in the high half
      std::vector<unsigned long long> pv(n);
      for (int i = 0; i < n; ++i) pv[i] = ((unsigned long long)(uint32_t)(key[i] + 0x40000000) << 32) | (unsigned)i;
      std::vector<uint16_t> Lp(n + 2), Rp(n + 2), cur(3 * cap), nxt(3 * cap), small(2 * (n / 2 + 1));
      int cnt[4] = {0, 0, 0, 0}, bstack[3 * 64];
      stlemu::sort_prefix_parallel(sx2, pv.data(), n, limit, big_cut, [](unsigned long long e) { return (uint32_t)(e >> 32); }, Lp.data(), Rp.data(),
                                   cur.data(), nxt.data(), small.data(), cnt, bstack);
      ++g_checked;
      bool ok = true;
      for (int i = 0; i < std::min(n, limit); ++i) ok = ok && (uint32_t)pv[i] == want[i];
      if (!ok) { ++g_bad; fprintf(stderr, "sort_prefix_parallel mismatch n=%d limit=%d big_cut=%d\n", n, limit, big_cut); }
    }
  }
  // one partition round by the workgroup forms (group form with the median found by one thread / by every thread -- the wide-beam
  // layouts' --, chunk form) against the serial split_with_median_pivot: same cut, same arrangement, heavy ties included
  if (n > 3 && n < 65536) {
    struct SeqX3 {
      int tid() const { return 0; }
      int nt() const { return 1; }
      void sync() {}
      int uni(int v) const { return v; }
      void block_scan_u32(uint32_t mine, uint32_t *base, uint32_t *total) { *base = 0; *total = mine; }
      int lanes() const { return 1; }
      uint32_t ballot(bool p) const { return p ? 1u : 0u; }
      int count(uint32_t m) const { return (int)m; }
      int count_below(uint32_t) const { return 0; }
      uint32_t first_lane(uint32_t v) const { return v; }
    } sx3;
    const int first = (int)(g() % (n - 3)), last = first + 4 + (int)(g() % (n - first - 3));
    auto key_of = [&](uint32_t e) { return key[e]; };
    std::vector<uint32_t> want = base;
    const int cut_want = stlemu::split_with_median_pivot(want.data(), first, last, cmp);
    for (int form = 0; form < 3; ++form) {
      std::vector<uint32_t> b = base;
      std::vector<uint16_t> Lp(n + 2), Rp(n + 2);
      int cutvar = -1;
      const int cut = form == 0 ? stlemu::hoare_round_parallel<false>(sx3, b.data(), first, last, key_of, Lp.data(), Rp.data(), &cutvar)
                    : form == 1 ? stlemu::hoare_round_parallel<true>(sx3, b.data(), first, last, key_of, Lp.data(), Rp.data(), &cutvar)
                                : stlemu::hoare_round_parallel_chunks(sx3, b.data(), first, last, key_of, Lp.data(), Rp.data(), &cutvar);
      ++g_checked;
      if (cut != cut_want || b != want) { ++g_bad; fprintf(stderr, "partition round mismatch form=%d n=%d [%d,%d) cut %d vs %d\n", form, n, first, last, cut, cut_want); }
    }
  }
  // nth_element at a few positions
  for (int rep = 0; rep < 4 && n > 0; ++rep) {
    int nth = rep == 0 ? n / 2 : rep == 1 ? std::min(n, 100) % (n + 1) : (int)(g() % (n + 1));
    std::vector<uint32_t> a = base, b = base;
    std::nth_element(a.begin(), a.begin() + nth, a.end(), cmp);
    stlemu::nth_element(b.data(), 0, nth, n, cmp);
    ++g_checked;
    if (a != b) { ++g_bad; fprintf(stderr, "nth_element mismatch n=%d nth=%d\n", n, nth); }
  }
  // partial_sort (reaches heap_select / sift directly)
  if (n > 0) {
    int mid = (int)(g() % (n + 1));
    std::vector<uint32_t> a = base, b = base;
    std::partial_sort(a.begin(), a.begin() + mid, a.end(), cmp);
    stlemu::partial_sort(b.data(), 0, mid, n, cmp);
    ++g_checked;
    if (a != b) { ++g_bad; fprintf(stderr, "partial_sort mismatch n=%d mid=%d\n", n, mid); }
  }
}

// McIlroy, "A Killer Adversary for Quicksort" (1999): decide values lazily while the real algorithm runs.
struct Adversary {
  std::vector<int> val;
  int gas, nsolid = 0, candidate = 0;
  explicit Adversary(int n) : val(n, n), gas(n) {}
  bool less(int x, int y) {
    if (val[x] == gas && val[y] == gas) {
      if (x == candidate) val[x] = nsolid++; else val[y] = nsolid++;
    }
    if (val[x] == gas) candidate = x; else if (val[y] == gas) candidate = y;
    return val[x] < val[y];
  }
};

static std::vector<int> killer_for_sort(int n, bool descending) {
  Adversary adv(n);
  std::vector<int> idx(n);
  std::iota(idx.begin(), idx.end(), 0);
  std::sort(idx.begin(), idx.end(), [&](int a, int b) { return descending ? adv.less(b, a) : adv.less(a, b); });
  for (int &v : adv.val) if (v == adv.gas) v = adv.nsolid++;
  return adv.val;
}

static std::vector<int> killer_for_nth(int n, int nth) {
  Adversary adv(n);
  std::vector<int> idx(n);
  std::iota(idx.begin(), idx.end(), 0);
  std::nth_element(idx.begin(), idx.begin() + nth, idx.end(), [&](int a, int b) { return adv.less(b, a); });
  for (int &v : adv.val) if (v == adv.gas) v = adv.nsolid++;
  return adv.val;
}

int main(int argc, char **argv) {
  int rounds = argc > 1 ? atoi(argv[1]) : 3000;
  std::mt19937 g(12345);
  for (int r = 0; r < rounds; ++r) {
    int n = r < 70 ? r : (int)(g() % 3200);
    int distinct = 1 + (int)(g() % (r % 3 == 0 ? 4 : r % 3 == 1 ? 64 : 100000));
    std::vector<int> key(n);
    for (int &k : key) k = (int)(g() % distinct);
    if (r % 5 == 0) std::sort(key.begin(), key.end());
    if (r % 7 == 0) std::sort(key.begin(), key.end(), std::greater<int>());
    check_all(key, g);
  }
  for (int n : {17, 33, 100, 257, 1000, 2900, 5000}) {
    check_all(killer_for_sort(n, false), g);
    check_all(killer_for_sort(n, true), g);
    check_all(killer_for_nth(n, std::min(100, n / 2)), g);
    // killers with ties folded in
    std::vector<int> k = killer_for_sort(n, true);
    for (int &v : k) v /= 3;
    check_all(k, g);
  }
  // make sure the adversarial inputs really exercised the fallbacks: count comparisons of the real sort
  {
    std::vector<int> k = killer_for_sort(5000, true);
    long long ncmp = 0;
    std::vector<int> idx(5000);
    std::iota(idx.begin(), idx.end(), 0);
    std::sort(idx.begin(), idx.end(), [&](int a, int b) { ++ncmp; return k[a] > k[b]; });
    printf("killer comparisons at n=5000: %lld (n lg n = %d)\n", ncmp, 5000 * 12);
  }
  printf("mismatches=%lld checked=%lld\n", g_bad, g_checked);
  return g_bad ? 1 : 0;
}
