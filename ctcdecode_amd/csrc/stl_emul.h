// stl_emul.h -- the selection / sorting algorithms the reference inherits from its toolchain,
// restated so that a GPU lane can run them and obtain the SAME permutation.
//
// Why this exists: the reference prunes the beam with std::nth_element
// (ctcdecode/src/ctc_beam_search_decoder.cpp:150-154) and orders the final beams with two
// std::sort calls (ctc_beam_search_decoder.cpp:188-190, decoder_utils.cpp:59) under a
// comparator (decoder_utils.cpp:122-132) for which exact ties are structural at long T
// (SURVEY.md 7.3-H2).  Which of several comparator-equivalent prefixes survives, and in which
// order equal beams are returned, is therefore decided by the algorithm inside libstdc++
// (GCC 11: introselect / introsort with median-of-three pivots, Hoare partition, heap fallback
// after 2*floor(lg n) levels, insertion-sort finish with threshold 16).  These are textbook
// algorithms; this header is an independent array-index formulation of them (no iterators, no
// recursion), validated element-for-element against the real std::nth_element / std::sort /
// std::partial_sort by tests/native/stl_emul_check.cpp (random inputs with heavy ties plus
// adversarial "quicksort killer" inputs that force the heap fallbacks).
//
// All functions work on v[first, last) of some trivially copyable T with a strict-weak
// "goes before" predicate `before(a, b)`.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define STLEMU_HD __host__ __device__ inline
#else
#define STLEMU_HD inline
#endif
// The serial building blocks are real functions in device code (one copy per element type and comparator instead of one per
// call site: the decode kernels carried 22 KB of inlined copies, a third of their code) -- they run on rare paths (the exact
// replay of a tie at the beam's boundary, the final ordering of the results), one lane at a time, and take their array through
// a generic pointer.
#if defined(__HIP_DEVICE_COMPILE__)
#define STLEMU_FN __host__ __device__ __attribute__((noinline))
#else
#define STLEMU_FN STLEMU_HD
#endif

namespace stlemu {

STLEMU_HD int floor_lg(int n) {  // n >= 1
  int k = 0;
  while (n > 1) {
    n >>= 1;
    ++k;
  }
  return k;
}

template <class T>
STLEMU_HD void exch(T *v, int a, int b) {
  T t = v[a];
  v[a] = v[b];
  v[b] = t;
}

// Put the median of v[a], v[b], v[c] into v[res].
template <class T, class C>
STLEMU_HD void median_to(T *v, int res, int a, int b, int c, C before) {
  if (before(v[a], v[b])) {
    if (before(v[b], v[c]))
      exch(v, res, b);
    else if (before(v[a], v[c]))
      exch(v, res, c);
    else
      exch(v, res, a);
  } else if (before(v[a], v[c])) {
    exch(v, res, a);
  } else if (before(v[b], v[c])) {
    exch(v, res, c);
  } else {
    exch(v, res, b);
  }
}

// Hoare partition of v[lo, hi) around the value at v[piv] (piv outside [lo, hi)); no bounds checks:
// the caller guarantees sentinels exist (median-of-three does).
template <class T, class C>
STLEMU_HD int hoare_split(T *v, int lo, int hi, int piv, C before) {
  for (;;) {
    while (before(v[lo], v[piv])) ++lo;
    --hi;
    while (before(v[piv], v[hi])) --hi;
    if (!(lo < hi)) return lo;
    exch(v, lo, hi);
    ++lo;
  }
}

template <class T, class C>
STLEMU_FN int split_with_median_pivot(T *v, int first, int last, C before) {
  int mid = first + (last - first) / 2;
  median_to(v, first, first + 1, mid, last - 1, before);
  return hoare_split(v, first + 1, last, first, before);
}

// Binary max-heap (w.r.t. `before` as "less") helpers on v[base, base+len).
template <class T, class C>
STLEMU_HD void sift(T *v, int base, int hole, int len, T value, C before) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (before(v[base + child], v[base + child - 1])) --child;
    v[base + hole] = v[base + child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    v[base + hole] = v[base + child - 1];
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && before(v[base + parent], value)) {
    v[base + hole] = v[base + parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  v[base + hole] = value;
}

template <class T, class C>
STLEMU_HD void heapify(T *v, int first, int last, C before) {
  const int len = last - first;
  if (len < 2) return;
  int parent = (len - 2) / 2;
  for (;;) {
    T value = v[first + parent];
    sift(v, first, parent, len, value, before);
    if (parent == 0) return;
    --parent;
  }
}

// Heap on [first, middle); replace the top by v[at] (which receives the old top).
template <class T, class C>
STLEMU_HD void pop_to(T *v, int first, int middle, int at, C before) {
  T value = v[at];
  v[at] = v[first];
  sift(v, first, 0, middle - first, value, before);
}

template <class T, class C>
STLEMU_FN void heap_select(T *v, int first, int middle, int last, C before) {
  heapify(v, first, middle, before);
  for (int i = middle; i < last; ++i)
    if (before(v[i], v[first])) pop_to(v, first, middle, i, before);
}

template <class T, class C>
STLEMU_FN void heap_sort_down(T *v, int first, int last, C before) {
  while (last - first > 1) {
    --last;
    pop_to(v, first, last, last, before);
  }
}

template <class T, class C>
STLEMU_FN void linear_insert_unguarded(T *v, int at, C before) {
  T value = v[at];
  int prev = at - 1;
  while (before(value, v[prev])) {
    v[at] = v[prev];
    at = prev;
    --prev;
  }
  v[at] = value;
}

template <class T, class C>
STLEMU_FN void insertion_sort(T *v, int first, int last, C before) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (before(v[i], v[first])) {
      T value = v[i];
      for (int j = i; j > first; --j) v[j] = v[j - 1];
      v[first] = value;
    } else {
      linear_insert_unguarded(v, i, before);
    }
  }
}

// The introselect loop of std::nth_element, resumable: `depth` is the remaining partition budget
// (2*floor_lg(n) at the start).  beam_core.h runs the first, large partitions workgroup-parallel and hands the
// remainder to this serial form.
template <class T, class C>
STLEMU_HD void introselect(T *v, int first, int nth, int last, int depth, C before) {
  while (last - first > 3) {
    if (depth == 0) {
      heap_select(v, first, nth + 1, last, before);
      exch(v, first, nth);
      return;
    }
    --depth;
    int cut = split_with_median_pivot(v, first, last, before);
    if (cut <= nth)
      first = cut;
    else
      last = cut;
  }
  insertion_sort(v, first, last, before);
}

// == std::nth_element(v+first, v+nth, v+last, before)
template <class T, class C>
STLEMU_HD void nth_element(T *v, int first, int nth, int last, C before) {
  if (first == last || nth == last) return;
  introselect(v, first, nth, last, 2 * floor_lg(last - first), before);
}

// == std::sort(v+first, v+last, before).  `stack` needs 3 * (2*floor_lg(n) + 2) ints.
template <class T, class C>
STLEMU_HD void sort(T *v, int first, int last, C before, int *stack) {
  if (first == last) return;
  const int lo0 = first, hi0 = last;
  int sp = 0;
  int depth = 2 * floor_lg(last - first);
  // Each pending range carries the depth budget it was created with.  Sub-ranges are disjoint, so the
  // order in which they are finished does not change the result.
  for (;;) {
    while (last - first > 16) {
      if (depth == 0) {
        heap_select(v, first, last, last, before);
        heap_sort_down(v, first, last, before);
        break;
      }
      --depth;
      int cut = split_with_median_pivot(v, first, last, before);
      stack[sp++] = cut;  // right part [cut, last) postponed
      stack[sp++] = last;
      stack[sp++] = depth;
      last = cut;
    }
    if (sp == 0) break;
    depth = stack[--sp];
    last = stack[--sp];
    first = stack[--sp];
  }
  first = lo0;
  last = hi0;
  if (last - first > 16) {
    insertion_sort(v, first, first + 16, before);
    for (int i = first + 16; i != last; ++i) linear_insert_unguarded(v, i, before);
  } else {
    insertion_sort(v, first, last, before);
  }
}

// == std::sort(v, v + n, before), element for element -- also where `before` ties -- run by a whole workgroup.
// Introsort's sub-ranges are disjoint once split, so the order in which they are finished cannot change the result:
// every thread takes one pending range per round (median-of-three Hoare split, or the heap sort once the depth budget is
// spent), and the closing insertion sort -- which never moves an element across a split point, because the comparison
// is strict -- is done per final range (at most 16 elements) by one thread each.
// X: execution policy with tid(), nt(), sync(), atomic_add(int *, int), uni(int).  Scratch (workgroup-shared): two task
// arrays of 3 * task_cap entries (task_cap >= n / 17 + 1), `small` of 2 * (n / 2 + 1) entries (entry type I: int, or
// uint16_t when n < 65536), cnt[3].  n > 16.
// seeded_tasks >= 0: the caller has already split the largest ranges itself (e.g. with a workgroup-parallel partition):
// cur[] holds that many pending ranges {first, last, depth budget}, small[] the cnt[2] final ranges found so far, and
// cnt[0] == cnt[1] == 0.
// limit: only the first `limit` positions of the sorted array are wanted (a prefix of std::sort's result, not a
// std::partial_sort): sub-ranges are disjoint once split and the closing insertion sort never moves an element across a
// split point, so a range that starts at or beyond `limit` can be left as it is.
template <class X, class T, class C, class I>
STLEMU_HD void sort_parallel(X &x, T *v, int n, C before, I *cur, I *nxt, I *small, int *cnt, int seeded_tasks = -1,
                             int limit = 0x7fffffff) {
  const int tid = x.tid(), nt = x.nt();
  int ntask = seeded_tasks;
  if (seeded_tasks < 0) {
    if (tid == 0) {
      cnt[0] = 0; cnt[1] = 0; cnt[2] = 0;  // [0], [1]: pending ranges of the next round (by round parity), [2]: final ranges
      cur[0] = (I)0; cur[1] = (I)n; cur[2] = (I)(2 * floor_lg(n));
    }
    x.sync();
    ntask = 1;
  }
  for (int round = 0; ntask > 0; ++round) {
    int *c_next = cnt + (round & 1);
    for (int k = tid; k < ntask; k += nt) {
      const int first = (int)cur[3 * k], last = (int)cur[3 * k + 1], depth = (int)cur[3 * k + 2];
      if (depth == 0) {
        heap_select(v, first, last, last, before);
        heap_sort_down(v, first, last, before);
        continue;
      }
      const int cut = split_with_median_pivot(v, first, last, before);
      for (int side = 0; side < 2; ++side) {
        const int a = side ? cut : first, e = side ? last : cut;
        if (a >= limit) continue;  // (only v[0, limit) is wanted: a range that starts beyond it never influences it)
        if (e - a > 16) {
          const int i = x.atomic_add(c_next, 1);
          nxt[3 * i] = (I)a; nxt[3 * i + 1] = (I)e; nxt[3 * i + 2] = (I)(depth - 1);
        } else if (e - a > 1) {
          const int i = x.atomic_add(cnt + 2, 1);
          small[2 * i] = (I)a; small[2 * i + 1] = (I)e;
        }
      }
    }
    x.sync();
    ntask = x.uni(*c_next);
    if (tid == 0) cnt[(round + 1) & 1] = 0;  // the counter of the round after next; nobody reads it any more
    I *t = cur; cur = nxt; nxt = t;
    x.sync();
  }
  const int nsmall = x.uni(cnt[2]);
  for (int k = tid; k < nsmall; k += nt) insertion_sort(v, (int)small[2 * k], (int)small[2 * k + 1], before);
  x.sync();
}

// One partition step of introsort / introselect -- median of three to the front, unguarded Hoare partition of the rest
// (split_with_median_pivot) -- on v[first, last) by a whole workgroup.  Elements are ordered by key_of(e), larger first.
// Every thread counts, in its own stretch of the range, the stops of the two scans (left-to-right: elements not better
// than the pivot; right-to-left: elements not worse); one prefix over the threads turns the counts into the stops' ranks;
// the t-th stop from the left is exchanged with the t-th from the right until the scans cross -- the exchanges of the
// serial loop.  X additionally provides block_scan_u32(mine, &exclusive_prefix, &total) (contains a barrier) and the group
// operations lanes(), ballot(bool) -> mask, count(mask), count_below(mask) (set bits of the lanes below the caller's),
// first_lane(v) (the value the group's first lane holds).
// Lp, Rp: scratch for last - first + 1 positions each (P = uint16_t: n < 65536, the two counts travel in one scanned
// word; P = uint32_t: any n, two scans); *cutvar: one shared word.  Returns the cut.
// MEDIAN_BY_ALL: every thread reads the three candidates and the front element and finds the median itself; one thread then
// stores the two elements that trade places.  (One thread doing median_to alone is a chain of four to five dependent memory
// round trips with the workgroup waiting -- in HBM, where the first rounds of a wide beam's list live, 4-5 us of a 6.5 us round.)
template <bool MEDIAN_BY_ALL = false, class X, class T, class KeyOf, class P>
STLEMU_HD int hoare_round_parallel(X &x, T *v, int first, int last, KeyOf key_of, P *Lp, P *Rp, int *cutvar) {
  const int tid = x.tid(), nt = x.nt();
  constexpr bool wide = sizeof(P) > 2;
  auto before = [&](const T &a, const T &b) { return key_of(a) > key_of(b); };
  const int lo = first + 1, m = last - lo;
  decltype(key_of(v[first])) kp;
  if (MEDIAN_BY_ALL) {
    const int pa = first + 1, pb = first + (last - first) / 2, pc = last - 1;
    const T va = v[pa], vb = v[pb], vc = v[pc], vf = v[first];
    int pm;  // (median_to's decision tree)
    if (before(va, vb)) pm = before(vb, vc) ? pb : (before(va, vc) ? pc : pa);
    else pm = before(va, vc) ? pa : (before(vb, vc) ? pc : pb);
    const T vm = pm == pa ? va : (pm == pb ? vb : vc);
    x.sync();  // (everyone has read before the two stores)
    if (tid == 0) { v[first] = vm; v[pm] = vf; }
    x.sync();
    kp = key_of(vm);
  } else {
    if (tid == 0) median_to(v, first, first + 1, first + (last - first) / 2, last - 1, before);
    x.sync();
    kp = key_of(v[first]);
  }
  // The range is cut into one stretch per GROUP of x.lanes() threads (a wavefront on the GPU, a single thread on the host),
  // the lanes of a group taking neighbouring elements: the group's loads are contiguous (the array is in HBM for the first
  // rounds of a wide beam) and a stop's rank within the stretch is a ballot + a count of the lanes below.
  const int W = x.lanes(), lane = tid % W, grp = tid / W, ng = nt / W;
  const int per = ((m + ng - 1) / ng + W - 1) / W * W;  // stretch length: a multiple of the group width
  const int g0 = grp * per < m ? grp * per : m, g1 = g0 + per < m ? g0 + per : m;
  uint32_t mineL = 0, mineR = 0;  // #left stops, #right stops of this group's stretch (identical in its lanes)
  for (int i = g0; i < g1; i += W) {
    const bool in = i + lane < g1;
    const auto k = in ? key_of(v[lo + i + lane]) : kp;
    mineL += (uint32_t)x.count(x.ballot(in && k <= kp));
    mineR += (uint32_t)x.count(x.ballot(in && k >= kp));
  }
  uint32_t runL, runR;
  int nL, nR;
  if (wide) {
    uint32_t totL, totR;
    x.block_scan_u32(lane == 0 ? mineL : 0u, &runL, &totL);
    x.block_scan_u32(lane == 0 ? mineR : 0u, &runR, &totR);
    nL = (int)totL; nR = (int)totR;
  } else {
    uint32_t run, tot;
    x.block_scan_u32(lane == 0 ? (mineL | (mineR << 16)) : 0u, &run, &tot);
    runL = run & 0xFFFFu; runR = run >> 16;
    nL = (int)(tot & 0xFFFFu); nR = (int)(tot >> 16);
  }
  runL = x.first_lane(runL); runR = x.first_lane(runR);  // (the group's base: the exclusive prefix its first lane received)
  for (int i = g0; i < g1; i += W) {
    const bool in = i + lane < g1;
    const auto k = in ? key_of(v[lo + i + lane]) : kp;
    const bool isL = in && k <= kp, isR = in && k >= kp;
    const auto bl = x.ballot(isL), br = x.ballot(isR);
    if (isL) Lp[runL + (uint32_t)x.count_below(bl)] = (P)(lo + i + lane);
    if (isR) Rp[nR - 1 - (int)(runR + (uint32_t)x.count_below(br))] = (P)(lo + i + lane);
    runL += (uint32_t)x.count(bl);
    runR += (uint32_t)x.count(br);
  }
  if (tid == 0) Rp[nR] = (P)first;  // the pivot itself stops the right-to-left scan
  x.sync();
  // Iteration t of the serial loop stops its left scan at min(Lp[t], Rp[t-1]) (the element swapped into Rp[t-1] is itself
  // a stop) and ends, returning that position, as soon as it is not left of the right scan's stop.
  const int tmax = nL < nR + 1 ? nL : nR + 1;
  auto crossed = [&](int t) { return t >= nL || t > nR || Lp[t] >= Rp[t]; };
  for (int t = tid; t <= tmax; t += nt) {
    if (!crossed(t)) {
      exch(v, (int)Lp[t], (int)Rp[t]);
    } else if (t == 0 || !crossed(t - 1)) {
      int c = t < nL ? (int)Lp[t] : 0x7fffffff;
      if (t > 0 && (int)Rp[t - 1] < c) c = (int)Rp[t - 1];
      *cutvar = c;
    }
  }
  x.sync();
  return x.uni(*cutvar);
}

// The same step with one contiguous chunk per THREAD (no group operations): the form for arrays in LDS that are a few
// elements per thread long -- measured: at 3 000 elements the ballots of the group form cost more than its neighbouring
// loads save (the north-star kernel's frame 3.7 % longer), at 15 000 elements in HBM the group form wins (wide beam -6 %).
template <class X, class T, class KeyOf, class P>
STLEMU_HD int hoare_round_parallel_chunks(X &x, T *v, int first, int last, KeyOf key_of, P *Lp, P *Rp, int *cutvar) {
  const int tid = x.tid(), nt = x.nt();
  constexpr bool wide = sizeof(P) > 2;
  auto before = [&](const T &a, const T &b) { return key_of(a) > key_of(b); };
  if (tid == 0) median_to(v, first, first + 1, first + (last - first) / 2, last - 1, before);
  x.sync();
  const int lo = first + 1, m = last - lo;
  const auto kp = key_of(v[first]);
  const int chunk = (m + nt - 1) / nt, i0 = tid * chunk < m ? tid * chunk : m, i1 = i0 + chunk < m ? i0 + chunk : m;
  uint32_t mineL = 0, mineR = 0;  // #left stops, #right stops of this thread's stretch
  for (int i = i0; i < i1; ++i) {
    const auto k = key_of(v[lo + i]);
    mineL += k <= kp ? 1u : 0u;
    mineR += k >= kp ? 1u : 0u;
  }
  uint32_t runL, runR;
  int nL, nR;
  if (wide) {
    uint32_t totL, totR;
    x.block_scan_u32(mineL, &runL, &totL);
    x.block_scan_u32(mineR, &runR, &totR);
    nL = (int)totL; nR = (int)totR;
  } else {
    uint32_t run, tot;
    x.block_scan_u32(mineL | (mineR << 16), &run, &tot);
    runL = run & 0xFFFFu; runR = run >> 16;
    nL = (int)(tot & 0xFFFFu); nR = (int)(tot >> 16);
  }
  for (int i = i0; i < i1; ++i) {
    const auto k = key_of(v[lo + i]);
    if (k <= kp) { Lp[runL] = (P)(lo + i); runL += 1u; }
    if (k >= kp) { Rp[nR - 1 - (int)runR] = (P)(lo + i); runR += 1u; }
  }
  if (tid == 0) Rp[nR] = (P)first;  // the pivot itself stops the right-to-left scan
  x.sync();
  // Iteration t of the serial loop stops its left scan at min(Lp[t], Rp[t-1]) (the element swapped into Rp[t-1] is itself
  // a stop) and ends, returning that position, as soon as it is not left of the right scan's stop.
  const int tmax = nL < nR + 1 ? nL : nR + 1;
  auto crossed = [&](int t) { return t >= nL || t > nR || Lp[t] >= Rp[t]; };
  for (int t = tid; t <= tmax; t += nt) {
    if (!crossed(t)) {
      exch(v, (int)Lp[t], (int)Rp[t]);
    } else if (t == 0 || !crossed(t - 1)) {
      int c = t < nL ? (int)Lp[t] : 0x7fffffff;
      if (t > 0 && (int)Rp[t - 1] < c) c = (int)Rp[t - 1];
      *cutvar = c;
    }
  }
  x.sync();
  return x.uni(*cutvar);
}

// The first `limit` places of std::sort(v, v + n) (key_of(e), larger first) by one workgroup, element for element.  Ranges
// longer than big_cut are split one at a time by all threads (hoare_round_parallel; a stack of pending long ranges in
// bstack, 3 * 64 ints), the ones that remain go through sort_parallel, one thread per range and round.  Ranges that start
// at or beyond `limit` are left alone.  n < 65536; scratch as for sort_parallel with 16-bit lists, cnt[4].
template <class X, class T, class KeyOf>
STLEMU_HD void sort_prefix_parallel(X &x, T *v, int n, int limit, int big_cut, KeyOf key_of, uint16_t *Lp, uint16_t *Rp, uint16_t *cur,
                                    uint16_t *nxt, uint16_t *small, int *cnt, int *bstack) {
  const int tid = x.tid();
  auto before = [&](const T &a, const T &b) { return key_of(a) > key_of(b); };
  if (n <= 16) {
    if (tid == 0) sort(v, 0, n, before, bstack);
    x.sync();
    return;
  }
  int sp = 0, ntask = 0, nsmall = 0;  // (identical in every thread)
  if (tid == 0) { bstack[0] = 0; bstack[1] = n; bstack[2] = 2 * floor_lg(n); }
  sp = 1;
  x.sync();
  while (sp > 0) {
    --sp;
    const int first = x.uni(bstack[3 * sp]), last = x.uni(bstack[3 * sp + 1]);
    int depth = x.uni(bstack[3 * sp + 2]);
    x.sync();  // (the slot is about to be overwritten by a push)
    if (last - first <= big_cut) {
      if (tid == 0) { cur[3 * ntask] = (uint16_t)first; cur[3 * ntask + 1] = (uint16_t)last; cur[3 * ntask + 2] = (uint16_t)depth; }
      ++ntask;
      continue;
    }
    if (depth == 0) {  // depth budget spent on a long range (adversarial input): the heap sort, by one thread
      if (tid == 0) { heap_select(v, first, last, last, before); heap_sort_down(v, first, last, before); }
      x.sync();
      continue;
    }
    --depth;
    const int cut = hoare_round_parallel(x, v, first, last, key_of, Lp, Rp, cnt + 3);
    for (int side = 0; side < 2; ++side) {
      const int a = side ? cut : first, e = side ? last : cut;
      if (a >= limit) continue;  // only the first `limit` places are wanted (sort_parallel)
      if (e - a > 16) {
        if (tid == 0) { bstack[3 * sp] = a; bstack[3 * sp + 1] = e; bstack[3 * sp + 2] = depth; }
        ++sp;
      } else if (e - a > 1) {
        if (tid == 0) { small[2 * nsmall] = (uint16_t)a; small[2 * nsmall + 1] = (uint16_t)e; }
        ++nsmall;
      }
    }
    x.sync();
  }
  if (tid == 0) { cnt[0] = 0; cnt[1] = 0; cnt[2] = nsmall; }
  x.sync();
  sort_parallel(x, v, n, before, cur, nxt, small, cnt, ntask, limit);
}

// == std::partial_sort(v+first, v+middle, v+last, before)   (used by the tests to reach the heap code directly)
template <class T, class C>
STLEMU_HD void partial_sort(T *v, int first, int middle, int last, C before) {
  heap_select(v, first, middle, last, before);
  heap_sort_down(v, first, middle, before);
}

}  // namespace stlemu
