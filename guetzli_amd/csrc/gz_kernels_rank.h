// The ranked input_order of ComputeBlockZeroingOrder (processor.cc:381-400) on the device:
// per block, every non-zero AC coefficient of the three components scored by the zeroing
// model (order.inc csf/bias, or the old zig-zag model) and std::sort-ed by score.
//
// Equal scores do occur and std::sort is not stable, so the permutation has to be the one
// libstdc++'s std::sort produces: introsort (median-of-3 moved to the front, unguarded Hoare
// partition, depth limit 2*floor(log2 n) with the heap-sort fall-back), then the final
// insertion sort over 16-element runs (bits/stl_algo.h, bits/stl_heap.h) -- restated here
// step by step.  At most 189 elements per block and tens of thousands of independent blocks:
// one lane sorts one block, its elements in LDS ([element][lane], so that lanes that are at
// the same element index hit different banks).
#pragma once
#include "gz_common.h"
#include "gz_kernels_block.h"   // GZ_CONST

namespace gz {

constexpr int kRankLanes = 64;
constexpr int kRankMax = 192;

// One lane's view of its (key, id) array in LDS.
struct RankArr {
  float* key;            // [kRankMax][kRankLanes]
  unsigned char* id;     // [kRankMax][kRankLanes]
  int t;
  GZ_DEVFN float k(int i) const { return key[i * kRankLanes + t]; }
  GZ_DEVFN unsigned char d(int i) const { return id[i * kRankLanes + t]; }
  GZ_DEVFN void set(int i, float kk, unsigned char dd) const {
    key[i * kRankLanes + t] = kk;
    id[i * kRankLanes + t] = dd;
  }
  GZ_DEVFN void move(int dst, int src) const { set(dst, k(src), d(src)); }
  GZ_DEVFN void swap(int a, int b) const {
    const float ka = k(a);
    const unsigned char da = d(a);
    set(a, k(b), d(b));
    set(b, ka, da);
  }
};

// std::__adjust_heap + std::__push_heap (bits/stl_heap.h), on [first, first + len)
GZ_DEVFN void rank_adjust_heap(const RankArr& a, int first, int hole, int len, float vk,
                               unsigned char vd) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (a.k(first + child) < a.k(first + child - 1)) --child;
    a.move(first + hole, first + child);
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    a.move(first + hole, first + child - 1);
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && a.k(first + parent) < vk) {
    a.move(first + hole, first + parent);
    hole = parent;
    parent = (hole - 1) / 2;
  }
  a.set(first + hole, vk, vd);
}

// std::__partial_sort(first, last, last): __heap_select (= __make_heap) + __sort_heap
GZ_DEVFN void rank_heap_sort(const RankArr& a, int first, int last) {
#ifdef GZ_RANK_COUNT_HEAP
  ++g_rank_heap_calls;   // tests/cpp/test_rank_sort.cc: the fall-back must have been exercised
#endif
  const int len = last - first;
  if (len >= 2) {
    int parent = (len - 2) / 2;
    for (;;) {
      rank_adjust_heap(a, first, parent, len, a.k(first + parent), a.d(first + parent));
      if (parent == 0) break;
      --parent;
    }
  }
  while (last - first > 1) {
    --last;
    const float vk = a.k(last);
    const unsigned char vd = a.d(last);
    a.move(last, first);
    rank_adjust_heap(a, first, 0, last - first, vk, vd);
  }
}

// std::__unguarded_partition_pivot: median of (first+1, mid, last-1) to first, then the
// unguarded Hoare partition of [first+1, last) around *first
GZ_DEVFN int rank_partition_pivot(const RankArr& a, int first, int last) {
  const int mid = first + (last - first) / 2;
  const int x = first + 1, y = mid, z = last - 1;
  const float kx = a.k(x), ky = a.k(y), kz = a.k(z);
  int m;
  if (kx < ky) {
    if (ky < kz) m = y;
    else if (kx < kz) m = z;
    else m = x;
  } else if (kx < kz) {
    m = x;
  } else if (ky < kz) {
    m = z;
  } else {
    m = y;
  }
  a.swap(first, m);
  const float pivot = a.k(first);
  int lo = first + 1, hi = last;
  for (;;) {
    while (a.k(lo) < pivot) ++lo;
    --hi;
    while (pivot < a.k(hi)) --hi;
    if (!(lo < hi)) return lo;
    a.swap(lo, hi);
    ++lo;
  }
}

// std::__unguarded_linear_insert
GZ_DEVFN void rank_linear_insert(const RankArr& a, int last) {
  const float vk = a.k(last);
  const unsigned char vd = a.d(last);
  int next = last - 1;
  while (vk < a.k(next)) {
    a.move(last, next);
    last = next;
    --next;
  }
  a.set(last, vk, vd);
}

// std::__insertion_sort
GZ_DEVFN void rank_insertion_sort(const RankArr& a, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (a.k(i) < a.k(first)) {
      const float vk = a.k(i);
      const unsigned char vd = a.d(i);
      for (int j = i; j > first; --j) a.move(j, j - 1);   // move_backward(first, i, i + 1)
      a.set(first, vk, vd);
    } else {
      rank_linear_insert(a, i);
    }
  }
}

// std::sort(first, first + n) with comp = (key <)
GZ_DEVFN void rank_std_sort(const RankArr& a, int n) {
  if (n <= 0) return;
  // __introsort_loop; the reference recurses into the right part and loops on the left one,
  // the ranges are disjoint, so a stack of pending right parts gives the same result
  int st_first[20], st_last[20], st_depth[20];
  int sp = 0;
  int lg = 0;
  while ((n >> (lg + 1)) != 0) ++lg;   // std::__lg(n)
  st_first[0] = 0; st_last[0] = n; st_depth[0] = 2 * lg;
  sp = 1;
  while (sp > 0) {
    --sp;
    int first = st_first[sp], last = st_last[sp], depth = st_depth[sp];
    while (last - first > 16) {
      if (depth == 0) {
        rank_heap_sort(a, first, last);
        break;
      }
      --depth;
      const int cut = rank_partition_pivot(a, first, last);
      st_first[sp] = cut; st_last[sp] = last; st_depth[sp] = depth;
      ++sp;
      last = cut;
    }
  }
  // __final_insertion_sort
  if (n > 16) {
    rank_insertion_sort(a, 0, 16);
    for (int i = 16; i != n; ++i) rank_linear_insert(a, i);
  } else {
    rank_insertion_sort(a, 0, n);
  }
}

struct RankArgs {
  const int16_t* coeffs;   // candidate, dequantised (frame layout)
  const int16_t* orig;     // original coefficients (same layout)
  const float* csf;        // kOrderCsf[192]  (order.inc)
  const float* bias;       // kOrderBias[192]
  int nb;                  // blocks of the search grid
  int coff[3];             // component c's block of grid position b = coff[c] + b
  int comp_mask;           // components that take part (processor.cc:382-383)
  int new_model;
  int32_t* cnt;            // [nb]
  uint8_t* idx;            // [nb][192]
};

GZ_CONST unsigned char kRankOldCsf[64] = {
    10, 10, 20, 40, 60, 70, 80, 90, 10, 20, 30, 60, 70, 80, 90, 90,
    20, 30, 60, 70, 80, 90, 90, 90, 40, 60, 70, 80, 90, 90, 90, 90,
    60, 70, 80, 90, 90, 90, 90, 90, 70, 80, 90, 90, 90, 90, 90, 90,
    80, 90, 90, 90, 90, 90, 90, 90, 90, 90, 90, 90, 90, 90, 90, 90};
GZ_CONST unsigned char kRankZigZag[64] = {   // kJPEGZigZagOrder, jpeg_data.h:75-84
    0, 1, 5, 6, 14, 15, 27, 28, 2, 4, 7, 13, 16, 26, 29, 42,
    3, 8, 12, 17, 25, 30, 41, 43, 9, 11, 18, 24, 31, 40, 44, 53,
    10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60,
    21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

// score of coefficient i = ch*64 + k of a block (processor.cc:386-396)
GZ_DEVFN float rank_score(int i, int orig_val, int new_model, const float* csf, const float* bias) {
  const int av = orig_val < 0 ? -orig_val : orig_val;
  if (new_model) return (float)av * csf[i] + bias[i];
  const int ch = i >> 6, k = i & 63;
  const double w = ch == 0 ? 1.0 : (ch == 1 ? 0.22 : 0.20);
  return (float)(((double)av - (double)kRankZigZag[k] / 64.0) * w / (double)kRankOldCsf[k]);
}

__global__ __launch_bounds__(kRankLanes) void k_rank_candidates(RankArgs a) {
  __shared__ float s_key[kRankMax * kRankLanes];
  __shared__ unsigned char s_id[kRankMax * kRankLanes];
  const int t = threadIdx.x;
  const int b = blockIdx.x * kRankLanes + t;
  if (b >= a.nb) return;   // no barrier in this kernel: every lane works on its own column
  RankArr arr{s_key, s_id, t};
  int n = 0;
  for (int ch = 0; ch < 3; ++ch) {
    if (!((a.comp_mask >> ch) & 1)) continue;
    const int16_t* blk = a.coeffs + ((size_t)a.coff[ch] + b) * 64;
    const int16_t* ob = a.orig + ((size_t)a.coff[ch] + b) * 64;
    for (int k = 1; k < 64; ++k) {
      if (blk[k] == 0) continue;
      const int i = ch * 64 + k;
      arr.set(n++, rank_score(i, (int)ob[k], a.new_model, a.csf, a.bias), (unsigned char)i);
    }
  }
  rank_std_sort(arr, n);
  a.cnt[b] = n;
  for (int i = 0; i < n; ++i) a.idx[(size_t)b * 192 + i] = arr.d(i);
}

// Test hook (gz_probe_rank_sort): std::sort of caller-supplied keys, ids 0..n-1.
__global__ __launch_bounds__(kRankLanes) void k_probe_rank_sort(const float* keys, const int32_t* cnt,
                                                                int narr, uint8_t* perm) {
  __shared__ float s_key[kRankMax * kRankLanes];
  __shared__ unsigned char s_id[kRankMax * kRankLanes];
  const int t = threadIdx.x;
  const int b = blockIdx.x * kRankLanes + t;
  if (b >= narr) return;
  RankArr arr{s_key, s_id, t};
  const int n = cnt[b];
  for (int i = 0; i < n; ++i) arr.set(i, keys[(size_t)b * 192 + i], (unsigned char)i);
  rank_std_sort(arr, n);
  for (int i = 0; i < n; ++i) perm[(size_t)b * 192 + i] = arr.d(i);
}

}  // namespace gz
