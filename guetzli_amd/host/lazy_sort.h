// std::sort, evaluated lazily from the front.
//
// Phase B of SelectFrequencyMasking (processor.cc:675-678) std::sort-s up to millions of
// (block, key) entries per iteration and then consumes only a short prefix before its
// size/count criterion stops the scan.  Keys tie exactly across different blocks (repeated
// image content), std::sort is not stable, and which tied block is served first feeds the
// JPEG bytes -- so the consumer must see exactly the permutation libstdc++'s introsort
// produces.  LazySorted does that without sorting the tail:
//
// libstdc++'s std::sort = introsort loop (median-of-3 of {first+1, mid, last-1} moved to
// `first`, unguarded Hoare partition, recursion on the right part, iteration on the left,
// depth limit 2*floor(lg n) with a heap-sort fallback) followed by one insertion-sort pass
// over the whole array.  Partitioning a range never touches elements outside it, and after
// a partition every element on the left is <= every element on the right, so (a) the
// order in which pending ranges are refined does not change the result and (b) the final
// insertion sort never moves an element across a range boundary: it is an independent,
// plain insertion sort of each final range of <= 16 elements.  Hence the front of the
// array can be finished first and the remaining ranges kept on a stack until somebody
// asks for them.  Element i of the result is identical to element i after std::sort on
// the same input (tests/test_host_logic.py checks this against std::sort itself,
// including tie-heavy and depth-limit inputs).
//
// Large ranges are partitioned in parallel with the same outcome: libstdc++'s unguarded
// Hoare partition swaps the k-th element from the left that is not less than the pivot with
// the k-th element from the right that is not greater, for as long as the former lies left
// of the latter, so the final arrangement and the cut follow from the two ordered lists of
// such positions (built by chunks), without replaying the scan.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <vector>

#include "parallel.h"

#if defined(__x86_64__) && defined(__GNUC__)
#include <immintrin.h>
#define GZ_LAZY_SORT_AVX2 1
#endif

namespace guetzli_amd {

// A comparator that orders 8-byte {int32, float} entries by the float (phase B's order) says so
// with a member constant `enum { float_second_key = 1 };`: the partitions' pass over the elements
// then runs eight entries at a time where the CPU has AVX2 (same lists, same arrangement).
template <class L, class = void>
struct HasFloatSecondKey { static constexpr bool value = false; };
template <class L>
struct HasFloatSecondKey<L, decltype((void)L::float_second_key)> { static constexpr bool value = L::float_second_key != 0; };

#ifdef GZ_LAZY_SORT_AVX2
namespace detail {
struct CompressTable {
  alignas(32) uint32_t perm[256][8];
  CompressTable() {
    for (int m = 0; m < 256; ++m) {
      int k = 0;
      for (int j = 0; j < 8; ++j)
        if (m & (1 << j)) perm[m][k++] = (uint32_t)j;
      for (; k < 8; ++k) perm[m][k] = 0;
    }
  }
};
inline const CompressTable& compress_table() {
  static const CompressTable t;
  return t;
}
inline bool cpu_has_avx2() {
  static const bool have = __builtin_cpu_supports("avx2");
  return have;
}
// Positions i (0 .. n-1) of the entries e with !(e.key < pv) -> lb, with !(pv < e.key) -> rb, each
// in ascending order; entries are 8 bytes {int32, float key}.  lb / rb have room for n + 8.
__attribute__((target("avx2"))) inline void StopperListsAvx2(const void* entries, size_t n, float pv,
                                                              uint32_t* lb, size_t* nl_out,
                                                              uint32_t* rb, size_t* nr_out) {
  const CompressTable& t = compress_table();
  const float* f = static_cast<const float*>(entries);
  const __m256 pvv = _mm256_set1_ps(pv);
  __m256i idx = _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7);
  const __m256i eight = _mm256_set1_epi32(8);
  size_t nl = 0, nr = 0, i = 0;
  for (; i + 8 <= n; i += 8) {
    const __m256 v0 = _mm256_loadu_ps(f + 2 * i);       // i0 k0 i1 k1 | i2 k2 i3 k3
    const __m256 v1 = _mm256_loadu_ps(f + 2 * i + 8);   // i4 k4 i5 k5 | i6 k6 i7 k7
    const __m256 sh = _mm256_shuffle_ps(v0, v1, _MM_SHUFFLE(3, 1, 3, 1));   // k0 k1 k4 k5 | k2 k3 k6 k7
    const __m256 keys = _mm256_castpd_ps(_mm256_permute4x64_pd(_mm256_castps_pd(sh), 0xD8));   // k0 .. k7
    const int ml = _mm256_movemask_ps(_mm256_cmp_ps(keys, pvv, _CMP_NLT_UQ));   // !(key < pv)
    const int mr = _mm256_movemask_ps(_mm256_cmp_ps(pvv, keys, _CMP_NLT_UQ));   // !(pv < key)
    const __m256i pl = _mm256_load_si256(reinterpret_cast<const __m256i*>(t.perm[ml]));
    const __m256i pr = _mm256_load_si256(reinterpret_cast<const __m256i*>(t.perm[mr]));
    _mm256_storeu_si256(reinterpret_cast<__m256i*>(lb + nl), _mm256_permutevar8x32_epi32(idx, pl));
    _mm256_storeu_si256(reinterpret_cast<__m256i*>(rb + nr), _mm256_permutevar8x32_epi32(idx, pr));
    nl += (size_t)__builtin_popcount((unsigned)ml);
    nr += (size_t)__builtin_popcount((unsigned)mr);
    idx = _mm256_add_epi32(idx, eight);
  }
  for (; i < n; ++i) {
    const float k = f[2 * i + 1];
    lb[nl] = (uint32_t)i;
    nl += !(k < pv);
    rb[nr] = (uint32_t)i;
    nr += !(pv < k);
  }
  *nl_out = nl;
  *nr_out = nr;
}
}  // namespace detail
#endif

// Optional back end for an array that starts out on the device (include/guetzli_amd.h,
// gz_order_*): ranges are partitioned there -- same arrangement and cut as Partition()
// below -- until they are small, then fetched into the host array and finished here.
class RangeDevice {
 public:
  virtual ~RangeDevice() {}
  // libstdc++'s __unguarded_partition_pivot on [lo, hi): median of {lo+1, mid, hi-1} to lo,
  // unguarded partition of [lo+1, hi) around it.
  virtual bool Partition(size_t lo, size_t hi, size_t* cut) = 0;
  virtual bool Fetch(size_t lo, size_t hi, void* dst) = 0;
};

template <class T, class Less>
class LazySorted {
 public:
  LazySorted(T* data, size_t n, Less less, int depth_override = -1,
             size_t parallel_threshold = 1 << 17, RangeDevice* device = nullptr,
             size_t device_threshold = 1 << 16)
      : a_(data), n_(n), less_(less), done_(0), par_threshold_(parallel_threshold),
        dev_(device), dev_threshold_(device_threshold < 16 ? 16 : device_threshold) {
    if (n_ == 0) return;
    int lg = 0;
    for (size_t m = n_; m > 1; m >>= 1) ++lg;
    pending_.push_back(Range{0, n_, depth_override >= 0 ? depth_override : 2 * lg, dev_ != nullptr});
  }
  size_t size() const { return n_; }
  bool failed() const { return failed_; }   // a device call failed; contents are undefined
  // Finalises positions [0, i] and returns element i.
  const T& operator[](size_t i) {
    while (done_ <= i) Refine();
    return a_[i];
  }
  void SortAll() {
    while (done_ < n_) Refine();
  }
  // Makes positions [0, upto) hold exactly the elements std::sort would place there, WITHOUT
  // ordering them (quickselect along introsort's own partitions), except that position
  // upto - 1 and everything after it up to the next range boundary is final and sorted.
  // Ranges that end at or below upto - 1 are dropped from the work list unsorted: a consumer
  // that only needs the SET of the first `upto` elements and element upto - 1 itself (phase
  // B's fast steps) saves the O(upto log upto) sort.  Afterwards operator[] is valid for
  // i >= upto - 1 only.
  void SelectPrefix(size_t upto) {
    if (upto == 0 || n_ == 0) return;
    if (upto > n_) upto = n_;
    const size_t last = upto - 1;
    while (!pending_.empty() && pending_.back().lo <= last) {
      const Range r = pending_.back();
      if (r.hi <= last) {            // entirely below: only its content matters
        pending_.pop_back();
        if (r.dev && !(dev_->Fetch(r.lo, r.hi, a_ + r.lo))) {
          failed_ = true;
          for (size_t i = r.lo; i < r.hi; ++i) a_[i] = T();
        }
        done_ = std::max(done_, r.hi);
        continue;
      }
      const size_t fin = RefineOn(&pending_, true);   // the range holding `last`: one step
      if (fin) done_ = std::max(done_, fin);
    }
  }

  // Finalises positions [0, upto) using the worker pool: the pending ranges that reach into
  // the prefix are split (in parallel) down to a grain, then finished independently -- the
  // ranges are disjoint and refining one never touches another, so the result is the one
  // the serial order of refinement gives.
  void EnsureSorted(size_t upto) {
    if (upto > n_) upto = n_;
    if (dev_) ResolveDevice(upto);
    WorkerPool& pool = WorkerPool::Get();
    if (upto <= done_ || pool.size() == 1 || upto - done_ < (1u << 16)) {
      while (done_ < upto) Refine();
      return;
    }
    std::vector<Range> work;
    while (!pending_.empty() && pending_.back().lo < upto) {
      work.push_back(pending_.back());
      pending_.pop_back();
    }
    const size_t grain = std::max<size_t>(8192, (upto - done_) / (8 * (size_t)pool.size()));
    // Right-hand pieces that start at or beyond `upto` are not needed yet: they go back to
    // the pending stack (they lie between the needed ranges and what is still pending).
    std::vector<Range> defer;
    for (size_t i = 0; i < work.size();) {
      const Range r = work[i];
      if (r.hi - r.lo > grain && r.hi - r.lo > 16 && r.depth > 0) {
        const size_t mid = r.lo + (r.hi - r.lo) / 2;
        MoveMedianToFirst(r.lo, r.lo + 1, mid, r.hi - 1);
        const size_t cut = Partition(r.lo + 1, r.hi, r.lo);
        work[i] = Range{r.lo, cut, r.depth - 1, false};
        if (cut < upto) work.push_back(Range{cut, r.hi, r.depth - 1, false});
        else defer.push_back(Range{cut, r.hi, r.depth - 1, false});
      } else {
        ++i;
      }
    }
    std::sort(defer.begin(), defer.end(), [](const Range& x, const Range& y) { return x.lo > y.lo; });
    for (const Range& r : defer) pending_.push_back(r);
    pool.Run((int)work.size(), [&](int w) {
      std::vector<Range> stack(1, work[w]);
      while (!stack.empty()) RefineOn(&stack, false);
    });
    for (const Range& r : work) done_ = std::max(done_, r.hi);
  }

 private:
  struct Range {
    size_t lo, hi;
    int depth;
    bool dev;   // still on the device (not yet in a_)
  };

  // One step on a device-resident range: partition it there while it is large, else bring
  // it to the host.  Pushes the result(s) on `stack`, leftmost on top.
  void RefineDevice(const Range& r, std::vector<Range>* stack) {
    if (r.hi - r.lo > dev_threshold_ && r.depth > 0) {
      size_t cut = 0;
      if (dev_->Partition(r.lo, r.hi, &cut) && cut > r.lo && cut <= r.hi) {
        stack->push_back(Range{cut, r.hi, r.depth - 1, true});
        stack->push_back(Range{r.lo, cut, r.depth - 1, true});
        return;
      }
      failed_ = true;
    }
    if (failed_ || !dev_->Fetch(r.lo, r.hi, a_ + r.lo)) {
      failed_ = true;
      for (size_t i = r.lo; i < r.hi; ++i) a_[i] = T();
    }
    stack->push_back(Range{r.lo, r.hi, r.depth, false});
  }

  // Brings every range that reaches into [0, upto) to the host (on the calling thread: the
  // device context is not shared with the pool).
  void ResolveDevice(size_t upto) {
    std::vector<Range> host;   // ascending lo
    while (!pending_.empty() && pending_.back().lo < upto) {
      const Range r = pending_.back();
      pending_.pop_back();
      if (r.dev) RefineDevice(r, &pending_);
      else host.push_back(r);
    }
    for (size_t i = host.size(); i-- > 0;) pending_.push_back(host[i]);
  }

  void MoveMedianToFirst(size_t result, size_t a, size_t b, size_t c) {
    if (less_(a_[a], a_[b])) {
      if (less_(a_[b], a_[c])) std::swap(a_[result], a_[b]);
      else if (less_(a_[a], a_[c])) std::swap(a_[result], a_[c]);
      else std::swap(a_[result], a_[a]);
    } else if (less_(a_[a], a_[c])) {
      std::swap(a_[result], a_[a]);
    } else if (less_(a_[b], a_[c])) {
      std::swap(a_[result], a_[c]);
    } else {
      std::swap(a_[result], a_[b]);
    }
  }

  // Same result as Partition() below, computed by chunks on the worker pool: one pass
  // records, per chunk, the positions not less than the pivot ("left stoppers") and not
  // greater than it ("right stoppers"); pair k = (k-th left stopper from the left, k-th
  // right stopper from the right) is swapped while the former lies left of the latter.
  size_t ParallelPartition(size_t first, size_t last, size_t pivot) {
    WorkerPool& pool = WorkerPool::Get();
    const T pv = a_[pivot];
    const size_t n = last - first;
    const int chunks = std::max(1, std::min<int>(4 * pool.size(), (int)(n / 8192)));
    const size_t per = (n + chunks - 1) / chunks;
    if (lbuf_.size() < n) {
      lbuf_.resize(n);
      rbuf_.resize(n);
    }
    std::vector<size_t> cnt_l(chunks + 1, 0), cnt_r(chunks + 1, 0);
    pool.Run(chunks, [&](int c) {
      const size_t lo = c * per, hi = std::min(n, lo + per);
      uint32_t* lp = &lbuf_[lo];
      uint32_t* rp = &rbuf_[lo];
      size_t nl = 0, nr = 0;
      for (size_t i = lo; i < hi; ++i) {
        const T& e = a_[first + i];
        lp[nl] = (uint32_t)i;
        nl += !less_(e, pv);
        rp[nr] = (uint32_t)i;
        nr += !less_(pv, e);
      }
      cnt_l[c + 1] = nl;
      cnt_r[c + 1] = nr;
    });
    for (int c = 0; c < chunks; ++c) {
      cnt_l[c + 1] += cnt_l[c];
      cnt_r[c + 1] += cnt_r[c];
    }
    const size_t nl = cnt_l[chunks], nr = cnt_r[chunks];
    // k-th (0-based) left stopper in ascending order / right stopper in ascending order
    auto left_at = [&](size_t k) {
      const int c = (int)(std::upper_bound(cnt_l.begin(), cnt_l.end(), k) - cnt_l.begin()) - 1;
      return (size_t)lbuf_[c * per + (k - cnt_l[c])];
    };
    auto right_at = [&](size_t k) {
      const int c = (int)(std::upper_bound(cnt_r.begin(), cnt_r.end(), k) - cnt_r.begin()) - 1;
      return (size_t)rbuf_[c * per + (k - cnt_r[c])];
    };
    // m = number of swapped pairs = max k with l_k < r_k (monotone in k)
    size_t lo_k = 0, hi_k = std::min(nl, nr);
    while (lo_k < hi_k) {
      const size_t k = (lo_k + hi_k + 1) / 2;
      if (left_at(k - 1) < right_at(nr - k)) lo_k = k; else hi_k = k - 1;
    }
    const size_t m = lo_k;
    const int schunks = m ? std::max(1, std::min<int>(4 * pool.size(), (int)(m / 4096))) : 0;
    const size_t sper = schunks ? (m + schunks - 1) / schunks : 0;
    pool.Run(schunks, [&](int sc) {
      const size_t k0 = sc * sper, k1 = std::min(m, k0 + sper);
      if (k0 >= k1) return;
      // cursors into the chunked lists: left ascending from k0, right descending from nr-1-k0
      int cl = (int)(std::upper_bound(cnt_l.begin(), cnt_l.end(), k0) - cnt_l.begin()) - 1;
      size_t il = k0 - cnt_l[cl];
      size_t rk = nr - 1 - k0;
      int cr = (int)(std::upper_bound(cnt_r.begin(), cnt_r.end(), rk) - cnt_r.begin()) - 1;
      size_t ir = rk - cnt_r[cr];
      for (size_t k = k0; k < k1; ++k) {
        while (il >= cnt_l[cl + 1] - cnt_l[cl]) { ++cl; il = 0; }
        std::swap(a_[first + lbuf_[cl * per + il]], a_[first + rbuf_[cr * per + ir]]);
        ++il;
        if (k + 1 < k1) {
          while (ir == 0) { --cr; ir = cnt_r[cr + 1] - cnt_r[cr]; }
          --ir;
        }
      }
    });
    // the scan stops at the next untouched left stopper or at the last swapped right one
    size_t cut = last;
    if (m < nl) cut = std::min(cut, first + left_at(m));
    if (m >= 1) cut = std::min(cut, first + right_at(nr - m));
    return cut;
  }

  size_t Partition(size_t first, size_t last, size_t pivot) {
    if (last - first >= par_threshold_ && last - first < (size_t)UINT32_MAX &&
        (WorkerPool::Get().size() > 1 || par_threshold_ < 1024))
      return ParallelPartition(first, last, pivot);
    return SerialPartition(first, last, pivot);
  }

  // std::__unguarded_partition.  Ranges of a few hundred elements and more take the same
  // two-list formulation as ParallelPartition on one thread: the branchy scan mispredicts on
  // every other element (4.2 ns per element measured), the branch-free compaction of stopper
  // positions plus the pairwise swaps takes 2.0 -- same arrangement, same cut.
  size_t SerialPartition(size_t first, size_t last, size_t pivot) {
    const size_t n = last - first;
    if (n >= 512 && n < (size_t)UINT32_MAX) {
      static thread_local std::vector<uint32_t> lb, rb;
      if (lb.size() < n + 8) {
        lb.resize(n + 8);
        rb.resize(n + 8);
      }
      const T pv = a_[pivot];
      size_t nl = 0, nr = 0;
      bool listed = false;
#ifdef GZ_LAZY_SORT_AVX2
      if constexpr (HasFloatSecondKey<Less>::value && sizeof(T) == 8) {
        if (detail::cpu_has_avx2()) {
          detail::StopperListsAvx2(a_ + first, n, pv.second, lb.data(), &nl, rb.data(), &nr);
          listed = true;
        }
      }
#endif
      if (!listed)
        for (size_t i = 0; i < n; ++i) {
          const T& e = a_[first + i];
          lb[nl] = (uint32_t)i;
          nl += !less_(e, pv);
          rb[nr] = (uint32_t)i;
          nr += !less_(pv, e);
        }
      size_t m = 0;   // pairs (m-th left stopper, m-th right stopper from the right) that swap
      while (m < nl && m < nr && lb[m] < rb[nr - 1 - m]) {
        std::swap(a_[first + lb[m]], a_[first + rb[nr - 1 - m]]);
        ++m;
      }
      size_t cut = last;
      if (m < nl) cut = std::min(cut, first + lb[m]);
      if (m >= 1) cut = std::min(cut, first + rb[nr - m]);
      return cut;
    }
    for (;;) {
      while (less_(a_[first], a_[pivot])) ++first;
      --last;
      while (less_(a_[pivot], a_[last])) --last;
      if (!(first < last)) return first;
      std::swap(a_[first], a_[last]);
      ++first;
    }
  }

  void InsertionSort(size_t lo, size_t hi) {
    for (size_t i = lo + 1; i < hi; ++i) {
      T val = a_[i];
      size_t j = i;
      while (j > lo && less_(val, a_[j - 1])) {
        a_[j] = a_[j - 1];
        --j;
      }
      a_[j] = val;
    }
  }

  // Takes the leftmost pending range one step further.
  void Refine() {
    const size_t fin = RefineOn(&pending_, true);
    if (fin) done_ = fin;
  }
  // One refinement step on the top of `stack`; returns the end of the range if it became
  // final, else 0.
  size_t RefineOn(std::vector<Range>* stack, bool may_use_pool) {
    Range r = stack->back();
    stack->pop_back();
    if (r.dev) {
      RefineDevice(r, stack);
      return 0;
    }
    if (r.hi - r.lo <= 16) {
      InsertionSort(r.lo, r.hi);
      return r.hi;
    }
    if (r.depth == 0) {
      std::partial_sort(a_ + r.lo, a_ + r.hi, a_ + r.hi, less_);   // the heap-sort fallback
      return r.hi;
    }
    const size_t mid = r.lo + (r.hi - r.lo) / 2;
    MoveMedianToFirst(r.lo, r.lo + 1, mid, r.hi - 1);
    const size_t cut = may_use_pool ? Partition(r.lo + 1, r.hi, r.lo)
                                    : SerialPartition(r.lo + 1, r.hi, r.lo);
    stack->push_back(Range{cut, r.hi, r.depth - 1, false});
    stack->push_back(Range{r.lo, cut, r.depth - 1, false});
    return 0;
  }

  T* a_;
  size_t n_;
  Less less_;
  size_t done_;
  size_t par_threshold_;
  RangeDevice* dev_;
  size_t dev_threshold_;
  bool failed_ = false;
  std::vector<uint32_t> lbuf_, rbuf_;   // per-chunk stopper positions (ParallelPartition)
  std::vector<Range> pending_;   // back() is the leftmost unsorted range
};

}  // namespace guetzli_amd
