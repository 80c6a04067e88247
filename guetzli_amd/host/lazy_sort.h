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
#pragma once
#include <algorithm>
#include <cstddef>
#include <vector>

namespace guetzli_amd {

template <class T, class Less>
class LazySorted {
 public:
  LazySorted(T* data, size_t n, Less less, int depth_override = -1)
      : a_(data), n_(n), less_(less), done_(0) {
    if (n_ == 0) return;
    int lg = 0;
    for (size_t m = n_; m > 1; m >>= 1) ++lg;
    pending_.push_back(Range{0, n_, depth_override >= 0 ? depth_override : 2 * lg});
  }
  size_t size() const { return n_; }
  // Finalises positions [0, i] and returns element i.
  const T& operator[](size_t i) {
    while (done_ <= i) Refine();
    return a_[i];
  }
  void SortAll() {
    while (done_ < n_) Refine();
  }

 private:
  struct Range {
    size_t lo, hi;
    int depth;
  };

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

  size_t Partition(size_t first, size_t last, size_t pivot) {
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
    Range r = pending_.back();
    pending_.pop_back();
    if (r.hi - r.lo <= 16) {
      InsertionSort(r.lo, r.hi);
      done_ = r.hi;
      return;
    }
    if (r.depth == 0) {
      std::partial_sort(a_ + r.lo, a_ + r.hi, a_ + r.hi, less_);   // the heap-sort fallback
      done_ = r.hi;
      return;
    }
    const size_t mid = r.lo + (r.hi - r.lo) / 2;
    MoveMedianToFirst(r.lo, r.lo + 1, mid, r.hi - 1);
    const size_t cut = Partition(r.lo + 1, r.hi, r.lo);
    pending_.push_back(Range{cut, r.hi, r.depth - 1});
    pending_.push_back(Range{r.lo, cut, r.depth - 1});
  }

  T* a_;
  size_t n_;
  Less less_;
  size_t done_;
  std::vector<Range> pending_;   // back() is the leftmost unsorted range
};

}  // namespace guetzli_amd
