// LazySorted (guetzli_amd/host/lazy_sort.h) must reproduce std::sort's permutation exactly,
// ties included.  Elements carry an id that the comparison ignores, so any deviation in the
// order of equal keys is visible.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <random>
#include <utility>
#include <vector>

#include "../../guetzli_amd/host/lazy_sort.h"

typedef std::pair<int, float> E;
struct Less {
#ifndef GZ_TEST_PLAIN_LESS
  enum { float_second_key = 1 };   // lazy_sort.h's AVX2 pass over the keys (-DGZ_TEST_PLAIN_LESS: the generic one)
#endif
  bool operator()(const E& a, const E& b) const { return a.second < b.second; }
};

static size_t g_par_threshold = 1 << 17;
static int g_ensure = 0;

static int check(std::vector<E> v, const char* what, size_t prefix_only = 0) {
  std::vector<E> ref = v;
  std::sort(ref.begin(), ref.end(), Less());
  guetzli_amd::LazySorted<E, Less> lazy(v.data(), v.size(), Less(), -1, g_par_threshold);
  const size_t upto = prefix_only ? std::min(prefix_only, v.size()) : v.size();
  size_t from = 0;
  if (g_ensure == 3) {
    // SelectPrefix: [0, f) is the right SET, element f-1 and everything read after it exact
    const size_t f = upto / 2;
    lazy.SelectPrefix(f);
    if (f > 0) {
      std::vector<std::pair<float, int> > a, b;
      for (size_t i = 0; i < f; ++i) {
        a.push_back(std::make_pair(v[i].second, v[i].first));
        b.push_back(std::make_pair(ref[i].second, ref[i].first));
      }
      std::sort(a.begin(), a.end());
      std::sort(b.begin(), b.end());
      if (a != b) { printf("FAIL %s n=%zu: SelectPrefix(%zu) set differs\n", what, v.size(), f); return 1; }
      from = f - 1;
    }
  } else if (g_ensure) {
    lazy.EnsureSorted(g_ensure == 1 ? upto : upto / 2);
  }
  for (size_t i = from; i < upto; ++i) {
    const E& e = lazy[i];
    if (e.first != ref[i].first || e.second != ref[i].second) {
      printf("FAIL %s n=%zu at %zu: lazy (%d,%g) std (%d,%g)\n", what, v.size(), i, e.first,
             e.second, ref[i].first, ref[i].second);
      return 1;
    }
  }
  return 0;
}

int run_all();
int main() {
  int fails = run_all();          // serial partitions below 128K elements, parallel above
  g_par_threshold = 24;           // parallel partition for (almost) every range
  fails += run_all();
  g_par_threshold = 1 << 17;
  g_ensure = 1;                   // pool-parallel EnsureSorted over the whole prefix
  fails += run_all();
  g_ensure = 2;                   // ... over half of it, the rest lazily
  fails += run_all();
  g_ensure = 3;                   // SelectPrefix over half of it (unordered set), the rest lazily
  fails += run_all();
  printf(fails ? "lazy_sort: %d FAILURES\n" : "lazy_sort: ok\n", fails);
  return fails ? 1 : 0;
}

int run_all() {
  std::mt19937 rng(12345);
  int fails = 0;
  const size_t sizes[] = {0, 1, 2, 3, 15, 16, 17, 18, 31, 32, 33, 100, 257, 1000, 4097, 65536, 300001, 2000003};
  for (size_t n : sizes) {
    for (int pattern = 0; pattern < 8; ++pattern) {
      std::vector<E> v(n);
      for (size_t i = 0; i < n; ++i) {
        float key;
        switch (pattern) {
          case 0: key = (float)(rng() % 1000003) / 7.0f; break;           // few ties
          case 1: key = (float)(rng() % 17); break;                       // heavy ties
          case 2: key = 1.0f; break;                                      // all equal
          case 3: key = (float)i; break;                                  // sorted
          case 4: key = (float)(n - i); break;                            // reversed
          case 5: key = (float)((i * 7919) % 101) * 0.25f; break;         // periodic ties
          case 6: key = (float)(i < n / 2 ? i : n - i); break;            // organ pipe
          default: key = (rng() % 4 == 0) ? 0.0f : ldexpf((float)(rng() % 1024), -(int)(rng() % 12));
        }
        v[i] = E((int)i, key);
      }
      fails += check(v, "full");
      if (n > 1000) fails += check(v, "prefix", n / 50 + 3);
    }
  }
  // median-of-3 killer (Musser): forces the introsort depth limit and the heap-sort fallback
  for (size_t n : {1024u, 65536u, 1u << 20}) {
    std::vector<E> v(n);
    const size_t k = n / 2;
    for (size_t i = 1; i <= k; ++i) {
      if (i % 2 == 1) {
        v[i - 1] = E((int)i, (float)i);
        v[i] = E((int)(i + 1), (float)(k + i));
      }
      v[k + i - 1] = E((int)(k + i), (float)(2 * i));
    }
    fails += check(v, "killer");
  }
  // forced depth limits: same algorithm as std::sort only for the natural limit, so here the
  // result is only required to be sorted and a permutation
  {
    std::vector<E> v(50000);
    for (size_t i = 0; i < v.size(); ++i) v[i] = E((int)i, (float)(rng() % 977));
    for (int depth : {0, 1, 3}) {
      std::vector<E> w = v;
      guetzli_amd::LazySorted<E, Less> lazy(w.data(), w.size(), Less(), depth);
      lazy.SortAll();
      long idsum = 0;
      for (size_t i = 0; i < w.size(); ++i) {
        idsum += w[i].first;
        if (i && w[i].second < w[i - 1].second) { printf("FAIL depth %d unsorted\n", depth); ++fails; break; }
      }
      if (idsum != (long)v.size() * ((long)v.size() - 1) / 2) { printf("FAIL depth %d not a permutation\n", depth); ++fails; }
    }
  }
  return fails;
}
