// The device-side ranking of zeroing candidates (guetzli_amd/csrc/gz_kernels_rank.h) must
// produce libstdc++'s std::sort permutation, ties and the heap-sort fall-back included.
//
//   test_rank_sort [libguetzli_amd*.so]
//
// Part 1 compiles the restatement for the host (GZ_EMU) and checks it against std::sort
// itself on random, tie-heavy, presorted and adversarial inputs (McIlroy's "killer adversary"
// run against std::sort drives introsort into its depth limit).  Part 2 sends the same arrays
// through gz_probe_rank_sort of the given library (emulation build on CPU, gfx950 on a GPU).
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <random>
#include <utility>
#include <vector>

static long g_rank_heap_calls = 0;
#define GZ_RANK_COUNT_HEAP 1
#include "../../guetzli_amd/csrc/gz_kernels_rank.h"
#include "../../include/guetzli_amd.h"

typedef std::pair<int, float> E;

static std::vector<float> killer(int n) {   // M. D. McIlroy, "A Killer Adversary for Quicksort"
  std::vector<int> val(n), ptr(n);
  const int gas = n - 1;
  int nsolid = 0, candidate = 0;
  for (int i = 0; i < n; ++i) { ptr[i] = i; val[i] = gas; }
  std::sort(ptr.begin(), ptr.end(), [&](int x, int y) {
    if (val[x] == gas && val[y] == gas) {
      if (x == candidate) val[x] = nsolid++; else val[y] = nsolid++;
    }
    if (val[x] == gas) candidate = x; else if (val[y] == gas) candidate = y;
    return val[x] < val[y];
  });
  std::vector<float> k(n);
  for (int i = 0; i < n; ++i) k[i] = (float)val[i];
  return k;
}

int main(int argc, char** argv) {
  std::mt19937 rng(12345);
  std::vector<std::vector<float> > arrays;
  for (int n = 0; n <= 192; ++n) {
    for (int variant = 0; variant < 6; ++variant) {
      std::vector<float> k(n);
      for (int i = 0; i < n; ++i) {
        switch (variant) {
          case 0: k[i] = (float)(rng() % 1000003) * 0.001f; break;       // mostly distinct
          case 1: k[i] = (float)(rng() % 7); break;                      // heavy ties
          case 2: k[i] = (float)i; break;                                // sorted
          case 3: k[i] = (float)(n - i); break;                          // reversed
          case 4: k[i] = (float)(i < n / 2 ? i : n - i); break;          // organ pipe
          default: k[i] = (float)((rng() % 3) * 100 + (i % 5)); break;   // clustered ties
        }
      }
      arrays.push_back(k);
    }
    if (n >= 17) {
      arrays.push_back(killer(n));
      std::vector<float> q = killer(n);
      for (float& v : q) v = (float)((int)v / 3);   // the adversary's shape, with ties
      arrays.push_back(q);
    }
  }
  const int narr = (int)arrays.size();
  std::vector<std::vector<uint8_t> > want(narr);
  for (int a = 0; a < narr; ++a) {
    std::vector<E> v;
    for (size_t i = 0; i < arrays[a].size(); ++i) v.push_back(E((int)i, arrays[a][i]));
    std::sort(v.begin(), v.end(), [](const E& x, const E& y) { return x.second < y.second; });
    for (const E& e : v) want[a].push_back((uint8_t)e.first);
  }
  // part 1: the restatement compiled for the host
  {
    std::vector<float> key(gz::kRankMax * gz::kRankLanes);
    std::vector<unsigned char> id(gz::kRankMax * gz::kRankLanes);
    for (int a = 0; a < narr; ++a) {
      gz::RankArr arr{key.data(), id.data(), a % gz::kRankLanes};
      const int n = (int)arrays[a].size();
      for (int i = 0; i < n; ++i) arr.set(i, arrays[a][i], (unsigned char)i);
      gz::rank_std_sort(arr, n);
      for (int i = 0; i < n; ++i)
        if (arr.d(i) != want[a][i]) {
          fprintf(stderr, "restatement: array %d (n=%d) differs from std::sort at %d\n", a, n, i);
          return 1;
        }
    }
    if (g_rank_heap_calls == 0) {
      fprintf(stderr, "the heap-sort fall-back was never reached: the adversarial inputs are too weak\n");
      return 1;
    }
    printf("restatement == std::sort on %d arrays (%ld heap-sort fall-backs)\n", narr, g_rank_heap_calls);
  }
  if (argc < 2) return 0;
  // part 2: the library's kernel
  void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!lib) { fprintf(stderr, "dlopen %s: %s\n", argv[1], dlerror()); return 2; }
  auto probe = (decltype(&gz_probe_rank_sort))dlsym(lib, "gz_probe_rank_sort");
  if (!probe) { fprintf(stderr, "gz_probe_rank_sort missing\n"); return 2; }
  std::vector<float> keys((size_t)narr * 192, 0.0f);
  std::vector<int32_t> cnt(narr);
  std::vector<uint8_t> perm((size_t)narr * 192, 0);
  for (int a = 0; a < narr; ++a) {
    cnt[a] = (int32_t)arrays[a].size();
    std::copy(arrays[a].begin(), arrays[a].end(), keys.begin() + (size_t)a * 192);
  }
  const int rc = probe(0, keys.data(), cnt.data(), narr, perm.data());
  if (rc != GZ_OK) { fprintf(stderr, "gz_probe_rank_sort: %d\n", rc); return 2; }
  for (int a = 0; a < narr; ++a)
    for (int i = 0; i < cnt[a]; ++i)
      if (perm[(size_t)a * 192 + i] != want[a][i]) {
        fprintf(stderr, "device: array %d (n=%d) differs from std::sort at %d\n", a, cnt[a], i);
        return 1;
      }
  printf("device == std::sort on %d arrays\n", narr);
  return 0;
}
