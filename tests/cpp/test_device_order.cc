// The device partition step of phase B (include/guetzli_amd.h: gz_order_upload /
// gz_order_partition / gz_order_fetch) driven by the product's LazySorted
// (guetzli_amd/host/lazy_sort.h) must reproduce std::sort's permutation exactly, ties
// included -- the check is against std::sort itself on the same input.
//
//   test_device_order <libguetzli_amd*.so> <max_n> [device_threshold ...]
//
// The library is dlopen-ed: the CPU suite passes the emulation build of the kernel sources
// (tests/emu), the GPU suite the gfx950 library.
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <random>
#include <utility>
#include <vector>

#include "../../guetzli_amd/host/lazy_sort.h"
#include "../../include/guetzli_amd.h"

typedef std::pair<int, float> E;
struct Less {
#ifndef GZ_TEST_PLAIN_LESS
  enum { float_second_key = 1 };   // lazy_sort.h's AVX2 pass over the keys (-DGZ_TEST_PLAIN_LESS: the generic one)
#endif
  bool operator()(const E& a, const E& b) const { return a.second < b.second; }
};

static decltype(&gz_create) p_create;
static decltype(&gz_destroy) p_destroy;
static decltype(&gz_order_upload) p_upload;
static decltype(&gz_order_partition) p_partition;
static decltype(&gz_order_fetch) p_fetch;
static decltype(&gz_order_descend) p_descend;
static decltype(&gz_last_error) p_last_error;
static int g_pattern = -1;
static gz_ctx* g_ctx;
static long g_partitions = 0, g_fetched = 0, g_replayed = 0;

struct Dev : guetzli_amd::RangeDevice {
  // partitions the device has already made (gz_order_descend), in the order it made them
  std::vector<uint64_t> log;
  size_t next = 0;
  bool Partition(size_t lo, size_t hi, size_t* cut) override {
    if (3 * next + 2 < log.size() && log[3 * next] == lo && log[3 * next + 1] == hi) {
      *cut = (size_t)log[3 * next + 2];
      ++next;
      ++g_replayed;
      return true;
    }
    uint64_t c = 0;
    if (p_partition(g_ctx, lo, hi, &c) != GZ_OK) return false;
    *cut = (size_t)c;
    ++g_partitions;
    return true;
  }
  bool Fetch(size_t lo, size_t hi, void* dst) override {
    g_fetched += (long)(hi - lo);
    return p_fetch(g_ctx, lo, hi, dst) == GZ_OK;
  }
};

static int check(const std::vector<E>& v, const char* what, size_t threshold, size_t prefix_only,
                 int ensure) {
  std::vector<E> ref = v;
  std::sort(ref.begin(), ref.end(), Less());
  if (p_upload(g_ctx, v.data(), v.size()) != GZ_OK) { printf("FAIL upload\n"); return 1; }
  std::vector<E> host(v.size(), E(-1, -1.0f));   // filled from the device range by range
  Dev dev;
  guetzli_amd::LazySorted<E, Less> lazy(host.data(), host.size(), Less(), -1, 1 << 17, &dev, threshold);
  const size_t upto = prefix_only ? std::min(prefix_only, v.size()) : v.size();
  size_t from = 0;
  if (ensure == 3 || ensure == 4) {   // SelectPrefix: the right set below f, exact from f-1 on
    const size_t f = upto / 2;
    if (ensure == 4 && f > 0) {
      // the descent towards position f - 1 made on the device in one go; LazySorted then
      // replays its log instead of asking for the partitions one by one
      dev.log.assign(3 * 12, 0);
      int levels = 0;
      if (p_descend(g_ctx, f - 1, threshold, 12, dev.log.data(), &levels) != GZ_OK) {
        printf("FAIL descend %s pattern %d n=%zu last=%zu threshold=%zu: %s\n", what, g_pattern, v.size(), f - 1, threshold,
               p_last_error ? p_last_error(g_ctx) : "");
        return 1;
      }
      dev.log.resize(3 * (size_t)levels);
    }
    lazy.SelectPrefix(f);
    if (ensure == 4 && dev.next != dev.log.size() / 3) {
      printf("FAIL %s n=%zu thr=%zu: %zu of %zu logged partitions replayed\n", what, v.size(), threshold, dev.next, dev.log.size() / 3);
      return 1;
    }
    if (f > 0) {
      std::vector<std::pair<float, int> > a, b;
      for (size_t i = 0; i < f; ++i) {
        a.push_back(std::make_pair(host[i].second, host[i].first));
        b.push_back(std::make_pair(ref[i].second, ref[i].first));
      }
      std::sort(a.begin(), a.end());
      std::sort(b.begin(), b.end());
      if (a != b) { printf("FAIL %s n=%zu thr=%zu: SelectPrefix(%zu) set differs\n", what, v.size(), threshold, f); return 1; }
      from = f - 1;
    }
  } else if (ensure) {
    lazy.EnsureSorted(ensure == 1 ? upto : upto / 2);
  }
  for (size_t i = from; i < upto; ++i) {
    const E& e = lazy[i];
    if (lazy.failed()) { printf("FAIL %s: device call failed\n", what); return 1; }
    if (e.first != ref[i].first || e.second != ref[i].second) {
      printf("FAIL %s n=%zu thr=%zu at %zu: device (%d,%g) std (%d,%g)\n", what, v.size(), threshold,
             i, e.first, e.second, ref[i].first, ref[i].second);
      return 1;
    }
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 3) { printf("usage: %s lib max_n [thresholds]\n", argv[0]); return 2; }
  void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) { printf("dlopen: %s\n", dlerror()); return 2; }
  p_create = (decltype(p_create))dlsym(h, "gz_create");
  p_destroy = (decltype(p_destroy))dlsym(h, "gz_destroy");
  p_upload = (decltype(p_upload))dlsym(h, "gz_order_upload");
  p_partition = (decltype(p_partition))dlsym(h, "gz_order_partition");
  p_fetch = (decltype(p_fetch))dlsym(h, "gz_order_fetch");
  p_descend = (decltype(p_descend))dlsym(h, "gz_order_descend");
  p_last_error = (decltype(p_last_error))dlsym(h, "gz_last_error");
  if (!p_create || !p_destroy || !p_upload || !p_partition || !p_fetch || !p_descend) { printf("missing symbol\n"); return 2; }
  const size_t max_n = (size_t)atol(argv[2]);
  std::vector<size_t> thresholds;
  for (int i = 3; i < argc; ++i) thresholds.push_back((size_t)atol(argv[i]));
  if (thresholds.empty()) thresholds = {16, 4096};
  std::vector<uint8_t> rgb(8 * 8 * 3, 128);
  int err = 0;
  g_ctx = p_create(0, 8, 8, rgb.data(), 1.0f, &err);
  if (!g_ctx) { printf("gz_create failed: %d\n", err); return 2; }

  std::mt19937 rng(4242);
  int fails = 0;
  const size_t sizes[] = {4, 5, 17, 18, 33, 100, 257, 1000, 2047, 2048, 2049, 4097, 20000,
                          32769 /* 16 chunks: the table's entry 16 is the next thread's */, 65536, 300001, 2000003,
                          5000011, 8388609 /* 4096 chunks: the descent's tables are full */,
                          12000017 /* beyond them: k_desc_swap's instantiation with the larger tables */};
  for (size_t thr : thresholds) {
    for (size_t n : sizes) {
      if (n > max_n || n > 4000 * thr) continue;   // keep the number of device calls bounded
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
          v[i] = E((int)(i % 977), key);   // block ids repeat, like the real order
        }
        // make ids distinguishable among equal keys
        for (size_t i = 0; i < n; ++i) v[i].first = (int)i;
        // (GZ_TEST_ONLY_N=<n>: the checks of that size only; the data of the others is still drawn)
        static const size_t only_n = getenv("GZ_TEST_ONLY_N") ? (size_t)atol(getenv("GZ_TEST_ONLY_N")) : 0;
        if (only_n && n != only_n) continue;
        g_pattern = pattern;
        fails += check(v, "full", thr, 0, 0);
        if (n > 1000) {
          fails += check(v, "prefix", thr, n / 50 + 3, 0);
          fails += check(v, "ensure", thr, n / 20 + 3, 1);
          fails += check(v, "ensure-half", thr, n / 20 + 3, 2);
          fails += check(v, "select-prefix", thr, n / 10 + 3, 3);
          fails += check(v, "descend", thr, n / 10 + 3, 4);
          fails += check(v, "descend-far", thr, n - n / 7, 4);
        }
        if (fails > 5) goto done;
      }
    }
  }
  // median-of-3 killer: deep recursion, depth limit and heap-sort fallback on fetched ranges
  for (size_t n : {1024u, 65536u}) {
    if (n > max_n) continue;
    std::vector<E> v(n);
    const size_t k = n / 2;
    for (size_t i = 1; i <= k; ++i) {
      if (i % 2 == 1) {
        v[i - 1] = E((int)i, (float)i);
        v[i] = E((int)(i + 1), (float)(k + i));
      }
      v[k + i - 1] = E((int)(k + i), (float)(2 * i));
    }
    fails += check(v, "killer", 16, 0, 0);
  }
done:
  p_destroy(g_ctx);
  printf("device partitions %ld (+ %ld replayed from descents), entries fetched %ld\n", g_partitions, g_replayed, g_fetched);
  printf(fails ? "device_order: %d FAILURES\n" : "device_order: ok\n", fails);
  return fails ? 1 : 0;
}
