// TEST INFRASTRUCTURE.  Exhaustive / randomised proof that the cheaper division sequences of
// gz_math.h (opsin sensitivity, butteraugli.h:548-615 / butteraugli.cc:351-353) give the bits
// of the reference's plain C++ divisions:
//
//   1. x01 = (v - kMin) / (kMax - kMin), v = double(float p): division by a CONSTANT, done as
//        q = a * y;  r = fma(-b, q, a);  q' = fma(r, y, q)     with y = RN(1 / b)
//      checked here for EVERY finite float p (2^32 inputs) against a / b.
//   2. s = float(G / double(p)) with G = double(float g): the double quotient of two
//      float-valued numbers rounded to float equals the float quotient g / p (a format of
//      >= 2*24 + 2 digits makes the second rounding innocuous); checked on random and
//      adversarial pairs.
// Build: g++ -O2 -mfma -ffp-contract=off -pthread verify_opsin_divisions.cc -o verify && ./verify
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <random>
#include <thread>
#include <vector>

static const double kMin = 0.971783, kMax = 590.188894;

int main(int argc, char** argv) {
  const double b = kMax - kMin;
  const double y = 1.0 / b;
  const unsigned nthreads = std::max(1u, std::thread::hardware_concurrency());
  const uint64_t limit = argc > 1 ? strtoull(argv[1], nullptr, 10) : (1ull << 32);   // inputs of check 1
  std::atomic<uint64_t> bad1(0), bad2(0);
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nthreads; ++t)
    th.emplace_back([&, t]() {
      uint64_t bad = 0;
      for (uint64_t u = t; u < limit; u += nthreads) {
        const uint32_t bits = (uint32_t)(u * 2654435761ull);   // a permutation of 2^32 when limit == 2^32
        float p;
        memcpy(&p, &bits, 4);
        if (!(fabsf(p) <= 3.0e38f)) continue;   // NaN / inf
        const double a = (double)p - kMin;
        const double ref = a / b;
        const double q = a * y;
        const double r = __builtin_fma(-b, q, a);
        const double q2 = __builtin_fma(r, y, q);
        if (memcmp(&ref, &q2, 8) != 0) ++bad;
      }
      bad1 += bad;
    });
  for (auto& x : th) x.join();
  th.clear();
  for (unsigned t = 0; t < nthreads; ++t)
    th.emplace_back([&, t]() {
      std::mt19937_64 rng(1234 + t);
      uint64_t bad = 0;
      const uint64_t n = limit / nthreads / 4 + 1000;
      for (uint64_t i = 0; i < n; ++i) {
        uint32_t gb = (uint32_t)rng(), pb = (uint32_t)(rng() >> 7);
        if (i & 1) {   // the range the kernel sees: sensitivities and absorbances of order 1 .. 1000
          gb = (gb & 0x007fffffu) | ((uint32_t)(120 + (rng() % 20)) << 23);
          pb = (pb & 0x007fffffu) | ((uint32_t)(120 + (rng() % 20)) << 23);
        }
        float g, p;
        memcpy(&g, &gb, 4);
        memcpy(&p, &pb, 4);
        if (!(fabsf(g) <= 3.0e38f) || !(fabsf(p) <= 3.0e38f) || p == 0.0f) continue;
        const float ref = (float)((double)g / (double)p);
        const float alt = g / p;
        if (memcmp(&ref, &alt, 4) != 0) ++bad;
      }
      bad2 += bad;
    });
  for (auto& x : th) x.join();
  printf("division by the constant (%llu float inputs): %llu mismatches\n", (unsigned long long)limit,
         (unsigned long long)bad1.load());
  printf("float(double(g) / double(p)) vs g / p: %llu mismatches\n", (unsigned long long)bad2.load());
  return (bad1.load() || bad2.load()) ? 1 : 0;
}
