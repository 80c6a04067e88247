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

// 3. gamma_poly_f of gz_math.h (shortened Clenshaw recursion, constant division by two FMAs)
//    against RationalPolynomial::operator() as the reference states it (butteraugli.h:548-615),
//    for EVERY float argument (NaNs included: both sides return the same bits or both a NaN).
#define GZ_EMU 1
#include "gz_common.h"
#include "gz_math.h"
static double ref_poly(double x, const double* c) {
  double b1 = 0.0, b2 = 0.0;
  for (int k = 5; k >= 1; --k) {
    const double x_b1 = x * b1;
    const double t = (x_b1 + x_b1) - b2 + c[k];
    b2 = b1;
    b1 = t;
  }
  const double x_b1 = x * b1;
  return x_b1 - b2 + c[0];
}
static float ref_gamma(double x) {
  static const double p[6] = {98.7821300963361, 164.273222212631, 92.948112871376,
                              33.8165311212688, 6.91626704983562, 0.556380877028234};
  static const double q[6] = {1, 1.64339473427892, 0.89392405219969, 0.298947051776379,
                              0.0507146002577288, 0.00226495093949756};
  const double x01 = (x - kMin) / (kMax - kMin);
  const double xc = 2.0 * x01 - 1.0;
  const double yp = ref_poly(xc, p), yq = ref_poly(xc, q);
  if (yq == 0.0) return (float)0.0;
  return static_cast<float>(yp / yq);
}

int main(int argc, char** argv) {
  const double b = kMax - kMin;
  const double y = 1.0 / b;
  const unsigned nthreads = std::max(1u, std::thread::hardware_concurrency());
  const uint64_t limit = argc > 1 ? strtoull(argv[1], nullptr, 10) : (1ull << 32);   // inputs of check 1
  std::atomic<uint64_t> bad1(0), bad2(0), bad3(0);
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
      uint64_t b3 = 0;
      for (uint64_t u = t; u < limit; u += nthreads) {
        const uint32_t bits = (uint32_t)(u * 2654435761ull);
        float p;
        memcpy(&p, &bits, 4);
        const float want = ref_gamma((double)p), got = gz::gamma_poly_f((double)p);
        if (memcmp(&want, &got, 4) != 0 && !(want != want && got != got)) ++b3;
      }
      bad3 += b3;
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
  printf("gamma_poly_f vs the reference's rational polynomial (%llu float inputs): %llu mismatches\n",
         (unsigned long long)limit, (unsigned long long)bad3.load());
  return (bad1.load() || bad2.load() || bad3.load()) ? 1 : 0;
}
