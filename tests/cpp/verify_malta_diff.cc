// TEST INFRASTRUCTURE.  Checks that the branch-free malta_diff of gz_math.h gives the bits of
// malta_diff_plain -- the reference's statement sequence (MaltaDiffMapImpl, butteraugli.cc:
// 1473-1529) -- for the six normalisations DiffmapPsychoImage uses, on random, structured and
// adversarial sample pairs (ratios around the 0.55 / 1.05 thresholds, zeros, signed zeros,
// denormals, huge values), and that div2_shared is the IEEE quotient for every denominator
// pattern it is offered, starting from a reciprocal estimate that is off by -1, 0, +1 ulp.
// Build: g++ -O2 -mfma -ffp-contract=off -pthread -DGZ_EMU -I guetzli_amd/csrc -I tests/emu
//        tests/cpp/verify_malta_diff.cc -o verify && ./verify [pairs per normalisation]
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <random>
#include <thread>
#include <vector>

#include "gz_common.h"
#include "gz_math.h"

using gz::MaltaNorm;

static MaltaNorm norm_of(bool lf, double w_0gt1, double w_0lt1, double norm1) {
  const double len = 3.75;
  const double mulli = lf ? 0.405371989604 : 0.354191303559;
  const float kWeight0 = 0.5;
  const float kWeight1 = 0.33;
  const double w_pre0gt1 = mulli * sqrt(kWeight0 * w_0gt1) / (len * 2 + 1);
  const double w_pre0lt1 = mulli * sqrt(kWeight1 * w_0lt1) / (len * 2 + 1);
  MaltaNorm n;
  n.norm2_0gt1 = w_pre0gt1 * norm1;
  n.norm2_0lt1 = w_pre0lt1 * norm1;
  n.norm1f = static_cast<float>(norm1);
  n.fast_div = 1;
  return n;
}

static uint32_t bits_of(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static float float_of(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }

int main(int argc, char** argv) {
  const uint64_t pairs = argc > 1 ? strtoull(argv[1], nullptr, 10) : 200000000ull;
  const float asym = 0.8f, sq = sqrtf(asym);
  const MaltaNorm norms[6] = {
      norm_of(false, 5.1409625726 * asym, 5.1409625726 / asym, 58.5001247061),
      norm_of(true, 153.671655716 * sq, 153.671655716 / sq, 83150785.9592),
      norm_of(true, 6841.81248144, 6841.81248144, 0.0135134962487),
      norm_of(false, 4.91743441556 * asym, 4.91743441556 / asym, 687196.39002),
      norm_of(true, 668.358918152 * sq, 668.358918152 / sq, 0.882954368025),
      norm_of(true, 813.901703816, 813.901703816, 16792.9322251)};
  const unsigned nthreads = std::max(1u, std::thread::hardware_concurrency());
  // ---- 1. div2_shared against the IEEE quotient, every mantissa of the denominator at a
  // spread of exponents in [2^-40, 2^40], reciprocal estimate off by -1 / 0 / +1 ulp
  uint64_t bad_div = 0, n_div = 0;
  for (int ulps = -1; ulps <= 1; ++ulps) {
    gz::gz_emu_rcp_ulps() = ulps;
    std::atomic<uint64_t> bad(0);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nthreads; ++t)
      th.emplace_back([&, t]() {
        uint64_t b = 0;
        for (int e = -40; e < 40; e += 3)
          for (uint32_t m = t; m < (1u << 23); m += nthreads) {
            const float d = float_of(((uint32_t)(e + 127) << 23) | m);
            for (int k = 0; k < 6; k += (m & 1) ? 5 : 1) {   // all six norms on even mantissas
              float q0, q1;
              gz::div2_shared(norms[k].norm2_0gt1, norms[k].norm2_0lt1, d, &q0, &q1);
              if (bits_of(q0) != bits_of(norms[k].norm2_0gt1 / d)) ++b;
              if (bits_of(q1) != bits_of(norms[k].norm2_0lt1 / d)) ++b;
            }
          }
        bad += b;
      });
    for (auto& x : th) x.join();
    bad_div += bad;
    n_div += (uint64_t)27 * (1u << 23) * 7;
  }
  printf("div2_shared: ~%llu quotients, %llu mismatches\n", (unsigned long long)n_div,
         (unsigned long long)bad_div);
  // ---- 2. malta_diff against malta_diff_plain
  uint64_t bad_md = 0;
  for (int ulps = -1; ulps <= 1; ++ulps) {
    gz::gz_emu_rcp_ulps() = ulps;
    std::atomic<uint64_t> bad(0);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nthreads; ++t)
      th.emplace_back([&, t]() {
        std::mt19937_64 rng(1234567 + 977 * t + 31 * (ulps + 1));
        uint64_t b = 0;
        const float specials[] = {0.0f, -0.0f, 1e-45f, -1e-45f, 1e-39f, 1.1754944e-38f, 1e-30f,
                                  7.8e-31f, 1.0f, -1.0f, 3.0e38f, -3.0e38f, 1e30f, 1e31f, 6e29f};
        const double ratios[] = {0.55, 1.05, -0.55, -1.05, 1.0, -1.0, 0.0, 0.5499999, 0.5500001,
                                 1.0499999, 1.0500001, 2.0, 1e-9, 1e9};
        const uint64_t per = pairs / nthreads / 3 + 1;
        for (uint64_t i = 0; i < per; ++i) {
          float a, c;
          const uint64_t r = rng();
          switch (r & 7) {
            case 0: {   // arbitrary bit patterns
              a = float_of((uint32_t)(r >> 8));
              c = float_of((uint32_t)(rng() >> 13));
              break;
            }
            case 1: {   // specials against anything
              a = specials[(r >> 8) % 15];
              c = (r & 0x100000) ? specials[(r >> 24) % 15] : float_of((uint32_t)(rng() >> 7));
              break;
            }
            case 2: case 3: {   // c = a * ratio near a threshold, nudged by a few ulps
              a = (float)((double)((int64_t)(r >> 20) % 2000001 - 1000000) * ldexp(1.0, (int)((r >> 8) % 40) - 30));
              const double q = ratios[(r >> 14) % 14];
              c = (float)((double)a * q);
              uint32_t u = bits_of(c);
              u += (uint32_t)((rng() % 9)) - 4;
              c = float_of(u);
              break;
            }
            default: {   // band-like magnitudes: both within a few decades
              const double sa = ldexp(1.0, (int)((r >> 8) % 28) - 14);
              a = (float)(((double)(int64_t)(rng() >> 11) / 4503599627370496.0 - 1.0) * sa);
              c = (float)(a * (1.0 + ((double)(int64_t)(rng() >> 11) / 4503599627370496.0 - 1.0) * 1.2));
              break;
            }
          }
          if (a != a || c != c) continue;   // NaNs: not produced by the chain
          const MaltaNorm& nm = norms[(r >> 3) % 6];
          if (bits_of(gz::malta_diff(a, c, nm)) != bits_of(gz::malta_diff_plain(a, c, nm))) {
            if (b < 3) fprintf(stderr, "mismatch a=%a b=%a\n", a, c);
            ++b;
          }
        }
        bad += b;
      });
    for (auto& x : th) x.join();
    bad_md += bad;
  }
  printf("malta_diff: ~%llu pairs, %llu mismatches\n", (unsigned long long)pairs,
         (unsigned long long)bad_md);
  return bad_div || bad_md ? 1 : 0;
}
