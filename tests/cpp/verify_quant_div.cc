// TEST INFRASTRUCTURE.  quant_div (guetzli_amd/csrc/gz_kernels_entropy.h) -- the entropy coder's coefficient / q as one
// float multiply by 1.0f / q and an integer correction -- against C++'s int division, for EVERY
// dividend an int16 coefficient (or the DC difference logic's operands) can be, |a| <= 32768, and
// every quantiser 1 .. 65535 (JPEG's 16-bit tables; the search uses <= 255).  The float product
// and the conversion are single IEEE operations on the device as here (contraction is off, the
// reciprocal is the correctly rounded quotient 1.0f / q on both sides).
#include <stdio.h>
#include <stdlib.h>

#include <thread>
#include <vector>

#include "hip_emu.h"
#include "gz_kernels_entropy.h"

int main() {
  const int nthreads = 8;
  std::vector<long> bad(nthreads, 0);
  std::vector<std::thread> pool;
  for (int t = 0; t < nthreads; ++t)
    pool.emplace_back([&, t] {
      for (int q = 1 + t; q <= 65535; q += nthreads) {
        const float rq = 1.0f / (float)q;
        for (int a = -32768; a <= 32768; ++a)
          if (gz::quant_div(a, q, rq) != a / q) ++bad[t];
      }
    });
  for (auto& th : pool) th.join();
  long total = 0;
  for (long b : bad) total += b;
  printf("quant_div: %ld mismatches over 65537 x 65535 pairs\n", total);
  return total == 0 ? 0 : 1;
}
