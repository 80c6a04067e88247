// Host part of ComputeBlockErrorAdjustmentWeights (butteraugli_comparator.cc:521-557):
// from the per-block maxima of the distance map (computed on the device, :505-520) to the
// per-block weights the search driver ranks candidates with.  O(nb * (2r+1)^2) on nb
// floats; header-only so that the C-ABI library and the host-logic replay harness of the
// test-suite (tests/replay) run the same code.
#pragma once
#include <algorithm>
#include <cstdlib>

namespace gz {

inline void block_weights_host(const float* bmax, int bw, int bh, float target, int direction,
                               int max_block_dist, double target_mul, float* block_weight) {
  const double target_distance = target * target_mul;
  for (int by = 0; by < bh; ++by) {
    for (int bx = 0; bx < bw; ++bx) {
      const int bix = by * bw + bx;
      float local = static_cast<float>(target_distance);
      const int x0 = std::max(0, bx - max_block_dist), y0 = std::max(0, by - max_block_dist);
      const int x1 = std::min(bw, bx + 1 + max_block_dist);
      const int y1 = std::min(bh, by + 1 + max_block_dist);
      for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) local = std::max(local, bmax[y * bw + x]);
      if (direction > 0) {
        if (bmax[bix] <= target_distance && local <= 1.1 * target_distance) block_weight[bix] = 1.0;
      } else {
        const double kLocalMaxWeight = 0.5;
        if (bmax[bix] <= (1 - kLocalMaxWeight) * target_distance + kLocalMaxWeight * local) continue;
        for (int y = y0; y < y1; ++y)
          for (int x = x0; x < x1; ++x) {
            const int d = std::max(std::abs(y - by), std::abs(x - bx));
            const int ix = y * bw + x;
            block_weight[ix] = std::max<float>(block_weight[ix], 1.0f / (d + 1.0f));
          }
      }
    }
  }
}

}  // namespace gz
