// Double-precision 8x8 DCT-II / DCT-III block kernels (dct_double.cc:28-85) and the two
// component-level users of them in the reference's 4:2:0 path:
//
//   k_dctd_blocks<INV>        ComputeBlockDCTDouble / ComputeBlockIDCTDouble on bare blocks
//   k_to_float_pixels         OutputImageComponent::ToFloatPixels   (output_image.cc:99-121)
//   k_set_downsampled_coeffs  SetDownsampledCoefficients            (output_image.cc:265-300)
//
// One 64-lane wavefront per 8x8 block, lane = output element, the block staged in LDS
// between the column and the row sweep.  FP64, bit-exact: every output is the reference's
// left-to-right sum `out = 0.0; out += M[..] * in[u]` for u = 0..7 (no FMA: the library is
// built with -ffp-contract=off), so results equal the x86-64 SSE2 reference bit for bit.
#pragma once
#include "gz_common.h"
#include "gz_kernels_block.h"   // GZ_CONST, kBlocksPerWG

namespace gz {

// kDCTMatrix[8*u+x] = 0.5*alpha(u)*cos((2x+1)u*pi/16) rounded to 10 digits as the reference
// stores it (dct_double.cc:28-45) -- data.
GZ_CONST double kDctD[64] = {
  0.3535533906,  0.3535533906,  0.3535533906,  0.3535533906,
  0.3535533906,  0.3535533906,  0.3535533906,  0.3535533906,
  0.4903926402,  0.4157348062,  0.2777851165,  0.0975451610,
 -0.0975451610, -0.2777851165, -0.4157348062, -0.4903926402,
  0.4619397663,  0.1913417162, -0.1913417162, -0.4619397663,
 -0.4619397663, -0.1913417162,  0.1913417162,  0.4619397663,
  0.4157348062, -0.0975451610, -0.4903926402, -0.2777851165,
  0.2777851165,  0.4903926402,  0.0975451610, -0.4157348062,
  0.3535533906, -0.3535533906, -0.3535533906,  0.3535533906,
  0.3535533906, -0.3535533906, -0.3535533906,  0.3535533906,
  0.2777851165, -0.4903926402,  0.0975451610,  0.4157348062,
 -0.4157348062, -0.0975451610,  0.4903926402, -0.2777851165,
  0.1913417162, -0.4619397663,  0.4619397663, -0.1913417162,
 -0.1913417162,  0.4619397663, -0.4619397663,  0.1913417162,
  0.0975451610, -0.2777851165,  0.4157348062, -0.4903926402,
  0.4903926402, -0.4157348062,  0.2777851165, -0.0975451610,
};

// TransformBlock (dct_double.cc:66-74) of the LDS-resident block `blk` (this wave's 64
// doubles), `tmp` = this wave's second 64 doubles.  Returns this lane's element of the
// result; every lane of the wave must call it (two workgroup barriers inside).
//   DCT1d  (:47-54): out[x] = sum_u M[8x+u] * in[u]
//   IDCT1d (:56-63): out[x] = sum_u M[8u+x] * in[u]
template <bool INV>
GZ_DEVFN double dctd_transform(const double* blk, double* tmp, int lane) {
  const int hi = lane >> 3, lo = lane & 7;
  // first sweep, f(&block[x], 8, &tmp[x]) for x = lo: tmp[8v + x] (v = hi) from column x
  double acc = 0.0;
#pragma unroll
  for (int u = 0; u < 8; ++u)
    acc += (INV ? kDctD[8 * u + hi] : kDctD[8 * hi + u]) * blk[8 * u + lo];
  tmp[lane] = acc;
  __syncthreads();
  // second sweep, f(&tmp[8y], 1, &block[8y]) for y = hi: block[8y + v] (v = lo) from row y
  acc = 0.0;
#pragma unroll
  for (int u = 0; u < 8; ++u)
    acc += (INV ? kDctD[8 * u + lo] : kDctD[8 * lo + u]) * tmp[8 * hi + u];
  __syncthreads();
  return acc;
}

template <bool INV>
__global__ __launch_bounds__(256) void k_dctd_blocks(double* __restrict__ blocks, int n) {
  __shared__ double s_blk[kBlocksPerWG][64];
  __shared__ double s_tmp[kBlocksPerWG][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int blk = blockIdx.x * kBlocksPerWG + wave;
  const bool live = blk < n;
  s_blk[wave][lane] = live ? blocks[(size_t)blk * 64 + lane] : 0.0;
  __syncthreads();
  const double v = dctd_transform<INV>(s_blk[wave], s_tmp[wave], lane);
  if (live) blocks[(size_t)blk * 64 + lane] = v;
}

// ToFloatPixels with stride 1 for one component at factor 1x1: coeffs [nb][64] int16 ->
// out[y*w + x] = float(idct_double(coeffs)[8*iy+ix] + 128.0), in-image pixels only.
__global__ __launch_bounds__(256) void k_to_float_pixels(const int16_t* __restrict__ coeffs,
                                                         int w, int h, int bw, int nb,
                                                         float* __restrict__ out) {
  __shared__ double s_blk[kBlocksPerWG][64];
  __shared__ double s_tmp[kBlocksPerWG][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int blk = blockIdx.x * kBlocksPerWG + wave;
  const bool live = blk < nb;
  s_blk[wave][lane] = live ? (double)coeffs[(size_t)blk * 64 + lane] : 0.0;
  __syncthreads();
  const double v = dctd_transform<true>(s_blk[wave], s_tmp[wave], lane);
  if (!live) return;
  const int x = 8 * (blk % bw) + (lane & 7), y = 8 * (blk / bw) + (lane >> 3);
  if (x < w && y < h) out[(size_t)y * w + x] = (float)(v + 128.0);
}

// SetDownsampledCoefficients: pixels = w*h floats of the full-resolution component; the
// component is reset to (fx, fy) subsampling: bw = ceil(w / (8 fx)), bh = ceil(h / (8 fy)).
// Per output sample: float average of the fx*fy source pixels (edge-clamped, summed j outer /
// i inner from 0.0f, then one float division), forward DCT in double, DC -= 1024, round()
// half away from zero, store as int16.
__global__ __launch_bounds__(256) void k_set_downsampled_coeffs(
    const float* __restrict__ pixels, int w, int h, int fx, int fy, int bw, int nb,
    int16_t* __restrict__ coeffs) {
  __shared__ double s_blk[kBlocksPerWG][64];
  __shared__ double s_tmp[kBlocksPerWG][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int blk = blockIdx.x * kBlocksPerWG + wave;
  const bool live = blk < nb;
  double in = 0.0;
  if (live) {
    const int ix = lane & 7, iy = lane >> 3;
    const int x0 = 8 * (blk % bw) * fx, y0 = 8 * (blk / bw) * fy;
    float avg = 0.0f;
    for (int j = 0; j < fy; ++j) {
      for (int i = 0; i < fx; ++i) {
        int x = x0 + ix * fx + i;
        x = x < w - 1 ? x : w - 1;
        int y = y0 + iy * fy + j;
        y = y < h - 1 ? y : h - 1;
        avg += pixels[(size_t)y * w + x];
      }
    }
    avg /= (float)(fx * fy);
    in = (double)avg;
  }
  s_blk[wave][lane] = in;
  __syncthreads();
  double v = dctd_transform<false>(s_blk[wave], s_tmp[wave], lane);
  if (!live) return;
  if (lane == 0) v -= 1024.0;
  coeffs[(size_t)blk * 64 + lane] = (int16_t)round(v);
}

}  // namespace gz
