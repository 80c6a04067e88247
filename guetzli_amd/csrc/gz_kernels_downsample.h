// OutputImage::Downsample (output_image.cc:304-340) without the "silver screen" option: the
// chroma pre-processing of guetzli/preprocess_downsample.cc:28-279 (PreProcessChannel: sharpen
// the channel where the image is red and dark, blur it where it is smooth) that runs before
// the two chroma components are averaged 2x2 and transformed again
// (SetDownsampledCoefficients, k_set_downsampled_coeffs in gz_kernels_dctd.h).
//
// Once per image, pointwise / small-stencil float work on three w*h planes (row pitch w): one
// thread per pixel, coalesced rows; HBM-bound and tiny next to the search (about twenty
// launches of ~10 us at 1080p).  Every expression keeps the reference's types: float where it
// computes in float, double where a double literal promotes (the image is compiled without
// FMA contraction), so results equal the x86-64 reference bit for bit.  Kernel taps
// (Normal(), :85-88: std::exp in double) are host-built.
#pragma once
#include "gz_common.h"

namespace gz {

// :164-168  yuv[0] /= 255.0 (double division, rounded to float); u, v: x / 255.0f - 0.5f
__global__ __launch_bounds__(256) void k_pp_normalize(float* __restrict__ y, float* __restrict__ u,
                                                      float* __restrict__ v, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  y[i] = (float)((double)y[i] / 255.0);
  u[i] = u[i] / 255.0f - 0.5f;
  v[i] = v[i] / 255.0f - 0.5f;
}

// :272-277  back to 0..255
__global__ __launch_bounds__(256) void k_pp_denormalize(float* __restrict__ y, float* __restrict__ u,
                                                        float* __restrict__ v, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  y[i] = (float)((double)y[i] * 255.0);
  u[i] = (u[i] + 0.5f) * 255.0f;
  v[i] = (v[i] + 0.5f) * 255.0f;
}

// darkmap (:171-192) and redmap (:199-215) before their erosions / dilations.
__global__ __launch_bounds__(256) void k_pp_maps(const float* __restrict__ yp, const float* __restrict__ up,
                                                 const float* __restrict__ vp, size_t n, int channel,
                                                 uint8_t* __restrict__ dark, uint8_t* __restrict__ red) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float y = yp[i], u = up[i], v = vp[i];
  const float r = y + 1.402f * v;
  const float g = y - 0.34414f * u - 0.71414f * v;
  const float b = y + 1.772f * u;
  bool d = false, rd = false;
  if (channel == 2) {
    d = (double)g < 0.85 && (double)b < 0.85 && (double)r < 0.9;
    rd = 2.116 * (double)v > -0.34414 * (double)u + 0.2 && 1.402 * (double)v > 1.772 * (double)u + 0.2;
  } else {
    d = (double)r < 0.85 && (double)g < 0.85 && (double)b < 0.9;
    rd = (double)v < 1.263 * (double)u - 0.1 && (double)u > -0.33741 * (double)v;
  }
  dark[i] = d ? 1 : 0;
  red[i] = rd ? 1 : 0;
}

// Erode (:110-121) / Dilate (:123-134): 5-point cross on the interior, the border row and
// column keep their value.
__global__ __launch_bounds__(256) void k_pp_morph(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                  int w, int h, int erode) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w || y >= h) return;
  const size_t i = (size_t)y * w + x;
  uint8_t v = in[i];
  if (x >= 1 && x + 1 < w && y >= 1 && y + 1 < h) {
    const uint8_t l = in[i - 1], r = in[i + 1], a = in[i - w], b = in[i + w];
    if (erode) v = (v && l && r && a && b) ? v : 0;
    else v = (v || l || r || a || b) ? 1 : v;
  }
  out[i] = v;
}

// sharpenmap (:222-228) = redmap && darkmap; blurmap before its erosions (:240-253): not
// sharpened, dark, |edge| < threshold with edge = Convolve2D(channel, {0,-1,0,-1,4,-1,0,-1,0})
// (:29-50: float accumulation over the nine taps in order, border pixels keep the image value),
// and v < -0.162 * u.
__global__ __launch_bounds__(256) void k_pp_edge_maps(const float* __restrict__ ch, const float* __restrict__ up,
                                                      const float* __restrict__ vp,
                                                      const uint8_t* __restrict__ dark,
                                                      const uint8_t* __restrict__ red, int w, int h,
                                                      double threshold, uint8_t* __restrict__ sharpen,
                                                      uint8_t* __restrict__ blur) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w || y >= h) return;
  const size_t i = (size_t)y * w + x;
  float edge = ch[i];
  if (x >= 1 && x + 1 < w && y >= 1 && y + 1 < h) {
    const float k[9] = {0.0f, -1.0f, 0.0f, -1.0f, 4.0f, -1.0f, 0.0f, -1.0f, 0.0f};
    float acc = 0;
#pragma unroll
    for (int j = 0; j < 9; ++j) acc += k[j] * ch[(size_t)(y + j / 3 - 1) * w + (x + j % 3 - 1)];
    edge = acc;
  }
  const bool sh = red[i] && dark[i];
  bool bl = false;
  if (!sh && dark[i]) bl = fabs((double)edge) < threshold && (double)vp[i] < -0.162 * (double)up[i];
  sharpen[i] = sh ? 1 : 0;
  blur[i] = bl ? 1 : 0;
}

struct PPTaps {
  float ks[5], kb[5];   // float(kernel[j]) of Sharpen (sigma = double(1.3f)) and of Blur (1.3)
  float mul_s, mul_b;   // float(1 / sum)
};

// First half of Convolve2X (:53-68) for both kernels: along x, columns 2..w-3.
__global__ __launch_bounds__(256) void k_pp_conv_h(const float* __restrict__ ch, int w, int h, PPTaps t,
                                                   float* __restrict__ tmp_s, float* __restrict__ tmp_b) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w || y >= h) return;
  const size_t i = (size_t)y * w + x;
  float s = ch[i], b = ch[i];
  if (!(x < 2 || x + 2 >= w)) {
    float vs = 0, vb = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const float p = ch[i + j - 2];
      vs += t.ks[j] * p;
      vb += t.kb[j] * p;
    }
    s = vs * t.mul_s;
    b = vb * t.mul_b;
  }
  tmp_s[i] = s;
  tmp_b[i] = b;
}

// Second half (:69-82) along y, the unsharp mask of Sharpen (:104-106), and the per-pixel
// choice (:258-270); writes the channel in place (only its own pixel of `ch` is read).
__global__ __launch_bounds__(256) void k_pp_conv_v_select(float* __restrict__ ch, const float* __restrict__ tmp_s,
                                                          const float* __restrict__ tmp_b,
                                                          const uint8_t* __restrict__ sharpen,
                                                          const uint8_t* __restrict__ blur, int w, int h,
                                                          PPTaps t, float amount, int do_sharpen, int do_blur) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w || y >= h) return;
  const size_t i = (size_t)y * w + x;
  float s = tmp_s[i], b = tmp_b[i];
  if (!(y < 2 || y + 2 >= h)) {
    float vs = 0, vb = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const size_t o = (size_t)(y + j - 2) * w + x;
      vs += t.ks[j] * tmp_s[o];
      vb += t.kb[j] * tmp_b[o];
    }
    s = vs * t.mul_s;
    b = vb * t.mul_b;
  }
  const float img = ch[i];
  const float sharpened = img + (img - s) * amount;
  float out = img;
  if (sharpen[i]) {
    if (do_sharpen) out = sharpened;
  } else if (blur[i]) {
    if (do_blur) out = b;
  }
  ch[i] = out;
}

}  // namespace gz
