// Separable Gaussian blur of butteraugli (Convolution / ConvolveBorderColumn / Blur,
// butteraugli.cc:156-233) as LDS-tiled row and column passes with coalesced global
// loads, plus the pointwise producers/consumers that are fused into them.
//
// Exact arithmetic kept:
//   interior sample (r <= x < n-r):  sum_{j=0..2r} in[x-r+j] * ks[j],  ks = k * (1/sum k),
//       f32, accumulated from 0.0f in ascending j
//   border sample:  (sum_{j in range} in[j] * k[j-x+r]) * scale(x)  with UNscaled taps;
//       scale(x) = 1 / ((1-br)*w_in_range + br*w_total) is precomputed on the host with
//       the same f32 operations.  Out-of-image taps are staged as 0.0f in LDS: adding
//       (+0.0f * k) never changes an f32 partial sum that started from +0.0f.
// The reference blurs along x first (writing the transpose), then along y.
#pragma once
#include "gz_common.h"
#include "gz_math.h"

namespace gz {

template <int R>
struct Taps {
  float k[2 * R + 1];    // ComputeKernel taps (unnormalised)
  float ks[2 * R + 1];   // taps * (1 / sum)
};

// Border scales for one axis of length n: lo[i] for position i (< R), hi[i] for position
// n-1-i (i < R).  Device pointers.
struct BorderScale {
  const float* lo;
  const float* hi;
};

// ----------------------------------------------------------------- source functors --
// A source yields the input sample at flat index `idx` (= y*pitch + x).
struct SrcPlain {
  const float* p;
  GZ_DEVFN float operator()(size_t idx) const { return p[idx]; }
};
struct SrcDiff {   // xyb - lf  (SeparateFrequencies, butteraugli.cc:512-517)
  const float* a;
  const float* b;
  GZ_DEVFN float operator()(size_t idx) const { return a[idx] - b[idx]; }
};
struct SrcSameNoise {   // butteraugli.cc:631-641
  const float* a;
  const float* b;
  GZ_DEVFN float operator()(size_t idx) const { return same_noise_pre(a[idx], b[idx]); }
};
template <class Src, int NC>
struct SrcPack {
  Src s[NC];
};
template <int NC>
struct PlanePack {
  float* p[NC];
};
template <int NC>
struct CPlanePack {
  const float* p[NC];
};

// Both passes work on a 64x64 output tile per workgroup of 256 threads and keep the taps'
// inputs in REGISTERS: a thread produces 16 consecutive outputs along the blur axis from a
// window of 16 + 2R staged samples, so each staged sample is read from LDS once per thread
// (not once per tap).  The four waves of a workgroup take the four 16-output groups, so
// whether an output is a border sample is the same for all lanes of a wave (a scalar
// branch).  Arithmetic per output is unchanged: f32, ascending taps from 0.0f.
constexpr int BT = 64;          // tile edge (outputs)
constexpr int BPT = BT / 4;     // outputs per thread along the blur axis

// One output of a window: interior = pre-scaled taps; border = raw taps, then one multiply.
template <int R>
GZ_DEVFN float blur_window_out(const float* win, int i, const Taps<R>& taps, bool border,
                               float scale) {
  float sum = 0.0f;
  if (!border) {
#pragma unroll
    for (int j = 0; j <= 2 * R; ++j) sum += win[i + j] * taps.ks[j];
  } else {
#pragma unroll
    for (int j = 0; j <= 2 * R; ++j) sum += win[i + j] * taps.k[j];
    sum = sum * scale;
  }
  return sum;
}

// ------------------------------------------------------------------------ row pass --
// grid = (ceil(w/64), ceil(h/64), NC).  The (64 rows) x (64 + 2R columns) input tile is
// staged with coalesced row loads; lane = row, wave = column group, so a lane walks along
// its row in LDS (odd pitch: conflict-free); the 64x64 results go back through LDS to be
// stored as full 256-byte row segments.
constexpr int HW = BT;
constexpr int HH = BT;

template <int R, class Src, int NC>
__global__ __launch_bounds__(256) void k_blur_h(SrcPack<Src, NC> src, PlanePack<NC> dst,
                                                int w, int h, int pitch, Taps<R> taps,
                                                BorderScale bs) {
  constexpr int TW = BT + 2 * R;                      // staged columns
  constexpr int P = (TW % 2 == 0) ? TW + 1 : TW;      // odd LDS pitch
  __shared__ float tile[BT * P];
  const int c = blockIdx.z;
  const Src s = src.s[c];
  float* __restrict__ out = dst.p[c];
  const int x0 = blockIdx.x * BT, y0 = blockIdx.y * BT;
  const int tid = threadIdx.x;
  for (int i = tid; i < BT * TW; i += 256) {
    const int ry = i / TW, rx = i - ry * TW;
    const int x = x0 - R + rx, y = y0 + ry;
    float v = 0.0f;
    if (x >= 0 && x < w && y < h) v = s((size_t)y * pitch + x);
    tile[ry * P + rx] = v;
  }
  __syncthreads();
  const int r = tid & 63;
  const int g = GZ_WAVE_UNIFORM(tid >> 6);
  float win[BPT + 2 * R];
#pragma unroll
  for (int i = 0; i < BPT + 2 * R; ++i) win[i] = tile[r * P + g * BPT + i];
  __syncthreads();   // every window is in registers: the tile can take the results
  constexpr int PO = BT + 1;
#pragma unroll
  for (int i = 0; i < BPT; ++i) {
    const int x = x0 + g * BPT + i;   // wave-uniform
    float sum = 0.0f;
    if (x < w) {
      const bool border = x < R || x >= w - R;
      float scale = 1.0f;
      if (border) scale = x < R ? bs.lo[x] : bs.hi[w - 1 - x];
      sum = blur_window_out<R>(win, i, taps, border, scale);
    }
    tile[r * PO + g * BPT + i] = sum;
  }
  __syncthreads();
  for (int i = tid; i < BT * BT; i += 256) {
    const int ry = i >> 6, rx = i & 63;
    const int x = x0 + rx, y = y0 + ry;
    if (x < w && y < h) out[(size_t)y * pitch + x] = tile[ry * PO + rx];
  }
}

// --------------------------------------------------------------------- column pass --
// grid = (ceil(w/VW), ceil(h/VH)); block = 256 = 64 columns x 4 row groups (one per wave);
// each thread produces VPT outputs of its column for every one of the NC planes, then hands
// the NC blurred values of each pixel to the Post functor (which may read/write other
// planes).
constexpr int VW = BT;
constexpr int VH = BT;
constexpr int VPT = BPT;   // outputs per thread

// With BM = true the values returned by the Post functor are additionally reduced to the
// per-8x8-block maxima of the tile (the tile origin is 8-aligned) and to one atomicMax per
// workgroup on the image maximum: first loop of ComputeBlockErrorAdjustmentWeights
// (butteraugli_comparator.cc:505-520) and ButteraugliScoreFromDiffmap
// (butteraugli.cc:1623-1633).  max is exact and order-free; values are >= 0, so the float
// order equals the order of the bit patterns.
struct BlockMaxOut {
  float* block_max;        // [nb] or null
  unsigned* image_max_bits;
  int bw;
};

template <int R, int NC, class Post, bool BM>
__global__ __launch_bounds__(256) void k_blur_v(CPlanePack<NC> src, Post post, int w, int h,
                                                int pitch, Taps<R> taps, BorderScale bs,
                                                BlockMaxOut bm) {
  __shared__ float tile[VH + 2 * R][VW];
  const int tx = threadIdx.x & 63;
  const int tg = GZ_WAVE_UNIFORM(threadIdx.x >> 6);
  const int x0 = blockIdx.x * VW, y0 = blockIdx.y * VH;
  const int x = x0 + tx;
  float acc[NC][VPT];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const float* __restrict__ in = src.p[c];
    if (c > 0) __syncthreads();
    for (int ry = tg; ry < VH + 2 * R; ry += 4) {
      const int y = y0 - R + ry;
      float v = 0.0f;
      if (x < w && y >= 0 && y < h) v = in[(size_t)y * pitch + x];
      tile[ry][tx] = v;
    }
    __syncthreads();
    float win[VPT + 2 * R];
#pragma unroll
    for (int i = 0; i < VPT + 2 * R; ++i) win[i] = tile[tg * VPT + i][tx];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int y = y0 + tg * VPT + i;   // wave-uniform
      float sum = 0.0f;
      if (y < h) {
        const bool border = y < R || y >= h - R;
        float scale = 1.0f;
        if (border) scale = y < R ? bs.lo[y] : bs.hi[h - 1 - y];
        sum = blur_window_out<R>(win, i, taps, border, scale);
      }
      acc[c][i] = sum;
    }
  }
  float res[VPT];
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int y = y0 + tg * VPT + i;
    res[i] = 0.0f;
    if (x < w && y < h) {
      float v[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) v[c] = acc[c][i];
      res[i] = post((size_t)y * pitch + x, v);
    }
  }
  if (BM) {
    __shared__ float s_bmax[64];
    __syncthreads();   // all column reads of the last plane are done
#pragma unroll
    for (int i = 0; i < VPT; ++i) tile[tg * VPT + i][tx] = res[i];
    __syncthreads();
    if (threadIdx.x < 64) {
      const int bxl = threadIdx.x & 7, byl = threadIdx.x >> 3;
      float m = 0.0f;
      for (int yy = 0; yy < 8; ++yy)
        for (int xx = 0; xx < 8; ++xx) {
          const float t = tile[8 * byl + yy][8 * bxl + xx];
          m = t > m ? t : m;
        }
      s_bmax[threadIdx.x] = m;
      const int gbx = x0 / 8 + bxl, gby = y0 / 8 + byl;
      if (bm.block_max && 8 * gbx < w && 8 * gby < h) bm.block_max[gby * bm.bw + gbx] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float m = 0.0f;
      for (int i = 0; i < 64; ++i) m = s_bmax[i] > m ? s_bmax[i] : m;
      atomicMax(bm.image_max_bits, __float_as_uint(m));
    }
  }
}

// ------------------------------------------------------------------- post functors --
template <int NC>
struct PostStore {
  float* out[NC];
  GZ_DEVFN float operator()(size_t idx, const float* v) const {
#pragma unroll
    for (int c = 0; c < NC; ++c) out[c][idx] = v[c];
    return v[0];
  }
};

// OpsinDynamicsImage, butteraugli.cc:337-363: v = blurred rgb; reads sharp rgb.
struct PostOpsin {
  const float* lin[3];
  float* xyb[3];
  GZ_DEVFN float operator()(size_t idx, const float* v) const {
    float x, y, z;
    opsin_pixel(v[0], v[1], v[2], lin[0][idx], lin[1][idx], lin[2][idx], &x, &y, &z);
    xyb[0][idx] = x;
    xyb[1][idx] = y;
    xyb[2][idx] = z;
    return y;
  }
};

// LF band (butteraugli.cc:510, :606-621): v = blur(xyb, sigma_lf).  Keeps the raw LF of
// X and Y (needed by the MF band and by the bright-area suppression) and writes the
// "vals" conversion of all three.
struct PostLF {
  float* lf_raw[2];
  float* lf_vals[3];
  GZ_DEVFN float operator()(size_t idx, const float* v) const {
    lf_raw[0][idx] = v[0];
    lf_raw[1][idx] = v[1];
    float vx, vy, vb;
    lf_to_vals(v[0], v[1], v[2], &vx, &vy, &vb);
    lf_vals[0][idx] = vx;
    lf_vals[1][idx] = vy;
    lf_vals[2][idx] = vb;
    return vy;
  }
};

// MF band of X and Y (butteraugli.cc:511-550) + SuppressXByY (:552-554):
// v = blur(xyb - lf, sigma_hf) for c = 0,1.  (The B channel's MF is never consumed:
// wmul[5] == 0, butteraugli.cc:873-883, and Malta runs on X and Y only.)
struct PostMF {
  const float* xyb[2];
  const float* lf_raw[2];
  float* mf[2];
  float* hf_pre[2];
  GZ_DEVFN float operator()(size_t idx, const float* v) const {
    const float band0 = xyb[0][idx] - lf_raw[0][idx];
    const float band1 = xyb[1][idx] - lf_raw[1][idx];
    const float h0 = band0 - v[0];
    const float h1 = band1 - v[1];
    mf[0][idx] = remove_range((float)0.120079806822, v[0]);
    mf[1][idx] = amplify_range((float)0.03430529365, v[1]);
    hf_pre[0][idx] = suppress_x_by_y(h0, h1);
    hf_pre[1][idx] = h1;
    return h1;
  }
};

// HF / UHF split (butteraugli.cc:556-603): v = blur(hf_pre, sigma_uhf) for c = 0,1.
struct PostHF {
  const float* hf_pre[2];
  const float* lf_raw_y;
  float* hf[2];
  float* uhf[2];
  GZ_DEVFN float operator()(size_t idx, const float* v) const {
    const float kMulSuppressHf = (float)1.10684769012;
    const float kMulRegHf = (float)0.478741530298;
    const float kRegHf = 2000 * kMulRegHf;
    const float kMulSuppressUhf = (float)1.76905001176;
    const float kMulRegUhf = (float)0.310148420674;
    const float kRegUhf = 2000 * kMulRegUhf;
    // X
    uhf[0][idx] = hf_pre[0][idx] - v[0];
    hf[0][idx] = remove_range((float)0.0287615200377, v[0]);
    // Y
    const float br = lf_raw_y[idx];
    float u = hf_pre[1][idx] - v[1];
    float hv = maximum_clamp(v[1], (float)78.8223237675);
    u = maximum_clamp(u, (float)5.8907152736);
    u = suppress_bright(u, br, kMulSuppressUhf, kRegUhf);
    hv = suppress_bright(hv, br, kMulSuppressHf, kRegHf);
    uhf[1][idx] = u;
    hf[1][idx] = hv;
    return hv;
  }
};

// Second half of CalculateDiffmap (butteraugli.cc:736-749): v = blur(d, 1.725, br 1.0).
struct PostDiffmapMix {
  const float* d;
  float* out;
  GZ_DEVFN float operator()(size_t idx, const float* v) const {
    const double mul1 = 0.458794906198;
    const float scale = (float)(1.0f / (1.0f + mul1));
    float r = d[idx];
    r = (float)((double)r + mul1 * (double)v[0]);
    r = r * scale;
    out[idx] = r;
    return r;
  }
};

// Mask Y blur pair (butteraugli.cc:1780-1790): v[0] = blur(diffY, r0=2.377) arrives from
// an earlier pass in `b1`; this one (sigma r1) combines them.  Kept as plain stores: the
// combination is done in k_combine, which needs both anyway.

}  // namespace gz
