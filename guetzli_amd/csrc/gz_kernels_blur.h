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
  GZ_DEVFN float operator()(size_t idx) const { return GZ_LDG(p, idx); }
  GZ_DEVFN gz_f4 load4(size_t idx) const { return GZ_LDG4(p, idx); }
};
struct SrcDiff {   // xyb - lf  (SeparateFrequencies, butteraugli.cc:512-517)
  const float* a;
  const float* b;
  GZ_DEVFN float operator()(size_t idx) const { return GZ_LDG(a, idx) - GZ_LDG(b, idx); }
  GZ_DEVFN gz_f4 load4(size_t idx) const {
    const gz_f4 u = GZ_LDG4(a, idx), v = GZ_LDG4(b, idx);
    gz_f4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.v[i] = u.v[i] - v.v[i];
    return r;
  }
};
struct SrcSameNoise {   // butteraugli.cc:631-641
  const float* a;
  const float* b;
  GZ_DEVFN float operator()(size_t idx) const { return same_noise_pre(GZ_LDG(a, idx), GZ_LDG(b, idx)); }
  GZ_DEVFN gz_f4 load4(size_t idx) const {
    const gz_f4 u = GZ_LDG4(a, idx), v = GZ_LDG4(b, idx);
    gz_f4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.v[i] = same_noise_pre(u.v[i], v.v[i]);
    return r;
  }
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

// ------------------------------------------------------------------------ row pass --
// grid = (ceil(w/HW), ceil(h/HH), NC); block = 256 threads; thread = one x, HH rows.
constexpr int HW = 256;
constexpr int HH = 4;

// TWO = true (NC == 2): the second plane has its own taps and border scales (taps1, bs1) -- two
// blurs of different sigma but equal radius in one launch (the mask's radius-20 pair).
template <int R, class Src, int NC, bool TWO = false>
__global__ __launch_bounds__(256) void k_blur_h(SrcPack<Src, NC> src, PlanePack<NC> dst,
                                                int w, int h, int pitch, Taps<R> taps0,
                                                BorderScale bs0, Taps<R> taps1, BorderScale bs1) {
  constexpr int RA = (R + 3) & ~3;      // halo rounded up to whole 16-byte vectors
  constexpr int TP = HW + 2 * RA;       // staged columns (multiple of 4)
  constexpr int OFF = RA - R;
  __shared__ __attribute__((aligned(16))) float tile[HH][TP];
  const GzTile bid = gz_xcd_tile();
  const int c = bid.z;
  // constant indices into the kernel arguments (a dynamic one would make the pointers generic)
  Src s = src.s[0];
  float* __restrict__ out = dst.p[0];
#pragma unroll
  for (int i = 1; i < NC; ++i)
    if (c == i) {
      s = src.s[i];
      out = dst.p[i];
    }
  const Taps<R>& taps = (TWO && c == 1) ? taps1 : taps0;
  const BorderScale& bs = (TWO && c == 1) ? bs1 : bs0;
  const int x0 = bid.x * HW, y0 = bid.y * HH;
  const int tid = threadIdx.x;
  // Interior tile (no border column, every staged sample inside the image, rows 16-byte
  // aligned): staged with aligned 16-byte loads; every thread produces 4 consecutive outputs
  // of one row from a register window of the staged samples -- each sample is read from LDS
  // once per thread instead of once per tap -- and stores them as one 16-byte vector.  Same
  // arithmetic per output: ascending taps from 0.0f.
  if (x0 >= RA && x0 + HW + RA <= w && y0 + HH <= h && (pitch & 3) == 0) {
    constexpr int NV = HH * (TP / 4);
#pragma unroll
    for (int k = 0; k < (NV + 255) / 256; ++k) {
      const int i = 256 * k + tid;
      if (256 * k + 255 < NV || i < NV) {
        const int ry = i / (TP / 4), q = i - ry * (TP / 4);
        const gz_f4 v = s.load4((size_t)(y0 + ry) * pitch + (x0 - RA + 4 * q));
        *reinterpret_cast<gz_f4*>(&tile[ry][4 * q]) = v;
      }
    }
    __syncthreads();
    const int ry = tid >> 6, xq = (tid & 63) * 4;
    float win[4 + 2 * RA];
#pragma unroll
    for (int i = 0; i < (4 + 2 * RA) / 4; ++i) {
      gz_f4 v = *reinterpret_cast<const gz_f4*>(&tile[ry][xq + 4 * i]);
      if (OFF > 0 && (i == 0 || i == (4 + 2 * RA) / 4 - 1)) GZ_KEEP_F4(v);   // (the ends no tap touches)
      win[4 * i] = v.v[0]; win[4 * i + 1] = v.v[1]; win[4 * i + 2] = v.v[2]; win[4 * i + 3] = v.v[3];
    }
    gz_f4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float sum = 0.0f;
#pragma unroll
      for (int j = 0; j <= 2 * R; ++j) sum += win[OFF + i + j] * taps.ks[j];
      o.v[i] = sum;
    }
    GZ_STG4(out, (size_t)(y0 + ry) * pitch + x0 + xq, o);
    return;
  }
  // generic tile: rows y0..y0+HH-1, columns x0-R .. x0+HW+R-1 (zero outside the image)
  for (int i = tid; i < HH * (HW + 2 * R); i += 256) {
    const int ry = i / (HW + 2 * R), rx = i - ry * (HW + 2 * R);
    const int x = x0 - R + rx, y = y0 + ry;
    float v = 0.0f;
    if (x >= 0 && x < w && y < h) v = s((size_t)y * pitch + x);
    tile[ry][rx] = v;
  }
  __syncthreads();
  const int x = x0 + tid;
  if (x >= w) return;
  const bool border = x < R || x >= w - R;
  float scale = 1.0f;
  if (border) scale = x < R ? bs.lo[x] : bs.hi[w - 1 - x];
#pragma unroll
  for (int ry = 0; ry < HH; ++ry) {
    const int y = y0 + ry;
    if (y >= h) break;
    float sum = 0.0f;
    if (!border) {
#pragma unroll
      for (int j = 0; j <= 2 * R; ++j) sum += tile[ry][tid + j] * taps.ks[j];
    } else {
#pragma unroll
      for (int j = 0; j <= 2 * R; ++j) sum += tile[ry][tid + j] * taps.k[j];
      sum = sum * scale;
    }
    out[(size_t)y * pitch + x] = sum;
  }
}

// Row pass with packed arithmetic: a workgroup takes HP = 8 rows as 4 row PAIRS; a thread
// produces 4 consecutive outputs of both rows of a pair, every multiply and every add as one
// packed instruction on (row 2p, row 2p + 1).  The pair's samples are staged interleaved,
// tile[p][x][2], so that one 16-byte LDS read yields two (upper, lower) pairs ready for use.
// Per output the operations and their order are those of k_blur_h: f32, ascending taps from
// 0.0f.  Tiles that are not interior take k_blur_h's generic path (8 rows of it).
constexpr int HP = 8;

template <int R, class Src, int NC, bool TWO = false>
__global__ __launch_bounds__(256) void k_blur_h_pk(SrcPack<Src, NC> src, PlanePack<NC> dst,
                                                   int w, int h, int pitch, Taps<R> taps0,
                                                   BorderScale bs0, Taps<R> taps1, BorderScale bs1) {
  constexpr int RA = (R + 3) & ~3;
  constexpr int TP = HW + 2 * RA;
  constexpr int OFF = RA - R;
  // Conflict-free window reads.  A thread's 16-byte reads walk its window in steps of one slot,
  // but neighbouring threads start 2 slots apart (4 outputs x 2 rows), so with one row pair x 256
  // columns per wavefront only 8 distinct slots (mod 16) are touched by the 16 lanes of a
  // ds_read_b128 group ({0-3, 12-15, 20-27}, ...) and every read takes twice its cycles
  // (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.32-0.63 in profiles/r02_compare_4k_sq_counters.csv).
  // A wavefront therefore takes TWO row pairs x 128 columns -- lanes with bit 3 clear the first
  // pair, set the second -- and consecutive pairs are staged an odd number of slots apart (4
  // floats of padding): each group then reads 8 even and 8 odd slots.  Same outputs per thread.
  constexpr int PS = TP * 2 + 4;   // floats per staged row pair
  __shared__ __attribute__((aligned(16))) float tile[(HP / 2) * (TP * 2 + 4)];
  const GzTile bid = gz_xcd_tile();
  const int c = bid.z;
  Src s = src.s[0];
  float* __restrict__ out = dst.p[0];
#pragma unroll
  for (int i = 1; i < NC; ++i)
    if (c == i) {
      s = src.s[i];
      out = dst.p[i];
    }
  const Taps<R>& taps = (TWO && c == 1) ? taps1 : taps0;
  const BorderScale& bs = (TWO && c == 1) ? bs1 : bs0;
  const int x0 = bid.x * HW, y0 = bid.y * HP;
  const int tid = threadIdx.x;
  if (x0 >= RA && x0 + HW + RA <= w && y0 + HP <= h && (pitch & 3) == 0) {
    constexpr int NQ = TP / 4;          // 16-byte vectors per staged row
    constexpr int NI = (HP / 2) * NQ;   // (pair, vector) items
#pragma unroll
    for (int k = 0; k < (NI + 255) / 256; ++k) {
      const int i = 256 * k + tid;
      if (256 * k + 255 < NI || i < NI) {
        const int p = i / NQ, q = i - p * NQ;
        const size_t g = (size_t)(y0 + 2 * p) * pitch + (x0 - RA + 4 * q);
        const gz_f4 a = s.load4(g), b = s.load4(g + pitch);
        gz_f4 lo, hi;
        lo.v[0] = a.v[0]; lo.v[1] = b.v[0]; lo.v[2] = a.v[1]; lo.v[3] = b.v[1];
        hi.v[0] = a.v[2]; hi.v[1] = b.v[2]; hi.v[2] = a.v[3]; hi.v[3] = b.v[3];
        float* t = &tile[p * PS + 4 * q * 2];
        *reinterpret_cast<gz_f4*>(t) = lo;
        *reinterpret_cast<gz_f4*>(t + 4) = hi;
      }
    }
    __syncthreads();
    const int wv = tid >> 6, l = tid & 63;
    const int p = 2 * (wv >> 1) + ((l >> 3) & 1);
    const int xq = ((wv & 1) * 32 + (((l >> 4) << 3) | (l & 7))) * 4;
    gz_f2 win[4 + 2 * RA];
#pragma unroll
    for (int i = 0; i < (4 + 2 * RA) / 2; ++i) {
      gz_f4 v = *reinterpret_cast<const gz_f4*>(&tile[p * PS + (xq + 2 * i) * 2]);
      if (OFF > 0 && (i == 0 || i == (4 + 2 * RA) / 2 - 1)) GZ_KEEP_F4(v);   // (the ends no tap touches)
      win[2 * i] = gz_f2{v.v[0], v.v[1]};
      win[2 * i + 1] = gz_f2{v.v[2], v.v[3]};
    }
    gz_f4 oa, ob;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      gz_f2 sum = gz_f2_splat(0.0f);
#pragma unroll
      for (int j = 0; j <= 2 * R; ++j) sum = sum + win[OFF + i + j] * gz_f2_splat(taps.ks[j]);
      oa.v[i] = sum[0];
      ob.v[i] = sum[1];
    }
    const size_t o = (size_t)(y0 + 2 * p) * pitch + x0 + xq;
    GZ_STG4(out, o, oa);
    GZ_STG4(out, o + pitch, ob);
    return;
  }
  // generic tile: rows y0..y0+HP-1, columns x0-R .. x0+HW+R-1 (zero outside the image)
  constexpr int GW = HW + 2 * R;
  for (int i = tid; i < HP * GW; i += 256) {
    const int ry = i / GW, rx = i - ry * GW;
    const int x = x0 - R + rx, y = y0 + ry;
    float v = 0.0f;
    if (x >= 0 && x < w && y < h) v = s((size_t)y * pitch + x);
    tile[ry * GW + rx] = v;
  }
  __syncthreads();
  const int x = x0 + tid;
  if (x >= w) return;
  const bool border = x < R || x >= w - R;
  float scale = 1.0f;
  if (border) scale = x < R ? bs.lo[x] : bs.hi[w - 1 - x];
#pragma unroll 1
  for (int ry = 0; ry < HP; ++ry) {
    const int y = y0 + ry;
    if (y >= h) break;
    const float* row = &tile[ry * GW + tid];
    float sum = 0.0f;
    if (!border) {
#pragma unroll
      for (int j = 0; j <= 2 * R; ++j) sum += row[j] * taps.ks[j];
    } else {
#pragma unroll
      for (int j = 0; j <= 2 * R; ++j) sum += row[j] * taps.k[j];
      sum = sum * scale;
    }
    out[(size_t)y * pitch + x] = sum;
  }
}

// --------------------------------------------------------------------- column pass --
// grid = (ceil(w/VW), ceil(h/VH)); block = 256 = 64 columns x 4 row groups; each thread
// produces VH/4 outputs of its column for every one of the NC planes, then hands the NC
// blurred values of each pixel to the Post functor (which may read/write other planes).
constexpr int VW = 64;
constexpr int VH = 64;
constexpr int VPT = VH / 4;   // outputs per thread

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
  // k_blur2d only, optional: the launch covers the LISTED tiles (tile = y * ceil(w / 64) + x, grid.x = their
  // number) instead of the image -- the opsin planes patched around the blocks the serial steps edited (chain.h,
  // opsin_ahead).  A tile's results do not depend on which launch computes it.
  const int* tiles;
};

// Per-8x8-block maxima and image maximum of the tile's results, straight from registers: a
// thread holds NR consecutive rows of one column (NR a multiple of 8, the tile origin is
// 8-aligned), i.e. one column of NR/8 blocks; the 8 columns of a block are 8 adjacent lanes.
// Values are >= 0, so their bit patterns order like the floats.
template <int NR>
GZ_DEVFN void block_max_from_registers(const float* res, int tx, int row0, int x0, int y0, int w,
                                       int h, const BlockMaxOut& bm) {
  int wave_max = 0;
#pragma unroll
  for (int g = 0; g < NR / 8; ++g) {
    float m = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) m = res[8 * g + i] > m ? res[8 * g + i] : m;
    int bits = (int)__float_as_uint(m);
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) {
      const int o = __shfl(bits, tx ^ d);
      bits = o > bits ? o : bits;
    }
    const int gbx = (x0 + tx) / 8, gby = (y0 + row0) / 8 + g;
    if ((tx & 7) == 0 && bm.block_max && 8 * gbx < w && 8 * gby < h)
      bm.block_max[gby * bm.bw + gbx] = __uint_as_float((unsigned)bits);
    wave_max = bits > wave_max ? bits : wave_max;
  }
#pragma unroll
  for (int d = 8; d < 64; d <<= 1) {
    const int o = __shfl(wave_max, tx ^ d);
    wave_max = o > wave_max ? o : wave_max;
  }
  // one atomic per workgroup (thousands of atomics on one address serialise in L2)
  __shared__ int s_wave_max[4];
  if (tx == 0) s_wave_max[threadIdx.x >> 6] = wave_max;
  __syncthreads();
  if (threadIdx.x == 0) {
    int m = s_wave_max[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) m = s_wave_max[i] > m ? s_wave_max[i] : m;
    // the maximum only grows: a workgroup that cannot raise the value it sees skips the atomic
    if ((unsigned)m > *(volatile unsigned*)bm.image_max_bits) atomicMax(bm.image_max_bits, (unsigned)m);
  }
}

// The same for a thread that holds FOUR consecutive rows of its column (16-row tiles: wavefront tg has
// rows 4 tg .. 4 tg + 3, so a block's eight rows are two wavefronts'): the column maxima of the upper
// and lower halves meet in LDS.  max is exact and order-free.
GZ_DEVFN void block_max_from_half_columns(const float* res, int tx, int tg, int x0, int y0, int w, int h,
                                          const BlockMaxOut& bm) {
  __shared__ int s_half[4][64];
  __shared__ int s_wave_max4[2];
  float m = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) m = res[i] > m ? res[i] : m;
  s_half[tg][tx] = (int)__float_as_uint(m);
  __syncthreads();
  int wave_max = 0;
  if ((tg & 1) == 0) {   // wavefronts 0 and 2: block rows 0 and 1 of the tile
    const int a = s_half[tg][tx], b = s_half[tg + 1][tx];
    int bits = a > b ? a : b;
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) {
      const int o = __shfl(bits, tx ^ d);
      bits = o > bits ? o : bits;
    }
    const int gbx = (x0 + tx) / 8, gby = y0 / 8 + (tg >> 1);
    if ((tx & 7) == 0 && bm.block_max && 8 * gbx < w && 8 * gby < h)
      bm.block_max[gby * bm.bw + gbx] = __uint_as_float((unsigned)bits);
    wave_max = bits;
#pragma unroll
    for (int d = 8; d < 64; d <<= 1) {
      const int o = __shfl(wave_max, tx ^ d);
      wave_max = o > wave_max ? o : wave_max;
    }
    if (tx == 0) s_wave_max4[tg >> 1] = wave_max;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int mm = s_wave_max4[0] > s_wave_max4[1] ? s_wave_max4[0] : s_wave_max4[1];
    if ((unsigned)mm > *(volatile unsigned*)bm.image_max_bits) atomicMax(bm.image_max_bits, (unsigned)mm);
  }
}

// ZCH = true (NC == 2, Post = PostStore<2>): the two planes are independent blurs with their own
// taps (taps1 / bs1 for the second); the grid's z index picks the plane, each workgroup does one.
template <int R, int NC, class Post, int TH, bool ZCH = false>
__global__ __launch_bounds__(256) void k_blur_v_compact(CPlanePack<NC> src, Post post, int w, int h,
                                                int pitch, Taps<R> taps0, BorderScale bs0,
                                                Taps<R> taps1, BorderScale bs1) {
  // TH = tile height (32, or 16 for small images: twice the workgroups to fill the chip)
  constexpr int VHt = TH, VPTt = TH / 4;
  constexpr int NCL = ZCH ? 1 : NC;   // planes per workgroup
  static_assert(!ZCH || NC == 2, "independent planes: two of them");
  __shared__ __attribute__((aligned(16))) float tile[VHt + 2 * R][VW];
  const int tx = threadIdx.x & 63, tg = threadIdx.x >> 6;
  const GzTile bid = gz_xcd_tile();
  const Taps<R>& taps = (ZCH && bid.z == 1) ? taps1 : taps0;
  const BorderScale& bs = (ZCH && bid.z == 1) ? bs1 : bs0;
  const int x0 = bid.x * VW, y0 = bid.y * VHt;
  const int x = x0 + tx;
  // staging with one aligned 16-byte load per lane when the tile's columns are all inside
  // the image and rows are 16-byte aligned (rows outside the image are zero)
  const bool vec = x0 + VW <= w && (pitch & 3) == 0;
  const int vq = (threadIdx.x & 15) * 4, vr = threadIdx.x >> 4;
  // Compact code on purpose: the channel and row loops stay loops (only the taps are unrolled):
  // 5 KB of instructions instead of the 34 KB of the unrolled form for <16, 3> -- every launch
  // starts with cold instruction caches, and some boxes miss three times as often
  // (profiles/r01_sq_counters_*_box.csv).  Tiles whose columns are all inside the image keep their
  // results in LDS and hand them to the Post functor by quads: 4 consecutive pixels of a row per
  // thread, 16-byte accesses (a dword per lane streams at two thirds of that rate); the other
  // tiles go pixel by pixel (one plane: straight from the row loop; several: through LDS).
  const bool quads = vec;
  __shared__ __attribute__((aligned(16))) float outv[NCL][VHt][VW];
#pragma unroll 1
  for (int c = 0; c < NCL; ++c) {
    const float* __restrict__ in = src.p[0];
#pragma unroll
    for (int k = 1; k < NC; ++k)
      if ((ZCH ? bid.z : c) == k) in = src.p[k];
    if (c > 0) __syncthreads();
    if (vec) {
#pragma unroll 1
      for (int k = 0; k < (VHt + 2 * R + 15) / 16; ++k) {
        const int ry = vr + 16 * k;
        if (ry < VHt + 2 * R) {
          const int y = y0 - R + ry;
          gz_f4 v;
          v.v[0] = v.v[1] = v.v[2] = v.v[3] = 0.0f;
          if (y >= 0 && y < h) v = GZ_LDG4(in, (size_t)y * pitch + x0 + vq);
          *reinterpret_cast<gz_f4*>(&tile[ry][vq]) = v;
        }
      }
    } else {
      for (int ry = tg; ry < VHt + 2 * R; ry += 4) {
        const int y = y0 - R + ry;
        float v = 0.0f;
        if (x < w && y >= 0 && y < h) v = in[(size_t)y * pitch + x];
        tile[ry][tx] = v;
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int i = 0; i < VPTt; ++i) {
      const int ly = tg * VPTt + i;   // local output row
      const int y = y0 + ly;
      float sum = 0.0f;
      if (y < h) {
        const bool border = y < R || y >= h - R;
        const float* col = &tile[ly][tx];
        if (!border) {
#pragma unroll
          for (int j = 0; j <= 2 * R; ++j) sum += col[j * VW] * taps.ks[j];
        } else {
#pragma unroll
          for (int j = 0; j <= 2 * R; ++j) sum += col[j * VW] * taps.k[j];
          sum = sum * (y < R ? bs.lo[y] : bs.hi[h - 1 - y]);
        }
      }
      if (quads) {
        outv[c][ly][tx] = sum;
      } else if constexpr (ZCH) {
        float* __restrict__ o = bid.z == 1 ? post.out[1] : post.out[0];
        if (x < w && y < h) o[(size_t)y * pitch + x] = sum;
      } else if (NC > 1) {
        outv[c][ly][tx] = sum;
      } else {
        float v1[1] = {sum};
        if (x < w && y < h) post((size_t)y * pitch + x, v1);
      }
    }
  }
  if (quads) {
    __syncthreads();
#pragma unroll 1
    for (int q = threadIdx.x; q < VHt * (VW / 4); q += 256) {
      const int row = q / (VW / 4), c4 = (q % (VW / 4)) * 4;
      const int y = y0 + row;
      if (y < h) {
        const size_t idx = (size_t)y * pitch + x0 + c4;
        gz_f4 v[NCL];
#pragma unroll
        for (int c = 0; c < NCL; ++c) v[c] = *reinterpret_cast<const gz_f4*>(&outv[c][row][c4]);
        if constexpr (ZCH) {
          float* __restrict__ o = bid.z == 1 ? post.out[1] : post.out[0];
          GZ_STG4(o, idx, v[0]);
        } else {
          post.quad(idx, v);
        }
      }
    }
    return;
  }
  // (each lane reads back what it wrote itself: no barrier needed)
  if constexpr (!ZCH && NC > 1) {
#pragma unroll 1
    for (int i = 0; i < VPTt; ++i) {
      const int ly = tg * VPTt + i;
      const int y = y0 + ly;
      if (x < w && y < h) {
        float v[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) v[c] = outv[c][ly][tx];
        (void)post((size_t)y * pitch + x, v);
      }
    }
  }
}


// Column pass with packed arithmetic: a thread takes two adjacent columns (an 8-byte LDS
// read yields the pair) and TH/8 consecutive rows from a register window of row pairs; every
// multiply and add is one packed instruction on (column x, column x + 1).  Tiles whose output
// rows are all interior and whose columns are all inside the image take this path, the
// others the per-output path with border handling (as a loop: it is rarely taken).
template <int R, int NC, class Post, int TH, bool ZCH = false>
__global__ __launch_bounds__(256) void k_blur_v_pk(CPlanePack<NC> src, Post post, int w, int h,
                                                   int pitch, Taps<R> taps0, BorderScale bs0,
                                                   Taps<R> taps1, BorderScale bs1) {
  constexpr int RPT = TH / 8;    // rows per thread on the packed path
  constexpr int VPTt = TH / 4;   // rows per thread on the generic path
  constexpr int NCL = ZCH ? 1 : NC;   // planes per workgroup (ZCH: see k_blur_v_compact)
  static_assert(!ZCH || NC == 2, "independent planes: two of them");
  __shared__ __attribute__((aligned(16))) float tile[TH + 2 * R][VW];
  const int tid = threadIdx.x;
  const int tx = tid & 63, tg = tid >> 6;
  const GzTile bid = gz_xcd_tile();
  const Taps<R>& taps = (ZCH && bid.z == 1) ? taps1 : taps0;
  const BorderScale& bs = (ZCH && bid.z == 1) ? bs1 : bs0;
  const int x0 = bid.x * VW, y0 = bid.y * TH;
  const bool vec = x0 + VW <= w && (pitch & 3) == 0;
  const bool inner = vec && y0 >= R && y0 + TH + R <= h;
  const int vq = (tid & 15) * 4, vr = tid >> 4;
  if (inner) {
    const int cp = tid & 31, rg = tid >> 5;
    gz_f2 acc[NCL][RPT];
#pragma unroll
    for (int c = 0; c < NCL; ++c) {
      const float* __restrict__ in = src.p[c];
      if constexpr (ZCH) in = bid.z == 1 ? src.p[1] : src.p[0];
      if (c > 0) __syncthreads();
#pragma unroll
      for (int k = 0; k < (TH + 2 * R + 15) / 16; ++k) {
        const int ry = vr + 16 * k;
        if ((k + 1) * 16 <= TH + 2 * R || ry < TH + 2 * R) {
          const gz_f4 v = GZ_LDG4(in, (size_t)(y0 - R + ry) * pitch + x0 + vq);
          *reinterpret_cast<gz_f4*>(&tile[ry][vq]) = v;
        }
      }
      __syncthreads();
      gz_f2 win[RPT + 2 * R];
#pragma unroll
      for (int i = 0; i < RPT + 2 * R; ++i)
        win[i] = *reinterpret_cast<const gz_f2*>(&tile[rg * RPT + i][2 * cp]);
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        gz_f2 sum = gz_f2_splat(0.0f);
#pragma unroll
        for (int j = 0; j <= 2 * R; ++j) sum = sum + win[i + j] * gz_f2_splat(taps.ks[j]);
        acc[c][i] = sum;
      }
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const size_t idx = (size_t)(y0 + rg * RPT + i) * pitch + x0 + 2 * cp;
      if constexpr (ZCH) {
        float* __restrict__ o = bid.z == 1 ? post.out[1] : post.out[0];
        GZ_STG2(o, idx, acc[0][i]);   // (idx is even: the pitch is a multiple of 4 on this path)
      } else {
        float v0[NC], v1[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          v0[c] = acc[c][i][0];
          v1[c] = acc[c][i][1];
        }
        post.pair(idx, v0, v1);
      }
    }
    return;
  }
  const int x = x0 + tx;
  __shared__ float outv[NCL > 1 ? NCL : 1][NCL > 1 ? TH : 1][VW];
#pragma unroll 1
  for (int c = 0; c < NCL; ++c) {
    const float* __restrict__ in = src.p[0];
#pragma unroll
    for (int k = 1; k < NC; ++k)
      if ((ZCH ? bid.z : c) == k) in = src.p[k];
    if (c > 0) __syncthreads();
    for (int ry = tg; ry < TH + 2 * R; ry += 4) {
      const int y = y0 - R + ry;
      float v = 0.0f;
      if (x < w && y >= 0 && y < h) v = in[(size_t)y * pitch + x];
      tile[ry][tx] = v;
    }
    __syncthreads();
#pragma unroll 1
    for (int i = 0; i < VPTt; ++i) {
      const int ly = tg * VPTt + i;
      const int y = y0 + ly;
      float sum = 0.0f;
      if (y < h) {
        const bool border = y < R || y >= h - R;
        const float* col = &tile[ly][tx];
        if (!border) {
#pragma unroll
          for (int j = 0; j <= 2 * R; ++j) sum += col[j * VW] * taps.ks[j];
        } else {
#pragma unroll
          for (int j = 0; j <= 2 * R; ++j) sum += col[j * VW] * taps.k[j];
          sum = sum * (y < R ? bs.lo[y] : bs.hi[h - 1 - y]);
        }
      }
      if constexpr (ZCH) {
        float* __restrict__ o = bid.z == 1 ? post.out[1] : post.out[0];
        if (x < w && y < h) o[(size_t)y * pitch + x] = sum;
      } else if (NC > 1) {
        outv[c][ly][tx] = sum;
      } else {
        float v1[1] = {sum};
        if (x < w && y < h) (void)post((size_t)y * pitch + x, v1);
      }
    }
  }
  if constexpr (NCL > 1) {
#pragma unroll 1
    for (int i = 0; i < VPTt; ++i) {
      const int ly = tg * VPTt + i;
      const int y = y0 + ly;
      if (x < w && y < h) {
        float v[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) v[c] = outv[c][ly][tx];   // (own writes)
        (void)post((size_t)y * pitch + x, v);
      }
    }
  }
}

// ------------------------------------------------------ fused row + column pass (2-D) --
// Blur = Convolution along x, then along y (butteraugli.cc:229-233), for one 64x64 output
// tile per workgroup without the intermediate plane ever leaving the chip:
//   1. the (64 + 2R) x (64 + 2RA) input tile (RA = R rounded up to 4, so that every row is
//      staged with aligned 16-byte loads) goes to LDS, with the source functor applied;
//   2. row pass, in place: a thread takes 4 consecutive outputs of a row from a register
//      window of the staged samples and writes them over the row's first 64 columns (a row is
//      only ever touched by lanes of one wave, whose loads precede its stores);
//   3. column pass from LDS: lane = column, 16 consecutive rows per thread from a register
//      window of 16 + 2R row-pass results; the NC blurred values of each pixel go to the Post
//      functor.
// Each output is computed with exactly the operations of the separate passes: f32, taps in
// ascending order from 0.0f; interior samples use the pre-scaled taps, border samples the raw
// taps and one multiply by the host-computed scale.  Tiles that touch the image border
// region (or images whose row pitch is not a multiple of 4) take a generic path with
// per-output border handling.  Compared with two passes this saves writing and re-reading
// the intermediate plane (2 of 4 plane passes per blur) at the price of (64+2R)/64 x the row
// pass arithmetic.
constexpr int T2 = 64;   // tile edge

// Without block maxima (BM = false: every blur but the chain's last) the channel loop stays a loop
// and the per-channel results wait in LDS instead of registers (ROLL): the code shrinks by about
// NC x (opsin blur 40 -> 6 KB, MF 32 -> 10, HF 23 -> 6), and the results leave by quads.
// (256, 4): registers for four wavefronts per SIMD -- the MF instance (PostMF's quad epilogue) took
// 143 VGPRs = three per SIMD without the bound; 1080p chain 0.358-0.362 -> 0.350 ms, 4K -0.4 %
// (profiles/r04_occupancy_experiments.log; five per SIMD: no further gain).
template <int R, int NC, class Src, class Post, bool BM, int TH>
__global__ __launch_bounds__(256, 4) void k_blur2d(SrcPack<Src, NC> src, Post post, int w, int h,
                                                int pitch, Taps<R> taps, BorderScale bsx,
                                                BorderScale bsy, BlockMaxOut bm) {
  constexpr bool ROLL = !BM;
  constexpr int RA = (R + 3) & ~3;
  constexpr int IW = T2 + 2 * RA;   // staged columns, multiple of 4
  constexpr int IH = TH + 2 * R;    // staged rows (TH = tile height: 64, or 32 for small images)
  constexpr int VPTt = TH / 4;      // column-pass outputs per thread
  constexpr int OFF = RA - R;       // window offset inside the aligned row
  __shared__ __attribute__((aligned(16))) float tile[IH][IW];
  const int tid = threadIdx.x;
  GzTile bid;
  if (bm.tiles) {
    const int t = bm.tiles[blockIdx.x], gx = (w + T2 - 1) / T2;
    bid.x = t % gx; bid.y = t / gx; bid.z = 0;
  } else {
    bid = gz_xcd_tile();
  }
  const int x0 = bid.x * T2, y0 = bid.y * TH;
  const bool interior = x0 >= RA && x0 + T2 + RA <= w && y0 >= R && y0 + TH + R <= h &&
                        (pitch & 3) == 0;
  const int tx = tid & 63, tg = tid >> 6;   // column pass: lane = column, wave = row group
  // Row pass: a thread takes 4 columns of rows hr + 16k; 16 lanes share a row.  A wave's
  // ds_read_b128 is served in four groups of 16 lanes that mix two ADJACENT rows
  // (MI355X_MICROARCH.md, LDS: {0-3, 12-15, 20-27}, ...); a row is IW / 4 = 18 or 20 sixteen-byte
  // slots, so with the plain mapping the second row's slots land 2 or 4 (mod 16) beyond the
  // first row's and collide with them (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.44-0.55,
  // profiles/r02_compare_4k_sq_counters.csv).  The odd rows therefore take their column quads
  // rotated back by that amount: every group then covers 16 distinct slots.  Which lane computes
  // which quad of a row changes nothing in the results.
  constexpr int kRowSlotShift = (IW / 4) & 15;
  const int hr = tid >> 4;                  // rows hr + 16k
  const int hq = (((tid & 15) - (hr & 1) * kRowSlotShift) & 15) * 4;
  float acc[ROLL ? 1 : NC][VPTt];
  __shared__ __attribute__((aligned(16))) float outv[ROLL ? NC : 1][ROLL ? TH : 1][T2];
  constexpr int kChannelUnroll = ROLL ? 1 : NC;
#pragma unroll kChannelUnroll
  for (int c = 0; c < NC; ++c) {
    Src s = src.s[0];   // constant indices into the kernel arguments
#pragma unroll
    for (int k = 1; k < NC; ++k)
      if (c == k) s = src.s[k];
    if (c > 0) __syncthreads();   // the column pass of the previous plane is done with the tile
    if (interior) {
#ifdef GZ_TAPS_VGPR
      float kv[2 * R + 1];
#pragma unroll
      for (int j = 0; j <= 2 * R; ++j) kv[j] = GZ_IN_VGPR(taps.ks[j]);
#else
      const float* kv = taps.ks;
#endif
      // ---- stage: aligned 16-byte loads, all of a thread's loads in flight before the
      // first LDS store (compile-time trip counts: the loads are issued back to back)
      constexpr int NV = IH * (IW / 4);            // 16-byte vectors in the tile
      constexpr int SB = 8;                        // vectors per thread per batch
#pragma unroll
      for (int b0 = 0; b0 < NV; b0 += 256 * SB) {
        gz_f4 buf[SB];
#pragma unroll
        for (int k = 0; k < SB; ++k) {
          const int i = b0 + 256 * k + tid;
          if (b0 + 256 * k + 255 < NV || i < NV) {
            const int ry = i / (IW / 4), q = i - ry * (IW / 4);
            buf[k] = s.load4((size_t)(y0 - R + ry) * pitch + (x0 - RA + 4 * q));
          }
        }
#pragma unroll
        for (int k = 0; k < SB; ++k) {
          const int i = b0 + 256 * k + tid;
          if (b0 + 256 * k + 255 < NV || i < NV) {
            const int ry = i / (IW / 4), q = i - ry * (IW / 4);
            *reinterpret_cast<gz_f4*>(&tile[ry][4 * q]) = buf[k];
          }
        }
      }
      __syncthreads();
      // ---- row pass (pre-scaled taps), in place: the 16 lanes that share a row read their
      // windows before any of them stores (one wavefront, one instruction stream)
#pragma unroll 1
      for (int k = 0; k < (IH + 15) / 16; ++k) {   // (same trip count for every lane)
        const int ry = hr + 16 * k;
        const bool on = ry < IH;
        float win[4 + 2 * RA];
        if (on) {
#pragma unroll
          for (int i = 0; i < (4 + 2 * RA) / 4; ++i) {
            const gz_f4 v = *reinterpret_cast<const gz_f4*>(&tile[ry][hq + 4 * i]);
            win[4 * i] = v.v[0]; win[4 * i + 1] = v.v[1]; win[4 * i + 2] = v.v[2]; win[4 * i + 3] = v.v[3];
          }
        }
        GZ_WAVE_LOCKSTEP();
        if (on) {
          gz_f4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float sum = 0.0f;
#pragma unroll
            for (int j = 0; j <= 2 * R; ++j) sum += win[OFF + i + j] * kv[j];
            o.v[i] = sum;
          }
          *reinterpret_cast<gz_f4*>(&tile[ry][hq]) = o;
        }
      }
      __syncthreads();
      // ---- column pass (pre-scaled taps)
      float win[VPTt + 2 * R];
#pragma unroll
      for (int i = 0; i < VPTt + 2 * R; ++i) win[i] = tile[tg * VPTt + i][tx];
#pragma unroll
      for (int i = 0; i < VPTt; ++i) {
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j <= 2 * R; ++j) sum += win[i + j] * kv[j];
        if (ROLL) outv[c][tg * VPTt + i][tx] = sum; else acc[c][i] = sum;
      }
    } else {
      // ---- generic tile: zero outside the image, per-output border handling
      for (int i = tid; i < IH * IW; i += 256) {
        const int ry = i / IW, rx = i - ry * IW;
        const int x = x0 - RA + rx, y = y0 - R + ry;
        float v = 0.0f;
        if (x >= 0 && x < w && y >= 0 && y < h) v = s((size_t)y * pitch + x);
        tile[ry][rx] = v;
      }
      __syncthreads();
#pragma unroll 1
      for (int k = 0; k < (IH + 15) / 16; ++k) {   // (same trip count for every lane)
        const int ry = hr + 16 * k;
        const bool on = ry < IH;
        float win[4 + 2 * R];
        if (on) {
#pragma unroll
          for (int i = 0; i < 4 + 2 * R; ++i) win[i] = tile[ry][hq + OFF + i];
        }
        GZ_WAVE_LOCKSTEP();
        if (on) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int x = x0 + hq + i;
            float sum = 0.0f;
            if (x < w) {
              const bool border = x < R || x >= w - R;
              if (!border) {
#pragma unroll
                for (int j = 0; j <= 2 * R; ++j) sum += win[i + j] * taps.ks[j];
              } else {
#pragma unroll
                for (int j = 0; j <= 2 * R; ++j) sum += win[i + j] * taps.k[j];
                sum = sum * (x < R ? bsx.lo[x] : bsx.hi[w - 1 - x]);
              }
            }
            tile[ry][hq + i] = sum;
          }
        }
      }
      __syncthreads();
      // (ROLL: the column loop of this rarely taken path stays a loop, results go to LDS)
      constexpr int kColUnroll = ROLL ? 1 : VPTt;
#pragma unroll kColUnroll
      for (int i = 0; i < VPTt; ++i) {
        const int ly = tg * VPTt + i;
        const int y = y0 + ly;
        float sum = 0.0f;
        if (y < h) {
          const bool border = y < R || y >= h - R;
          const float* col = &tile[ly][tx];
          if (!border) {
#pragma unroll
            for (int j = 0; j <= 2 * R; ++j) sum += col[j * IW] * taps.ks[j];
          } else {
#pragma unroll
            for (int j = 0; j <= 2 * R; ++j) sum += col[j * IW] * taps.k[j];
            sum = sum * (y < R ? bsy.lo[y] : bsy.hi[h - 1 - y]);
          }
        }
        if (ROLL) outv[c][ly][tx] = sum; else acc[c][i] = sum;
      }
    }
  }
  const int x = x0 + tx;
  if (ROLL && !BM) {
    if (x0 + T2 <= w && (pitch & 3) == 0) {
      // Epilogue by quads: a thread takes 4 consecutive pixels of a row from the results in
      // LDS, so that everything the Post functor reads and writes moves as 16-byte accesses
      // (a dword per lane streams at two thirds of that rate, tools/ubench/bw.hip).
      __syncthreads();
#pragma unroll 1
      for (int q = tid; q < TH * (T2 / 4); q += 256) {
        const int row = q / (T2 / 4), c4 = (q % (T2 / 4)) * 4;
        const int y = y0 + row;
        if (y < h) {
          gz_f4 v[NC];
#pragma unroll
          for (int c = 0; c < NC; ++c) v[c] = *reinterpret_cast<const gz_f4*>(&outv[c][row][c4]);
          post.quad((size_t)y * pitch + x0 + c4, v);
        }
      }
      return;
    }
    // the Post functor once in the code, not once per row of the thread
#pragma unroll 1
    for (int i = 0; i < VPTt; ++i) {
      const int y = y0 + tg * VPTt + i;
      if (x < w && y < h) {
        float v[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) v[c] = outv[c][tg * VPTt + i][tx];   // (own writes)
        (void)post((size_t)y * pitch + x, v);
      }
    }
    return;
  }
  float res[VPTt];
#pragma unroll
  for (int i = 0; i < VPTt; ++i) {
    const int y = y0 + tg * VPTt + i;
    res[i] = 0.0f;
    if (x < w && y < h) {
      float v[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) v[c] = ROLL ? outv[c][tg * VPTt + i][tx] : acc[c][i];   // (own writes)
      res[i] = post((size_t)y * pitch + x, v);
    }
  }
  if constexpr (BM && VPTt == 4) block_max_from_half_columns(res, tx, tg, x0, y0, w, h, bm);
  else if constexpr (BM) block_max_from_registers<VPTt>(res, tx, tg * VPTt, x0, y0, w, h, bm);
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
  // four consecutive pixels of a row at once (idx a multiple of 4): 16-byte accesses
  GZ_DEVFN void quad(size_t idx, const gz_f4* v) const {
#pragma unroll
    for (int c = 0; c < NC; ++c) GZ_STG4(out[c], idx, v[c]);
  }
  // two consecutive pixels (idx even): 8-byte stores
  GZ_DEVFN void pair(size_t idx, const float* v0, const float* v1) const {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const gz_f2 t = {v0[c], v1[c]};
      GZ_STG2(out[c], idx, t);
    }
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
  GZ_DEVFN void pair(size_t idx, const float* v0, const float* v1) const {
    (void)(*this)(idx, v0);
    (void)(*this)(idx + 1, v1);
  }
  GZ_DEVFN void quad(size_t idx, const gz_f4* v) const {
    const gz_f4 l0 = GZ_LDG4(lin[0], idx), l1 = GZ_LDG4(lin[1], idx), l2 = GZ_LDG4(lin[2], idx);
    gz_f4 x, y, z;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      opsin_pixel(v[0].v[i], v[1].v[i], v[2].v[i], l0.v[i], l1.v[i], l2.v[i], &x.v[i], &y.v[i], &z.v[i]);
    GZ_STG4(xyb[0], idx, x);
    GZ_STG4(xyb[1], idx, y);
    GZ_STG4(xyb[2], idx, z);
  }
};

// LF band (butteraugli.cc:510, :606-621): v = blur(xyb, sigma_lf).  Keeps the raw LF of X and Y
// (needed by the MF band and by the bright-area suppression) and writes the "vals" conversion of
// all three -- in two parts, for the X / Y planes and for the B plane (XybLowFreqToVals mixes the
// blurred Y into B, butteraugli.cc:386-389: the B part reads the raw LF of Y the first part
// wrote): the B plane is needed by k_combine only, so its blur runs on a side stream beside the
// MF / HF bands instead of in front of them.
struct PostLFxy {
  float* lf_raw[2];
  float* lf_vals[2];
  GZ_DEVFN float operator()(size_t idx, const float* v) const {
    lf_raw[0][idx] = v[0];
    lf_raw[1][idx] = v[1];
    float vx, vy, vb;
    lf_to_vals(v[0], v[1], 0.0f, &vx, &vy, &vb);
    lf_vals[0][idx] = vx;
    lf_vals[1][idx] = vy;
    return vy;
  }
  GZ_DEVFN void pair(size_t idx, const float* v0, const float* v1) const {
    (void)(*this)(idx, v0);
    (void)(*this)(idx + 1, v1);
  }
  GZ_DEVFN void quad(size_t idx, const gz_f4* v) const {
    GZ_STG4(lf_raw[0], idx, v[0]);
    GZ_STG4(lf_raw[1], idx, v[1]);
    gz_f4 vx, vy;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float vb;
      lf_to_vals(v[0].v[i], v[1].v[i], 0.0f, &vx.v[i], &vy.v[i], &vb);
    }
    GZ_STG4(lf_vals[0], idx, vx);
    GZ_STG4(lf_vals[1], idx, vy);
  }
};
struct PostLFb {
  const float* lf_raw_y;
  float* lf_vals_b;
  GZ_DEVFN float operator()(size_t idx, const float* v) const {
    float vx, vy, vb;
    lf_to_vals(0.0f, lf_raw_y[idx], v[0], &vx, &vy, &vb);
    lf_vals_b[idx] = vb;
    return vb;
  }
  GZ_DEVFN void pair(size_t idx, const float* v0, const float* v1) const {
    (void)(*this)(idx, v0);
    (void)(*this)(idx + 1, v1);
  }
  GZ_DEVFN void quad(size_t idx, const gz_f4* v) const {
    const gz_f4 y = GZ_LDG4(lf_raw_y, idx);
    gz_f4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float vx, vy;
      lf_to_vals(0.0f, y.v[i], v[0].v[i], &vx, &vy, &o.v[i]);
    }
    GZ_STG4(lf_vals_b, idx, o);
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
  GZ_DEVFN void pair(size_t idx, const float* v0, const float* v1) const {
    (void)(*this)(idx, v0);
    (void)(*this)(idx + 1, v1);
  }
  GZ_DEVFN void quad(size_t idx, const gz_f4* v) const {
    const gz_f4 x0 = GZ_LDG4(xyb[0], idx), x1 = GZ_LDG4(xyb[1], idx);
    const gz_f4 l0 = GZ_LDG4(lf_raw[0], idx), l1 = GZ_LDG4(lf_raw[1], idx);
    gz_f4 m0, m1, p0, p1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float band0 = x0.v[i] - l0.v[i];
      const float band1 = x1.v[i] - l1.v[i];
      const float h0 = band0 - v[0].v[i];
      const float h1 = band1 - v[1].v[i];
      m0.v[i] = remove_range((float)0.120079806822, v[0].v[i]);
      m1.v[i] = amplify_range((float)0.03430529365, v[1].v[i]);
      p0.v[i] = suppress_x_by_y(h0, h1);
      p1.v[i] = h1;
    }
    GZ_STG4(mf[0], idx, m0);
    GZ_STG4(mf[1], idx, m1);
    GZ_STG4(hf_pre[0], idx, p0);
    GZ_STG4(hf_pre[1], idx, p1);
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
  GZ_DEVFN void pair(size_t idx, const float* v0, const float* v1) const {
    (void)(*this)(idx, v0);
    (void)(*this)(idx + 1, v1);
  }
  GZ_DEVFN void quad(size_t idx, const gz_f4* v) const {
    const float kMulSuppressHf = (float)1.10684769012;
    const float kMulRegHf = (float)0.478741530298;
    const float kRegHf = 2000 * kMulRegHf;
    const float kMulSuppressUhf = (float)1.76905001176;
    const float kMulRegUhf = (float)0.310148420674;
    const float kRegUhf = 2000 * kMulRegUhf;
    const gz_f4 p0 = GZ_LDG4(hf_pre[0], idx), p1 = GZ_LDG4(hf_pre[1], idx), br4 = GZ_LDG4(lf_raw_y, idx);
    gz_f4 u0, h0, u1, h1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u0.v[i] = p0.v[i] - v[0].v[i];
      h0.v[i] = remove_range((float)0.0287615200377, v[0].v[i]);
      const float br = br4.v[i];
      float u = p1.v[i] - v[1].v[i];
      float hv = maximum_clamp(v[1].v[i], (float)78.8223237675);
      u = maximum_clamp(u, (float)5.8907152736);
      u = suppress_bright(u, br, kMulSuppressUhf, kRegUhf);
      hv = suppress_bright(hv, br, kMulSuppressHf, kRegHf);
      u1.v[i] = u;
      h1.v[i] = hv;
    }
    GZ_STG4(uhf[0], idx, u0);
    GZ_STG4(hf[0], idx, h0);
    GZ_STG4(uhf[1], idx, u1);
    GZ_STG4(hf[1], idx, h1);
  }
};

// Second half of CalculateDiffmap (butteraugli.cc:736-749): v = blur(d, 1.725, br 1.0).
// `out` may be null: the search loop consumes only the per-block maxima and the image maximum
// (BlockMaxOut), so the distance map itself is stored only for a caller that asked for it
// (gz_compare with distmap != NULL, the stage probes) -- 4 B/px less per Compare.
struct PostDiffmapMix {
  const float* d;
  float* out;
  GZ_DEVFN float operator()(size_t idx, const float* v) const {
    const double mul1 = 0.458794906198;
    const float scale = (float)(1.0f / (1.0f + mul1));
    float r = d[idx];
    r = (float)((double)r + mul1 * (double)v[0]);
    r = r * scale;
    if (out) out[idx] = r;
    return r;
  }
  GZ_DEVFN void pair(size_t idx, const float* v0, const float* v1) const {
    (void)(*this)(idx, v0);
    (void)(*this)(idx + 1, v1);
  }
  GZ_DEVFN void quad(size_t idx, const gz_f4* v) const {   // (unused: this functor runs with block maxima)
    for (int i = 0; i < 4; ++i) { float one[1] = {v[0].v[i]}; (void)(*this)(idx + i, one); }
  }
};

// Mask Y blur pair (butteraugli.cc:1780-1790): v[0] = blur(diffY, r0=2.377) arrives from
// an earlier pass in `b1`; this one (sigma r1) combines them.  Kept as plain stores: the
// combination is done in k_combine, which needs both anyway.

}  // namespace gz
