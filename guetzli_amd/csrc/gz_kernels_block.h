// Block kernels: one 64-lane wavefront per 8x8 block, lane = pixel / coefficient index,
// coefficients staged in LDS between the column and the row pass.
//
//   k_encode_rgb   RGB -> YCbCr -> integer FDCT -> /16        (jpeg_data_encoder.cc:40-117,
//                                                              fdct.cc:29-240)
//   k_quantize     Quantize() over all coefficients           (quantize.h:24-29)
//   k_reconstruct  integer IDCT -> YCbCr->RGB -> sRGB LUT     (idct.cc:29-161,
//                                                              color_transform.h:211-219,
//                                                              output_image.cc:411-440)
// All integer: results must be bit-identical to the reference.
#pragma once
#include "gz_common.h"

namespace gz {

#ifdef GZ_EMU
#define GZ_CONST static const
#else
#define GZ_CONST __constant__ const
#endif

// idct.cc:29-38: kIDCTMatrix[8*x+u]; rows 4..7 are the (-1)^u mirror of rows 3..0, which
// is what Compute1dIDCT's butterflies (:41-137) evaluate.
GZ_CONST int kIdctM[64] = {
  8192,  11363,  10703,   9633,   8192,   6437,   4433,   2260,
  8192,   9633,   4433,  -2259,  -8192, -11362, -10704,  -6436,
  8192,   6437,  -4433, -11362,  -8192,   2261,  10704,   9633,
  8192,   2260, -10703,  -6436,   8192,   9633,  -4433, -11363,
  8192,  -2260, -10703,   6436,   8192,  -9633,  -4433,  11363,
  8192,  -6437,  -4433,  11362,  -8192,  -2261,  10704,  -9633,
  8192,  -9633,   4433,   2259,  -8192,  11362, -10704,   6436,
  8192, -11363,  10703,  -9633,   8192,  -6437,   4433,  -2260,
};

// The same matrix with two 16-bit entries per word, (kIdctM[8*r+2p] | kIdctM[8*r+2p+1] << 16):
// operand of v_dot2c_i32_i16, which multiplies two pairs of 16-bit integers and adds both
// products to a 32-bit accumulator in ONE instruction (the entries fit 15 bits, coefficients and
// the column pass's results are int16 by definition, idct.cc:140,147).  Integer sums modulo
// 2^32: the same value as eight separate multiply-adds in any order.
#define GZ_PK16(a, b) ((uint32_t)((a) & 0xffff) | ((uint32_t)((b) & 0xffff) << 16))
GZ_CONST uint32_t kIdctMP[32] = {
  GZ_PK16(8192,  11363), GZ_PK16( 10703,   9633), GZ_PK16( 8192,   6437), GZ_PK16(  4433,   2260),
  GZ_PK16(8192,   9633), GZ_PK16(  4433,  -2259), GZ_PK16(-8192, -11362), GZ_PK16(-10704,  -6436),
  GZ_PK16(8192,   6437), GZ_PK16( -4433, -11362), GZ_PK16(-8192,   2261), GZ_PK16( 10704,   9633),
  GZ_PK16(8192,   2260), GZ_PK16(-10703,  -6436), GZ_PK16( 8192,   9633), GZ_PK16( -4433, -11363),
  GZ_PK16(8192,  -2260), GZ_PK16(-10703,   6436), GZ_PK16( 8192,  -9633), GZ_PK16( -4433,  11363),
  GZ_PK16(8192,  -6437), GZ_PK16( -4433,  11362), GZ_PK16(-8192,  -2261), GZ_PK16( 10704,  -9633),
  GZ_PK16(8192,  -9633), GZ_PK16(  4433,   2259), GZ_PK16(-8192,  11362), GZ_PK16(-10704,   6436),
  GZ_PK16(8192, -11363), GZ_PK16( 10703,  -9633), GZ_PK16( 8192,  -6437), GZ_PK16(  4433,  -2260),
};
#ifdef GZ_EMU
static inline int gz_sdot2(uint32_t a, uint32_t b, int c) {
  const unsigned p0 = (unsigned)((int)(short)(a & 0xffff) * (int)(short)(b & 0xffff));
  const unsigned p1 = (unsigned)((int)(short)(a >> 16) * (int)(short)(b >> 16));
  return (int)((unsigned)c + p0 + p1);
}
#else
typedef short gz_s2 __attribute__((ext_vector_type(2)));
GZ_DEVFN int gz_sdot2(uint32_t a, uint32_t b, int c) {
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(gz_s2, a), __builtin_bit_cast(gz_s2, b), c, false);
}
#endif
struct alignas(16) gz_u4 {
  uint32_t v[4];
};
// 16 bytes of packed int16 from LDS / constant memory (16-byte aligned)
GZ_DEVFN gz_u4 gz_load_u4(const void* p) {
#ifdef GZ_EMU
  gz_u4 r;
  __builtin_memcpy(&r, p, sizeof(r));
  return r;
#else
  return *reinterpret_cast<const gz_u4*>(p);
#endif
}
// sum_{u<8} M[row][u] * x[u] for 8 int16 values x packed in q (two per word)
GZ_DEVFN int idct_dot8(const gz_u4& m, const gz_u4& q) {
  int acc = gz_sdot2(m.v[0], q.v[0], 0);
  acc = gz_sdot2(m.v[1], q.v[1], acc);
  acc = gz_sdot2(m.v[2], q.v[2], acc);
  return gz_sdot2(m.v[3], q.v[3], acc);
}

// fdct.cc:29-36, indexed by row class {0/4, 1/7, 2/6, 3/5}.
GZ_CONST short kFdctRowTab[4][8] = {
  {22725, 21407, 19266, 16384, 12873,  8867, 4520, 0},
  {31521, 29692, 26722, 22725, 17855, 12299, 6270, 0},
  {29692, 27969, 25172, 21407, 16819, 11585, 5906, 0},
  {26722, 25172, 22654, 19266, 15137, 10426, 5315, 0},
};

constexpr int kBlocksPerWG = 4;   // 4 waves = 256 threads per workgroup

GZ_DEVFN int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// --------------------------------------------------------------------- reconstruct --
// coeffs: [3][nb][64] int16.  Outputs (either may be null): lin = 3 float planes with
// row pitch `pitch`, plane stride `pstride`; srgb = packed u8.
//
// One workgroup = 8 blocks side by side = a strip of 64 x 8 pixels: each wavefront runs the
// integer IDCT of two of the blocks (lane = pixel, three components each), the converted
// pixels go through LDS, and the strip is written as whole rows: 16 bytes per lane, 256
// contiguous bytes per row and plane.  (One wavefront per block writing its own 8 x 8 floats
// -- eight 32-byte fragments per plane, a dword per lane -- ran this streaming kernel at 1.9
// TB/s; tools/ubench/bw.hip: 3.3 TB/s for dword accesses against 5.0 for 16-byte ones.)
constexpr int kReconBlocks = 8;   // blocks per workgroup (grid = bh * ceil(bw / 8))

// libjpeg YCbCr->RGB (color_transform.h; tables == these formulas, tools/gen_tables.py)
GZ_DEVFN void ycc_to_rgb(int yy, int pcb, int pcr, int* r, int* g, int* b) {
  const int cb = pcb - 128, cr = pcr - 128, half = 1 << 15;
  *r = clamp255(yy + ((GZ_MUL24(91881, cr) + half) >> 16));
  *g = clamp255(yy + ((GZ_MUL24(-46802, cr) + (GZ_MUL24(-22554, cb) + half)) >> 16));
  *b = clamp255(yy + ((GZ_MUL24(116130, cb) + half) >> 16));
}

GZ_DEVFN void idct3_to_rgb(const int (*s_in)[64], int (*s_col)[64], int lane, int* r, int* g, int* b) {
  const int iy = lane >> 3, ix = lane & 7;
  int px[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    // row pass (idct.cc:150-160): out = clamp((sum + (257 << 17)) >> 18)
    int acc = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += GZ_MUL24(kIdctM[8 * ix + u], s_col[c][8 * iy + u]);
    px[c] = clamp255((acc + (257 << 17)) >> 18);
  }
  (void)s_in;
  ycc_to_rgb(px[0], px[1], px[2], r, g, b);
}

// Both IDCT passes as four v_dot2c_i32_i16 per output instead of eight multiply-adds, the
// coefficients kept as int16 in LDS -- transposed by the staging store, so that the eight values
// a lane needs (a column of the block, then a row of the column pass's results) are one
// 16-byte read instead of eight dword reads.  (Round 3's 24-bit multiply-add form of this kernel,
// 72 instead of 59 us at 4K, was removed in round 4; idct3_to_rgb keeps that arithmetic for the
// 4:2:0 and block-search kernels.)
// strips_per_wg: a workgroup takes that many consecutive strips of its block row, one after the
// other, with the NEXT strip's coefficients already requested while it transforms the current one
// (six 2-byte loads per lane, held in registers): twice the bytes in flight per wavefront.  Round 3's
// one-strip workgroups waited for their loads 73 % of their cycles (SQ_WAIT_ANY) at 2.6 TB/s.
// (strips_per_wg is the caller's: 4 where that still leaves several workgroups per CU.)
__global__ __launch_bounds__(256) void k_reconstruct(
    const int16_t* __restrict__ coeffs, int w, int h, int bw, int nb, int pitch,
    size_t pstride, const float* __restrict__ srgb_lut, float* __restrict__ lin,
    uint8_t* __restrict__ srgb, unsigned* __restrict__ clear_word, int strips_per_wg) {
  const int kReconStrips = strips_per_wg;
  // first kernel of a Compare: also resets the distance accumulator of its last kernel
  if (clear_word && blockIdx.x == 0 && threadIdx.x == 0) *clear_word = 0u;
  __shared__ __attribute__((aligned(16))) short s_in16[kReconBlocks][3][64];    // [ix][u]
  __shared__ __attribute__((aligned(16))) short s_col16[kReconBlocks][3][64];   // [iy][u]
  __shared__ __attribute__((aligned(16))) float s_px[3][8][kReconBlocks * 8];   // [plane][row][x]
  __shared__ uint8_t s_u8[8][kReconBlocks * 8][3];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int strips = (bw + kReconBlocks - 1) / kReconBlocks;
  const int groups = (strips + kReconStrips - 1) / kReconStrips;
  const int by = blockIdx.x / groups, s0 = (blockIdx.x % groups) * kReconStrips;
  const int ns = strips - s0 < kReconStrips ? strips - s0 : kReconStrips;
  const int iy = lane >> 3, ix = lane & 7;
  // rows iy / ix of the matrix, packed
  const gz_u4 m_row = gz_load_u4(&kIdctMP[4 * iy]), m_col = gz_load_u4(&kIdctMP[4 * ix]);
  // this lane's coefficient of the wavefront's two blocks x three components of strip `st`
  auto fetch = [&](int st, int16_t (&v)[2][3]) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int bx = (s0 + st) * kReconBlocks + 2 * wave + k;
      const bool live = bx < bw;
      const size_t blk = (size_t)by * bw + bx;
#pragma unroll
      for (int c = 0; c < 3; ++c) v[k][c] = live ? coeffs[((size_t)c * nb + blk) * 64 + lane] : (int16_t)0;
    }
  };
  int16_t cur[2][3], nxt[2][3];
  fetch(0, nxt);
#pragma unroll 1
  for (int st = 0; st < ns; ++st) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int c = 0; c < 3; ++c) cur[k][c] = nxt[k][c];
    if (st + 1 < ns) fetch(st + 1, nxt);   // in flight during this strip's arithmetic
    const int bx0 = (s0 + st) * kReconBlocks;
    // the three components of both blocks together: one memory latency, two barriers
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int j = 2 * wave + k;
#pragma unroll
      for (int c = 0; c < 3; ++c) s_in16[j][c][8 * ix + iy] = cur[k][c];   // lane = 8 * (row u) + column: transposed
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int j = 2 * wave + k;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        // column pass (idct.cc:143-149): colidcts[8*y+x] = int16((sum + 2^10) >> 11)
        const gz_u4 q = gz_load_u4(&s_in16[j][c][8 * ix]);
        s_col16[j][c][lane] = (short)((idct_dot8(m_row, q) + (1 << 10)) >> 11);
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int j = 2 * wave + k;
      int r, g, b;
      int px[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        // row pass (idct.cc:150-160): out = clamp((sum + (257 << 17)) >> 18)
        const gz_u4 q = gz_load_u4(&s_col16[j][c][8 * iy]);
        px[c] = clamp255((idct_dot8(m_col, q) + (257 << 17)) >> 18);
      }
      ycc_to_rgb(px[0], px[1], px[2], &r, &g, &b);
      const int x = 8 * j + ix;
      if (lin) {
        s_px[0][iy][x] = srgb_lut[r];
        s_px[1][iy][x] = srgb_lut[g];
        s_px[2][iy][x] = srgb_lut[b];
      }
      if (srgb) {
        s_u8[iy][x][0] = (uint8_t)r;
        s_u8[iy][x][1] = (uint8_t)g;
        s_u8[iy][x][2] = (uint8_t)b;
      }
    }
    __syncthreads();
    const int x0 = 8 * bx0, y0 = 8 * by;
    if (lin) {
      // 3 planes x 8 rows x 16 four-pixel groups = 384 16-byte stores
      for (int i = threadIdx.x; i < 3 * 8 * (kReconBlocks * 2); i += 256) {
        const int q = i % (kReconBlocks * 2), row = (i / (kReconBlocks * 2)) % 8, pl = i / (kReconBlocks * 16);
        const int x = x0 + 4 * q, y = y0 + row;
        if (y >= h || x >= w) continue;
        const size_t o = (size_t)pl * pstride + (size_t)y * pitch + x;
        if (x + 3 < w && (pitch & 3) == 0) {
          GZ_STG4(lin, o, *reinterpret_cast<const gz_f4*>(&s_px[pl][row][4 * q]));
        } else {
          for (int e = 0; e < 4 && x + e < w; ++e) lin[o + e] = s_px[pl][row][4 * q + e];
        }
      }
    }
    if (srgb) {
      for (int i = threadIdx.x; i < 8 * kReconBlocks * 8; i += 256) {
        const int xx = i % (kReconBlocks * 8), row = i / (kReconBlocks * 8);
        const int x = x0 + xx, y = y0 + row;
        if (x < w && y < h) {
          uint8_t* p = srgb + ((size_t)y * w + x) * 3;
          p[0] = s_u8[row][xx][0];
          p[1] = s_u8[row][xx][1];
          p[2] = s_u8[row][xx][2];
        }
      }
    }
    // (the next strip's first barrier stands between these reads of s_px / s_u8 and its writes
    // to them, which come two barriers later)
  }
}

// ------------------------------------------------------------ reconstruct: patches --
// Phase B's candidate of iteration i + 1 is iteration i's with a few coefficients of a fifth of the
// blocks changed (processor.cc:704-736), and a block position's pixels depend on its own three
// coefficient blocks only (4:4:4): the linear planes of the previous candidate stay where they are
// and only the changed positions are transformed again -- by the kernels that change them
// (k_reconstruct_listed behind k_apply_steps_hist / k_apply_coeff_edits).  The arithmetic is
// k_reconstruct's, statement for statement; what the reference does per changed block in
// OutputImageComponent::SetCoeffBlock -> UpdatePixelsForBlock (output_image.cc:123-132,146-160).
struct PatchPlanes {
  int bw, w, h, pitch;
  size_t pstride;
  const float* srgb_lut;
  float* lin;   // null: no patching
};

// One wavefront, lane = pixel of block position b.  blk3 = the position's three coefficient blocks in
// LDS ([3][64], natural order); tr, col = [3][64] shorts of the wavefront's own LDS (16-byte aligned).
// Every lane of the workgroup passes the same two synchronisation points (the emulation's rule);
// `live` = false transforms and stores nothing.
GZ_DEVFN void wave_reconstruct_block(const short* blk3, short* tr, short* col, int lane, int b, bool live,
                                     const PatchPlanes& pp) {
  const int iy = lane >> 3, ix = lane & 7;
#pragma unroll
  for (int c = 0; c < 3; ++c) tr[c * 64 + 8 * ix + iy] = blk3[c * 64 + lane];   // transposed
  GZ_WAVE_SYNC();
  const gz_u4 m_row = gz_load_u4(&kIdctMP[4 * iy]), m_col = gz_load_u4(&kIdctMP[4 * ix]);
#pragma unroll
  for (int c = 0; c < 3; ++c) {   // column pass (idct.cc:143-149)
    const gz_u4 q = gz_load_u4(&tr[c * 64 + 8 * ix]);
    col[c * 64 + lane] = (short)((idct_dot8(m_row, q) + (1 << 10)) >> 11);
  }
  GZ_WAVE_SYNC();
  int px[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {   // row pass (idct.cc:150-160)
    const gz_u4 q = gz_load_u4(&col[c * 64 + 8 * iy]);
    px[c] = clamp255((idct_dot8(m_col, q) + (257 << 17)) >> 18);
  }
  int r, g, bl;
  ycc_to_rgb(px[0], px[1], px[2], &r, &g, &bl);
  const int x = 8 * (b % pp.bw) + ix, y = 8 * (b / pp.bw) + iy;
  if (live && x < pp.w && y < pp.h) {
    const size_t o = (size_t)y * pp.pitch + x;
    pp.lin[o] = pp.srgb_lut[r];
    pp.lin[pp.pstride + o] = pp.srgb_lut[g];
    pp.lin[2 * pp.pstride + o] = pp.srgb_lut[bl];
  }
}

// The block positions of a list transformed again, one wavefront per entry.  POS: the entries are coefficient
// positions (gz_apply_coeff_edits: pos[i] indexes [3][nb][64]), else block positions (gz_apply_candidate_steps'
// blocks); an entry whose block position is its predecessor's is skipped, other repetitions store the same values twice.
template <bool POS>
__global__ __launch_bounds__(256) void k_reconstruct_listed(const int* __restrict__ pos, int n,
                                                            const int16_t* __restrict__ coeffs, int nb,
                                                            PatchPlanes pp) {
  __shared__ __attribute__((aligned(16))) short s_blk[kBlocksPerWG][3 * 64];
  __shared__ __attribute__((aligned(16))) short s_tr[kBlocksPerWG][3 * 64];
  __shared__ __attribute__((aligned(16))) short s_col[kBlocksPerWG][3 * 64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * kBlocksPerWG + wave;
  const int b = i < n ? (POS ? (pos[i] >> 6) % nb : pos[i]) : 0;
  const bool live = i < n && !(i > 0 && (POS ? (pos[i - 1] >> 6) % nb : pos[i - 1]) == b);
#pragma unroll
  for (int c = 0; c < 3; ++c) s_blk[wave][c * 64 + lane] = coeffs[((size_t)c * nb + b) * 64 + lane];
  GZ_WAVE_SYNC();
  wave_reconstruct_block(s_blk[wave], s_tr[wave], s_col[wave], lane, b, live, pp);
}

// gz_config.patch_reconstruct == 2: words in which the patched planes differ from a full reconstruction.
__global__ __launch_bounds__(256) void k_count_differing_words(const unsigned* __restrict__ a,
                                                               const unsigned* __restrict__ b, size_t n,
                                                               unsigned* __restrict__ count) {
  unsigned bad = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) bad += a[i] != b[i];
  if (bad) atomicAdd(count, bad);
}

// Bare-block IDCT probe (gz_probe_idct_blocks): the arithmetic of k_reconstruct.
__global__ __launch_bounds__(256) void k_idct_blocks(const int16_t* __restrict__ blocks,
                                                     int n, uint8_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) short s_in16[kBlocksPerWG][64];
  __shared__ __attribute__((aligned(16))) short s_col16[kBlocksPerWG][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int blk = blockIdx.x * kBlocksPerWG + wave;
  const bool live = blk < n;
  const int iy = lane >> 3, ix = lane & 7;
  s_in16[wave][8 * ix + iy] = live ? blocks[(size_t)blk * 64 + lane] : (int16_t)0;
  __syncthreads();
  const gz_u4 m_row = gz_load_u4(&kIdctMP[4 * iy]);
  const gz_u4 m_col = gz_load_u4(&kIdctMP[4 * ix]);
  const gz_u4 q = gz_load_u4(&s_in16[wave][8 * ix]);
  s_col16[wave][lane] = (short)((idct_dot8(m_row, q) + (1 << 10)) >> 11);
  __syncthreads();
  const gz_u4 q2 = gz_load_u4(&s_col16[wave][8 * iy]);
  if (live) out[(size_t)blk * 64 + lane] = (uint8_t)clamp255((idct_dot8(m_col, q2) + (257 << 17)) >> 18);
}

// ---------------------------------------------------------------- 4:2:0 reconstruct --
// A 2x2-subsampled component's pixel cache (OutputImageComponent::UpdatePixelsForBlock,
// output_image.cc:146-203) is a pure function of its coefficients: every update recovers the
// neighbouring subsampled samples from the upsampled pixels by inverting the "fancy
// upsampler", and because every sample is idct << 4 -- a multiple of 16 -- the 9-3-3-1
// sums are multiples of 16, the >> 4 is exact and so is the recovery.  Hence
//   pixels_[y][x] = (9 S(kx,ky) + 3 S(kx,ky+dy) + 3 S(kx+dx,ky) + S(kx+dx,ky+dy)) >> 4,
//   kx = x/2, dx = x odd ? +1 : -1 (likewise y), S = idct << 4 of the block the sample lies
//   in, sample coordinates clamped to [0, (w-1)/2] x [0, (h-1)/2] (the replication rules of
//   :162-169),
// whatever sequence of SetCoeffBlock calls produced the image (checked against the reference
// with random update orders, tests/test_oracle_vs_ref.py).  ToPixels (:82) then rounds:
// (p + 8 - (x & 1)) >> 4.
//
// k_chroma_samples: integer IDCT of every block of the two subsampled components into two
// u8 sample planes [cbh*8][cbw*8] (Cb, then Cr, plane stride cbw*8*cbh*8).
__global__ __launch_bounds__(256) void k_chroma_samples(const int16_t* __restrict__ cb_blocks,
                                                        const int16_t* __restrict__ cr_blocks,
                                                        int cbw, int nbc,
                                                        uint8_t* __restrict__ samples) {
  __shared__ int s_in[2][kBlocksPerWG][64];
  __shared__ int s_col[2][kBlocksPerWG][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int blk = blockIdx.x * kBlocksPerWG + wave;
  const bool live = blk < nbc;
  const int iy = lane >> 3, ix = lane & 7;
  s_in[0][wave][lane] = live ? (int)cb_blocks[(size_t)blk * 64 + lane] : 0;
  s_in[1][wave][lane] = live ? (int)cr_blocks[(size_t)blk * 64 + lane] : 0;
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    int acc = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += GZ_MUL24(kIdctM[8 * iy + u], s_in[c][wave][8 * u + ix]);
    s_col[c][wave][lane] = (int)(short)((acc + (1 << 10)) >> 11);
  }
  __syncthreads();
  if (!live) return;
  const int sw = cbw * 8;
  const size_t pl = (size_t)sw * (size_t)(nbc / cbw) * 8;
  const size_t o = (size_t)(8 * (blk / cbw) + iy) * sw + 8 * (blk % cbw) + ix;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    int acc = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += GZ_MUL24(kIdctM[8 * ix + u], s_col[c][wave][8 * iy + u]);
    samples[c * pl + o] = (uint8_t)clamp255((acc + (257 << 17)) >> 18);
  }
}

// Upsampled, rounded pixel (ToPixels) of a 2x2-subsampled component at (x, y) from its sample
// plane (row pitch sw); mxs, mys = (w-1)/2, (h-1)/2.
GZ_DEVFN int chroma420_pixel(const uint8_t* __restrict__ sp, int sw, int mxs, int mys, int x, int y) {
  const int kx = x >> 1, ky = y >> 1;
  int kx2 = (x & 1) ? kx + 1 : kx - 1, ky2 = (y & 1) ? ky + 1 : ky - 1;
  kx2 = kx2 < 0 ? 0 : (kx2 > mxs ? mxs : kx2);
  ky2 = ky2 < 0 ? 0 : (ky2 > mys ? mys : ky2);
  const int a = (int)sp[(size_t)ky * sw + kx] << 4, b = (int)sp[(size_t)ky2 * sw + kx] << 4;
  const int c = (int)sp[(size_t)ky * sw + kx2] << 4, d = (int)sp[(size_t)ky2 * sw + kx2] << 4;
  const int p = (a * 9 + b * 3 + c * 3 + d) >> 4;
  return (p + 8 - (x & 1)) >> 4;
}

// k_reconstruct for a 4:2:0 frame: luma IDCT of a strip of 8 blocks as in k_reconstruct (which
// follows below), chroma from the sample planes of k_chroma_samples, whole rows out.
__global__ __launch_bounds__(256) void k_reconstruct420(
    const int16_t* __restrict__ ycoeffs, const uint8_t* __restrict__ samples, int w, int h, int bw,
    int nb, int cbw, int cbh, int pitch, size_t pstride, const float* __restrict__ srgb_lut,
    float* __restrict__ lin, uint8_t* __restrict__ srgb, unsigned* __restrict__ clear_word) {
  if (clear_word && blockIdx.x == 0 && threadIdx.x == 0) *clear_word = 0u;
  constexpr int NB = 8;   // blocks per workgroup (== kReconBlocks)
  __shared__ int s_in[NB][64];
  __shared__ int s_col[NB][64];
  __shared__ __attribute__((aligned(16))) float s_px[3][8][NB * 8];
  __shared__ uint8_t s_u8[8][NB * 8][3];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int strips = (bw + NB - 1) / NB;
  const int by = blockIdx.x / strips, bx0 = (blockIdx.x % strips) * NB;
  const int iy = lane >> 3, ix = lane & 7;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int j = 2 * wave + k, bx = bx0 + j;
    s_in[j][lane] = bx < bw ? (int)ycoeffs[((size_t)by * bw + bx) * 64 + lane] : 0;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int j = 2 * wave + k;
    int acc = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += GZ_MUL24(kIdctM[8 * iy + u], s_in[j][8 * u + ix]);
    s_col[j][lane] = (int)(short)((acc + (1 << 10)) >> 11);
  }
  __syncthreads();
  const int sw = cbw * 8;
  const size_t pl_bytes = (size_t)sw * cbh * 8;
  const int mxs = (w - 1) >> 1, mys = (h - 1) >> 1;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int j = 2 * wave + k;
    int acc = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += GZ_MUL24(kIdctM[8 * ix + u], s_col[j][8 * iy + u]);
    const int yy = clamp255((acc + (257 << 17)) >> 18);
    const int x = 8 * (bx0 + j) + ix, y = 8 * by + iy;
    int r = 0, g = 0, b = 0;
    if (x < w && y < h) {
      const int cb = chroma420_pixel(samples, sw, mxs, mys, x, y) - 128;
      const int cr = chroma420_pixel(samples + pl_bytes, sw, mxs, mys, x, y) - 128;
      const int half = 1 << 15;
      r = clamp255(yy + ((GZ_MUL24(91881, cr) + half) >> 16));
      g = clamp255(yy + ((GZ_MUL24(-46802, cr) + (GZ_MUL24(-22554, cb) + half)) >> 16));
      b = clamp255(yy + ((GZ_MUL24(116130, cb) + half) >> 16));
    }
    const int xs = 8 * j + ix;
    if (lin) {
      s_px[0][iy][xs] = srgb_lut[r];
      s_px[1][iy][xs] = srgb_lut[g];
      s_px[2][iy][xs] = srgb_lut[b];
    }
    if (srgb) {
      s_u8[iy][xs][0] = (uint8_t)r;
      s_u8[iy][xs][1] = (uint8_t)g;
      s_u8[iy][xs][2] = (uint8_t)b;
    }
  }
  __syncthreads();
  const int x0 = 8 * bx0, y0 = 8 * by;
  if (lin) {
    for (int i = threadIdx.x; i < 3 * 8 * (NB * 2); i += 256) {
      const int q = i % (NB * 2), row = (i / (NB * 2)) % 8, pl = i / (NB * 16);
      const int x = x0 + 4 * q, y = y0 + row;
      if (y >= h || x >= w) continue;
      const size_t o = (size_t)pl * pstride + (size_t)y * pitch + x;
      if (x + 3 < w && (pitch & 3) == 0) {
        GZ_STG4(lin, o, *reinterpret_cast<const gz_f4*>(&s_px[pl][row][4 * q]));
      } else {
        for (int e = 0; e < 4 && x + e < w; ++e) lin[o + e] = s_px[pl][row][4 * q + e];
      }
    }
  }
  if (srgb) {
    for (int i = threadIdx.x; i < 8 * NB * 8; i += 256) {
      const int xx = i % (NB * 8), row = i / (NB * 8);
      const int x = x0 + xx, y = y0 + row;
      if (x < w && y < h) {
        uint8_t* p = srgb + ((size_t)y * w + x) * 3;
        p[0] = s_u8[row][xx][0];
        p[1] = s_u8[row][xx][1];
        p[2] = s_u8[row][xx][2];
      }
    }
  }
}

// ------------------------------------------------------------------------ quantize --
// quantize.h:24-29 applied to every coefficient of the original; q = int[3][64].  b1, b2 =
// first block of components 1 and 2 in the arrays, nblk = blocks of all three components.
__global__ __launch_bounds__(256) void k_quantize(const int16_t* __restrict__ orig,
                                                  int16_t* __restrict__ cand, int b1, int b2,
                                                  int nblk, const int* __restrict__ q) {
  const size_t total = (size_t)nblk * 64;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int blk = (int)(i >> 6);
    const int c = blk >= b2 ? 2 : (blk >= b1 ? 1 : 0);
    const int quant = q[c * 64 + (int)(i & 63)];
    const int raw = orig[i];
    const int r = raw % quant;
    const int delta = 2 * r > quant ? quant - r : ((-2) * r > quant ? -quant - r : -r);
    cand[i] = (int16_t)(raw + (int)(short)delta);
  }
}

// -------------------------------------------------------------------------- encode --
GZ_DEVFN int mulhi16(int a, int b) { return (a * b) >> 16; }   // MULT, fdct.cc:150

// COLUMN_DCT8 (fdct.cc:68-145) on one column of an LDS-resident block (stride 8).
GZ_DEVFN void fdct_column(short* col) {
  const int i0 = col[0], i1 = col[8], i2 = col[16], i3 = col[24];
  const int i4 = col[32], i5 = col[40], i6 = col[48], i7 = col[56];
  int d07 = i0 - i7, s07 = i0 + i7;
  int d25 = i2 - i5, s25 = i2 + i5;
  int d34 = i3 - i4, s34 = i3 + i4;
  int d16 = i1 - i6, s16 = i1 + i6;
  int e0 = s07 - s34, e1 = s07 + s34;
  int e2 = s16 - s25, e3 = s16 + s25;
  e1 *= 8;
  e3 *= 8;
  col[0] = (short)(e1 + e3);
  col[32] = (short)(e1 - e3);
  e0 *= 8;
  e2 *= 8;
  d34 *= 8;
  d07 *= 8;
  const int kTan1 = 13036, kTan2 = 27146, kTan3m1 = -21746, k2Sqrt2 = 23170;
  col[16] = (short)(mulhi16(kTan2, e2) + e0);
  col[48] = (short)(mulhi16(kTan2, e0) - e2);
  d25 *= 16;
  d16 *= 16;
  const int p = mulhi16(d16 + d25, k2Sqrt2);
  const int q = mulhi16(d16 - d25, k2Sqrt2);
  int m3 = d34 - q, m1 = d34 + q;
  const int m0 = d07 - p, m2 = d07 + p;
  const int m7 = m3, m6 = m1;
  m3 = mulhi16(m3, kTan3m1) + m7 + 1;
  m1 = mulhi16(m1, kTan1) + m2 + 1;
  const int m4 = mulhi16(kTan3m1, m0) + m0;
  const int m5 = mulhi16(kTan1, m2);
  col[8] = (short)m1;
  col[24] = (short)(m0 - m3);
  col[40] = (short)(m7 + m4);
  col[56] = (short)(m5 - m6);
}

// RowDct (fdct.cc:173-208) on one row.
GZ_DEVFN void fdct_row(short* in, const short* t) {
  const int a0 = in[0] + in[7], b0 = in[0] - in[7];
  const int a1 = in[1] + in[6], b1 = in[1] - in[6];
  const int a2 = in[2] + in[5], b2 = in[2] - in[5];
  const int a3 = in[3] + in[4], b3 = in[3] - in[4];
  const int C1 = t[0], C2 = t[1], C3 = t[2], C4 = t[3], C5 = t[4], C6 = t[5], C7 = t[6];
  const int c0 = a0 + a3, c1 = a0 - a3, c2 = a1 + a2, c3 = a1 - a2;
  in[0] = (short)((C4 * (c0 + c2)) >> 16);
  in[4] = (short)((C4 * (c0 - c2)) >> 16);
  in[2] = (short)((C2 * c1 + C6 * c3) >> 16);
  in[6] = (short)((C6 * c1 - C2 * c3) >> 16);
  in[1] = (short)((C1 * b0 + C3 * b1 + C5 * b2 + C7 * b3) >> 16);
  in[3] = (short)((C3 * b0 - C7 * b1 - C1 * b2 - C5 * b3) >> 16);
  in[5] = (short)((C5 * b0 - C1 * b1 + C7 * b2 + C3 * b3) >> 16);
  in[7] = (short)((C7 * b0 - C5 * b1 + C3 * b2 - C1 * b3) >> 16);
}

GZ_DEVFN int fdct_row_class(int r) { return r == 0 || r == 4 ? 0 : (r == 1 || r == 7 ? 1 : (r == 2 || r == 6 ? 2 : 3)); }

// EncodeRGBToJpeg with all-ones quant (jpeg_data_encoder.cc:66-117).
__global__ __launch_bounds__(256) void k_encode_rgb(const uint8_t* __restrict__ rgb, int w,
                                                    int h, int bw, int nb,
                                                    int16_t* __restrict__ coeffs) {
  __shared__ short s_blk[kBlocksPerWG][3][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int blk = blockIdx.x * kBlocksPerWG + wave;
  const bool live = blk < nb;
  if (live) {
    const int iy = lane >> 3, ix = lane & 7;
    int y = 8 * (blk / bw) + iy, x = 8 * (blk % bw) + ix;
    y = y < h - 1 ? y : h - 1;
    x = x < w - 1 ? x : w - 1;
    const uint8_t* p = rgb + ((size_t)y * w + x) * 3;
    const int r = p[0], g = p[1], b = p[2], HALF = 1 << 15;
    s_blk[wave][0][lane] = (short)((19595 * r + 38469 * g + 7471 * b - (128 << 16) + HALF) >> 16);
    s_blk[wave][1][lane] = (short)((-11059 * r - 21709 * g + 32768 * b + HALF - 1) >> 16);
    s_blk[wave][2][lane] = (short)((32768 * r - 27439 * g - 5329 * b + HALF - 1) >> 16);
  }
  __syncthreads();
  if (live && lane < 24) fdct_column(&s_blk[wave][lane >> 3][lane & 7]);
  __syncthreads();
  if (live && lane < 24)
    fdct_row(&s_blk[wave][lane >> 3][8 * (lane & 7)], kFdctRowTab[fdct_row_class(lane & 7)]);
  __syncthreads();
  if (live) {
    for (int c = 0; c < 3; ++c) {
      const int v = s_blk[wave][c][lane];
      coeffs[((size_t)c * nb + blk) * 64 + lane] = (int16_t)((v * 65537 + 0x80000) >> 20);
    }
  }
}

// Bare-block FDCT probe.
__global__ __launch_bounds__(256) void k_fdct_blocks(int16_t* __restrict__ blocks, int n) {
  __shared__ short s_blk[kBlocksPerWG][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int blk = blockIdx.x * kBlocksPerWG + wave;
  const bool live = blk < n;
  if (live) s_blk[wave][lane] = blocks[(size_t)blk * 64 + lane];
  __syncthreads();
  if (live && lane < 8) fdct_column(&s_blk[wave][lane]);
  __syncthreads();
  if (live && lane < 8) fdct_row(&s_blk[wave][8 * lane], kFdctRowTab[fdct_row_class(lane)]);
  __syncthreads();
  if (live) blocks[(size_t)blk * 64 + lane] = s_blk[wave][lane];
}

// OutputImageComponent::SetCoeffBlock for a batch of blocks (output_image.cc:123-132):
// blocks[i] = 3 x 64 coefficients (Y, Cb, Cr) of block index[i].  One wave per block.
__global__ __launch_bounds__(256) void k_scatter_blocks(const int32_t* __restrict__ index,
                                                        const int16_t* __restrict__ blocks,
                                                        int n, int nb,
                                                        int16_t* __restrict__ coeffs) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * kBlocksPerWG + wave;
  if (i >= n) return;
  const int b = index[i];
#pragma unroll
  for (int c = 0; c < 3; ++c)
    coeffs[((size_t)c * nb + b) * 64 + lane] = blocks[((size_t)i * 3 + c) * 64 + lane];
}

// sRGB u8 packed -> 3 linear float planes (LinearRgb, butteraugli_comparator.cc:33-47).
__global__ __launch_bounds__(256) void k_linear_from_rgb8(const uint8_t* __restrict__ rgb,
                                                          int w, int h, int pitch,
                                                          size_t pstride,
                                                          const float* __restrict__ lut,
                                                          float* __restrict__ lin) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w || y >= h) return;
  const uint8_t* p = rgb + ((size_t)y * w + x) * 3;
  const size_t o = (size_t)y * pitch + x;
  lin[o] = lut[p[0]];
  lin[pstride + o] = lut[p[1]];
  lin[2 * pstride + o] = lut[p[2]];
}

}  // namespace gz
