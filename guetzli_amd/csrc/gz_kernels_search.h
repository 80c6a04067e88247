// Batched phase A of Processor::SelectFrequencyMasking (processor.cc:554-590): for every
// 8x8 block, the greedy coefficient-zeroing order of ComputeBlockZeroingOrder
// (processor.cc:364-467) with the per-candidate error of
// ButteraugliComparator::CompareBlock (butteraugli_comparator.cc:457-488).
//
// One workgroup = one 64-lane wavefront = one block position (Y, Cb, Cr blocks together);
// lane = pixel.  Everything a block needs for its <= 189 sequential steps x <= 3 look-ahead
// candidates lives in LDS / registers: the three coefficient blocks, the pixel cache of the
// current state, the original block's 8x8 opsin image, the ranked candidate list.  HBM
// traffic is the compulsory ~1.5 KB per block; the kernel is bound by the dependent
// FP64 chains (8x8 opsin polynomial, DJB FFT butterflies, the two in-order sums), not by
// bandwidth.
//
// Per candidate (CompareBlock): integer IDCT of the changed component -> edge replicate
// (ToPixels, output_image.cc:85-96) -> YCbCr->RGB -> sRGB LUT -> 8x8 OpsinDynamicsImage
// (r=2 blur with border renormalisation on columns/rows 0,1,6,7) -> per-channel
// difference in double -> ButteraugliBlockDiff (:382-411): 4*mean^2 with the mean summed
// in index order, 2-D real FFT (RealFFT8 rows, FFT8 / RealFFT8 columns, :154-380) and the
// CSF-weighted power sum in index order -> sqrt(sum_c diff_c * mask_c(corner)).
#pragma once
#include "gz_common.h"
#include "gz_kernels_block.h"
#include "gz_kernels_blur.h"
#include "gz_math.h"

namespace gz {

// GetContrastSensitivityMatrix, butteraugli_comparator.cc:93-134 (entries 4..36 are read).
GZ_CONST double kCsf8x8[37] = {
  0.0, 0.0, 0.0, 0.0,
  0.3831134973, 0.676303603859, 1.1550451483, 8,
  8, 0.692062533689, 0.847511538605, 0.498250875965, 0.36198671102, 0.308982169883,
  0.1312701920435, 4.71274312228,
  1.1550451483, 0.847511538605, 4.71274312228, 0.991205724152, 1.30229591239,
  0.627264168628, 0.4, 0.1312701920435,
  0.676303603859, 0.498250875965, 0.991205724152, 0.5, 0.3831134973, 0.349686450518,
  0.627264168628, 0.308982169883,
  0.3831134973, 0.36198671102, 1.30229591239, 0.3831134973, 0.323078800177,
};

struct Cpx {
  double re, im;
};

#define GZ_SQRT_HALF 0.70710678118654752440084436210484903

// RealFFT8 (butteraugli_comparator.cc:282-353) in single-assignment form with the final
// output order; one rounding per written operation, same association as the reference.
GZ_DEVFN void real_fft8(const double a0, const double a1, const double a2, const double a3,
                        const double a4, const double a5, const double a6, const double a7,
                        Cpx* F) {
  const double d26 = a2 - a6, s26 = a6 + a2;
  const double d04 = a0 - a4, s04 = a4 + a0;
  const double d15 = a1 - a5, s15 = a5 + a1;
  const double d37 = a3 - a7, s37 = a7 + a3;
  const double nd37 = -d37, nd26 = -d26;
  const double m6 = (d15 - d37) * GZ_SQRT_HALF;
  const double m1 = (d15 + d37) * GZ_SQRT_HALF;
  const double m5 = (nd37 - d15) * GZ_SQRT_HALF;
  const double m2 = (nd37 + d15) * GZ_SQRT_HALF;
  const double e = s26 + s04, o = s37 + s15;
  const double t3 = s15 - s37, t1 = s04 - s26;
  F[0].re = e + o;      F[0].im = 0.0;
  F[1].re = m2 + d04;   F[1].im = m5 + nd26;
  F[2].re = t1;         F[2].im = -t3;
  F[3].re = d04 - m6;   F[3].im = d26 - m1;
  F[4].re = e - o;      F[4].im = 0.0;
  F[5].re = d04 - m2;   F[5].im = nd26 - m5;
  F[6].re = t1;         F[6].im = t3;
  F[7].re = m6 + d04;   F[7].im = m1 + d26;
}

// FFT8 + FFT4 (butteraugli_comparator.cc:154-277), complex input, final output order.
GZ_DEVFN void fft8(const Cpx* a, Cpx* F) {
  const double I0 = a[4].im + a[0].im, dI04 = a[0].im - a[4].im;
  const double R2 = a[6].re + a[2].re, dR26 = a[2].re - a[6].re;
  const double a6im = dI04 - dR26, a4im = dI04 + dR26;
  const double dI26 = a[2].im - a[6].im, I2 = a[6].im + a[2].im;
  const double dR04 = a[0].re - a[4].re, R0 = a[4].re + a[0].re;
  const double a4re = dR04 - dI26, a6re = dR04 + dI26;
  const double dR15 = a[1].re - a[5].re, R1 = a[5].re + a[1].re;
  const double dI37 = a[3].im - a[7].im, I3 = a[7].im + a[3].im;
  const double u1 = dR15 - dI37, u3 = dR15 + dI37;
  const double dI15 = a[1].im - a[5].im, I1 = a[5].im + a[1].im;
  const double dR37 = a[3].re - a[7].re, R3 = a[7].re + a[3].re;
  const double u2 = dI15 - dR37, u4 = dI15 + dR37;
  const double m6 = (u1 - u4) * GZ_SQRT_HALF;
  const double m1 = (u1 + u4) * GZ_SQRT_HALF;
  const double m5 = (u2 - u3) * GZ_SQRT_HALF;
  const double m2 = (u2 + u3) * GZ_SQRT_HALF;
  const double e = R2 + R0, o = R3 + R1, t1 = R0 - R2, t3 = R1 - R3;
  const double f = I2 + I0, g = I3 + I1, t2 = I0 - I2, t4 = I1 - I3;
  F[0].re = e + o;       F[0].im = f + g;
  F[1].re = m2 + a6re;   F[1].im = m5 + a6im;
  F[2].re = t1 + t4;     F[2].im = t2 - t3;
  F[3].re = a4re - m6;   F[3].im = a4im - m1;
  F[4].re = e - o;       F[4].im = f - g;
  F[5].re = a6re - m2;   F[5].im = a6im - m5;
  F[6].re = t1 - t4;     F[6].im = t2 + t3;
  F[7].re = m6 + a4re;   F[7].im = m1 + a4im;
}

struct SearchArgs {
  const int16_t* coeffs;        // candidate image coefficients (frame layout, coff[] below)
  const int32_t* rank_cnt;      // [grid blocks]
  const uint8_t* rank_idx;      // [grid blocks][192]: ranked input_order (processor.cc:381-400), k_rank_candidates
  const uint8_t* rgb;           // original sRGB image
  const float* srgb_lut;        // 256 floats
  const float* block_mask;      // [3][nb]: mask_xyz_[c](8*by, 8*bx) on the 8x8 (luma) grid
  int w, h, bw, nb;             // luma grid
  int coff[3];                  // first block of each component in coeffs
  int cbw;                      // 4:2:0 frames: width of the chroma block grid
  const uint8_t* samples;       // 4:2:0 frames: the chroma sample planes of k_chroma_samples
  int lookahead;                // Params::zeroing_greedy_lookahead
  float limit;                  // Comparator::BlockErrorLimit()
  Taps<2> taps;                 // sigma 1.2
  float scale_lo[2], scale_hi[2];   // border scales for an axis of length 8, border_ratio 0
  int32_t* out_cnt;             // [grid blocks]
  uint8_t* out_idx;             // [grid blocks][192]
  float* out_err;               // [grid blocks][192]
};

// Candidates of one greedy step that are evaluated together: their wide stages (IDCT, colour,
// opsin -- 64 busy lanes each) run one after the other, their narrow stages (in-order means,
// FFTs, in-order CSF sums -- 3 to 24 busy lanes and long dependent FP64 chains per candidate)
// run side by side on lanes that would otherwise idle.  3 = the default look-ahead.
constexpr int kEvalBatch = 3;

struct SearchLds {
  short coef[192];
  // the same coefficients transposed, [c][8 * column + row]: the eight values of a column are one
  // 16-byte read (the IDCT's packed dot products, gz_kernels_block.h)
  alignas(16) short coefT[192];
  alignas(16) short col16[64];   // the IDCT's column pass, [8 * row + column]
  // [0..2]: pixel cache of the current (processed) block (values 0..255); [3]: the changed component
  // of the candidate being evaluated
  unsigned char ycc[4][64];
  unsigned char list[192];
  unsigned char oidx[192];
  float oerr[192];
  int cpx[64];          // 4:2:0 chroma search: the changed component's samples
  // row-blurred planes between the two passes of the 8x8 blur: 12 rows of 8, rows 0, 1, 10, 11 hold
  // +0.0f for the whole kernel (blur8_init), so the column pass reads its five taps at constant offsets
  float tmp[3][96];
  // per candidate of the batch: opsin differences; after the row stage their row transforms in
  // place, 8 doubles per row: F0.re, F4.re, F1, F2, F3 (F0, F4 of a real row are real); after the
  // column stage |.|^2 * 0.000064 for flat indices 0..39 at the front of each plane
  double d[kEvalBatch][3][64];
  double red[kEvalBatch][3];
  float err[kEvalBatch];
  int num;
#ifdef GZ_SEARCH_LDS_PAD
  char pad[GZ_SEARCH_LDS_PAD];   // (experiment: fewer wavefronts per CU)
#endif
};
// Round 4 took the struct from 12 672 to 10 064 bytes: the sRGB table is read from global memory (three
// L1-resident loads per evaluation instead of three LDS reads at random banks), the pixel cache holds
// bytes, and what only the 4:2:0 chroma search needs lives in a struct of its own.  Round 5: the
// original's opsin image and the candidate's linear RGB live in registers; the blur's plane between its
// passes carries four rows of zeros, the coefficients are kept transposed as well, and the column
// stage's power terms replace its input in place (8 416 bytes: LDS no longer decides the occupancy).
#ifndef GZ_SEARCH_LDS_PAD
static_assert(sizeof(SearchLds) <= 160 * 1024 / 16, "k_block_search: four wavefronts per SIMD need <= 10 KB each");
#endif

// 4:2:0 chroma search only (MODE 2): the 10x10 subsampled samples around the 16x16 block
// (UpdatePixelsForBlock's `subsampled`, output_image.cc:150-183) per chroma component.
template <bool ON>
struct SearchLds420 {
  int s10[2][ON ? 100 : 1];
  short cellsrc[ON ? 100 : 2];   // >= 0: the cell is sample cellsrc of the block itself; -1: a neighbour's
};

// Integer IDCT of component c with coefficient `zero_k` forced to 0 (or -1: none); the result of
// this lane's pixel is left in `dst[lane]` after the trailing barrier.  Both passes as four packed
// 16-bit dot products per output (idct_dot8; the same integers as eight multiply-adds).
template <class T>
GZ_DEVFN void idct_component(SearchLds& s, int c, int zero_k, int lane, T* dst) {
  const int iy = lane >> 3, ix = lane & 7;
  const gz_u4 m_row = gz_load_u4(&kIdctMP[4 * iy]), m_col = gz_load_u4(&kIdctMP[4 * ix]);
  const int zt = ((zero_k & 7) << 3) | ((zero_k >> 3) & 7);   // its place in the transposed block
  if (lane == 0 && zero_k >= 0) s.coefT[64 * c + zt] = 0;
  __syncthreads();
  // column pass (idct.cc:143-149): colidcts[8*y+x] = int16((sum + 2^10) >> 11)
  const gz_u4 q = gz_load_u4(&s.coefT[64 * c + 8 * ix]);
  s.col16[lane] = (short)((idct_dot8(m_row, q) + (1 << 10)) >> 11);
  __syncthreads();
  if (lane == 0 && zero_k >= 0) s.coefT[64 * c + zt] = s.coef[64 * c + zero_k];   // (every lane has read it)
  // row pass (idct.cc:150-160): out = clamp((sum + (257 << 17)) >> 18)
  const gz_u4 r = gz_load_u4(&s.col16[8 * iy]);
  dst[lane] = (T)clamp255((idct_dot8(m_col, r) + (257 << 17)) >> 18);
  __syncthreads();
}

// The 8x8 window's blur (Convolution on an 8-wide image: positions 2..5 interior -- pre-scaled taps --,
// 0, 1, 6, 7 border -- raw taps of the samples inside, times a scale) without a branch: every lane
// holds its five weights per axis (a tap that falls outside the window gets weight +0.0f: the
// samples are finite and non-negative, so its product is +0.0f and adding it changes nothing) and
// its scale (1.0f inside: exact).  The sums run in the reference's order, from 0.0f.
struct Blur8 {
  float wx[5], wy[5], sx, sy;
};
GZ_DEVFN void blur8_axis(int pos, const SearchArgs& a, float* w, float* scale) {
  const bool interior = pos >= 2 && pos < 6;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int p = pos - 2 + j;
    w[j] = interior ? a.taps.ks[j] : (p >= 0 && p < 8 ? a.taps.k[j] : 0.0f);
  }
  // (selected, not indexed: an array indexed by the lane would be fetched from memory)
  const float lo = pos == 0 ? a.scale_lo[0] : a.scale_lo[1];
  const float hi = pos == 7 ? a.scale_hi[0] : a.scale_hi[1];
  *scale = interior ? 1.0f : (pos < 2 ? lo : hi);
}
GZ_DEVFN Blur8 blur8_init(SearchLds& s, int lane, const SearchArgs& a) {
  Blur8 w;
  blur8_axis(lane & 7, a, w.wx, &w.sx);
  blur8_axis(lane >> 3, a, w.wy, &w.sy);
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    w.wx[j] = GZ_IN_VGPR(w.wx[j]);
    w.wy[j] = GZ_IN_VGPR(w.wy[j]);
  }
  if (lane < 16) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      s.tmp[c][lane] = 0.0f;        // rows 0, 1
      s.tmp[c][80 + lane] = 0.0f;   // rows 10, 11
    }
  }
  return w;
}
// Row pass: the neighbours of a pixel sit in the neighbouring lanes (8 pixels of a row = 8
// consecutive lanes), so the taps come through the lane-shift operand of the multiply, not LDS.
GZ_DEVFN float blur8_row(float v, const Blur8& w) {
  float sum = 0.0f;
  sum += gz_row16_shift<-2>(v) * w.wx[0];
  sum += gz_row16_shift<-1>(v) * w.wx[1];
  sum += v * w.wx[2];
  sum += gz_row16_shift<1>(v) * w.wx[3];
  sum += gz_row16_shift<2>(v) * w.wx[4];
  return sum * w.sx;
}
GZ_DEVFN float blur8_col(const float* t, int lane, const Blur8& w) {
  float sum = 0.0f;
#pragma unroll
  for (int j = 0; j < 5; ++j) sum += t[lane + 8 * j] * w.wy[j];
  return sum * w.sy;
}

// 8x8 OpsinDynamicsImage of this lane's linear (r, g, b) and its window -> this lane's (x, y, b).
GZ_DEVFN void opsin8x8(SearchLds& s, int lane, const Blur8& w, float r, float g, float bl,
                       float* ox, float* oy, float* ob) {
  s.tmp[0][16 + lane] = blur8_row(r, w);
  s.tmp[1][16 + lane] = blur8_row(g, w);
  s.tmp[2][16 + lane] = blur8_row(bl, w);
  __syncthreads();
  float b[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) b[c] = blur8_col(s.tmp[c], lane, w);
  opsin_pixel(b[0], b[1], b[2], r, g, bl, ox, oy, ob);
  __syncthreads();
}

// 4:2:0 chroma search: the 10x10 sample array of chroma component cc (1, 2) with the block's
// own samples taken from `own` (an 8x8 IDCT result), and this lane's upsampled + rounded pixel
// (UpdatePixelsForBlock's fancy upsampler :192-203, then ToPixels :82) of the 8x8 sub-block
// (off_x, off_y) of the 16x16 area.
GZ_DEVFN void fill_s10(SearchLds420<true>& q, int cc, const int* own, const int* ring, int lane) {
  for (int cell = lane; cell < 100; cell += 64) {
    const int src = q.cellsrc[cell];
    q.s10[cc - 1][cell] = src >= 0 ? own[src] : ring[cell];
  }
  __syncthreads();
}
GZ_DEVFN int upsampled_pixel(const int* s10, int lx, int ly) {
  const int col = (lx >> 1) + 1, row = (ly >> 1) + 1;
  const int dx = (lx & 1) ? 1 : -1, dy = (ly & 1) ? 10 : -10;
  const int i = row * 10 + col;
  // the reference keeps samples as idct << 4, which makes the >> 4 of its 9-3-3-1 sum exact:
  // pixels_ = 9a + 3b + 3c + d on the plain idct values; ToPixels rounds that
  const int p = s10[i] * 9 + s10[i + dy] * 3 + s10[i + dx] * 3 + s10[i + dx + dy];
  return (p + 8 - (lx & 1)) >> 4;
}

// What a lane keeps for the whole search of its block: the blur's weights and the original block's
// opsin image at its pixel (per_block_pregamma_, widened once).
struct SearchLane {
  Blur8 w;
  double x0[3];
};
// SwitchBlock (butteraugli_comparator.cc:427-455): the original block, clamped gather.
GZ_DEVFN SearchLane search_lane_init(SearchLds& s, int lane, int xmin, int ymin, const SearchArgs& a) {
  SearchLane L;
  L.w = blur8_init(s, lane, a);
  const int iy = lane >> 3, ix = lane & 7;
  const int x = xmin + ix < a.w - 1 ? xmin + ix : a.w - 1;
  const int y = ymin + iy < a.h - 1 ? ymin + iy : a.h - 1;
  const uint8_t* p = a.rgb + ((size_t)y * a.w + x) * 3;
  __syncthreads();   // (the blur's zero rows)
  float x0, y0, z0;
  opsin8x8(s, lane, L.w, GZ_LDG(a.srgb_lut, p[0]), GZ_LDG(a.srgb_lut, p[1]), GZ_LDG(a.srgb_lut, p[2]),
           &x0, &y0, &z0);
  L.x0[0] = (double)x0;
  L.x0[1] = (double)y0;
  L.x0[2] = (double)z0;
  return L;
}

// What differs between the three searches.
//   MODE 0: 4:4:4 frame (any component mask: the candidate list decides) -- 8x8 grid
//   MODE 1: 4:2:0 frame, luma candidates -- 8x8 grid, chroma pixels fixed
//   MODE 2: 4:2:0 frame, chroma candidates -- 16x16 grid, four wavefronts per block, one per
//           8x8 sub-block; the block's error is the maximum over its in-image sub-blocks
//           (processor.cc:420-430)
struct SearchView {
  int vw, vh;            // in-image width / height of this wavefront's 8x8 window
  int off_x, off_y;      // MODE 2: sub-block
  float m0, m1, m2;      // mask at the window's corner
  bool in_image;         // MODE 2: the sub-block is compared at all
};

// Wide stages of CompareBlock for this wavefront's 8x8 window with coefficient `ci`
// (= c*64+k) zeroed: the opsin differences of the candidate go to s.d[slot].
template <int MODE>
GZ_DEVFN void eval_wide(SearchLds& s, SearchLds420<MODE == 2>& q, int ci, int slot, int lane,
                        const SearchView& v, const SearchLane& L, const int* ring, const SearchArgs& a) {
  // ci == 64 * 3: nothing zeroed (the luma component is simply recomputed)
  const int cc = ci >= 192 ? 0 : ci >> 6, kk = ci >= 192 ? -1 : ci & 63;
  // edge replication + colour + LUT
  const int iy = lane >> 3, ix = lane & 7;
  const int sx = ix < v.vw ? ix : v.vw - 1, sy = iy < v.vh ? iy : v.vh - 1;
  const int sp = 8 * sy + sx;
  int py, pcb, pcr;
  if constexpr (MODE == 2) {
    idct_component(s, cc, kk, lane, s.cpx);
    fill_s10(q, cc, s.cpx, ring + 100 * (cc - 1), lane);
    const int px = upsampled_pixel(q.s10[cc - 1], 8 * v.off_x + sx, 8 * v.off_y + sy);
    py = s.ycc[0][sp];
    pcb = (cc == 1 ? px : s.ycc[1][sp]) - 128;
    pcr = (cc == 2 ? px : s.ycc[2][sp]) - 128;
  } else {
    // the changed component into plane 3 of the pixel cache; which plane a channel reads is a
    // wavefront-uniform offset (no branch, one byte read each)
    idct_component(s, cc, kk, lane, s.ycc[3]);
    const unsigned char* px = &s.ycc[0][0];
    py = px[(cc == 0 ? 192 : 0) + sp];
    pcb = (int)px[(cc == 1 ? 192 : 64) + sp] - 128;
    pcr = (int)px[(cc == 2 ? 192 : 128) + sp] - 128;
  }
  const int half = 1 << 15;
  const int r = clamp255(py + ((GZ_MUL24(91881, pcr) + half) >> 16));
  const int g = clamp255(py + ((GZ_MUL24(-46802, pcr) + (GZ_MUL24(-22554, pcb) + half)) >> 16));
  const int b = clamp255(py + ((GZ_MUL24(116130, pcb) + half) >> 16));
  float x, y, z;
  opsin8x8(s, lane, L.w, GZ_LDG(a.srgb_lut, r), GZ_LDG(a.srgb_lut, g), GZ_LDG(a.srgb_lut, b), &x, &y, &z);
  s.d[slot][0][lane] = L.x0[0] - (double)x;
  s.d[slot][1][lane] = L.x0[1] - (double)y;
  s.d[slot][2][lane] = L.x0[2] - (double)z;
  __syncthreads();
}

// Narrow stages (ButteraugliBlockDiff, butteraugli_comparator.cc:382-411, and CompareBlock's
// masked sum) of the nc <= kEvalBatch candidates whose differences are in s.d[0..nc): every
// candidate's arithmetic is what it would be alone -- lane assignments only -- and its error
// lands in s.err[slot]; s.red[slot] keeps the three channel terms.
GZ_DEVFN void eval_narrow(SearchLds& s, int nc, int lane, const SearchView& v) {
  // mean term in index order: lane -> (slot, channel)
  const int slot9 = lane / 3, ch9 = lane - 3 * slot9;
  double dc_term = 0.0;
  if (lane < 3 * nc) {
    const double* p = s.d[slot9][ch9];
    double sum = 0.0;
    for (int i = 0; i < 64; ++i) sum += p[i];
    const double avg = sum / 64;
    dc_term = (4.0 * avg) * avg;
  }
  __syncthreads();
  // row transforms, in place: task -> (slot, channel, row)
  for (int t = lane; t < 24 * nc; t += 64) {
    double* p = &s.d[t / 24][(t % 24) >> 3][8 * (t & 7)];
    Cpx F[8];
    real_fft8(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], F);
    p[0] = F[0].re; p[1] = F[4].re;
    p[2] = F[1].re; p[3] = F[1].im;
    p[4] = F[2].re; p[5] = F[2].im;
    p[6] = F[3].re; p[7] = F[3].im;
  }
  __syncthreads();
  // column stage: task -> (slot, channel, transposed row k)
  double pw8[8];
  if (lane < 15 * nc) {
    const int sl = lane / 15, r = lane - 15 * sl;
    const int ch = r / 5, k = r - 5 * ch;
    const double* col = s.d[sl][ch];
    Cpx F[8];
    if (k == 0 || k == 4) {
      const int j = k == 0 ? 0 : 1;
      real_fft8(col[j], col[8 + j], col[16 + j], col[24 + j], col[32 + j], col[40 + j],
                col[48 + j], col[56 + j], F);
    } else {
      Cpx in[8];
#pragma unroll
      for (int x2 = 0; x2 < 8; ++x2) {
        in[x2].re = col[8 * x2 + 2 * k];
        in[x2].im = col[8 * x2 + 2 * k + 1];
      }
      fft8(in, F);
    }
#pragma unroll
    for (int x2 = 0; x2 < 8; ++x2) {
      const double pv = F[x2].re * F[x2].re + F[x2].im * F[x2].im;
      pw8[x2] = pv * 0.000064;
    }
  }
  __syncthreads();   // (every column task has read its plane)
  if (lane < 15 * nc) {
    const int sl = lane / 15, r = lane - 15 * sl;
    const int ch = r / 5, k = r - 5 * ch;
#pragma unroll
    for (int x2 = 0; x2 < 8; ++x2) s.d[sl][ch][8 * k + x2] = pw8[x2];
  }
  __syncthreads();
  if (lane < 3 * nc) {
    const double* pw = s.d[slot9][ch9];
    double acc = dc_term;   // diff_xyb[c] starts at 0.0: 0.0 + 4*avg*avg
    for (int i = 4; i < 37; ++i) acc += kCsf8x8[i] * pw[i];
    s.red[slot9][ch9] = acc;
  }
  __syncthreads();
  if (lane < nc) {
    double diff = 0.0;
    diff += s.red[lane][0] * (double)v.m0;
    diff += s.red[lane][1] * (double)v.m1;
    diff += s.red[lane][2] * (double)v.m2;
    s.err[lane] = (float)sqrt(diff);
  }
  __syncthreads();
}

// grid = one workgroup per block of the search grid; 64 threads (MODE 0, 1) or 256 (MODE 2).
// Occupancy decides: the same code at 12 / 14 / 16 wavefronts per CU (LDS padded) runs 27.0 / 23.0 / 22.2 ms
// at 4K (profiles/r05_block_search_variants.log; round 4's kernel: 30.9 at 12) -- hence at most 128
// registers for MODE 0 / 1 (they take 126 / 120 unforced) and a wavefront's LDS below 10 KB.  (Round 4's
// kernel, capped at 128 registers, was 5 % slower at 16 than at 12: it spilled into longer chains.)
// The sRGB table in LDS instead of L1: 22.7 ms, not kept.  MODE 2 (four wavefronts per block, 144 registers
// unforced): capped at 128 it spills nine registers and is still faster, 23.4 -> 20.6 ms for a 4K 4:2:0 frame.
template <int MODE>
__global__ __launch_bounds__(MODE == 2 ? 256 : 64, 4) void k_block_search(SearchArgs a) {
  constexpr int NW = MODE == 2 ? 4 : 1;
  __shared__ SearchLds sh[NW];
  __shared__ SearchLds420<MODE == 2> sq[NW];
  __shared__ int s_ring[MODE == 2 ? 200 : 1];      // neighbours' samples around the block, fixed
  __shared__ float s_err[kEvalBatch][4];
  const int wave = MODE == 2 ? (int)(threadIdx.x >> 6) : 0, lane = threadIdx.x & 63;
  SearchLds& s = sh[wave];
  SearchLds420<MODE == 2>& q = sq[wave];
  const int blk = blockIdx.x;
  const int gw = MODE == 2 ? a.cbw : a.bw;      // width of the search grid
  const int gbx = blk % gw, gby = blk / gw;
  SearchView v;
  v.off_x = wave & 1;
  v.off_y = wave >> 1;
  const int bx = MODE == 2 ? 2 * gbx + v.off_x : gbx, by = MODE == 2 ? 2 * gby + v.off_y : gby;
  const int xmin = 8 * bx, ymin = 8 * by;
  v.in_image = xmin < a.w && ymin < a.h;
  v.vw = a.w - xmin < 8 ? a.w - xmin : 8;   // in-image width / height of the window
  v.vh = a.h - ymin < 8 ? a.h - ymin : 8;
  if (v.vw < 1) v.vw = 1;                   // (sub-blocks outside the image: evaluated, ignored)
  if (v.vh < 1) v.vh = 1;
  const int iy = lane >> 3, ix = lane & 7;
  for (int c = 0; c < 3; ++c) {
    const bool mine = MODE == 0 || (MODE == 1 && c == 0) || (MODE == 2 && c > 0);
    short cv = mine ? a.coeffs[((size_t)a.coff[c] + blk) * 64 + lane] : (short)0;
    if (MODE == 2 && c == 0) {
      // the luma block under this wavefront's 8x8 sub-block (fixed during the chroma search; no
      // candidate of this mode names component 0)
      const int lb = v.in_image ? by * a.bw + bx : 0;
      cv = a.coeffs[((size_t)a.coff[0] + lb) * 64 + lane];
    }
    s.coef[64 * c + lane] = cv;
    s.coefT[64 * c + 8 * ix + iy] = cv;
  }
  const size_t r0 = (size_t)blk * 192;
  int n = a.rank_cnt[blk];
  for (int i = lane; i < n; i += 64) s.list[i] = a.rank_idx[r0 + i];
  __syncthreads();
  const SearchLane L = search_lane_init(s, lane, xmin, ymin, a);
  const int mxs = (a.w - 1) >> 1, mys = (a.h - 1) >> 1;
  if constexpr (MODE == 0) {
    for (int c = 0; c < 3; ++c) idct_component(s, c, -1, lane, s.ycc[c]);
  } else if constexpr (MODE == 1) {
    idct_component(s, 0, -1, lane, s.ycc[0]);
    const int sw = a.cbw * 8;
    const size_t pl = (size_t)sw * (size_t)(((a.h + 15) >> 4) * 8);
    const int x = xmin + ix < a.w - 1 ? xmin + ix : a.w - 1;
    const int y = ymin + iy < a.h - 1 ? ymin + iy : a.h - 1;
    s.ycc[1][lane] = (unsigned char)chroma420_pixel(a.samples, sw, mxs, mys, x, y);
    s.ycc[2][lane] = (unsigned char)chroma420_pixel(a.samples + pl, sw, mxs, mys, x, y);
    __syncthreads();
  } else {
    idct_component(s, 0, -1, lane, s.ycc[0]);   // luma pixels of this wavefront's 8x8 sub-block
    // the 10x10 neighbourhood: which cells are the block's own samples (after the replication
    // rules, i.e. clamping the sample coordinates into the image), and the others' values
    const int sw = a.cbw * 8;
    const size_t pl = (size_t)sw * (size_t)(((a.h + 15) >> 4) * 8);
    for (int cell = lane; cell < 100; cell += 64) {
      int gx = 8 * gbx + cell % 10 - 1, gy = 8 * gby + cell / 10 - 1;
      gx = gx < 0 ? 0 : (gx > mxs ? mxs : gx);
      gy = gy < 0 ? 0 : (gy > mys ? mys : gy);
      const int lx = gx - 8 * gbx, ly = gy - 8 * gby;
      const bool own = lx >= 0 && lx < 8 && ly >= 0 && ly < 8;
      q.cellsrc[cell] = own ? (short)(8 * ly + lx) : (short)-1;
      if (wave == 0) {
        s_ring[cell] = (int)a.samples[(size_t)gy * sw + gx];
        s_ring[100 + cell] = (int)a.samples[pl + (size_t)gy * sw + gx];
      }
    }
    __syncthreads();
    for (int c = 1; c < 3; ++c) {
      idct_component(s, c, -1, lane, s.cpx);
      fill_s10(q, c, s.cpx, s_ring + 100 * (c - 1), lane);
      s.ycc[c][lane] = (unsigned char)upsampled_pixel(q.s10[c - 1], 8 * v.off_x + ix, 8 * v.off_y + iy);
      __syncthreads();
    }
  }
  {
    const int mb = v.in_image ? by * a.bw + bx : 0;
    v.m0 = a.block_mask[mb];
    v.m1 = a.block_mask[a.nb + mb];
    v.m2 = a.block_mask[2 * a.nb + mb];
  }
  int m = 0;
  while (n > 0) {
    float best_err = 1e17f;
    int best_i = 0;
    const int tries = n < a.lookahead ? n : a.lookahead;
    for (int base = 0; base < tries; base += kEvalBatch) {
      const int nc = tries - base < kEvalBatch ? tries - base : kEvalBatch;
      for (int j = 0; j < nc; ++j) eval_wide<MODE>(s, q, (int)s.list[base + j], j, lane, v, L, s_ring, a);
      eval_narrow(s, nc, lane, v);
      if (MODE == 2) {
        if (lane < nc) s_err[lane][wave] = v.in_image ? s.err[lane] : 0.0f;
        __syncthreads();
      }
      for (int j = 0; j < nc; ++j) {
        float max_err = 0.0f;
        if (MODE == 2) {
          for (int k = 0; k < 4; ++k) max_err = s_err[j][k] > max_err ? s_err[j][k] : max_err;   // std::max(max_err, err)
        } else {
          const float e = s.err[j];
          max_err = e > 0.0f ? e : 0.0f;   // std::max(0, err)
        }
        if (max_err < best_err) {
          best_err = max_err;
          best_i = base + j;
        }
      }
      if (MODE == 2) __syncthreads();
    }
    const int ci = (int)s.list[best_i];
    __syncthreads();
    if (lane == 0) {
      s.coef[ci] = 0;
      s.coefT[(ci & 192) + ((ci & 7) << 3) + ((ci >> 3) & 7)] = 0;
      s.oidx[m] = (unsigned char)ci;
      s.oerr[m] = best_err;
    }
    // erase list[best_i]
    for (int base = 0; base < n; base += 64) {
      const int j = base + lane;
      unsigned char vv = 0;
      const bool mv = j >= best_i && j < n - 1;
      if (mv) vv = s.list[j + 1];
      __syncthreads();
      if (mv) s.list[j] = vv;
      __syncthreads();
    }
    ++m;
    --n;
    if constexpr (MODE == 2) {
      const int c = ci >> 6;
      idct_component(s, c, -1, lane, s.cpx);
      fill_s10(q, c, s.cpx, s_ring + 100 * (c - 1), lane);
      s.ycc[c][lane] = (unsigned char)upsampled_pixel(q.s10[c - 1], 8 * v.off_x + ix, 8 * v.off_y + iy);
      __syncthreads();
    } else {
      idct_component(s, ci >> 6, -1, lane, s.ycc[ci >> 6]);
    }
  }
  __syncthreads();
  // monotone minimum from the end + cut at the block error limit (processor.cc:447-459)
  // (MODE 2: the four wavefronts hold identical lists; the first one writes the result)
  if (lane == 0) {
    float min_err = 1e10f;
    for (int i = m - 1; i >= 0; --i) {
      min_err = s.oerr[i] < min_err ? s.oerr[i] : min_err;
      s.oerr[i] = min_err;
    }
    int num = 0;
    while (num < m && s.oerr[num] <= a.limit) ++num;
    s.num = num;
    if (wave == 0) a.out_cnt[blk] = num;
  }
  __syncthreads();
  const int num = wave == 0 ? s.num : 0;
  for (int i = lane; i < num; i += 64) {
    a.out_idx[(size_t)blk * 192 + i] = s.oidx[i];
    a.out_err[(size_t)blk * 192 + i] = s.oerr[i];
  }
}

// ButteraugliComparator::SwitchBlock + CompareBlock (butteraugli_comparator.cc:427-488) with
// factor 1 for n independent (block position, 3 x 64 coefficients) pairs: the per-block form of
// the Comparator seam (gz_compare_blocks), one wavefront per pair.  Returns the double
// CompareBlock returns (before the caller's cast to float).
__global__ __launch_bounds__(64) void k_compare_blocks(SearchArgs a, const int32_t* __restrict__ block_xy,
                                                       const int16_t* __restrict__ blocks, int n,
                                                       double* __restrict__ out) {
  __shared__ SearchLds s;
  const int i = blockIdx.x, lane = threadIdx.x;
  const int bx = block_xy[2 * i], by = block_xy[2 * i + 1];
  const int xmin = 8 * bx, ymin = 8 * by;
  for (int c = 0; c < 3; ++c) {
    const short cv = blocks[((size_t)i * 3 + c) * 64 + lane];
    s.coef[64 * c + lane] = cv;
    s.coefT[64 * c + 8 * (lane & 7) + (lane >> 3)] = cv;
  }
  __syncthreads();
  const SearchLane L = search_lane_init(s, lane, xmin, ymin, a);
  for (int c = 1; c < 3; ++c) idct_component(s, c, -1, lane, s.ycc[c]);
  SearchView v;
  v.off_x = v.off_y = 0;
  v.in_image = true;
  v.vw = a.w - xmin < 8 ? a.w - xmin : 8;
  v.vh = a.h - ymin < 8 ? a.h - ymin : 8;
  const int mb = by * a.bw + bx;
  v.m0 = a.block_mask[mb];
  v.m1 = a.block_mask[a.nb + mb];
  v.m2 = a.block_mask[2 * a.nb + mb];
  // candidate index 192: the luma component is recomputed with no coefficient zeroed
  SearchLds420<false> q;   // (unused by this mode)
  eval_wide<0>(s, q, 192, 0, lane, v, L, nullptr, a);
  eval_narrow(s, 1, lane, v);
  if (lane == 0) {
    double diff = 0.0;
    diff += s.red[0][0] * (double)v.m0;
    diff += s.red[0][1] * (double)v.m1;
    diff += s.red[0][2] * (double)v.m2;
    out[i] = sqrt(diff);
  }
}

// The same for windows given by their YCbCr PIXELS (what CompareBlock reads of the image:
// OutputImage::ToLinearRGB(xmin, ymin, 8, 8) <- OutputImageComponent::ToPixels, output_image.cc:
// 69-96, edge replication included) -- the form of the seam that serves every frame: the pixels
// of a subsampled component depend on its neighbours' blocks, which only the caller's
// OutputImage holds.  ycc: n x 3 x 64 bytes (Y, Cb, Cr of a window one after the other).
__global__ __launch_bounds__(64) void k_compare_block_pixels(SearchArgs a, const int32_t* __restrict__ block_xy,
                                                             const uint8_t* __restrict__ ycc, int n,
                                                             double* __restrict__ out) {
  __shared__ SearchLds s;
  const int i = blockIdx.x, lane = threadIdx.x;
  const int bx = block_xy[2 * i], by = block_xy[2 * i + 1];
  const int xmin = 8 * bx, ymin = 8 * by;
  const SearchLane L = search_lane_init(s, lane, xmin, ymin, a);
  {
    const uint8_t* p = ycc + (size_t)i * 192;
    int r, g, b;
    ycc_to_rgb((int)p[lane], (int)p[64 + lane], (int)p[128 + lane], &r, &g, &b);
    float x, y, z;
    opsin8x8(s, lane, L.w, GZ_LDG(a.srgb_lut, r), GZ_LDG(a.srgb_lut, g), GZ_LDG(a.srgb_lut, b), &x, &y, &z);
    s.d[0][0][lane] = L.x0[0] - (double)x;
    s.d[0][1][lane] = L.x0[1] - (double)y;
    s.d[0][2][lane] = L.x0[2] - (double)z;
    __syncthreads();
  }
  SearchView v;
  v.off_x = v.off_y = 0;
  v.in_image = true;
  v.vw = v.vh = 8;
  const int mb = by * a.bw + bx;
  v.m0 = a.block_mask[mb];
  v.m1 = a.block_mask[a.nb + mb];
  v.m2 = a.block_mask[2 * a.nb + mb];
  eval_narrow(s, 1, lane, v);
  if (lane == 0) {
    double diff = 0.0;
    diff += s.red[0][0] * (double)v.m0;
    diff += s.red[0][1] * (double)v.m1;
    diff += s.red[0][2] * (double)v.m2;
    out[i] = sqrt(diff);
  }
}

// The search's result as the CSR arrays the caller takes (offsets + packed candidate indices), made on
// the device: the host then copies `total` bytes instead of 192 per block and compacts nothing.
// k_csr_offsets: exclusive scan of n counts by ONE workgroup of 1024 threads, off[0..n].
__global__ __launch_bounds__(1024) void k_csr_offsets(const int32_t* __restrict__ cnt, int n,
                                                      int32_t* __restrict__ off) {
  __shared__ int s_part[1024];
  const int t = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int lo = t * per < n ? t * per : n, hi = lo + per < n ? lo + per : n;
  int sum = 0;
  for (int i = lo; i < hi; ++i) sum += cnt[i];
  s_part[t] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = t >= d ? s_part[t - d] : 0;
    __syncthreads();
    s_part[t] += v;
    __syncthreads();
  }
  int run = s_part[t] - sum;
  for (int i = lo; i < hi; ++i) {
    off[i] = run;
    run += cnt[i];
  }
  if (t == 1023) off[n] = s_part[1023];
}
// one wavefront per block: its cnt[b] candidate indices to out[off[b]..]
__global__ __launch_bounds__(256) void k_csr_pack(const int32_t* __restrict__ cnt, const int32_t* __restrict__ off,
                                                  const uint8_t* __restrict__ idx192, int n,
                                                  uint8_t* __restrict__ out) {
  const int b = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= n) return;
  const int c = cnt[b], o = off[b];
  for (int i = lane; i < c; i += 64) out[o + i] = idx192[(size_t)b * 192 + i];
}

// Picks mask planes at block corners: out[c][blk] = mask[c](8*by, 8*bx).
__global__ __launch_bounds__(256) void k_gather_block_corners(const float* m0, const float* m1,
                                                              const float* m2, int pitch, int bw,
                                                              int nb, float* out) {
  const int blk = blockIdx.x * blockDim.x + threadIdx.x;
  if (blk >= nb) return;
  const size_t idx = (size_t)(8 * (blk / bw)) * pitch + 8 * (blk % bw);
  out[blk] = m0[idx];
  out[nb + blk] = m1[idx];
  out[2 * nb + blk] = m2[idx];
}

}  // namespace gz
