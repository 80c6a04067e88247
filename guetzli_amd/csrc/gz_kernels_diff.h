// Difference-side kernels of ButteraugliComparator::DiffmapPsychoImage
// (butteraugli.cc:817-908): Malta line filters, the L2 / noise terms, the visual mask and
// the channel combination.
#pragma once
#include "gz_common.h"
#include "gz_kernels_block.h"   // GZ_CONST
#include "gz_math.h"
#define GZ_TABLE GZ_CONST
#include "tables_generated.h"

namespace gz {

// ------------------------------------------------------------------ DiffPrecompute --
// Input of the mask blurs.  MaskPsychoImage (butteraugli.cc:753-782) feeds
// a*uhf + b*hf (X: a = 0) of both images; StartBlockComparisons
// (butteraugli_comparator.cc:415-421) feeds the raw XYB planes of the original twice.
struct MaskIn {
  const float* uhf;   // may be null when a == 0 (0*uhf + b*hf == b*hf up to the sign of 0,
                      // which the fabs() differences below cannot see)
  const float* hf;
  double a, b;
  int plain;          // 1: value = hf[idx] unchanged
  GZ_DEVFN float mix(float u, float v) const {   // u = uhf sample (unused when uhf == null)
    if (plain) return v;
    if (uhf == nullptr) return (float)(b * (double)v);
    return (float)(a * (double)u + b * (double)v);
  }
  GZ_DEVFN float operator()(size_t idx) const {
    return mix(uhf ? GZ_LDG(uhf, idx) : 0.0f, GZ_LDG(hf, idx));
  }
  GZ_DEVFN gz_f4 load4(size_t idx) const {
    const gz_f4 v = GZ_LDG4(hf, idx);
    gz_f4 u = v, r;
    if (uhf) u = GZ_LDG4(uhf, idx);
#pragma unroll
    for (int i = 0; i < 4; ++i) r.v[i] = mix(u.v[i], v.v[i]);
    return r;
  }
};
// The two halves of DiffPrecompute apart: the ORIGINAL's |c - right| + |c - down| is the same for
// every candidate of an image, so gz_set_rgb computes it once (k_mask_sup: two planes) and a
// Compare reads those two planes instead of the original's three band planes (k_mask_pre: 5 plane
// reads + 2 writes instead of 6 + 2, a third fewer memory instructions: 84 -> ~60 us at 4K).
struct MaskSupPack {
  MaskIn in[2];     // [X, Y] of one image
  float* out[2];
};
struct MaskPrePack {
  const float* sup0[2];   // k_mask_sup of image 0
  MaskIn in1[2];          // [X, Y] of image 1
  float* out[2];
};

// This thread's four mixed samples of row y at columns x .. x + 3, of the row below (mirrored at
// the last row) and the sample right of the fourth (the next lane's first, or -- last lane of a
// wavefront, last quad of a row -- loaded; mirrored at the last column: butteraugli.cc:1706-1725).
struct MaskQuad { gz_f4 c, d; float r; };
GZ_DEVFN MaskQuad mask_quad(const MaskIn& a, int x, int y, int w, int h, int pitch) {
  const int y2 = y + 1 < h ? y + 1 : (y > 0 ? y - 1 : y);
  MaskQuad q;
  q.c = a.load4((size_t)y * pitch + x);
  q.d = a.load4((size_t)y2 * pitch + x);
  const int lane = (int)(threadIdx.x & 63);
  // (called by the lanes whose whole quad lies inside the image: the next lane takes part iff its
  // quad does too)
  const float next = __uint_as_float((unsigned)__shfl((int)__float_as_uint(q.c.v[0]), lane + 1));
  if (lane != 63 && x + 7 < w) {
    q.r = next;
  } else {
    const int xr = x + 4 < w ? x + 4 : x + 2;
    q.r = a((size_t)y * pitch + xr);
  }
  return q;
}

// grid = (ceil(w/1024), h, 2): a thread takes 4 consecutive pixels of a row -- 16-byte loads of
// the row and of the row below, one 16-byte store -- when the row pitch allows it.  Rows in
// XCD-aware order: row y + 1 (read by this row's workgroups and by the next row's) then comes
// from the same L2.
__global__ __launch_bounds__(256) void k_mask_sup(MaskSupPack pk, int w, int h, int pitch) {
  const GzTile bid = gz_xcd_tile();
  const int x = (bid.x * (int)blockDim.x + (int)threadIdx.x) * 4, y = bid.y;
  if (x >= w || y >= h) return;
  const int c = bid.z;
  MaskIn a = pk.in[0];
  float* out = pk.out[0];
  if (c == 1) { a = pk.in[1]; out = pk.out[1]; }
  const int y2 = y + 1 < h ? y + 1 : (y > 0 ? y - 1 : y);
  if ((pitch & 3) == 0 && x + 3 < w) {
    const MaskQuad q = mask_quad(a, x, y, w, h, pitch);
    gz_f4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o.v[k] = diff_sup(q.c.v[k], k < 3 ? q.c.v[k < 3 ? k + 1 : 3] : q.r, q.d.v[k]);
    GZ_STG4(out, (size_t)y * pitch + x, o);
    return;
  }
  for (int k = 0; k < 4 && x + k < w; ++k) {
    const int xx = x + k;
    const int x2 = xx + 1 < w ? xx + 1 : (xx > 0 ? xx - 1 : xx);
    const size_t i = (size_t)y * pitch + xx;
    out[i] = diff_sup(a(i), a((size_t)y * pitch + x2), a((size_t)y2 * pitch + xx));
  }
}

__global__ __launch_bounds__(256) void k_mask_pre(MaskPrePack pk, int w, int h, int pitch) {
  const GzTile bid = gz_xcd_tile();
  const int x = (bid.x * (int)blockDim.x + (int)threadIdx.x) * 4, y = bid.y;
  if (x >= w || y >= h) return;
  const int c = bid.z;
  MaskIn b = pk.in1[0];
  const float* sup0 = pk.sup0[0];
  float* out = pk.out[0];
  if (c == 1) { b = pk.in1[1]; sup0 = pk.sup0[1]; out = pk.out[1]; }
  const int y2 = y + 1 < h ? y + 1 : (y > 0 ? y - 1 : y);
  if ((pitch & 3) == 0 && x + 3 < w) {
    const size_t i = (size_t)y * pitch + x;
    const gz_f4 s0 = GZ_LDG4(sup0, i);
    const MaskQuad q = mask_quad(b, x, y, w, h, pitch);
    gz_f4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      o.v[k] = diff_from_sups(s0.v[k], diff_sup(q.c.v[k], k < 3 ? q.c.v[k < 3 ? k + 1 : 3] : q.r, q.d.v[k]));
    GZ_STG4(out, i, o);
    return;
  }
  for (int k = 0; k < 4 && x + k < w; ++k) {
    const int xx = x + k;
    const int x2 = xx + 1 < w ? xx + 1 : (xx > 0 ? xx - 1 : xx);
    const size_t i = (size_t)y * pitch + xx;
    out[i] = diff_from_sups(sup0[i], diff_sup(b(i), b((size_t)y * pitch + x2), b((size_t)y2 * pitch + xx)));
  }
}

// ---------------------------------------------------------------------------- Malta --
// One workgroup = 64x32 output pixels.  Per pass the per-pixel "diffs" value
// (MaltaDiffMapImpl, butteraugli.cc:1468-1529) is computed for the tile plus a 4-pixel
// halo straight into LDS (0 outside the image, PaddedMaltaUnit :1439-1457), then every
// thread sums the 16 oriented line filters (MaltaUnit :914-1424, taps and order from
// tables_generated.h) for its 8 pixels.  A channel's passes accumulate in registers in the
// reference's order, followed by that channel's pointwise terms.
constexpr int MW = 64;
constexpr int MH = 32;
constexpr int MPT = MH / 4;

struct MaltaPass {
  const float* p0;
  const float* p1;
  MaltaNorm nm;
  int lf;   // 1: MaltaTagLF (5-tap), 0: MaltaTag (up to 9-tap)
};

template <bool LF>
GZ_DEVFN float malta_unit(const float (*t)[MW + 8], int ly, int lx) {
  // (ly, lx) are tile coordinates of the centre inside the haloed tile.
  float ret = 0.0f;
#pragma unroll
  for (int o = 0; o < 16; ++o) {
    float sum = 0.0f;
    if (LF) {
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const float v = t[ly + kMaltaLF[o][k][0]][lx + kMaltaLF[o][k][1]];
        sum = k == 0 ? v : sum + v;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        if (k < kMaltaHFCount[o]) {
          const float v = t[ly + kMaltaHF[o][k][0]][lx + kMaltaHF[o][k][1]];
          sum = k == 0 ? v : sum + v;
        }
      }
    }
    ret += sum * sum;
  }
  return ret;
}

template <int NPASS>
struct MaltaArgs {
  MaltaPass pass[NPASS];
  float* out;
};

// grid = (ceil(w/MW), ceil(h/MH), 2): blockIdx.z = channel (a0: Y, a1: X) -- the two channels are
// independent, one launch fills the chip better than two.  The loop over a thread's 8 pixels stays a
// loop and the pixels' accumulators live in LDS: one pixel's 16 line sums need a third of the
// registers all eight need side by side (59 VGPRs: 8 wavefronts per SIMD), and the kernel's three
// phases per pass are bound by how many wavefronts a SIMD holds, not by a unit.  (Round 3 measured
// the unrolled form, 120 VGPRs, and the line sums from a per-thread register window, 125 VGPRs and
// 512 threads, against it: profiles/r03_chain_kernel_experiments.log; both were removed in round 4.)
template <int NPASS>
__global__ __launch_bounds__(256, 8) void k_malta_rolled(MaltaArgs<NPASS> a0, MaltaArgs<NPASS> a1, int w,
                                               int h, int pitch) {
  const GzTile bid = gz_xcd_tile();
  const MaltaArgs<NPASS>& a = bid.z ? a1 : a0;
  __shared__ __attribute__((aligned(16))) float tile[MH + 8][MW + 8];
  const int tx = threadIdx.x & 63, tg = threadIdx.x >> 6;
  const int x0 = bid.x * MW, y0 = bid.y * MH;
  __shared__ float accs[MPT][256];
#pragma unroll
  for (int i = 0; i < MPT; ++i) accs[i][threadIdx.x] = 0.0f;   // (own slots: no barrier needed)
  // the haloed tile starts at x0 - 4: rows can be staged with aligned 16-byte loads when the
  // tile lies inside the image horizontally and the pitch allows it
  const bool vec = x0 >= 4 && x0 + MW + 4 <= w && (pitch & 3) == 0;
  for (int ps = 0; ps < NPASS; ++ps) {
    const MaltaPass P = a.pass[ps];
    // the pass's three constants in vector registers: malta_diff adds / multiplies them into every
    // staged sample, and a scalar-register operand makes those 4-cycle instructions (GZ_IN_VGPR)
    MaltaNorm nm = P.nm;
    nm.norm1f = GZ_IN_VGPR(nm.norm1f);
    nm.norm2_0gt1 = GZ_IN_VGPR(nm.norm2_0gt1);
    nm.norm2_0lt1 = GZ_IN_VGPR(nm.norm2_0lt1);
    if (ps > 0) __syncthreads();
    if (vec) {
      constexpr int NV = (MH + 8) * ((MW + 8) / 4);   // 16-byte vectors in the tile
#pragma unroll 1
      for (int k = 0; k < (NV + 255) / 256; ++k) {
        const int i = 256 * k + (int)threadIdx.x;
        if (i < NV) {
          const int ry = i / ((MW + 8) / 4), q = i - ry * ((MW + 8) / 4);
          const int y = y0 - 4 + ry;
          gz_f4 v;
          v.v[0] = v.v[1] = v.v[2] = v.v[3] = 0.0f;
          if (y >= 0 && y < h) {
            const size_t idx = (size_t)y * pitch + (x0 - 4 + 4 * q);
            const gz_f4 u0 = GZ_LDG4(P.p0, idx), u1 = GZ_LDG4(P.p1, idx);
#pragma unroll
            for (int e = 0; e < 4; ++e) v.v[e] = malta_diff(u0.v[e], u1.v[e], nm);
          }
          *reinterpret_cast<gz_f4*>(&tile[ry][4 * q]) = v;
        }
      }
    } else {
      for (int i = threadIdx.x; i < (MH + 8) * (MW + 8); i += 256) {
        const int ry = i / (MW + 8), rx = i - ry * (MW + 8);
        const int x = x0 - 4 + rx, y = y0 - 4 + ry;
        float v = 0.0f;
        if (x >= 0 && x < w && y >= 0 && y < h) {
          const size_t idx = (size_t)y * pitch + x;
          v = malta_diff(GZ_LDG(P.p0, idx), GZ_LDG(P.p1, idx), nm);
        }
        tile[ry][rx] = v;
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int i = 0; i < MPT; ++i) {
      const int ly = tg * MPT + i;
      const float r = P.lf ? malta_unit<true>(tile, ly + 4, tx + 4)
                           : malta_unit<false>(tile, ly + 4, tx + 4);
      accs[i][threadIdx.x] += r;
    }
  }
  const int x = x0 + tx;
  if (x >= w) return;
#pragma unroll
  for (int i = 0; i < MPT; ++i) {
    const int y = y0 + tg * MPT + i;
    if (y >= h) break;
    const size_t idx = (size_t)y * pitch + x;
    const float v = accs[i][threadIdx.x];
    GZ_STG(a.out, idx, v);
  }
}

// -------------------------------------------------- mask LUTs + combine + sqrt stage --
// Mask second half (butteraugli.cc:1780-1816), L2Diff on the LF planes (:899),
// CombineChannels (:1597-1621) and the first half of CalculateDiffmap (:718-735).
struct CombineArgs {
  const float* mask_x_blur;   // blur(diffX, 9.24)
  const float* mask_y_blur1;  // blur(diffY, 2.377)
  const float* mask_y_blur2;  // blur(diffY, 9.04)
  const float* ac0;           // block_diff_ac[0]
  const float* ac1;           // block_diff_ac[1]
  const float* lf0_x;         // pi0.lf[0] / pi1.lf[0] (vals space)
  const float* lf1_x;
  const float* lf0_b;         // pi0.lf[2] / pi1.lf[2]
  const float* lf1_b;
  // Y channel only: SameNoiseLevels second half (butteraugli.cc:644-651) and L2DiffAsymmetric
  // on HF-Y (:672-714), added to block_diff_ac[1] after its three Malta passes.  Applied here
  // rather than in k_malta so that Malta does not wait for the SameNoise blur.
  const float* sn_blur;       // may be null: ac1 is used as is
  const float* hf0_y;
  const float* hf1_y;
  double w_sn, w_0gt1, w_0lt1;
  const double* luts;         // [4][512]: MaskX, MaskY, MaskDcX, MaskDcY
  float* out;                 // sqrt-stage diffmap (input of the final blur)
  float* mask_out[3];         // optional: mask planes (block search / probes), may be null
  float* mask_dc_out[3];      // optional
  unsigned* clear_word;       // optional: the distance accumulator of the chain's last kernel, reset here when the
                              // Compare has no full reconstruction in front (whose first workgroup resets it otherwise)
};

GZ_DEVFN void mask_p0p1(float bx, float by1, float by2, double* p0, double* p1) {
  const double muls0 = 0.207017089891, muls1 = 0.267138152891;
  const double normalizer = 1.0 / (muls0 + muls1);
  const float my = (float)(normalizer * (muls0 * (double)by1 + muls1 * (double)by2));
  const double s0 = (double)bx, s1 = (double)my;
  const double mul0 = 16.6963293877, mul1 = 2.1364621982;
  const double w00 = 36.4671237619, w11 = 2.1887170895, p1_to_p0 = 0.0513061271723;
  *p1 = (mul1 * w11) * s1;
  *p0 = (mul0 * w00) * s0 + p1_to_p0 * (*p1);
}

// One pixel of the stage: the mask values from the three mask blurs, the LF and HF-Y terms,
// CombineChannels and the square-root stage.  Writes the optional mask planes itself; returns
// the value of the sqrt-stage diffmap.
GZ_DEVFN float combine_px(const CombineArgs& a, size_t i, float mxb, float myb1, float myb2, float ac0,
                          float ac1_in, float lf0x, float lf1x, float lf0b, float lf1b, float snb,
                          float hf0y, float hf1y) {
  double p0, p1;
  mask_p0p1(mxb, myb1, myb2, &p0, &p1);
  const double w_ytob_hf = 0.086624184478, w_ytob_lf = 21.6804277046;
  const float m0 = (float)interp_lut512(a.luts, p0);
  const double my = interp_lut512(a.luts + 512, p1);
  const float m1 = (float)my;
  const float mdc0 = (float)interp_lut512(a.luts + 1024, p0);
  const double mdcy = interp_lut512(a.luts + 1536, p1);
  const float mdc1 = (float)mdcy;
  const float mdc2 = (float)(w_ytob_lf * mdcy);
  if (a.mask_out[0]) {
    a.mask_out[0][i] = m0;
    a.mask_out[1][i] = m1;
    a.mask_out[2][i] = (float)(w_ytob_hf * my);
  }
  if (a.mask_dc_out[0]) {
    a.mask_dc_out[0][i] = mdc0;
    a.mask_dc_out[1][i] = mdc1;
    a.mask_dc_out[2][i] = mdc2;
  }
  if (a.out == nullptr) return 0.0f;
  // block_diff_dc: only X (wmul[6]) and B (wmul[8]) are non-zero (butteraugli.cc:873-883)
  const float dc0 = l2diff_acc(0.0f, lf0x, lf1x, 1.01370836411);
  const float dc1 = 0.0f;
  const float dc2 = l2diff_acc(0.0f, lf0b, lf1b, 1.74566011615);
  const float ac2 = 0.0f;
  float ac1 = ac1_in;
  if (a.sn_blur) {
    const double d = (double)snb;
    ac1 = (float)((double)ac1 + (a.w_sn * d) * d);
    ac1 = l2diff_asym_acc(ac1, hf0y, hf1y, a.w_0gt1, a.w_0lt1);
  }
  const float m2 = (float)(w_ytob_hf * my);
  // CombineChannels: DotProduct(diff_dc, dc_mask) + DotProduct(diff_ac, mask)
  const float sdc = (dc0 * mdc0 + dc1 * mdc1) + dc2 * mdc2;
  const float sac = (ac0 * m0 + ac1 * m1) + ac2 * m2;
  const float v = sdc + sac;
  const float kInitialSlope = 100.0f;
  return v < (1.0f / (kInitialSlope * kInitialSlope)) ? kInitialSlope * v : sqrtf(v);
}

// grid = (ceil(w / (4 * 256)), h): a thread takes 4 consecutive pixels of a row -- the twelve
// input planes and the output move as 16-byte accesses -- when the pitch allows it.
__global__ __launch_bounds__(256) void k_combine(CombineArgs a, int w, int h, int pitch) {
  const int x = (blockIdx.x * blockDim.x + threadIdx.x) * 4, y = blockIdx.y;
  if (a.clear_word && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *a.clear_word = 0u;
  if (x >= w || y >= h) return;
  const size_t i = (size_t)y * pitch + x;
  if ((pitch & 3) == 0 && x + 3 < w && a.out != nullptr) {
    const gz_f4 mxb = GZ_LDG4(a.mask_x_blur, i), myb1 = GZ_LDG4(a.mask_y_blur1, i), myb2 = GZ_LDG4(a.mask_y_blur2, i);
    const gz_f4 ac0 = GZ_LDG4(a.ac0, i), ac1 = GZ_LDG4(a.ac1, i);
    const gz_f4 lf0x = GZ_LDG4(a.lf0_x, i), lf1x = GZ_LDG4(a.lf1_x, i), lf0b = GZ_LDG4(a.lf0_b, i), lf1b = GZ_LDG4(a.lf1_b, i);
    gz_f4 snb = ac0, hf0y = ac0, hf1y = ac0;
    if (a.sn_blur) { snb = GZ_LDG4(a.sn_blur, i); hf0y = GZ_LDG4(a.hf0_y, i); hf1y = GZ_LDG4(a.hf1_y, i); }
    gz_f4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      o.v[k] = combine_px(a, i + k, mxb.v[k], myb1.v[k], myb2.v[k], ac0.v[k], ac1.v[k], lf0x.v[k], lf1x.v[k],
                          lf0b.v[k], lf1b.v[k], snb.v[k], hf0y.v[k], hf1y.v[k]);
    GZ_STG4(a.out, i, o);
    return;
  }
  for (int k = 0; k < 4 && x + k < w; ++k) {
    const size_t j = i + k;
    const bool full = a.out != nullptr;   // (mask-only calls read the three blurs alone)
    const float v = combine_px(a, j, a.mask_x_blur[j], a.mask_y_blur1[j], a.mask_y_blur2[j],
                               full ? a.ac0[j] : 0.0f, full ? a.ac1[j] : 0.0f,
                               full ? a.lf0_x[j] : 0.0f, full ? a.lf1_x[j] : 0.0f,
                               full ? a.lf0_b[j] : 0.0f, full ? a.lf1_b[j] : 0.0f,
                               full && a.sn_blur ? a.sn_blur[j] : 0.0f,
                               full && a.sn_blur ? a.hf0_y[j] : 0.0f, full && a.sn_blur ? a.hf1_y[j] : 0.0f);
    if (full) a.out[j] = v;
  }
}

// Arithmetic self-check (gz_probe_arith).
__global__ void k_probe_arith(int op, const void* a, const void* b, const void* c, void* out,
                              int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  switch (op) {
    case 0: ((float*)out)[i] = ((const float*)a)[i] / ((const float*)b)[i]; break;
    case 1: ((float*)out)[i] = sqrtf(((const float*)a)[i]); break;
    case 2: ((double*)out)[i] = ((const double*)a)[i] / ((const double*)b)[i]; break;
    case 3: ((double*)out)[i] = sqrt(((const double*)a)[i]); break;
    case 4: ((float*)out)[i] = ((const float*)a)[i] * ((const float*)b)[i] + ((const float*)c)[i]; break;
    case 5: ((double*)out)[i] = ((const double*)a)[i] * ((const double*)b)[i] + ((const double*)c)[i]; break;
    case 6: ((float*)out)[i] = (float)((const double*)a)[i]; break;
    default: break;
  }
}

}  // namespace gz
