// TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference hot path (see
// gz_oracle.h for the rules and for how its parity is pinned).
//
// Written flat, one function per stage, in the shape the HIP kernels use (pixel- and
// block-parallel loops, separable blur as an x pass then a y pass, Malta taps from a
// table), while keeping every float/double promotion and every accumulation order of
// the reference.  Must be compiled with -ffp-contract=off and without
// fast-math (oracle/Makefile does).  "ref:" comments cite /root/reference paths.

#include "gz_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "malta_offsets.inc"

extern "C" void orc_comparator_block_mask(void* p, float* mask3);

namespace {

typedef std::vector<float> Plane;

// ===================================================================== block path ==

// ref: guetzli/idct.cc:29-38.  Only rows 0..3 are used (as in Compute1dIDCT); rows
// 4..7 follow from the even/odd symmetry out[7-x] = sum_u (-1)^u M[x][u] in[u].
const int kIdctRows[4][8] = {
  {8192, 11363, 10703,   9633,  8192,   6437,   4433,  2260},
  {8192,  9633,  4433,  -2259, -8192, -11362, -10704, -6436},
  {8192,  6437, -4433, -11362, -8192,   2261,  10704,  9633},
  {8192,  2260, -10703, -6436,  8192,   9633,  -4433, -11363},
};

// ref: idct.cc:41-137 (Compute1dIDCT) -- exact int32 sums, order irrelevant.
inline void idct_1d(const int16_t* in, int stride, int out[8]) {
  for (int x = 0; x < 4; ++x) {
    int even = 0, odd = 0;
    for (int u = 0; u < 8; u += 2) even += kIdctRows[x][u] * in[u * stride];
    for (int u = 1; u < 8; u += 2) odd += kIdctRows[x][u] * in[u * stride];
    out[x] = even + odd;
    out[7 - x] = even - odd;
  }
}

// ref: fdct.cc:29-36
const int16_t kFdctRowTab[4][7] = {
  {22725, 21407, 19266, 16384, 12873,  8867, 4520},   // rows 0,4
  {31521, 29692, 26722, 22725, 17855, 12299, 6270},   // rows 1,7
  {29692, 27969, 25172, 21407, 16819, 11585, 5906},   // rows 2,6
  {26722, 25172, 22654, 19266, 15137, 10426, 5315},   // rows 3,5
};
const int kFdctRowSel[8] = {0, 1, 2, 3, 0, 3, 2, 1};  // fdct.cc:230-240

inline int mulhi16(int a, int b) { return (a * b) >> 16; }  // MULT, fdct.cc:150

// ref: fdct.cc:68-145 (COLUMN_DCT8) as straight-line code on one column.
inline void fdct_column(int16_t* col /*stride 8*/) {
  const int i0 = col[0], i1 = col[8], i2 = col[16], i3 = col[24];
  const int i4 = col[32], i5 = col[40], i6 = col[48], i7 = col[56];
  int d07 = i0 - i7, s07 = i0 + i7;
  int d25 = i2 - i5, s25 = i2 + i5;
  int d34 = i3 - i4, s34 = i3 + i4;
  int d16 = i1 - i6, s16 = i1 + i6;
  int e0 = s07 - s34, e1 = s07 + s34;   // BUTTERFLY(m7, m4)
  int e2 = s16 - s25, e3 = s16 + s25;   // BUTTERFLY(m6, m5)
  e1 *= 8;
  e3 *= 8;
  col[0] = (int16_t)(e1 + e3);
  col[32] = (int16_t)(e1 - e3);
  e0 *= 8;
  e2 *= 8;
  d34 *= 8;
  d07 *= 8;
  const int kTan1 = 13036, kTan2 = 27146, kTan3m1 = -21746, k2Sqrt2 = 23170;
  col[16] = (int16_t)(mulhi16(kTan2, e2) + e0);
  col[48] = (int16_t)(mulhi16(kTan2, e0) - e2);
  d25 *= 16;
  d16 *= 16;
  int p = mulhi16(d16 + d25, k2Sqrt2);   // m2 after BUTTERFLY(m1,m2); MULT
  int q = mulhi16(d16 - d25, k2Sqrt2);   // m1
  int m3 = d34 - q, m1 = d34 + q;        // BUTTERFLY(m3, m1)
  int m0 = d07 - p, m2 = d07 + p;        // BUTTERFLY(m0, m2)
  int m7 = m3, m6 = m1;
  m3 = mulhi16(m3, kTan3m1) + m7 + 1;
  m1 = mulhi16(m1, kTan1) + m2 + 1;
  int m4 = mulhi16(kTan3m1, m0) + m0;
  int m5 = mulhi16(kTan1, m2);
  col[8] = (int16_t)m1;
  col[24] = (int16_t)(m0 - m3);
  col[40] = (int16_t)(m7 + m4);
  col[56] = (int16_t)(m5 - m6);
}

// ref: fdct.cc:173-208 (RowDct)
inline void fdct_row(int16_t* in, const int16_t* t) {
  int a[4], b[4];
  for (int k = 0; k < 4; ++k) {
    a[k] = in[k] + in[7 - k];
    b[k] = in[k] - in[7 - k];
  }
  const int C1 = t[0], C2 = t[1], C3 = t[2], C4 = t[3], C5 = t[4], C6 = t[5], C7 = t[6];
  const int c0 = a[0] + a[3], c1 = a[0] - a[3], c2 = a[1] + a[2], c3 = a[1] - a[2];
  in[0] = (int16_t)((C4 * (c0 + c2)) >> 16);
  in[4] = (int16_t)((C4 * (c0 - c2)) >> 16);
  in[2] = (int16_t)((C2 * c1 + C6 * c3) >> 16);
  in[6] = (int16_t)((C6 * c1 - C2 * c3) >> 16);
  in[1] = (int16_t)((C1 * b[0] + C3 * b[1] + C5 * b[2] + C7 * b[3]) >> 16);
  in[3] = (int16_t)((C3 * b[0] - C7 * b[1] - C1 * b[2] - C5 * b[3]) >> 16);
  in[5] = (int16_t)((C5 * b[0] - C1 * b[1] + C7 * b[2] + C3 * b[3]) >> 16);
  in[7] = (int16_t)((C7 * b[0] - C5 * b[1] + C3 * b[2] - C1 * b[3]) >> 16);
}

// ref: color_transform.h:211-219; tables regenerated from the libjpeg formulas
// (tools/gen_tables.py checks them against the reference's literal tables).
inline uint8_t clamp255(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
inline void ycc_to_rgb(int y, int cb, int cr, uint8_t* rgb) {
  const int half = 1 << 15;
  rgb[0] = clamp255(y + ((91881 * (cr - 128) + half) >> 16));
  rgb[1] = clamp255(y + ((-46802 * (cr - 128) + (-22554 * (cb - 128) + half)) >> 16));
  rgb[2] = clamp255(y + ((116130 * (cb - 128) + half) >> 16));
}

// ref: gamma_correct.cc:23-38
const double* srgb_table() {
  static double table[256];
  static bool init = false;
  if (!init) {
    int i = 0;
    for (; i < 11; ++i) table[i] = i / 12.92;
    for (; i < 256; ++i) table[i] = 255.0 * std::pow(((i / 255.0) + 0.055) / 1.055, 2.4);
    init = true;
  }
  return table;
}

// ref: quantize.h:24-29
inline int16_t quantize_coeff(int16_t raw, int quant) {
  const int r = raw % quant;
  const int16_t delta =
      (int16_t)(2 * r > quant ? quant - r : (-2) * r > quant ? -quant - r : -r);
  return (int16_t)(raw + delta);
}

// ============================================================== butteraugli: blur ==

struct BlurPlan {
  int r;
  std::vector<float> k;    // unnormalised taps   (ComputeKernel)
  std::vector<float> ks;   // taps * (1/sum)      (Convolution interior)
  float wsum;
};

// ref: butteraugli.cc:145-154.  <math.h> is included there, so fabs/exp resolve to the
// float overloads.
BlurPlan make_plan(float sigma) {
  BlurPlan p;
  const float m = 2.25;
  const float scaler = -1.0 / (2 * sigma * sigma);
  const int diff = std::max<int>(1, m * std::fabs(sigma));
  p.r = diff;
  p.k.resize(2 * diff + 1);
  for (int i = -diff; i <= diff; ++i) p.k[i + diff] = std::exp(scaler * i * i);
  float w = 0.0f;
  for (size_t j = 0; j < p.k.size(); ++j) w += p.k[j];
  p.wsum = w;
  const float s = 1.0f / w;
  p.ks = p.k;
  for (size_t j = 0; j < p.ks.size(); ++j) p.ks[j] *= s;
  return p;
}

// One output sample of Convolution (interior, :207-218) / ConvolveBorderColumn
// (:156-181) along a line of `n` samples with stride `st`, at position x.
inline float conv_at(const float* line, int st, int n, int x, const BlurPlan& p,
                     float border_ratio) {
  const int r = p.r;
  if (x >= r && x < n - r) {
    float sum = 0.0f;
    const float* s = line + (x - r) * st;
    for (int j = 0; j <= 2 * r; ++j) sum += s[j * st] * p.ks[j];
    return sum;
  }
  const int lo = x < r ? 0 : x - r;
  const int hi = std::min(n - 1, x + r);
  float weight = 0.0f;
  for (int j = lo; j <= hi; ++j) weight += p.k[j - x + r];
  weight = (1.0f - border_ratio) * weight + border_ratio * p.wsum;
  const float scale = 1.0f / weight;
  float sum = 0.0f;
  for (int j = lo; j <= hi; ++j) sum += line[j * st] * p.k[j - x + r];
  return sum * scale;
}

// ref: butteraugli.cc:229-233 (Blur = Convolution along x, then along y).
void blur(const float* in, int w, int h, float sigma, float border_ratio, float* out) {
  const BlurPlan p = make_plan(sigma);
  Plane tmp((size_t)w * h);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
      tmp[(size_t)y * w + x] = conv_at(in + (size_t)y * w, 1, w, x, p, border_ratio);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
      out[(size_t)y * w + x] = conv_at(tmp.data() + x, w, h, y, p, border_ratio);
}

// ============================================================= butteraugli: opsin ==

// ref: butteraugli.h:498-534 (V = float: constants rounded to float first)
inline void opsin_absorbance(float r, float g, float b, float out[3]) {
  const float m0 = 0.254462330846, m1 = 0.488238255095, m2 = 0.0635278003854,
              m3 = 1.01681026909;
  const float m4 = 0.195214015766, m5 = 0.568019861857, m6 = 0.0860755536007,
              m7 = 1.1510118369;
  const float m8 = 0.07374607900105684, m9 = 0.06142425304154509,
              m10 = 0.24416850520714256, m11 = 1.20481945273;
  out[0] = m0 * r + m1 * g + m2 * b + m3;
  out[1] = m4 * r + m5 * g + m6 * b + m7;
  out[2] = m8 * r + m9 * g + m10 * b + m11;
}

// ref: butteraugli.h:548-591 (ClenshawRecursion, RationalPolynomial::operator())
inline double clenshaw6(double x, const double* c) {
  double b1 = 0.0, b2 = 0.0;
  for (int i = 5; i >= 1; --i) {
    const double x_b1 = x * b1;
    const double t = (x_b1 + x_b1) - b2 + c[i];
    b2 = b1;
    b1 = t;
  }
  const double x_b1 = x * b1;
  return x_b1 - b2 + c[0];
}

// ref: butteraugli.h:601-615 (GammaPolynomial)
inline double gamma_poly(double v) {
  static const double kMin = 0.971783, kMax = 590.188894;
  static const double p[6] = {98.7821300963361, 164.273222212631, 92.948112871376,
                              33.8165311212688, 6.91626704983562, 0.556380877028234};
  static const double q[6] = {1, 1.64339473427892, 0.89392405219969,
                              0.298947051776379, 0.0507146002577288,
                              0.00226495093949756};
  const double x01 = (v - kMin) / (kMax - kMin);
  const double xc = 2.0 * x01 - 1.0;
  const double yp = clenshaw6(xc, p);
  const double yq = clenshaw6(xc, q);
  if (yq == 0.0) return 0.0;
  return static_cast<float>(yp / yq);
}

// ref: butteraugli.cc:324-366 (OpsinDynamicsImage)
void opsin(const float* rgb, int w, int h, float* xyb) {
  const size_t n = (size_t)w * h;
  std::vector<Plane> bl(3, Plane(n));
  const double kSigma = 1.2;
  for (int c = 0; c < 3; ++c) blur(rgb + c * n, w, h, kSigma, 0.0f, bl[c].data());
  for (size_t i = 0; i < n; ++i) {
    float pre[3], cur[3], sens[3];
    opsin_absorbance(bl[0][i], bl[1][i], bl[2][i], pre);
    for (int c = 0; c < 3; ++c) sens[c] = gamma_poly(pre[c]) / pre[c];
    opsin_absorbance(rgb[i], rgb[n + i], rgb[2 * n + i], cur);
    for (int c = 0; c < 3; ++c) cur[c] *= sens[c];
    xyb[i] = cur[0] - cur[1];
    xyb[n + i] = cur[0] + cur[1];
    xyb[2 * n + i] = cur[2];
  }
}

// =================================================== butteraugli: frequency bands ==

inline float remove_range(float w, float x) {   // butteraugli.cc:369
  return x > w ? x - w : x < -w ? x + w : 0.0f;
}
inline float amplify_range(float w, float x) {  // :374
  return x > w ? x + w : x < -w ? x - w : 2.0f * x;
}
inline float maximum_clamp(float v, float maxval) {  // :432-444
  static const double kMul = 0.688059627878;
  if (v >= maxval) {
    v -= maxval;
    v *= kMul;
    v += maxval;
  } else if (v < -maxval) {
    v += maxval;
    v *= kMul;
    v -= maxval;
  }
  return v;
}
inline float suppress_bright(float hf, float brightness, float mul, float reg) {  // :420-430
  float scaler = mul * reg / (reg + brightness);
  return scaler * hf;
}

struct Psycho {
  Plane lf[3], mf[3], hf[2], uhf[2];
};

// ref: butteraugli.cc:489-622 (SeparateFrequencies)
void separate_frequencies(const float* xyb, int w, int h, Psycho* ps) {
  const size_t n = (size_t)w * h;
  static const double kSigmaLf = 7.46953768697, kSigmaHf = 3.734768843485,
                      kSigmaUhf = 1.8673844217425;
  static const double border_lf = -0.00457628248637, border_mf = -0.271277366628,
                      border_hf = 0.147068973249;
  for (int i = 0; i < 3; ++i) {
    const float* src = xyb + i * n;
    ps->lf[i].resize(n);
    blur(src, w, h, kSigmaLf, border_lf, ps->lf[i].data());
    Plane band(n);
    for (size_t p = 0; p < n; ++p) band[p] = src[p] - ps->lf[i][p];
    ps->mf[i].resize(n);
    blur(band.data(), w, h, kSigmaHf, border_mf, ps->mf[i].data());
    if (i == 2) break;
    ps->hf[i] = band;
    static const double w0 = 0.120079806822, w1 = 0.03430529365;
    for (size_t p = 0; p < n; ++p) {
      ps->hf[i][p] -= ps->mf[i][p];
      ps->mf[i][p] = i == 0 ? remove_range(w0, ps->mf[i][p])
                            : amplify_range(w1, ps->mf[i][p]);
    }
  }
  // SuppressXByY (:470-487), all double per pixel.
  {
    static const double suppress = 2.96534974403, s = 0.745954517135;
    for (size_t p = 0; p < n; ++p) {
      const double xval = ps->hf[0][p];
      const double yval = ps->hf[1][p];
      const double scaler = s + (suppress * (1.0 - s)) / (suppress + yval * yval);
      ps->hf[0][p] = scaler * xval;
    }
  }
  for (int i = 0; i < 2; ++i) {
    ps->uhf[i] = ps->hf[i];
    Plane blurred(n);
    blur(ps->hf[i].data(), w, h, kSigmaUhf, border_hf, blurred.data());
    ps->hf[i] = blurred;
    static const double kRemoveHfRange = 0.0287615200377;
    static const double kMaxclampHf = 78.8223237675;
    static const double kMaxclampUhf = 5.8907152736;
    static const float kMulSuppressHf = 1.10684769012;
    static const float kMulRegHf = 0.478741530298;
    static const float kRegHf = 2000 * kMulRegHf;
    static const float kMulSuppressUhf = 1.76905001176;
    static const float kMulRegUhf = 0.310148420674;
    static const float kRegUhf = 2000 * kMulRegUhf;
    for (size_t p = 0; p < n; ++p) {
      float& uhf = ps->uhf[i][p];
      float& hf = ps->hf[i][p];
      uhf -= hf;
      if (i == 0) {
        hf = remove_range(kRemoveHfRange, hf);
      } else {
        const float br = ps->lf[1][p];   // raw LF-Y, before the vals conversion
        hf = maximum_clamp(hf, kMaxclampHf);
        uhf = maximum_clamp(uhf, kMaxclampUhf);
        uhf = suppress_bright(uhf, br, kMulSuppressUhf, kRegUhf);
        hf = suppress_bright(hf, br, kMulSuppressHf, kRegHf);
      }
    }
  }
  // XybLowFreqToVals (:382-399), V = float.
  {
    const float xmul = 5.57547552483, ymul = 1.20828034498, bmul = 6.08319517575;
    const float y_to_b_mul = -0.628811683685;
    for (size_t p = 0; p < n; ++p) {
      const float x = ps->lf[0][p], y = ps->lf[1][p], b_arg = ps->lf[2][p];
      const float b = b_arg + y_to_b_mul * y;
      ps->lf[2][p] = b * bmul;
      ps->lf[0][p] = x * xmul;
      ps->lf[1][p] = y * ymul;
    }
  }
}

// ============================================================= butteraugli: Malta ==

// ref: butteraugli.cc:1460-1568 (MaltaDiffMapImpl) + :914-1458 (MaltaUnit,
// PaddedMaltaUnit: taps outside the image read 0).
void malta(const float* lum0, const float* lum1, int w, int h, bool lf, double w_0gt1,
           double w_0lt1, double norm1, float* acc) {
  const double len = 3.75;
  const double mulli = lf ? 0.405371989604 : 0.354191303559;  // :1570-1595
  const float kWeight0 = 0.5;
  const float kWeight1 = 0.33;
  const double w_pre0gt1 = mulli * sqrt(kWeight0 * w_0gt1) / (len * 2 + 1);
  const double w_pre0lt1 = mulli * sqrt(kWeight1 * w_0lt1) / (len * 2 + 1);
  const float norm2_0gt1 = w_pre0gt1 * norm1;
  const float norm2_0lt1 = w_pre0lt1 * norm1;
  const size_t n = (size_t)w * h;
  Plane diffs(n);
  for (size_t i = 0; i < n; ++i) {
    const float a = lum0[i], b = lum1[i];
    const float absval = 0.5 * std::abs(a) + 0.5 * std::abs(b);
    const float diff = a - b;
    const float scaler = norm2_0gt1 / (static_cast<float>(norm1) + absval);
    float d = scaler * diff;
    const float scaler2 = norm2_0lt1 / (static_cast<float>(norm1) + absval);
    const double fabs0 = std::fabs(a);
    const double too_small = 0.55 * fabs0;
    const double too_big = 1.05 * fabs0;
    double impact = 0.0;
    bool hit = true;
    if (a < 0) {
      if (b > -too_small) impact = scaler2 * (b + too_small);
      else if (b < -too_big) impact = scaler2 * (-b - too_big);
      else hit = false;
    } else {
      if (b < too_small) impact = scaler2 * (too_small - b);
      else if (b > too_big) impact = scaler2 * (b - too_big);
      else hit = false;
    }
    if (hit) {
      if (diff < 0) d -= impact; else d += impact;
    }
    diffs[i] = d;
  }
  const int ntap = lf ? 5 : 9;
  for (int y = 0; y < h; ++y) {
    for (int x = 0; x < w; ++x) {
      float ret = 0;
      for (int o = 0; o < 16; ++o) {
        const int cnt = lf ? kMaltaLFCount[o] : kMaltaHFCount[o];
        float sum = 0;
        bool first = true;
        for (int t = 0; t < cnt && t < ntap; ++t) {
          const int dy = lf ? kMaltaLF[o][t][0] : kMaltaHF[o][t][0];
          const int dx = lf ? kMaltaLF[o][t][1] : kMaltaHF[o][t][1];
          const int yy = y + dy, xx = x + dx;
          const float v = (yy < 0 || yy >= h || xx < 0 || xx >= w)
                              ? 0.0f : diffs[(size_t)yy * w + xx];
          if (first) { sum = v; first = false; } else { sum += v; }
        }
        ret += sum * sum;
      }
      acc[(size_t)y * w + x] += ret;
    }
  }
}

// ============================================================== butteraugli: mask ==

const double kGlobalScale = 1.0 / 20.35;  // butteraugli.cc:138-139

struct MaskLuts { double x[512], y[512], dcx[512], dcy[512]; };

// ref: butteraugli.cc:1638-1697 (MakeMask + MaskX/MaskY/MaskDcX/MaskDcY)
void make_mask_lut(double extmul, double extoff, double mul, double offset,
                   double scaler, double* lut) {
  for (int i = 0; i < 512; ++i) {
    const double c = mul / ((0.01 * scaler * i) + offset);
    lut[i] = kGlobalScale * (1.0 + extmul * (c + extoff));
    if (lut[i] < 1e-5) lut[i] = 1e-5;
    lut[i] *= lut[i];
  }
}
const MaskLuts& mask_luts() {
  static MaskLuts l;
  static bool init = false;
  if (!init) {
    make_mask_lut(2.59885507073, 3.08805636789, 5.62939030582, 0.315424196682,
                  16.2770141832, l.x);
    make_mask_lut(0.9613705131, -0.581933100068, 6.64307621174, 1.00846207765,
                  2.2342321176, l.y);
    make_mask_lut(10.0470705878, 3.18472654033, 0.373092999662, 0.0551512255218, 70.0,
                  l.dcx);
    make_mask_lut(0.0115640939227, 45.9483175519, 2.52611324247, 0.0142290066313, 5.0,
                  l.dcy);
    init = true;
  }
  return l;
}
// ref: butteraugli.cc:236-251
inline double interp_clamp_neg(const double* a, int size, double ix) {
  if (ix < 0) ix = 0;
  const int base = static_cast<int>(ix);
  if (base >= size - 1) return a[size - 1];
  const double mix = ix - base;
  return a[base] + mix * (a[base + 1] - a[base]);
}

// ref: butteraugli.cc:1699-1739 (DiffPrecompute)
void diff_precompute(const float* p0, const float* p1, int w, int h, float* out) {
  for (int y = 0; y < h; ++y) {
    const int y2 = y + 1 < h ? y + 1 : y > 0 ? y - 1 : y;
    for (int x = 0; x < w; ++x) {
      const int x2 = x + 1 < w ? x + 1 : x > 0 ? x - 1 : x;
      const size_t i = (size_t)y * w + x, ir = (size_t)y * w + x2, id = (size_t)y2 * w + x;
      double sup0 = (std::fabs(p0[i] - p0[ir]) + std::fabs(p0[i] - p0[id]));
      double sup1 = (std::fabs(p1[i] - p1[ir]) + std::fabs(p1[i] - p1[id]));
      static const double mul0 = 0.918416534734;
      float v = mul0 * std::min(sup0, sup1);
      static const double cutoff = 55.0184555849;
      if (v >= cutoff) v = cutoff;
      out[i] = v;
    }
  }
}

// ref: butteraugli.cc:1741-1817 (Mask); xyb0/xyb1 need only planes 0 and 1.
void mask(const float* x0, const float* y0, const float* x1, const float* y1, int w,
          int h, float* mask3, float* mask_dc3) {
  const size_t n = (size_t)w * h;
  const double muls[2] = {0.207017089891, 0.267138152891};
  const double normalizer = 1.0 / (muls[0] + muls[1]);
  static const double r0 = 2.3770330432, r1 = 9.04353323561, r2 = 9.24456601467;
  static const double border_ratio = -0.0724948220913;
  Plane diff(n), mx(n), b1(n), b2(n), my(n);
  diff_precompute(x0, x1, w, h, diff.data());
  blur(diff.data(), w, h, r2, border_ratio, mx.data());
  diff_precompute(y0, y1, w, h, diff.data());
  blur(diff.data(), w, h, r0, border_ratio, b1.data());
  blur(diff.data(), w, h, r1, border_ratio, b2.data());
  for (size_t i = 0; i < n; ++i) {
    const double val = normalizer * (muls[0] * b1[i] + muls[1] * b2[i]);
    my[i] = val;
  }
  static const double mul[2] = {16.6963293877, 2.1364621982};
  static const double w00 = 36.4671237619, w11 = 2.1887170895;
  static const double w_ytob_hf = std::max<double>(0.086624184478, 0.0);
  static const double w_ytob_lf = 21.6804277046;
  static const double p1_to_p0 = 0.0513061271723;
  const MaskLuts& l = mask_luts();
  for (size_t i = 0; i < n; ++i) {
    const double s0 = mx[i];
    const double s1 = my[i];
    const double p1 = mul[1] * w11 * s1;
    const double p0 = mul[0] * w00 * s0 + p1_to_p0 * p1;
    mask3[i] = interp_clamp_neg(l.x, 512, p0);
    mask3[n + i] = interp_clamp_neg(l.y, 512, p1);
    mask3[2 * n + i] = w_ytob_hf * interp_clamp_neg(l.y, 512, p1);
    if (mask_dc3) {
      mask_dc3[i] = interp_clamp_neg(l.dcx, 512, p0);
      mask_dc3[n + i] = interp_clamp_neg(l.dcy, 512, p1);
      mask_dc3[2 * n + i] = w_ytob_lf * interp_clamp_neg(l.dcy, 512, p1);
    }
  }
}

// ref: butteraugli.cc:753-782 (MaskPsychoImage)
void mask_psycho(const Psycho& pi0, const Psycho& pi1, int w, int h, float* mask3,
                 float* mask_dc3) {
  const size_t n = (size_t)w * h;
  static const double muls[4] = {0, 1.64178305129, 0.831081703362, 3.23680933546};
  Plane m0[2], m1[2];
  for (int i = 0; i < 2; ++i) {
    const double a = muls[2 * i], b = muls[2 * i + 1];
    m0[i].resize(n);
    m1[i].resize(n);
    for (size_t p = 0; p < n; ++p) {
      m0[i][p] = a * pi0.uhf[i][p] + b * pi0.hf[i][p];
      m1[i][p] = a * pi1.uhf[i][p] + b * pi1.hf[i][p];
    }
  }
  mask(m0[0].data(), m0[1].data(), m1[0].data(), m1[1].data(), w, h, mask3, mask_dc3);
}

// ======================================================== butteraugli: the diffmap ==

// ref: butteraugli.cc:654-668
void l2diff(const Plane& i0, const Plane& i1, double w, Plane* acc) {
  if (w == 0) return;
  for (size_t p = 0; p < i0.size(); ++p) {
    double diff = i0[p] - i1[p];
    (*acc)[p] += w * diff * diff;
  }
}
// ref: butteraugli.cc:672-714
void l2diff_asym(const Plane& i0, const Plane& i1, double w_0gt1, double w_0lt1,
                 Plane* acc) {
  if (w_0gt1 == 0 && w_0lt1 == 0) return;
  w_0gt1 *= 0.8;
  w_0lt1 *= 0.8;
  for (size_t p = 0; p < i0.size(); ++p) {
    const float a = i0[p], b = i1[p];
    float& out = (*acc)[p];
    double diff = a - b;
    out += w_0gt1 * diff * diff;
    const double fabs0 = std::fabs(a);
    const double too_small = 0.4 * fabs0;
    const double too_big = 1.0 * fabs0;
    if (a < 0) {
      if (b > -too_small) {
        double v = b + too_small;
        out += w_0lt1 * v * v;
      } else if (b < -too_big) {
        double v = -b - too_big;
        out += w_0lt1 * v * v;
      }
    } else {
      if (b < too_small) {
        double v = too_small - b;
        out += w_0lt1 * v * v;
      } else if (b > too_big) {
        double v = b - too_big;
        out += w_0lt1 * v * v;
      }
    }
  }
}
// ref: butteraugli.cc:624-652
void same_noise_levels(const Plane& i0, const Plane& i1, int w, int h, double kSigma,
                       double wgt, double maxclamp, Plane* acc) {
  const size_t n = i0.size();
  Plane t(n), blurred(n);
  for (size_t p = 0; p < n; ++p) {
    double v0 = std::fabs(i0[p]);
    double v1 = std::fabs(i1[p]);
    if (v0 > maxclamp) v0 = maxclamp;
    if (v1 > maxclamp) v1 = maxclamp;
    t[p] = v0 - v1;
  }
  blur(t.data(), w, h, kSigma, 0.0, blurred.data());
  for (size_t p = 0; p < n; ++p) {
    double diff = blurred[p];
    (*acc)[p] += wgt * diff * diff;
  }
}

// ref: butteraugli.cc:817-908 (DiffmapPsychoImage) + :1597-1621 + :718-751
void diffmap_psycho(const Psycho& pi0, const Psycho& pi1, int w, int h, float* result) {
  const size_t n = (size_t)w * h;
  const float hf_asymmetry_ = 0.8f;  // NB: sqrt(hf_asymmetry_) below is the FLOAT sqrt (math.h overload)
  Plane dc[3], ac[3];
  for (int c = 0; c < 3; ++c) {
    dc[c].assign(n, 0.0f);
    ac[c].assign(n, 0.0f);
  }
  static const double wUhfMalta = 5.1409625726, norm1Uhf = 58.5001247061;
  malta(pi0.uhf[1].data(), pi1.uhf[1].data(), w, h, false, wUhfMalta * hf_asymmetry_,
        wUhfMalta / hf_asymmetry_, norm1Uhf, ac[1].data());
  static const double wUhfMaltaX = 4.91743441556, norm1UhfX = 687196.39002;
  malta(pi0.uhf[0].data(), pi1.uhf[0].data(), w, h, false, wUhfMaltaX * hf_asymmetry_,
        wUhfMaltaX / hf_asymmetry_, norm1UhfX, ac[0].data());
  static const double wHfMalta = 153.671655716, norm1Hf = 83150785.9592;
  malta(pi0.hf[1].data(), pi1.hf[1].data(), w, h, true, wHfMalta * std::sqrt(hf_asymmetry_),
        wHfMalta / std::sqrt(hf_asymmetry_), norm1Hf, ac[1].data());
  static const double wHfMaltaX = 668.358918152, norm1HfX = 0.882954368025;
  malta(pi0.hf[0].data(), pi1.hf[0].data(), w, h, true, wHfMaltaX * std::sqrt(hf_asymmetry_),
        wHfMaltaX / std::sqrt(hf_asymmetry_), norm1HfX, ac[0].data());
  static const double wMfMalta = 6841.81248144, norm1Mf = 0.0135134962487;
  malta(pi0.mf[1].data(), pi1.mf[1].data(), w, h, true, wMfMalta, wMfMalta, norm1Mf,
        ac[1].data());
  static const double wMfMaltaX = 813.901703816, norm1MfX = 16792.9322251;
  malta(pi0.mf[0].data(), pi1.mf[0].data(), w, h, true, wMfMaltaX, wMfMaltaX, norm1MfX,
        ac[0].data());
  static const double wmul[9] = {0, 32.4449876135, 0, 0, 0, 0, 1.01370836411, 0,
                                 1.74566011615};
  static const double maxclamp = 85.7047444518, kSigmaHfX = 10.6666499623,
                      wsn = 884.809801415;
  same_noise_levels(pi0.hf[1], pi1.hf[1], w, h, kSigmaHfX, wsn, maxclamp, &ac[1]);
  for (int c = 0; c < 3; ++c) {
    if (c < 2)
      l2diff_asym(pi0.hf[c], pi1.hf[c], wmul[c] * hf_asymmetry_, wmul[c] / hf_asymmetry_,
                  &ac[c]);
    l2diff(pi0.mf[c], pi1.mf[c], wmul[3 + c], &ac[c]);
    l2diff(pi0.lf[c], pi1.lf[c], wmul[6 + c], &dc[c]);
  }
  Plane m(3 * n), mdc(3 * n);
  mask_psycho(pi0, pi1, w, h, m.data(), mdc.data());
  // CombineChannels + the first half of CalculateDiffmap
  Plane d(n), blurred(n);
  static const float kInitialSlope = 100.0f;
  for (size_t p = 0; p < n; ++p) {
    const float a = dc[0][p] * mdc[p] + dc[1][p] * mdc[n + p] + dc[2][p] * mdc[2 * n + p];
    const float b = ac[0][p] * m[p] + ac[1][p] * m[n + p] + ac[2][p] * m[2 * n + p];
    const float v = a + b;
    d[p] = v < (1.0f / (kInitialSlope * kInitialSlope)) ? kInitialSlope * v : std::sqrt(v);
  }
  static const double kSigma = 1.72547472444, mul1 = 0.458794906198;
  static const float scale = 1.0f / (1.0f + mul1);
  static const double border_ratio = 1.0;
  blur(d.data(), w, h, kSigma, border_ratio, blurred.data());
  for (size_t p = 0; p < n; ++p) {
    float v = d[p];
    v += mul1 * blurred[p];
    v *= scale;
    result[p] = v;
  }
}

// ======================================================== guetzli-side comparator ==

struct Comparator {
  int w, h;
  float target;
  std::vector<uint8_t> rgb;
  Psycho pi0;            // butteraugli::ButteraugliComparator::pi0_
};

void linear_from_rgb8(const uint8_t* rgb, int w, int h, float* planes) {
  const double* lut = srgb_table();
  const size_t n = (size_t)w * h;
  for (int c = 0; c < 3; ++c)
    for (size_t p = 0; p < n; ++p) planes[c * n + p] = lut[rgb[3 * p + c]];
}

void reconstruct(const int16_t* coeffs, int w, int h, const int* q, int16_t* coeffs_out,
                 uint8_t* srgb, float* linear) {
  const int bw = (w + 7) / 8, bh = (h + 7) / 8, nb = bw * bh;
  const size_t n = (size_t)w * h;
  const double* lut = srgb_table();
  std::vector<uint8_t> ycc(3 * n);
  for (int c = 0; c < 3; ++c) {
    for (int by = 0; by < bh; ++by) {
      for (int bx = 0; bx < bw; ++bx) {
        int16_t blk[64];
        const size_t off = ((size_t)c * nb + (size_t)by * bw + bx) * 64;
        memcpy(blk, coeffs + off, sizeof(blk));
        if (q) orc_quantize_block(blk, q + 64 * c);
        if (coeffs_out) memcpy(coeffs_out + off, blk, sizeof(blk));
        uint8_t px[64];
        orc_idct_block(blk, px);
        for (int iy = 0; iy < 8; ++iy)
          for (int ix = 0; ix < 8; ++ix) {
            const int x = 8 * bx + ix, y = 8 * by + iy;
            if (x < w && y < h) ycc[((size_t)y * w + x) * 3 + c] = px[8 * iy + ix];
          }
      }
    }
  }
  orc_ycbcr_to_rgb(ycc.data(), (int)n);
  if (srgb) memcpy(srgb, ycc.data(), 3 * n);
  if (linear)
    for (int c = 0; c < 3; ++c)
      for (size_t p = 0; p < n; ++p)
        linear[c * n + p] = static_cast<float>(lut[ycc[3 * p + c]]);
}


// ============================================================ block search (a18, a21) ==

// ref: butteraugli_comparator.cc:93-134 (GetContrastSensitivityMatrix); entries 4..36 are
// the ones ButteraugliBlockDiff reads.
const double kCsf8x8[37] = {
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

struct Cpx { double re, im; };
const double kSqrtHalf = 0.70710678118654752440084436210484903;

// ref: butteraugli_comparator.cc:282-353 (RealFFT8) in single-assignment form, outputs
// already in the final (reordered) positions F[0..7].  Every + - * is one rounding, in
// the reference's association.
inline void real_fft8(const double* a, Cpx* F) {
  const double d26 = a[2] - a[6], s26 = a[6] + a[2];
  const double d04 = a[0] - a[4], s04 = a[4] + a[0];
  const double d15 = a[1] - a[5], s15 = a[5] + a[1];
  const double d37 = a[3] - a[7], s37 = a[7] + a[3];
  const double nd37 = -d37, nd26 = -d26;
  const double m6 = (d15 - d37) * kSqrtHalf;
  const double m1 = (d15 + d37) * kSqrtHalf;
  const double m5 = (nd37 - d15) * kSqrtHalf;
  const double m2 = (nd37 + d15) * kSqrtHalf;
  const double e = s26 + s04, o = s37 + s15;
  const double t3 = s15 - s37, t1 = s04 - s26;
  F[0] = {e + o, 0.0};
  F[1] = {m2 + d04, m5 + nd26};   // pre-reorder out[6]
  F[2] = {t1, -t3};               // out[3]
  F[3] = {d04 - m6, d26 - m1};    // out[5]
  F[4] = {e - o, 0.0};            // out[1]
  F[5] = {d04 - m2, nd26 - m5};   // out[7]
  F[6] = {t1, t3};                // out[2]
  F[7] = {m6 + d04, m1 + d26};    // out[4]
}

// ref: butteraugli_comparator.cc:154-277 (FFT4 + FFT8), complex in, final order out.
inline void fft8(const Cpx* a, Cpx* F) {
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
  const double m6 = (u1 - u4) * kSqrtHalf;
  const double m1 = (u1 + u4) * kSqrtHalf;
  const double m5 = (u2 - u3) * kSqrtHalf;
  const double m2 = (u2 + u3) * kSqrtHalf;
  const Cpx o5 = {a4re - m6, a4im - m1};
  const Cpx o4 = {m6 + a4re, m1 + a4im};
  const Cpx o7 = {a6re - m2, a6im - m5};
  const Cpx o6 = {m2 + a6re, m5 + a6im};
  // FFT4 on (R0,I0) (R1,I1) (R2,I2) (R3,I3)
  const double e = R2 + R0, o = R3 + R1, t1 = R0 - R2, t3 = R1 - R3;
  const double f = I2 + I0, g = I3 + I1, t2 = I0 - I2, t4 = I1 - I3;
  const Cpx o0 = {e + o, f + g};
  const Cpx o1 = {e - o, f - g};
  const Cpx o2 = {t1 - t4, t2 + t3};
  const Cpx o3 = {t1 + t4, t2 - t3};
  F[0] = o0; F[1] = o6; F[2] = o3; F[3] = o5; F[4] = o1; F[5] = o7; F[6] = o2; F[7] = o4;
}

// ref: butteraugli_comparator.cc:357-380 (ButteraugliFFTSquared): fills p[4..36].
void fft_squared(const double* block, double* p) {
  Cpx C[64], T[64];
  for (int y = 0; y < 8; ++y) real_fft8(block + 8 * y, C + 8 * y);
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j) T[8 * i + j] = C[8 * j + i];
  double r0[8], r1[8];
  for (int x = 0; x < 8; ++x) {
    r0[x] = T[x].re;
    r1[x] = T[32 + x].re;
  }
  real_fft8(r0, T);
  real_fft8(r1, T + 32);
  for (int y = 1; y < 4; ++y) {
    Cpx in[8];
    memcpy(in, T + 8 * y, sizeof(in));
    fft8(in, T + 8 * y);
  }
  for (int i = 4; i < 37; ++i) {
    double v = T[i].re * T[i].re + T[i].im * T[i].im;
    v *= 0.000064;
    p[i] = v;
  }
}

// ref: butteraugli_comparator.cc:382-411 (ButteraugliBlockDiff)
void block_diff(const double* xyb0, const double* xyb1, double* diff_xyb) {
  double avg[3] = {0.0, 0.0, 0.0};
  for (int i = 0; i < 192; ++i) avg[i / 64] += xyb0[i] - xyb1[i];
  for (int c = 0; c < 3; ++c) {
    const double a = avg[c] / 64;
    diff_xyb[c] += 4.0 * a * a;
  }
  double d[3][64], p[3][64];
  for (int c = 0; c < 3; ++c) {
    for (int i = 0; i < 64; ++i) d[c][i] = xyb0[64 * c + i] - xyb1[64 * c + i];
    fft_squared(d[c], p[c]);
  }
  for (int i = 4; i < 37; ++i) {
    const double w = kCsf8x8[i];
    diff_xyb[0] += w * p[0][i];
    diff_xyb[1] += w * p[1][i];
    diff_xyb[2] += w * p[2][i];
  }
}

// 8x8 OpsinDynamicsImage of a packed 3x64 linear-RGB block (SwitchBlock :427-455 and
// CompareBlock :466-470 both go through the general image code).
void opsin8x8(const float* lin3x64, float* xyb3x64) { opsin(lin3x64, 8, 8, xyb3x64); }

struct BlockSearch {
  const Comparator* cmp;
  Plane mask;            // mask_xyz_ (3 planes)
  float orig_xyb[192];   // per_block_pregamma_ of the current block
  int bx, by;
};

// ref: butteraugli_comparator.cc:427-455
void switch_block(BlockSearch* bs, int bx, int by) {
  const Comparator* c = bs->cmp;
  bs->bx = bx;
  bs->by = by;
  const double* lut = srgb_table();
  float lin[192];
  for (int iy = 0, i = 0; iy < 8; ++iy)
    for (int ix = 0; ix < 8; ++ix, ++i) {
      const int x = std::min(8 * bx + ix, c->w - 1);
      const int y = std::min(8 * by + iy, c->h - 1);
      const size_t px = (size_t)y * c->w + x;
      for (int ch = 0; ch < 3; ++ch) lin[64 * ch + i] = lut[c->rgb[3 * px + ch]];
    }
  opsin8x8(lin, bs->orig_xyb);
}

// ref: butteraugli_comparator.cc:457-488 (CompareBlock) for a candidate 3x64 coefficient
// block; pixels past the image edge replicate the last in-image column / row
// (OutputImageComponent::ToPixels, output_image.cc:85-96).
double compare_block(const BlockSearch* bs, const int16_t* block192) {
  const Comparator* c = bs->cmp;
  const double* lut = srgb_table();
  uint8_t ycc[192];
  for (int ch = 0; ch < 3; ++ch) orc_idct_block(block192 + 64 * ch, ycc + 64 * ch);
  const int xmin = 8 * bs->bx, ymin = 8 * bs->by;
  float lin[192];
  for (int iy = 0, i = 0; iy < 8; ++iy)
    for (int ix = 0; ix < 8; ++ix, ++i) {
      const int sx = std::min(xmin + ix, c->w - 1) - xmin;
      const int sy = std::min(ymin + iy, c->h - 1) - ymin;
      uint8_t px[3] = {ycc[8 * sy + sx], ycc[64 + 8 * sy + sx], ycc[128 + 8 * sy + sx]};
      ycc_to_rgb(px[0], px[1], px[2], px);
      for (int ch = 0; ch < 3; ++ch) lin[64 * ch + i] = static_cast<float>(lut[px[ch]]);
    }
  float xyb1[192];
  opsin8x8(lin, xyb1);
  double b0[192], b1[192];
  for (int i = 0; i < 192; ++i) {
    b0[i] = bs->orig_xyb[i];
    b1[i] = xyb1[i];
  }
  double diff_xyz[3] = {0.0, 0.0, 0.0};
  block_diff(b0, b1, diff_xyz);
  const size_t n = (size_t)c->w * c->h;
  double diff = 0.0;
  for (int ch = 0; ch < 3; ++ch)
    diff += diff_xyz[ch] * bs->mask[ch * n + (size_t)ymin * c->w + xmin];
  return sqrt(diff);
}

void init_block_search(BlockSearch* bs, const Comparator* c) {
  bs->cmp = c;
  bs->mask.resize((size_t)3 * c->w * c->h);
  orc_comparator_block_mask((void*)c, bs->mask.data());
}

}  // namespace

// ==================================================================== C surface ==
extern "C" {

void orc_fdct_block(int16_t* block) {
  for (int i = 0; i < 8; ++i) fdct_column(block + i);
  for (int r = 0; r < 8; ++r) fdct_row(block + 8 * r, kFdctRowTab[kFdctRowSel[r]]);
}

// ref: idct.cc:139-161
void orc_idct_block(const int16_t* block, uint8_t* out) {
  int16_t cols[64];
  for (int x = 0; x < 8; ++x) {
    int t[8];
    idct_1d(block + x, 8, t);
    for (int y = 0; y < 8; ++y) cols[8 * y + x] = (int16_t)((t[y] + (1 << 10)) >> 11);
  }
  for (int y = 0; y < 8; ++y) {
    int t[8];
    idct_1d(cols + 8 * y, 1, t);
    for (int x = 0; x < 8; ++x)
      out[8 * y + x] = clamp255((t[x] + (257 << 17)) >> 18);
  }
}

// ---- double-precision DCT (a8) ------------------------------------------------
// ref: dct_double.cc:28-45 -- the 10-digit basis values are data of the reference.
static const double kDctBasis[8][8] = {
  { 0.3535533906,  0.3535533906,  0.3535533906,  0.3535533906,  0.3535533906,  0.3535533906,  0.3535533906,  0.3535533906},
  { 0.4903926402,  0.4157348062,  0.2777851165,  0.0975451610, -0.0975451610, -0.2777851165, -0.4157348062, -0.4903926402},
  { 0.4619397663,  0.1913417162, -0.1913417162, -0.4619397663, -0.4619397663, -0.1913417162,  0.1913417162,  0.4619397663},
  { 0.4157348062, -0.0975451610, -0.4903926402, -0.2777851165,  0.2777851165,  0.4903926402,  0.0975451610, -0.4157348062},
  { 0.3535533906, -0.3535533906, -0.3535533906,  0.3535533906,  0.3535533906, -0.3535533906, -0.3535533906,  0.3535533906},
  { 0.2777851165, -0.4903926402,  0.0975451610,  0.4157348062, -0.4157348062, -0.0975451610,  0.4903926402, -0.2777851165},
  { 0.1913417162, -0.4619397663,  0.4619397663, -0.1913417162, -0.1913417162,  0.4619397663, -0.4619397663,  0.1913417162},
  { 0.0975451610, -0.2777851165,  0.4157348062, -0.4903926402,  0.4903926402, -0.4157348062,  0.2777851165, -0.0975451610},
};

// ref: dct_double.cc:47-85.  Both transforms are two sweeps of an 8-point matrix-vector
// product, columns first, each output accumulated from 0.0 over u = 0..7 in order; the
// forward transform uses basis[out][u], the inverse basis[u][out].
void orc_dct_double(double* block, int inverse) {
  double mid[64];
  for (int x = 0; x < 8; ++x)          // column x
    for (int v = 0; v < 8; ++v) {
      double acc = 0.0;
      for (int u = 0; u < 8; ++u)
        acc += (inverse ? kDctBasis[u][v] : kDctBasis[v][u]) * block[8 * u + x];
      mid[8 * v + x] = acc;
    }
  for (int y = 0; y < 8; ++y)          // row y
    for (int v = 0; v < 8; ++v) {
      double acc = 0.0;
      for (int u = 0; u < 8; ++u)
        acc += (inverse ? kDctBasis[u][v] : kDctBasis[v][u]) * mid[8 * y + u];
      block[8 * y + v] = acc;
    }
}

// ref: output_image.cc:99-121 (stride 1, factor 1x1 component)
void orc_to_float_pixels(const int16_t* coeffs, int w, int h, float* out) {
  const int bw = (w + 7) / 8, bh = (h + 7) / 8;
  for (int by = 0; by < bh; ++by)
    for (int bx = 0; bx < bw; ++bx) {
      double d[64];
      for (int k = 0; k < 64; ++k) d[k] = coeffs[((size_t)by * bw + bx) * 64 + k];
      orc_dct_double(d, 1);
      for (int iy = 0; iy < 8 && 8 * by + iy < h; ++iy)
        for (int ix = 0; ix < 8 && 8 * bx + ix < w; ++ix)
          out[(size_t)(8 * by + iy) * w + 8 * bx + ix] = static_cast<float>(d[8 * iy + ix] + 128.0);
    }
}

// ref: output_image.cc:265-300 (and Reset, :40-49, for the block grid of the subsampled
// component).  Returns the number of blocks written.
int orc_set_downsampled(const float* pixels, int w, int h, int fx, int fy, int16_t* out) {
  const int bw = (w + 8 * fx - 1) / (8 * fx), bh = (h + 8 * fy - 1) / (8 * fy);
  for (int by = 0; by < bh; ++by)
    for (int bx = 0; bx < bw; ++bx) {
      double d[64];
      for (int iy = 0; iy < 8; ++iy)
        for (int ix = 0; ix < 8; ++ix) {
          float avg = 0.0;
          for (int j = 0; j < fy; ++j)
            for (int i = 0; i < fx; ++i) {
              const int x = std::min(8 * bx * fx + ix * fx + i, w - 1);
              const int y = std::min(8 * by * fy + iy * fy + j, h - 1);
              avg += pixels[(size_t)y * w + x];
            }
          avg /= fx * fy;
          d[8 * iy + ix] = avg;
        }
      orc_dct_double(d, 0);
      d[0] -= 1024.0;
      for (int k = 0; k < 64; ++k)
        out[((size_t)by * bw + bx) * 64 + k] = static_cast<int16_t>(std::round(d[k]));
    }
  return bw * bh;
}

int orc_quantize_block(int16_t* block, const int* q) {
  int changed = 0;
  for (int k = 0; k < 64; ++k) {
    const int16_t c = quantize_coeff(block[k], q[k]);
    changed |= (c != block[k]);
    block[k] = c;
  }
  return changed;
}

void orc_ycbcr_to_rgb(uint8_t* px, int npix) {
  for (int i = 0; i < npix; ++i) {
    uint8_t* p = px + 3 * i;
    ycc_to_rgb(p[0], p[1], p[2], p);
  }
}

void orc_srgb_to_linear_table(double* out256) {
  memcpy(out256, srgb_table(), 256 * sizeof(double));
}

// ref: jpeg_data_encoder.cc:33-117
int orc_encode_rgb(const uint8_t* rgb, int w, int h, int16_t* coeffs) {
  if (w < 0 || w >= 1 << 16 || h < 0 || h >= 1 << 16) return -1;
  const int bw = (w + 7) / 8, bh = (h + 7) / 8, nb = bw * bh;
  for (int by = 0; by < bh; ++by) {
    for (int bx = 0; bx < bw; ++bx) {
      int16_t blk[192];
      for (int iy = 0; iy < 8; ++iy) {
        for (int ix = 0; ix < 8; ++ix) {
          const int y = std::min(h - 1, 8 * by + iy);
          const int x = std::min(w - 1, 8 * bx + ix);
          const uint8_t* p = rgb + 3 * ((size_t)y * w + x);
          const int r = p[0], g = p[1], b = p[2];
          const int HALF = 1 << 15;
          int16_t* o = blk + 8 * iy + ix;
          o[0] = (int16_t)((19595 * r + 38469 * g + 7471 * b - (128 << 16) + HALF) >> 16);
          o[64] = (int16_t)((-11059 * r - 21709 * g + 32768 * b + HALF - 1) >> 16);
          o[128] = (int16_t)((32768 * r - 27439 * g - 5329 * b + HALF - 1) >> 16);
        }
      }
      for (int c = 0; c < 3; ++c) {
        orc_fdct_block(blk + 64 * c);
        int16_t* dst = coeffs + ((size_t)c * nb + (size_t)by * bw + bx) * 64;
        // Quantize with q = 1: iquant = 65537, (v*iquant + 0x80000) >> 20
        for (int k = 0; k < 64; ++k)
          dst[k] = (int16_t)((blk[64 * c + k] * 65537 + 0x80000) >> 20);
      }
    }
  }
  return 0;
}

void orc_reconstruct(const int16_t* coeffs, int w, int h, const int* q,
                     int16_t* coeffs_out, uint8_t* srgb, float* linear) {
  reconstruct(coeffs, w, h, q, coeffs_out, srgb, linear);
}

int orc_compute_kernel(float sigma, float* taps, int cap) {
  BlurPlan p = make_plan(sigma);
  if ((int)p.k.size() > cap) return -(int)p.k.size();
  memcpy(taps, p.k.data(), p.k.size() * sizeof(float));
  return (int)p.k.size();
}

void orc_blur(const float* in, int w, int h, float sigma, float border_ratio,
              float* out) {
  blur(in, w, h, sigma, border_ratio, out);
}

void orc_opsin(const float* rgb, int w, int h, float* xyb) { opsin(rgb, w, h, xyb); }

void orc_separate_frequencies(const float* xyb, int w, int h, float* out10) {
  Psycho ps;
  separate_frequencies(xyb, w, h, &ps);
  const size_t n = (size_t)w * h;
  for (int i = 0; i < 3; ++i) memcpy(out10 + i * n, ps.lf[i].data(), n * 4);
  for (int i = 0; i < 3; ++i) memcpy(out10 + (3 + i) * n, ps.mf[i].data(), n * 4);
  for (int i = 0; i < 2; ++i) memcpy(out10 + (6 + i) * n, ps.hf[i].data(), n * 4);
  for (int i = 0; i < 2; ++i) memcpy(out10 + (8 + i) * n, ps.uhf[i].data(), n * 4);
}

void orc_mask(const float* xyb0, const float* xyb1, int w, int h, float* m,
              float* mdc) {
  const size_t n = (size_t)w * h;
  mask(xyb0, xyb0 + n, xyb1, xyb1 + n, w, h, m, mdc);
}

void orc_malta(const float* lum0, const float* lum1, int w, int h, int lf,
               double w_0gt1, double w_0lt1, double norm1, float* acc) {
  malta(lum0, lum1, w, h, lf != 0, w_0gt1, w_0lt1, norm1, acc);
}

double orc_diffmap(const float* rgb0, const float* rgb1, int w, int h, float* diffmap) {
  const size_t n = (size_t)w * h;
  Plane xyb0(3 * n), xyb1(3 * n), d(n);
  Psycho p0, p1;
  opsin(rgb0, w, h, xyb0.data());
  separate_frequencies(xyb0.data(), w, h, &p0);
  opsin(rgb1, w, h, xyb1.data());
  separate_frequencies(xyb1.data(), w, h, &p1);
  diffmap_psycho(p0, p1, w, h, d.data());
  float mx = 0.0f;
  for (size_t p = 0; p < n; ++p) mx = std::max(mx, d[p]);
  if (diffmap) memcpy(diffmap, d.data(), n * 4);
  return mx;
}

// ref: butteraugli_comparator.cc:51-61
void* orc_comparator_create(const uint8_t* rgb, int w, int h, float target) {
  Comparator* c = new Comparator;
  c->w = w;
  c->h = h;
  c->target = target;
  c->rgb.assign(rgb, rgb + (size_t)3 * w * h);
  const size_t n = (size_t)w * h;
  Plane lin(3 * n), xyb(3 * n);
  linear_from_rgb8(rgb, w, h, lin.data());
  opsin(lin.data(), w, h, xyb.data());
  separate_frequencies(xyb.data(), w, h, &c->pi0);
  return c;
}
void orc_comparator_destroy(void* p) { delete (Comparator*)p; }

// ref: butteraugli_comparator.cc:63-75 (the dead rgb0 opsin at :64-65 is skipped)
float orc_comparator_compare(void* p, const int16_t* coeffs, float* distmap) {
  Comparator* c = (Comparator*)p;
  const size_t n = (size_t)c->w * c->h;
  Plane lin(3 * n), xyb(3 * n), d(n);
  reconstruct(coeffs, c->w, c->h, nullptr, nullptr, nullptr, lin.data());
  opsin(lin.data(), c->w, c->h, xyb.data());
  Psycho p1;
  separate_frequencies(xyb.data(), c->w, c->h, &p1);
  diffmap_psycho(c->pi0, p1, c->w, c->h, d.data());
  float mx = 0.0f;
  for (size_t i = 0; i < n; ++i) mx = std::max(mx, d[i]);
  if (distmap) memcpy(distmap, d.data(), n * 4);
  return mx;
}

// ref: butteraugli_comparator.cc:494-558
void orc_comparator_block_weights(void* p, int direction, int max_block_dist,
                                  double target_mul, const float* distmap,
                                  float* block_weight) {
  Comparator* c = (Comparator*)p;
  const int w = c->w, h = c->h;
  const double target_distance = c->target * target_mul;
  const int bw = (w + 7) / 8, bh = (h + 7) / 8;
  std::vector<float> bmax((size_t)bw * bh);
  for (int by = 0; by < bh; ++by)
    for (int bx = 0; bx < bw; ++bx) {
      float m = 0.0;
      for (int y = 8 * by; y < std::min(h, 8 * by + 8); ++y)
        for (int x = 8 * bx; x < std::min(w, 8 * bx + 8); ++x)
          m = std::max(m, distmap[(size_t)y * w + x]);
      bmax[(size_t)by * bw + bx] = m;
    }
  for (int by = 0; by < bh; ++by)
    for (int bx = 0; bx < bw; ++bx) {
      const int bix = by * bw + bx;
      float local = static_cast<float>(target_distance);
      const int x0 = std::max(0, bx - max_block_dist), y0 = std::max(0, by - max_block_dist);
      const int x1 = std::min(bw, bx + 1 + max_block_dist);
      const int y1 = std::min(bh, by + 1 + max_block_dist);
      for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) local = std::max(local, bmax[(size_t)y * bw + x]);
      if (direction > 0) {
        if (bmax[bix] <= target_distance && local <= 1.1 * target_distance)
          block_weight[bix] = 1.0;
      } else {
        constexpr double kLocalMaxWeight = 0.5;
        if (bmax[bix] <= (1 - kLocalMaxWeight) * target_distance + kLocalMaxWeight * local)
          continue;
        for (int y = y0; y < y1; ++y)
          for (int x = x0; x < x1; ++x) {
            const int d = std::max(std::abs(y - by), std::abs(x - bx));
            const int ix = y * bw + x;
            block_weight[ix] = std::max<float>(block_weight[ix], 1.0f / (d + 1.0f));
          }
      }
    }
}

// ref: butteraugli_comparator.cc:415-421 (StartBlockComparisons -> mask_xyz_)
void orc_comparator_block_mask(void* p, float* mask3) {
  Comparator* c = (Comparator*)p;
  const size_t n = (size_t)c->w * c->h;
  Plane lin(3 * n), xyb(3 * n);
  linear_from_rgb8(c->rgb.data(), c->w, c->h, lin.data());
  opsin(lin.data(), c->w, c->h, xyb.data());
  mask(xyb.data(), xyb.data() + n, xyb.data(), xyb.data() + n, c->w, c->h, mask3, nullptr);
}


double orc_comparator_compare_block(void* p, const int16_t* coeffs, int bx, int by) {
  Comparator* c = (Comparator*)p;
  BlockSearch bs;
  init_block_search(&bs, c);
  switch_block(&bs, bx, by);
  const int bw = (c->w + 7) / 8, nb = bw * ((c->h + 7) / 8);
  int16_t blk[192];
  for (int ch = 0; ch < 3; ++ch)
    memcpy(blk + 64 * ch, coeffs + ((size_t)ch * nb + (size_t)by * bw + bx) * 64, 128);
  return compare_block(&bs, blk);
}

// ref: processor.cc:364-467 (ComputeBlockZeroingOrder) over all blocks as in
// SelectFrequencyMasking phase A (:554-590), comp_mask 7, 4:4:4.
int orc_block_zeroing_orders(void* p, const int16_t* coeffs, const int16_t* orig,
                             int lookahead, int new_model, int32_t* offsets, uint8_t* idx,
                             float* err, int cap) {
  Comparator* c = (Comparator*)p;
  const int bw = (c->w + 7) / 8, bh = (c->h + 7) / 8, nb = bw * bh;
  BlockSearch bs;
  init_block_search(&bs, c);
  static const uint8_t oldCsf[64] = {
      10, 10, 20, 40, 60, 70, 80, 90, 10, 20, 30, 60, 70, 80, 90, 90,
      20, 30, 60, 70, 80, 90, 90, 90, 40, 60, 70, 80, 90, 90, 90, 90,
      60, 70, 80, 90, 90, 90, 90, 90, 70, 80, 90, 90, 90, 90, 90, 90,
      80, 90, 90, 90, 90, 90, 90, 90, 90, 90, 90, 90, 90, 90, 90, 90};
  static const int zigzag[64] = {
      0, 1, 5, 6, 14, 15, 27, 28, 2, 4, 7, 13, 16, 26, 29, 42,
      3, 8, 12, 17, 25, 30, 41, 43, 9, 11, 18, 24, 31, 40, 44, 53,
      10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60,
      21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};
  static const double kWeight[3] = {1.0, 0.22, 0.20};
  int total = 0;
  for (int by = 0, bix = 0; by < bh; ++by) {
    for (int bx = 0; bx < bw; ++bx, ++bix) {
      int16_t block[192], oblock[192];
      for (int ch = 0; ch < 3; ++ch) {
        memcpy(block + 64 * ch, coeffs + ((size_t)ch * nb + bix) * 64, 128);
        memcpy(oblock + 64 * ch, orig + ((size_t)ch * nb + bix) * 64, 128);
      }
      std::vector<std::pair<int, float> > input_order;
      for (int ch = 0; ch < 3; ++ch)
        for (int k = 1; k < 64; ++k) {
          const int i = ch * 64 + k;
          if (block[i] != 0) {
            float score;
            if (new_model)
              score = std::abs(oblock[i]) * kOrderCsf[i] + kOrderBias[i];
            else
              score = static_cast<float>((std::abs(oblock[i]) - zigzag[k] / 64.0) *
                                         kWeight[ch] / oldCsf[k]);
            input_order.push_back(std::make_pair(i, score));
          }
        }
      std::sort(input_order.begin(), input_order.end(),
                [](const std::pair<int, float>& a, const std::pair<int, float>& b) {
                  return a.second < b.second; });
      int16_t processed[192];
      memcpy(processed, block, sizeof(processed));
      switch_block(&bs, bx, by);
      std::vector<std::pair<int, float> > out;
      while (!input_order.empty()) {
        float best_err = 1e17f;
        int best_i = 0;
        for (size_t i = 0; i < std::min<size_t>(lookahead, input_order.size()); ++i) {
          int16_t cand[192];
          memcpy(cand, processed, sizeof(cand));
          cand[input_order[i].first] = 0;
          float max_err = 0;
          if (8 * bx < c->w && 8 * by < c->h) {
            const float e = static_cast<float>(compare_block(&bs, cand));
            max_err = std::max(max_err, e);
          }
          if (max_err < best_err) {
            best_err = max_err;
            best_i = (int)i;
          }
        }
        const int ci = input_order[best_i].first;
        processed[ci] = 0;
        input_order.erase(input_order.begin() + best_i);
        out.push_back(std::make_pair(ci, best_err));
      }
      float min_err = 1e10;
      for (int i = (int)out.size() - 1; i >= 0; --i) {
        min_err = std::min(min_err, out[i].second);
        out[i].second = min_err;
      }
      size_t num = 0;
      while (num < out.size() && out[num].second <= c->target) ++num;
      offsets[bix] = total;
      for (size_t i = 0; i < num; ++i) {
        if (total < cap) {
          idx[total] = (uint8_t)out[i].first;
          err[total] = out[i].second;
        }
        ++total;
      }
    }
  }
  offsets[nb] = total;
  return total <= cap ? total : -total;
}


// =============================================================== YUV 4:2:0 ==
// Frame layout of a 4:2:0 image across this repository: nb luma blocks (8x8 grid), then
// nbc = ceil(w/16)*ceil(h/16) blocks of Cb, then nbc of Cr.
//
// The restatement keeps the reference's STATEFUL pixel cache: every SetCoeffBlock of a 2x2
// component rebuilds the neighbouring subsampled samples from the current upsampled pixels and
// rewrites an 18x18 window (output_image.cc:146-203).  The product computes the same pixels in
// closed form (gz_kernels_block.h); agreeing with this restatement after arbitrary update
// sequences is what shows that the closed form is right.
namespace {

struct Comp {   // OutputImageComponent, output_image.h:27-97
  int w, h, f, bw, bh;
  std::vector<int16_t> coeffs;
  std::vector<uint16_t> pixels;
  void reset(int w_, int h_, int factor) {   // :35-49
    w = w_; h = h_; f = factor;
    bw = (w + 8 * f - 1) / (8 * f);
    bh = (h + 8 * f - 1) / (8 * f);
    coeffs.assign((size_t)bw * bh * 64, 0);
    pixels.assign((size_t)w * h, 128 << 4);
  }
  // ref: output_image.cc:123-209
  void set_block(int bx, int by, const int16_t* block) {
    memcpy(&coeffs[((size_t)by * bw + bx) * 64], block, 128);
    uint8_t idct[64];
    orc_idct_block(block, idct);
    if (f == 1) {
      for (int iy = 0; iy < 8; ++iy)
        for (int ix = 0; ix < 8; ++ix) {
          const int x = 8 * bx + ix, y = 8 * by + iy;
          if (x < w && y < h) pixels[(size_t)y * w + x] = (uint16_t)(idct[8 * iy + ix] << 4);
        }
      return;
    }
    // the 10x10 subsampled area: rows = the 8 of the block, the one below, the one above;
    // columns = the 8 of the block, the one to the right, the one to the left (:150-183)
    uint16_t sub[100];
    for (int j = 0; j < 10; ++j) {
      const int y0 = by * 16 + (j < 9 ? j * 2 : -2);
      for (int i = 0; i < 10; ++i) {
        const int ix = (j < 9 ? (j + 1) * 10 : 0) + (i < 9 ? i + 1 : 0);
        const int x0 = bx * 16 + (i < 9 ? i * 2 : -2);
        if (x0 < 0) sub[ix] = sub[ix + 1];
        else if (y0 < 0) sub[ix] = sub[ix + 10];
        else if (x0 >= w) sub[ix] = sub[ix - 1];
        else if (y0 >= h) sub[ix] = sub[ix - 10];
        else if (i < 8 && j < 8) sub[ix] = (uint16_t)(idct[j * 8 + i] << 4);
        else {   // the inverse of the fancy upsampler on the pixels as they stand
          const int y1 = std::max(y0 - 1, 0), x1 = std::max(x0 - 1, 0);
          sub[ix] = (uint16_t)((pixels[(size_t)y0 * w + x0] * 9 + pixels[(size_t)y1 * w + x1] +
                                pixels[(size_t)y0 * w + x1] * -3 + pixels[(size_t)y1 * w + x0] * -3) >> 2);
        }
      }
    }
    const int xmin = std::max(bx * 16 - 1, 0), xmax = std::min(bx * 16 + 16, w - 1);
    const int ymin = std::max(by * 16 - 1, 0), ymax = std::min(by * 16 + 16, h - 1);
    for (int y = ymin; y <= ymax; ++y) {
      const int r0 = ((y & ~1) / 2 - by * 8 + 1) * 10;
      const int dy = ((y & 1) * 2 - 1) * 10;
      for (int x = xmin; x <= xmax; ++x) {
        const int c0 = (x & ~1) / 2 - bx * 8 + 1;
        const int dx = (x & 1) * 2 - 1;
        const int ix = c0 + r0;
        pixels[(size_t)y * w + x] =
            (uint16_t)((sub[ix] * 9 + sub[ix + dy] * 3 + sub[ix + dx] * 3 + sub[ix + dx + dy]) >> 4);
      }
    }
  }
  // ref: output_image.cc:67-97 (ToPixels): window may extend past the image, edge replicated
  void to_pixels(int xmin, int ymin, int xs, int ys, uint8_t* out, int stride) const {
    for (int iy = 0; iy < ys; ++iy)
      for (int ix = 0; ix < xs; ++ix) {
        const int x = std::min(xmin + ix, w - 1), y = std::min(ymin + iy, h - 1);
        out[(size_t)(iy * xs + ix) * stride] = (uint8_t)((pixels[(size_t)y * w + x] + 8 - (x & 1)) >> 4);
      }
  }
};

struct Image3 {   // OutputImage
  int w, h;
  Comp c[3];
  size_t off[3];   // first block of each component in the frame layout
  void init(int w_, int h_, int chroma_factor) {
    w = w_; h = h_;
    size_t at = 0;
    for (int i = 0; i < 3; ++i) {
      c[i].reset(w, h, i == 0 ? 1 : chroma_factor);
      off[i] = at;
      at += (size_t)c[i].bw * c[i].bh;
    }
  }
  size_t nblocks() const { return off[2] + (size_t)c[2].bw * c[2].bh; }
  void fill(const int16_t* coeffs) {
    for (int i = 0; i < 3; ++i)
      for (int by = 0; by < c[i].bh; ++by)
        for (int bx = 0; bx < c[i].bw; ++bx)
          c[i].set_block(bx, by, coeffs + (off[i] + (size_t)by * c[i].bw + bx) * 64);
  }
  // ref: output_image.cc:232-243,342-346
  void quantize(const int* q) {
    for (int i = 0; i < 3; ++i)
      for (int by = 0; by < c[i].bh; ++by)
        for (int bx = 0; bx < c[i].bw; ++bx) {
          int16_t blk[64];
          memcpy(blk, &c[i].coeffs[((size_t)by * c[i].bw + bx) * 64], 128);
          if (orc_quantize_block(blk, q + 64 * i)) c[i].set_block(bx, by, blk);
        }
  }
  void dump(int16_t* out) const {
    for (int i = 0; i < 3; ++i) memcpy(out + off[i] * 64, c[i].coeffs.data(), c[i].coeffs.size() * 2);
  }
  // ref: output_image.cc:411-440 (ToSRGB / ToLinearRGB of a window)
  void to_srgb(int xmin, int ymin, int xs, int ys, uint8_t* rgb) const {
    for (int i = 0; i < 3; ++i) c[i].to_pixels(xmin, ymin, xs, ys, rgb + i, 3);
    orc_ycbcr_to_rgb(rgb, xs * ys);
  }
};

// CompareBlock (butteraugli_comparator.cc:457-488) of the 8x8 window at (8*bxx, 8*byy) of an
// image, against the original block switch_block() prepared.
double compare_block_image(const BlockSearch* bs, const Image3& img, int bxx, int byy) {
  const Comparator* c = bs->cmp;
  const double* lut = srgb_table();
  uint8_t rgb[192];
  img.to_srgb(8 * bxx, 8 * byy, 8, 8, rgb);
  float lin[192];
  for (int i = 0; i < 64; ++i)
    for (int ch = 0; ch < 3; ++ch) lin[64 * ch + i] = static_cast<float>(lut[rgb[3 * i + ch]]);
  float xyb1[192];
  opsin8x8(lin, xyb1);
  double b0[192], b1[192];
  for (int i = 0; i < 192; ++i) {
    b0[i] = bs->orig_xyb[i];
    b1[i] = xyb1[i];
  }
  double diff_xyz[3] = {0.0, 0.0, 0.0};
  block_diff(b0, b1, diff_xyz);
  const size_t n = (size_t)c->w * c->h;
  double diff = 0.0;
  for (int ch = 0; ch < 3; ++ch)
    diff += diff_xyz[ch] * bs->mask[ch * n + (size_t)(8 * byy) * c->w + 8 * bxx];
  return sqrt(diff);
}

// ---- preprocess_downsample.cc:28-279 ----
typedef std::vector<float> Fl;

Fl convolve2x(const Fl& image, int w, int h, const double* kernel, double mul) {   // :53-83, size 5
  Fl temp = image;
  for (size_t i = 0; i < image.size(); ++i) {
    const int x = (int)(i % w), y = (int)(i / w);
    if (x < 2 || x + 2 >= w) continue;
    float v = 0;
    for (int j = 0; j < 5; ++j) v += static_cast<float>(kernel[j]) * image[(size_t)y * w + x + j - 2];
    temp[i] = v * static_cast<float>(mul);
  }
  Fl result = temp;
  for (size_t i = 0; i < temp.size(); ++i) {
    const int x = (int)(i % w), y = (int)(i / w);
    if (y < 2 || y + 2 >= h) continue;
    float v = 0;
    for (int j = 0; j < 5; ++j) v += static_cast<float>(kernel[j]) * temp[(size_t)(y + j - 2) * w + x];
    result[i] = v * static_cast<float>(mul);
  }
  return result;
}
void normal_kernel(double sigma, double kernel[5], double* mul) {   // :85-100
  double sum = 0;
  for (int i = 0; i < 5; ++i) {
    const double x = 1.0 * i - 2;
    kernel[i] = std::exp(-x * x / (2 * sigma * sigma)) * 0.3989422804014327 / sigma;
    sum += kernel[i];
  }
  *mul = 1.0 / sum;
}
void morph(int w, int h, std::vector<char>* image, bool erode) {   // :110-134
  const std::vector<char> t = *image;
  for (int y = 1; y + 1 < h; ++y)
    for (int x = 1; x + 1 < w; ++x) {
      const size_t i = (size_t)y * w + x;
      const bool all = t[i] && t[i - 1] && t[i + 1] && t[i - w] && t[i + w];
      const bool any = t[i] || t[i - 1] || t[i + 1] || t[i - w] || t[i + w];
      if (erode) { if (!all) (*image)[i] = 0; }
      else if (any) (*image)[i] = 1;
    }
}
// ref: PreProcessChannel, :157-279, with blur and sharpen both on
void preprocess_channel(int w, int h, int channel, float sigma, float amount, Fl yuv[3]) {
  const size_t n = (size_t)w * h;
  for (size_t i = 0; i < n; ++i) {
    yuv[0][i] /= 255.0;
    yuv[1][i] = yuv[1][i] / 255.0f - 0.5f;
    yuv[2][i] = yuv[2][i] / 255.0f - 0.5f;
  }
  std::vector<char> dark(n, 0), red(n, 0), sharp(n, 0), blurm(n, 0);
  for (size_t i = 0; i < n; ++i) {
    const float y = yuv[0][i], u = yuv[1][i], v = yuv[2][i];
    const float r = y + 1.402f * v;
    const float g = y - 0.34414f * u - 0.71414f * v;
    const float b = y + 1.772f * u;
    if (channel == 2 && g < 0.85 && b < 0.85 && r < 0.9) dark[i] = 1;
    if (channel == 1 && r < 0.85 && g < 0.85 && b < 0.9) dark[i] = 1;
    if (channel == 2 && 2.116 * v > -0.34414 * u + 0.2 && 1.402 * v > 1.772 * u + 0.2) red[i] = 1;
    if (channel == 1 && v < 1.263 * u - 0.1 && u > -0.33741 * v) red[i] = 1;
  }
  for (int k = 0; k < 3; ++k) morph(w, h, &dark, true);
  for (int k = 0; k < 3; ++k) morph(w, h, &red, false);
  for (size_t i = 0; i < n; ++i) sharp[i] = red[i] && dark[i];
  const double threshold = (channel == 2 ? 0.02 : 1.0) * 127.5;
  static const double kEdge[9] = {0, -1, 0, -1, 4, -1, 0, -1, 0};
  Fl edge = yuv[channel];   // Convolve2D, :29-50
  for (int y = 1; y + 1 < h; ++y)
    for (int x = 1; x + 1 < w; ++x) {
      float v = 0;
      for (int j = 0; j < 9; ++j)
        v += static_cast<float>(kEdge[j]) * yuv[channel][(size_t)(y + j / 3 - 1) * w + x + j % 3 - 1];
      edge[(size_t)y * w + x] = v;
    }
  for (size_t i = 0; i < n; ++i) {
    if (sharp[i] || !dark[i]) continue;
    if (fabs(edge[i]) < threshold && yuv[2][i] < -0.162 * yuv[1][i]) blurm[i] = 1;
  }
  morph(w, h, &blurm, true);
  morph(w, h, &blurm, true);
  double ks[5], kb[5], ms, mb;
  normal_kernel(sigma, ks, &ms);   // Sharpen: float sigma promoted
  normal_kernel(1.3, kb, &mb);     // Blur: kSigma
  Fl sharpened = convolve2x(yuv[channel], w, h, ks, ms);
  for (size_t i = 0; i < n; ++i) sharpened[i] = yuv[channel][i] + (yuv[channel][i] - sharpened[i]) * amount;
  const Fl blurred = convolve2x(yuv[channel], w, h, kb, mb);
  for (size_t i = 0; i < n; ++i) {
    if (sharp[i]) yuv[channel][i] = sharpened[i];
    else if (blurm[i]) yuv[channel][i] = blurred[i];
  }
  for (size_t i = 0; i < n; ++i) {
    yuv[0][i] *= 255.0;
    yuv[1][i] = (yuv[1][i] + 0.5f) * 255.0f;
    yuv[2][i] = (yuv[2][i] + 0.5f) * 255.0f;
  }
}

}  // namespace

// ref: OutputImage::Downsample (output_image.cc:304-340) with the default DownsampleConfig and
// use_silver_screen == false.  In: 4:4:4 coefficients; out: the 4:2:0 frame (or the unchanged
// 4:4:4 one for a greyscale image).  Returns the number of blocks written.
int orc_downsample(const int16_t* coeffs, int w, int h, int use_silver_screen, int16_t* out) {
  const int bw = (w + 7) / 8, bh = (h + 7) / 8, nb = bw * bh;
  if (use_silver_screen) return -1;   // not restated
  bool grey = true;
  for (size_t i = (size_t)nb * 64; i < (size_t)3 * nb * 64 && grey; ++i) grey = coeffs[i] == 0;
  if (grey) {
    memcpy(out, coeffs, (size_t)3 * nb * 128);
    return 3 * nb;
  }
  Fl yuv[3];
  for (int c = 0; c < 3; ++c) {
    yuv[c].resize((size_t)w * h);
    orc_to_float_pixels(coeffs + (size_t)c * nb * 64, w, h, yuv[c].data());
  }
  preprocess_channel(w, h, 2, 1.3f, 0.5f, yuv);
  preprocess_channel(w, h, 1, 1.3f, 0.5f, yuv);
  memcpy(out, coeffs, (size_t)nb * 128);
  const int nbc = ((w + 15) / 16) * ((h + 15) / 16);
  orc_set_downsampled(yuv[1].data(), w, h, 2, 2, out + (size_t)nb * 64);
  orc_set_downsampled(yuv[2].data(), w, h, 2, 2, out + ((size_t)nb + nbc) * 64);
  return nb + 2 * nbc;
}

void orc_reconstruct420(const int16_t* coeffs, int w, int h, const int* q, int shuffle,
                        int16_t* coeffs_out, uint8_t* srgb, float* linear) {
  Image3 img;
  img.init(w, h, 2);
  img.fill(coeffs);
  if (shuffle) {   // extra updates in a scrambled order: the result must not depend on them
    unsigned rng = (unsigned)shuffle;
    auto next = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
    for (int rep = 0; rep < 2; ++rep)
      for (int c = 1; c < 3; ++c) {
        Comp& comp = img.c[c];
        const int n = comp.bw * comp.bh;
        for (int t = 0; t < n; ++t) {
          const int b = next() % n;
          int16_t junk[64];
          for (int k = 0; k < 64; ++k) junk[k] = (int16_t)((int)(next() % 301) - 150);
          comp.set_block(b % comp.bw, b / comp.bw, junk);
          comp.set_block(b % comp.bw, b / comp.bw, coeffs + (img.off[c] + b) * 64);
        }
      }
  }
  if (q) img.quantize(q);
  if (coeffs_out) img.dump(coeffs_out);
  const size_t n = (size_t)w * h;
  std::vector<uint8_t> rgb(3 * n);
  img.to_srgb(0, 0, w, h, rgb.data());
  if (srgb) memcpy(srgb, rgb.data(), 3 * n);
  if (linear) {
    const double* lut = srgb_table();
    for (int c = 0; c < 3; ++c)
      for (size_t p = 0; p < n; ++p) linear[c * n + p] = static_cast<float>(lut[rgb[3 * p + c]]);
  }
}

float orc_comparator_compare420(void* p, const int16_t* coeffs, float* distmap) {
  Comparator* c = (Comparator*)p;
  const size_t n = (size_t)c->w * c->h;
  Plane lin(3 * n), xyb(3 * n), d(n);
  orc_reconstruct420(coeffs, c->w, c->h, nullptr, 0, nullptr, nullptr, lin.data());
  opsin(lin.data(), c->w, c->h, xyb.data());
  Psycho p1;
  separate_frequencies(xyb.data(), c->w, c->h, &p1);
  diffmap_psycho(c->pi0, p1, c->w, c->h, d.data());
  float mx = 0.0f;
  for (size_t i = 0; i < n; ++i) mx = std::max(mx, d[i]);
  if (distmap) memcpy(distmap, d.data(), n * 4);
  return mx;
}

// ref: butteraugli_comparator.cc:494-558 with factor_x = factor_y = factor
void orc_comparator_block_weights_factor(void* p, int direction, int max_block_dist,
                                         double target_mul, int factor, const float* distmap,
                                         float* block_weight) {
  Comparator* c = (Comparator*)p;
  const int w = c->w, h = c->h, s = 8 * factor;
  const double target_distance = c->target * target_mul;
  const int bw = (w + s - 1) / s, bh = (h + s - 1) / s;
  std::vector<float> bmax((size_t)bw * bh);
  for (int by = 0; by < bh; ++by)
    for (int bx = 0; bx < bw; ++bx) {
      float m = 0.0;
      for (int y = s * by; y < std::min(h, s * (by + 1)); ++y)
        for (int x = s * bx; x < std::min(w, s * (bx + 1)); ++x)
          m = std::max(m, distmap[(size_t)y * w + x]);
      bmax[(size_t)by * bw + bx] = m;
    }
  for (int by = 0; by < bh; ++by)
    for (int bx = 0; bx < bw; ++bx) {
      const int bix = by * bw + bx;
      float local = static_cast<float>(target_distance);
      const int x0 = std::max(0, bx - max_block_dist), y0 = std::max(0, by - max_block_dist);
      const int x1 = std::min(bw, bx + 1 + max_block_dist);
      const int y1 = std::min(bh, by + 1 + max_block_dist);
      for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) local = std::max(local, bmax[(size_t)y * bw + x]);
      if (direction > 0) {
        if (bmax[bix] <= target_distance && local <= 1.1 * target_distance)
          block_weight[bix] = 1.0;
      } else {
        constexpr double kLocalMaxWeight = 0.5;
        if (bmax[bix] <= (1 - kLocalMaxWeight) * target_distance + kLocalMaxWeight * local)
          continue;
        for (int y = y0; y < y1; ++y)
          for (int x = x0; x < x1; ++x) {
            const int d = std::max(std::abs(y - by), std::abs(x - bx));
            const int ix = y * bw + x;
            block_weight[ix] = std::max<float>(block_weight[ix], 1.0f / (d + 1.0f));
          }
      }
    }
}

// ref: processor.cc:364-467 over the grid of SelectFrequencyMasking's phase A (:539-590) for
// any comp_mask, on a 4:4:4 (frame420 == 0) or 4:2:0 frame: every candidate is set into the
// image (SetCoeffBlock of each component in the mask), compared on the factor x factor 8x8
// blocks of its area that lie inside the image, and scored by the largest error.
int orc_block_zeroing_orders_masked(void* p, const int16_t* coeffs, const int16_t* orig,
                                    int frame420, int comp_mask, int lookahead, int new_model,
                                    int32_t* offsets, uint8_t* idx, float* err, int cap) {
  Comparator* c = (Comparator*)p;
  Image3 img;
  img.init(c->w, c->h, frame420 ? 2 : 1);
  img.fill(coeffs);
  int last_c = 0;
  for (int i = 0; i < 3; ++i)
    if (comp_mask & (1 << i)) last_c = i;
  const int factor = img.c[last_c].f;
  const int gw = (c->w + 8 * factor - 1) / (8 * factor), gh = (c->h + 8 * factor - 1) / (8 * factor);
  BlockSearch bs;
  init_block_search(&bs, c);
  static const uint8_t oldCsf[64] = {
      10, 10, 20, 40, 60, 70, 80, 90, 10, 20, 30, 60, 70, 80, 90, 90,
      20, 30, 60, 70, 80, 90, 90, 90, 40, 60, 70, 80, 90, 90, 90, 90,
      60, 70, 80, 90, 90, 90, 90, 90, 70, 80, 90, 90, 90, 90, 90, 90,
      80, 90, 90, 90, 90, 90, 90, 90, 90, 90, 90, 90, 90, 90, 90, 90};
  static const int zigzag[64] = {
      0, 1, 5, 6, 14, 15, 27, 28, 2, 4, 7, 13, 16, 26, 29, 42,
      3, 8, 12, 17, 25, 30, 41, 43, 9, 11, 18, 24, 31, 40, 44, 53,
      10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60,
      21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};
  static const double kWeight[3] = {1.0, 0.22, 0.20};
  int total = 0;
  std::vector<std::vector<float> > sub_xyb((size_t)factor * factor, std::vector<float>(192));
  for (int by = 0, bix = 0; by < gh; ++by) {
    for (int bx = 0; bx < gw; ++bx, ++bix) {
      int16_t block[192] = {0}, oblock[192] = {0};
      for (int ch = 0; ch < 3; ++ch) {
        if (!(comp_mask & (1 << ch))) continue;
        memcpy(block + 64 * ch, coeffs + (img.off[ch] + bix) * 64, 128);
        memcpy(oblock + 64 * ch, orig + (img.off[ch] + bix) * 64, 128);
      }
      std::vector<std::pair<int, float> > input_order;
      for (int ch = 0; ch < 3; ++ch) {
        if (!(comp_mask & (1 << ch))) continue;
        for (int k = 1; k < 64; ++k) {
          const int i = ch * 64 + k;
          if (block[i] != 0) {
            float score;
            if (new_model)
              score = std::abs(oblock[i]) * kOrderCsf[i] + kOrderBias[i];
            else
              score = static_cast<float>((std::abs(oblock[i]) - zigzag[k] / 64.0) *
                                         kWeight[ch] / oldCsf[k]);
            input_order.push_back(std::make_pair(i, score));
          }
        }
      }
      std::sort(input_order.begin(), input_order.end(),
                [](const std::pair<int, float>& a, const std::pair<int, float>& b) {
                  return a.second < b.second; });
      int16_t processed[192];
      memcpy(processed, block, sizeof(processed));
      // SwitchBlock: the original's opsin image of every sub-block
      for (int oy = 0, s = 0; oy < factor; ++oy)
        for (int ox = 0; ox < factor; ++ox, ++s) {
          switch_block(&bs, bx * factor + ox, by * factor + oy);
          memcpy(sub_xyb[s].data(), bs.orig_xyb, sizeof(bs.orig_xyb));
        }
      auto set_blocks = [&](const int16_t* b192) {
        for (int ch = 0; ch < 3; ++ch)
          if (comp_mask & (1 << ch)) img.c[ch].set_block(bx, by, b192 + 64 * ch);
      };
      std::vector<std::pair<int, float> > out;
      while (!input_order.empty()) {
        float best_err = 1e17f;
        int best_i = 0;
        for (size_t i = 0; i < std::min<size_t>(lookahead, input_order.size()); ++i) {
          int16_t cand[192];
          memcpy(cand, processed, sizeof(cand));
          cand[input_order[i].first] = 0;
          set_blocks(cand);
          float max_err = 0;
          for (int oy = 0, s = 0; oy < factor; ++oy)
            for (int ox = 0; ox < factor; ++ox, ++s) {
              const int bxx = bx * factor + ox, byy = by * factor + oy;
              if (8 * bxx < c->w && 8 * byy < c->h) {
                memcpy(bs.orig_xyb, sub_xyb[s].data(), sizeof(bs.orig_xyb));
                const float e = static_cast<float>(compare_block_image(&bs, img, bxx, byy));
                max_err = std::max(max_err, e);
              }
            }
          if (max_err < best_err) {
            best_err = max_err;
            best_i = (int)i;
          }
        }
        const int ci = input_order[best_i].first;
        processed[ci] = 0;
        input_order.erase(input_order.begin() + best_i);
        out.push_back(std::make_pair(ci, best_err));
        set_blocks(processed);
      }
      float min_err = 1e10;
      for (int i = (int)out.size() - 1; i >= 0; --i) {
        min_err = std::min(min_err, out[i].second);
        out[i].second = min_err;
      }
      size_t num = 0;
      while (num < out.size() && out[num].second <= c->target) ++num;
      offsets[bix] = total;
      for (size_t i = 0; i < num; ++i) {
        if (total < cap) {
          idx[total] = (uint8_t)out[i].first;
          err[total] = out[i].second;
        }
        ++total;
      }
      set_blocks(block);   // the image as it was (processor.cc:460-466)
    }
  }
  offsets[gw * gh] = total;
  return total <= cap ? total : -total;
}

}  // extern "C"
