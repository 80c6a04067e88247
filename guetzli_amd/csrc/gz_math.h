// Per-pixel arithmetic of the butteraugli chain, as device functions.
//
// Bit-exactness contract: the reference is x86-64 SSE2 code without FMA
// and without fast-math, so every operation rounds once to its static C++ type.  This
// file is compiled with -ffp-contract=off; every expression below keeps the reference's
// operand types, promotions and association.  gfx950 provides correctly rounded
// f32/f64 + - * / sqrt and conversions, and denormals are preserved (HIP default).
// All transcendental work (exp for blur taps, pow for the sRGB LUT, the mask LUTs) is
// done on the host and uploaded.
#pragma once
#include "gz_common.h"

namespace gz {

// ---- OpsinAbsorbance<float>, butteraugli.h:498-534 -------------------------------
GZ_DEVFN void opsin_absorbance(float r, float g, float b, float* o0, float* o1,
                               float* o2) {
  const float m0 = (float)0.254462330846, m1 = (float)0.488238255095,
              m2 = (float)0.0635278003854, m3 = (float)1.01681026909;
  const float m4 = (float)0.195214015766, m5 = (float)0.568019861857,
              m6 = (float)0.0860755536007, m7 = (float)1.1510118369;
  const float m8 = (float)0.07374607900105684, m9 = (float)0.06142425304154509,
              m10 = (float)0.24416850520714256, m11 = (float)1.20481945273;
  *o0 = ((m0 * r + m1 * g) + m2 * b) + m3;
  *o1 = ((m4 * r + m5 * g) + m6 * b) + m7;
  *o2 = ((m8 * r + m9 * g) + m10 * b) + m11;
}

// ---- GammaPolynomial, butteraugli.h:548-615 (Clenshaw, degree 5/5, all double) ----
// The reference's recursion, statement by statement (the form the host check compares with).
GZ_DEVFN double clenshaw6_plain(double x, double c0, double c1, double c2, double c3, double c4,
                                double c5) {
  double b1 = 0.0, b2 = 0.0, xb, t;
  xb = x * b1; t = ((xb + xb) - b2) + c5; b2 = b1; b1 = t;
  xb = x * b1; t = ((xb + xb) - b2) + c4; b2 = b1; b1 = t;
  xb = x * b1; t = ((xb + xb) - b2) + c3; b2 = b1; b1 = t;
  xb = x * b1; t = ((xb + xb) - b2) + c2; b2 = b1; b1 = t;
  xb = x * b1; t = ((xb + xb) - b2) + c1; b2 = b1; b1 = t;
  xb = x * b1;
  return (xb - b2) + c0;
}
// The same values with 14 operations instead of 23 (x finite, x2 = x + x):
//  * the first step has b1 = b2 = 0: (x*0 + x*0) - 0 + c5 is c5;
//  * the second has b2 = 0: y - 0 is y;
//  * (x*b + x*b) is 2 * RN(x*b), and so is RN((x + x) * b): scaling by two commutes with
//    rounding as long as the product is a normal number, and x*b is either exactly 0 or
//    (x of order 1, b a sum of order-1 constants, at worst cancelled down to 2^-52) far from
//    the denormal range.
// tests/cpp/verify_opsin_divisions.cc compares gamma_poly_f built on this form with the plain
// one for EVERY float argument.
GZ_DEVFN double clenshaw6(double x, double x2, double c0, double c1, double c2, double c3,
                          double c4, double c5) {
  double b2 = c5;
  double b1 = x2 * c5 + c4;
  double t;
  t = (x2 * b1 - b2) + c3; b2 = b1; b1 = t;
  t = (x2 * b1 - b2) + c2; b2 = b1; b1 = t;
  t = (x2 * b1 - b2) + c1; b2 = b1; b1 = t;
  return (x * b1 - b2) + c0;
}

// float(yp / yq) of RationalPolynomial::operator() (butteraugli.h:581-591) -- what its
// `static_cast<float>` leaves before the caller widens it again.
//
// Two of the three FP64 divisions per channel of the reference's sensitivity are evaluated by
// cheaper sequences that give the SAME bits (tests/cpp/verify_opsin_divisions.cc):
//  * x01 = (v - kMin) / (kMax - kMin) divides by a constant b: with y = RN(1/b),
//    q = a*y, r = fma(-b, q, a) (exact), q' = fma(r, y, q) is the correctly rounded a/b --
//    verified for every finite float v exhaustively (2^32 inputs, 0 mismatches);
//  * float(double(g) / double(p)) for float g, p is the float quotient g / p: a double has
//    53 >= 2*24 + 2 digits, which makes the second rounding innocuous (Figueroa); opsin_pixel
//    below divides in f32 (about 10 instructions against about 30 for an FP64 division).
// The division yp / yq of two computed doubles stays an FP64 division.
GZ_DEVFN float gamma_poly_f(double v) {
  const double kMin = 0.971783, kMax = 590.188894;
  const double b = kMax - kMin;
  const double y = 1.0 / b;
  const double a = v - kMin;
  const double q = a * y;
  const double r = __builtin_fma(-b, q, a);
  const double x01 = __builtin_fma(r, y, q);
  const double xc = 2.0 * x01 - 1.0;
  double yp, yq;
  if (xc * 0.0 == 0.0) {   // finite (always, for the finite absorbances the chain produces)
    const double x2 = xc + xc;
    yp = clenshaw6(xc, x2, 98.7821300963361, 164.273222212631, 92.948112871376,
                   33.8165311212688, 6.91626704983562, 0.556380877028234);
    yq = clenshaw6(xc, x2, 1, 1.64339473427892, 0.89392405219969, 0.298947051776379,
                   0.0507146002577288, 0.00226495093949756);
  } else {
    yp = clenshaw6_plain(xc, 98.7821300963361, 164.273222212631, 92.948112871376,
                         33.8165311212688, 6.91626704983562, 0.556380877028234);
    yq = clenshaw6_plain(xc, 1, 1.64339473427892, 0.89392405219969, 0.298947051776379,
                         0.0507146002577288, 0.00226495093949756);
  }
  if (yq == 0.0) return 0.0f;
  return (float)(yp / yq);
}

// One pixel of OpsinDynamicsImage (butteraugli.cc:337-363): blurred rgb -> sensitivity,
// sharp rgb -> mixed * sensitivity -> XYB.
GZ_DEVFN void opsin_pixel(float br, float bg, float bb, float r, float g, float b,
                          float* x, float* y, float* z) {
  float p0, p1, p2, c0, c1, c2;
  opsin_absorbance(br, bg, bb, &p0, &p1, &p2);
  // (float)(Gamma((double)p) / (double)p), butteraugli.cc:351-353
  const float s0 = gamma_poly_f((double)p0) / p0;
  const float s1 = gamma_poly_f((double)p1) / p1;
  const float s2 = gamma_poly_f((double)p2) / p2;
  opsin_absorbance(r, g, b, &c0, &c1, &c2);
  c0 *= s0;
  c1 *= s1;
  c2 *= s2;
  *x = c0 - c1;
  *y = c0 + c1;
  *z = c2;
}

// ---- SeparateFrequencies helpers, butteraugli.cc:369-487 -------------------------
GZ_DEVFN float remove_range(float w, float x) {
  return x > w ? x - w : x < -w ? x + w : 0.0f;
}
GZ_DEVFN float amplify_range(float w, float x) {
  return x > w ? x + w : x < -w ? x - w : 2.0f * x;
}
GZ_DEVFN float maximum_clamp(float v, float maxval) {
  const double kMul = 0.688059627878;
  if (v >= maxval) {
    v -= maxval;
    v = (float)((double)v * kMul);
    v += maxval;
  } else if (v < -maxval) {
    v += maxval;
    v = (float)((double)v * kMul);
    v -= maxval;
  }
  return v;
}
GZ_DEVFN float suppress_bright(float hf, float brightness, float mul, float reg) {
  const float scaler = (mul * reg) / (reg + brightness);
  return scaler * hf;
}
GZ_DEVFN float suppress_x_by_y(float xv, float yv) {
  const double suppress = 2.96534974403, s = 0.745954517135;
  const double xval = xv, yval = yv;
  const double scaler = s + (suppress * (1.0 - s)) / (suppress + yval * yval);
  return (float)(scaler * xval);
}
// XybLowFreqToVals<float>, butteraugli.cc:382-399
GZ_DEVFN void lf_to_vals(float x, float y, float b_arg, float* vx, float* vy, float* vb) {
  const float xmul = (float)5.57547552483, ymul = (float)1.20828034498,
              bmul = (float)6.08319517575, y_to_b_mul = (float)-0.628811683685;
  const float b = b_arg + y_to_b_mul * y;
  *vb = b * bmul;
  *vx = x * xmul;
  *vy = y * ymul;
}

// ---- Malta per-pixel "diffs" value, butteraugli.cc:1473-1529 ---------------------
struct MaltaNorm {
  float norm2_0gt1, norm2_0lt1, norm1f;
  int fast_div;   // 1: norm2_* are in [2^-40, 2^40] (malta_diff may share one reciprocal)
};
// The reference's statement sequence, kept as the definition the fast form below is checked
// against (tests/cpp/verify_malta_diff.cc on the host; tools/ubench/divcheck.hip for the
// division on the device).
GZ_DEVFN float malta_diff_plain(float a, float b, const MaltaNorm nm) {
  const float absval = (float)(0.5 * (double)fabsf(a) + 0.5 * (double)fabsf(b));
  const float diff = a - b;
  const float scaler = nm.norm2_0gt1 / (nm.norm1f + absval);
  float d = scaler * diff;
  const float scaler2 = nm.norm2_0lt1 / (nm.norm1f + absval);
  const double fabs0 = (double)fabsf(a);
  const double too_small = 0.55 * fabs0;
  const double too_big = 1.05 * fabs0;
  const double bd = (double)b;
  double impact;
  bool hit = true;
  if (a < 0) {
    if (bd > -too_small) impact = (double)scaler2 * (bd + too_small);
    else if (bd < -too_big) impact = (double)scaler2 * (-bd - too_big);
    else { hit = false; impact = 0.0; }
  } else {
    if (bd < too_small) impact = (double)scaler2 * (too_small - bd);
    else if (bd > too_big) impact = (double)scaler2 * (bd - too_big);
    else { hit = false; impact = 0.0; }
  }
  if (hit) d = diff < 0 ? (float)((double)d - impact) : (float)((double)d + impact);
  return d;
}

#ifdef GZ_EMU
inline int& gz_emu_rcp_ulps() {
  static int v = 0;
  return v;
}
#endif
// n0 / d and n1 / d, both correctly rounded, from ONE reciprocal estimate: the refinement
// steps of the compiler's own expansion of an IEEE f32 division (reciprocal, one Newton step
// on it, quotient, two residual corrections -- all fused multiply-adds), without the operand
// scaling and the special-case fix-up that surround it there.  Those two are the identity when
// neither operand nor the quotient comes near the ends of the exponent range, which the caller
// guarantees (2^-40 <= d, n <= 2^40).
GZ_DEVFN void div2_shared(float n0, float n1, float d, float* q0, float* q1) {
#ifdef GZ_EMU
  // (v_rcp_f32 is accurate to 1 ulp: the host check runs the refinement from RN(1/d) and from
  // its two neighbours, gz_emu_rcp_ulps() = -1, 0, +1)
  float r0 = 1.0f / d;
  if (gz_emu_rcp_ulps() > 0) r0 = __builtin_nextafterf(r0, __builtin_inff());
  if (gz_emu_rcp_ulps() < 0) r0 = __builtin_nextafterf(r0, -__builtin_inff());
#else
  const float r0 = __builtin_amdgcn_rcpf(d);
#endif
  const float e0 = __builtin_fmaf(-d, r0, 1.0f);
  const float r = __builtin_fmaf(e0, r0, r0);
  {
    const float q = n0 * r;
    const float e1 = __builtin_fmaf(-d, q, n0);
    const float qq = __builtin_fmaf(e1, r, q);
    const float e2 = __builtin_fmaf(-d, qq, n0);
    *q0 = __builtin_fmaf(e2, r, qq);
  }
  {
    const float q = n1 * r;
    const float e1 = __builtin_fmaf(-d, q, n1);
    const float qq = __builtin_fmaf(e1, r, q);
    const float e2 = __builtin_fmaf(-d, qq, n1);
    *q1 = __builtin_fmaf(e2, r, qq);
  }
}

// Same value as malta_diff_plain, bit for bit, with a third of the instructions and no
// divergent branches (a wave of malta_diff_plain costs ~375 cycles, most of it in the four
// exec-masked regions of the if-ladder):
//  * absval: for floats x, y >= 0, float(0.5*double(x) + 0.5*double(y)) == 0.5f * (x + y)
//    whenever the float sum s = x + y is a normal number away from the range ends: the halves
//    and -- for exponents at most 28 apart -- their sum are exact in double, so both sides
//    round the exact (x + y) / 2 once to 24 bits; further apart, the smaller operand is below
//    2^-5 ulp of the larger in both.  Sums outside [2^-100, 2^100] take the plain form
//    (0 + 0 included: both give 0).
//  * the two quotients share their reciprocal (div2_shared) when the denominator is in
//    [2^-40, 2^40] (the numerators are per-pass constants, checked on the host:
//    MaltaNorm::fast_div).
//  * the ladder: with bb = (a < 0 ? -b : b) its four cases are  bb < too_small -> scaler2 *
//    (too_small - bb),  else bb > too_big -> scaler2 * (bb - too_big)  -- negation is exact,
//    x - (-y) == x + y == y + x -- and d -/+ impact is d + (-/+ impact); round 5: the two cases as
//    one maximum (below), the FP64 form of absval behind a wavefront-uniform branch.
GZ_DEVFN float malta_diff(float a, float b, const MaltaNorm nm) {
#ifdef GZ_MALTA_DIFF_PLAIN
  return malta_diff_plain(a, b, nm);
#endif
  const float fa = fabsf(a), fb = fabsf(b);
  const float s = fa + fb;
  float absval = 0.5f * s;
  const bool odd = !(s >= 0x1p-100f && s <= 0x1p100f);
  if (GZ_ANY_LANE(odd)) {   // (0 + 0 included: flat regions take this path, whole wavefronts of them)
    GZ_RARE_PATH();
    if (odd) absval = (float)(0.5 * (double)fa + 0.5 * (double)fb);
  }
  const float diff = a - b;
  const float den = nm.norm1f + absval;
  float scaler, scaler2;
  // (fast_div also says norm1f >= 2^-40, and absval >= 0: the sum is not below 2^-40; a NaN fails the test)
  if (nm.fast_div && den <= 0x1p40f) {
    div2_shared(nm.norm2_0gt1, nm.norm2_0lt1, den, &scaler, &scaler2);
  } else {
    scaler = nm.norm2_0gt1 / den;
    scaler2 = nm.norm2_0lt1 / den;
  }
  const float d = scaler * diff;
  const double fabs0 = (double)fa;
  const double too_small = 0.55 * fabs0;
  const double too_big = 1.05 * fabs0;
  const double bb = (double)(a < 0 ? -b : b);
  // too_small <= too_big (RN is monotone), so of  d1 = too_small - bb  and  d2 = bb - too_big  at most
  // one is positive: bb < too_small  <=>  d1 > 0 (then d2 < 0), else bb > too_big  <=>  d2 > 0 (then
  // d1 <= 0) -- the ladder's (u - v) is the larger of the two, and it hits iff that is positive.
  // (NaNs: inf - inf on both sides or a NaN sample make both NaN; "> 0" is false, as the ladder's tests.)
  const double d1 = too_small - bb, d2 = bb - too_big;
  const double mx = __builtin_fmax(d1, d2);
  const bool hit = mx > 0.0;
  double impact = (double)scaler2 * mx;
  if (diff < 0) impact = -impact;
  const float r = (float)((double)d + impact);
  return hit ? r : d;
}

// ---- L2Diff / L2DiffAsymmetric / SameNoiseLevels accumulations -------------------
// butteraugli.cc:654-668
GZ_DEVFN float l2diff_acc(float acc, float a, float b, double w) {
  const double diff = (double)(a - b);
  return (float)((double)acc + (w * diff) * diff);
}
// butteraugli.cc:672-714; w_0gt1 / w_0lt1 are already multiplied by 0.8.
GZ_DEVFN float l2diff_asym_acc(float acc, float a, float b, double w_0gt1, double w_0lt1) {
  const double diff = (double)(a - b);
  acc = (float)((double)acc + (w_0gt1 * diff) * diff);
  const double fabs0 = (double)fabsf(a);
  const double too_small = 0.4 * fabs0;
  const double too_big = 1.0 * fabs0;
  const double bd = (double)b;
  if (a < 0) {
    if (bd > -too_small) {
      const double v = bd + too_small;
      acc = (float)((double)acc + (w_0lt1 * v) * v);
    } else if (bd < -too_big) {
      const double v = -bd - too_big;
      acc = (float)((double)acc + (w_0lt1 * v) * v);
    }
  } else {
    if (bd < too_small) {
      const double v = too_small - bd;
      acc = (float)((double)acc + (w_0lt1 * v) * v);
    } else if (bd > too_big) {
      const double v = bd - too_big;
      acc = (float)((double)acc + (w_0lt1 * v) * v);
    }
  }
  return acc;
}
// butteraugli.cc:631-641 (input of the SameNoiseLevels blur)
GZ_DEVFN float same_noise_pre(float a, float b) {
  const double maxclamp = 85.7047444518;
  double v0 = (double)fabsf(a);
  double v1 = (double)fabsf(b);
  if (v0 > maxclamp) v0 = maxclamp;
  if (v1 > maxclamp) v1 = maxclamp;
  return (float)(v0 - v1);
}

// ---- Mask: DiffPrecompute core (butteraugli.cc:1723-1733) and LUT interpolation ---
// One image's half: |c - right| + |c - down| in float (fabs(float) is the float overload,
// butteraugli.cc:1726-1729), the other's likewise; then min, scale, clamp.
GZ_DEVFN float diff_sup(float c, float r, float d) { return fabsf(c - r) + fabsf(c - d); }
GZ_DEVFN float diff_from_sups(float sup0f, float sup1f) {
  const double sup0 = (double)sup0f, sup1 = (double)sup1f;
  const double mul0 = 0.918416534734, cutoff = 55.0184555849;
  float v = (float)(mul0 * (sup0 < sup1 ? sup0 : sup1));   // std::min(sup0, sup1)
  if ((double)v >= cutoff) v = (float)cutoff;
  return v;
}
// InterpolateClampNegative, butteraugli.cc:236-251 (size 512)
GZ_DEVFN double interp_lut512(const double* a, double ix) {
  if (ix < 0) ix = 0;
  const int base = (int)ix;
  if (base >= 511) return a[511];
  const double mix = ix - (double)base;
  return a[base] + mix * (a[base + 1] - a[base]);
}

}  // namespace gz
