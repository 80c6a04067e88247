// Blur plans (taps and border scales, host-built with the reference's float exp), Malta normalisations, the PsychoImage plane set.
// (part of the one translation unit gz_api.hip, which includes these files in order; split by
// concern in round 5 -- no declaration here is visible outside libguetzli_amd.so but the C ABI)
#pragma once

namespace {

// ------------------------------------------------------------------ blur plans ------
// Host-side restatement of ComputeKernel (butteraugli.cc:145-154) and of the border
// normalisation of ConvolveBorderColumn (:156-181).  exp() is evaluated on the host with
// the same libm float overload the reference uses; the device never recomputes taps.
struct BlurCfg {
  float sigma, border_ratio;
  int r;
  std::vector<float> k, ks;
  float wsum;
  // device border scales for the x axis (length w) and the y axis (length h)
  float* d_scale = nullptr;   // 4*r floats: x.lo, x.hi, y.lo, y.hi
  BorderScale bx, by;
};

void make_taps_host(float sigma, BlurCfg* c) {
  const float m = 2.25;
  const float scaler = -1.0 / (2 * sigma * sigma);
  const int diff = std::max<int>(1, m * fabsf(sigma));
  c->sigma = sigma;
  c->r = diff;
  c->k.resize(2 * diff + 1);
  for (int i = -diff; i <= diff; ++i) c->k[i + diff] = expf(scaler * i * i);
  float w = 0.0f;
  for (size_t j = 0; j < c->k.size(); ++j) w += c->k[j];
  c->wsum = w;
  const float s = 1.0f / w;
  c->ks = c->k;
  for (size_t j = 0; j < c->ks.size(); ++j) c->ks[j] *= s;
}

// scale(x) for a border position x on an axis of length n.
float border_scale(const BlurCfg& c, int n, int x) {
  const int r = c.r;
  const int lo = x < r ? 0 : x - r;
  const int hi = std::min(n - 1, x + r);
  float weight = 0.0f;
  for (int j = lo; j <= hi; ++j) weight += c.k[j - x + r];
  weight = (1.0f - c.border_ratio) * weight + c.border_ratio * c.wsum;
  return 1.0f / weight;
}

void border_scales_host(const BlurCfg& c, int n, std::vector<float>* lo,
                        std::vector<float>* hi) {
  lo->assign(c.r, 1.0f);
  hi->assign(c.r, 1.0f);
  for (int i = 0; i < c.r; ++i) {
    if (i < n) (*lo)[i] = border_scale(c, n, i);
    if (n - 1 - i >= 0) (*hi)[i] = border_scale(c, n, n - 1 - i);
  }
}

template <int R>
Taps<R> taps_of(const BlurCfg& c) {
  Taps<R> t;
  for (int j = 0; j <= 2 * R; ++j) {
    t.k[j] = c.k[j];
    t.ks[j] = c.ks[j];
  }
  return t;
}

enum BlurId { B_OPSIN, B_LF, B_MF, B_HF, B_SN, B_MASKX, B_MASKY0, B_MASKY1, B_FINAL, B_COUNT };
struct BlurSpec { double sigma, border; int r; };
// sigmas / border ratios: butteraugli.cc:329, :497-508, :885, :1757-1760, :737-740
const BlurSpec kBlurSpecs[B_COUNT] = {
  {1.2, 0.0, 2},
  {7.46953768697, -0.00457628248637, 16},
  {3.734768843485, -0.271277366628, 8},
  {1.8673844217425, 0.147068973249, 4},
  {10.6666499623, 0.0, 23},
  {9.24456601467, -0.0724948220913, 20},
  {2.3770330432, -0.0724948220913, 5},
  {9.04353323561, -0.0724948220913, 20},
  {1.72547472444, 1.0, 3},
};

// MakeMask (butteraugli.cc:1638-1653) for MaskX / MaskY / MaskDcX / MaskDcY (:1655-1697)
void make_mask_lut(double extmul, double extoff, double mul, double offset, double scaler,
                   double* lut) {
  const double kGlobalScale = 1.0 / 20.35;
  for (int i = 0; i < 512; ++i) {
    const double c = mul / ((0.01 * scaler * i) + offset);
    lut[i] = kGlobalScale * (1.0 + extmul * (c + extoff));
    if (lut[i] < 1e-5) lut[i] = 1e-5;
    lut[i] *= lut[i];
  }
}

// Malta normalisation constants (MaltaDiffMapImpl, butteraugli.cc:1468-1476)
MaltaNorm malta_norm(bool lf, double w_0gt1, double w_0lt1, double norm1) {
  const double len = 3.75;
  const double mulli = lf ? 0.405371989604 : 0.354191303559;
  const float kWeight0 = 0.5;
  const float kWeight1 = 0.33;
  const double w_pre0gt1 = mulli * sqrt(kWeight0 * w_0gt1) / (len * 2 + 1);
  const double w_pre0lt1 = mulli * sqrt(kWeight1 * w_0lt1) / (len * 2 + 1);
  MaltaNorm n;
  n.norm2_0gt1 = w_pre0gt1 * norm1;
  n.norm2_0lt1 = w_pre0lt1 * norm1;
  n.norm1f = static_cast<float>(norm1);
  auto mid = [](float x) { return x >= 0x1p-40f && x <= 0x1p40f; };
  // (norm1f: malta_diff then needs one comparison for its denominator, norm1f + absval >= norm1f)
  n.fast_div = mid(n.norm2_0gt1) && mid(n.norm2_0lt1) && n.norm1f >= 0x1p-40f && n.norm1f <= 0x1p39f ? 1 : 0;
  return n;
}

// The six Malta passes of DiffmapPsychoImage (butteraugli.cc:835-874): [channel X/Y][band
// UHF, HF, MF] -> normalisation and tap pattern.
struct MaltaSpec {
  MaltaNorm nm;
  int lf;
};
void malta_specs(MaltaSpec out[2][3]) {
  const float hf_asymmetry_ = 0.8f;
  const double wUhfMalta = 5.1409625726, norm1Uhf = 58.5001247061;
  const double wUhfMaltaX = 4.91743441556, norm1UhfX = 687196.39002;
  const double wHfMalta = 153.671655716, norm1Hf = 83150785.9592;
  const double wHfMaltaX = 668.358918152, norm1HfX = 0.882954368025;
  const double wMfMalta = 6841.81248144, norm1Mf = 0.0135134962487;
  const double wMfMaltaX = 813.901703816, norm1MfX = 16792.9322251;
  const float sqrt_asym = sqrtf(hf_asymmetry_);   // float sqrt overload in the reference
  out[1][0] = {malta_norm(false, wUhfMalta * hf_asymmetry_, wUhfMalta / hf_asymmetry_, norm1Uhf), 0};
  out[1][1] = {malta_norm(true, wHfMalta * sqrt_asym, wHfMalta / sqrt_asym, norm1Hf), 1};
  out[1][2] = {malta_norm(true, wMfMalta, wMfMalta, norm1Mf), 1};
  out[0][0] = {malta_norm(false, wUhfMaltaX * hf_asymmetry_, wUhfMaltaX / hf_asymmetry_, norm1UhfX), 0};
  out[0][1] = {malta_norm(true, wHfMaltaX * sqrt_asym, wHfMaltaX / sqrt_asym, norm1HfX), 1};
  out[0][2] = {malta_norm(true, wMfMaltaX, wMfMaltaX, norm1MfX), 1};
}

struct Psycho {   // device planes of one image's PsychoImage (butteraugli.h:418-423)
  float* lfv[3];  // lf in "vals" space
  float* mf[2];   // X, Y  (mf[2] of the reference is dead: wmul[5] == 0)
  float* hf[2];
  float* uhf[2];
};

}  // namespace
