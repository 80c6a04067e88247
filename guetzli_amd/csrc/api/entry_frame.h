// C ABI: context-free block transforms (gz_encode_rgb_only, double-precision DCT and its two users), the frame layout switch, OutputImage::Downsample (4:2:0).
// (part of the one translation unit gz_api.hip, which includes these files in order; split by
// concern in round 5 -- no declaration here is visible outside libguetzli_amd.so but the C ABI)
#pragma once

extern "C" {

static int probe_device(int device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev ||
      hipSetDevice(device) != hipSuccess)
    return GZ_E_NO_DEVICE;
  return GZ_OK;
}


// ----------------------------------------------------------- double-precision DCT ----
namespace {
struct DevBuf {   // scoped device allocation for the context-free entry points
  void* p = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) (void)hipFree(p); }
  bool alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1) == hipSuccess; }
};
}  // namespace

int gz_encode_rgb_only(int device, const uint8_t* rgb, int w, int h, int16_t* coeffs_out) {
  if (!rgb || !coeffs_out || w <= 0 || h <= 0 || w >= (1 << 16) || h >= (1 << 16)) return GZ_E_ARG;
  if (probe_device(device) != GZ_OK) return GZ_E_NO_DEVICE;
  const int bw = (w + 7) / 8, bh = (h + 7) / 8, nb = bw * bh;
  DevBuf drgb, dco;
  if (!drgb.alloc((size_t)3 * w * h) || !dco.alloc((size_t)3 * nb * 128)) return GZ_E_NOMEM;
  if (hipMemcpy(drgb.p, rgb, (size_t)3 * w * h, hipMemcpyHostToDevice) != hipSuccess) return GZ_E_HIP;
  const uint8_t* d_rgb = (const uint8_t*)drgb.p;
  int16_t* d_co = (int16_t*)dco.p;
  GZ_LAUNCH(k_encode_rgb, dim3(gz_div_up(nb, kBlocksPerWG)), dim3(256), (hipStream_t)0, d_rgb, w, h,
            bw, nb, d_co);
  if (hipGetLastError() != hipSuccess) return GZ_E_HIP;
  if (hipMemcpy(coeffs_out, dco.p, (size_t)3 * nb * 128, hipMemcpyDeviceToHost) != hipSuccess)
    return GZ_E_HIP;
  return GZ_OK;
}

int gz_dct_double_blocks(int device, double* blocks, int n, int inverse) {
  if (!blocks || n <= 0) return GZ_E_ARG;
  if (probe_device(device) != GZ_OK) return GZ_E_NO_DEVICE;
  DevBuf d;
  const size_t bytes = (size_t)n * 64 * sizeof(double);
  if (!d.alloc(bytes)) return GZ_E_NOMEM;
  if (hipMemcpy(d.p, blocks, bytes, hipMemcpyHostToDevice) != hipSuccess) return GZ_E_HIP;
  double* dblk = (double*)d.p;
  if (inverse) {
    GZ_LAUNCH((k_dctd_blocks<true>), dim3(gz_div_up(n, kBlocksPerWG)), dim3(256), (hipStream_t)0,
              dblk, n);
  } else {
    GZ_LAUNCH((k_dctd_blocks<false>), dim3(gz_div_up(n, kBlocksPerWG)), dim3(256), (hipStream_t)0,
              dblk, n);
  }
  if (hipGetLastError() != hipSuccess) return GZ_E_HIP;
  if (hipMemcpy(blocks, d.p, bytes, hipMemcpyDeviceToHost) != hipSuccess) return GZ_E_HIP;
  return GZ_OK;
}

int gz_component_to_float_pixels(int device, const int16_t* coeffs, int w, int h, float* out) {
  if (!coeffs || !out || w <= 0 || h <= 0 || w >= (1 << 16) || h >= (1 << 16)) return GZ_E_ARG;
  if (probe_device(device) != GZ_OK) return GZ_E_NO_DEVICE;
  const int bw = (w + 7) / 8, bh = (h + 7) / 8, nb = bw * bh;
  DevBuf dc, dp;
  if (!dc.alloc((size_t)nb * 128) || !dp.alloc((size_t)w * h * sizeof(float))) return GZ_E_NOMEM;
  if (hipMemcpy(dc.p, coeffs, (size_t)nb * 128, hipMemcpyHostToDevice) != hipSuccess) return GZ_E_HIP;
  const int16_t* dcoef = (const int16_t*)dc.p;
  float* dpix = (float*)dp.p;
  GZ_LAUNCH(k_to_float_pixels, dim3(gz_div_up(nb, kBlocksPerWG)), dim3(256), (hipStream_t)0,
            dcoef, w, h, bw, nb, dpix);
  if (hipGetLastError() != hipSuccess) return GZ_E_HIP;
  if (hipMemcpy(out, dp.p, (size_t)w * h * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
    return GZ_E_HIP;
  return GZ_OK;
}

int gz_component_set_downsampled(int device, const float* pixels, int w, int h, int fx, int fy,
                                 int16_t* coeffs_out) {
  if (!pixels || !coeffs_out || w <= 0 || h <= 0 || w >= (1 << 16) || h >= (1 << 16) ||
      fx < 1 || fy < 1 || fx > 4 || fy > 4)
    return GZ_E_ARG;
  if (probe_device(device) != GZ_OK) return GZ_E_NO_DEVICE;
  const int bw = (w + 8 * fx - 1) / (8 * fx), bh = (h + 8 * fy - 1) / (8 * fy), nb = bw * bh;
  DevBuf dc, dp;
  if (!dc.alloc((size_t)nb * 128) || !dp.alloc((size_t)w * h * sizeof(float))) return GZ_E_NOMEM;
  if (hipMemcpy(dp.p, pixels, (size_t)w * h * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
    return GZ_E_HIP;
  const float* dpix = (const float*)dp.p;
  int16_t* dcoef = (int16_t*)dc.p;
  GZ_LAUNCH(k_set_downsampled_coeffs, dim3(gz_div_up(nb, kBlocksPerWG)), dim3(256), (hipStream_t)0,
            dpix, w, h, fx, fy, bw, nb, dcoef);
  if (hipGetLastError() != hipSuccess) return GZ_E_HIP;
  if (hipMemcpy(coeffs_out, dc.p, (size_t)nb * 128, hipMemcpyDeviceToHost) != hipSuccess)
    return GZ_E_HIP;
  return GZ_OK;
}


int gz_set_frame(gz_ctx* c, int chroma_factor) {
  DeviceScope ds_(c);
  if (!c || (chroma_factor != 1 && chroma_factor != 2)) return GZ_E_ARG;
  if (c->cfac == chroma_factor) return GZ_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  set_frame(c, chroma_factor);
  c->have_cand = false;
  c->lin_is_cand = c->xyb_is_cand = false;
  c->have_orig = false;   // the original coefficients on the device belonged to the other frame
  return GZ_OK;
}


// OutputImage::Downsample (output_image.cc:304-340), cfg defaults of Processor::DownsampleImage
// (processor.cc:97-104) without the silver-screen option, on the ORIGINAL coefficients of a
// 4:4:4 frame: ToFloatPixels of the three components, PreProcessChannel on V then on U
// (preprocess_downsample.cc:157-279), SetDownsampledCoefficients of U and V by 2 x 2.
static void normal_taps(double sigma, float k[5], float* mul) {   // Normal(), :85-88; Sharpen / Blur :92-100,138-146
  double kernel[5], sum = 0;
  for (size_t i = 0; i < 5; ++i) {
    const double x = 1.0 * i - 5 / 2;
    static const double kInvSqrt2Pi = 0.3989422804014327;
    kernel[i] = exp(-x * x / (2 * sigma * sigma)) * kInvSqrt2Pi / sigma;
  }
  for (size_t i = 0; i < 5; ++i) sum += kernel[i];
  for (size_t i = 0; i < 5; ++i) k[i] = static_cast<float>(kernel[i]);
  *mul = static_cast<float>(1.0 / sum);
}

int gz_downsample(gz_ctx* c, int16_t* coeffs_out) {
  DeviceScope ds_(c);
  if (!c) return GZ_E_ARG;
  if (!c->have_orig || c->cfac != 1) { c->err = "gz_downsample needs the original coefficients of a 4:4:4 frame"; return GZ_E_STATE; }
  const int w = c->w, h = c->h;
  const size_t n = (size_t)w * h;
  // scratch: planes of the candidate's evaluation (nothing of it is in flight here)
  c->xyb_is_cand = false;
  float* yuv[3] = {c->xyb[0], c->xyb[1], c->xyb[2]};
  float* tmp_s = c->tmp[0];
  float* tmp_b = c->tmp[1];
  // byte planes: four per float plane
  uint8_t* bp0 = reinterpret_cast<uint8_t*>(c->tmp[2]);
  uint8_t* bp1 = reinterpret_cast<uint8_t*>(c->lf_raw[0]);
  uint8_t* dark_a = bp0; uint8_t* dark_b = bp0 + n; uint8_t* red_a = bp0 + 2 * n; uint8_t* red_b = bp0 + 3 * n;
  uint8_t* sharpen = bp1; uint8_t* blurm = bp1 + n; uint8_t* blur_t = bp1 + 2 * n;
  for (int i = 0; i < 3; ++i) {
    GZ_LAUNCH(k_to_float_pixels, dim3(gz_div_up(c->nb, kBlocksPerWG)), dim3(256), c->stream,
              (const int16_t*)(c->d_orig + (size_t)c->coff[i] * 64), w, h, c->bw, c->nb, yuv[i]);
    KCHK(c);
  }
  PPTaps taps;
  normal_taps((double)1.3f, taps.ks, &taps.mul_s);   // Sharpen(sigma = 1.3f)
  normal_taps(1.3, taps.kb, &taps.mul_b);            // Blur: kSigma = 1.3
  const dim3 g1(gz_div_up((int)std::min<size_t>(n, 0x7fffffff), 256)), g2(gz_div_up(w, 256), h);
  const int channels[2] = {2, 1};   // :326-329
  for (int pass = 0; pass < 2; ++pass) {
    const int channel = channels[pass];
    GZ_LAUNCH(k_pp_normalize, g1, dim3(256), c->stream, yuv[0], yuv[1], yuv[2], n);
    KCHK(c);
    GZ_LAUNCH(k_pp_maps, g1, dim3(256), c->stream, (const float*)yuv[0], (const float*)yuv[1],
              (const float*)yuv[2], n, channel, dark_a, red_a);
    KCHK(c);
    // Erode x3 (darkmap, :194-196): a -> b -> a -> b; Dilate x3 (redmap, :217-219) likewise
    for (int i = 0; i < 3; ++i) {
      GZ_LAUNCH(k_pp_morph, g2, dim3(256), c->stream, (const uint8_t*)(i & 1 ? dark_b : dark_a),
                i & 1 ? dark_a : dark_b, w, h, 1);
      KCHK(c);
      GZ_LAUNCH(k_pp_morph, g2, dim3(256), c->stream, (const uint8_t*)(i & 1 ? red_b : red_a),
                i & 1 ? red_a : red_b, w, h, 0);
      KCHK(c);
    }
    const double threshold = (channel == 2 ? 0.02 : 1.0) * 127.5;
    GZ_LAUNCH(k_pp_edge_maps, g2, dim3(256), c->stream, (const float*)yuv[channel], (const float*)yuv[1],
              (const float*)yuv[2], (const uint8_t*)dark_b, (const uint8_t*)red_b, w, h,
              threshold, sharpen, blurm);
    KCHK(c);
    // Erode x2 (blurmap, :254-255)
    GZ_LAUNCH(k_pp_morph, g2, dim3(256), c->stream, (const uint8_t*)blurm, blur_t, w, h, 1);
    KCHK(c);
    GZ_LAUNCH(k_pp_morph, g2, dim3(256), c->stream, (const uint8_t*)blur_t, blurm, w, h, 1);
    KCHK(c);
    GZ_LAUNCH(k_pp_conv_h, g2, dim3(256), c->stream, (const float*)yuv[channel], w, h, taps, tmp_s, tmp_b);
    KCHK(c);
    GZ_LAUNCH(k_pp_conv_v_select, g2, dim3(256), c->stream, yuv[channel], (const float*)tmp_s,
              (const float*)tmp_b, (const uint8_t*)sharpen, (const uint8_t*)blurm, w, h, taps, 0.5f, 1, 1);
    KCHK(c);
    GZ_LAUNCH(k_pp_denormalize, g1, dim3(256), c->stream, yuv[0], yuv[1], yuv[2], n);
    KCHK(c);
  }
  // the two chroma components, 2 x 2 subsampled, replace the 4:4:4 ones (luma is kept as it is)
  const int cbw = (w + 15) / 16, cbh = (h + 15) / 16, nbc = cbw * cbh;
  for (int i = 1; i < 3; ++i) {
    int16_t* dst = c->d_orig + ((size_t)c->nb + (size_t)(i - 1) * nbc) * 64;
    GZ_LAUNCH(k_set_downsampled_coeffs, dim3(gz_div_up(nbc, kBlocksPerWG)), dim3(256), c->stream,
              (const float*)yuv[i], w, h, 2, 2, cbw, nbc, dst);
    KCHK(c);
  }
  set_frame(c, 2);
  c->have_cand = false;
  c->lin_is_cand = c->xyb_is_cand = false;
  c->have_distmap = false;
  if (coeffs_out)
    HIPCHK(c, hipMemcpyAsync(coeffs_out, c->d_orig, (size_t)c->nblk * 128, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GZ_OK;
}

int gz_downsample_planes(gz_ctx* c, const float* y, const float* u, const float* v, int16_t* coeffs_out) {
  DeviceScope ds_(c);
  if (!c || !y || !u || !v) return GZ_E_ARG;
  if (!c->have_orig || c->cfac != 1) { c->err = "gz_downsample_planes needs the original coefficients of a 4:4:4 frame"; return GZ_E_STATE; }
  const int w = c->w, h = c->h;
  const size_t n = (size_t)w * h;
  const float* src[3] = {y, u, v};
  c->xyb_is_cand = false;   // (xyb[] as scratch)
  for (int i = 0; i < 3; ++i)
    HIPCHK(c, hipMemcpyAsync(c->xyb[i], src[i], n * sizeof(float), hipMemcpyHostToDevice, c->stream));
  // output_image.cc:314-316: every component from its plane, luma included (factor 1), the
  // chroma blocks packed behind the nb luma blocks
  const int cbw = (w + 15) / 16, cbh = (h + 15) / 16, nbc = cbw * cbh;
  GZ_LAUNCH(k_set_downsampled_coeffs, dim3(gz_div_up(c->nb, kBlocksPerWG)), dim3(256), c->stream,
            (const float*)c->xyb[0], w, h, 1, 1, c->bw, c->nb, c->d_orig);
  KCHK(c);
  for (int i = 1; i < 3; ++i) {
    int16_t* dst = c->d_orig + ((size_t)c->nb + (size_t)(i - 1) * nbc) * 64;
    GZ_LAUNCH(k_set_downsampled_coeffs, dim3(gz_div_up(nbc, kBlocksPerWG)), dim3(256), c->stream,
              (const float*)c->xyb[i], w, h, 2, 2, cbw, nbc, dst);
    KCHK(c);
  }
  set_frame(c, 2);
  c->have_cand = false;
  c->lin_is_cand = c->xyb_is_cand = false;
  c->have_distmap = false;
  if (coeffs_out)
    HIPCHK(c, hipMemcpyAsync(coeffs_out, c->d_orig, (size_t)c->nblk * 128, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GZ_OK;
}


}  // extern "C"
