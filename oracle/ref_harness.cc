// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or executed from the
// product path (guetzli_amd/).  Only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may load the library built from this file.
//
// This translation unit is a thin extern "C" shim around the UNMODIFIED reference
// sources, compiled from where they lie under /root/reference by oracle/Makefile
// into oracle/_ref/libgz_ref.so.  No reference source is copied: the two .cc files
// whose internal (static / anonymous-namespace) functions we need to probe are
// #include-d so that their internals are visible to this TU.
//
// What it exposes: every stage of the hot path as a plain-pointer C
// function, so that the restatement in oracle/gz_oracle.cc and the HIP kernels can be
// compared stage by stage with the real thing.

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

// Make the anonymous-namespace Processor class and the private members of the
// butteraugli comparator reachable from the probes below.
#define private public
#define protected public
#include "butteraugli/butteraugli.cc"   // butteraugli internals (SeparateFrequencies, ...)
#include "guetzli/processor.cc"         // guetzli::Processor internals (phase A)
#undef private
#undef protected

#include "guetzli/butteraugli_comparator.h"
#include "guetzli/color_transform.h"
#include "guetzli/dct_double.h"
#include "guetzli/fdct.h"
#include "guetzli/gamma_correct.h"
#include "guetzli/idct.h"
#include "guetzli/jpeg_data_encoder.h"
#include "guetzli/jpeg_data_reader.h"
#include "guetzli/jpeg_data_writer.h"
#include "guetzli/output_image.h"
#include "guetzli/quality.h"
#include "guetzli/quantize.h"
#include "guetzli/score.h"

using butteraugli::ImageF;

namespace {

std::vector<ImageF> PlanesFromFlat(const float* src, int w, int h, int n) {
  std::vector<ImageF> planes = butteraugli::CreatePlanes<float>(w, h, n);
  for (int c = 0; c < n; ++c)
    for (int y = 0; y < h; ++y)
      memcpy(planes[c].Row(y), src + ((size_t)c * h + y) * w, sizeof(float) * w);
  return planes;
}

void FlatFromPlane(const ImageF& p, float* dst) {
  for (size_t y = 0; y < p.ysize(); ++y)
    memcpy(dst + y * p.xsize(), p.Row(y), sizeof(float) * p.xsize());
}

void FlatFromPlanes(const std::vector<ImageF>& planes, float* dst) {
  for (size_t c = 0; c < planes.size(); ++c)
    FlatFromPlane(planes[c],
                  dst + c * planes[c].xsize() * planes[c].ysize());
}

// coeffs layout used across this repo: component-major, block-major, 64 per block:
//   coeffs[(c * nblocks + block_y * bw + block_x) * 64 + k]
void FillOutputImage(guetzli::OutputImage* img, const int16_t* coeffs) {
  for (int c = 0; c < 3; ++c) {
    guetzli::OutputImageComponent& comp = img->component(c);
    const int bw = comp.width_in_blocks(), bh = comp.height_in_blocks();
    const int16_t* src = coeffs + (size_t)c * bw * bh * 64;
    for (int by = 0; by < bh; ++by)
      for (int bx = 0; bx < bw; ++bx)
        comp.SetCoeffBlock(bx, by, src + ((size_t)by * bw + bx) * 64);
  }
}

void JpegDataFromCoeffs(const int16_t* coeffs, int w, int h,
                        guetzli::JPEGData* jpg) {
  guetzli::InitJPEGDataForYUV444(w, h, jpg);
  guetzli::AddApp0Data(jpg);
  for (int c = 0; c < 3; ++c) {
    guetzli::JPEGComponent& comp = jpg->components[c];
    memcpy(comp.coeffs.data(), coeffs + (size_t)c * comp.num_blocks * 64,
           sizeof(int16_t) * comp.num_blocks * 64);
    for (int k = 0; k < 64; ++k) jpg->quant[c].values[k] = 1;
  }
}

}  // namespace

extern "C" {

// ---------------------------------------------------------------- block path ----
void ref_fdct_block(int16_t* block) { guetzli::ComputeBlockDCT(block); }
void ref_idct_block(const int16_t* block, uint8_t* out) {
  guetzli::ComputeBlockIDCT(block, out);
}
int ref_quantize_block(int16_t* block, const int* q) {
  return guetzli::QuantizeBlock(block, q) ? 1 : 0;
}
void ref_dct_double(double* block) { guetzli::ComputeBlockDCTDouble(block); }
void ref_idct_double(double* block) { guetzli::ComputeBlockIDCTDouble(block); }
// OutputImageComponent::ToFloatPixels (output_image.cc:99-121), stride 1: one component's
// coefficients [nb][64] -> w*h floats.
void ref_to_float_pixels(const int16_t* coeffs, int w, int h, float* out) {
  guetzli::OutputImageComponent comp(w, h);
  const int bw = comp.width_in_blocks(), bh = comp.height_in_blocks();
  for (int by = 0; by < bh; ++by)
    for (int bx = 0; bx < bw; ++bx)
      comp.SetCoeffBlock(bx, by, coeffs + ((size_t)by * bw + bx) * 64);
  comp.ToFloatPixels(out, 1);
}
// OutputImage::Downsample (output_image.cc:304-340) with sharpening and blurring off, i.e.
// ToFloatPixels + SetDownsampledCoefficients(:265-300) of the two chroma components by
// (fx, fy).  coeffs = 3 components at 4:4:4; out_u / out_v receive
// ceil(w/(8fx))*ceil(h/(8fy)) blocks each.  Returns the number of blocks per component.
int ref_downsample_plain(const int16_t* coeffs, int w, int h, int fx, int fy,
                         int16_t* out_u, int16_t* out_v) {
  guetzli::OutputImage img(w, h);
  FillOutputImage(&img, coeffs);
  guetzli::OutputImage::DownsampleConfig cfg;
  cfg.u_factor_x = cfg.v_factor_x = fx;
  cfg.u_factor_y = cfg.v_factor_y = fy;
  cfg.u_sharpen = cfg.u_blur = cfg.v_sharpen = cfg.v_blur = false;
  cfg.use_silver_screen = false;
  img.Downsample(cfg);
  int nb = 0;
  for (int c = 1; c < 3; ++c) {
    const guetzli::OutputImageComponent& comp = img.component(c);
    nb = comp.width_in_blocks() * comp.height_in_blocks();
    int16_t* dst = c == 1 ? out_u : out_v;
    for (int by = 0; by < comp.height_in_blocks(); ++by)
      for (int bx = 0; bx < comp.width_in_blocks(); ++bx)
        comp.GetCoeffBlock(bx, by, dst + ((size_t)by * comp.width_in_blocks() + bx) * 64);
  }
  return nb;
}
void ref_ycbcr_to_rgb(uint8_t* pixels, int npix) {
  for (int i = 0; i < npix; ++i) guetzli::ColorTransformYCbCrToRGB(pixels + 3 * i);
}
void ref_srgb_to_linear_table(double* out256) {
  memcpy(out256, guetzli::Srgb8ToLinearTable(), 256 * sizeof(double));
}
double ref_butteraugli_score_for_quality(double q) {
  return guetzli::ButteraugliScoreForQuality(q);
}
double ref_score_jpeg(double dist, int size, double target) {
  return guetzli::ScoreJPEG(dist, size, target);
}

// rgb (packed u8, w*h*3) -> unquantised YCbCr DCT coefficients (EncodeRGBToJpeg).
int ref_encode_rgb(const uint8_t* rgb, int w, int h, int16_t* coeffs) {
  std::vector<uint8_t> v(rgb, rgb + (size_t)3 * w * h);
  guetzli::JPEGData jpg;
  if (!guetzli::EncodeRGBToJpeg(v, w, h, &jpg)) return -1;
  for (int c = 0; c < 3; ++c) {
    const guetzli::JPEGComponent& comp = jpg.components[c];
    memcpy(coeffs + (size_t)c * comp.num_blocks * 64, comp.coeffs.data(),
           sizeof(int16_t) * comp.num_blocks * 64);
  }
  return 0;
}

// coeffs (dequantised) -> optional global quantisation q[3][64] (may be null) ->
// IDCT -> sRGB u8 (packed, may be null) and linear RGB float planes (3*N, may be null).
// coeffs_out (may be null) receives the coefficients after quantisation.
void ref_reconstruct(const int16_t* coeffs, int w, int h, const int* q,
                     int16_t* coeffs_out, uint8_t* srgb, float* linear) {
  guetzli::OutputImage img(w, h);
  FillOutputImage(&img, coeffs);
  if (q) {
    int qq[3][guetzli::kDCTBlockSize];
    memcpy(qq, q, sizeof(qq));
    img.ApplyGlobalQuantization(qq);
  }
  if (coeffs_out) {
    for (int c = 0; c < 3; ++c) {
      const guetzli::OutputImageComponent& comp = img.component(c);
      size_t n = (size_t)comp.width_in_blocks() * comp.height_in_blocks() * 64;
      memcpy(coeffs_out + c * n, comp.coeffs(), n * sizeof(int16_t));
    }
  }
  if (srgb) {
    std::vector<uint8_t> v = img.ToSRGB();
    memcpy(srgb, v.data(), v.size());
  }
  if (linear) {
    std::vector<std::vector<float> > rgb(3, std::vector<float>((size_t)w * h));
    img.ToLinearRGB(&rgb);
    for (int c = 0; c < 3; ++c)
      memcpy(linear + (size_t)c * w * h, rgb[c].data(), sizeof(float) * w * h);
  }
}

// ---------------------------------------------------------- butteraugli stages ----
int ref_compute_kernel(float sigma, float* taps, int cap) {
  std::vector<float> k = butteraugli::ComputeKernel(sigma);
  if ((int)k.size() > cap) return -(int)k.size();
  memcpy(taps, k.data(), k.size() * sizeof(float));
  return (int)k.size();
}

void ref_blur(const float* in, int w, int h, float sigma, float border_ratio,
              float* out) {
  std::vector<ImageF> p = PlanesFromFlat(in, w, h, 1);
  ImageF b = butteraugli::Blur(p[0], sigma, border_ratio);
  FlatFromPlane(b, out);
}

// linear rgb planes (3N) -> xyb planes (3N)   (OpsinDynamicsImage)
void ref_opsin(const float* rgb, int w, int h, float* xyb) {
  std::vector<ImageF> p = PlanesFromFlat(rgb, w, h, 3);
  std::vector<ImageF> x = butteraugli::OpsinDynamicsImage(p);
  FlatFromPlanes(x, xyb);
}

// xyb (3N) -> 10 planes: lf[0..2], mf[0..2], hf[0..1], uhf[0..1]
void ref_separate_frequencies(const float* xyb, int w, int h, float* out10) {
  std::vector<ImageF> p = PlanesFromFlat(xyb, w, h, 3);
  butteraugli::PsychoImage ps;
  butteraugli::SeparateFrequencies(w, h, p, ps);
  size_t n = (size_t)w * h;
  FlatFromPlanes(ps.lf, out10);
  FlatFromPlanes(ps.mf, out10 + 3 * n);
  FlatFromPlanes(ps.hf, out10 + 6 * n);
  FlatFromPlanes(ps.uhf, out10 + 8 * n);
}

// butteraugli::Mask(xyb0, xyb1) -> mask[3N], mask_dc[3N]
void ref_mask(const float* xyb0, const float* xyb1, int w, int h, float* mask,
              float* mask_dc) {
  std::vector<ImageF> a = PlanesFromFlat(xyb0, w, h, 3);
  std::vector<ImageF> b = PlanesFromFlat(xyb1, w, h, 3);
  std::vector<ImageF> m, mdc;
  butteraugli::Mask(a, b, &m, &mdc);
  FlatFromPlanes(m, mask);
  FlatFromPlanes(mdc, mask_dc);
}

// One Malta accumulation: acc += Malta{,LF}(lum0, lum1).  lf != 0 selects MaltaDiffMapLF.
void ref_malta(const float* lum0, const float* lum1, int w, int h, int lf,
               double w_0gt1, double w_0lt1, double norm1, float* acc) {
  std::vector<ImageF> a = PlanesFromFlat(lum0, w, h, 1);
  std::vector<ImageF> b = PlanesFromFlat(lum1, w, h, 1);
  std::vector<ImageF> c = PlanesFromFlat(acc, w, h, 1);
  // The comparator object is only used for its xsize_/ysize_.
  std::vector<ImageF> dummy = butteraugli::CreatePlanes<float>(w, h, 3);
  for (int i = 0; i < 3; ++i)
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) dummy[i].Row(y)[x] = 1.0f;
  butteraugli::ButteraugliComparator cmp(dummy);
  if (lf)
    cmp.MaltaDiffMapLF(a[0], b[0], w_0gt1, w_0lt1, norm1, &c[0]);
  else
    cmp.MaltaDiffMap(a[0], b[0], w_0gt1, w_0lt1, norm1, &c[0]);
  FlatFromPlane(c[0], acc);
}

// Full butteraugli: linear rgb0, rgb1 (3N each) -> diffmap (N); returns max score.
double ref_diffmap(const float* rgb0, const float* rgb1, int w, int h,
                   float* diffmap) {
  std::vector<ImageF> a = PlanesFromFlat(rgb0, w, h, 3);
  std::vector<ImageF> b = PlanesFromFlat(rgb1, w, h, 3);
  butteraugli::ButteraugliComparator cmp(a);
  ImageF d;
  cmp.Diffmap(b, d);
  if (diffmap) FlatFromPlane(d, diffmap);
  return butteraugli::ButteraugliScoreFromDiffmap(d);
}

// ------------------------------------------------ guetzli comparator (a17-a19) ----
struct RefComparator {
  std::vector<uint8_t> rgb;
  guetzli::ProcessStats stats;
  guetzli::ButteraugliComparator* cmp;
  int w, h;
};

void* ref_comparator_create(const uint8_t* rgb, int w, int h, float target) {
  RefComparator* r = new RefComparator;
  r->rgb.assign(rgb, rgb + (size_t)3 * w * h);
  r->w = w;
  r->h = h;
  r->cmp = new guetzli::ButteraugliComparator(w, h, &r->rgb, target, &r->stats);
  return r;
}
void ref_comparator_destroy(void* p) {
  RefComparator* r = (RefComparator*)p;
  delete r->cmp;
  delete r;
}
// Compare(): coeffs are dequantised coefficients of the candidate.
float ref_comparator_compare(void* p, const int16_t* coeffs, float* distmap) {
  RefComparator* r = (RefComparator*)p;
  guetzli::OutputImage img(r->w, r->h);
  FillOutputImage(&img, coeffs);
  r->cmp->Compare(img);
  if (distmap) {
    std::vector<float> d = r->cmp->distmap();
    memcpy(distmap, d.data(), d.size() * sizeof(float));
  }
  return r->cmp->distmap_aggregate();
}
void ref_comparator_block_weights(void* p, int direction, int max_block_dist,
                                  double target_mul, const float* distmap,
                                  float* block_weight) {
  RefComparator* r = (RefComparator*)p;
  std::vector<float> d(distmap, distmap + (size_t)r->w * r->h);
  int bw = (r->w + 7) / 8, bh = (r->h + 7) / 8;
  std::vector<float> wgt(block_weight, block_weight + (size_t)bw * bh);
  r->cmp->ComputeBlockErrorAdjustmentWeights(direction, max_block_dist, target_mul,
                                             1, 1, d, &wgt);
  memcpy(block_weight, wgt.data(), wgt.size() * sizeof(float));
}
// mask_xyz_ planes used by CompareBlock (StartBlockComparisons): 3N floats.
void ref_comparator_block_mask(void* p, float* mask3) {
  RefComparator* r = (RefComparator*)p;
  r->cmp->StartBlockComparisons();
  FlatFromPlanes(r->cmp->mask_xyz_, mask3);
  r->cmp->FinishBlockComparisons();
}
// One CompareBlock evaluation for block (bx,by) of the image given by coeffs.
double ref_comparator_compare_block(void* p, const int16_t* coeffs, int bx, int by) {
  RefComparator* r = (RefComparator*)p;
  guetzli::OutputImage img(r->w, r->h);
  FillOutputImage(&img, coeffs);
  r->cmp->StartBlockComparisons();
  r->cmp->SwitchBlock(bx, by, 1, 1);
  double d = r->cmp->CompareBlock(img, 0, 0);
  r->cmp->FinishBlockComparisons();
  return d;
}

// Phase A of SelectFrequencyMasking (processor.cc:554-590) for comp_mask 7, 4:4:4:
// `coeffs` = current (globally quantised, dequantised values) image, `orig` = the
// unquantised coefficients.  Outputs CSR arrays; returns total candidate count, or
// -needed if cap is too small.
int ref_block_zeroing_orders(void* p, const int16_t* coeffs, const int16_t* orig,
                             int lookahead, int new_model, int32_t* offsets,
                             uint8_t* idx, float* err, int cap) {
  RefComparator* r = (RefComparator*)p;
  const int w = r->w, h = r->h;
  const int bw = (w + 7) / 8, bh = (h + 7) / 8, nb = bw * bh;
  guetzli::OutputImage img(w, h);
  FillOutputImage(&img, coeffs);
  guetzli::Processor proc;
  proc.params_.zeroing_greedy_lookahead = lookahead;
  proc.params_.new_zeroing_model = new_model != 0;
  proc.comparator_ = r->cmp;
  proc.stats_ = &r->stats;
  r->cmp->StartBlockComparisons();
  std::vector<guetzli::CoeffData> order;
  int total = 0;
  for (int by = 0, bix = 0; by < bh; ++by) {
    for (int bx = 0; bx < bw; ++bx, ++bix) {
      int16_t block[192], oblock[192];
      for (int c = 0; c < 3; ++c) {
        memcpy(block + 64 * c, coeffs + ((size_t)c * nb + bix) * 64, 128);
        memcpy(oblock + 64 * c, orig + ((size_t)c * nb + bix) * 64, 128);
      }
      order.clear();
      proc.ComputeBlockZeroingOrder(block, oblock, bx, by, 1, 1, 7, &img, &order);
      offsets[bix] = total;
      for (size_t i = 0; i < order.size(); ++i) {
        if (total < cap) {
          idx[total] = (uint8_t)order[i].idx;
          err[total] = order[i].block_err;
        }
        ++total;
      }
    }
  }
  offsets[nb] = total;
  r->cmp->FinishBlockComparisons();
  return total <= cap ? total : -total;
}

// ------------------------------------------------------------- whole encoder ----
// guetzli::Process(params, stats, rgb, w, h, &out).  If trace != null the --verbose
// trace is copied there (NUL-terminated, truncated to trace_cap).
// Returns output size (bytes written to out if <= cap), or -1 on failure.
long ref_process(const uint8_t* rgb, int w, int h, float butteraugli_target,
                 uint8_t* out, long cap, char* trace, long trace_cap) {
  std::vector<uint8_t> v(rgb, rgb + (size_t)3 * w * h);
  guetzli::Params params;
  params.butteraugli_target = butteraugli_target;
  guetzli::ProcessStats stats;
  std::string dbg;
  if (trace) stats.debug_output = &dbg;
  std::string jpg;
  if (!guetzli::Process(params, &stats, v, w, h, &jpg)) return -1;
  if ((long)jpg.size() <= cap) memcpy(out, jpg.data(), jpg.size());
  if (trace && trace_cap > 0) {
    size_t n = std::min<size_t>(dbg.size(), trace_cap - 1);
    memcpy(trace, dbg.data(), n);
    trace[n] = 0;
  }
  return (long)jpg.size();
}

// Serialise an image given by (already quantised-and-dequantised) coefficients and
// its quant tables exactly as Processor::TryQuantMatrix does
// (SaveToJpegData + WriteJpeg).  Returns size; bytes copied if they fit.
long ref_write_jpeg(const int16_t* coeffs, int w, int h, const int* q, uint8_t* out,
                    long cap) {
  guetzli::OutputImage img(w, h);
  FillOutputImage(&img, coeffs);
  int qq[3][guetzli::kDCTBlockSize];
  memcpy(qq, q, sizeof(qq));
  img.ApplyGlobalQuantization(qq);
  guetzli::JPEGData jpg;
  std::vector<int16_t> zero((size_t)3 * ((w + 7) / 8) * ((h + 7) / 8) * 64);
  JpegDataFromCoeffs(zero.data(), w, h, &jpg);
  img.SaveToJpegData(&jpg);
  std::string s;
  guetzli::JPEGOutput o(guetzli::GuetzliStringOut, &s);
  if (!guetzli::WriteJpeg(jpg, true, o)) return -1;
  if ((long)s.size() <= cap) memcpy(out, s.data(), s.size());
  return (long)s.size();
}

// ReadJpeg(data, JPEG_READ_ALL) as the canonical dump gzh_read_jpeg also produces (see
// guetzli_amd/host/processor.cc); -1 if the reference rejects the stream.
long ref_read_jpeg(const uint8_t* data, long len, uint8_t* out, long cap) {
  guetzli::JPEGData jpg;
  if (!guetzli::ReadJpeg(data, (size_t)len, guetzli::JPEG_READ_ALL, &jpg)) return -1;
  std::string d;
  auto put32 = [&](int32_t v) { d.append((const char*)&v, 4); };
  auto puts = [&](const std::string& s) { put32((int32_t)s.size()); d.append(s); };
  put32(jpg.width); put32(jpg.height); put32((int32_t)jpg.components.size());
  for (const auto& c : jpg.components) {
    put32(c.id); put32(c.h_samp_factor); put32(c.v_samp_factor); put32((int32_t)c.quant_idx);
    put32(c.width_in_blocks); put32(c.height_in_blocks);
  }
  put32((int32_t)jpg.quant.size());
  for (const auto& q : jpg.quant) {
    put32(q.index); put32(q.precision);
    for (int k = 0; k < 64; ++k) put32(q.values[k]);
  }
  put32((int32_t)jpg.app_data.size());
  for (const auto& a : jpg.app_data) puts(a);
  put32((int32_t)jpg.com_data.size());
  for (const auto& a : jpg.com_data) puts(a);
  puts(jpg.tail_data);
  for (const auto& c : jpg.components) d.append((const char*)c.coeffs.data(), c.coeffs.size() * 2);
  if ((long)d.size() <= cap) memcpy(out, d.data(), d.size());
  return (long)d.size();
}

// guetzli::Process(params, stats, jpeg_data, &out) (processor.cc:890-924), optional trace.
long ref_process_jpeg(const uint8_t* data, long len, float butteraugli_target,
                      int clear_metadata, uint8_t* out, long cap, char* trace, long trace_cap) {
  guetzli::Params params;
  params.butteraugli_target = butteraugli_target;
  params.clear_metadata = clear_metadata != 0;
  guetzli::ProcessStats stats;
  std::string dbg;
  if (trace) stats.debug_output = &dbg;
  std::string in((const char*)data, (size_t)len), jpg;
  if (!guetzli::Process(params, &stats, in, &jpg)) return -1;
  if ((long)jpg.size() <= cap) memcpy(out, jpg.data(), jpg.size());
  if (trace && trace_cap > 0) {
    const size_t n = std::min<size_t>(dbg.size(), (size_t)trace_cap - 1);
    memcpy(trace, dbg.data(), n);
    trace[n] = 0;
  }
  return (long)jpg.size();
}


// ------------------------------------------------------------------ YUV 4:2:0 ----
// Frame layout used across this repo for a 4:2:0 image: nb luma blocks (8x8 grid), then
// nbc = ceil(w/16)*ceil(h/16) blocks of Cb, then nbc of Cr.
namespace {
void FillOutputImage420(guetzli::OutputImage* img, const int16_t* coeffs) {
  size_t at = 0;
  for (int c = 0; c < 3; ++c) {
    guetzli::OutputImageComponent& comp = img->component(c);
    if (c > 0) comp.Reset(2, 2);
    const int bw = comp.width_in_blocks(), bh = comp.height_in_blocks();
    for (int by = 0; by < bh; ++by)
      for (int bx = 0; bx < bw; ++bx, ++at) comp.SetCoeffBlock(bx, by, coeffs + at * 64);
  }
}
void DumpOutputImage(const guetzli::OutputImage& img, int16_t* out) {
  size_t at = 0;
  for (int c = 0; c < 3; ++c) {
    const guetzli::OutputImageComponent& comp = img.component(c);
    const size_t n = (size_t)comp.width_in_blocks() * comp.height_in_blocks() * 64;
    memcpy(out + at, comp.coeffs(), n * sizeof(int16_t));
    at += n;
  }
}
}  // namespace

// OutputImage::Downsample with Processor::DownsampleImage's configuration
// (processor.cc:97-104): 4:4:4 coefficients in, 4:2:0 frame out.  Returns the number of
// blocks written (nb + 2*nbc; 3*nb if the image is greyscale and nothing happens).
int ref_downsample(const int16_t* coeffs, int w, int h, int use_silver_screen, int16_t* out) {
  guetzli::OutputImage img(w, h);
  FillOutputImage(&img, coeffs);
  guetzli::OutputImage::DownsampleConfig cfg;
  cfg.use_silver_screen = use_silver_screen != 0;
  img.Downsample(cfg);
  DumpOutputImage(img, out);
  int n = 0;
  for (int c = 0; c < 3; ++c) n += img.component(c).width_in_blocks() * img.component(c).height_in_blocks();
  return n;
}

// 4:2:0 coefficients -> optional quantisation -> pixels (OutputImage::ToSRGB / ToLinearRGB with
// the 2x2 pixel model of UpdatePixelsForBlock).  `shuffle` != 0 first applies random extra
// SetCoeffBlock calls (garbage, then the real block again) in a random order: the pixel cache
// must not depend on the update history.
void ref_reconstruct420(const int16_t* coeffs, int w, int h, const int* q, int shuffle,
                        int16_t* coeffs_out, uint8_t* srgb, float* linear) {
  guetzli::OutputImage img(w, h);
  FillOutputImage420(&img, coeffs);
  if (shuffle) {
    unsigned rng = (unsigned)shuffle;
    auto next = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
    const int nb = img.component(0).width_in_blocks() * img.component(0).height_in_blocks();
    for (int rep = 0; rep < 3; ++rep)
      for (int c = 1; c < 3; ++c) {
        guetzli::OutputImageComponent& comp = img.component(c);
        const int bw = comp.width_in_blocks(), bh = comp.height_in_blocks();
        const int16_t* base = coeffs + ((size_t)nb + (size_t)(c - 1) * bw * bh) * 64;
        for (int t = 0; t < bw * bh; ++t) {
          const int b = next() % (bw * bh);
          int16_t junk[64];
          for (int k = 0; k < 64; ++k) junk[k] = (int16_t)((int)(next() % 401) - 200);
          comp.SetCoeffBlock(b % bw, b / bw, junk);
          const int b2 = next() % (bw * bh);
          comp.SetCoeffBlock(b2 % bw, b2 / bw, base + (size_t)b2 * 64);
          comp.SetCoeffBlock(b % bw, b / bw, base + (size_t)b * 64);
        }
      }
  }
  if (q) {
    int qq[3][guetzli::kDCTBlockSize];
    memcpy(qq, q, sizeof(qq));
    img.ApplyGlobalQuantization(qq);
  }
  if (coeffs_out) DumpOutputImage(img, coeffs_out);
  if (srgb) {
    std::vector<uint8_t> v = img.ToSRGB();
    memcpy(srgb, v.data(), v.size());
  }
  if (linear) {
    std::vector<std::vector<float> > rgb(3, std::vector<float>((size_t)w * h));
    img.ToLinearRGB(&rgb);
    for (int c = 0; c < 3; ++c)
      memcpy(linear + (size_t)c * w * h, rgb[c].data(), sizeof(float) * w * h);
  }
}

float ref_comparator_compare420(void* p, const int16_t* coeffs, float* distmap) {
  RefComparator* r = (RefComparator*)p;
  guetzli::OutputImage img(r->w, r->h);
  FillOutputImage420(&img, coeffs);
  r->cmp->Compare(img);
  if (distmap) {
    std::vector<float> d = r->cmp->distmap();
    memcpy(distmap, d.data(), d.size() * sizeof(float));
  }
  return r->cmp->distmap_aggregate();
}

void ref_comparator_block_weights_factor(void* p, int direction, int max_block_dist,
                                         double target_mul, int factor, const float* distmap,
                                         float* block_weight) {
  RefComparator* r = (RefComparator*)p;
  std::vector<float> d(distmap, distmap + (size_t)r->w * r->h);
  const int s = 8 * factor;
  const int bw = (r->w + s - 1) / s, bh = (r->h + s - 1) / s;
  std::vector<float> wgt(block_weight, block_weight + (size_t)bw * bh);
  r->cmp->ComputeBlockErrorAdjustmentWeights(direction, max_block_dist, target_mul,
                                             factor, factor, d, &wgt);
  memcpy(block_weight, wgt.data(), wgt.size() * sizeof(float));
}

// Phase A of SelectFrequencyMasking (processor.cc:554-590) for any comp_mask on a 4:4:4
// (frame420 == 0) or 4:2:0 frame: CSR arrays over the grid of the mask's last component.
int ref_block_zeroing_orders_masked(void* p, const int16_t* coeffs, const int16_t* orig,
                                    int frame420, int comp_mask, int lookahead, int new_model,
                                    int32_t* offsets, uint8_t* idx, float* err, int cap) {
  RefComparator* r = (RefComparator*)p;
  const int w = r->w, h = r->h;
  guetzli::OutputImage img(w, h);
  if (frame420) FillOutputImage420(&img, coeffs); else FillOutputImage(&img, coeffs);
  int last_c = 0;
  for (int c = 0; c < 3; ++c) if (comp_mask & (1 << c)) last_c = c;
  const int factor = img.component(last_c).factor_x();
  const int gw = (w + 8 * factor - 1) / (8 * factor), gh = (h + 8 * factor - 1) / (8 * factor);
  size_t coff[3] = {0, 0, 0};
  for (int c = 1; c < 3; ++c)
    coff[c] = coff[c - 1] + (size_t)img.component(c - 1).width_in_blocks() * img.component(c - 1).height_in_blocks();
  guetzli::Processor proc;
  proc.params_.zeroing_greedy_lookahead = lookahead;
  proc.params_.new_zeroing_model = new_model != 0;
  proc.comparator_ = r->cmp;
  proc.stats_ = &r->stats;
  r->cmp->StartBlockComparisons();
  std::vector<guetzli::CoeffData> order;
  int total = 0;
  for (int by = 0, bix = 0; by < gh; ++by) {
    for (int bx = 0; bx < gw; ++bx, ++bix) {
      int16_t block[192] = {0}, oblock[192] = {0};
      for (int c = 0; c < 3; ++c) {
        if (!(comp_mask & (1 << c))) continue;
        memcpy(block + 64 * c, coeffs + (coff[c] + bix) * 64, 128);
        memcpy(oblock + 64 * c, orig + (coff[c] + bix) * 64, 128);
      }
      order.clear();
      proc.ComputeBlockZeroingOrder(block, oblock, bx, by, factor, factor, (uint8_t)comp_mask, &img, &order);
      offsets[bix] = total;
      for (size_t i = 0; i < order.size(); ++i) {
        if (total < cap) {
          idx[total] = (uint8_t)order[i].idx;
          err[total] = order[i].block_err;
        }
        ++total;
      }
    }
  }
  offsets[gw * gh] = total;
  r->cmp->FinishBlockComparisons();
  return total <= cap ? total : -total;
}

// SaveToJpegData + WriteJpeg of a 4:2:0 image (coefficients before quantisation by q).
long ref_write_jpeg420(const int16_t* coeffs, int w, int h, const int* q, uint8_t* out, long cap) {
  guetzli::OutputImage img(w, h);
  FillOutputImage420(&img, coeffs);
  int qq[3][guetzli::kDCTBlockSize];
  memcpy(qq, q, sizeof(qq));
  img.ApplyGlobalQuantization(qq);
  guetzli::JPEGData jpg;
  std::vector<int16_t> zero((size_t)3 * ((w + 7) / 8) * ((h + 7) / 8) * 64);
  JpegDataFromCoeffs(zero.data(), w, h, &jpg);
  img.SaveToJpegData(&jpg);
  std::string s;
  guetzli::JPEGOutput o(guetzli::GuetzliStringOut, &s);
  if (!guetzli::WriteJpeg(jpg, true, o)) return -1;
  if ((long)s.size() <= cap) memcpy(out, s.data(), s.size());
  return (long)s.size();
}

// guetzli::Process with every field of Params (processor.h:29-37).  jpeg_len < 0: `data` is
// packed RGB of w x h; otherwise JPEG bytes.
long ref_process_params(const uint8_t* data, long jpeg_len, int w, int h, float butteraugli_target,
                        int clear_metadata, int try_420, int force_420, int use_silver_screen,
                        int lookahead, int new_model, uint8_t* out, long cap, char* trace,
                        long trace_cap) {
  guetzli::Params params;
  params.butteraugli_target = butteraugli_target;
  params.clear_metadata = clear_metadata != 0;
  params.try_420 = try_420 != 0;
  params.force_420 = force_420 != 0;
  params.use_silver_screen = use_silver_screen != 0;
  params.zeroing_greedy_lookahead = lookahead;
  params.new_zeroing_model = new_model != 0;
  guetzli::ProcessStats stats;
  std::string dbg;
  if (trace) stats.debug_output = &dbg;
  std::string jpg;
  bool ok;
  if (jpeg_len < 0) {
    std::vector<uint8_t> v(data, data + (size_t)3 * w * h);
    ok = guetzli::Process(params, &stats, v, w, h, &jpg);
  } else {
    std::string in((const char*)data, (size_t)jpeg_len);
    ok = guetzli::Process(params, &stats, in, &jpg);
  }
  if (!ok) return -1;
  if ((long)jpg.size() <= cap) memcpy(out, jpg.data(), jpg.size());
  if (trace && trace_cap > 0) {
    const size_t n = std::min<size_t>(dbg.size(), (size_t)trace_cap - 1);
    memcpy(trace, dbg.data(), n);
    trace[n] = 0;
  }
  return (long)jpg.size();
}

}  // extern "C"
