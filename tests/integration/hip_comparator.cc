// See hip_comparator.h.  Every method is the device counterpart of the same method of
// guetzli::ButteraugliComparator (guetzli/butteraugli_comparator.cc), cited per method.
#include "hip_comparator.h"

#include <assert.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "guetzli/debug_print.h"
#include "guetzli/score.h"

namespace guetzli {

HipButteraugliComparator::HipButteraugliComparator(int width, int height,
                                                   const std::vector<uint8_t>* rgb,
                                                   float target_distance, ProcessStats* stats,
                                                   int device)
    : width_(width), height_(height), target_distance_(target_distance), stats_(stats) {
  // butteraugli_comparator.cc:51-61: the original's linear RGB and PsychoImage
  int err = 0;
  ctx_ = gz_create(device, width, height, rgb->data(), target_distance, &err);
  if (!ctx_) fprintf(stderr, "gz_create: %s\n", gz_strerror(err));
}

HipButteraugliComparator::~HipButteraugliComparator() { gz_destroy(ctx_); }

void HipButteraugliComparator::Die(const char* what, int rc) const {
  fprintf(stderr, "%s: %s (%s)\n", what, gz_strerror(rc), gz_last_error(ctx_));
  abort();   // the reference's interface has no error path (comparator.h); its asserts abort too
}

// The frame of an image: 1 = 4:4:4, 2 = 4:2:0 (the two the reference's Processor produces,
// processor.cc:97-104,811-815).
int HipButteraugliComparator::FrameOf(const OutputImage& img) const {
  const int fx = img.component(1).factor_x(), fy = img.component(1).factor_y();
  if (img.component(0).factor_x() != 1 || img.component(0).factor_y() != 1 || fx != fy ||
      img.component(2).factor_x() != fx || img.component(2).factor_y() != fy || (fx != 1 && fx != 2))
    Die("unsupported sampling factors", GZ_E_ARG);
  return fx;
}

// OutputImage keeps every component's coefficients contiguous and block-major
// (output_image.h:33-40): the three arrays one after the other are the C ABI's coefficient
// layout, for a 4:4:4 frame and for a 4:2:0 one (chroma on its own, smaller block grid).
void HipButteraugliComparator::GatherCoeffs(const OutputImage& img, std::vector<int16_t>* out) const {
  out->clear();
  for (int c = 0; c < 3; ++c) {
    const OutputImageComponent& comp = img.component(c);
    const size_t n = (size_t)comp.width_in_blocks() * comp.height_in_blocks() * kDCTBlockSize;
    out->insert(out->end(), comp.coeffs(), comp.coeffs() + n);
  }
}

// butteraugli_comparator.cc:63-75
void HipButteraugliComparator::Compare(const OutputImage& img) {
  std::vector<int16_t> coeffs;
  GatherCoeffs(img, &coeffs);
  int rc = gz_set_frame(ctx_, FrameOf(img));
  if (rc != GZ_OK) Die("gz_set_frame", rc);
  rc = gz_set_coeffs(ctx_, coeffs.data());
  if (rc != GZ_OK) Die("gz_set_coeffs", rc);
  distmap_.resize((size_t)width_ * height_);
  rc = gz_compare(ctx_, &distance_, distmap_.data(), nullptr);
  if (rc != GZ_OK) Die("gz_compare", rc);
  ++compare_calls_;
  GUETZLI_LOG(stats_, " BA[100.00%%] D[%6.4f]", distance_);
}

// :415-425 -- the mask of the original is prepared by the library on first use
void HipButteraugliComparator::StartBlockComparisons() {}
void HipButteraugliComparator::FinishBlockComparisons() {}

// :427-455 -- the original's 8x8 opsin images of the macro-block are computed by the device
// call that compares them
void HipButteraugliComparator::SwitchBlock(int block_x, int block_y, int factor_x, int factor_y) {
  block_x_ = block_x;
  block_y_ = block_y;
  factor_x_ = factor_x;
  factor_y_ = factor_y;
}

// :457-488 -- one round trip per call: correct, and two orders of magnitude slower than the
// batched gz_block_zeroing_orders the repository's own driver uses.
// What CompareBlock reads of the image are the pixels of the 8x8 window (ToLinearRGB(xmin, ymin,
// 8, 8), :467): they go to the device as they are, whatever the frame.
double HipButteraugliComparator::CompareBlock(const OutputImage& img, int off_x, int off_y) const {
  const int bx = block_x_ * factor_x_ + off_x, by = block_y_ * factor_y_ + off_y;
  uint8_t ycc[3 * kDCTBlockSize];
  for (int c = 0; c < 3; ++c) img.component(c).ToPixels(8 * bx, 8 * by, 8, 8, &ycc[c * kDCTBlockSize], 1);
  const int32_t xy[2] = {bx, by};
  double d = 0.0;
  const int rc = gz_compare_block_pixels(ctx_, 1, xy, ycc, &d);
  if (rc != GZ_OK) Die("gz_compare_block_pixels", rc);
  ++compare_block_calls_;
  return d;
}

#ifdef GUETZLI_BATCHED_BLOCK_SEARCH
// processor.cc:554-590 for the whole grid at once.  The original's coefficients are jpg's
// (:566-574; for a 4:2:0 frame jpg carries MCU padding, SaveToJpegData output_image.cc:348-409:
// blocks are picked by their position), the candidate's are img's.
bool HipButteraugliComparator::ComputeAllBlockZeroingOrders(
    const JPEGData& jpg, const OutputImage& img, uint8_t comp_mask, int lookahead,
    bool new_zeroing_model, std::vector<int>* candidate_coeff_offsets,
    std::vector<uint8_t>* candidate_coeffs, std::vector<float>* candidate_coeff_errors) {
  if (jpg.components.size() != 3) return false;   // (greyscale output: let the per-block loop do it)
  const int frame = FrameOf(img);
  std::vector<int16_t> orig, cand;
  for (int c = 0; c < 3; ++c) {
    const OutputImageComponent& comp = img.component(c);
    const JPEGComponent& jc = jpg.components[c];
    for (int by = 0; by < comp.height_in_blocks(); ++by)
      for (int bx = 0; bx < comp.width_in_blocks(); ++bx) {
        const coeff_t* src = &jc.coeffs[((size_t)by * jc.width_in_blocks + bx) * kDCTBlockSize];
        orig.insert(orig.end(), src, src + kDCTBlockSize);
      }
  }
  GatherCoeffs(img, &cand);
  int rc = frame == 2 ? gz_set_orig_coeffs_420(ctx_, orig.data()) : gz_set_orig_coeffs(ctx_, orig.data());
  if (rc != GZ_OK) Die("gz_set_orig_coeffs", rc);
  rc = gz_set_coeffs(ctx_, cand.data());
  if (rc != GZ_OK) Die("gz_set_coeffs", rc);
  const int last_c = comp_mask >= 4 ? 2 : (comp_mask >= 2 ? 1 : 0);
  const OutputImageComponent& grid = img.component(last_c);
  const int nb = grid.width_in_blocks() * grid.height_in_blocks();
  std::vector<int32_t> off((size_t)nb + 1);
  std::vector<uint8_t> idx((size_t)nb * 189);
  std::vector<float> err((size_t)nb * 189);
  rc = gz_block_zeroing_orders_masked(ctx_, comp_mask, lookahead, new_zeroing_model ? 1 : 0, off.data(),
                                      idx.data(), err.data(), nb * 189);
  if (rc != GZ_OK) Die("gz_block_zeroing_orders_masked", rc);
  candidate_coeff_offsets->assign(off.begin(), off.end());
  candidate_coeffs->assign(idx.begin(), idx.begin() + off[nb]);
  candidate_coeff_errors->assign(err.begin(), err.begin() + off[nb]);
  ++batched_search_calls_;
  return true;
}
#endif

double HipButteraugliComparator::ScoreOutputSize(int size) const {   // :560-562
  return ScoreJPEG(distance_, size, target_distance_);
}
bool HipButteraugliComparator::DistanceOK(double target_mul) const {   // .h:51-53
  return distance_ <= target_mul * target_distance_;
}
const std::vector<float> HipButteraugliComparator::distmap() const { return distmap_; }
float HipButteraugliComparator::distmap_aggregate() const { return distance_; }
float HipButteraugliComparator::BlockErrorLimit() const { return target_distance_; }   // :490-492

// :494-558.  Processor passes either an all-zero map (its first "up" iteration,
// processor.cc:626-630) or distmap() of the last Compare: in both cases the per-block maxima the
// weights are made of are on the device already.
void HipButteraugliComparator::ComputeBlockErrorAdjustmentWeights(
    int direction, int max_block_dist, double target_mul, int factor_x, int factor_y,
    const std::vector<float>& distmap, std::vector<float>* block_weight) {
  bool zero = true;
  for (size_t i = 0; i < distmap.size() && zero; ++i) zero = distmap[i] == 0.0f;
  if (!zero && (distmap.size() != distmap_.size() ||
                memcmp(distmap.data(), distmap_.data(), distmap.size() * sizeof(float)) != 0))
    Die("ComputeBlockErrorAdjustmentWeights: a distance map other than the last Compare's", GZ_E_ARG);
  if (factor_x != factor_y) Die("ComputeBlockErrorAdjustmentWeights: factor_x != factor_y", GZ_E_ARG);
  const int rc = gz_block_weights_factor(ctx_, direction, max_block_dist, target_mul, zero ? 0 : 1,
                                         factor_x, block_weight->data());
  if (rc != GZ_OK) Die("gz_block_weights_factor", rc);
}

}  // namespace guetzli
