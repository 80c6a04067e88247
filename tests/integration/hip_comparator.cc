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

// butteraugli_comparator.cc:63-75.  OutputImage keeps every component's coefficients
// contiguous and block-major (output_image.h:33-40): the three arrays one after the other are
// the C ABI's coefficient layout.
void HipButteraugliComparator::Compare(const OutputImage& img) {
  std::vector<int16_t> coeffs;
  for (int c = 0; c < 3; ++c) {
    const OutputImageComponent& comp = img.component(c);
    if (comp.factor_x() != 1 || comp.factor_y() != 1) Die("Compare: YUV420 frames go through gz_downsample", GZ_E_STATE);
    const size_t n = (size_t)comp.width_in_blocks() * comp.height_in_blocks() * kDCTBlockSize;
    coeffs.insert(coeffs.end(), comp.coeffs(), comp.coeffs() + n);
  }
  int rc = gz_set_coeffs(ctx_, coeffs.data());
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

// :427-455
void HipButteraugliComparator::SwitchBlock(int block_x, int block_y, int factor_x, int factor_y) {
  if (factor_x != 1 || factor_y != 1) Die("SwitchBlock: factor 1 only through the per-block seam", GZ_E_ARG);
  block_x_ = block_x;
  block_y_ = block_y;
}

// :457-488 -- one round trip per call: correct, and two orders of magnitude slower than the
// batched gz_block_zeroing_orders the repository's own driver uses
double HipButteraugliComparator::CompareBlock(const OutputImage& img, int off_x, int off_y) const {
  int16_t blocks[3 * kDCTBlockSize];
  for (int c = 0; c < 3; ++c) img.component(c).GetCoeffBlock(block_x_ + off_x, block_y_ + off_y, &blocks[c * kDCTBlockSize]);
  const int32_t xy[2] = {block_x_ + off_x, block_y_ + off_y};
  double d = 0.0;
  const int rc = gz_compare_blocks(ctx_, 1, xy, blocks, &d);
  if (rc != GZ_OK) Die("gz_compare_blocks", rc);
  ++compare_block_calls_;
  return d;
}

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
