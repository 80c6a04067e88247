// TEST INFRASTRUCTURE (like oracle/_ref): the reference-side binding of INTEGRATION.md section 1,
// compiled -- a `guetzli::Comparator` (guetzli/comparator.h:29-96) that drives libguetzli_amd.so
// through its C ABI, so that the UNMODIFIED reference `guetzli::ProcessJpegData`
// (guetzli/processor.h:48-50) runs with every butteraugli evaluation on the MI355X.
// tests/integration/Makefile builds it together with the reference's own sources from
// /root/reference; nothing of the reference is copied here.
//
// This class is what a guetzli maintainer adds to the reference tree
// (guetzli/hip_comparator.{h,cc}); it includes only the reference's public headers and
// include/guetzli_amd.h.
#ifndef GUETZLI_HIP_COMPARATOR_H_
#define GUETZLI_HIP_COMPARATOR_H_

#include <stdint.h>

#include <vector>

#include "guetzli/comparator.h"
#include "guetzli/jpeg_data.h"
#include "guetzli/output_image.h"
#include "guetzli/stats.h"
#include "guetzli_amd.h"

namespace guetzli {

class HipButteraugliComparator : public Comparator {
 public:
  // Same arguments as ButteraugliComparator (butteraugli_comparator.h:34-36) + a HIP ordinal.
  HipButteraugliComparator(int width, int height, const std::vector<uint8_t>* rgb,
                           float target_distance, ProcessStats* stats, int device = 0);
  ~HipButteraugliComparator() override;
  bool ok() const { return ctx_ != nullptr; }

  void Compare(const OutputImage& img) override;
  void StartBlockComparisons() override;
  void FinishBlockComparisons() override;
  void SwitchBlock(int block_x, int block_y, int factor_x, int factor_y) override;
  double CompareBlock(const OutputImage& img, int off_x, int off_y) const override;
  double ScoreOutputSize(int size) const override;
  bool DistanceOK(double target_mul) const override;
  const std::vector<float> distmap() const override;
  float distmap_aggregate() const override;
  float BlockErrorLimit() const override;
  void ComputeBlockErrorAdjustmentWeights(int direction, int max_block_dist, double target_mul,
                                          int factor_x, int factor_y,
                                          const std::vector<float>& distmap,
                                          std::vector<float>* block_weight) override;
#ifdef GUETZLI_BATCHED_BLOCK_SEARCH
  // The batched hook of INTEGRATION.md section 2 (declared in comparator.h by
  // tests/integration/patch_reference.py, called once per SelectFrequencyMasking instead of its
  // per-block loop, processor.cc:554-590): phase A for every block of the grid in one device call.
  bool ComputeAllBlockZeroingOrders(const JPEGData& jpg, const OutputImage& img, uint8_t comp_mask,
                                    int lookahead, bool new_zeroing_model,
                                    std::vector<int>* candidate_coeff_offsets,
                                    std::vector<uint8_t>* candidate_coeffs,
                                    std::vector<float>* candidate_coeff_errors) override;
#endif

  // counters for the test: how the seam was used
  long compare_calls() const { return compare_calls_; }
  long compare_block_calls() const { return compare_block_calls_; }
  long batched_search_calls() const { return batched_search_calls_; }

 private:
  void Die(const char* what, int rc) const;
  gz_ctx* ctx_ = nullptr;
  const int width_, height_;
  const float target_distance_;
  ProcessStats* stats_;
  float distance_ = 0.0f;
  std::vector<float> distmap_;
  int block_x_ = 0, block_y_ = 0, factor_x_ = 1, factor_y_ = 1;
  // the image's coefficient arrays in the C ABI's layout, and the frame they have
  int FrameOf(const OutputImage& img) const;
  void GatherCoeffs(const OutputImage& img, std::vector<int16_t>* out) const;
  long compare_calls_ = 0, batched_search_calls_ = 0;
  mutable long compare_block_calls_ = 0;
};

}  // namespace guetzli

#endif  // GUETZLI_HIP_COMPARATOR_H_
