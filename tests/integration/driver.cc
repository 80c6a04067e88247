// TEST INFRASTRUCTURE: guetzli::Process(params, stats, rgb, w, h, &out) (guetzli/processor.cc:
// 926-948) with its `new ButteraugliComparator(...)` replaced by HipButteraugliComparator --
// the one-line change of INTEGRATION.md -- behind an extern "C" entry for the test-suite.
// Everything else is the reference's own code, compiled where it lies.
#include <stdint.h>
#include <string.h>

#include <memory>
#include <string>
#include <vector>

#include "guetzli/jpeg_data.h"
#include "guetzli/jpeg_data_encoder.h"
#include "guetzli/processor.h"
#include "guetzli/stats.h"
#include "hip_comparator.h"

static long run(const uint8_t* rgb, int w, int h, float butteraugli_target, int device, int force_420,
                int try_420, uint8_t* out, long cap, char* trace, long trace_cap, long* calls) {
  std::vector<uint8_t> v(rgb, rgb + (size_t)3 * w * h);
  guetzli::Params params;
  params.butteraugli_target = butteraugli_target;
  params.force_420 = force_420 != 0;
  params.try_420 = try_420 != 0;
  guetzli::ProcessStats stats;
  std::string dbg;
  if (trace) stats.debug_output = &dbg;
  guetzli::JPEGData jpg;
  if (!guetzli::EncodeRGBToJpeg(v, w, h, &jpg)) return -1;
  guetzli::GuetzliOutput result;
  std::unique_ptr<guetzli::HipButteraugliComparator> comparator;
  if (jpg.width >= 32 && jpg.height >= 32) {
    comparator.reset(new guetzli::HipButteraugliComparator(jpg.width, jpg.height, &v,
                                                           params.butteraugli_target, &stats, device));
    if (!comparator->ok()) return -2;
  }
  if (!guetzli::ProcessJpegData(params, jpg, comparator.get(), &result, &stats)) return -1;
  if (calls && comparator) {
    calls[0] = comparator->compare_calls();
    calls[1] = comparator->compare_block_calls();
    calls[2] = comparator->batched_search_calls();
  }
  const std::string& s = result.jpeg_data;
  if ((long)s.size() <= cap) memcpy(out, s.data(), s.size());
  if (trace && trace_cap > 0) {
    const size_t n = dbg.size() < (size_t)trace_cap - 1 ? dbg.size() : (size_t)trace_cap - 1;
    memcpy(trace, dbg.data(), n);
    trace[n] = 0;
  }
  return (long)s.size();
}

// calls: [Compare, CompareBlock, batched phase-A] call counts of the comparator
extern "C" long gzi_process(const uint8_t* rgb, int w, int h, float butteraugli_target, int device,
                            uint8_t* out, long cap, char* trace, long trace_cap, long* calls) {
  return run(rgb, w, h, butteraugli_target, device, 0, 0, out, cap, trace, trace_cap, calls);
}
// the same with Params::force_420 / try_420 (processor.h:36-37)
extern "C" long gzi_process_params(const uint8_t* rgb, int w, int h, float butteraugli_target, int device,
                                   int force_420, int try_420, uint8_t* out, long cap, char* trace,
                                   long trace_cap, long* calls) {
  return run(rgb, w, h, butteraugli_target, device, force_420, try_420, out, cap, trace, trace_cap, calls);
}
