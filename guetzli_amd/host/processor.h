// Host search driver over the C ABI of include/guetzli_amd.h: the drop-in for
// guetzli::Process(params, stats, rgb, w, h, &out) (processor.h:54-56,
// processor.cc:926-948) with all per-pixel / per-block numeric work on the MI355X.
//
// Same names, argument meaning and error behaviour as the reference API: bool return,
// diagnostics on stderr, no exceptions; `ProcessStats::debug_output` receives the same
// --verbose trace text (one line per candidate evaluation), which is what the parity
// tests compare first.
//
// What stays on the host is the serial decision logic of the reference (quant-matrix
// bisection, global coefficient ordering, entropy-size model) and the JPEG writer; it is
// written from scratch around flat coefficient arrays.  Scope: everything guetzli::Process
// accepts -- RGB input (the BASELINE configurations), YUV 4:4:4 and 4:2:0 JPEG input, every
// field of Params (try_420 / force_420 / use_silver_screen included).
#pragma once
#include <stdint.h>
#include <stdio.h>

#include <map>
#include <string>
#include <vector>

namespace guetzli_amd {

struct Params {                       // guetzli::Params, processor.h:29-37
  float butteraugli_target = 1.0;
  bool clear_metadata = true;
  bool try_420 = false;
  bool force_420 = false;
  bool use_silver_screen = false;
  int zeroing_greedy_lookahead = 3;
  bool new_zeroing_model = true;
  int device = 0;                     // HIP device ordinal (not in the reference)
};

static const char* const kNumItersCnt = "number of iterations";
static const char* const kNumItersUpCnt = "number of iterations up";
static const char* const kNumItersDownCnt = "number of iterations down";

struct ProcessStats {                 // guetzli::ProcessStats, stats.h:34-41
  std::map<std::string, int> counters;
  std::string* debug_output = nullptr;
  FILE* debug_output_file = nullptr;
  std::string filename;
  // wall-clock breakdown in seconds, filled by Process (not in the reference)
  std::map<std::string, double> timers;
};

double ButteraugliScoreForQuality(double quality);                       // quality.cc:78-85
double ScoreJPEG(double butteraugli_distance, int size, double target);  // score.cc:23-41

// Sets *out to a JPEG that decodes to an image visually indistinguishable from rgb
// (packed 8-bit sRGB, w*h*3).  Returns false (message on stderr) on failure.
bool Process(const Params& params, ProcessStats* stats, const std::vector<uint8_t>& rgb,
             int w, int h, std::string* out);

// The same from an existing JPEG (guetzli::Process(params, stats, jpeg_data, &out),
// processor.h:39-41): the input is parsed on the host (jpeg_reader.h), its coefficients become
// the original, its decoded pixels the comparator's reference image, its quantisation the
// first candidate; Params::clear_metadata decides whether APPn / COM / trailing bytes are
// carried over.  YUV 4:4:4 and 4:2:0 input (other sampling factors are refused, as the reference
// refuses them, processor.cc:811-823).
bool Process(const Params& params, ProcessStats* stats, const std::string& jpeg_data,
             std::string* out);

}  // namespace guetzli_amd
