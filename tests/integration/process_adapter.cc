// REFERENCE-SIDE BINDING: what a maintainer of google/guetzli adds so
// that the UNMODIFIED front end guetzli/guetzli.cc (main(), :232-326) encodes on the MI355X.
//
// It is compiled INSTEAD of guetzli/processor.cc and defines the reference's two public entry
// points with the reference's own declarations (guetzli/processor.h:39-41,54-56) by forwarding to
// the host search driver of this repository (guetzli_amd/host/processor.h, libguetzli_amd_host.so):
// same Params fields, same bool + stderr error behaviour, ProcessStats::counters /
// debug_output / debug_output_file / filename carried both ways, so that `--verbose` prints the
// reference's trace.  Test infrastructure: tests/integration/Makefile builds `guetzli_hip` from
// it; nothing under guetzli_amd/ depends on it.
#include "guetzli/processor.h"            // the reference's header, where it lies

#include "guetzli_amd/host/processor.h"   // this repository's driver

namespace guetzli {
namespace {

guetzli_amd::Params Convert(const Params& p) {
  guetzli_amd::Params q;
  q.butteraugli_target = p.butteraugli_target;
  q.clear_metadata = p.clear_metadata;
  q.try_420 = p.try_420;
  q.force_420 = p.force_420;
  q.use_silver_screen = p.use_silver_screen;
  q.zeroing_greedy_lookahead = p.zeroing_greedy_lookahead;
  q.new_zeroing_model = p.new_zeroing_model;
  if (const char* d = getenv("GUETZLI_HIP_DEVICE")) q.device = atoi(d);   // which GPU (default 0)
  return q;
}

template <typename Call>
bool Forward(ProcessStats* stats, Call call) {
  guetzli_amd::ProcessStats s;
  if (stats) {
    s.counters = stats->counters;
    s.debug_output = stats->debug_output;
    s.debug_output_file = stats->debug_output_file;
    s.filename = stats->filename;
  }
  const bool ok = call(&s);
  if (stats) stats->counters = s.counters;
  return ok;
}

}  // namespace

bool Process(const Params& params, ProcessStats* stats, const std::vector<uint8_t>& rgb, int w, int h,
             std::string* out) {
  return Forward(stats, [&](guetzli_amd::ProcessStats* s) {
    return guetzli_amd::Process(Convert(params), s, rgb, w, h, out);
  });
}

bool Process(const Params& params, ProcessStats* stats, const std::string& in_data, std::string* out_data) {
  return Forward(stats, [&](guetzli_amd::ProcessStats* s) {
    return guetzli_amd::Process(Convert(params), s, in_data, out_data);
  });
}

}  // namespace guetzli
