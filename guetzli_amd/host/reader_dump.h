// ReadJpeg's result as a canonical byte string (test hook shared by gzh_read_jpeg and tests/cpp/fuzz_readers.cc;
// oracle/ref_harness.cc writes the same format from the reference's JPEGData): int32 w, h, ncomp; per component
// id, h_samp, v_samp, quant_idx, width_in_blocks, height_in_blocks; int32 nquant; per table index, precision, 64
// values; int32 napp; per entry int32 size + bytes; int32 ncom; likewise; int32 tail size + bytes; then the int16
// coefficients of every component.
#pragma once
#include <stdint.h>

#include <string>

#include "jpeg_reader.h"

namespace guetzli_amd {

inline std::string DumpJpegInput(const JpegInput& jpg) {
  std::string d;
  auto put32 = [&](int32_t v) { d.append((const char*)&v, 4); };
  auto puts = [&](const std::string& s) { put32((int32_t)s.size()); d.append(s); };
  put32(jpg.width); put32(jpg.height); put32((int32_t)jpg.components.size());
  for (const auto& c : jpg.components) {
    put32(c.id); put32(c.h_samp); put32(c.v_samp); put32(c.quant_idx);
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
  return d;
}

}  // namespace guetzli_amd
