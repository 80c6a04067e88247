// JPEG input: baseline / extended-sequential / progressive Huffman JPEG
// -> quantised DCT coefficients + tables + metadata, what guetzli::ReadJpeg(data,
// JPEG_READ_ALL, &jpg) (jpeg_data_reader.cc:931-1073) leaves in JPEGData for
// guetzli::Process(params, stats, jpeg_data, &out) (processor.cc:890-924).
//
// Written from ITU-T T.81: marker parsing (B.2), Huffman table construction (C),
// sequential decoding (F.2.2) and progressive decoding with successive approximation
// (G.1.2), restart intervals (E.2.4).  Host code: parsing an entropy-coded stream is serial.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace guetzli_amd {

struct JpegQuant {
  int values[64];   // natural (row-major) order
  int precision;    // 0: 8 bit, 1: 16 bit
  int index;        // Tq, 0..3
};

struct JpegComponentIn {
  int id = 0;
  int h_samp = 1, v_samp = 1;
  int quant_idx = 0;                 // position in JpegInput::quant (after FixupIndexes, :890-909)
  int width_in_blocks = 0, height_in_blocks = 0;
  std::vector<int16_t> coeffs;       // [height_in_blocks][width_in_blocks][64], quantised
};

struct JpegInput {
  int width = 0, height = 0;
  int max_h_samp = 1, max_v_samp = 1;
  int mcu_rows = 0, mcu_cols = 0;
  int restart_interval = 0;
  bool progressive = false;
  std::vector<JpegComponentIn> components;
  std::vector<JpegQuant> quant;         // every DQT table in file order (:346-375)
  std::vector<std::string> app_data;    // marker byte + segment (length included), :396-408
  std::vector<std::string> com_data;    // segment with its length bytes, :411-422
  std::string tail_data;                // bytes after EOI (:1046-1049)

  bool Is444() const;                   // jpeg_data.cc:36-46
  bool Is420() const;                   // jpeg_data.cc:24-34
};

// false on anything the stream does not allow; *error (optional) gets a short description.
bool ReadJpeg(const uint8_t* data, size_t len, JpegInput* jpg, std::string* error = nullptr);

}  // namespace guetzli_amd
