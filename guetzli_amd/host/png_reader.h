// PNG input: what the reference's command-line front end does before it
// calls guetzli::Process(params, stats, rgb, w, h, &out) -- ReadPNG, guetzli.cc:47-152:
// libpng's png_read_png with PNG_TRANSFORM_PACKING | PNG_TRANSFORM_EXPAND |
// PNG_TRANSFORM_STRIP_16 (1/2/4-bit samples unpacked, palettes -> RGB, grey -> 8 bit, tRNS ->
// alpha, 16 -> 8 bit by dropping the low byte), then grey / grey+alpha / RGB / RGBA rows ->
// packed RGB with alpha blended on black (BlendOnBlack, guetzli.cc:42-44).
//
// libpng is not part of this code base: the container and the filters are written from the
// PNG specification (ISO/IEC 15948: chunk layout and CRC 5, IHDR 11.2.2, PLTE 11.2.3, tRNS
// 11.3.2.1, filtering 9, Adam7 interlace 8.2), the deflate stream is inflated with zlib.
// tests/test_png_reader.py holds it to the reference's own ReadPNG over libpng 1.6.37
// (oracle/ref_png_harness.cc), pixels and accept / reject decisions alike.
// Host code: a deflate stream is serial.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace guetzli_amd {

// false on anything libpng would raise an error for (bad signature / CRC of a critical chunk /
// IHDR values, missing PLTE, truncated or corrupt image data, missing IEND, dimensions above
// libpng's default user limit of 1 000 000); *error (optional) gets a short description.
bool ReadPng(const uint8_t* data, size_t len, int* xsize, int* ysize, std::vector<uint8_t>* rgb,
             std::string* error = nullptr);

}  // namespace guetzli_amd
