#include "png_reader.h"

#include <stdlib.h>
#include <string.h>
#include <zlib.h>

namespace guetzli_amd {
namespace {

inline uint32_t Be32(const uint8_t* p) {
  return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}
inline uint32_t Be16(const uint8_t* p) { return ((uint32_t)p[0] << 8) | p[1]; }

inline uint8_t BlendOnBlack(uint8_t val, uint8_t alpha) {   // guetzli.cc:42-44
  return (uint8_t)(((int)val * (int)alpha + 128) / 255);
}

struct Header {
  uint32_t width = 0, height = 0;
  int bit_depth = 0, color_type = 0, interlace = 0;
  int channels = 0;          // samples per pixel in the file
  int bits_per_pixel = 0;
};

struct Transparency {
  bool present = false;
  uint8_t alpha[256];        // palette images
  int num = 0;
  uint32_t gray = 0, red = 0, green = 0, blue = 0;   // 16-bit values as stored
};

bool Fail(std::string* error, const char* what) {
  if (error) *error = what;
  return false;
}

// Filter type 4 predictor (ISO/IEC 15948 9.4).
inline int Paeth(int a, int b, int c) {
  const int p = a + b - c;
  const int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  if (pa <= pb && pa <= pc) return a;
  return pb <= pc ? b : c;
}

// Reverses the filter of one scanline in place; prev = the unfiltered previous scanline of the
// same pass (zeros for the first one).  bpp = bytes per complete pixel, at least 1.
bool Unfilter(int type, uint8_t* row, const uint8_t* prev, size_t n, size_t bpp) {
  switch (type) {
    case 0:
      return true;
    case 1:
      for (size_t i = bpp; i < n; ++i) row[i] = (uint8_t)(row[i] + row[i - bpp]);
      return true;
    case 2:
      for (size_t i = 0; i < n; ++i) row[i] = (uint8_t)(row[i] + prev[i]);
      return true;
    case 3:
      for (size_t i = 0; i < n; ++i) {
        const int left = i >= bpp ? row[i - bpp] : 0;
        row[i] = (uint8_t)(row[i] + ((left + prev[i]) >> 1));
      }
      return true;
    case 4:
      for (size_t i = 0; i < n; ++i) {
        const int left = i >= bpp ? row[i - bpp] : 0;
        const int up_left = i >= bpp ? prev[i - bpp] : 0;
        row[i] = (uint8_t)(row[i] + Paeth(left, prev[i], up_left));
      }
      return true;
    default:
      return false;
  }
}

// Sample s (0-based, in file order) of a scanline with `depth` bits per sample, as stored.
inline uint32_t Sample(const uint8_t* row, size_t s, int depth) {
  switch (depth) {
    case 8: return row[s];
    case 16: return Be16(row + 2 * s);
    case 4: return (row[s >> 1] >> (4 * (1 - (s & 1)))) & 0x0f;
    case 2: return (row[s >> 2] >> (2 * (3 - (s & 3)))) & 0x03;
    default: return (row[s >> 3] >> (7 - (s & 7))) & 0x01;
  }
}

// One decoded scanline -> packed RGB at (x0 + i * dx, y): unpacking, palette / grey expansion,
// tRNS -> alpha, 16 -> 8 bits (high byte), blend on black.
void EmitRow(const Header& hd, const uint8_t* palette, const Transparency& tr, const uint8_t* row,
             uint32_t pass_width, uint32_t x0, uint32_t dx, uint8_t* out_row) {
  const int depth = hd.bit_depth;
  for (uint32_t i = 0; i < pass_width; ++i) {
    uint8_t r, g, b, a = 255;
    switch (hd.color_type) {
      case 3: {
        const uint32_t idx = Sample(row, i, depth);
        r = palette[3 * idx]; g = palette[3 * idx + 1]; b = palette[3 * idx + 2];
        if (tr.present) a = tr.alpha[idx];
        break;
      }
      case 0:
      case 4: {
        const size_t s = hd.color_type == 4 ? 2 * (size_t)i : i;
        const uint32_t v = Sample(row, s, depth);
        uint8_t v8;
        uint32_t key = tr.gray;
        switch (depth) {   // png_do_expand: low-depth grey (and its tRNS key) scaled to 8 bits
          case 1: v8 = (uint8_t)(v * 0xff); key = (key & 0x01) * 0xff; break;
          case 2: v8 = (uint8_t)(v * 0x55); key = (key & 0x03) * 0x55; break;
          case 4: v8 = (uint8_t)(v * 0x11); key = (key & 0x0f) * 0x11; break;
          case 8: v8 = (uint8_t)v; key &= 0xff; break;
          default: v8 = (uint8_t)(v >> 8); break;   // 16 bits: compared in full, then chopped
        }
        r = g = b = v8;
        if (hd.color_type == 4) {
          const uint32_t av = Sample(row, s + 1, depth);
          a = (uint8_t)(depth == 16 ? av >> 8 : av);
        } else if (tr.present) {
          a = (depth == 16 ? v == (key & 0xffff) : v8 == key) ? 0 : 255;
        }
        break;
      }
      default: {   // 2: RGB, 6: RGBA
        const size_t s = (size_t)i * (hd.color_type == 6 ? 4 : 3);
        const uint32_t rv = Sample(row, s, depth), gv = Sample(row, s + 1, depth),
                       bv = Sample(row, s + 2, depth);
        if (depth == 16) { r = (uint8_t)(rv >> 8); g = (uint8_t)(gv >> 8); b = (uint8_t)(bv >> 8); }
        else { r = (uint8_t)rv; g = (uint8_t)gv; b = (uint8_t)bv; }
        if (hd.color_type == 6) {
          const uint32_t av = Sample(row, s + 3, depth);
          a = (uint8_t)(depth == 16 ? av >> 8 : av);
        } else if (tr.present) {
          const bool hit = depth == 16
              ? (rv == (tr.red & 0xffff) && gv == (tr.green & 0xffff) && bv == (tr.blue & 0xffff))
              : (rv == (tr.red & 0xff) && gv == (tr.green & 0xff) && bv == (tr.blue & 0xff));
          a = hit ? 0 : 255;
        }
        break;
      }
    }
    uint8_t* o = out_row + 3 * ((size_t)x0 + (size_t)i * dx);
    const bool has_alpha = hd.color_type == 4 || hd.color_type == 6 || tr.present;
    if (has_alpha) {
      o[0] = BlendOnBlack(r, a); o[1] = BlendOnBlack(g, a); o[2] = BlendOnBlack(b, a);
    } else {
      o[0] = r; o[1] = g; o[2] = b;
    }
  }
}

}  // namespace

bool ReadPng(const uint8_t* data, size_t len, int* xsize, int* ysize, std::vector<uint8_t>* rgb,
             std::string* error) {
  static const uint8_t kSignature[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (!data || len < 8 || memcmp(data, kSignature, 8) != 0) return Fail(error, "not a PNG file");
  Header hd;
  Transparency tr;
  uint8_t palette[3 * 256];
  memset(palette, 0, sizeof(palette));   // libpng's palette is zero-filled up to 256 entries
  memset(tr.alpha, 0xff, sizeof(tr.alpha));
  bool have_ihdr = false, have_plte = false, have_idat = false, have_iend = false;
  // The image data = the FIRST run of consecutive IDAT chunks (libpng reads rows from it and stops at the first
  // chunk of another type; IDAT chunks behind that are skipped with a warning), kept as the chunks they are.
  std::vector<std::pair<const uint8_t*, size_t> > idat;
  size_t idat_bytes = 0;
  bool idat_run_over = false;
  size_t pos = 8;
  while (!have_iend) {
    if (len - pos < 12) return Fail(error, "unexpected end of data");
    const uint32_t clen = Be32(data + pos);
    const uint8_t* type = data + pos + 4;
    if (clen > 0x7fffffffu) return Fail(error, "chunk length out of range");
    if (len - pos - 12 < clen) return Fail(error, "unexpected end of data");
    const uint8_t* body = data + pos + 8;
    for (int i = 0; i < 4; ++i) {
      const uint8_t ch = type[i];
      if (!((ch >= 'A' && ch <= 'Z') || (ch >= 'a' && ch <= 'z'))) return Fail(error, "invalid chunk type");
    }
    const bool critical = (type[0] & 0x20) == 0;
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), type, 4 + clen);
    const bool crc_ok = crc == Be32(body + clen);
    pos += 12 + (size_t)clen;
    if (have_idat && memcmp(type, "IDAT", 4) != 0) idat_run_over = true;
    if (!crc_ok) {
      if (critical) return Fail(error, "CRC error in a critical chunk");
      continue;   // ancillary: discarded with a warning
    }
    if (!have_ihdr && memcmp(type, "IHDR", 4) != 0) return Fail(error, "missing IHDR");
    if (memcmp(type, "IHDR", 4) == 0) {
      if (have_ihdr) return Fail(error, "duplicate IHDR");
      if (clen != 13) return Fail(error, "invalid IHDR length");
      hd.width = Be32(body);
      hd.height = Be32(body + 4);
      hd.bit_depth = body[8];
      hd.color_type = body[9];
      hd.interlace = body[12];
      if (hd.width == 0 || hd.height == 0 || hd.width > 1000000u || hd.height > 1000000u)
        return Fail(error, "image dimensions out of range");
      if (body[10] != 0 || body[11] != 0 || hd.interlace > 1) return Fail(error, "invalid IHDR method fields");
      const int d = hd.bit_depth;
      bool ok = false;
      switch (hd.color_type) {
        case 0: ok = d == 1 || d == 2 || d == 4 || d == 8 || d == 16; hd.channels = 1; break;
        case 2: ok = d == 8 || d == 16; hd.channels = 3; break;
        case 3: ok = d == 1 || d == 2 || d == 4 || d == 8; hd.channels = 1; break;
        case 4: ok = d == 8 || d == 16; hd.channels = 2; break;
        case 6: ok = d == 8 || d == 16; hd.channels = 4; break;
        default: break;
      }
      if (!ok) return Fail(error, "invalid colour type / bit depth combination");
      hd.bits_per_pixel = hd.channels * d;
      have_ihdr = true;
    } else if (memcmp(type, "PLTE", 4) == 0) {
      if (have_plte) return Fail(error, "duplicate PLTE");
      if (have_idat) continue;   // out of place: ignored (the missing palette was refused at the first IDAT)
      if (hd.color_type == 0 || hd.color_type == 4) continue;   // ignored in greyscale PNGs
      if (clen > 3 * 256 || clen % 3 != 0) {
        if (hd.color_type == 3) return Fail(error, "invalid palette");
        continue;
      }
      size_t n = clen / 3;
      if (hd.color_type == 3 && n > ((size_t)1 << hd.bit_depth)) n = (size_t)1 << hd.bit_depth;
      memcpy(palette, body, 3 * n);
      have_plte = true;
    } else if (memcmp(type, "tRNS", 4) == 0) {
      if (have_idat || tr.present) continue;   // out of place / duplicate: ignored
      if (hd.color_type == 3) {
        if (!have_plte || clen > 256 || clen == 0) continue;
        memcpy(tr.alpha, body, clen);
        tr.num = (int)clen;
        tr.present = true;
      } else if (hd.color_type == 0) {
        if (clen != 2) continue;
        tr.gray = Be16(body);
        tr.present = true;
      } else if (hd.color_type == 2) {
        if (clen != 6) continue;
        tr.red = Be16(body); tr.green = Be16(body + 2); tr.blue = Be16(body + 4);
        tr.present = true;
      }   // not allowed with an alpha channel: ignored
    } else if (memcmp(type, "IDAT", 4) == 0) {
      if (hd.color_type == 3 && !have_plte) return Fail(error, "missing PLTE before IDAT");
      if (!idat_run_over) {
        idat.push_back(std::make_pair(body, (size_t)clen));
        idat_bytes += clen;
      }
      have_idat = true;
    } else if (memcmp(type, "IEND", 4) == 0) {
      have_iend = true;
    } else if (critical) {
      return Fail(error, "unknown critical chunk");
    }
  }
  if (!have_idat) return Fail(error, "missing IDAT");

  // ---- geometry of the (up to seven) passes ----
  static const int kX0[7] = {0, 4, 0, 2, 0, 1, 0}, kY0[7] = {0, 0, 4, 0, 2, 0, 1};
  static const int kDx[7] = {8, 8, 4, 4, 2, 2, 1}, kDy[7] = {8, 8, 8, 4, 4, 2, 2};
  const int npass = hd.interlace ? 7 : 1;
  uint32_t pw[7], ph[7];
  size_t total = 0, max_rowbytes = 0;
  for (int p = 0; p < npass; ++p) {
    if (hd.interlace) {
      pw[p] = (hd.width + kDx[p] - 1 - kX0[p]) / kDx[p];
      ph[p] = (hd.height + kDy[p] - 1 - kY0[p]) / kDy[p];
    } else {
      pw[p] = hd.width;
      ph[p] = hd.height;
    }
    if (pw[p] == 0 || ph[p] == 0) continue;
    const size_t rowbytes = ((size_t)pw[p] * hd.bits_per_pixel + 7) / 8;
    if (rowbytes > max_rowbytes) max_rowbytes = rowbytes;
    total += (rowbytes + 1) * ph[p];
  }

  // ---- inflate ----
  // deflate expands by at most ~1032:1 (a stored/RLE limit of the format), so a header that
  // declares more samples than the IDAT data can possibly inflate to is rejected before
  // anything of that size is allocated (a 60-byte file declaring 10^6 x 10^6 RGBA16 would
  // otherwise ask for 8 TB here); 3 * width * height <= 24 * total is bounded with it.
  if (total / 1040 > idat_bytes + 64) return Fail(error, "not enough image data");
  std::vector<uint8_t> raw(total);
  {
    // WHICH damaged streams still count as an image is libpng's call (png_read_IDAT_data / png_read_finish_IDAT,
    // pngrutil.c), and it depends on how libpng feeds zlib: input in pieces of at most 8192 bytes (PNG_ZBUF_SIZE)
    // that never span two IDAT chunks, output one scanline (filter byte + row) at a time; a scanline that cannot be
    // completed is an error; behind the last scanline the rest of the stream is inflated into a 1024-byte scratch
    // buffer until it ends -- running out of IDAT data before the stream's end is an error there too ("Not
    // enough image data"), unless nothing at all came out of that last step, while damage in that part only
    // warns.  Restated here call for call (tests/test_fuzz_readers.py holds it to libpng's verdicts).
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, 0) != Z_OK) return Fail(error, "zlib initialisation failed");   // window size: the stream's own
    size_t chunk = 0, left = idat.empty() ? 0 : idat[0].second;
    const uint8_t* next = idat.empty() ? nullptr : idat[0].first;
    bool first_byte = true, ended = false;
    auto refill = [&]() -> bool {   // more input: false when the run of IDAT chunks is used up
      while (left == 0) {
        if (++chunk >= idat.size()) return false;
        next = idat[chunk].first;
        left = idat[chunk].second;
      }
      const size_t n = left < 8192 ? left : 8192;
      zs.next_in = const_cast<Bytef*>(next);
      zs.avail_in = (uInt)n;
      next += n;
      left -= n;
      return true;
    };
    auto step = [&]() -> int {      // one inflate call as libpng makes it (png_zlib_inflate: the CMF byte's window bits)
      if (first_byte && zs.avail_in > 0) {
        if ((zs.next_in[0] >> 4) > 7) return Z_DATA_ERROR;
        first_byte = false;
      }
      return inflate(&zs, Z_NO_FLUSH);
    };
    const char* problem = nullptr;
    size_t out_pos = 0;
    for (int p = 0; p < npass && !problem; ++p) {
      if (pw[p] == 0 || ph[p] == 0) continue;
      const size_t rowbytes = ((size_t)pw[p] * hd.bits_per_pixel + 7) / 8;
      for (uint32_t j = 0; j < ph[p] && !problem; ++j) {
        size_t want = rowbytes + 1;
        zs.next_out = raw.data() + out_pos;
        out_pos += want;
        if (ended) { problem = "not enough image data"; break; }
        while (want > 0) {
          if (zs.avail_in == 0 && !refill()) { problem = "not enough image data"; break; }
          const uInt out = want > 0x7fffffffu ? 0x7fffffffu : (uInt)want;
          zs.avail_out = out;
          const int zrc = step();
          want -= out - zs.avail_out;
          if (zrc == Z_STREAM_END) { ended = true; break; }
          if (zrc != Z_OK) { problem = "corrupt compressed image data"; break; }
        }
        if (!problem && want > 0) problem = "not enough image data";
      }
    }
    if (!problem && !ended) {   // the rest of the stream (png_read_finish_IDAT)
      uint8_t scratch[1024];
      size_t extra = 0;
      do {
        if (zs.avail_in == 0 && !refill()) { problem = "not enough image data"; break; }
        zs.next_out = scratch;
        zs.avail_out = (uInt)sizeof(scratch);
        const int zrc = step();
        extra += sizeof(scratch) - zs.avail_out;
        if (zrc != Z_OK) break;   // the end, or damage behind the image: a warning at most
      } while (extra > 0);
    }
    inflateEnd(&zs);
    if (problem) return Fail(error, problem);
  }

  // ---- unfilter + convert ----
  const size_t npix = (size_t)hd.width * hd.height;
  rgb->assign(3 * npix, 0);
  const size_t bpp = hd.bits_per_pixel >= 8 ? (size_t)hd.bits_per_pixel / 8 : 1;
  std::vector<uint8_t> prev(max_rowbytes);
  size_t at = 0;
  for (int p = 0; p < npass; ++p) {
    if (pw[p] == 0 || ph[p] == 0) continue;
    const size_t rowbytes = ((size_t)pw[p] * hd.bits_per_pixel + 7) / 8;
    memset(prev.data(), 0, rowbytes);
    for (uint32_t j = 0; j < ph[p]; ++j) {
      const int filter = raw[at];
      uint8_t* row = &raw[at + 1];
      if (!Unfilter(filter, row, prev.data(), rowbytes, bpp)) return Fail(error, "bad adaptive filter value");
      const uint32_t y = hd.interlace ? (uint32_t)kY0[p] + j * (uint32_t)kDy[p] : j;
      EmitRow(hd, palette, tr, row, pw[p], hd.interlace ? (uint32_t)kX0[p] : 0u,
              hd.interlace ? (uint32_t)kDx[p] : 1u, rgb->data() + 3 * (size_t)y * hd.width);
      memcpy(prev.data(), row, rowbytes);
      at += rowbytes + 1;
    }
  }
  *xsize = (int)hd.width;
  *ysize = (int)hd.height;
  return true;
}

}  // namespace guetzli_amd
