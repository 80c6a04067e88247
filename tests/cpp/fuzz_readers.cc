// TEST INFRASTRUCTURE.  Differential mutation fuzzing of the two parsers the product owns
// (guetzli_amd/host/jpeg_reader.cc, png_reader.cc) under AddressSanitizer + UndefinedBehaviorSanitizer,
// with the UNMODIFIED reference as the judge of every verdict (oracle/_ref/libgz_ref.so: guetzli::ReadJpeg;
// libgz_ref_png.so: guetzli.cc's ReadPNG over libpng) -- the surface the reference ships a fuzz entry for
// (fuzz_target.cc:6-29; parser jpeg_data_reader.cc:931-1081).  Built and driven by tests/test_fuzz_readers.py:
//
//   fuzz_readers jpeg|png REF_SO MUTATIONS RNG_SEED SEED_FILE...
//
// Every mutated stream goes through the product's reader (instrumented) and the reference's (dlopen'ed, not
// instrumented): both must refuse, or both accept with identical content (the canonical dump of reader_dump.h /
// the RGB pixels).  Streams whose header announces more than 16 MPix go through the product alone (the
// reference would allocate the announced image: its own fuzz target skips large images for that reason).
// Every stream goes through the product twice over differently scribbled stacks (uninitialised state shows as a
// difference).  Exit code 0 and a one-line summary; 1 with the offending stream written to fuzz_fail.bin on a mismatch; a
// sanitizer report aborts the process.
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../guetzli_amd/host/jpeg_reader.h"
#include "../../guetzli_amd/host/png_reader.h"
#include "../../guetzli_amd/host/reader_dump.h"

typedef std::vector<uint8_t> Bytes;

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) {}
  uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
  uint32_t below(uint32_t n) { return n ? (uint32_t)(next() % n) : 0; }
};

static Bytes read_file(const char* path) {
  Bytes b;
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
  uint8_t buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + n);
  fclose(f);
  return b;
}

// ---------------------------------------------------------------------------------- JPEG --
struct Seg { size_t pos, len; };   // marker position (the 0xFF) and total length incl. marker and length field

// Marker segments from SOI up to (and including) the first SOS header; entropy-coded data is not split.
static std::vector<Seg> jpeg_segments(const Bytes& d) {
  std::vector<Seg> out;
  size_t p = 2;
  while (p + 4 <= d.size() && d[p] == 0xFF) {
    const int m = d[p + 1];
    if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { out.push_back({p, 2}); p += 2; continue; }
    const size_t len = ((size_t)d[p + 2] << 8) | d[p + 3];
    if (len < 2 || p + 2 + len > d.size()) break;
    out.push_back({p, 2 + len});
    p += 2 + len;
    if (m == 0xDA) break;
  }
  return out;
}

static uint64_t jpeg_announced_pixels(const Bytes& d) {
  for (const Seg& s : jpeg_segments(d)) {
    const int m = d[s.pos + 1];
    if ((m == 0xC0 || m == 0xC1 || m == 0xC2) && s.len >= 9) {
      const uint64_t h = ((uint64_t)d[s.pos + 5] << 8) | d[s.pos + 6], w = ((uint64_t)d[s.pos + 7] << 8) | d[s.pos + 8];
      return w * h;
    }
  }
  return 0;
}

static void mutate_bytes(Bytes* d, Rng* r) {   // the format-agnostic part
  if (d->empty()) return;
  switch (r->below(6)) {
    case 0: for (uint32_t k = 1 + r->below(4); k--;) (*d)[r->below((uint32_t)d->size())] ^= (uint8_t)(1u << r->below(8)); break;
    case 1: (*d)[r->below((uint32_t)d->size())] = (uint8_t)(r->below(3) == 0 ? 0xFF : r->below(2) ? 0x00 : r->below(256)); break;
    case 2: d->resize(r->below((uint32_t)d->size() + 1)); break;                                   // truncation
    case 3: {                                                                                        // inserted bytes
      const size_t at = r->below((uint32_t)d->size() + 1);
      Bytes ins(1 + r->below(8));
      for (auto& b : ins) b = (uint8_t)r->below(256);
      d->insert(d->begin() + at, ins.begin(), ins.end());
      break;
    }
    case 4: {                                                                                        // a run removed
      const size_t at = r->below((uint32_t)d->size());
      const size_t n = std::min<size_t>(1 + r->below(16), d->size() - at);
      d->erase(d->begin() + at, d->begin() + at + n);
      break;
    }
    default: {                                                                                       // a run overwritten by another
      const size_t n = 1 + r->below(32);
      if (d->size() > 2 * n) {
        const size_t a = r->below((uint32_t)(d->size() - n)), b = r->below((uint32_t)(d->size() - n));
        memmove(d->data() + a, d->data() + b, n);
      }
    }
  }
}

static Bytes mutate_jpeg(const Bytes& seed, Rng* r) {
  Bytes d = seed;
  const std::vector<Seg> segs = jpeg_segments(d);
  const uint32_t kind = r->below(10);
  if (kind < 4 || segs.empty()) { mutate_bytes(&d, r); return d; }
  const Seg s = segs[r->below((uint32_t)segs.size())];
  switch (kind) {
    case 4: {   // length field edited
      if (s.len >= 4) {
        int len = ((int)d[s.pos + 2] << 8) | d[s.pos + 3];
        len = r->below(3) == 0 ? (int)r->below(65536) : len + (int)r->below(17) - 8;
        d[s.pos + 2] = (uint8_t)(len >> 8); d[s.pos + 3] = (uint8_t)len;
      }
      break;
    }
    case 5: {   // segment duplicated
      Bytes copy(d.begin() + s.pos, d.begin() + s.pos + s.len);
      d.insert(d.begin() + s.pos + s.len, copy.begin(), copy.end());
      break;
    }
    case 6: d.erase(d.begin() + s.pos, d.begin() + s.pos + s.len); break;   // segment dropped
    case 7: {   // two segments swapped
      const Seg t = segs[r->below((uint32_t)segs.size())];
      if (t.pos != s.pos) {
        const Seg a = s.pos < t.pos ? s : t, b = s.pos < t.pos ? t : s;
        Bytes out(d.begin(), d.begin() + a.pos);
        out.insert(out.end(), d.begin() + b.pos, d.begin() + b.pos + b.len);
        out.insert(out.end(), d.begin() + a.pos + a.len, d.begin() + b.pos);
        out.insert(out.end(), d.begin() + a.pos, d.begin() + a.pos + a.len);
        out.insert(out.end(), d.begin() + b.pos + b.len, d.end());
        d.swap(out);
      }
      break;
    }
    case 8: {   // a byte inside the segment's payload (tables, frame and scan headers)
      if (s.len > 4) d[s.pos + 4 + r->below((uint32_t)(s.len - 4))] = (uint8_t)r->below(256);
      break;
    }
    default: {  // the marker byte itself
      d[s.pos + 1] = (uint8_t)(r->below(2) ? 0xC0 + r->below(16) : r->below(256));
    }
  }
  if (r->below(4) == 0) mutate_bytes(&d, r);
  return d;
}

// ----------------------------------------------------------------------------------- PNG --
struct Chunk { size_t pos, len; };   // position of the length field, body length

static std::vector<Chunk> png_chunks(const Bytes& d) {
  std::vector<Chunk> out;
  size_t p = 8;
  while (p + 12 <= d.size()) {
    const size_t n = ((size_t)d[p] << 24) | ((size_t)d[p + 1] << 16) | ((size_t)d[p + 2] << 8) | d[p + 3];
    if (p + 12 + n > d.size()) break;
    out.push_back({p, n});
    p += 12 + n;
  }
  return out;
}
static void png_fix_crc(Bytes* d, const Chunk& c) {
  const uint32_t crc = (uint32_t)crc32(0, d->data() + c.pos + 4, (uInt)(4 + c.len));
  uint8_t* q = d->data() + c.pos + 8 + c.len;
  q[0] = (uint8_t)(crc >> 24); q[1] = (uint8_t)(crc >> 16); q[2] = (uint8_t)(crc >> 8); q[3] = (uint8_t)crc;
}
static uint64_t png_announced_pixels(const Bytes& d) {
  if (d.size() < 33) return 0;
  auto be = [&](size_t p) { return ((uint64_t)d[p] << 24) | ((uint64_t)d[p + 1] << 16) | ((uint64_t)d[p + 2] << 8) | d[p + 3]; };
  return be(16) * be(20);
}

// The image data re-written with damage INSIDE the deflate stream's payload: every IDAT inflated, a few
// bytes of the filtered scanlines changed (filter type bytes included), deflated again into one IDAT.
static bool png_mutate_scanlines(Bytes* d, Rng* r) {
  const std::vector<Chunk> ch = png_chunks(*d);
  Bytes z;
  size_t first = 0, last = 0;
  bool any = false;
  for (const Chunk& c : ch)
    if (!memcmp(d->data() + c.pos + 4, "IDAT", 4)) {
      if (!any) first = c.pos;
      last = c.pos + 12 + c.len;
      any = true;
      z.insert(z.end(), d->begin() + c.pos + 8, d->begin() + c.pos + 8 + c.len);
    }
  if (!any) return false;
  Bytes raw(1 << 22);
  uLongf n = (uLongf)raw.size();
  if (uncompress(raw.data(), &n, z.data(), (uLong)z.size()) != Z_OK || n == 0) return false;
  raw.resize(n);
  for (uint32_t k = 1 + r->below(3); k--;) raw[r->below((uint32_t)raw.size())] = (uint8_t)r->below(256);
  if (r->below(4) == 0) raw.resize(r->below((uint32_t)raw.size() + 1));   // too few scanlines
  Bytes packed(compressBound((uLong)raw.size()));
  uLongf m = (uLongf)packed.size();
  if (compress2(packed.data(), &m, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
  packed.resize(m);
  Bytes chunk(8);
  chunk[0] = (uint8_t)(m >> 24); chunk[1] = (uint8_t)(m >> 16); chunk[2] = (uint8_t)(m >> 8); chunk[3] = (uint8_t)m;
  memcpy(chunk.data() + 4, "IDAT", 4);
  chunk.insert(chunk.end(), packed.begin(), packed.end());
  chunk.resize(chunk.size() + 4);
  Bytes out(d->begin(), d->begin() + first);
  const size_t at = out.size();
  out.insert(out.end(), chunk.begin(), chunk.end());
  out.insert(out.end(), d->begin() + last, d->end());
  d->swap(out);
  png_fix_crc(d, Chunk{at, (size_t)m});
  return true;
}

static Bytes mutate_png(const Bytes& seed, Rng* r) {
  Bytes d = seed;
  const std::vector<Chunk> ch = png_chunks(d);
  const uint32_t kind = r->below(10);
  if (kind < 2 || ch.empty()) { mutate_bytes(&d, r); return d; }
  if (kind < 4 && png_mutate_scanlines(&d, r)) return d;
  const Chunk c = ch[r->below((uint32_t)ch.size())];
  const bool fix = r->below(5) != 0;   // mostly with the CRC recomputed, so that the damage reaches the parser
  switch (kind) {
    case 4: case 5: case 6:   // a byte (or bit) of the chunk's body
      if (c.len) {
        uint8_t& b = d[c.pos + 8 + r->below((uint32_t)c.len)];
        if (r->below(2)) b ^= (uint8_t)(1u << r->below(8)); else b = (uint8_t)r->below(256);
        if (fix) png_fix_crc(&d, c);
      }
      break;
    case 7: {                 // the length field
      uint32_t n = (uint32_t)c.len + r->below(9) - 4;
      if (r->below(4) == 0) n = (uint32_t)r->next();
      d[c.pos] = (uint8_t)(n >> 24); d[c.pos + 1] = (uint8_t)(n >> 16); d[c.pos + 2] = (uint8_t)(n >> 8); d[c.pos + 3] = (uint8_t)n;
      break;
    }
    case 8: {                 // chunk duplicated or dropped
      if (r->below(2)) {
        Bytes copy(d.begin() + c.pos, d.begin() + c.pos + 12 + c.len);
        d.insert(d.begin() + c.pos + 12 + c.len, copy.begin(), copy.end());
      } else {
        d.erase(d.begin() + c.pos, d.begin() + c.pos + 12 + c.len);
      }
      break;
    }
    default: {                // the chunk type
      static const char* kTypes[] = {"IHDR", "PLTE", "IDAT", "IEND", "tRNS", "gAMA", "bKGD", "zzZz", "IDAt"};
      memcpy(d.data() + c.pos + 4, kTypes[r->below(9)], 4);
      if (fix) png_fix_crc(&d, c);
    }
  }
  return d;
}

// The product's readers keep their state in stack objects; ASan / UBSan do not see a read of a member nobody wrote.
// Every stream therefore goes through the product twice, over a stack scribbled with two different patterns: a
// verdict or a content that depends on uninitialised memory differs between the two runs.
__attribute__((noinline)) static void scribble_stack(int pattern) {
  volatile uint8_t pad[96 * 1024];
  for (size_t i = 0; i < sizeof pad; i += 1) pad[i] = (uint8_t)pattern;
}

static bool product_read(bool png, const uint8_t* p, size_t n, std::string* got) {
  got->clear();
  if (png) {
    std::vector<uint8_t> rgb;
    int w = 0, h = 0;
    if (!guetzli_amd::ReadPng(p, n, &w, &h, &rgb)) return false;
    got->assign((const char*)&w, 4);
    got->append((const char*)&h, 4);
    got->append((const char*)rgb.data(), rgb.size());
    return true;
  }
  guetzli_amd::JpegInput jpg;
  if (!guetzli_amd::ReadJpeg(p, n, &jpg)) return false;
  *got = guetzli_amd::DumpJpegInput(jpg);
  return true;
}

// ---------------------------------------------------------------------------------- main --
typedef long (*RefJpegFn)(const uint8_t*, long, uint8_t*, long);
typedef long (*RefPngFn)(const unsigned char*, long, int*, unsigned char*, long);

int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: fuzz_readers jpeg|png REF_SO MUTATIONS RNG_SEED SEED_FILE...\n"); return 2; }
  const bool png = !strcmp(argv[1], "png");
  void* so = dlopen(argv[2], RTLD_NOW | RTLD_LOCAL);
  if (!so) { fprintf(stderr, "dlopen %s: %s\n", argv[2], dlerror()); return 2; }
  RefJpegFn ref_jpeg = png ? nullptr : (RefJpegFn)dlsym(so, "ref_read_jpeg");
  RefPngFn ref_png = png ? (RefPngFn)dlsym(so, "ref_read_png") : nullptr;
  if (!ref_jpeg && !ref_png) { fprintf(stderr, "the reference entry point is missing\n"); return 2; }
  const int total = atoi(argv[3]);
  Rng rng((uint64_t)atoll(argv[4]));
  std::vector<Bytes> seeds;
  for (int i = 5; i < argc; ++i) seeds.push_back(read_file(argv[i]));
  const uint64_t kMaxPixels = 16u << 20;
  Bytes ref_out((size_t)64 << 20);
  int accepted = 0, refused = 0, product_only = 0, mismatches = 0;
  const bool keep_going = getenv("FUZZ_KEEP_GOING") != nullptr;   // (triage: every mismatch to its own file)
  for (int k = -(int)seeds.size(); k < total; ++k) {
    const Bytes& seed = seeds[(size_t)(k < 0 ? -k - 1 : k) % seeds.size()];
    const Bytes d = k < 0 ? seed : (png ? mutate_png(seed, &rng) : mutate_jpeg(seed, &rng));   // (first: the seeds themselves)
    const uint8_t* p = d.empty() ? (const uint8_t*)"" : d.data();
    std::string got, again;
    scribble_stack(0x00);
    const bool got_ok = product_read(png, p, d.size(), &got);
    scribble_stack(0xff);
    const bool again_ok = product_read(png, p, d.size(), &again);
    if (got_ok != again_ok || got != again) {
      fprintf(stderr, "NONDETERMINISTIC at mutation %d: the product's result depends on uninitialised memory\n", k);
      FILE* f = fopen("fuzz_fail.bin", "wb");
      if (f) { fwrite(d.data(), 1, d.size(), f); fclose(f); }
      return 1;
    }
    if ((png ? png_announced_pixels(d) : jpeg_announced_pixels(d)) > kMaxPixels) { ++product_only; continue; }
    bool exp_ok;
    std::string exp;
    if (png) {
      int wh[2] = {0, 0};
      const long n = ref_png(p, (long)d.size(), wh, ref_out.data(), (long)ref_out.size());
      exp_ok = n >= 0;
      if (exp_ok) { exp.assign((const char*)&wh[0], 4); exp.append((const char*)&wh[1], 4); exp.append((const char*)ref_out.data(), (size_t)std::min<long>(n, (long)ref_out.size())); }
    } else {
      const long n = ref_jpeg(p, (long)d.size(), ref_out.data(), (long)ref_out.size());
      exp_ok = n >= 0;
      if (exp_ok) exp.assign((const char*)ref_out.data(), (size_t)std::min<long>(n, (long)ref_out.size()));
    }
    if (k < 0 && !exp_ok) { fprintf(stderr, "seed %d is refused by the reference\n", -k - 1); return 2; }
    if (got_ok != exp_ok || got != exp) {
      fprintf(stderr, "MISMATCH at mutation %d: product %s, reference %s%s\n", k, got_ok ? "accepts" : "refuses",
              exp_ok ? "accepts" : "refuses", got_ok && exp_ok ? " (different content)" : "");
      char name[64];
      snprintf(name, sizeof name, keep_going ? "fuzz_fail_%d.bin" : "fuzz_fail.bin", k);
      FILE* f = fopen(name, "wb");
      if (f) { fwrite(d.data(), 1, d.size(), f); fclose(f); }
      if (!keep_going) return 1;
      ++mismatches;
      continue;
    }
    if (k >= 0) { if (got_ok) ++accepted; else ++refused; }
  }
  printf("%s: %d mutations of %zu seeds: %d accepted with the reference's content, %d refused like the reference, "
         "%d product-only (announce > 16 MPix)\n", argv[1], total, seeds.size(), accepted, refused, product_only);
  if (mismatches) { fprintf(stderr, "%d MISMATCHES\n", mismatches); return 1; }
  return 0;
}
