#include "jpeg_reader.h"

#include <string.h>

#include <algorithm>

namespace guetzli_amd {

bool JpegInput::Is444() const {
  if (components.size() != 3 || max_h_samp != 1 || max_v_samp != 1) return false;
  for (int c = 0; c < 3; ++c)
    if (components[c].h_samp != 1 || components[c].v_samp != 1) return false;
  return true;
}

bool JpegInput::Is420() const {
  return components.size() == 3 && max_h_samp == 2 && max_v_samp == 2 &&
         components[0].h_samp == 2 && components[0].v_samp == 2 && components[1].h_samp == 1 &&
         components[1].v_samp == 1 && components[2].h_samp == 1 && components[2].v_samp == 1;
}

namespace {

const int kZigZagToNatural[64 + 16] = {
  0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
  35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
  58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
  63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63,   // overrun guard
};

struct Fail {
  std::string* sink;
  bool operator()(const char* what) const {
    if (sink) *sink = what;
    return false;
  }
};

// WHICH streams are accepted follows the reference's reader decision by decision (jpeg_data_reader.cc; cited per
// rule below), not T.81 alone: Process(jpeg) must refuse exactly what guetzli::Process refuses and read the same
// coefficients from every oddity it tolerates (tests/test_fuzz_readers.py: mutated streams, the reference as judge).

// Canonical Huffman decoding table (T.81 C.2, F.2.2.3) with a 9-bit first-level lookup.  A bit pattern that is
// no symbol's code decodes to -1: the reference appends one invalid symbol behind the last code of the longest
// length and leaves every other unassigned pattern invalid too (ProcessDHT :287-330), which is the same set.
struct HuffTable {
  bool seen = false;   // some DHT named this slot (ProcessSOS's "table found", :229-252)
  int maxcode[18];     // largest code of each length, -1 if none
  int valptr[17];
  int mincode[17];
  uint8_t values[256];
  int num_values = 0;
  uint16_t fast[512];  // ((length) << 8 | symbol) + 1 for codes of <= 9 bits, 0 otherwise

  // A slot no DHT has filled decodes nothing: a sequential scan whose header names a band without DC passes the
  // "table defined" check (it goes by the header's band, ProcessSOS :229-252) and then meets the reference's
  // never-built lookup table, whose every entry is the invalid symbol (found by the 2 M-mutation campaign).
  HuffTable() {
    memset(fast, 0, sizeof(fast));
    memset(values, 0, sizeof(values));
    for (int l = 0; l < 18; ++l) maxcode[l] = -1;
    for (int l = 0; l < 17; ++l) { valptr[l] = 0; mincode[l] = 0; }
    maxcode[17] = 0x7fffffff;
  }

  void Build(const int* counts /*[1..16]*/, const uint8_t* vals, int n) {
    num_values = n;
    memcpy(values, vals, (size_t)n);
    memset(fast, 0, sizeof(fast));
    int code = 0, k = 0;
    for (int len = 1; len <= 16; ++len) {
      valptr[len] = k;
      mincode[len] = code;
      for (int i = 0; i < counts[len]; ++i, ++k, ++code) {
        if (len <= 9) {
          const int first = code << (9 - len), count = 1 << (9 - len);
          for (int j = 0; j < count; ++j) fast[first + j] = (uint16_t)(((len << 8) | vals[k]) + 1);
        }
      }
      maxcode[len] = counts[len] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
  }
};

// MSB-first bit reader over an entropy-coded segment with the reference's view of where it ends
// (BitReaderState, :431-505): the data stop at the first 0xff that is not followed by 0x00 -- or two bytes
// before the end of the file, whatever stands there -- and a reader that runs on gets that 0xff and then zero
// bytes; Finish() gives whole unread bytes back and tells whether a bit beyond the stop was consumed.
struct BitReader {
  const uint8_t* data;
  size_t len, pos = 0, stop = 0;
  uint64_t acc = 0;
  int bits = 0;

  BitReader(const uint8_t* d, size_t l, size_t p) : data(d), len(l) { Reset(p); }
  void Reset(size_t p) {
    pos = p;
    acc = 0;
    bits = 0;
    stop = len - 2;
  }
  uint8_t NextByte() {
    if (pos >= stop) {
      ++pos;
      return 0;
    }
    const uint8_t c = data[pos++];
    if (c == 0xff) {
      if (data[pos] == 0) ++pos;   // (pos <= stop <= len - 2: inside the buffer)
      else stop = pos - 1;         // a marker: its 0xff was the last byte handed out
    }
    return c;
  }
  void Fill() {
    while (bits <= 56) {
      acc = (acc << 8) | NextByte();
      bits += 8;
    }
  }
  int Peek(int n) {
    if (bits < n) Fill();
    return (int)((acc >> (bits - n)) & ((1u << n) - 1));
  }
  void Drop(int n) { bits -= n; }
  int Read(int n) {
    if (n == 0) return 0;
    const int v = Peek(n);
    Drop(n);
    return v;
  }
  // FinishStream (:478-497): false when the scan consumed bits the stream does not have.
  bool Finish(size_t* next) {
    for (int whole = bits >> 3; whole > 0; --whole) {
      --pos;
      if (pos < stop && data[pos] == 0 && data[pos - 1] == 0xff) --pos;   // a stuffed pair goes back as one
    }
    bits = 0;
    acc = 0;
    if (pos > stop) return false;
    *next = pos;
    return true;
  }
};

int DecodeSymbol(const HuffTable& t, BitReader* br) {
  const int look = br->Peek(9);
  const uint16_t f = t.fast[look];
  if (f) {
    br->Drop((f - 1) >> 8);
    return (f - 1) & 0xff;
  }
  // codes longer than 9 bits: extend bit by bit (F.2.2.3)
  int code = look, len = 9;
  br->Drop(9);
  while (len <= 16 && (t.maxcode[len] < 0 || code > t.maxcode[len] || code < t.mincode[len])) {
    code = (code << 1) | br->Read(1);
    ++len;
  }
  if (len > 16) return -1;
  const int idx = t.valptr[len] + code - t.mincode[len];
  if (idx < 0 || idx >= t.num_values) return -1;
  return t.values[idx];
}

inline int Extend(int v, int nbits) {   // F.2.2.1 EXTEND
  return v < (1 << (nbits - 1)) ? v - (1 << nbits) + 1 : v;
}
inline int ShiftLeftSigned(int v, int s) { return v >= 0 ? v << s : -((-v) << s); }

struct ScanComp {
  int comp;
  int dc_tbl, ac_tbl;
};

struct Decoder {
  const uint8_t* data;
  size_t len;
  JpegInput* jpg;
  Fail fail;
  HuffTable dc_tables[4], ac_tables[4];
  bool found_sof = false;
  int tables_defined = 0;               // Huffman tables in all DHT segments (:1057-1069)
  uint16_t progression[4][64] = {{0}};  // per component and coefficient: the bit planes the scans so far covered

  uint16_t Be16(size_t p) const { return (uint16_t)((data[p] << 8) | data[p + 1]); }

  // SOF0-2 (ProcessSOF, :86-164).
  bool ProcessSOF(size_t* pos, int marker) {
    if (found_sof) return fail("duplicate SOF");
    if (*pos + 8 > len) return fail("truncated SOF");
    const size_t seg = Be16(*pos);
    size_t p = *pos + 2;
    const int precision = data[p++];
    const int height = Be16(p); p += 2;
    const int width = Be16(p); p += 2;
    const int nc = data[p++];
    if (precision != 8) return fail("unsupported sample precision");
    if (height < 1 || width < 1) return fail("bad dimensions");
    if (nc < 1 || nc > 4) return fail("bad component count");
    if (p + 3 * (size_t)nc > len) return fail("truncated SOF");
    jpg->height = height;
    jpg->width = width;
    jpg->progressive = marker == 0xc2;
    jpg->components.assign(nc, JpegComponentIn());
    for (int i = 0; i < nc; ++i) {
      JpegComponentIn& c = jpg->components[i];
      c.id = data[p++];
      for (int j = 0; j < i; ++j)
        if (jpg->components[j].id == c.id) return fail("duplicate component id");
      const int hv = data[p++];
      c.h_samp = hv >> 4;
      c.v_samp = hv & 15;
      if (c.h_samp < 1 || c.v_samp < 1) return fail("bad sampling factor");
      c.quant_idx = data[p++];   // (a table that does not exist is found out at the end of the stream, as there)
      jpg->max_h_samp = std::max(jpg->max_h_samp, c.h_samp);
      jpg->max_v_samp = std::max(jpg->max_v_samp, c.v_samp);
    }
    jpg->mcu_cols = (jpg->width + 8 * jpg->max_h_samp - 1) / (8 * jpg->max_h_samp);
    jpg->mcu_rows = (jpg->height + 8 * jpg->max_v_samp - 1) / (8 * jpg->max_v_samp);
    for (int i = 0; i < nc; ++i) {
      JpegComponentIn& c = jpg->components[i];
      if (jpg->max_h_samp % c.h_samp != 0 || jpg->max_v_samp % c.v_samp != 0)
        return fail("non-integer subsampling ratio");
      c.width_in_blocks = jpg->mcu_cols * c.h_samp;
      c.height_in_blocks = jpg->mcu_rows * c.v_samp;
      const uint64_t nblocks = (uint64_t)c.width_in_blocks * c.height_in_blocks;
      if (nblocks > (1ull << 21)) return fail("image too large");
      c.coeffs.assign((size_t)nblocks * 64, 0);
    }
    if (*pos + seg != p) return fail("bad SOF length");
    *pos = p;
    found_sof = true;
    return true;
  }

  // DHT (ProcessDHT, :262-340): DC alphabets of at most 12 symbols 0..11, AC of at most 256, no symbol twice,
  // an EMPTY table allowed, and the code lengths must leave room for one more code of the longest length used
  // (a complete code is refused: the all-ones code word is reserved).
  bool ProcessDHT(size_t* pos) {
    if (*pos + 2 > len) return fail("truncated DHT");
    const size_t seg = Be16(*pos);
    if (seg == 2) return fail("empty DHT");
    size_t p = *pos + 2;
    const size_t end = *pos + seg;
    while (p < end) {
      if (p + 17 > len) return fail("truncated DHT table");
      const int slot = data[p++];
      const bool is_ac = (slot & 0x10) != 0;
      const int th = is_ac ? slot - 0x10 : slot;
      if (th < 0 || th > 3) return fail("bad Huffman table id");
      int counts[17] = {0};
      int total = 0, longest = 1;
      long space = 1L << 16;
      for (int l = 1; l <= 16; ++l) {
        counts[l] = data[p++];
        if (counts[l]) longest = l;
        total += counts[l];
        space -= (long)counts[l] << (16 - l);
      }
      if (total > (is_ac ? 256 : 12)) return fail("bad Huffman table size");
      if (p + (size_t)total > len) return fail("truncated DHT table");
      bool used[256] = {false};
      for (int i = 0; i < total; ++i) {
        const int v = data[p + i];
        if (!is_ac && v > 11) return fail("bad DC symbol in a Huffman table");
        if (used[v]) return fail("symbol twice in a Huffman table");
        used[v] = true;
      }
      space -= 1L << (16 - longest);
      if (space < 0) return fail("invalid Huffman code lengths");
      HuffTable& t = is_ac ? ac_tables[th] : dc_tables[th];
      t.Build(counts, data + p, total);
      t.seen = true;
      ++tables_defined;
      p += total;
    }
    if (p != end) return fail("bad DHT length");
    *pos = p;
    return true;
  }

  // DQT (ProcessDQT, :344-377): at most four tables in the whole stream; any non-zero Pq means 16-bit entries.
  bool ProcessDQT(size_t* pos) {
    if (*pos + 2 > len) return fail("truncated DQT");
    const size_t seg = Be16(*pos);
    if (seg == 2) return fail("empty DQT");
    size_t p = *pos + 2;
    const size_t end = *pos + seg;
    while (p < end && jpg->quant.size() < 4) {
      if (p + 1 > len) return fail("truncated DQT");
      JpegQuant t;
      t.precision = data[p] >> 4;
      t.index = data[p] & 15;
      ++p;
      if (t.index > 3) return fail("bad quantisation table id");
      if (p + (t.precision ? 128 : 64) > len) return fail("truncated quantisation table");
      for (int i = 0; i < 64; ++i) {
        int v;
        if (t.precision) { v = Be16(p); p += 2; } else { v = data[p++]; }
        if (v < 1) return fail("zero quantiser");
        t.values[kZigZagToNatural[i]] = v;
      }
      jpg->quant.push_back(t);
    }
    if (p != end) return fail("bad DQT length");   // (also: a fifth table)
    *pos = p;
    return true;
  }

  bool ProcessDRI(size_t* pos) {
    if (jpg->restart_interval > 0) return fail("duplicate DRI");   // as the reference (:379-393)
    if (*pos + 4 > len || Be16(*pos) != 4) return fail("bad DRI");
    jpg->restart_interval = Be16(*pos + 2);
    *pos += 4;
    return true;
  }

  bool SaveSegment(size_t* pos, bool app) {
    if (*pos + 2 > len) return fail("truncated segment");
    const size_t seg = Be16(*pos);
    if (seg < 2 || *pos + seg > len) return fail("bad segment length");
    if (app) jpg->app_data.push_back(std::string((const char*)data + *pos - 1, seg + 1));
    else jpg->com_data.push_back(std::string((const char*)data + *pos, seg));
    *pos += seg;
    return true;
  }

  // One block of a first pass over the band ss..se at bit position al (sequential: 0..63 at 0): DecodeDCTBlock,
  // :531-618.  *eobrun counts the blocks an end-of-band run still covers.
  bool DecodeBlockFirst(const HuffTable& dc, const HuffTable& ac, int ss, int se, int al, int16_t* coeffs,
                        int* last_dc, int* eobrun, BitReader* br) {
    const bool eobrun_allowed = ss > 0;
    if (ss == 0) {
      int s = DecodeSymbol(dc, br);
      if (s < 0 || s > 11) return fail("bad DC symbol");
      if (s) s = Extend(br->Read(s), s);
      s += *last_dc;
      const int v = ShiftLeftSigned(s, al);
      if (v != (int16_t)v) return fail("DC coefficient out of range");
      coeffs[0] = (int16_t)v;
      *last_dc = s;
      ++ss;
    }
    if (ss > se) return true;
    if (*eobrun > 0) {
      --*eobrun;
      return true;
    }
    for (int k = ss; k <= se; ++k) {
      const int rs = DecodeSymbol(ac, br);
      if (rs < 0) return fail("bad AC symbol");
      const int r = rs >> 4, sz = rs & 15;
      if (sz) {
        k += r;
        if (k > se) return fail("AC run past the band");
        if (sz + al >= 12) return fail("AC coefficient out of range");
        coeffs[kZigZagToNatural[k]] = (int16_t)ShiftLeftSigned(Extend(br->Read(sz), sz), al);
      } else if (r == 15) {
        k += 15;
      } else {
        *eobrun = 1 << r;
        if (r) {
          if (!eobrun_allowed) return fail("end-of-band run in a scan that carries DC");
          *eobrun += br->Read(r);
        }
        break;
      }
    }
    --*eobrun;
    return true;
  }

  // One block of a refinement pass (RefineDCTBlock, :620-730).
  bool DecodeBlockRefine(const HuffTable& ac, int ss, int se, int al, int16_t* coeffs, int* eobrun, BitReader* br) {
    const bool eobrun_allowed = ss > 0;
    if (ss == 0) {
      coeffs[0] = (int16_t)(coeffs[0] | (br->Read(1) << al));
      ++ss;
    }
    if (ss > se) return true;
    const int p1 = 1 << al, m1 = -(1 << al);
    auto refine = [&](int16_t* c) {
      if (br->Read(1) && (*c & p1) == 0) *c = (int16_t)(*c >= 0 ? *c + p1 : *c + m1);
    };
    int k = ss;
    bool after_zero_run = false;
    if (*eobrun <= 0) {
      for (; k <= se; ++k) {
        const int rs = DecodeSymbol(ac, br);
        if (rs < 0) return fail("bad AC symbol");
        int r = rs >> 4;
        const int sz = rs & 15;
        int value = 0;
        if (sz) {
          if (sz != 1) return fail("bad refinement symbol");
          value = br->Read(1) ? p1 : m1;
          after_zero_run = false;
        } else {
          if (r != 15) {
            *eobrun = 1 << r;
            if (r) {
              if (!eobrun_allowed) return fail("end-of-band run in a scan that carries DC");
              *eobrun += br->Read(r);
            }
            break;
          }
          after_zero_run = true;
        }
        // r coefficients without history are passed, the ones with history refined on the way
        do {
          int16_t* c = &coeffs[kZigZagToNatural[k]];
          if (*c != 0) refine(c);
          else if (--r < 0) break;
          ++k;
        } while (k <= se);
        if (value) {
          if (k > se) return fail("refinement past the band");
          coeffs[kZigZagToNatural[k]] = (int16_t)value;
        }
      }
    }
    if (after_zero_run) return fail("zero run before the end of the band");
    if (*eobrun > 0) {
      for (; k <= se; ++k) {
        int16_t* c = &coeffs[kZigZagToNatural[k]];
        if (*c != 0) refine(c);
      }
    }
    --*eobrun;
    return true;
  }

  // SOS + its entropy-coded segment (ProcessSOS :168-259, ProcessScan :755-888).
  bool ProcessScan(size_t* pos) {
    if (*pos + 3 > len) return fail("truncated SOS");
    const size_t seg = Be16(*pos);
    size_t p = *pos + 2;
    const int ns = data[p++];
    if (ns < 1 || ns > (int)jpg->components.size()) return fail("bad SOS");   // (also: SOS before SOF)
    if (p + 2 * (size_t)ns > len) return fail("truncated SOS");
    ScanComp sc[4];
    bool id_seen[256] = {false};
    for (int i = 0; i < ns; ++i) {
      const int id = data[p++];
      if (id_seen[id]) return fail("SOS names a component twice");
      id_seen[id] = true;
      sc[i].comp = -1;
      for (size_t j = 0; j < jpg->components.size(); ++j)
        if (jpg->components[j].id == id) sc[i].comp = (int)j;
      if (sc[i].comp < 0) return fail("SOS names an unknown component");
      const int tbl = data[p++];
      sc[i].dc_tbl = tbl >> 4;
      sc[i].ac_tbl = tbl & 15;
      if (sc[i].dc_tbl > 3 || sc[i].ac_tbl > 3) return fail("bad table selector");
    }
    if (p + 3 > len) return fail("truncated SOS");
    const int hss = data[p++], hse = data[p++];
    if (hss > 63 || hse < hss || hse > 63) return fail("bad spectral selection");
    const int hah = data[p] >> 4, hal = data[p] & 15;
    ++p;
    for (int i = 0; i < ns; ++i) {   // (by the header's band, also in a sequential frame)
      if (hss == 0 && !dc_tables[sc[i].dc_tbl].seen) return fail("scan uses an undefined DC Huffman table");
      if (hse > 0 && !ac_tables[sc[i].ac_tbl].seen) return fail("scan uses an undefined AC Huffman table");
    }
    if (*pos + seg != p) return fail("bad SOS length");
    *pos = p;

    // a sequential frame's scans cover everything at once, whatever their header says
    const int ss = jpg->progressive ? hss : 0, se = jpg->progressive ? hse : 63;
    const int ah = jpg->progressive ? hah : 0, al = jpg->progressive ? hal : 0;
    // no bit plane of a coefficient twice, and none below one that was refined already (:784-803)
    const uint16_t planes = (uint16_t)(ah == 0 ? 0xffffu << al : 1u << al), below = (uint16_t)((1u << al) - 1);
    for (int i = 0; i < ns; ++i)
      for (int k = ss; k <= se; ++k) {
        uint16_t& had = progression[sc[i].comp][k];
        if (had & planes) return fail("overlapping scans");
        if (had & below) return fail("a finer scan of this coefficient came first");
        had |= planes;
      }
    if (al > 10) return fail("unsupported successive approximation position");

    // geometry of the scan: interleaved = MCUs; single component = its own block grid
    const bool interleaved = ns > 1;
    int rows, cols;
    if (interleaved) {
      rows = jpg->mcu_rows;
      cols = jpg->mcu_cols;
    } else {
      const JpegComponentIn& c = jpg->components[sc[0].comp];
      cols = (jpg->width * c.h_samp + 8 * jpg->max_h_samp - 1) / (8 * jpg->max_h_samp);
      rows = (jpg->height * c.v_samp + 8 * jpg->max_v_samp - 1) / (8 * jpg->max_v_samp);
    }
    BitReader br(data, len, *pos);
    int last_dc[4] = {0, 0, 0, 0};
    int eobrun = -1;
    int restarts_left = jpg->restart_interval;
    int next_rst = 0;
    for (int my = 0; my < rows; ++my) {
      for (int mx = 0; mx < cols; ++mx) {
        if (jpg->restart_interval > 0) {
          if (restarts_left == 0) {
            size_t q = 0;
            if (!br.Finish(&q)) return fail("entropy-coded data ends early");
            if (q + 2 > len || data[q] != 0xff || data[q + 1] != 0xd0 + next_rst)
              return fail("missing restart marker");
            br.Reset(q + 2);
            next_rst = (next_rst + 1) & 7;
            restarts_left = jpg->restart_interval;
            memset(last_dc, 0, sizeof(last_dc));
            if (eobrun > 0) return fail("end-of-band run across a restart");
            eobrun = -1;
          }
          --restarts_left;
        }
        for (int i = 0; i < ns; ++i) {
          JpegComponentIn& c = jpg->components[sc[i].comp];
          const int nh = interleaved ? c.h_samp : 1, nv = interleaved ? c.v_samp : 1;
          for (int by = 0; by < nv; ++by) {
            for (int bx = 0; bx < nh; ++bx) {
              const int x = mx * nh + bx, y = my * nv + by;
              int16_t* coeffs = &c.coeffs[((size_t)y * c.width_in_blocks + x) * 64];
              const bool ok = ah == 0
                  ? DecodeBlockFirst(dc_tables[sc[i].dc_tbl], ac_tables[sc[i].ac_tbl], ss, se, al, coeffs,
                                     &last_dc[sc[i].comp], &eobrun, &br)
                  : DecodeBlockRefine(ac_tables[sc[i].ac_tbl], ss, se, al, coeffs, &eobrun, &br);
              if (!ok) return false;
            }
          }
        }
      }
    }
    if (eobrun > 0) return fail("end-of-band run past the scan");
    if (!br.Finish(pos)) return fail("entropy-coded data ends early");
    if (*pos > len) return fail("entropy-coded data ends early");
    return true;
  }

  static bool IsKnownMarker(int m) {
    return m == 0xc0 || m == 0xc1 || m == 0xc2 || m == 0xc4 || (m >= 0xd0 && m <= 0xd7) || m == 0xd9 || m == 0xda ||
           m == 0xdb || m == 0xdd || (m >= 0xe0 && m <= 0xef) || m == 0xfe;
  }

  bool Run() {
    size_t pos = 0;
    if (len < 2 || data[0] != 0xff || data[1] != 0xd8) return fail("no SOI marker");
    pos = 2;
    int marker = 0;
    do {
      // skip fill bytes / garbage up to the next marker the reference knows (FindNextMarker, :911-927: SOF0-2,
      // DHT, RST0-7, EOI, SOS, DQT, DRI, APP0-15, COM; anything else -- a second SOI, SOF3.., 0xf0-0xfd -- is
      // skipped like garbage, found by tests/test_fuzz_readers.py)
      while (pos + 1 < len && !(data[pos] == 0xff && IsKnownMarker(data[pos + 1]))) ++pos;
      if (pos + 2 > len || data[pos] != 0xff) return fail("marker expected");
      marker = data[pos + 1];
      pos += 2;
      bool ok = true;
      switch (marker) {
        case 0xc0: case 0xc1: case 0xc2: ok = ProcessSOF(&pos, marker); break;
        case 0xc4: ok = ProcessDHT(&pos); break;
        case 0xd0: case 0xd1: case 0xd2: case 0xd3: case 0xd4: case 0xd5: case 0xd6: case 0xd7:
        case 0xd9: break;
        case 0xda: ok = ProcessScan(&pos); break;
        case 0xdb: ok = ProcessDQT(&pos); break;
        case 0xdd: ok = ProcessDRI(&pos); break;
        case 0xfe: ok = SaveSegment(&pos, false); break;
        default:
          if (marker >= 0xe0 && marker <= 0xef) ok = SaveSegment(&pos, true);
          else return fail("unsupported marker");
      }
      if (!ok) return false;
    } while (marker != 0xd9);
    if (!found_sof) return fail("no SOF marker");
    if (pos < len) jpg->tail_data.assign((const char*)data + pos, len - pos);
    // FixupIndexes (:890-909): Tq -> position of the first table with that id
    for (size_t i = 0; i < jpg->components.size(); ++i) {
      JpegComponentIn& c = jpg->components[i];
      int found = -1;
      for (size_t j = 0; j < jpg->quant.size() && found < 0; ++j)
        if (jpg->quant[j].index == c.quant_idx) found = (int)j;
      if (found < 0) return fail("quantisation table not found");
      c.quant_idx = found;
    }
    if (tables_defined == 0) return fail("no Huffman table");          // (:1057-1063)
    if (tables_defined >= 512) return fail("too many Huffman tables");  // (:1064-1069)
    return true;
  }
};

}  // namespace

bool ReadJpeg(const uint8_t* data, size_t len, JpegInput* jpg, std::string* error) {
  *jpg = JpegInput();
  Decoder d;
  d.data = data;
  d.len = len;
  d.jpg = jpg;
  d.fail = Fail{error};
  return d.Run();
}

}  // namespace guetzli_amd
