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

// Canonical Huffman decoding table (T.81 C.2, F.2.2.3) with a 9-bit first-level lookup.
struct HuffTable {
  bool defined = false;
  int maxcode[18];     // largest code of each length, -1 if none
  int valptr[17];
  int mincode[17];
  uint8_t values[256];
  int num_values = 0;
  uint16_t fast[512];  // (length << 8) | symbol for codes of <= 9 bits, 0 otherwise

  bool Build(const uint8_t* counts /*[1..16]*/, const uint8_t* vals, int n) {
    num_values = n;
    memcpy(values, vals, n);
    memset(fast, 0, sizeof(fast));
    int code = 0, k = 0;
    for (int len = 1; len <= 16; ++len) {
      valptr[len] = k;
      mincode[len] = code;
      for (int i = 0; i < counts[len]; ++i, ++k, ++code) {
        if (len <= 9) {
          const int first = code << (9 - len), count = 1 << (9 - len);
          if (first + count > 512) return false;
          for (int j = 0; j < count; ++j) fast[first + j] = (uint16_t)((len << 8) | vals[k]);
        }
      }
      maxcode[len] = counts[len] ? code - 1 : -1;
      if (code > (1 << len)) return false;   // over-subscribed
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    defined = true;
    return true;
  }
};

// MSB-first bit reader over the entropy-coded segment: 0xFF00 is a stuffed 0xFF, any other
// marker ends the data (further bits read as zero, T.81 F.2.2.5 behaviour of decoders).
struct BitReader {
  const uint8_t* data;
  size_t len, pos;
  uint64_t acc = 0;
  int bits = 0;
  bool hit_marker = false;
  int overrun = 0;   // zero bytes supplied past the data

  BitReader(const uint8_t* d, size_t l, size_t p) : data(d), len(l), pos(p) {}
  void Fill() {
    while (bits <= 56) {
      uint8_t b = 0;
      if (!hit_marker && pos < len) {
        b = data[pos];
        if (b == 0xff) {
          if (pos + 1 < len && data[pos + 1] == 0x00) {
            pos += 2;
          } else {
            hit_marker = true;
            b = 0;
            ++overrun;
          }
        } else {
          ++pos;
        }
      } else {
        hit_marker = true;
        ++overrun;
      }
      acc = (acc << 8) | b;
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
  // Position of the next unread byte of the stream once whole bytes still in the accumulator
  // are given back (used at restart markers and at the end of the scan).
  bool TooFarPastEnd() const { return overrun * 8 > bits; }   // consumed bits that are not in the stream
  void AlignAndRewind() {
    // bytes fetched but not consumed go back; they were real bytes only if not overrun
    int whole = bits / 8;
    while (whole > 0 && overrun > 0) { --whole; --overrun; }
    while (whole > 0) {
      // step back over one stream byte (a stuffed 0xFF00 pair counts as one)
      if (pos >= 2 && data[pos - 1] == 0x00 && data[pos - 2] == 0xff) pos -= 2;
      else --pos;
      --whole;
    }
    acc = 0;
    bits = 0;
    hit_marker = false;
    overrun = 0;
  }
};

int DecodeSymbol(const HuffTable& t, BitReader* br) {
  const int look = br->Peek(9);
  const uint16_t f = t.fast[look];
  if (f) {
    br->Drop(f >> 8);
    return f & 0xff;
  }
  // codes longer than 9 bits: extend bit by bit (F.2.2.3)
  int code = look, len = 9;
  br->Drop(9);
  while (len <= 16 && (t.maxcode[len] < 0 || code > t.maxcode[len])) {
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

  uint16_t Be16(size_t p) const { return (uint16_t)((data[p] << 8) | data[p + 1]); }

  bool ProcessSOF(size_t* pos, int marker) {
    if (found_sof) return fail("duplicate SOF");
    if (*pos + 2 > len) return fail("truncated SOF");
    const size_t seg = Be16(*pos);
    if (seg < 8 || *pos + seg > len) return fail("bad SOF length");
    size_t p = *pos + 2;
    const int precision = data[p++];
    if (precision != 8) return fail("unsupported sample precision");
    jpg->height = Be16(p); p += 2;
    jpg->width = Be16(p); p += 2;
    const int nc = data[p++];
    if (jpg->height < 1 || jpg->width < 1) return fail("bad dimensions");
    if (nc < 1 || nc > 4 || seg != (size_t)(8 + 3 * nc)) return fail("bad component count");
    jpg->progressive = marker == 0xc2;
    jpg->components.assign(nc, JpegComponentIn());
    for (int i = 0; i < nc; ++i) {
      JpegComponentIn& c = jpg->components[i];
      c.id = data[p++];
      const int hv = data[p++];
      c.h_samp = hv >> 4;
      c.v_samp = hv & 15;
      c.quant_idx = data[p++];
      if (c.h_samp < 1 || c.h_samp > 15 || c.v_samp < 1 || c.v_samp > 15 || c.quant_idx > 3)
        return fail("bad component parameters");
      for (int j = 0; j < i; ++j)
        if (jpg->components[j].id == c.id) return fail("duplicate component id");
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
    *pos += seg;
    found_sof = true;
    return true;
  }

  bool ProcessDHT(size_t* pos) {
    if (*pos + 2 > len) return fail("truncated DHT");
    const size_t seg = Be16(*pos);
    if (seg < 2 || *pos + seg > len) return fail("bad DHT length");
    size_t p = *pos + 2;
    const size_t end = *pos + seg;
    if (p == end) return fail("empty DHT");
    while (p < end) {
      if (p + 17 > end) return fail("truncated DHT table");
      const int tc = data[p] >> 4, th = data[p] & 15;
      ++p;
      if (tc > 1 || th > 3) return fail("bad Huffman table id");
      uint8_t counts[17] = {0};
      int total = 0;
      for (int l = 1; l <= 16; ++l) {
        counts[l] = data[p++];
        total += counts[l];
      }
      if (total < 1 || total > 256 || p + total > end) return fail("bad Huffman table size");
      HuffTable& t = tc == 0 ? dc_tables[th] : ac_tables[th];
      if (!t.Build(counts, data + p, total)) return fail("invalid Huffman code");
      p += total;
    }
    *pos += seg;
    return true;
  }

  bool ProcessDQT(size_t* pos) {
    if (*pos + 2 > len) return fail("truncated DQT");
    const size_t seg = Be16(*pos);
    if (seg < 2 || *pos + seg > len) return fail("bad DQT length");
    size_t p = *pos + 2;
    const size_t end = *pos + seg;
    if (p == end) return fail("empty DQT");
    while (p < end) {
      JpegQuant t;
      t.precision = data[p] >> 4;
      t.index = data[p] & 15;
      ++p;
      if (t.index > 3 || t.precision > 1) return fail("bad quantisation table id");
      if (p + (t.precision ? 128 : 64) > end) return fail("truncated quantisation table");
      for (int i = 0; i < 64; ++i) {
        int v;
        if (t.precision) { v = Be16(p); p += 2; } else { v = data[p++]; }
        if (v < 1) return fail("zero quantiser");
        t.values[kZigZagToNatural[i]] = v;
      }
      jpg->quant.push_back(t);
    }
    *pos += seg;
    return true;
  }

  bool ProcessDRI(size_t* pos) {
    if (*pos + 4 > len || Be16(*pos) != 4) return fail("bad DRI");
    if (jpg->restart_interval > 0) return fail("duplicate DRI");   // as the reference (:379-393)
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

  // One block of a sequential scan (F.2.2).
  bool DecodeBlockSequential(const HuffTable& dc, const HuffTable& ac, int16_t* coeffs,
                             int* last_dc, BitReader* br) {
    int s = DecodeSymbol(dc, br);
    if (s < 0 || s > 11) return fail("bad DC symbol");
    int diff = 0;
    if (s) diff = Extend(br->Read(s), s);
    *last_dc += diff;
    coeffs[0] = (int16_t)*last_dc;
    for (int k = 1; k < 64;) {
      const int rs = DecodeSymbol(ac, br);
      if (rs < 0) return fail("bad AC symbol");
      const int r = rs >> 4, sz = rs & 15;
      if (sz == 0) {
        if (r == 15) { k += 16; continue; }
        break;   // EOB
      }
      k += r;
      if (k > 63) return fail("AC run past the block");
      coeffs[kZigZagToNatural[k]] = (int16_t)Extend(br->Read(sz), sz);
      ++k;
    }
    return true;
  }

  // One block of a progressive scan (G.1.2).
  bool DecodeBlockProgressive(const HuffTable& dc, const HuffTable& ac, int ss, int se, int ah,
                              int al, int16_t* coeffs, int* last_dc, int* eobrun,
                              BitReader* br) {
    if (ss == 0) {
      if (ah == 0) {   // DC first
        const int s = DecodeSymbol(dc, br);
        if (s < 0 || s > 11) return fail("bad DC symbol");
        int diff = 0;
        if (s) diff = Extend(br->Read(s), s);
        *last_dc += diff;
        coeffs[0] = (int16_t)(*last_dc * (1 << al));
      } else {         // DC refinement
        if (br->Read(1)) coeffs[0] = (int16_t)(coeffs[0] | (1 << al));
      }
      return true;
    }
    if (ah == 0) {     // AC first
      if (*eobrun > 0) {
        --*eobrun;
        return true;
      }
      for (int k = ss; k <= se;) {
        const int rs = DecodeSymbol(ac, br);
        if (rs < 0) return fail("bad AC symbol");
        const int r = rs >> 4, sz = rs & 15;
        if (sz == 0) {
          if (r == 15) { k += 16; continue; }
          *eobrun = (1 << r) - 1;
          if (r) *eobrun += br->Read(r);
          break;
        }
        k += r;
        if (k > se) return fail("AC run past the band");
        coeffs[kZigZagToNatural[k]] = (int16_t)(Extend(br->Read(sz), sz) * (1 << al));
        ++k;
      }
      return true;
    }
    // AC refinement (G.1.2.3)
    const int p1 = 1 << al, m1 = -(1 << al);
    int k = ss;
    if (*eobrun == 0) {
      for (; k <= se;) {
        const int rs = DecodeSymbol(ac, br);
        if (rs < 0) return fail("bad AC symbol");
        int r = rs >> 4;
        const int sz = rs & 15;
        int value = 0;
        if (sz == 0) {
          if (r != 15) {
            *eobrun = 1 << r;
            if (r) *eobrun += br->Read(r);
            break;
          }
        } else if (sz == 1) {
          value = br->Read(1) ? p1 : m1;
        } else {
          return fail("bad refinement symbol");
        }
        // skip r zero-history coefficients, refining the non-zero ones on the way
        for (; k <= se; ++k) {
          int16_t* c = &coeffs[kZigZagToNatural[k]];
          if (*c != 0) {
            if (br->Read(1) && (*c & p1) == 0) *c = (int16_t)(*c >= 0 ? *c + p1 : *c + m1);
          } else {
            if (r == 0) break;
            --r;
          }
        }
        if (value) {
          if (k > se) return fail("refinement past the band");
          coeffs[kZigZagToNatural[k]] = (int16_t)value;
        }
        ++k;
      }
    }
    if (*eobrun > 0) {
      for (; k <= se; ++k) {
        int16_t* c = &coeffs[kZigZagToNatural[k]];
        if (*c != 0 && br->Read(1) && (*c & p1) == 0) *c = (int16_t)(*c >= 0 ? *c + p1 : *c + m1);
      }
      --*eobrun;
    }
    return true;
  }

  bool ProcessScan(size_t* pos) {
    if (!found_sof) return fail("SOS before SOF");
    if (*pos + 3 > len) return fail("truncated SOS");
    const size_t seg = Be16(*pos);
    size_t p = *pos + 2;
    const int ns = data[p++];
    if (ns < 1 || ns > (int)jpg->components.size() || seg != (size_t)(6 + 2 * ns) ||
        *pos + seg > len)
      return fail("bad SOS");
    ScanComp sc[4];
    for (int i = 0; i < ns; ++i) {
      const int id = data[p++];
      const int tbl = data[p++];
      sc[i].comp = -1;
      for (size_t j = 0; j < jpg->components.size(); ++j)
        if (jpg->components[j].id == id) sc[i].comp = (int)j;
      if (sc[i].comp < 0) return fail("SOS names an unknown component");
      for (int j = 0; j < i; ++j)
        if (sc[j].comp >= sc[i].comp) return fail("SOS components out of order");
      sc[i].dc_tbl = tbl >> 4;
      sc[i].ac_tbl = tbl & 15;
      if (sc[i].dc_tbl > 3 || sc[i].ac_tbl > 3) return fail("bad table selector");
    }
    const int ss = data[p++], se = data[p++];
    const int ah = data[p] >> 4, al = data[p] & 15;
    ++p;
    if (jpg->progressive) {
      if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || al > 13 ||
          (ah != 0 && ah != al + 1))
        return fail("bad progression parameters");
    } else if (ss != 0 || se != 63 || ah != 0 || al != 0) {
      return fail("bad spectral selection for a sequential scan");
    }
    for (int i = 0; i < ns; ++i) {
      const bool need_dc = ss == 0, need_ac = jpg->progressive ? ss > 0 : true;
      if ((need_dc && !(jpg->progressive && ah) && !dc_tables[sc[i].dc_tbl].defined) ||
          (need_ac && !ac_tables[sc[i].ac_tbl].defined))
        return fail("scan uses an undefined Huffman table");
    }
    *pos += seg;

    // geometry of the scan: interleaved = MCUs; single component = its own block grid
    const bool interleaved = ns > 1;
    int rows, cols;
    if (interleaved) {
      rows = jpg->mcu_rows;
      cols = jpg->mcu_cols;
    } else {
      const JpegComponentIn& c = jpg->components[sc[0].comp];
      const int wpx = (jpg->width * c.h_samp + jpg->max_h_samp - 1) / jpg->max_h_samp;
      const int hpx = (jpg->height * c.v_samp + jpg->max_v_samp - 1) / jpg->max_v_samp;
      cols = (wpx + 7) / 8;
      rows = (hpx + 7) / 8;
    }
    BitReader br(data, len, *pos);
    int last_dc[4] = {0, 0, 0, 0};
    int eobrun = 0;
    int restarts_left = jpg->restart_interval;
    int next_rst = 0;
    for (int my = 0; my < rows; ++my) {
      for (int mx = 0; mx < cols; ++mx) {
        if (jpg->restart_interval > 0) {
          if (restarts_left == 0) {
            br.AlignAndRewind();
            size_t q = br.pos;
            if (q + 2 > len || data[q] != 0xff || data[q + 1] != 0xd0 + next_rst)
              return fail("missing restart marker");
            br.pos = q + 2;
            next_rst = (next_rst + 1) & 7;
            restarts_left = jpg->restart_interval;
            memset(last_dc, 0, sizeof(last_dc));
            eobrun = 0;
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
              const HuffTable& dc = dc_tables[sc[i].dc_tbl];
              const HuffTable& ac = ac_tables[sc[i].ac_tbl];
              const bool ok = jpg->progressive
                  ? DecodeBlockProgressive(dc, ac, ss, se, ah, al, coeffs, &last_dc[sc[i].comp],
                                           &eobrun, &br)
                  : DecodeBlockSequential(dc, ac, coeffs, &last_dc[sc[i].comp], &br);
              if (!ok) return false;
              if (br.TooFarPastEnd()) return fail("entropy-coded data ends early");
            }
          }
        }
      }
    }
    br.AlignAndRewind();
    *pos = br.pos;
    return true;
  }

  bool Run() {
    size_t pos = 0;
    if (len < 4 || data[0] != 0xff || data[1] != 0xd8) return fail("no SOI marker");
    pos = 2;
    int marker = 0;
    do {
      // skip fill bytes / garbage up to the next marker (FindNextMarker, :911-927)
      while (pos + 1 < len && !(data[pos] == 0xff && data[pos + 1] >= 0xc0 && data[pos + 1] != 0xff))
        ++pos;
      if (pos + 2 > len) return fail("marker expected");
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
