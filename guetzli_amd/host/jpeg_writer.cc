#include "jpeg_writer.h"

#include <string.h>

#include <algorithm>
#include <cstdlib>

namespace guetzli_amd {

const int kNaturalOrder[64] = {   // kJPEGNaturalOrder, jpeg_data.h:62-73
  0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
  35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
  58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
const int kZigZagOrder[64] = {    // kJPEGZigZagOrder, jpeg_data.h:75-84
  0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42,
  3,  8,  12, 17, 25, 30, 41, 43, 9,  11, 18, 24, 31, 40, 44, 53,
  10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60,
  21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

namespace {

inline int FloorLog2(uint32_t n) { return 31 ^ __builtin_clz(n); }   // fast_log.h:25-39
inline int BitLength(uint32_t n) { return n == 0 ? 0 : FloorLog2(n) + 1; }

}  // namespace

void SymbolHistogram::Clear() {
  memset(counts, 0, sizeof(counts));
  counts[kHistoSize - 1] = 1;
}
void SymbolHistogram::Merge(const SymbolHistogram& other) {
  for (int i = 0; i + 1 < kHistoSize; ++i) counts[i] += other.counts[i];
  counts[kHistoSize - 1] = 1;
}
int SymbolHistogram::NumSymbols() const {
  int n = 0;
  for (int i = 0; i + 1 < kHistoSize; ++i) n += counts[i] > 0;
  return n;
}

// ------------------------------------------------------------------ Huffman depths ----
// Two-queue Huffman construction over leaves sorted by (count ascending, symbol
// descending); on equal weight the leaf queue is preferred; if the tree is deeper than
// tree_limit, all counts are raised to a doubling floor and the construction repeats --
// exactly the procedure of CreateHuffmanTree / SetDepth (entropy_encode.cc:25-145), whose
// tie rules decide the code lengths and therefore the bytes.
namespace {
struct HuffNode {
  uint32_t weight;
  int left;    // -1 for a leaf
  int right;   // child index, or the symbol for a leaf
};
// The size model of phase B asks for the depths of the same few histograms over and over
// (every 10 coefficient steps, processor.cc:741-743), and most steps touch one component:
// remember the last answers per thread.
struct HuffMemo {
  static const int kSlots = 8;
  uint32_t counts[kSlots][kHistoSize];
  uint8_t depth[kSlots][kHistoSize];
  int limit[kSlots];
  bool valid[kSlots];
  int next;
};
}  // namespace

static void HuffmanDepthsUncached(const uint32_t* counts, size_t length, int tree_limit,
                                  uint8_t* depth, int stream);

void HuffmanDepths(const uint32_t* counts, size_t length, int tree_limit, uint8_t* depth, int stream) {
  if (length != (size_t)kHistoSize) {
    HuffmanDepthsUncached(counts, length, tree_limit, depth, -1);
    return;
  }
  static thread_local HuffMemo memo = {};
  for (int s = 0; s < HuffMemo::kSlots; ++s)
    if (memo.valid[s] && memo.limit[s] == tree_limit &&
        memcmp(memo.counts[s], counts, sizeof(memo.counts[s])) == 0) {
      // depth[] entries of absent symbols are left untouched by the construction: same here
      for (int i = 0; i < kHistoSize; ++i)
        if (counts[i]) depth[i] = memo.depth[s][i];
      return;
    }
  HuffmanDepthsUncached(counts, length, tree_limit, depth, stream);
  const int s = memo.next;
  memo.next = (memo.next + 1) % HuffMemo::kSlots;
  memcpy(memo.counts[s], counts, sizeof(memo.counts[s]));
  memcpy(memo.depth[s], depth, sizeof(memo.depth[s]));
  memo.limit[s] = tree_limit;
  memo.valid[s] = true;
}

namespace {
// The order of the present symbols by (count, symbol descending) of the last construction on a
// "stream" of histograms (phase B asks for the codes of the same component again after ten
// coefficient steps: the counts have moved by a few units and the order hardly at all).
struct HuffOrder {
  int n;
  int16_t sym_id[260];
  int16_t order[260];
};
const int kHuffStreams = 8;
}  // namespace

static void HuffmanDepthsUncached(const uint32_t* counts, size_t length, int tree_limit,
                                  uint8_t* depth, int stream) {
  // Called ~10^4 times per image by phase B's size model, typically with ~8 repetitions each
  // (AC counts span 1..10^6, so the unconstrained tree is ~22 deep): flat arrays, one sort (or
  // the last order of the stream, repaired), leaves that are a fill and a copy, and no tree walk
  // for the attempts that fail.
  const size_t kMax = 2 * 260 + 2;
  if (length > 260) {   // not a JPEG histogram: refuse loudly
    fprintf(stderr, "guetzli_amd: HuffmanDepths on %zu symbols is not supported\n", length);
    abort();
  }
  uint32_t sym_count[260];
  int16_t sym_id[260];
  size_t n = 0;
  for (size_t i = length; i-- > 0;)   // present symbols, symbol descending
    if (counts[i]) { sym_count[n] = counts[i]; sym_id[n] = (int16_t)i; ++n; }
  if (n == 0) return;
  if (n == 1) {
    depth[sym_id[0]] = 1;
    return;
  }
  // The leaves of one construction are ordered by (weight ascending, symbol descending) with
  // weight = max(count, floor).  All leaves at the floor therefore come first in symbol
  // order, and the rest follow in (count, symbol) order whatever the floor is: one order
  // serves every repetition.  order[j] = position in sym_* of the j-th smallest (count, position).
  int16_t order[260];
  static thread_local HuffOrder last[kHuffStreams];
  HuffOrder* memo = stream >= 0 && stream < kHuffStreams ? &last[stream] : nullptr;
  if (memo && memo->n == (int)n && memcmp(memo->sym_id, sym_id, n * sizeof(int16_t)) == 0) {
    // the same symbols as last time: their last order, repaired by insertion (a total order:
    // whichever way it is reached, the result is the one a sort gives)
    order[0] = memo->order[0];
    for (size_t j = 1; j < n; ++j) {
      const int16_t p = memo->order[j];
      const uint32_t c = sym_count[p];
      size_t k = j;
      while (k > 0 && (sym_count[order[k - 1]] > c || (sym_count[order[k - 1]] == c && order[k - 1] > p))) {
        order[k] = order[k - 1];
        --k;
      }
      order[k] = p;
    }
  } else {
    uint64_t by_count[260];   // (count, position in sym_*) ascending
    for (size_t i = 0; i < n; ++i) by_count[i] = ((uint64_t)sym_count[i] << 32) | (uint64_t)i;
    std::sort(by_count, by_count + n);
    for (size_t j = 0; j < n; ++j) order[j] = (int16_t)(by_count[j] & 0xffffffffu);
  }
  if (memo) {
    memo->n = (int)n;
    memcpy(memo->sym_id, sym_id, n * sizeof(int16_t));
    memcpy(memo->order, order, n * sizeof(int16_t));
  }
  uint32_t sorted_count[260];
  for (size_t j = 0; j < n; ++j) sorted_count[j] = sym_count[order[j]];
  uint32_t weight[kMax];
  int16_t child[kMax][2];
  uint8_t height[kMax];   // of the subtree under a node, <= tree_limit while building
  size_t above = 0;   // first entry of the order whose count exceeds the floor
  for (uint32_t floor_count = 1;; floor_count *= 2) {
    // a floor below every count changes nothing: same (too deep) tree as the attempt before
    if (floor_count > 1 && sorted_count[0] >= floor_count) continue;
    while (above < n && sorted_count[above] <= floor_count) ++above;
    // the leaves' weights: `above` leaves at the floor, then the counts above it (which symbols
    // they are matters only for the attempt that succeeds)
    for (size_t j = 0; j < above; ++j) weight[j] = floor_count;
    memcpy(weight + above, sorted_count + above, (n - above) * sizeof(uint32_t));
    // two queues: leaves [0, n) and inner nodes [n + 1, ...), each closed by a sentinel; on
    // equal weight the leaf is taken
    weight[n] = ~0u;
    weight[n + 1] = ~0u;
    memset(height, 0, n + 1);
    size_t leaf = 0, inner = n + 1;
    bool too_deep = false;
    for (size_t made = 0; made + 1 < n; ++made) {
      size_t pick[2];
      for (int k = 0; k < 2; ++k) {
        pick[k] = weight[leaf] <= weight[inner] ? leaf++ : inner++;
      }
      const size_t at = n + 1 + made;
      weight[at] = weight[pick[0]] + weight[pick[1]];
      child[at][0] = (int16_t)pick[0];
      child[at][1] = (int16_t)pick[1];
      // a subtree higher than the limit puts a leaf deeper than the limit whatever is built
      // above it (the reference finds that out in SetDepth, entropy_encode.cc:25-63)
      const int hh = 1 + std::max(height[pick[0]], height[pick[1]]);
      if (hh > tree_limit) { too_deep = true; break; }
      height[at] = (uint8_t)hh;
      weight[at + 1] = ~0u;
    }
    if (too_deep) continue;   // raise the floor
    const size_t root = 2 * n - 1;
    // children are made before their parents: one pass from the root down assigns the levels
    uint8_t level[kMax];
    level[root] = 0;
    for (size_t at = root; at > n; --at) {
      const uint8_t l = (uint8_t)(level[at] + 1);
      level[child[at][0]] = l;
      level[child[at][1]] = l;
    }
    // leaf m: the floor's leaves in symbol order (descending), then the order's rest
    size_t m = 0;
    if (above > 0)
      for (size_t i = 0; i < n; ++i)
        if (sym_count[i] <= floor_count) depth[sym_id[i]] = level[m++];
    for (size_t j = above; j < n; ++j) depth[sym_id[order[j]]] = level[m++];
    return;
  }
}

size_t HistogramHeaderBits(const SymbolHistogram& h) {
  size_t bits = 17 * 8;
  for (int i = 0; i + 1 < kHistoSize; ++i) bits += h.counts[i] > 0 ? 8 : 0;
  return bits;
}

size_t HistogramEntropyBits(const SymbolHistogram& h, const uint8_t* depth) {
  return EntropyBitsFromRaw(HistogramRawBits(h, depth));
}

size_t ClusterHistograms(SymbolHistogram* histo, size_t* num, int* histo_indexes,
                         uint8_t* depth) {
  memset(depth, 0, *num * kHistoSize);
  size_t costs[4];
  for (size_t i = 0; i < *num; ++i) {
    histo_indexes[i] = (int)i;
    HuffmanDepths(histo[i].counts, kHistoSize, 16, &depth[i * kHistoSize], (int)i);
    costs[i] = HistogramHeaderBits(histo[i]) + HistogramEntropyBits(histo[i], &depth[i * kHistoSize]);
  }
  const size_t orig_num = *num;
  while (*num > 1) {
    const size_t last = *num - 1, prev = *num - 2;
    SymbolHistogram both(histo[last]);
    both.Merge(histo[prev]);
    uint8_t depth_both[kHistoSize] = {0};
    HuffmanDepths(both.counts, kHistoSize, 16, depth_both, 3 + (int)(orig_num - *num));
    const size_t cost_both = HistogramHeaderBits(both) + HistogramEntropyBits(both, depth_both);
    if (cost_both >= costs[last] + costs[prev]) break;
    histo[prev] = both;
    histo[last] = SymbolHistogram();
    costs[prev] = cost_both;
    memcpy(&depth[prev * kHistoSize], depth_both, sizeof(depth_both));
    for (size_t i = 0; i < orig_num; ++i)
      if (histo_indexes[i] == (int)last) histo_indexes[i] = (int)prev;
    --(*num);
  }
  size_t total = 0;
  for (size_t i = 0; i < *num; ++i) total += costs[i];
  return (total + 7) / 8;
}

// Entropy-size model of the AC coefficients (processor.cc:497-525).
size_t EntropyCodes(const SymbolHistogram* histo, int n, uint8_t* depths /*3*257*/) {
  SymbolHistogram clustered[3] = {histo[0], histo[1], histo[2]};
  size_t num = (size_t)n;
  int indexes[3];
  uint8_t cdepths[3 * kHistoSize];
  ClusterHistograms(clustered, &num, indexes, cdepths);
  for (int i = 0; i < n; ++i)
    memcpy(&depths[i * kHistoSize], &cdepths[indexes[i] * kHistoSize], kHistoSize);
  size_t header = 0;
  for (size_t i = 0; i < num; ++i) header += HistogramHeaderBits(clustered[i]) / 8;
  return header;
}
size_t EntropyDataSize(const SymbolHistogram* histo, int n, const uint8_t* depths) {
  size_t bits = 0;
  for (int i = 0; i < n; ++i) bits += HistogramEntropyBits(histo[i], &depths[i * kHistoSize]);
  return (bits + 7) / 8;
}

void AddBlockACSymbols(const int16_t* block, const int* q, int weight, SymbolHistogram* h,
                       const uint8_t* depth, int64_t* raw_bits) {
  int run = 0;
  int64_t bits = 0;   // what the symbols cost under `depth`: HistogramEntropyBits' sum, term by term
  auto add = [&](int symbol) {
    h->Add(symbol, weight);
    if (depth) bits += depth[symbol] + (symbol & 0xf);
  };
  for (int k = 1; k < 64; ++k) {
    const int nat = kNaturalOrder[k];
    const int v = block[nat];
    if (v == 0) {
      ++run;
      continue;
    }
    while (run > 15) {
      add(0xf0);
      run -= 16;
    }
    const int mag = std::abs(q ? v / q[nat] : v);
    add((run << 4) + FloorLog2((uint32_t)mag) + 1);
    run = 0;
  }
  if (run > 0) add(0);
  if (raw_bits) *raw_bits += weight * bits;
}

// The same update for ONE coefficient that changes (natural index k >= 1, blk still holds the old
// value): the AC symbols of a block differ only between the non-zero coefficient before position
// k of the scan and the one after it -- the coefficient's own symbol, its successor's zero run and
// the end-of-block code.  Equal to AddBlockACSymbols(blk, q, -1) + the store + AddBlockACSymbols
// (blk, q, +1), for a few coefficients' work instead of two passes over the block.
namespace {
// natural index -> zig-zag position; built once, under C++11's thread-safe initialisation of a
// function-local static (several encodes run as threads of one process: batch.run_config5)
const int* ZigzagOfNatural() {
  struct Inverse {
    int at[64];
    Inverse() { for (int z = 0; z < 64; ++z) at[kNaturalOrder[z]] = z; }
  };
  static const Inverse inverse;
  return inverse.at;
}

// add(symbol, weight) for every AC symbol occurrence that leaves (weight -1) or enters (+1) the
// block's scan when the coefficient at natural index k >= 1 changes from blk[k] to newval.
template <class Add>
inline void ForCoeffACSymbolChanges(const int16_t* blk, const int* q, int k, int newval, Add&& add) {
  const int* zigzag_of = ZigzagOfNatural();
  const int oldval = blk[k];
  if (oldval == newval) return;
  const int z = zigzag_of[k];
  int pz = z - 1, nz = z + 1;             // the non-zero neighbours in scan order: 0 / 64 if none
  while (pz >= 1 && blk[kNaturalOrder[pz]] == 0) --pz;
  while (nz <= 63 && blk[kNaturalOrder[nz]] == 0) ++nz;
  auto coeff = [&](int run, int v, int nat, int weight) {   // `run` zeros, then the value v at nat
    while (run > 15) {
      add(0xf0, weight);
      run -= 16;
    }
    const int mag = std::abs(q ? v / q[nat] : v);
    add((run << 4) + FloorLog2((uint32_t)mag) + 1, weight);
  };
  auto window = [&](int v, int weight) {   // the symbols between pz and nz with v at position z
    int last = pz;                         // last non-zero position so far
    if (v != 0) {
      coeff(z - 1 - pz, v, k, weight);
      last = z;
    }
    if (nz <= 63) coeff(nz - 1 - last, blk[kNaturalOrder[nz]], kNaturalOrder[nz], weight);
    else if (last < 63) add(0, weight);    // trailing zeros: the end-of-block code
  };
  window(oldval, -1);
  window(newval, 1);
}
}  // namespace

void ReplaceCoeffACSymbols(const int16_t* blk, const int* q, int k, int newval, SymbolHistogram* h,
                           const uint8_t* depth, int64_t* raw_bits) {
  int64_t bits = 0;
  ForCoeffACSymbolChanges(blk, q, k, newval, [&](int symbol, int weight) {
    h->Add(symbol, weight);
    if (depth) bits += weight * (depth[symbol] + (symbol & 0xf));
  });
  if (raw_bits) *raw_bits += bits;
}

int CoeffACSymbolChanges(const int16_t* blk, const int* q, int k, int newval, int16_t* changes) {
  int n = 0;
  ForCoeffACSymbolChanges(blk, q, k, newval, [&](int symbol, int weight) {
    changes[n++] = (int16_t)(weight > 0 ? symbol + 1 : -(symbol + 1));
  });
  return n;
}

int64_t HistogramRawBits(const SymbolHistogram& h, const uint8_t* depth) {
  int64_t bits = 0;
  for (int i = 0; i + 1 < kHistoSize; ++i) bits += (int64_t)(h.counts[i] / 2) * (depth[i] + (i & 0xf));
  return bits;
}

size_t EntropyBitsFromRaw(int64_t raw) {
  size_t bits = (size_t)raw;
  bits += (bits * 3 + 512) >> 10;   // estimated 0xff escapes
  return bits;
}

// ------------------------------------------------------------------------- frames ----
namespace {

void AssignQuantTables(const int q[3][64], Frame* f) {   // SaveQuantTables, jpeg_data.cc:71-102
  f->quant.clear();
  for (int c = 0; c < f->ncomp; ++c) {
    int found = -1;
    for (size_t j = 0; j < f->quant.size(); ++j)
      if (memcmp(q[c], f->quant[j].values, sizeof(int) * 64) == 0) {
        found = (int)j;
        break;
      }
    if (found < 0) {
      QuantTable t;
      memcpy(t.values, q[c], sizeof(t.values));
      t.precision = 0;
      for (int k = 0; k < 64; ++k)
        if (t.values[k] > 0xff) t.precision = 1;
      t.index = (int)f->quant.size();
      found = t.index;
      f->quant.push_back(t);
    }
    f->quant_idx[c] = found;
  }
}

bool AllZero(const int16_t* p, size_t n) {
  for (size_t i = 0; i < n; ++i)
    if (p[i]) return false;
  return true;
}

}  // namespace

// Sampling geometry of a frame with `ncomp` components written and chroma factor `factor`
// (SaveToJpegData :353-380: with one component everything is 1 x 1).
static void SetGeometry(int w, int h, int ncomp, int factor, Frame* f) {
  f->width = w;
  f->height = h;
  f->bw = (w + 7) / 8;
  f->bh = (h + 7) / 8;
  f->ncomp = ncomp;
  const int fac = ncomp == 1 ? 1 : factor;
  f->mcu_cols = (w + 8 * fac - 1) / (8 * fac);
  f->mcu_rows = (h + 8 * fac - 1) / (8 * fac);
  for (int c = 0; c < 3; ++c) {
    f->samp[c] = c == 0 ? fac : 1;
    f->cw[c] = f->mcu_cols * f->samp[c];
    f->ch[c] = f->mcu_rows * f->samp[c];
  }
}

void FrameFromImage(const int16_t* coeffs, const int q[3][64], int w, int h, Frame* f) {
  FrameFromImageFactor(coeffs, q, w, h, 1, f);
}

void FrameFromImageFactor(const int16_t* coeffs, const int q[3][64], int w, int h, int factor,
                          Frame* f) {
  const int bw = (w + 7) / 8, bh = (h + 7) / 8;
  const int cbw = (w + 8 * factor - 1) / (8 * factor), cbh = (h + 8 * factor - 1) / (8 * factor);
  const size_t nb = (size_t)bw * bh, nbc = (size_t)cbw * cbh;
  // a single component is written when both chroma planes are entirely zero
  const int ncomp = AllZero(coeffs + nb * 64, 2 * nbc * 64) ? 1 : 3;
  SetGeometry(w, h, ncomp, factor, f);
  for (int c = 0; c < 3; ++c) f->coeffs[c].clear();
  for (int c = 0; c < ncomp; ++c) {
    const int rw = c == 0 ? bw : cbw, rh = c == 0 ? bh : cbh;   // the component's real blocks
    const int16_t* src = coeffs + (c == 0 ? 0 : (nb + (size_t)(c - 1) * nbc)) * 64;
    f->coeffs[c].assign((size_t)f->cw[c] * f->ch[c] * 64, 0);
    int16_t* dst = f->coeffs[c].data();
    int last_dc = 0;
    for (int by = 0; by < f->ch[c]; ++by)
      for (int bx = 0; bx < f->cw[c]; ++bx, dst += 64) {
        if (by >= rh || bx >= rw) {
          dst[0] = (int16_t)last_dc;   // the AC part stays zero
        } else {
          const int16_t* s = src + ((size_t)by * rw + bx) * 64;
          for (int k = 0; k < 64; ++k) dst[k] = (int16_t)(s[k] / q[c][k]);
        }
        last_dc = dst[0];
      }
  }
  AssignQuantTables(q, f);
}

void FrameTables(const int q[3][64], int w, int h, int ncomp, Frame* f) {
  FrameTablesFactor(q, w, h, ncomp, 1, f);
}

void FrameTablesFactor(const int q[3][64], int w, int h, int ncomp, int factor, Frame* f) {
  SetGeometry(w, h, ncomp, factor, f);
  for (int c = 0; c < 3; ++c) f->coeffs[c].clear();
  if (q) {
    AssignQuantTables(q, f);
  } else {   // the q=1 frame of EncodeRGBToJpeg: three all-ones tables, all with index 0
    f->quant.clear();
    for (int c = 0; c < 3; ++c) {
      QuantTable t;
      for (int k = 0; k < 64; ++k) t.values[k] = 1;
      t.precision = 0;
      t.index = 0;
      f->quant.push_back(t);
      f->quant_idx[c] = c;
    }
  }
}

void FrameFromOriginal(const int16_t* coeffs, int w, int h, Frame* f) {
  SetGeometry(w, h, 3, 1, f);
  const size_t n = (size_t)f->bw * f->bh * 64;
  f->quant.clear();
  for (int c = 0; c < 3; ++c) {
    f->coeffs[c].assign(coeffs + (size_t)c * n, coeffs + (size_t)(c + 1) * n);
    QuantTable t;
    for (int k = 0; k < 64; ++k) t.values[k] = 1;
    t.precision = 0;
    t.index = 0;   // JPEGQuantTable() default; EncodeRGBToJpeg never sets it
    f->quant.push_back(t);
    f->quant_idx[c] = c;
  }
}

void BuildDCHistograms(const Frame& f, SymbolHistogram* histo) {
  for (int c = 0; c < f.ncomp; ++c) {
    int last = 0;
    const int16_t* p = f.coeffs[c].data();
    const int sp = f.samp[c];
    for (int my = 0; my < f.mcu_rows; ++my)
      for (int mx = 0; mx < f.mcu_cols; ++mx)
        for (int iy = 0; iy < sp; ++iy)
          for (int ix = 0; ix < sp; ++ix) {
            const size_t b = (size_t)(my * sp + iy) * f.cw[c] + (mx * sp + ix);
            const int dc = p[b * 64];
            histo[c].Add(BitLength((uint32_t)std::abs(dc - last)));
            last = dc;
          }
  }
}

void BuildACHistograms(const Frame& f, SymbolHistogram* histo) {
  for (int c = 0; c < f.ncomp; ++c) {
    const size_t nb = (size_t)f.cw[c] * f.ch[c];   // padding blocks count too (an EOB each)
    for (size_t b = 0; b < nb; ++b) AddBlockACSymbols(&f.coeffs[c][b * 64], nullptr, 1, &histo[c]);
  }
}

size_t HeaderSize(const Frame& f) {
  size_t n = 2 + 4;        // SOI, DQT marker + length
  if (f.meta && !f.meta->strip) {
    for (const std::string& a : f.meta->app_data) n += 1 + a.size();
    for (const std::string& a : f.meta->com_data) n += 2 + a.size();
  } else {
    n += 18;               // the fixed APP0
  }
  if (f.meta) n += f.meta->tail_data.size();   // counted whether or not it is written (:291)
  for (size_t i = 0; i < f.quant.size(); ++i) n += 1 + (f.quant[i].precision ? 2 : 1) * 64;
  n += 10 + 3 * f.ncomp;   // SOF
  n += 4;                  // DHT without the code data
  n += 8 + 2 * f.ncomp;    // SOS
  n += 2;                  // EOI
  return n;
}

size_t EstimateDCSize(const Frame& f) {
  SymbolHistogram histo[3];
  BuildDCHistograms(f, histo);
  size_t num = f.ncomp;
  int indexes[3];
  uint8_t depths[3 * kHistoSize];
  return ClusterHistograms(histo, &num, indexes, depths);
}

// -------------------------------------------------------------------- bit packing ----
namespace {

struct CodeTable {
  uint8_t depth[256];
  int code[256];
};

// Canonical JPEG code from depths: symbols ordered by (depth, value); the last code of the
// reserved symbol is dropped (BuildHuffmanCode + BuildHuffmanCodeTable, :158-208).
void CanonicalCode(const uint8_t* depth, int counts[17], int values[kHistoSize],
                   CodeTable* table) {
  memset(counts, 0, sizeof(int) * 17);
  for (int i = 0; i < kHistoSize; ++i)
    if (depth[i] > 0) ++counts[depth[i]];
  int offset[17] = {0};
  for (int l = 1; l <= 16; ++l) offset[l] = offset[l - 1] + counts[l - 1];
  for (int i = 0; i < kHistoSize; ++i)
    if (depth[i] > 0) values[offset[depth[i]]++] = i;
  for (int j = 0; j < 256; ++j) table->depth[j] = 255;
  int total = 0;
  for (int l = 1; l <= 16; ++l) total += counts[l];
  if (total == 0) return;
  int code = 0, p = 0;
  for (int l = 1; l <= 16; ++l) {
    for (int i = 0; i < counts[l]; ++i, ++p) {
      if (p < total - 1) {
        table->depth[values[p]] = (uint8_t)l;
        table->code[values[p]] = code;
      }
      ++code;
    }
    code <<= 1;
  }
}

// 64-bit accumulator, MSB first, 0xff byte stuffing (BitWriter, jpeg_bit_writer.h:31-108).
class BitSink {
 public:
  explicit BitSink(std::string* out) : out_(out), acc_(0), free_(64) {}
  void Put(int nbits, uint64_t bits) {
    free_ -= nbits;
    acc_ |= bits << free_;
    if (free_ <= 16) {
      for (int s = 56; s >= 16; s -= 8) Byte((int)((acc_ >> s) & 0xff));
      acc_ <<= 48;
      free_ += 48;
    }
  }
  void Finish() {
    while (free_ <= 56) {
      Byte((int)((acc_ >> 56) & 0xff));
      acc_ <<= 8;
      free_ += 8;
    }
    if (free_ < 64) {
      const int pad = 0xff >> (64 - free_);
      Byte((int)(((acc_ >> 56) & ~(uint64_t)pad) | pad));
    }
    acc_ = 0;
    free_ = 64;
  }

 private:
  void Byte(int b) {
    out_->push_back((char)b);
    if (b == 0xff) out_->push_back((char)0);
  }
  std::string* out_;
  uint64_t acc_;
  int free_;
};

void PutBlock(const int16_t* blk, const CodeTable& dc, const CodeTable& ac, int* last_dc,
              BitSink* sink) {   // EncodeDCTBlockSequential, :446-497
  int16_t diff = (int16_t)(blk[0] - *last_dc);
  *last_dc = blk[0];
  int16_t mag = diff, low = diff;
  if (diff < 0) {
    mag = (int16_t)-diff;
    low = (int16_t)(diff - 1);
  }
  int nbits = BitLength((uint32_t)(int)mag);
  sink->Put(dc.depth[nbits], (uint64_t)dc.code[nbits]);
  if (nbits > 0) sink->Put(nbits, (uint64_t)(low & ((1 << nbits) - 1)));
  int run = 0;
  for (int k = 1; k < 64; ++k) {
    int16_t v = blk[kNaturalOrder[k]];
    if (v == 0) {
      ++run;
      continue;
    }
    int16_t bits_v;
    if (v < 0) {
      v = (int16_t)-v;
      bits_v = (int16_t)~v;
    } else {
      bits_v = v;
    }
    while (run > 15) {
      sink->Put(ac.depth[0xf0], (uint64_t)ac.code[0xf0]);
      run -= 16;
    }
    nbits = FloorLog2((uint32_t)(int)v) + 1;
    const int sym = (run << 4) + nbits;
    sink->Put(ac.depth[sym], (uint64_t)ac.code[sym]);
    sink->Put(nbits, (uint64_t)(bits_v & ((1 << nbits) - 1)));
    run = 0;
  }
  if (run > 0) sink->Put(ac.depth[0], (uint64_t)ac.code[0]);
}

inline void Push16(std::string* s, size_t v) {
  s->push_back((char)(v >> 8));
  s->push_back((char)(v & 0xff));
}

}  // namespace

bool BuildJpegHead(const Frame& f, const SymbolHistogram* dc_histo,
                   const SymbolHistogram* ac_histo, JpegHead* head) {
  std::string* out = &head->bytes;
  out->clear();
  const int nc = f.ncomp;
  head->ncomp = nc;
  memset(head->depth, 0, sizeof(head->depth));
  memset(head->code, 0, sizeof(head->code));
  out->push_back((char)0xff);
  out->push_back((char)0xd8);
  if (f.meta && !f.meta->strip) {   // EncodeMetadata (:52-72)
    for (const std::string& a : f.meta->app_data) {
      out->push_back((char)0xff);
      out->append(a);
    }
    for (const std::string& a : f.meta->com_data) {
      out->push_back((char)0xff);
      out->push_back((char)0xfe);
      out->append(a);
    }
  } else {
    static const unsigned char kApp0[] = {0xff, 0xe0, 0x00, 0x10, 0x4a, 0x46, 0x49, 0x46, 0x00,
                                          0x01, 0x01, 0x00, 0x00, 0x01, 0x00, 0x01, 0x00, 0x00};
    out->append((const char*)kApp0, sizeof(kApp0));
  }
  {  // DQT (EncodeDQT :74-98)
    size_t len = 2;
    for (size_t i = 0; i < f.quant.size(); ++i) len += 1 + (f.quant[i].precision ? 2 : 1) * 64;
    out->push_back((char)0xff);
    out->push_back((char)0xdb);
    Push16(out, len);
    for (size_t i = 0; i < f.quant.size(); ++i) {
      const QuantTable& t = f.quant[i];
      out->push_back((char)((t.precision << 4) + t.index));
      for (int k = 0; k < 64; ++k) {
        const int v = t.values[kNaturalOrder[k]];
        if (t.precision) out->push_back((char)(v >> 8));
        out->push_back((char)(v & 0xff));
      }
    }
  }
  {  // SOF1 (EncodeSOF :100-127)
    out->push_back((char)0xff);
    out->push_back((char)0xc1);
    Push16(out, 8 + 3 * nc);
    out->push_back((char)8);
    Push16(out, (size_t)f.height);
    Push16(out, (size_t)f.width);
    out->push_back((char)nc);
    for (int c = 0; c < nc; ++c) {
      out->push_back((char)f.comp_id[c]);
      out->push_back((char)((f.samp[c] << 4) | f.samp[c]));
      if (f.quant_idx[c] >= (int)f.quant.size()) return false;
      out->push_back((char)f.quant[f.quant_idx[c]].index);
    }
  }
  // Huffman codes (BuildAndEncodeHuffmanCodes :361-444)
  std::vector<SymbolHistogram> histo(dc_histo, dc_histo + nc);
  size_t num_dc = nc;
  int dc_index[4], ac_index[4];
  std::vector<uint8_t> depths((size_t)nc * kHistoSize);
  ClusterHistograms(histo.data(), &num_dc, dc_index, depths.data());
  histo.resize(num_dc);
  histo.insert(histo.end(), ac_histo, ac_histo + nc);
  depths.resize((num_dc + nc) * kHistoSize);
  size_t num_ac = nc;
  ClusterHistograms(&histo[num_dc], &num_ac, ac_index, &depths[num_dc * kHistoSize]);
  const int num_histo = (int)(num_dc + num_ac);
  histo.resize(num_histo);
  int total_symbols = 0;
  for (int i = 0; i < num_histo; ++i) total_symbols += histo[i].NumSymbols();
  out->push_back((char)0xff);
  out->push_back((char)0xc4);
  Push16(out, 2 + (size_t)num_histo * 17 + total_symbols);
  for (int i = 0; i < num_histo; ++i) {
    const bool is_dc = (size_t)i < num_dc;
    const int idx = is_dc ? i : i - (int)num_dc;
    int counts[17], values[kHistoSize] = {0};
    CodeTable table;
    CanonicalCode(&depths[(size_t)i * kHistoSize], counts, values, &table);
    for (int c = 0; c < nc; ++c) {
      if ((is_dc && dc_index[c] == idx) || (!is_dc && ac_index[c] == idx)) {
        for (int j = 0; j < 256; ++j) {
          head->depth[is_dc ? 0 : 1][c][j] = table.depth[j];
          head->code[is_dc ? 0 : 1][c][j] = table.depth[j] == 255 ? 0 : (uint16_t)table.code[j];
        }
      }
    }
    int max_len = 16;
    while (max_len > 0 && counts[max_len] == 0) --max_len;
    --counts[max_len];   // the reserved all-ones code is not announced
    int listed = 0;
    for (int l = 0; l <= max_len; ++l) listed += l ? counts[l] : 0;
    out->push_back((char)(is_dc ? i : idx + 0x10));
    for (int l = 1; l <= 16; ++l) out->push_back((char)counts[l]);
    for (int j = 0; j < listed; ++j) out->push_back((char)values[j]);
  }
  {  // SOS
    out->push_back((char)0xff);
    out->push_back((char)0xda);
    Push16(out, 6 + 2 * nc);
    out->push_back((char)nc);
    for (int c = 0; c < nc; ++c) {
      out->push_back((char)f.comp_id[c]);
      out->push_back((char)((dc_index[c] << 4) | ac_index[c]));
    }
    out->push_back((char)0);
    out->push_back((char)63);
    out->push_back((char)0);
  }
  return true;
}

bool WriteJpeg(const Frame& f, std::string* out) {
  const int nc = f.ncomp;
  SymbolHistogram dc_histo[3], ac_histo[3];
  BuildDCHistograms(f, dc_histo);
  BuildACHistograms(f, ac_histo);
  JpegHead head;
  if (!BuildJpegHead(f, dc_histo, ac_histo, &head)) return false;
  *out = head.bytes;
  CodeTable dc_table[3], ac_table[3];
  for (int c = 0; c < nc; ++c)
    for (int j = 0; j < 256; ++j) {
      dc_table[c].depth[j] = head.depth[0][c][j];
      dc_table[c].code[j] = head.code[0][c][j];
      ac_table[c].depth[j] = head.depth[1][c][j];
      ac_table[c].code[j] = head.code[1][c][j];
    }
  // scan (EncodeScan :499-536): MCU by MCU, per component its samp x samp blocks
  {
    BitSink sink(out);
    int last_dc[3] = {0, 0, 0};
    for (int my = 0; my < f.mcu_rows; ++my)
      for (int mx = 0; mx < f.mcu_cols; ++mx)
        for (int c = 0; c < nc; ++c)
          for (int iy = 0; iy < f.samp[c]; ++iy)
            for (int ix = 0; ix < f.samp[c]; ++ix) {
              const size_t b = (size_t)(my * f.samp[c] + iy) * f.cw[c] + (mx * f.samp[c] + ix);
              PutBlock(&f.coeffs[c][b * 64], dc_table[c], ac_table[c], &last_dc[c], &sink);
            }
    sink.Finish();
  }
  out->push_back((char)0xff);
  out->push_back((char)0xd9);
  if (f.meta && !f.meta->strip) out->append(f.meta->tail_data);
  return true;
}

}  // namespace guetzli_amd
