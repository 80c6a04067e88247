// Baseline-sequential JPEG entropy coding of the candidate image on the device: what
// BuildDCHistograms / BuildACHistograms (jpeg_data_writer.cc:241-275) count and what
// EncodeScan / EncodeDCTBlockSequential (:446-536) emit through the BitWriter
// (jpeg_bit_writer.h:31-108), for 4:4:4 frames (one block per component per MCU) and 4:2:0
// frames (2 x 2 luma blocks + one block of each chroma component per MCU, padded).
//
// The search evaluates ~150 candidates per image and needs the EXACT size of each one's
// JPEG (it feeds ScoreJPEG); the bytes themselves are only wanted for the winner.  The
// candidate's coefficients already live in HBM, so the scan is produced there:
//   k_jpeg_histograms   symbol statistics                    (host builds the Huffman codes)
//   k_jpeg_block_bits   bits per MCU under those codes       -> exclusive scan = bit offsets
//   k_jpeg_emit         every MCU writes its bits at its offset (MSB-first)
//   k_jpeg_count_ff     bytes equal to 0xFF (each costs one stuffed 0x00)
// One 64-lane wavefront per MCU at a time; lane = zig-zag position; runs of zeros come
// from a ballot over "coefficient != 0", bit positions inside the MCU from a wavefront
// prefix sum.  Integer work only; byte-exact by contract.
#pragma once
#include "gz_common.h"
#include "gz_kernels_block.h"   // GZ_CONST

namespace gz {

GZ_CONST unsigned char kNaturalOrderDev[64] = {   // kJPEGNaturalOrder, jpeg_data.h:62-73
  0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
  35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
  58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

GZ_DEVFN int bit_length(unsigned v) { return v == 0 ? 0 : 32 - __clz((int)v); }

// Geometry of the frame the coefficient arrays describe: per component a grid of real blocks
// (bw x bh, first block coff) and the number of blocks it has per MCU along each axis (samp;
// 4:4:4: 1,1,1 -- 4:2:0: 2,1,1).  A component's blocks in the JPEG cover whole MCUs
// (mcu_cols*samp x mcu_rows*samp); the blocks beyond the real grid are the padding
// OutputImage::SaveToJpegData writes (output_image.cc:386-404): all AC zero, DC = the DC of
// the block before it in raster order of the padded grid.
struct FrameGeom {
  int bw[3], bh[3], coff[3], samp[3];
  int mcu_cols, mcu_rows, ncomp;
};

// Blocks ("units") of one MCU in scan order: component, sub-block position.
GZ_DEVFN int geom_units_per_mcu(const FrameGeom& g) {
  int n = 0;
  for (int c = 0; c < g.ncomp; ++c) n += g.samp[c] * g.samp[c];
  return n;
}
GZ_DEVFN void geom_unit(const FrameGeom& g, int u, int* c, int* ix, int* iy) {
  int cc = 0;
  while (u >= g.samp[cc] * g.samp[cc]) { u -= g.samp[cc] * g.samp[cc]; ++cc; }
  *c = cc;
  *ix = u % g.samp[cc];
  *iy = u / g.samp[cc];
}

// Quantised DC of block (bx, by) of component c in the padded grid.
GZ_DEVFN int geom_dc(const int16_t* __restrict__ coeffs, const int* __restrict__ q,
                     const FrameGeom& g, int c, int bx, int by) {
  int rx = bx < g.bw[c] ? bx : g.bw[c] - 1, ry = by;
  if (by >= g.bh[c]) { rx = g.bw[c] - 1; ry = g.bh[c] - 1; }
  return (int)coeffs[((size_t)g.coff[c] + (size_t)ry * g.bw[c] + rx) * 64] / q[c * 64];
}

// What lane k contributes for one block.
struct LaneSyms {
  int zrl;      // number of 0xF0 symbols in front (run / 16)
  int sym;      // Huffman symbol, -1: nothing (zero AC coefficient)
  int nbits;    // extra bits that follow the symbol
  unsigned extra;
  int eob;      // lane 63 only: an end-of-block symbol closes the block
  int is_dc;
};

// The AC symbols of one block (lanes 1..63; lane 0 gets an empty entry marked is_dc) from
// this lane's quantised value v (0 everywhere for a padding block).
GZ_DEVFN LaneSyms lane_ac_symbols(int v, int lane) {
  const unsigned long long mask = __ballot(lane >= 1 && v != 0);
  LaneSyms s;
  s.zrl = 0; s.sym = -1; s.nbits = 0; s.extra = 0; s.eob = 0; s.is_dc = lane == 0;
  if (lane >= 1 && v != 0) {
    const unsigned long long below = mask & ((1ull << lane) - 1ull);
    const int prev = below ? 63 - __clzll((long long)below) : 0;
    const int run = lane - prev - 1;
    const int mag = v < 0 ? -v : v;
    const int bits_v = v < 0 ? ~mag : mag;
    s.zrl = run >> 4;
    s.nbits = bit_length((unsigned)mag);
    int sym = ((run & 15) << 4) + s.nbits;
    s.sym = sym > 255 ? 255 : sym;   // unreachable for 8-bit image data (|coeff| < 2^15)
    s.extra = (unsigned)bits_v & ((1u << s.nbits) - 1u);
  }
  if (lane == 63) {
    const int last = mask ? 63 - __clzll((long long)mask) : 0;
    s.eob = last < 63;
  }
  return s;
}

GZ_DEVFN void lane_set_dc(LaneSyms* s, int dc, int prev) {
  const int diff = (int)(short)(dc - prev);   // int16 arithmetic of the writer (:448-455)
  const int mag = diff < 0 ? -diff : diff;
  const int low = diff < 0 ? diff - 1 : diff;
  s->nbits = bit_length((unsigned)mag);
  s->sym = s->nbits;
  s->extra = (unsigned)low & ((1u << s->nbits) - 1u);
}

// coeffs: dequantised blocks of the frame; q: int[3][64].  quantised value = coeff / q (C++
// `/`, OutputImage::SaveToJpegData, output_image.cc:395-400).  Unit (c, ix, iy) of MCU
// (mx, my); the DC difference is taken against the component's previous block in scan order
// (EncodeScan, jpeg_data_writer.cc:499-536; BuildDCHistograms, :241-265).
GZ_DEVFN LaneSyms lane_symbols(const int16_t* __restrict__ coeffs, const int* __restrict__ q,
                               const FrameGeom& g, int c, int ix, int iy, int mx, int my, int lane) {
  const int sp = g.samp[c];
  const int bx = mx * sp + ix, by = my * sp + iy;
  const bool real = bx < g.bw[c] && by < g.bh[c];
  const int nat = kNaturalOrderDev[lane];
  int v = 0;
  if (real) v = (int)coeffs[((size_t)g.coff[c] + (size_t)by * g.bw[c] + bx) * 64 + nat] / q[c * 64 + nat];
  LaneSyms s = lane_ac_symbols(v, lane);
  if (lane == 0) {
    const int dc = real ? v : geom_dc(coeffs, q, g, c, bx, by);
    int prev = 0;
    if (ix > 0) prev = geom_dc(coeffs, q, g, c, bx - 1, by);
    else if (iy > 0) prev = geom_dc(coeffs, q, g, c, mx * sp + sp - 1, by - 1);
    else if (mx > 0) prev = geom_dc(coeffs, q, g, c, mx * sp - 1, my * sp + sp - 1);
    else if (my > 0) prev = geom_dc(coeffs, q, g, c, g.mcu_cols * sp - 1, my * sp - 1);
    lane_set_dc(&s, dc, prev);
  }
  return s;
}

// ------------------------------------------------------------------- histograms ------
// hist: uint32 [2][3][256] (DC, AC) x component, raw occurrence counts; zeroed by the
// caller.  Persistent workgroups of four wavefronts (each wavefront strides over block
// positions, one block at a time) sharing one LDS histogram: few workgroups keep the final
// merge short (its global atomics all land on the same ~300 counters), four wavefronts per
// workgroup keep enough loads in flight to hide their latency.
constexpr int kHistWaves = 4;

__global__ __launch_bounds__(64 * kHistWaves) void k_jpeg_histograms(const int16_t* __restrict__ coeffs,
                                                                     const int* __restrict__ q, FrameGeom g,
                                                                     unsigned* __restrict__ hist) {
  __shared__ unsigned s_hist[2 * 3 * 256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2 * 3 * 256; i += 64 * kHistWaves) s_hist[i] = 0;
  __syncthreads();
  const int nmcu = g.mcu_cols * g.mcu_rows, upm = geom_units_per_mcu(g);
  // the same trip count for the four wavefronts (a wavefront past the end repeats the last
  // MCU and drops the result): the lane exchanges inside lane_symbols stay workgroup-uniform
  for (int m0 = blockIdx.x * kHistWaves; m0 < nmcu; m0 += gridDim.x * kHistWaves) {
    const bool live = m0 + wv < nmcu;
    const int m = live ? m0 + wv : nmcu - 1;
    const int mx = m % g.mcu_cols, my = m / g.mcu_cols;
    for (int u = 0; u < upm; ++u) {
      int c, ix, iy;
      geom_unit(g, u, &c, &ix, &iy);
      const LaneSyms s = lane_symbols(coeffs, q, g, c, ix, iy, mx, my, lane);
      if (!live) continue;
      unsigned* h = &s_hist[((s.is_dc ? 0 : 1) * 3 + c) * 256];
      if (s.zrl) atomicAdd(&h[0xf0], (unsigned)s.zrl);
      if (s.sym >= 0) atomicAdd(&h[s.sym], 1u);
      if (s.eob) atomicAdd(&s_hist[(3 + c) * 256], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * 3 * 256; i += 64 * kHistWaves)
    if (s_hist[i]) atomicAdd(&hist[i], s_hist[i]);
}

// --------------------------------------------------------------------- bit counts ----
struct JpegCodes {          // device pointers; [2][3][256]: (DC, AC) x component
  const unsigned char* depth;
  const unsigned short* code;
};

// Inclusive prefix sum over the 64 lanes.
GZ_DEVFN int wave_inclusive_sum(int v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(v, d);
    if (lane >= d) v += o;
  }
  return v;
}

// The two kernels that walk the coefficients (bits per MCU here, k_jpeg_emit below) are bound by
// the instructions a wavefront issues, and the entropy coder runs beside the Compare chain: every
// issue slot it takes is one the chain waits for.  Round 2's versions spent some 300 instructions
// per 8x8 block (an integer division per coefficient, two more for the DC prediction on lane 0,
// 64-bit index arithmetic, a six-step shuffle scan per block, the frame's geometry interpreted at
// run time); these do the same work with
//   * the MCU's shape as a template constant (UPM = blocks per MCU: 1 luma alone, 3 4:4:4,
//     6 4:2:0): the loop over the blocks is unrolled, their loads are issued together;
//   * coefficient / q as one float multiply and an exact integer correction (quant_div);
//   * everything uniform across a wavefront (the DC quantisers, the ZRL and EOB code lengths, the
//     lane's zig-zag position and quantisers) loaded once per wavefront, which then serves
//     kMcuPerWave consecutive MCUs;
//   * ONE reduction per MCU in the counting kernel (the blocks' lengths are summed per lane first);
//   * a symbol's code and its extra bits written as one value (at most 32 bits).
// 4K, 4:4:4: counting 133 -> 72 us, emitting 208 -> 149 us (profiles/r03_chain_kernel_experiments.log).
constexpr int kMcuPerWave = 4;

// trunc(a / q) for |a| <= 32768, q >= 1, rq = 1.0f / q: the float product is within 2^-8 of the
// quotient, so its truncation is off by at most one; the remainder says which way.
GZ_DEVFN int quant_div(int a, int q, float rq) {
  const int m = a < 0 ? -a : a;
  int k = (int)((float)m * rq);
  const int r = m - k * q;
  k += r >= q ? 1 : 0;
  k -= r < 0 ? 1 : 0;
  return a < 0 ? -k : k;
}

template <int UPM> struct McuShape {
  static constexpr int kComps = UPM == 1 ? 1 : 3;
  GZ_DEVFN static constexpr int comp(int u) { return UPM == 6 ? (u < 4 ? 0 : u - 3) : u; }
  GZ_DEVFN static constexpr int ix(int u) { return UPM == 6 && u < 4 ? (u & 1) : 0; }
  GZ_DEVFN static constexpr int iy(int u) { return UPM == 6 && u < 4 ? (u >> 1) : 0; }
  GZ_DEVFN static constexpr int samp(int c) { return UPM == 6 && c == 0 ? 2 : 1; }
};

// What a wavefront keeps for all its MCUs.
template <int UPM> struct WaveTables {
  int nat;               // this lane's natural index
  int q[3];              // this lane's quantisers, per component
  float rq[3];
  int q0[3];             // the DC quantisers (uniform)
  float rq0[3];
  int zrl_len[3], eob_len[3];
  unsigned long long lt; // lanes below this one
};

template <int UPM>
GZ_DEVFN void wave_tables_load(const int* __restrict__ q, const unsigned char* __restrict__ depth,
                               int lane, WaveTables<UPM>* t) {
  t->nat = kNaturalOrderDev[lane];
  t->lt = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
  for (int c = 0; c < McuShape<UPM>::kComps; ++c) {
    t->q[c] = q[c * 64 + t->nat];
    t->rq[c] = 1.0f / (float)t->q[c];
    t->q0[c] = q[c * 64];
    t->rq0[c] = 1.0f / (float)t->q0[c];
    t->zrl_len[c] = depth[(3 + c) * 256 + 0xf0];
    t->eob_len[c] = depth[(3 + c) * 256];
  }
}

// The raw (dequantised) values one block needs: this lane's coefficient, the block's DC as the
// writer sees it (padding blocks repeat the DC before them, geom_dc) and the DC it is predicted
// from (0: the component's first block).
struct UnitRaw {
  int v, dc, prev;
};

template <int UPM>
GZ_DEVFN UnitRaw unit_load(const int16_t* __restrict__ coeffs, const FrameGeom& g, int u, int mx,
                           int my, int nat) {
  typedef McuShape<UPM> S;
  const int c = S::comp(u), sp = S::samp(c), ix = S::ix(u), iy = S::iy(u);
  const int bw = g.bw[c], bh = g.bh[c];
  const int16_t* base = coeffs + (size_t)g.coff[c] * 64;
  const int bx = mx * sp + ix, by = my * sp + iy;
  auto dc_of = [&](int x, int y) {   // geom_dc's clamping, without the division
    int rx = x < bw ? x : bw - 1, ry = y;
    if (y >= bh) { rx = bw - 1; ry = bh - 1; }
    return (int)base[(unsigned)(ry * bw + rx) * 64u];
  };
  UnitRaw r;
  const bool real = bx < bw && by < bh;
  r.v = real ? (int)base[(unsigned)(by * bw + bx) * 64u + (unsigned)nat] : 0;
  r.dc = dc_of(bx, by);
  // the component's previous block in scan order (EncodeScan, jpeg_data_writer.cc:499-536)
  int px = 0, py = 0;
  bool has = true;
  if (ix > 0) { px = bx - 1; py = by; }
  else if (iy > 0) { px = mx * sp + sp - 1; py = by - 1; }
  else if (mx > 0) { px = mx * sp - 1; py = my * sp + sp - 1; }
  else if (my > 0) { px = g.mcu_cols * sp - 1; py = my * sp - 1; }
  else has = false;
  r.prev = has ? dc_of(px, py) : 0;
  return r;
}

// The block's symbols from the raw values (lane_symbols above, on what unit_load fetched).
template <int UPM>
GZ_DEVFN LaneSyms unit_symbols(const UnitRaw& r, const WaveTables<UPM>& t, int c, int lane) {
  const int v = quant_div(r.v, t.q[c], t.rq[c]);
  const bool nz = lane >= 1 && v != 0;
  const unsigned long long mask = __ballot(nz);
  LaneSyms s;
  s.zrl = 0; s.sym = -1; s.nbits = 0; s.extra = 0; s.eob = 0; s.is_dc = lane == 0;
  if (nz) {
    const unsigned long long below = mask & t.lt;
    const int prev = below ? 63 - __clzll((long long)below) : 0;
    const int run = lane - prev - 1;
    const int mag = v < 0 ? -v : v;
    const int bits_v = v < 0 ? ~mag : mag;
    s.zrl = run >> 4;
    s.nbits = bit_length((unsigned)mag);
    const int sym = ((run & 15) << 4) + s.nbits;
    s.sym = sym > 255 ? 255 : sym;   // unreachable for 8-bit image data (|coeff| < 2^15)
    s.extra = (unsigned)bits_v & ((1u << s.nbits) - 1u);
  }
  if (lane == 63) s.eob = mask ? (63 - __clzll((long long)mask)) < 63 : 1;
  if (lane == 0)
    lane_set_dc(&s, quant_div(r.dc, t.q0[c], t.rq0[c]), quant_div(r.prev, t.q0[c], t.rq0[c]));
  return s;
}

// kMcuWaves wavefronts per workgroup: with a workgroup per MCU the 129 600 single-wavefront
// workgroups of a 4K frame were bound by the rate at which workgroups can be dispatched (49 us =
// one per clock), not by what they compute.
constexpr int kMcuWaves = 4;

// bits[m] = number of scan bits of MCU m.
template <int UPM, int W>
__global__ __launch_bounds__(64 * W) void k_jpeg_block_bits(const int16_t* __restrict__ coeffs,
                                                        const int* __restrict__ q, FrameGeom g,
                                                        JpegCodes codes, unsigned* __restrict__ bits) {
  typedef McuShape<UPM> S;
  const int lane = threadIdx.x & 63;
  const int nmcu = g.mcu_cols * g.mcu_rows;
  const int m0 = (blockIdx.x * W + (int)(threadIdx.x >> 6)) * kMcuPerWave;
  if (m0 >= nmcu) return;   // (a whole wavefront)
  WaveTables<UPM> t;
  wave_tables_load<UPM>(q, codes.depth, lane, &t);
  int mx = m0 % g.mcu_cols, my = m0 / g.mcu_cols;
  for (int m = m0; m < m0 + kMcuPerWave && m < nmcu; ++m) {
    UnitRaw raw[UPM];
#pragma unroll
    for (int u = 0; u < UPM; ++u) raw[u] = unit_load<UPM>(coeffs, g, u, mx, my, t.nat);
    int len = 0;
#pragma unroll
    for (int u = 0; u < UPM; ++u) {
      const int c = S::comp(u);
      const LaneSyms s = unit_symbols<UPM>(raw[u], t, c, lane);
      if (s.sym >= 0) {
        const unsigned char* d = codes.depth + (s.is_dc ? c : 3 + c) * 256;
        len += s.zrl * t.zrl_len[c] + d[s.sym] + s.nbits;
      }
      if (s.eob) len += t.eob_len[c];
    }
    const int total = __shfl(wave_inclusive_sum(len, lane), 63);
    if (lane == 0) bits[m] = (unsigned)total;
    if (++mx == g.mcu_cols) { mx = 0; ++my; }
  }
}

// The symbol statistics with the same machinery (round 5): the MCU's shape as a template constant, the
// blocks' loads issued together, coefficient / q as quant_div instead of an integer division per lane
// (and two more on lane 0 for the DC prediction), the frame's geometry not interpreted per block.
// k_jpeg_histograms above (kept: the tests' reference for this one on small frames) took 162 us for a 4K
// frame -- on the host's path once per quantisation trial and at the start of phase B.
template <int UPM>
__global__ __launch_bounds__(64 * kHistWaves) void k_jpeg_histograms_t(const int16_t* __restrict__ coeffs,
                                                                       const int* __restrict__ q, FrameGeom g,
                                                                       unsigned* __restrict__ hist) {
  typedef McuShape<UPM> S;
  __shared__ unsigned s_hist[2 * 3 * 256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2 * 3 * 256; i += 64 * kHistWaves) s_hist[i] = 0;
  __syncthreads();
  WaveTables<UPM> t;
  t.nat = kNaturalOrderDev[lane];
  t.lt = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
  for (int c = 0; c < S::kComps; ++c) {
    t.q[c] = q[c * 64 + t.nat];
    t.rq[c] = 1.0f / (float)t.q[c];
    t.q0[c] = q[c * 64];
    t.rq0[c] = 1.0f / (float)t.q0[c];
    t.zrl_len[c] = 0;
    t.eob_len[c] = 0;
  }
  const int nmcu = g.mcu_cols * g.mcu_rows;
  // (the same trip count for the four wavefronts: a wavefront past the end repeats the last MCU and
  // drops the result, so that unit_symbols' ballots stay whole-wavefront operations)
  for (int m0 = blockIdx.x * kHistWaves; m0 < nmcu; m0 += gridDim.x * kHistWaves) {
    const bool live = m0 + wv < nmcu;
    const int m = live ? m0 + wv : nmcu - 1;
    const int mx = m % g.mcu_cols, my = m / g.mcu_cols;
    UnitRaw raw[UPM];
#pragma unroll
    for (int u = 0; u < UPM; ++u) raw[u] = unit_load<UPM>(coeffs, g, u, mx, my, t.nat);
#pragma unroll
    for (int u = 0; u < UPM; ++u) {
      const int c = S::comp(u);
      const LaneSyms s = unit_symbols<UPM>(raw[u], t, c, lane);
      if (!live) continue;
      unsigned* h = &s_hist[((s.is_dc ? 0 : 1) * 3 + c) * 256];
      if (s.zrl) atomicAdd(&h[0xf0], (unsigned)s.zrl);
      if (s.sym >= 0) atomicAdd(&h[s.sym], 1u);
      if (s.eob) atomicAdd(&s_hist[(3 + c) * 256], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * 3 * 256; i += 64 * kHistWaves)
    if (s_hist[i]) atomicAdd(&hist[i], s_hist[i]);
}

// off[0..n] = exclusive prefix sums of bits[0..n) (64-bit), in one pass over the array by
// ceil(n / 2048) workgroups ("decoupled look-back"): a workgroup scans its tile of 2048 values
// (8 per thread, two 16-byte loads), publishes the tile's sum, then adds up what the tiles
// before it have published -- their sums, or from the nearest one that already knows it the
// inclusive prefix -- 64 predecessors at a time, one per lane.  Tiles are taken in ticket order
// (a workgroup never waits for one that has not started), the flags carry the launch's epoch so
// that nothing has to be cleared between launches.  A value is at most an MCU's scan bits or a
// block's candidate count: a tile's sum stays far below 2^31; prefixes are 64-bit.
// (Round 1 walked the array with ONE workgroup of 1024: 22-36 us for the 32 400 MCUs of a
// 1080p image, twice per phase-B iteration on its critical path.)
constexpr int kScanTile = 2048;

struct ScanState {             // device scratch of one context: ceil(n_max / kScanTile) tiles
  unsigned long long* agg;     // sum of tile t
  unsigned long long* incl;    // inclusive prefix up to and including tile t
  unsigned* status;            // epoch << 2 | {1: agg valid, 2: incl valid too}
  unsigned* ticket;
};

struct alignas(16) ScanU4 { unsigned x, y, z, w; };

GZ_DEVFN void scan_load4(const unsigned* __restrict__ bits, int nb, int at, unsigned v[4]) {
  if (at + 3 < nb) {
    const ScanU4 u = *reinterpret_cast<const ScanU4*>(bits + at);
    v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = at + e < nb ? bits[at + e] : 0u;
  }
}

__global__ __launch_bounds__(256) void k_scan_offsets(const unsigned* __restrict__ bits, int n,
                                                      unsigned long long* __restrict__ off,
                                                      ScanState st, unsigned epoch) {
  __shared__ unsigned s_tile;
  __shared__ int wave_total[4];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int ntiles = (int)gridDim.x;
  if (t == 0) {
    s_tile = atomicAdd(st.ticket, 1u);
    if (s_tile == (unsigned)(ntiles - 1)) *st.ticket = 0u;   // every ticket is taken: ready for the next launch
  }
  __syncthreads();
  const int tile = (int)s_tile;
  const int at = tile * kScanTile + 8 * t;
  unsigned v[8];
  scan_load4(bits, n, at, v);
  scan_load4(bits, n, at + 4, v + 4);
  int loc[8];
  int run = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) { run += (int)v[e]; loc[e] = run; }   // inclusive within the thread
  const int inc = wave_inclusive_sum(run, lane);
  if (lane == 63) wave_total[wv] = inc;
  __syncthreads();
  int before = 0, all = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int w = wave_total[k];
    if (k < wv) before += w;
    all += w;
  }
  if (t == 0) {
    st.agg[tile] = (unsigned long long)all;
    if (tile == 0) st.incl[0] = (unsigned long long)all;
    GZ_STORE_RELEASE(&st.status[tile], (epoch << 2) | (tile == 0 ? 2u : 1u));
  }
  // look-back: every wavefront does the same (lane = one predecessor), thread 0 publishes
  unsigned long long prefix = 0;
  for (int hi = tile - 1; hi >= 0;) {
    const int p = hi - lane;
    unsigned stt = 0;
    if (p >= 0) {
      do { stt = GZ_LOAD_ACQUIRE(&st.status[p]); } while ((stt >> 2) != epoch || (stt & 3u) == 0u);
    }
    const unsigned long long has_incl = __ballot(p >= 0 && (stt & 3u) == 2u);
    // lanes nearer than the nearest tile with an inclusive prefix contribute their sums
    int stop = 64;
    if (has_incl) {
      stop = 0;
      while (!((has_incl >> stop) & 1ull)) ++stop;
    }
    unsigned long long mine = 0;
    if (p >= 0 && lane < stop) mine = st.agg[p];
    else if (p >= 0 && lane == stop) mine = st.incl[p];
    // 64-lane sum of 64-bit values through two 32-bit halves
    unsigned lo32 = (unsigned)mine, hi32 = (unsigned)(mine >> 32);
    unsigned long long sum = 0;
    {
      // wave_inclusive_sum works on int: sum the 16-bit quarters separately (no overflow)
      const int q0 = (int)(lo32 & 0xffffu), q1 = (int)(lo32 >> 16), q2 = (int)(hi32 & 0xffffu), q3 = (int)(hi32 >> 16);
      const unsigned long long s0 = (unsigned)__shfl(wave_inclusive_sum(q0, lane), 63);
      const unsigned long long s1 = (unsigned)__shfl(wave_inclusive_sum(q1, lane), 63);
      const unsigned long long s2 = (unsigned)__shfl(wave_inclusive_sum(q2, lane), 63);
      const unsigned long long s3 = (unsigned)__shfl(wave_inclusive_sum(q3, lane), 63);
      sum = s0 + (s1 << 16) + (s2 << 32) + (s3 << 48);
    }
    prefix += sum;
    if (has_incl) break;
    hi -= 64;
  }
  if (t == 0 && tile > 0) {
    st.incl[tile] = prefix + (unsigned long long)all;
    GZ_STORE_RELEASE(&st.status[tile], (epoch << 2) | 2u);
  }
  const unsigned long long excl = prefix + (unsigned long long)(before + inc - run);
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if (at + e < n) off[at + e] = excl + (unsigned long long)(loc[e] - (int)v[e]);
  if (tile == ntiles - 1 && t == 255) off[n] = prefix + (unsigned long long)all;
}

// --------------------------------------------------------------------------- emit ----
// The bit buffer is an array of 32-bit words, stream bit p at word p/32, bit 31-(p%32)
// (so that stream byte j = (word[j/4] >> (24 - 8*(j%4))) & 0xff).  It must be zero before
// the launch.  Each MCU assembles its bits in LDS and ORs whole words out (the first and
// last word of an MCU are shared with its neighbours, hence atomicOr); MCUs too long for
// the staging buffer OR their pieces straight into global memory.
constexpr int kStageWords = 256;

GZ_DEVFN void or_bits(unsigned* words, unsigned long long pos, unsigned value, int len) {
  if (len == 0) return;
  const unsigned long long w = pos >> 5;
  const int sh = (int)(pos & 31);
  const unsigned long long x = (unsigned long long)value << (64 - len - sh);
  const unsigned hi = (unsigned)(x >> 32), lo = (unsigned)x;
  if (hi) atomicOr(&words[w], hi);
  if (lo) atomicOr(&words[w + 1], lo);
}

// `value`'s low `len` (1..32) bits at stream position pos: at most two words.
GZ_DEVFN void or_bits32(unsigned* words, unsigned long long pos, unsigned value, int len) {
  const unsigned long long w = pos >> 5;
  const int sh = (int)(pos & 31);
  const unsigned long long x = (unsigned long long)value << (64 - len - sh);
  const unsigned hi = (unsigned)(x >> 32), lo = (unsigned)x;
  if (hi) atomicOr(&words[w], hi);
  if (lo) atomicOr(&words[w + 1], lo);
}

template <int UPM, int W>
__global__ __launch_bounds__(64 * W) void k_jpeg_emit(const int16_t* __restrict__ coeffs,
                                                  const int* __restrict__ q, FrameGeom g,
                                                  JpegCodes codes,
                                                  const unsigned long long* __restrict__ off,
                                                  unsigned* __restrict__ words,
                                                  unsigned long long cap_words) {
  typedef McuShape<UPM> S;
  // every wavefront has its own staging area (no workgroup barrier: the wavefronts of a
  // workgroup have nothing to do with each other)
  __shared__ unsigned stage_all[W][kStageWords + 2];
  unsigned* stage = stage_all[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
  const int nmcu = g.mcu_cols * g.mcu_rows;
  const int m0 = (blockIdx.x * W + (int)(threadIdx.x >> 6)) * kMcuPerWave;
  if (m0 >= nmcu) return;   // (a whole wavefront)
  WaveTables<UPM> t;
  wave_tables_load<UPM>(q, codes.depth, lane, &t);
  int mx = m0 % g.mcu_cols, my = m0 / g.mcu_cols;
  for (int m = m0; m < m0 + kMcuPerWave && m < nmcu; ++m) {
    const unsigned long long start = off[m], end = off[m + 1];
    // the last MCU also writes the 1-padding up to the byte boundary (BitWriter::JumpToByteBoundary)
    const int pad = m == nmcu - 1 ? (int)((8 - (end & 7)) & 7) : 0;
    // `words` is sized for valid code lengths (<= 16 bits); with anything else the scan can be
    // longer, the host reports that afterwards, and nothing is written past the buffer here
    if (((end + pad + 63) >> 5) + 1 > cap_words) return;
    UnitRaw raw[UPM];
#pragma unroll
    for (int u = 0; u < UPM; ++u) raw[u] = unit_load<UPM>(coeffs, g, u, mx, my, t.nat);
    const unsigned long long word0 = start >> 5;
    const unsigned long long span = (end + pad) - (word0 << 5);   // bits from word0's first bit
    const bool staged = span <= (unsigned long long)kStageWords * 32;
    const int nwords = (int)((span + 31) >> 5);
    if (staged) {
      for (int i = lane; i < nwords + 1; i += 64) stage[i] = 0;
      GZ_WAVE_SYNC();
    }
    unsigned* dst = staged ? stage : words;
    unsigned long long base = staged ? start - (word0 << 5) : start;
#pragma unroll
    for (int u = 0; u < UPM; ++u) {
      const int c = S::comp(u);
      const LaneSyms s = unit_symbols<UPM>(raw[u], t, c, lane);
      const int tab = (s.is_dc ? c : 3 + c) * 256;
      int dl = 0;
      unsigned cd = 0;
      if (s.sym >= 0) {
        dl = codes.depth[tab + s.sym];
        cd = codes.code[tab + s.sym];
      }
      int len = s.sym >= 0 ? s.zrl * t.zrl_len[c] + dl + s.nbits : 0;
      if (s.eob) len += t.eob_len[c];
      const int incl = wave_inclusive_sum(len, lane);
      const int total = __shfl(incl, 63);
      unsigned long long pos = base + (unsigned long long)(incl - len);
      if (s.sym >= 0) {
        if (s.zrl) {
          const unsigned zc = codes.code[(3 + c) * 256 + 0xf0];
          for (int z = 0; z < s.zrl; ++z) {
            or_bits(dst, pos, zc, t.zrl_len[c]);
            pos += t.zrl_len[c];
          }
        }
        // the code and the extra bits that follow it as one value
        if (dl + s.nbits <= 32) {
          if (dl + s.nbits > 0) or_bits32(dst, pos, (cd << s.nbits) | s.extra, dl + s.nbits);
        } else {   // (code lengths no JPEG has: the host refuses the result)
          or_bits(dst, pos, cd, dl);
          or_bits(dst, pos + dl, s.extra, s.nbits);
        }
        pos += dl + s.nbits;
      }
      if (s.eob) or_bits(dst, pos, codes.code[(3 + c) * 256], t.eob_len[c]);
      base += (unsigned long long)total;
    }
    if (pad && lane == 0) or_bits(dst, base, (1u << pad) - 1u, pad);
    if (staged) {
      GZ_WAVE_SYNC();
      // whole words of the MCU's own are stored; the first and the last are shared with the
      // neighbours (the buffer is zero before the launch)
      for (int i = lane; i < nwords; i += 64) {
        const unsigned x = stage[i];
        if (i == 0 || i == nwords - 1) { if (x) atomicOr(&words[word0 + i], x); }
        else words[word0 + i] = x;
      }
      GZ_WAVE_SYNC();   // (the staging area is cleared for the next MCU)
    }
    if (++mx == g.mcu_cols) { mx = 0; ++my; }
  }
}

// The scan's size is only known on the device (off[nb], the scan's bit count) when the words
// are cleared and counted: both kernels read it there, so that one gz_jpeg_scan needs a single
// host synchronisation.
// Clears the words the scan will occupy (k_jpeg_emit ORs into them) and the 0xFF counter.
__global__ __launch_bounds__(256) void k_jpeg_clear_words(unsigned* __restrict__ words,
                                                          const unsigned long long* __restrict__ total_bits,
                                                          unsigned long long cap_words,
                                                          unsigned long long* __restrict__ count) {
  const unsigned long long nbytes = (*total_bits + 7) >> 3;
  unsigned long long nwords = (nbytes >> 2) + 4;
  if (nwords > cap_words) nwords = cap_words;
  for (unsigned long long w = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; w < nwords;
       w += (unsigned long long)gridDim.x * blockDim.x)
    words[w] = 0u;
  if (blockIdx.x == 0 && threadIdx.x == 0) *count = 0ull;
}

// Number of stream bytes equal to 0xFF, added to *count.
__global__ __launch_bounds__(256) void k_jpeg_count_ff(const unsigned* __restrict__ words,
                                                       const unsigned long long* __restrict__ total_bits,
                                                       unsigned long long* __restrict__ count) {
  __shared__ unsigned s_cnt[256];
  const unsigned long long nbytes = (*total_bits + 7) >> 3;
  const unsigned long long nwords = (nbytes + 3) >> 2;
  unsigned n = 0;
  for (unsigned long long w = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; w < nwords;
       w += (unsigned long long)gridDim.x * blockDim.x) {
    const unsigned v = words[w];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * w + j < nbytes && ((v >> (24 - 8 * j)) & 0xffu) == 0xffu) ++n;
  }
  s_cnt[threadIdx.x] = n;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) s_cnt[threadIdx.x] += s_cnt[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0 && s_cnt[0]) atomicAdd(count, (unsigned long long)s_cnt[0]);
}

}  // namespace gz
