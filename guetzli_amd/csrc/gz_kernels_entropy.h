// Baseline-sequential JPEG entropy coding of the candidate image on the device: what
// BuildDCHistograms / BuildACHistograms (jpeg_data_writer.cc:241-275) count and what
// EncodeScan / EncodeDCTBlockSequential (:446-536) emit through the BitWriter
// (jpeg_bit_writer.h:31-108), for a 4:4:4 frame (one block per component per MCU).
//
// The search evaluates ~150 candidates per image and needs the EXACT size of each one's
// JPEG (it feeds ScoreJPEG); the bytes themselves are only wanted for the winner.  The
// candidate's coefficients already live in HBM, so the scan is produced there:
//   k_jpeg_histograms   symbol statistics                    (host builds the Huffman codes)
//   k_jpeg_block_bits   bits per MCU under those codes       -> exclusive scan = bit offsets
//   k_jpeg_emit         every MCU writes its bits at its offset (MSB-first)
//   k_jpeg_count_ff     bytes equal to 0xFF (each costs one stuffed 0x00)
// One 64-lane wavefront per block position; lane = zig-zag position; runs of zeros come
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

// What lane k contributes for component c of block b.
struct LaneSyms {
  int zrl;      // number of 0xF0 symbols in front (run / 16)
  int sym;      // Huffman symbol, -1: nothing (zero AC coefficient)
  int nbits;    // extra bits that follow the symbol
  unsigned extra;
  int eob;      // lane 63 only: an end-of-block symbol closes the block
  int is_dc;
};

// coeffs: dequantised [3][nb][64]; q: int[3][64].  quantised value = coeff / q (C++ `/`,
// FrameFromImage == OutputImage::SaveToJpegData, output_image.cc:348-409).
GZ_DEVFN LaneSyms lane_symbols(const int16_t* __restrict__ coeffs, const int* __restrict__ q,
                               int nb, int c, int b, int lane) {
  const int nat = kNaturalOrderDev[lane];
  const int16_t* blk = coeffs + ((size_t)c * nb + b) * 64;
  const int v = (int)blk[nat] / q[c * 64 + nat];
  const unsigned long long mask = __ballot(lane >= 1 && v != 0);
  LaneSyms s;
  s.zrl = 0; s.sym = -1; s.nbits = 0; s.extra = 0; s.eob = 0; s.is_dc = lane == 0;
  if (lane == 0) {
    const int prev = b > 0 ? (int)blk[-64] / q[c * 64] : 0;
    const int diff = (int)(short)(v - prev);   // int16 arithmetic of the writer (:448-455)
    const int mag = diff < 0 ? -diff : diff;
    const int low = diff < 0 ? diff - 1 : diff;
    s.nbits = bit_length((unsigned)mag);
    s.sym = s.nbits;
    s.extra = (unsigned)low & ((1u << s.nbits) - 1u);
  } else if (v != 0) {
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

// ------------------------------------------------------------------- histograms ------
// hist: uint32 [2][3][256] (DC, AC) x component, raw occurrence counts; zeroed by the
// caller.  Persistent workgroups of four wavefronts (each wavefront strides over block
// positions, one block at a time) sharing one LDS histogram: few workgroups keep the final
// merge short (its global atomics all land on the same ~300 counters), four wavefronts per
// workgroup keep enough loads in flight to hide their latency.
constexpr int kHistWaves = 4;

__global__ __launch_bounds__(64 * kHistWaves) void k_jpeg_histograms(const int16_t* __restrict__ coeffs,
                                                                     const int* __restrict__ q, int nb,
                                                                     unsigned* __restrict__ hist) {
  __shared__ unsigned s_hist[2 * 3 * 256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2 * 3 * 256; i += 64 * kHistWaves) s_hist[i] = 0;
  __syncthreads();
  // the same trip count for the four wavefronts (a wavefront past the end repeats the last
  // block and drops the result): the lane exchanges inside lane_symbols stay workgroup-uniform
  for (int b0 = blockIdx.x * kHistWaves; b0 < nb; b0 += gridDim.x * kHistWaves) {
    const bool live = b0 + wv < nb;
    const int b = live ? b0 + wv : nb - 1;
    LaneSyms s[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) s[c] = lane_symbols(coeffs, q, nb, c, b, lane);
    if (!live) continue;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      unsigned* h = &s_hist[((s[c].is_dc ? 0 : 1) * 3 + c) * 256];
      if (s[c].zrl) atomicAdd(&h[0xf0], (unsigned)s[c].zrl);
      if (s[c].sym >= 0) atomicAdd(&h[s[c].sym], 1u);
      if (s[c].eob) atomicAdd(&s_hist[(3 + c) * 256], 1u);
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

GZ_DEVFN int lane_bits(const LaneSyms& s, const unsigned char* depth_dc,
                       const unsigned char* depth_ac) {
  int len = 0;
  if (s.sym >= 0) {
    const unsigned char* d = s.is_dc ? depth_dc : depth_ac;
    len = s.zrl * depth_ac[0xf0] + d[s.sym] + s.nbits;
  }
  if (s.eob) len += depth_ac[0];
  return len;
}

// Inclusive prefix sum over the 64 lanes.
GZ_DEVFN int wave_inclusive_sum(int v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(v, d);
    if (lane >= d) v += o;
  }
  return v;
}

// bits[b] = number of scan bits of MCU b (ncomp components).
__global__ __launch_bounds__(64) void k_jpeg_block_bits(const int16_t* __restrict__ coeffs,
                                                        const int* __restrict__ q, int nb,
                                                        int ncomp, JpegCodes codes,
                                                        unsigned* __restrict__ bits) {
  const int lane = threadIdx.x, b = blockIdx.x;
  int total = 0;
  for (int c = 0; c < ncomp; ++c) {
    const LaneSyms s = lane_symbols(coeffs, q, nb, c, b, lane);
    const int len = lane_bits(s, codes.depth + c * 256, codes.depth + (3 + c) * 256);
    total += __shfl(wave_inclusive_sum(len, lane), 63);
  }
  if (lane == 0) bits[b] = (unsigned)total;
}

// off[0..nb] = exclusive prefix sums of bits[0..nb) (64-bit).  One workgroup of 1024 walks
// the array in tiles of 4096: a lane takes 4 consecutive values (one 16-byte load, issued one
// tile ahead), the tile is scanned in 32 bits (a value is at most a block's scan bits or its
// candidate count: 4096 of them stay far below 2^31) with wavefront shuffles plus one LDS
// exchange between the 16 wavefronts, and the running total is carried in 64 bits.  (A
// variant that stages 32768 values at a time in 135 KB of LDS measured 42 us against this
// one's 22 us inside an encode: a workgroup that needs most of a CU's LDS waits for the
// Compare kernels that share the GPU with it.)
constexpr int kScanTile = 4096;

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

__global__ __launch_bounds__(1024) void k_jpeg_scan_offsets(const unsigned* __restrict__ bits,
                                                            int nb,
                                                            unsigned long long* __restrict__ off) {
  __shared__ int wave_total[16];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  unsigned long long carry = 0;
  unsigned cur[4], nxt[4];
  scan_load4(bits, nb, 4 * t, cur);
  for (int base = 0; base < nb; base += kScanTile) {
    const int at = base + 4 * t;
    if (base + kScanTile < nb) scan_load4(bits, nb, at + kScanTile, nxt);
    const int s0 = (int)cur[0], s1 = s0 + (int)cur[1], s2 = s1 + (int)cur[2], s3 = s2 + (int)cur[3];
    const int inc = wave_inclusive_sum(s3, lane);
    if (lane == 63) wave_total[wv] = inc;
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int w = wave_total[k];
      if (k < wv) before += w;
      all += w;
    }
    const unsigned long long excl = carry + (unsigned long long)(before + inc - s3);
    if (at < nb) off[at] = excl;
    if (at + 1 < nb) off[at + 1] = excl + (unsigned long long)s0;
    if (at + 2 < nb) off[at + 2] = excl + (unsigned long long)s1;
    if (at + 3 < nb) off[at + 3] = excl + (unsigned long long)s2;
    carry += (unsigned long long)all;
    __syncthreads();   // wave_total is rewritten by the next tile
#pragma unroll
    for (int e = 0; e < 4; ++e) cur[e] = nxt[e];
  }
  if (t == 0) off[nb] = carry;
}

// --------------------------------------------------------------------------- emit ----
// The bit buffer is an array of 32-bit words, stream bit p at word p/32, bit 31-(p%32)
// (so that stream byte j = (word[j/4] >> (24 - 8*(j%4))) & 0xff).  It must be zero before
// the launch.  Each MCU assembles its bits in LDS and ORs whole words out (the first and
// last word of an MCU are shared with its neighbours, hence atomicOr); MCUs too long for
// the staging buffer OR their pieces straight into global memory.
constexpr int kStageWords = 256;

GZ_DEVFN void or_bits(unsigned* words, bool lds, unsigned long long pos, unsigned value, int len) {
  if (len == 0) return;
  const unsigned long long w = pos >> 5;
  const int sh = (int)(pos & 31);
  const unsigned long long x = (unsigned long long)value << (64 - len - sh);
  const unsigned hi = (unsigned)(x >> 32), lo = (unsigned)x;
  (void)lds;
  if (hi) atomicOr(&words[w], hi);
  if (lo) atomicOr(&words[w + 1], lo);
}

__global__ __launch_bounds__(64) void k_jpeg_emit(const int16_t* __restrict__ coeffs,
                                                  const int* __restrict__ q, int nb, int ncomp,
                                                  JpegCodes codes,
                                                  const unsigned long long* __restrict__ off,
                                                  unsigned* __restrict__ words,
                                                  unsigned long long cap_words) {
  __shared__ unsigned stage[kStageWords + 2];
  const int lane = threadIdx.x, b = blockIdx.x;
  const unsigned long long start = off[b], end = off[b + 1];
  // the last MCU also writes the 1-padding up to the byte boundary (BitWriter::JumpToByteBoundary)
  const int pad = b == nb - 1 ? (int)((8 - (end & 7)) & 7) : 0;
  // `words` is sized for valid code lengths (<= 16 bits); with anything else the scan can be
  // longer, the host reports that afterwards, and nothing is written past the buffer here
  if (((end + pad + 63) >> 5) + 1 > cap_words) return;
  const unsigned long long word0 = start >> 5;
  const unsigned long long span = (end + pad) - (word0 << 5);   // bits from word0's first bit
  const bool staged = span <= (unsigned long long)kStageWords * 32;
  if (staged) {
    for (int i = lane; i < kStageWords + 2; i += 64) stage[i] = 0;
    __syncthreads();
  }
  unsigned* dst = staged ? stage : words;
  unsigned long long base = staged ? start - (word0 << 5) : start;
  for (int c = 0; c < ncomp; ++c) {
    const LaneSyms s = lane_symbols(coeffs, q, nb, c, b, lane);
    const unsigned char* ddc = codes.depth + c * 256;
    const unsigned char* dac = codes.depth + (3 + c) * 256;
    const unsigned short* cdc = codes.code + c * 256;
    const unsigned short* cac = codes.code + (3 + c) * 256;
    const int len = lane_bits(s, ddc, dac);
    const int incl = wave_inclusive_sum(len, lane);
    const int total = __shfl(incl, 63);
    unsigned long long pos = base + (unsigned long long)(incl - len);
    if (s.sym >= 0) {
      for (int z = 0; z < s.zrl; ++z) {
        or_bits(dst, staged, pos, cac[0xf0], dac[0xf0]);
        pos += dac[0xf0];
      }
      const int dl = s.is_dc ? ddc[s.sym] : dac[s.sym];
      or_bits(dst, staged, pos, s.is_dc ? cdc[s.sym] : cac[s.sym], dl);
      pos += dl;
      or_bits(dst, staged, pos, s.extra, s.nbits);
      pos += s.nbits;
    }
    if (s.eob) or_bits(dst, staged, pos, cac[0], dac[0]);
    base += (unsigned long long)total;
  }
  if (pad && lane == 0) or_bits(dst, staged, base, (1u << pad) - 1u, pad);
  if (staged) {
    __syncthreads();
    const int nwords = (int)((span + 31) >> 5);
    for (int i = lane; i < nwords; i += 64)
      if (stage[i]) atomicOr(&words[word0 + i], stage[i]);
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
