// Phase B of Processor::SelectFrequencyMasking on the device: the global candidate order
// (processor.cc:622-663) and the partition step of the std::sort the reference runs on it
// (processor.cc:675-678).
//
// The reference builds `global_order` = (block, val) for every remaining candidate of
// every block with a non-zero weight, std::sort-s it by val and consumes a prefix.  Equal
// vals occur across different blocks, std::sort is not stable, and which tied block is
// served first feeds the JPEG bytes; so the search driver (guetzli_amd/host/lazy_sort.h)
// evaluates libstdc++'s introsort lazily from the front.  The expensive part of that --
// the unguarded Hoare partitions of the multi-million-entry ranges -- runs here, with the
// identical outcome:
//
//   libstdc++'s __unguarded_partition(first, last, pivot) swaps the k-th element from the
//   left that is not less than the pivot ("left stopper" L_k) with the k-th element from the
//   right that is not greater ("right stopper" R_k) for k = 0, 1, ... while L_k < R_k.  With
//   lr(i) = number of left stoppers before i and ra(i) = number of right stoppers after i,
//   a left stopper i (rank k = lr(i)) is swapped iff ra(i) > k, a right stopper j (rank
//   k = ra(j)) iff lr(j) > k, and partners share k.  Both conditions are local given two
//   prefix counts, so the partition is two streaming passes + a pairwise swap.  The cut is
//   min(first unswapped left stopper, last swapped right stopper) -- where the serial scan
//   stops.
//
// Entries are 8 bytes {int block; float val}, the layout of std::pair<int, float>.
#pragma once
#include "gz_common.h"
#include "gz_kernels_entropy.h"   // lane_symbols

namespace gz {

struct OrderEntry {
  int block;
  float val;
};

// comp = [](a, b) { return a.second < b.second; }  (processor.cc:676-678)
GZ_DEVFN bool order_less(const OrderEntry& a, const OrderEntry& b) { return a.val < b.val; }

// ----------------------------------------------------------------- building the order --
// The order is built without a scan kernel and without atomics on the critical path: the kernel
// that knows the blocks' entry counts n_b (k_weights_gather, or k_order_sizes when the weights come
// from the host) also writes, per workgroup of kOrderGroup blocks, the group's sum of n_b and its
// number of blocks with n_b > 0; k_order_fill's workgroups add up the groups before theirs for
// themselves (a few hundred numbers) and find their blocks' offsets from there.  (Round 3: an
// atomic per wavefront on the one counter -- 2 000 device-scope atomics on one address, 26 us --
// and a decoupled look-back scan of 64 tiles, 15 us, both on phase B's critical path.)
constexpr int kOrderGroup = 256;   // = the block size of the kernels that write the group sums

// Sum over the wavefront; every lane gets it.
GZ_DEVFN int wave_sum(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(v, (unsigned)d);
    if (lane >= d) v += o;
  }
  return __shfl(v, 63);
}

// The group's two sums and every block's offset inside its group (the entries of the group's
// blocks before it); every thread of the workgroup calls this (n = 0 beyond the last block).
GZ_DEVFN void order_group_sums(bool valid, int b, int n, unsigned* __restrict__ group_sums,
                               unsigned* __restrict__ blk_off) {
  __shared__ int lds8[8];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int incl = n, cnt = n > 0 ? 1 : 0;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, (unsigned)d), oc = __shfl_up(cnt, (unsigned)d);
    if (lane >= d) { incl += o; cnt += oc; }
  }
  if (lane == 63) { lds8[wave] = incl; lds8[4 + wave] = cnt; }
  __syncthreads();
  int before = 0;
  for (int wv = 0; wv < wave; ++wv) before += lds8[wv];
  if (valid) blk_off[b] = (unsigned)(before + incl - n);
  if (t == 0) {
    group_sums[2 * blockIdx.x] = (unsigned)(lds8[0] + lds8[1] + lds8[2] + lds8[3]);
    group_sums[2 * blockIdx.x + 1] = (unsigned)(lds8[4] + lds8[5] + lds8[6] + lds8[7]);
  }
}

// n_b[b] = number of entries block b contributes (processor.cc:638-661): none if its weight
// is 0; the candidates from next_cand[b] on for "up"; the next_cand[b] applied ones for
// "down".
GZ_DEVFN int order_size_of(bool valid, int b, float w, const int* __restrict__ cnt,
                           const int* __restrict__ next_cand, int direction,
                           unsigned* __restrict__ n_b) {
  int n = 0;
  if (valid) {
    if (!(w == 0)) {
      const int at = next_cand[b];
      n = direction > 0 ? cnt[b] - at : at;
      if (n < 0) n = 0;
    }
    n_b[b] = (unsigned)n;
  }
  return n;
}

__global__ __launch_bounds__(256) void k_order_sizes(const int* __restrict__ cnt,
                                                     const int* __restrict__ next_cand,
                                                     const float* __restrict__ weight,
                                                     int direction, int nb,
                                                     unsigned* __restrict__ n_b,
                                                     unsigned* __restrict__ group_sums,
                                                     unsigned* __restrict__ blk_off) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = b < nb;
  const int n = order_size_of(valid, b, valid ? weight[b] : 0.0f, cnt, next_cand, direction, n_b);
  order_group_sums(valid, b, n, group_sums, blk_off);
}

// Sixteen lanes per block (a block has a few dozen entries at most; a whole wavefront per block
// left most of its lanes idle and was bound by the number of wavefronts): entry j of block b is
// (b, (err[at+j] - max_err) / weight) for "up", (b, (max_err - err[at-1-j]) / weight) for "down"
// (float arithmetic, processor.cc:649-657).  err has a fixed stride of 192 per block
// (k_block_search's layout).  A block's offset = the sums of the groups before its own (added up
// by every wavefront for itself: a few hundred numbers, no barrier) + its offset inside the group.
// The first wavefront of the grid also writes the order's size (*total) and blocks_to_change
// (counters[0]).  counters[1] += number of vals < limit when count_below (the partition_point of
// processor.cc:690-696 counts exactly these once the order is sorted).
constexpr int kFillLanes = 16;
constexpr int kFillBlocks = 256 / kFillLanes;
__global__ __launch_bounds__(256) void k_order_fill(const float* __restrict__ err,
                                                    const int* __restrict__ next_cand,
                                                    const float* __restrict__ weight,
                                                    const float* __restrict__ max_err,
                                                    const unsigned* __restrict__ n_b,
                                                    const unsigned* __restrict__ group_sums,
                                                    const unsigned* __restrict__ blk_off,
                                                    int direction, int nb, int count_below,
                                                    float limit, OrderEntry* __restrict__ out,
                                                    unsigned long long* __restrict__ total,
                                                    unsigned* __restrict__ counters) {
  const int t = threadIdx.x, lane = t & 63;
  const int group = t / kFillLanes, sub = t % kFillLanes;
  const int b0 = blockIdx.x * kFillBlocks, b = b0 + group;
  const int g = b0 / kOrderGroup, ngroups = (nb + kOrderGroup - 1) / kOrderGroup;
  if (blockIdx.x == 0 && t < 64) {   // (one wavefront)
    int all = 0, changed = 0;
    for (int i = lane; i < ngroups; i += 64) {
      all += (int)group_sums[2 * i];
      changed += (int)group_sums[2 * i + 1];
    }
    all = wave_sum(all);
    changed = wave_sum(changed);
    if (lane == 0) {
      *total = (unsigned long long)(unsigned)all;
      counters[0] = (unsigned)changed;
    }
  }
  // everything the block needs is asked for at once: its scalars, its offset, its first two
  // rounds of entries, the sums of the groups before
  const bool valid = b < nb;
  int n = 0, at = 0;
  unsigned inside = 0;
  float base = 0.0f, wb = 1.0f;
  if (valid) {
    n = (int)n_b[b];
    at = next_cand[b];
    base = max_err[b];
    wb = weight[b];
    inside = blk_off[b];
  }
  const float* e = err + (size_t)(valid ? b : 0) * 192;
  auto entry = [&](int j) { return direction > 0 ? e[at + j] : e[at - 1 - j]; };
  float e0 = 0.0f, e1 = 0.0f;
  if (sub < n) e0 = entry(sub);
  if (sub + kFillLanes < n) e1 = entry(sub + kFillLanes);
  int part = 0;
  for (int i = lane; i < g; i += 64) part += (int)group_sums[2 * i];
  const unsigned long long o = (unsigned)wave_sum(part) + inside;
  if (n == 0) return;
  unsigned below = 0;
  auto put = [&](int j, float ev) {
    OrderEntry v;
    v.block = b;
    v.val = direction > 0 ? (ev - base) / wb : (base - ev) / wb;
    out[o + j] = v;
    below += (count_below && v.val < limit) ? 1u : 0u;
  };
  if (sub < n) put(sub, e0);
  if (sub + kFillLanes < n) put(sub + kFillLanes, e1);
  for (int j = sub + 2 * kFillLanes; j < n; j += kFillLanes) put(j, entry(j));
  if (count_below && below) atomicAdd(&counters[1], below);
}

// ------------------------------------------------ block weights on the device (a19) --
// ComputeBlockErrorAdjustmentWeights (butteraugli_comparator.cc:521-557) from the per-block
// maxima of the distance map, as two gather passes (the reference scatters 1/(d+1) around
// every block that exceeds its local limit; the maximum a block receives is 1/(dmin+1) for
// the nearest such block within max_block_dist, since 1/(d+1) decreases with d).
//   pass 1: flag[b] = direction > 0 ? (bmax <= td && local <= 1.1 td)
//                                   : (bmax > 0.5 td + 0.5 local),   local = max(td_f, nbhd max)
//   pass 2: weight[b] = direction > 0 ? flag : 1/(dmin+1) or 0
// td = target * target_mul in double, td_f = (float)td; comparisons promote as the reference
// does.  use_bmax == 0: the distance map is all zero (first "up" iteration, processor.cc:627).
__global__ __launch_bounds__(256) void k_weights_flag(const float* __restrict__ bmax, int use_bmax,
                                                      int bw, int bh, float target,
                                                      double target_mul, int direction, int r,
                                                      unsigned char* __restrict__ flag,
                                                      unsigned* __restrict__ clear2) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  // first kernel of an order build: also resets the build's two counters
  if (clear2 && b == 0) { clear2[0] = 0u; clear2[1] = 0u; }
  if (b >= bw * bh) return;
  const int bx = b % bw, by = b / bw;
  const double td = target * target_mul;
  float local = static_cast<float>(td);
  const int x0 = bx - r > 0 ? bx - r : 0, y0 = by - r > 0 ? by - r : 0;
  const int x1 = bx + 1 + r < bw ? bx + 1 + r : bw, y1 = by + 1 + r < bh ? by + 1 + r : bh;
  if (use_bmax)
    for (int y = y0; y < y1; ++y)
      for (int x = x0; x < x1; ++x) {
        const float v = bmax[y * bw + x];
        local = v > local ? v : local;   // std::max(local, v)
      }
  const float own = use_bmax ? bmax[b] : 0.0f;
  bool f;
  if (direction > 0) {
    f = (double)own <= td && (double)local <= 1.1 * td;
  } else {
    const double kLocalMaxWeight = 0.5;
    f = !((double)own <= (1 - kLocalMaxWeight) * td + kLocalMaxWeight * (double)local);
  }
  flag[b] = f ? 1 : 0;
}

// The entry count of the block and the group sums (k_order_sizes) come out of the same pass when
// cnt is given.
// adv_direction != 0: the previous iteration's max_block_error update (k_order_advance, below) is still due and is
// made here, with the weight this kernel is about to replace -- one launch less between an iteration's last step
// and its evaluation.
__global__ __launch_bounds__(256) void k_weights_gather(const unsigned char* __restrict__ flag,
                                                        int bw, int bh, int direction, int r,
                                                        float* __restrict__ weight,
                                                        const int* __restrict__ cnt,
                                                        const int* __restrict__ next_cand,
                                                        unsigned* __restrict__ n_b,
                                                        unsigned* __restrict__ group_sums,
                                                        unsigned* __restrict__ blk_off,
                                                        float* __restrict__ max_err, float adv_threshold,
                                                        int adv_direction) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = b < bw * bh;
  float w = 0.0f;
  if (valid) {
    if (adv_direction) max_err[b] += weight[b] * adv_threshold * adv_direction;
    if (direction > 0) {
      w = flag[b] ? 1.0f : 0.0f;
    } else {
      const int bx = b % bw, by = b / bw;
      const int x0 = bx - r > 0 ? bx - r : 0, y0 = by - r > 0 ? by - r : 0;
      const int x1 = bx + 1 + r < bw ? bx + 1 + r : bw, y1 = by + 1 + r < bh ? by + 1 + r : bh;
      int dmin = r + 1;
      for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x)
          if (flag[y * bw + x]) {
            const int dy = y > by ? y - by : by - y, dx = x > bx ? x - bx : bx - x;
            const int d = dy > dx ? dy : dx;
            dmin = d < dmin ? d : dmin;
          }
      w = dmin <= r ? 1.0f / (dmin + 1.0f) : 0.0f;
    }
    weight[b] = w;
  }
  if (cnt) order_group_sums(valid, b, order_size_of(valid, b, w, cnt, next_cand, direction, n_b), group_sums, blk_off);
}

// max_block_error[i] += block_weight[i] * val_threshold * direction  (processor.cc:754-756)
__global__ __launch_bounds__(256) void k_order_advance(float* __restrict__ max_err,
                                                       const float* __restrict__ weight,
                                                       float val_threshold, int direction, int nb) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  max_err[b] += weight[b] * val_threshold * direction;
}

// OutputImageComponent::SetCoeffBlock for single coefficients: coeffs[pos[i]] = val[i]
// (positions distinct).
__global__ __launch_bounds__(256) void k_apply_coeff_edits(const int* __restrict__ pos,
                                                           const short* __restrict__ val, int n,
                                                           short* __restrict__ coeffs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) coeffs[pos[i]] = val[i];
}

// The coefficient changes of phase B's global loop (processor.cc:704-736) for whole blocks:
// entry i advances block blocks[i] by counts[i] steps in `direction`.  Step j of a block
// applies its candidate next_cand + j ("up": the coefficient is zeroed) or next_cand - 1 - j
// ("down": the coefficient gets its quantised original value back), unless it is one of the
// two "precious" coefficients.  A block's candidates are distinct coefficients, so its steps
// are independent: one wave per block, one lane per step.
// Blocks are positions of the search grid (gz_block_zeroing_orders): component c's block of
// grid position b is block coff[c] + b of the coefficient arrays (4:4:4, any mask: the three
// components share the grid; 4:2:0: the luma grid for mask 1, the chroma grid for mask 6).
struct StepGeom {
  int coff[3];
  int comp_mask;
};

GZ_DEVFN int step_new_value(int direction, const short* __restrict__ ob, int k, int quant) {
  int newval = 0;
  if (direction < 0) {   // Quantize(), quantize.h:24-29
    const int raw = ob[k];
    const int r = raw % quant;
    const int delta = 2 * r > quant ? quant - r : ((-2) * r > quant ? -quant - r : -r);
    newval = (short)(raw + (int)(short)delta);
  }
  return newval;
}
GZ_DEVFN bool step_is_precious(const short* __restrict__ ob, int k) {   // processor.cc:722-733
  if (k != 1 && k != 8) return false;
  int sum_of_hf = 0;
  for (int ii = 3; ii < 64; ++ii) {
    if ((ii & 7) < 3 && ii < 3 * 8) continue;
    const int v = ob[ii];
    sum_of_hf += v < 0 ? -v : v;
  }
  const int limit = sum_of_hf < 60 ? 4 : 8;
  const int v = ob[k];
  return (v < 0 ? -v : v) >= limit;
}

__global__ __launch_bounds__(256) void k_apply_steps(const int* __restrict__ blocks,
                                                     const int* __restrict__ counts, int n,
                                                     int direction, const int* __restrict__ next_cand,
                                                     const unsigned char* __restrict__ cand_idx,
                                                     const short* __restrict__ orig,
                                                     short* __restrict__ cand,
                                                     const int* __restrict__ q, StepGeom sg) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave;
  if (i >= n) return;
  const int b = blocks[i], cnt = counts[i], nx = next_cand[b];
  for (int j = lane; j < cnt; j += 64) {
    const int p = direction > 0 ? nx + j : nx - 1 - j;
    const int idx = cand_idx[(size_t)b * 192 + p];
    const int c = idx >> 6, k = idx & 63;
    const short* ob = orig + ((size_t)sg.coff[c] + b) * 64;
    const int newval = step_new_value(direction, ob, k, q[c * 64 + k]);
    if (!(newval == 0 && step_is_precious(ob, k))) cand[((size_t)sg.coff[c] + b) * 64 + k] = (short)newval;
  }
}

// The same steps, one wavefront per block, together with what they do to the AC symbol
// statistics (BuildACHistograms of the image before minus after, for the touched blocks only):
// the block's coefficient blocks go through LDS, their symbols are counted out of the
// histogram, the steps are applied, the symbols are counted back in.  delta: [3][256] counters
// (wrapping unsigned arithmetic = signed differences) x kStepDeltaCopies, zero between launches; jq = the quantiser
// the symbols are defined under (gz_jpeg_histograms' matrix).  After ~6000 steps an iteration
// the host's size model needs the statistics again; recounting the 32 400 blocks of a 1080p
// image took 40-50 us, the touched blocks take a fraction of that.
GZ_DEVFN void steps_count_symbols(const short* blk3, const int* __restrict__ jq, int comp_mask,
                                  int lane, bool live, unsigned sign, unsigned* s_delta) {
  const int nat = kNaturalOrderDev[lane];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (!((comp_mask >> c) & 1)) continue;   // (uniform)
    const LaneSyms s = lane_ac_symbols((int)blk3[c * 64 + nat] / jq[c * 64 + nat], lane);
    if (!live) continue;
    unsigned* h = &s_delta[c * 256];
    if (s.zrl) atomicAdd(&h[0xf0], sign * (unsigned)s.zrl);
    if (s.sym >= 0 && !s.is_dc) atomicAdd(&h[s.sym], sign);
    if (s.eob) atomicAdd(&h[0], sign);
  }
}

// Grid-stride over the touched blocks (a wavefront takes blocks wv, wv + total wavefronts, ...):
// a workgroup's LDS histogram collects the changes of all its blocks and goes to global memory
// ONCE, into one of kStepDeltaCopies copies (the caller sums them).  With a workgroup per four
// blocks, 26 000 touched blocks of a 4K iteration meant 6500 workgroups x ~100 atomics on the same
// 768 words: the launch spent its 81 us queueing at a dozen L2 lines.  The wavefronts of a
// workgroup share nothing but that histogram, so they synchronise only around it.
constexpr int kStepDeltaCopies = 32;
constexpr int kStepHistGrid = 2048;   // workgroups at most (1024 / 16 copies: 7 ms of a 4K encode waiting for the kernel, 2048 / 32: 6, 4096 / 64: 9.5)
// list_out (optional): the entries' blocks again, in device memory -- for k_reconstruct_listed behind this kernel,
// which transforms the touched block positions of the candidate's linear planes again while the host looks at the
// statistics.  (The same work inside this kernel's wavefronts, the coefficients being in LDS anyway, made the launch
// the host waits for 44 -> 62 us at 4K: profiles/r06_chain_experiments.log section 14.)
__global__ __launch_bounds__(256) void k_apply_steps_hist(const int* __restrict__ blocks,
                                                          const int* __restrict__ counts, int n,
                                                          int direction, const int* __restrict__ next_cand,
                                                          const unsigned char* __restrict__ cand_idx,
                                                          const short* __restrict__ orig,
                                                          short* __restrict__ cand,
                                                          const int* __restrict__ q,
                                                          const int* __restrict__ jq, StepGeom sg,
                                                          unsigned* __restrict__ delta, int* __restrict__ list_out) {
  __shared__ short s_blk[4][3 * 64];
  __shared__ unsigned s_delta[3 * 256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int k = threadIdx.x; k < 3 * 256; k += 256) s_delta[k] = 0u;
  __syncthreads();
  // Wavefront g of the grid takes the `per` consecutive entries from g * per (the same trip count for
  // every wavefront: one past the end repeats the last block and drops the result, so that every thread
  // of a workgroup passes the same sequence of synchronisation points, which the test emulation relies
  // on).  `blocks` / `counts` are the caller's page-locked staging buffer, read over the bus: a wavefront
  // fetches its entries 64 at a time, one per lane (consecutive addresses), and hands them round.
  const int per = (n + (int)gridDim.x * 4 - 1) / ((int)gridDim.x * 4);
  const int base = ((int)blockIdx.x * 4 + wave) * per;
  int pre_b = 0, pre_c = 0;
  for (int it = 0; it < per; ++it) {
    if ((it & 63) == 0) {
      const int i2 = base + it + lane;
      const bool lv = it + lane < per && i2 < n;
      pre_b = blocks[lv ? i2 : n - 1];
      pre_c = lv ? counts[i2] : 0;
      if (list_out && lv) list_out[i2] = pre_b;
    }
    const bool live = base + it < n;
    const int b = __shfl(pre_b, it & 63), cnt = __shfl(pre_c, it & 63), nx = next_cand[b];
#pragma unroll
    for (int c = 0; c < 3; ++c)
      if ((sg.comp_mask >> c) & 1) s_blk[wave][c * 64 + lane] = cand[((size_t)sg.coff[c] + b) * 64 + lane];
    GZ_WAVE_SYNC();
    steps_count_symbols(s_blk[wave], jq, sg.comp_mask, lane, live, 0xffffffffu, s_delta);
    GZ_WAVE_SYNC();
    for (int j = lane; j < cnt; j += 64) {
      const int p = direction > 0 ? nx + j : nx - 1 - j;
      const int idx = cand_idx[(size_t)b * 192 + p];
      const int c = idx >> 6, k = idx & 63;
      const short* ob = orig + ((size_t)sg.coff[c] + b) * 64;
      const int newval = step_new_value(direction, ob, k, q[c * 64 + k]);
      if (!(newval == 0 && step_is_precious(ob, k))) s_blk[wave][c * 64 + k] = (short)newval;
    }
    GZ_WAVE_SYNC();
    steps_count_symbols(s_blk[wave], jq, sg.comp_mask, lane, live, 1u, s_delta);
    if (live) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
        if ((sg.comp_mask >> c) & 1) cand[((size_t)sg.coff[c] + b) * 64 + lane] = s_blk[wave][c * 64 + lane];
    }
    GZ_WAVE_SYNC();   // (the block's copy is overwritten by the wavefront's next block)
  }
  __syncthreads();
  unsigned* out = delta + (blockIdx.x % kStepDeltaCopies) * 768;
  for (int k = threadIdx.x; k < 3 * 256; k += 256)
    if (s_delta[k]) atomicAdd(&out[k], s_delta[k]);
}

// The copies' sums into the caller's page-locked result (mapped into the device's address space: no copy
// command behind the kernels), the copies left zeroed for the next launch (no memset command before it).
// A kernel of its own, one workgroup: a "last workgroup done" ticket inside k_apply_steps_hist needs an
// agent-scope fence per workgroup -- a write-back of the XCD's L2 each -- and cost 75 us per launch.
__global__ __launch_bounds__(256) void k_steps_hist_sum(unsigned* __restrict__ delta, int* __restrict__ result) {
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int k = (int)threadIdx.x + 256 * j;
    unsigned v[kStepDeltaCopies];
#pragma unroll
    for (int r = 0; r < kStepDeltaCopies; ++r) v[r] = delta[r * 768 + k];   // (all in flight together)
    unsigned sum = 0;
#pragma unroll
    for (int r = 0; r < kStepDeltaCopies; ++r) {
      sum += v[r];
      if (v[r]) delta[r * 768 + k] = 0u;
    }
    result[k] = (int)sum;
  }
}

// Per-block maxima of the distance map over fx x fy groups of 8x8 blocks: the first loop of
// ComputeBlockErrorAdjustmentWeights (butteraugli_comparator.cc:505-520) for a subsampled
// search grid, from the 8x8 maxima the final blur kernel leaves (max is exact in any order).
__global__ __launch_bounds__(256) void k_block_max_group(const float* __restrict__ bmax, int bw, int bh,
                                                         int gw, int gh, int f,
                                                         float* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= gw * gh) return;
  const int gx = b % gw, gy = b / gw;
  float m = 0.0f;
  for (int y = gy * f; y < gy * f + f && y < bh; ++y)
    for (int x = gx * f; x < gx * f + f && x < bw; ++x) {
      const float v = bmax[y * bw + x];
      m = v > m ? v : m;
    }
  out[b] = m;
}

// -------------------------------------------------------------- one introsort partition --
struct PartScalars {
  unsigned m;      // number of swapped pairs
  unsigned cut_l;  // first unswapped left stopper (relative to first), 0xffffffff if none
  unsigned cut_r;  // last swapped right stopper (the smallest position), 0xffffffff if none
  unsigned pad;
  OrderEntry pivot;
};

constexpr int kPartItems = 8;                       // consecutive entries per thread
constexpr int kPartChunk = 256 * kPartItems;        // entries per workgroup

GZ_DEVFN void order_swap(OrderEntry* a, size_t i, size_t j) {
  const OrderEntry t = a[i];
  a[i] = a[j];
  a[j] = t;
}

// std::__move_median_to_first(first, first+1, mid, last-1) of libstdc++'s
// __unguarded_partition_pivot on [lo, hi), then the pivot value for the passes below.
__global__ void k_part_median(OrderEntry* __restrict__ a, size_t lo, size_t hi,
                              PartScalars* __restrict__ s) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const size_t r = lo, x = lo + 1, y = lo + (hi - lo) / 2, z = hi - 1;
  if (order_less(a[x], a[y])) {
    if (order_less(a[y], a[z])) order_swap(a, r, y);
    else if (order_less(a[x], a[z])) order_swap(a, r, z);
    else order_swap(a, r, x);
  } else if (order_less(a[x], a[z])) {
    order_swap(a, r, x);
  } else if (order_less(a[y], a[z])) {
    order_swap(a, r, z);
  } else {
    order_swap(a, r, y);
  }
  s->pivot = a[lo];
  s->m = 0;
  s->cut_l = 0xffffffffu;
  s->cut_r = 0xffffffffu;
}

// Workgroup-wide inclusive prefix sum of one unsigned per thread (256 threads).
GZ_DEVFN unsigned wg_inclusive_scan(unsigned v, unsigned* lds) {
  const int t = threadIdx.x;
  lds[t] = v;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    unsigned o = 0;
    if (t >= d) o = lds[t - d];
    __syncthreads();
    lds[t] += o;
    __syncthreads();
  }
  const unsigned r = lds[t];
  __syncthreads();
  return r;
}

// The same with wavefront shuffles: a scan inside every wavefront, then the (at most four)
// wavefront totals through LDS -- two barriers instead of sixteen.
GZ_DEVFN unsigned wg_inclusive_scan_fast(unsigned v, unsigned* lds4) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  unsigned x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned o = (unsigned)__shfl_up((int)x, (unsigned)d);
    if (lane >= d) x += o;
  }
  if (lane == 63) lds4[wave] = x;
  __syncthreads();
  unsigned base = 0;
  for (int wv = 0; wv < wave; ++wv) base += lds4[wv];
  __syncthreads();
  return x + base;
}

// Pass 1: per chunk, the number of left stoppers (!(e < pivot)) and right stoppers
// (!(pivot < e)) among a[first .. first+n).
__global__ __launch_bounds__(256) void k_part_count(const OrderEntry* __restrict__ a,
                                                    size_t first, unsigned n,
                                                    const PartScalars* __restrict__ s,
                                                    unsigned* __restrict__ cnt_l,
                                                    unsigned* __restrict__ cnt_r) {
  __shared__ unsigned lds[256];
  const OrderEntry pv = s->pivot;
  const unsigned base = blockIdx.x * (unsigned)kPartChunk + threadIdx.x * (unsigned)kPartItems;
  unsigned nl = 0, nr = 0;
#pragma unroll
  for (int i = 0; i < kPartItems; ++i) {
    const unsigned p = base + i;
    if (p < n) {
      const OrderEntry e = a[first + p];
      nl += !order_less(e, pv) ? 1u : 0u;
      nr += !order_less(pv, e) ? 1u : 0u;
    }
  }
  const unsigned tl = wg_inclusive_scan(nl, lds);
  const unsigned tr = wg_inclusive_scan(nr, lds);
  if (threadIdx.x == 255) {
    cnt_l[blockIdx.x] = tl;
    cnt_r[blockIdx.x] = tr;
  }
}

// base_l[c] = left stoppers in chunks before c; base_r[c] = right stoppers in chunks after c.
// One workgroup of 1024.
__global__ __launch_bounds__(1024) void k_part_scan(const unsigned* __restrict__ cnt_l,
                                                    const unsigned* __restrict__ cnt_r,
                                                    int nchunks, unsigned* __restrict__ base_l,
                                                    unsigned* __restrict__ base_r) {
  __shared__ unsigned part[1024];
  const int t = threadIdx.x;
  const int per = (nchunks + 1023) / 1024;
  const int lo = t * per < nchunks ? t * per : nchunks;
  const int hi = lo + per < nchunks ? lo + per : nchunks;
  for (int pass = 0; pass < 2; ++pass) {
    // pass 0: forward over cnt_l; pass 1: the same over cnt_r read back to front
    unsigned sum = 0;
    for (int i = lo; i < hi; ++i) sum += pass == 0 ? cnt_l[i] : cnt_r[nchunks - 1 - i];
    part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      unsigned o = 0;
      if (t >= d) o = part[t - d];
      __syncthreads();
      part[t] += o;
      __syncthreads();
    }
    unsigned run = part[t] - sum;
    for (int i = lo; i < hi; ++i) {
      if (pass == 0) {
        base_l[i] = run;
        run += cnt_l[i];
      } else {
        base_r[nchunks - 1 - i] = run;
        run += cnt_r[nchunks - 1 - i];
      }
    }
    __syncthreads();
  }
}

// Pass 2: ranks of the stoppers; swapped ones record their position under their rank.
__global__ __launch_bounds__(256) void k_part_scatter(const OrderEntry* __restrict__ a,
                                                      size_t first, unsigned n,
                                                      PartScalars* __restrict__ s,
                                                      const unsigned* __restrict__ base_l,
                                                      const unsigned* __restrict__ base_r,
                                                      unsigned* __restrict__ pos_l,
                                                      unsigned* __restrict__ pos_r) {
  __shared__ unsigned lds[256];
  __shared__ unsigned red[3];
  const OrderEntry pv = s->pivot;
  const int t = threadIdx.x;
  const unsigned base = blockIdx.x * (unsigned)kPartChunk + t * (unsigned)kPartItems;
  unsigned fl = 0, fr = 0, nl = 0, nr = 0;   // flag bit masks and counts of this thread
#pragma unroll
  for (int i = 0; i < kPartItems; ++i) {
    const unsigned p = base + i;
    if (p < n) {
      const OrderEntry e = a[first + p];
      if (!order_less(e, pv)) { fl |= 1u << i; ++nl; }
      if (!order_less(pv, e)) { fr |= 1u << i; ++nr; }
    }
  }
  if (t == 0) { red[0] = 0; red[1] = 0xffffffffu; red[2] = 0xffffffffu; }
  // left stoppers before this thread's first entry / right stoppers after its last one
  const unsigned incl_l = wg_inclusive_scan(nl, lds);
  const unsigned incl_r = wg_inclusive_scan(nr, lds);
  __shared__ unsigned chunk_r;
  if (t == 255) chunk_r = incl_r;
  __syncthreads();
  unsigned lr = base_l[blockIdx.x] + (incl_l - nl);
  unsigned ra = base_r[blockIdx.x] + (chunk_r - incl_r) + nr;   // + stoppers after entry i, below
  unsigned my_m = 0, my_cl = 0xffffffffu, my_cr = 0xffffffffu;
#pragma unroll
  for (int i = 0; i < kPartItems; ++i) {
    const unsigned p = base + i;
    const bool is_l = (fl >> i) & 1u, is_r = (fr >> i) & 1u;
    if (is_r) --ra;            // ra = right stoppers strictly after entry i
    if (is_l) {
      if (ra > lr) {           // L_k < R_k with k = lr
        pos_l[lr] = p;
        my_m = lr + 1;
      } else if (p < my_cl) {
        my_cl = p;
      }
    }
    if (is_r && lr > ra) {     // partner exists on the left: k = ra
      pos_r[ra] = p;
      if (p < my_cr) my_cr = p;
    }
    if (is_l) ++lr;            // lr = left stoppers strictly before the next entry
  }
  if (my_m) atomicMax(&red[0], my_m);
  if (my_cl != 0xffffffffu) atomicMin(&red[1], my_cl);
  if (my_cr != 0xffffffffu) atomicMin(&red[2], my_cr);
  __syncthreads();
  if (t == 0) {
    if (red[0]) atomicMax(&s->m, red[0]);
    if (red[1] != 0xffffffffu) atomicMin(&s->cut_l, red[1]);
    if (red[2] != 0xffffffffu) atomicMin(&s->cut_r, red[2]);
  }
}

// Pass 3: swap pair k for k < m.
__global__ __launch_bounds__(256) void k_part_swap(OrderEntry* __restrict__ a, size_t first,
                                                   const PartScalars* __restrict__ s,
                                                   const unsigned* __restrict__ pos_l,
                                                   const unsigned* __restrict__ pos_r) {
  const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= s->m) return;
  order_swap(a, first + pos_l[k], first + pos_r[k]);
}

// ------------------------------------- quick-select descent decided on the device (round 3) --
// What the search driver needs of the sorted order before its stopping rule can fire is the SET
// of the first `last + 1` entries (guetzli_amd/host/lazy_sort.h: SelectPrefix): libstdc++'s
// introsort partitions the whole range, then the part that holds position `last`, and so on
// until that part is small enough to be finished on the host.  Driven from the host, every one
// of those partitions was five dependent launches and a round trip (~75 us each, 2-6 per
// iteration of phase B).  Here the whole descent is enqueued at once, right behind the
// construction of the order: per level TWO launches and no host in between --
//   k_desc_count  every workgroup finds the pivot for itself (median of three: four reads),
//                 flags the left / right stoppers of its chunk against it and writes their
//                 positions, in order, into the chunk's own segments of two lists;
//   k_desc_swap   every workgroup sums the chunk counts for itself (a few thousand numbers:
//                 no scan kernel, no look-back chain across the chip), thread k finds the k-th
//                 left stopper and the k-th right stopper from the right through those sums and
//                 swaps them if they have not crossed; the one thread that sees "pair k - 1
//                 swapped, pair k not" knows the cut, decides which side holds `last` and
//                 writes the next level's range.
// Round 4: a wavefront takes ROWS of 64 consecutive entries (k_desc_count) / pairs (k_desc_swap) --
// loads of 512 contiguous bytes, a stopper's rank from the rows' lane masks -- and both kernels run
// on a bounded grid that loops over the chunks / groups of pairs (a workgroup per group of the
// largest possible order queued the idle ones behind the working ones: DESIGN.md 3.6).
// libstdc++ first moves the median to the front of the range (std::__move_median_to_first);
// both kernels read the range as if that swap had happened (position `med` holds the old front
// element) and k_desc_swap's first thread makes it real.  The arrangement and the cuts are those
// of gz_order_partition (and of std::sort): tests/cpp/test_device_order.cc.
constexpr int kDescMaxChunks = 4096;      // chunk tables of a workgroup: ranges up to 8.4 M entries (33 KB of LDS)
constexpr int kDescMaxChunksBig = 16384;  // the instantiation for larger orders: up to 33.5 M entries (131 KB: one workgroup per CU)
constexpr int kDescMaxLevels = 12;
constexpr int kDescCountGrid = 2048;   // workgroups of k_desc_count (eight fit on a CU)
constexpr int kDescSwapGrid = 1024;    // workgroups of k_desc_swap (four fit on a CU: 33 KB of LDS)

struct DescState {                 // the range before level l (level 0: derived, see desc_load)
  unsigned long long lo, hi;       // the range that holds `last`
  unsigned long long last;
  unsigned long long cut;          // cut of the partition that led here
  int depth;                       // introsort's depth budget
  unsigned epoch;                  // valid iff equal to the descent's epoch
};
struct DescPivot {
  OrderEntry pivot, a_lo;          // the median of three; the element it changes places with
  unsigned long long med;          // where the median was found
};
struct DescArgs {
  OrderEntry* a;
  DescState* st;                   // [kDescMaxLevels + 3]: the ranges, the results slot (publish), k_desc_export's
  DescPivot* pv;                   // [kDescMaxLevels]
  unsigned* cnt_l;                 // [chunks]
  unsigned* cnt_r;
  unsigned* lpos;                  // [chunks * kPartChunk]: positions relative to lo + 1
  unsigned* rpos;
  unsigned max_chunks;             // of the k_desc_swap instantiation that is launched (its tables)
  unsigned epoch;
  unsigned long long threshold;    // ranges up to this size are left to the host
  int derive;                      // 0: n0 / last0; 1: from the order construction's results
  unsigned long long n0, last0;
  const unsigned long long* total; // number of entries (the offsets scan's last element)
  const unsigned* counters;        // [0] blocks_to_change
  float per_block;                 // coefficients to change per block (processor.cc:685-687)
  // publish: the first level also copies what the host waits for at this point of an iteration
  // into st[kDescMaxLevels + 1] -- lo = the order's size, hi = blocks_to_change, last = entries
  // below the limit, cut = the bits of the last Compare's distance, depth = 1 -- so that ONE
  // transfer of st brings everything (three small device-to-host copies less on the stream).
  int publish;
  const unsigned* max_bits;
};

// The range level `level` works on; false = nothing to do at this level.
GZ_DEVFN bool desc_load(const DescArgs& A, int level, DescState* s) {
  if (level == 0) {
    const unsigned long long n = A.derive ? *A.total : A.n0;
    if (n == 0) return false;
    s->lo = 0;
    s->hi = n;
    int lg = 0;
    for (unsigned long long m = n; m > 1; m >>= 1) ++lg;
    s->depth = 2 * lg;
    if (A.derive) {
      // min_coeffs_to_change = coeffs_to_change_per_block * blocks_to_change (float product,
      // truncated: processor.cc:685-687); the stopping rule cannot fire before that many steps,
      // the entropy codes are refreshed every 10th (:739-741)
      int min_coeffs = (int)(A.per_block * (float)(int)A.counters[0]);
      if (min_coeffs < 0) min_coeffs = 0;
      unsigned long long last_needed = (unsigned long long)min_coeffs;
      if (last_needed > n - 1) last_needed = n - 1;
      const unsigned long long fast_until = last_needed / 10 * 10;
      s->last = fast_until ? fast_until - 1 : 0;
    } else {
      s->last = A.last0;
    }
    s->cut = 0;
    s->epoch = A.epoch;
  } else {
    *s = A.st[level];
    if (s->epoch != A.epoch) return false;
  }
  const unsigned long long len = s->hi - s->lo;
  return len > A.threshold && len > 16 && s->depth > 0 &&
         len - 1 <= (unsigned long long)A.max_chunks * kPartChunk;
}

GZ_DEVFN OrderEntry desc_read(const OrderEntry* a, unsigned long long p, unsigned long long med,
                              const OrderEntry& a_lo) {
  return p == med ? a_lo : a[p];
}

__global__ __launch_bounds__(256) void k_desc_count(DescArgs A, int level) {
  __shared__ unsigned wsum[2][8];
  if (level == 0 && A.publish && blockIdx.x == 0 && threadIdx.x == 0) {
    DescState r;
    r.lo = *A.total;
    r.hi = A.counters[0];
    r.last = A.counters[1];
    r.cut = A.max_bits ? *A.max_bits : 0u;
    r.depth = 1;
    r.epoch = A.epoch;
    A.st[kDescMaxLevels + 1] = r;
  }
  DescState s;
  if (!desc_load(A, level, &s)) return;
  const unsigned long long first = s.lo + 1;
  const unsigned n = (unsigned)(s.hi - first);
  const unsigned nchunks = (n + kPartChunk - 1) / kPartChunk;
  if (blockIdx.x >= nchunks) return;   // (the grid: at most kDescCountGrid workgroups, looping over the chunks)
  // std::__move_median_to_first(lo, lo + 1, mid, hi - 1)
  const unsigned long long x = s.lo + 1, y = s.lo + (s.hi - s.lo) / 2, z = s.hi - 1;
  const OrderEntry ex = A.a[x], ey = A.a[y], ez = A.a[z], e0 = A.a[s.lo];
  unsigned long long med;
  if (order_less(ex, ey)) {
    if (order_less(ey, ez)) med = y;
    else if (order_less(ex, ez)) med = z;
    else med = x;
  } else if (order_less(ex, ez)) {
    med = x;
  } else if (order_less(ey, ez)) {
    med = z;
  } else {
    med = y;
  }
  const OrderEntry pv = med == x ? ex : (med == y ? ey : ez);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    DescPivot o;
    o.pivot = pv;
    o.a_lo = e0;
    o.med = med;
    A.pv[level] = o;
  }
  // A wavefront takes kPartItems rows of 64 consecutive entries (512 bytes per load instruction);
  // the rank of a stopper inside the chunk comes from the rows' lane masks.
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int par = 0;   // (two sets of the wavefront sums: a chunk's are read while the next one's are written)
  for (unsigned chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x, par ^= 1) {
  const unsigned wbase = chunk * (unsigned)kPartChunk + (unsigned)wave * (64u * kPartItems);
  unsigned long long bl[kPartItems], br[kPartItems];
  unsigned tl = 0, tr = 0;
#pragma unroll
  for (int i = 0; i < kPartItems; ++i) {
    const unsigned p = wbase + (unsigned)i * 64u + (unsigned)lane;
    bool l = false, r = false;
    if (p < n) {
      const OrderEntry e = desc_read(A.a, first + p, med, e0);
      l = !order_less(e, pv);
      r = !order_less(pv, e);
    }
    bl[i] = __ballot(l);
    br[i] = __ballot(r);
    tl += (unsigned)GZ_POPC64(bl[i]);
    tr += (unsigned)GZ_POPC64(br[i]);
  }
  if (lane == 0) {
    wsum[par][wave] = tl;
    wsum[par][4 + wave] = tr;
  }
  __syncthreads();
  unsigned ol = chunk * (unsigned)kPartChunk, orr = ol;
  for (int wv = 0; wv < wave; ++wv) {
    ol += wsum[par][wv];
    orr += wsum[par][4 + wv];
  }
  const unsigned long long below = GZ_LANES_BELOW(lane);
#pragma unroll
  for (int i = 0; i < kPartItems; ++i) {
    const unsigned p = wbase + (unsigned)i * 64u + (unsigned)lane;
    if ((bl[i] >> lane) & 1ull) A.lpos[ol + (unsigned)GZ_POPC64(bl[i] & below)] = p;
    if ((br[i] >> lane) & 1ull) A.rpos[orr + (unsigned)GZ_POPC64(br[i] & below)] = p;
    ol += (unsigned)GZ_POPC64(bl[i]);
    orr += (unsigned)GZ_POPC64(br[i]);
  }
  if (t == 0) {
    A.cnt_l[chunk] = wsum[par][0] + wsum[par][1] + wsum[par][2] + wsum[par][3];
    A.cnt_r[chunk] = wsum[par][4] + wsum[par][5] + wsum[par][6] + wsum[par][7];
  }
  }   // for chunk
}

// First index in tab[0 .. n] whose value exceeds k, minus one (tab ascending, tab[0] <= k < tab[n]).
GZ_DEVFN unsigned desc_chunk_of(const unsigned* tab, unsigned n, unsigned k) {
  unsigned lo = 0, hi = n;   // tab[lo] <= k < tab[hi]
  while (hi - lo > 1) {
    const unsigned mid = (lo + hi) >> 1;
    if (tab[mid] <= k) lo = mid; else hi = mid;
  }
  return lo;
}

template <int MAXC>
__global__ __launch_bounds__(256) void k_desc_swap(DescArgs A, int level) {
  __shared__ unsigned PL[MAXC + 1];   // left stoppers in the chunks before c
  __shared__ unsigned RR[MAXC + 1];   // right stoppers in the chunks after nchunks-1-i
  __shared__ unsigned lds[256];
  DescState s;
  if (!desc_load(A, level, &s)) return;
  const unsigned long long first = s.lo + 1;
  const unsigned n = (unsigned)(s.hi - first);
  const unsigned nchunks = (n + kPartChunk - 1) / kPartChunk;
  // (the grid is sized for the largest order: there are at most n pairs)
  if ((unsigned long long)blockIdx.x * (unsigned)kPartChunk > n) return;
  const DescPivot pvt = A.pv[level];   // (asked for here: it arrives while the tables are built)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  {  // the two tables, by every workgroup for itself
    const unsigned per = (nchunks + 255) / 256;
    const unsigned c0 = t * per < nchunks ? t * per : nchunks;
    const unsigned c1 = c0 + per < nchunks ? c0 + per : nchunks;
    unsigned sl = 0, sr = 0;
    for (unsigned c = c0; c < c1; ++c) {
      sl += A.cnt_l[c];
      sr += A.cnt_r[nchunks - 1 - c];
    }
    unsigned run_l = wg_inclusive_scan_fast(sl, lds) - sl;
    unsigned run_r = wg_inclusive_scan_fast(sr, lds) - sr;
    for (unsigned c = c0; c < c1; ++c) {
      PL[c] = run_l;
      RR[c] = run_r;
      run_l += A.cnt_l[c];
      run_r += A.cnt_r[nchunks - 1 - c];
    }
    if (c1 == nchunks && c0 < nchunks) {
      PL[nchunks] = run_l;
      RR[nchunks] = run_r;
    }
    __syncthreads();
  }
  // (no workgroup barrier below this line: the wavefronts go their own ways)
  const unsigned total_l = PL[nchunks], total_r = RR[nchunks];
  const unsigned K = total_l < total_r ? total_l : total_r;   // pairs 0 .. K-1 exist; "pair" K ends the scan
  // the wavefront's pairs: kPartItems rows of 64 consecutive k (their positions sit next to each
  // other in the chunks' lists, and the stoppers themselves nearly so)
  // The grid has at most kDescSwapGrid workgroups, each with its tables built once: group g of
  // kPartChunk pairs goes to workgroup g % gridDim.x.  (A workgroup per group of the LARGEST
  // possible order -- 3 400 at 4K, 33 KB of LDS each, four to a CU -- made the groups that have
  // nothing to do queue behind the ones that work: 47 us for the first level where the same
  // kernel on a grid of the order's own size takes 21, profiles/r04_occupancy_experiments.log.)
  for (unsigned g = blockIdx.x;; g += gridDim.x) {
  const unsigned kw = g * (unsigned)kPartChunk + (unsigned)wave * (64u * kPartItems);
  if (kw > K) return;   // (and so are the workgroup's later groups)
  // positions (relative to `first`) of the k-th left stopper / the k-th right stopper from the right
  auto find_l = [&](unsigned k) {
    const unsigned c = desc_chunk_of(PL, nchunks, k);
    return A.lpos[c * (unsigned)kPartChunk + (k - PL[c])];
  };
  auto find_r = [&](unsigned k) {
    const unsigned i = desc_chunk_of(RR, nchunks, k);
    const unsigned c = nchunks - 1 - i, cnt = RR[i + 1] - RR[i];
    return A.rpos[c * (unsigned)kPartChunk + (cnt - 1 - (k - RR[i]))];
  };
  // The chunks of the pair BEFORE the wavefront's first one, found by the wavefront together (64
  // table entries per step instead of one): every lane walks on from there -- its k is at most
  // 64 * kPartItems further.
  const unsigned kp = kw > 0 ? kw - 1 : 0;
  auto chunk_by_wave = [&](const unsigned* tab, unsigned k) {
    // tab[0 .. nchunks] ascending, tab[0] <= k < tab[nchunks]: the last index with tab[i] <= k
    unsigned lo = 0, len = nchunks;               // the answer lies in [lo, lo + len)
    while (len > 1) {
      const unsigned step = (len + 63) / 64;      // lane j looks at lo + j * step
      const unsigned i = lo + (unsigned)lane * step;
      const unsigned long long le = __ballot(i < lo + len && tab[i] <= k);
      const unsigned j = (unsigned)GZ_POPC64(le) - 1u;   // (lane 0 always holds: tab[lo] <= k)
      const unsigned nlo = lo + j * step;
      len = nlo + step <= lo + len ? step : lo + len - nlo;
      lo = nlo;
    }
    return lo;
  };
  unsigned cl = 0, ir = 0;
  if (kp < total_l) cl = chunk_by_wave(PL, kp);
  if (kp < total_r) ir = chunk_by_wave(RR, kp);
  // the predecessor pair and the kPartItems rows in ONE round of look-ups (they do not depend on
  // one another): a wavefront behind the crossing finds out a little later than it could, the
  // others save a trip to memory
  unsigned prev_pl = 0, prev_pr = 0;
  if (kw > 0) {   // (kp < K: both exist)
    prev_pl = A.lpos[cl * (unsigned)kPartChunk + (kp - PL[cl])];
    const unsigned c = nchunks - 1 - ir, cnt = RR[ir + 1] - RR[ir];
    prev_pr = A.rpos[c * (unsigned)kPartChunk + (cnt - 1 - (kp - RR[ir]))];
  }
  unsigned pl[kPartItems], pr[kPartItems];
  bool sw[kPartItems];
#pragma unroll
  for (int i = 0; i < kPartItems; ++i) {
    const unsigned k = kw + (unsigned)i * 64u + (unsigned)lane;
    const bool has_l = k <= K && k < total_l, has_r = k <= K && k < total_r;
    pl[i] = 0xffffffffu;
    pr[i] = 0u;
    if (has_l) {
      while (PL[cl + 1] <= k) ++cl;
      pl[i] = A.lpos[cl * (unsigned)kPartChunk + (k - PL[cl])];
    }
    if (has_r) {
      while (RR[ir + 1] <= k) ++ir;
      const unsigned c = nchunks - 1 - ir, cnt = RR[ir + 1] - RR[ir];
      pr[i] = A.rpos[c * (unsigned)kPartChunk + (cnt - 1 - (k - RR[ir]))];
    }
    sw[i] = has_l && has_r && pl[i] < pr[i];
  }
  // Left stoppers ascend and right stoppers descend with k: once a pair has crossed, every later
  // one has.  A wavefront behind the crossing has nothing to swap and no cut to find -- about
  // half of them (the pivot is a median of three).
  const bool prev_swapped = kw == 0 || prev_pl < prev_pr;
  if (!prev_swapped) return;   // (the later groups are behind the crossing too)
  OrderEntry vl[kPartItems], vr[kPartItems];
#pragma unroll
  for (int i = 0; i < kPartItems; ++i)
    if (sw[i]) {
      vl[i] = desc_read(A.a, first + pl[i], pvt.med, pvt.a_lo);
      vr[i] = desc_read(A.a, first + pr[i], pvt.med, pvt.a_lo);
    }
#pragma unroll
  for (int i = 0; i < kPartItems; ++i)
    if (sw[i]) {
      A.a[first + pl[i]] = vr[i];
      A.a[first + pr[i]] = vl[i];
    }
  // the one pair that is not swapped while its predecessor was: where the serial scan stops -- at
  // the next untouched left stopper or at the last swapped right one, whichever comes first
  // (libstdc++ __unguarded_partition returns `first`).  The swapped pairs of a wavefront are a
  // prefix of its rows and lanes: the first row whose mask is not full holds that pair.
  bool all_before = true;
#pragma unroll
  for (int i = 0; i < kPartItems; ++i) {
    const unsigned long long m = __ballot(sw[i]);
    if (all_before && m != ~0ull) {   // (the same in every lane)
      all_before = false;
      const int j = GZ_POPC64(m);     // lanes 0 .. j-1 swapped: lane j holds the pair
      unsigned last_pr = prev_pr;     // the right stopper of the pair before it
      if (j > 0) last_pr = (unsigned)__shfl((int)pr[i], j - 1);
      else if (i > 0) last_pr = (unsigned)__shfl((int)pr[i > 0 ? i - 1 : 0], 63);
      const unsigned k = kw + (unsigned)i * 64u + (unsigned)lane;
      if (lane == j && k <= K) {
        unsigned long long cut = s.hi;
        if (pl[i] != 0xffffffffu && first + pl[i] < cut) cut = first + pl[i];
        if (k >= 1 && first + last_pr < cut) cut = first + last_pr;
        DescState nx;
        if (s.last < cut) { nx.lo = s.lo; nx.hi = cut; } else { nx.lo = cut; nx.hi = s.hi; }
        nx.last = s.last;
        nx.cut = cut;
        nx.depth = s.depth - 1;
        nx.epoch = A.epoch;
        A.st[level + 1] = nx;
        if (level == 0) A.st[0] = s;   // (the derived level-0 range, for the host's replay)
      }
    }
  }
  if (g == 0 && t == 0) {
    // the median's move to the front, made real: the front gets the pivot; the place the pivot
    // came from gets the old front element unless a pair above has already put a partner there
    A.a[s.lo] = pvt.pivot;
    const unsigned pm = (unsigned)(pvt.med - first);
    const unsigned c = pm / (unsigned)kPartChunk, ci = nchunks - 1 - c;
    bool moved = false;
    if (!order_less(pvt.a_lo, pvt.pivot)) {   // a left stopper: its rank among them
      const unsigned* seg = A.lpos + c * (unsigned)kPartChunk;
      unsigned lo = 0, hi = PL[c + 1] - PL[c];
      while (lo < hi) { const unsigned mid = (lo + hi) >> 1; if (seg[mid] < pm) lo = mid + 1; else hi = mid; }
      const unsigned k = PL[c] + lo;
      if (k < total_r && pm < find_r(k)) moved = true;
    }
    if (!moved && !order_less(pvt.pivot, pvt.a_lo)) {   // a right stopper
      const unsigned* seg = A.rpos + c * (unsigned)kPartChunk;
      const unsigned cnt = RR[ci + 1] - RR[ci];
      unsigned lo = 0, hi = cnt;
      while (lo < hi) { const unsigned mid = (lo + hi) >> 1; if (seg[mid] < pm) lo = mid + 1; else hi = mid; }
      const unsigned k = RR[ci] + (cnt - 1 - lo);
      if (k < total_l && find_l(k) < pm) moved = true;
    }
    if (!moved) A.a[pvt.med] = pvt.a_lo;
  }
  }   // for g
}

// Behind the last level of a descent: the leading entries of the order up to the end of the range
// the descent ended in -- what the search driver fetches next (guetzli_amd/host/processor.cc:
// DeviceOrder::Prefetch) -- written straight into the context's page-locked host mirror, when the
// descent got as far as the driver's own condition asks (range <= threshold) and the prefix fits
// max_entries.  st[kDescMaxLevels + 2] tells the host: lo = entries exported (0: none), depth = 2.
// One dispatch on the stream instead of a host round trip after it.
__global__ __launch_bounds__(256) void k_desc_export(DescArgs A, int levels, OrderEntry* __restrict__ dst,
                                                     unsigned long long max_entries,
                                                     DescState* __restrict__ host_state) {
  __shared__ unsigned long long s_hi;
  __shared__ DescState s_own;
  if (threadIdx.x == 0) {
    unsigned long long lo = 0, hi = 0;
    bool any = false;
    for (int l = 1; l <= levels; ++l) {
      const DescState st = A.st[l];
      if (st.epoch != A.epoch) break;
      lo = st.lo;
      hi = st.hi;
      any = true;
    }
    const bool ok = any && hi - lo <= A.threshold && hi <= max_entries;
    s_hi = ok ? hi : 0ull;
    if (blockIdx.x == 0) {
      DescState r;
      r.lo = s_hi; r.hi = 0; r.last = 0; r.cut = 0; r.depth = 2; r.epoch = A.epoch;
      A.st[kDescMaxLevels + 2] = r;
      s_own = r;
    }
  }
  __syncthreads();
  // the descent's state (ranges, results slot, this kernel's own slot) for the host, straight into
  // its page-locked copy: thread 0 of the first workgroup has just written the last of it
  if (blockIdx.x == 0 && host_state && threadIdx.x < kDescMaxLevels + 3)
    host_state[threadIdx.x] = threadIdx.x == kDescMaxLevels + 2 ? s_own : A.st[threadIdx.x];
  const unsigned long long n = s_hi;
  // 16 bytes (two entries) per thread and step
  const unsigned long long pairs = n >> 1;
  struct alignas(16) Two { OrderEntry a, b; };
  const Two* src2 = reinterpret_cast<const Two*>(A.a);
  Two* dst2 = reinterpret_cast<Two*>(dst);
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < pairs;
       i += (unsigned long long)gridDim.x * blockDim.x)
    dst2[i] = src2[i];
  if ((n & 1ull) && blockIdx.x == 0 && threadIdx.x == 0) dst[n - 1] = A.a[n - 1];
}

}  // namespace gz
