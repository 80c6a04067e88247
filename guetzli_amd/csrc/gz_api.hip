// C-ABI implementation (include/guetzli_amd.h): per-image context, device plane arena,
// blur plans, and the kernel sequences for the block path and for
// ButteraugliComparator::Compare.  Host code only orchestrates; every per-pixel /
// per-block operation is in the gz_kernels_*.h kernels.
//
// Built by hipcc for gfx950 with -ffp-contract=off (guetzli_amd/build.py).  There is no
// CPU path: without a usable HIP device gz_create fails with GZ_E_NO_DEVICE.

#include "../../include/guetzli_amd.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <unordered_map>
#include <string>
#include <sched.h>
#include <thread>
#include <utility>
#include <vector>

#include "gz_common.h"
#include "gz_kernels_block.h"
#include "gz_kernels_blur.h"
#include "gz_kernels_diff.h"
#include "gz_kernels_search.h"
#include "gz_kernels_entropy.h"
#include "gz_kernels_dctd.h"
#include "gz_kernels_downsample.h"
#include "gz_kernels_order.h"
#include "gz_kernels_rank.h"
#include "gz_host_weights.h"
#include "order_tables_generated.h"   // host-side csf/bias of order.inc

using namespace gz;

namespace {

// ------------------------------------------------------------------ blur plans ------
// Host-side restatement of ComputeKernel (butteraugli.cc:145-154) and of the border
// normalisation of ConvolveBorderColumn (:156-181).  exp() is evaluated on the host with
// the same libm float overload the reference uses; the device never recomputes taps.
struct BlurCfg {
  float sigma, border_ratio;
  int r;
  std::vector<float> k, ks;
  float wsum;
  // device border scales for the x axis (length w) and the y axis (length h)
  float* d_scale = nullptr;   // 4*r floats: x.lo, x.hi, y.lo, y.hi
  BorderScale bx, by;
};

void make_taps_host(float sigma, BlurCfg* c) {
  const float m = 2.25;
  const float scaler = -1.0 / (2 * sigma * sigma);
  const int diff = std::max<int>(1, m * fabsf(sigma));
  c->sigma = sigma;
  c->r = diff;
  c->k.resize(2 * diff + 1);
  for (int i = -diff; i <= diff; ++i) c->k[i + diff] = expf(scaler * i * i);
  float w = 0.0f;
  for (size_t j = 0; j < c->k.size(); ++j) w += c->k[j];
  c->wsum = w;
  const float s = 1.0f / w;
  c->ks = c->k;
  for (size_t j = 0; j < c->ks.size(); ++j) c->ks[j] *= s;
}

// scale(x) for a border position x on an axis of length n.
float border_scale(const BlurCfg& c, int n, int x) {
  const int r = c.r;
  const int lo = x < r ? 0 : x - r;
  const int hi = std::min(n - 1, x + r);
  float weight = 0.0f;
  for (int j = lo; j <= hi; ++j) weight += c.k[j - x + r];
  weight = (1.0f - c.border_ratio) * weight + c.border_ratio * c.wsum;
  return 1.0f / weight;
}

void border_scales_host(const BlurCfg& c, int n, std::vector<float>* lo,
                        std::vector<float>* hi) {
  lo->assign(c.r, 1.0f);
  hi->assign(c.r, 1.0f);
  for (int i = 0; i < c.r; ++i) {
    if (i < n) (*lo)[i] = border_scale(c, n, i);
    if (n - 1 - i >= 0) (*hi)[i] = border_scale(c, n, n - 1 - i);
  }
}

template <int R>
Taps<R> taps_of(const BlurCfg& c) {
  Taps<R> t;
  for (int j = 0; j <= 2 * R; ++j) {
    t.k[j] = c.k[j];
    t.ks[j] = c.ks[j];
  }
  return t;
}

enum BlurId { B_OPSIN, B_LF, B_MF, B_HF, B_SN, B_MASKX, B_MASKY0, B_MASKY1, B_FINAL, B_COUNT };
struct BlurSpec { double sigma, border; int r; };
// sigmas / border ratios: butteraugli.cc:329, :497-508, :885, :1757-1760, :737-740
const BlurSpec kBlurSpecs[B_COUNT] = {
  {1.2, 0.0, 2},
  {7.46953768697, -0.00457628248637, 16},
  {3.734768843485, -0.271277366628, 8},
  {1.8673844217425, 0.147068973249, 4},
  {10.6666499623, 0.0, 23},
  {9.24456601467, -0.0724948220913, 20},
  {2.3770330432, -0.0724948220913, 5},
  {9.04353323561, -0.0724948220913, 20},
  {1.72547472444, 1.0, 3},
};

// MakeMask (butteraugli.cc:1638-1653) for MaskX / MaskY / MaskDcX / MaskDcY (:1655-1697)
void make_mask_lut(double extmul, double extoff, double mul, double offset, double scaler,
                   double* lut) {
  const double kGlobalScale = 1.0 / 20.35;
  for (int i = 0; i < 512; ++i) {
    const double c = mul / ((0.01 * scaler * i) + offset);
    lut[i] = kGlobalScale * (1.0 + extmul * (c + extoff));
    if (lut[i] < 1e-5) lut[i] = 1e-5;
    lut[i] *= lut[i];
  }
}

// Malta normalisation constants (MaltaDiffMapImpl, butteraugli.cc:1468-1476)
MaltaNorm malta_norm(bool lf, double w_0gt1, double w_0lt1, double norm1) {
  const double len = 3.75;
  const double mulli = lf ? 0.405371989604 : 0.354191303559;
  const float kWeight0 = 0.5;
  const float kWeight1 = 0.33;
  const double w_pre0gt1 = mulli * sqrt(kWeight0 * w_0gt1) / (len * 2 + 1);
  const double w_pre0lt1 = mulli * sqrt(kWeight1 * w_0lt1) / (len * 2 + 1);
  MaltaNorm n;
  n.norm2_0gt1 = w_pre0gt1 * norm1;
  n.norm2_0lt1 = w_pre0lt1 * norm1;
  n.norm1f = static_cast<float>(norm1);
  auto mid = [](float x) { return x >= 0x1p-40f && x <= 0x1p40f; };
  // (norm1f: malta_diff then needs one comparison for its denominator, norm1f + absval >= norm1f)
  n.fast_div = mid(n.norm2_0gt1) && mid(n.norm2_0lt1) && n.norm1f >= 0x1p-40f && n.norm1f <= 0x1p39f ? 1 : 0;
  return n;
}

// The six Malta passes of DiffmapPsychoImage (butteraugli.cc:835-874): [channel X/Y][band
// UHF, HF, MF] -> normalisation and tap pattern.
struct MaltaSpec {
  MaltaNorm nm;
  int lf;
};
void malta_specs(MaltaSpec out[2][3]) {
  const float hf_asymmetry_ = 0.8f;
  const double wUhfMalta = 5.1409625726, norm1Uhf = 58.5001247061;
  const double wUhfMaltaX = 4.91743441556, norm1UhfX = 687196.39002;
  const double wHfMalta = 153.671655716, norm1Hf = 83150785.9592;
  const double wHfMaltaX = 668.358918152, norm1HfX = 0.882954368025;
  const double wMfMalta = 6841.81248144, norm1Mf = 0.0135134962487;
  const double wMfMaltaX = 813.901703816, norm1MfX = 16792.9322251;
  const float sqrt_asym = sqrtf(hf_asymmetry_);   // float sqrt overload in the reference
  out[1][0] = {malta_norm(false, wUhfMalta * hf_asymmetry_, wUhfMalta / hf_asymmetry_, norm1Uhf), 0};
  out[1][1] = {malta_norm(true, wHfMalta * sqrt_asym, wHfMalta / sqrt_asym, norm1Hf), 1};
  out[1][2] = {malta_norm(true, wMfMalta, wMfMalta, norm1Mf), 1};
  out[0][0] = {malta_norm(false, wUhfMaltaX * hf_asymmetry_, wUhfMaltaX / hf_asymmetry_, norm1UhfX), 0};
  out[0][1] = {malta_norm(true, wHfMaltaX * sqrt_asym, wHfMaltaX / sqrt_asym, norm1HfX), 1};
  out[0][2] = {malta_norm(true, wMfMaltaX, wMfMaltaX, norm1MfX), 1};
}

struct Psycho {   // device planes of one image's PsychoImage (butteraugli.h:418-423)
  float* lfv[3];  // lf in "vals" space
  float* mf[2];   // X, Y  (mf[2] of the reference is dead: wmul[5] == 0)
  float* hf[2];
  float* uhf[2];
};

}  // namespace

// ------------------------------------------------------------ caching allocator ------
// One image = one context = ~45 device allocations (0.5 GB at 1080p, 2 GB at 4K) and three
// pinned host buffers; hipMalloc / hipFree (which also synchronises the device) of those cost
// ~10 ms per image.  Freed blocks are kept in exact-size free lists per device and handed to
// the next context that asks for the same size -- a batch of same-sized images allocates once.
// GZ_POOL_MB bounds the cached device bytes per device (default 16384; 0 = no caching);
// gz_trim_pool() releases everything cached.  The emulation build allocates directly, so that
// its poisoning of fresh memory keeps catching reads of never-written buffers.
namespace {
// page-locked AND mapped into the device's address space: k_apply_coeff_edits and k_desc_export access
// staging buffers directly (the default flags give that on ROCm; said explicitly)
#ifdef GZ_EMU
constexpr unsigned kHostAllocFlags = 0;
#else
constexpr unsigned kHostAllocFlags = hipHostMallocMapped;
#endif
struct MemPool {
  std::mutex mu;
  std::unordered_map<void*, std::pair<int, size_t> > live;          // ptr -> (device, bytes)
  std::multimap<std::pair<int, size_t>, void*> idle;                // (device, bytes) -> ptr
  std::unordered_map<int, size_t> idle_bytes;                       // per device
};
MemPool& dev_pool() { static MemPool p; return p; }
MemPool& host_pool() { static MemPool p; return p; }
size_t pool_limit_bytes() {
  static const size_t lim = [] {
    const char* e = getenv("GZ_POOL_MB");
    return (size_t)(e ? std::max(0L, atol(e)) : 16384L) << 20;
  }();
  return lim;
}
void pool_release_idle(MemPool& p, bool host, int device /* -1: all */) {
  for (auto it = p.idle.begin(); it != p.idle.end();) {
    if (device >= 0 && it->first.first != device) { ++it; continue; }
    if (host) (void)hipHostFree(it->second); else (void)hipFree(it->second);
    p.idle_bytes[it->first.first] -= it->first.second;
    it = p.idle.erase(it);
  }
}
hipError_t pool_alloc(MemPool& p, bool host, void** out, size_t bytes) {
  if (bytes == 0) bytes = 1;
#ifdef GZ_EMU
  return host ? hipHostMalloc(out, bytes, kHostAllocFlags) : hipMalloc(out, bytes);
#else
  int device = 0;
  (void)hipGetDevice(&device);
  std::lock_guard<std::mutex> lk(p.mu);
  auto it = p.idle.find(std::make_pair(device, bytes));
  if (it != p.idle.end()) {
    *out = it->second;
    p.idle.erase(it);
    p.idle_bytes[device] -= bytes;
    p.live[*out] = std::make_pair(device, bytes);
    return hipSuccess;
  }
  hipError_t e = host ? hipHostMalloc(out, bytes, kHostAllocFlags) : hipMalloc(out, bytes);
  if (e != hipSuccess) {   // make room: drop what is cached on this device and try once more
    (void)hipGetLastError();
    pool_release_idle(p, host, device);
    e = host ? hipHostMalloc(out, bytes, kHostAllocFlags) : hipMalloc(out, bytes);
  }
  if (e == hipSuccess) p.live[*out] = std::make_pair(device, bytes);
  return e;
#endif
}
void pool_release(MemPool& p, bool host, void* ptr) {
  if (!ptr) return;
#ifdef GZ_EMU
  if (host) (void)hipHostFree(ptr); else (void)hipFree(ptr);
#else
  std::lock_guard<std::mutex> lk(p.mu);
  auto it = p.live.find(ptr);
  if (it == p.live.end()) {   // not ours
    if (host) (void)hipHostFree(ptr); else (void)hipFree(ptr);
    return;
  }
  const std::pair<int, size_t> key = it->second;
  p.live.erase(it);
  if (key.second <= pool_limit_bytes() && p.idle_bytes[key.first] + key.second > pool_limit_bytes())
    pool_release_idle(p, host, key.first);   // full of sizes nobody asks for any more: start over
  if (p.idle_bytes[key.first] + key.second <= pool_limit_bytes()) {
    p.idle.insert(std::make_pair(key, ptr));
    p.idle_bytes[key.first] += key.second;
  } else if (host) {
    (void)hipHostFree(ptr);
  } else {
    (void)hipFree(ptr);
  }
#endif
}
inline hipError_t pool_malloc(void** out, size_t bytes) { return pool_alloc(dev_pool(), false, out, bytes); }
inline void pool_free(void* ptr) { pool_release(dev_pool(), false, ptr); }
inline hipError_t pool_host_malloc(void** out, size_t bytes) { return pool_alloc(host_pool(), true, out, bytes); }
inline void pool_host_free(void* ptr) { pool_release(host_pool(), true, ptr); }

// Streams and (timing-less) events are pooled the same way: creating and destroying a
// context's four streams costs several milliseconds.  Only idle ones come back (the context
// synchronises its streams before it returns them).
struct HandlePool {
  std::mutex mu;
  std::multimap<int, hipStream_t> streams;   // device * 4 + priority class -> stream
  std::multimap<int, hipEvent_t> events;
};
HandlePool& handle_pool() { static HandlePool p; return p; }
// prio: 0 = default, +1 = the device's highest priority, -1 = its lowest.  Who takes which:
// create_context.
static bool stream_priorities() { return true; }
// Contexts alive per device: adds `delta`, returns the count before.
static int live_contexts(int device, int delta) {
  static std::mutex mu;
  static std::map<int, int> live;
  std::lock_guard<std::mutex> lk(mu);
  const int before = live[device];
  live[device] = before + delta;
  return before;
}
hipError_t pool_stream_create(hipStream_t* out, int prio = 0) {
#ifndef GZ_EMU
  if (!stream_priorities()) prio = 0;
  int device = 0;
  (void)hipGetDevice(&device);
  const int key = device * 4 + (prio + 1);
  {
    HandlePool& p = handle_pool();
    std::lock_guard<std::mutex> lk(p.mu);
    auto it = p.streams.find(key);
    if (it != p.streams.end()) { *out = it->second; p.streams.erase(it); return hipSuccess; }
  }
  if (prio != 0) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
      return hipStreamCreateWithPriority(out, hipStreamDefault, prio > 0 ? greatest : least);
  }
#endif
  return hipStreamCreate(out);
}
void pool_stream_destroy(hipStream_t s_, int prio = 0) {
  if (!s_) return;
#ifndef GZ_EMU
  if (!stream_priorities()) prio = 0;
  if (pool_limit_bytes() != 0) {
    int device = 0;
    (void)hipGetDevice(&device);
    const int key = device * 4 + (prio + 1);
    HandlePool& p = handle_pool();
    std::lock_guard<std::mutex> lk(p.mu);
    if (p.streams.count(key) < 64) { p.streams.insert(std::make_pair(key, s_)); return; }
  }
#endif
  (void)hipStreamDestroy(s_);
}
hipError_t pool_event_create(hipEvent_t* out) {
#ifndef GZ_EMU
  int device = 0;
  (void)hipGetDevice(&device);
  {
    HandlePool& p = handle_pool();
    std::lock_guard<std::mutex> lk(p.mu);
    auto it = p.events.find(device);
    if (it != p.events.end()) { *out = it->second; p.events.erase(it); return hipSuccess; }
  }
#endif
  return hipEventCreateWithFlags(out, hipEventDisableTiming);
}
void pool_event_destroy(hipEvent_t e_) {
  if (!e_) return;
#ifndef GZ_EMU
  if (pool_limit_bytes() != 0) {
    int device = 0;
    (void)hipGetDevice(&device);
    HandlePool& p = handle_pool();
    std::lock_guard<std::mutex> lk(p.mu);
    if (p.events.count(device) < 256) { p.events.insert(std::make_pair(device, e_)); return; }
  }
#endif
  (void)hipEventDestroy(e_);
}
}  // namespace

// Pinned host staging for the small per-iteration uploads (step lists, coefficient edits,
// next_cand, Huffman codes): the caller's buffer is copied here, the H2D copy is asynchronous
// and nobody has to wait for it -- the buffer is only waited for when it is reused.
struct HostStage {
  void* h = nullptr;
  size_t cap = 0;
  hipEvent_t ev = nullptr;
  bool busy = false;
};

struct gz_ctx {
  int device = 0;
  int w = 0, h = 0, bw = 0, bh = 0, nb = 0, pitch = 0;
  size_t plane = 0;   // floats per plane
  // The current frame (OutputImage's component layout): chroma subsampling factor 1 (4:4:4)
  // or 2 (4:2:0: OutputImageComponent::Reset(2, 2), output_image.cc:40-49), the chroma block
  // grid under it, the first block of every component in d_orig / d_cand, blocks in total.
  int cfac = 1, cbw = 0, cbh = 0, nbc = 0, coff[3] = {0, 0, 0}, nblk = 0;
  uint8_t* d_csamp = nullptr;      // 4:2:0: IDCT samples of the two chroma components (k_chroma_samples)
  // grid of the last block search (gz_block_zeroing_orders*), which phase B's order works on
  int sg_w = 0, sg_h = 0, sg_n = 0, sg_factor = 1, sg_mask = 7;
  float* d_gmax = nullptr;         // per-16x16 maxima of the distance map (sg_factor == 2)
  // scratch of k_scan_offsets, one set per stream that runs it (main: order build; entropy: scan)
  void* d_scan_state[2] = {nullptr, nullptr};
  unsigned scan_epoch[2] = {0, 0};
  float target = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  // second stream for the branch of Compare that does not depend on the Malta path (the
  // mask: DiffPrecompute + three blurs), forked and joined with events
  hipStream_t side_stream = nullptr, side_stream2 = nullptr;
  bool prio_streams = false, counted_live = false;   // (see create_context)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join2 = nullptr, ev_mask_pre = nullptr;
  hipEvent_t ev_next_cand = nullptr;   // next_cand uploaded beside a Compare chain in flight
  hipEvent_t ev_xyb = nullptr, ev_lfy = nullptr;   // B plane's LF blur on side stream 2 (stage_separate)
  // the entropy coder's kernels (gz_jpeg_scan) run on their own stream, beside a Compare that
  // gz_compare_begin has put on the main stream: both only read the candidate coefficients
  hipStream_t entropy_stream = nullptr;
  hipEvent_t ev_candidate = nullptr;   // main stream: the candidate is in place
  std::string err;

  uint8_t* d_rgb = nullptr;
  int16_t* d_orig = nullptr;   // [3][nb][64] original coefficients
  int16_t* d_cand = nullptr;   // candidate coefficients
  int* d_q = nullptr;          // [3][64]
  float* d_srgb_lut = nullptr; // float(Srgb8ToLinearTable[i])
  double* d_mask_luts = nullptr;
  float* d_block_max = nullptr;
  unsigned* d_max_bits = nullptr;
  uint8_t* d_srgb_out = nullptr;
  int32_t* d_blkidx = nullptr; size_t blkidx_cap = 0;
  int16_t* d_blkdata = nullptr;

  float* arena = nullptr;
  float* extra_arena = nullptr;   // probe-only planes (ensure_pip)
  std::vector<float*> free_planes;
  BlurCfg blur[B_COUNT];

  Psycho pi0;   // original
  Psycho pi1;   // candidate
  Psycho pip;   // probe "image 0" (allocated lazily)
  bool have_pip = false;
  // scratch
  float *lin[3], *tmp[3], *xyb[3], *lf_raw[2], *hfp[2];
  float *snb, *diffx, *diffy, *mxb, *myb1, *myb2, *ac[2], *dsq, *distmap;
  float* sup0[2];   // the original's half of DiffPrecompute (k_mask_sup of pi0: X, Y), per image
  float* sup_scratch[2] = {nullptr, nullptr};   // the same for Mask() on raw planes (block mask, probe)
  float *mask_out[3], *mask_dc_out[3];
  bool have_mask_out = false;

  float* d_block_mask = nullptr;   // [3][nb] mask_xyz_ at block corners (StartBlockComparisons)
  bool have_block_mask = false;
  int32_t* d_rank_cnt = nullptr; uint8_t* d_rank_idx = nullptr; float* d_rank_tables = nullptr;
  int32_t* d_out_cnt = nullptr; uint8_t* d_out_idx = nullptr; float* d_out_err = nullptr;

  // device entropy coder (gz_kernels_entropy.h)
  int* d_jq = nullptr;                    // [3][64] quant matrices of the frame being written
  unsigned* d_hist = nullptr;             // [2][3][256]
  unsigned char* d_code_depth = nullptr;  // [2][3][256]
  unsigned short* d_code_bits = nullptr;  // [2][3][256]
  unsigned* d_mcu_bits = nullptr;         // [nb]
  unsigned long long* d_mcu_off = nullptr;   // [nb+1]
  unsigned long long* d_ff_count = nullptr;
  unsigned* d_words = nullptr; size_t words_cap = 0;        // scan bits of the last gz_jpeg_scan
  unsigned* d_words_kept = nullptr; size_t words_kept_cap = 0;
  unsigned long long scan_bits = 0, scan_ff = 0, kept_bits = 0, kept_ff = 0;
  bool have_jq = false, have_scan = false, have_kept = false;
  bool scan_pending = false;           // between gz_jpeg_scan_begin and _end
  void* h_scan_result = nullptr;       // pinned: total bits, 0xFF count of the scan in flight

  // global candidate order of phase B (gz_kernels_order.h)
  OrderEntry* d_order = nullptr; size_t order_cap = 0; size_t order_n = 0;
  unsigned* d_pos_l = nullptr; unsigned* d_pos_r = nullptr;       // [order_cap]
  unsigned* d_chunk = nullptr; size_t chunk_cap = 0;              // cnt_l, cnt_r, base_l, base_r
  PartScalars* d_part = nullptr;
  // gz_order_build_auto_begin .. _end: results land here (pinned; not the shared landing area,
  // which gz_compare_end uses in between)
  struct OrderPending { unsigned long long total; unsigned counters[2]; };
  OrderPending* h_order_pending = nullptr;
  bool order_pending = false;
  // quick-select descent decided on the device (gz_order_descend*): per-level ranges and pivots,
  // the ranges' pinned copy for the host's replay
  DescState* d_desc_st = nullptr; DescPivot* d_desc_pv = nullptr; DescState* h_desc = nullptr;
  unsigned desc_epoch = 0; int desc_levels = 0; bool desc_pending = false;
  // gz_order_build_auto_descend_begin: the order's counters (and the distance of the Compare in
  // flight) arrive with the descent's state, in h_desc[kDescMaxLevels + 1]
  void* h_order_mirror = nullptr;      // gz_order_host_mirror: pinned, order entries land in it directly
  size_t order_mirror_cap = 0;         // entries
  bool results_in_desc = false, distance_in_desc = false;
  unsigned results_epoch = 0;          // the descent (desc_epoch) that published them
  unsigned export_epoch = 0;           // the descent whose k_desc_export wrote into the host mirror (0: none)
  unsigned* d_order_nb = nullptr;                                 // [nb]
  unsigned long long* d_order_off = nullptr;                      // [nb+1]: [nb] = the order's size; the first 4 nb BYTES: every block's offset inside its group
  unsigned* d_order_counters = nullptr;                           // [2]
  unsigned* d_order_groups = nullptr;                             // [2 * ceil(nb / kOrderGroup)]: sum of n_b, blocks with n_b > 0
  int* d_next_cand = nullptr; float* d_weight = nullptr; float* d_max_err = nullptr;   // [nb]
  bool have_search = false;
  unsigned char* d_wflag = nullptr;                               // [nb]
  int* d_edit_pos = nullptr; short* d_edit_val = nullptr; size_t edit_cap = 0;

  bool have_orig = false, have_cand = false, have_distmap = false;
  std::vector<float> h_block_max;
  bool h_block_max_valid = false;
  bool compare_pending = false;
  int h_jq[192] = {0};       // the matrix d_jq holds
  unsigned* d_step_delta = nullptr; bool have_step_delta = false;   // AC statistics change of the last bulk steps
  HostStage stage_main, stage_entropy;
  HostStage stage_edits;   // gz_apply_coeff_edits' own: its kernel reads the buffer, and the next order's upload (stage_main) must not wait for it
  // pinned landing area for the small results every call waits for (a copy into pageable
  // memory costs 27 us per round trip on this system, into pinned memory 15)
  void* h_res = nullptr; size_t h_res_cap = 0;
  void* d_cmp_stage = nullptr; size_t cmp_stage_cap = 0;   // gz_compare_blocks / _block_pixels staging
  size_t search_total = 0;   // candidates phase A produced (bounds every global order)
  unsigned long long search_evaluations = 0;   // CompareBlock evaluations of the last block search
  float last_distance = 0.0f;
};

// Every context entry point runs with the context's device current and leaves the caller's
// device as it found it: a thread may own contexts on several GPUs (the pools key on the
// current device, kernels launch on it).
struct DeviceScope {
  int prev = -1;
  explicit DeviceScope(const gz_ctx* c) {
    if (!c) return;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); cur = -1; }
    if (cur != c->device) {
      (void)hipSetDevice(c->device);
      prev = cur;
    }
  }
  ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
};

namespace {

#define HIPCHK(ctx, call)                                                            \
  do {                                                                               \
    hipError_t e_ = (call);                                                          \
    if (e_ != hipSuccess) {                                                          \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                \
      return GZ_E_HIP;                                                               \
    }                                                                                \
  } while (0)

#define KCHK(ctx)                                                                    \
  do {                                                                               \
    hipError_t e_ = hipGetLastError();                                               \
    if (e_ != hipSuccess) {                                                          \
      (ctx)->err = std::string("kernel launch: ") + hipGetErrorString(e_);           \
      return GZ_E_HIP;                                                               \
    }                                                                                \
  } while (0)

const int kNumPlanes = 9 + 9 + 3 + 3 + 3 + 2 + 2 + 10 + 2;   // pi0, pi1, lin, tmp, xyb, lf_raw, hfp, 10 singles, sup0[2]

void set_frame(gz_ctx* c, int factor) {
  c->cfac = factor;
  c->cbw = (c->w + 8 * factor - 1) / (8 * factor);
  c->cbh = (c->h + 8 * factor - 1) / (8 * factor);
  c->nbc = c->cbw * c->cbh;
  c->coff[0] = 0;
  c->coff[1] = c->nb;
  c->coff[2] = c->nb + c->nbc;
  c->nblk = c->nb + 2 * c->nbc;
  c->have_search = false;
  // whatever was pending or kept belonged to the old frame: a stale gz_order_build_auto_end /
  // gz_order_descend_end / gz_compare_end / gz_jpeg_scan_end must fail, not return its data
  c->order_pending = false;
  c->results_in_desc = false;
  c->desc_pending = false;
  c->distance_in_desc = false;
  c->compare_pending = false;
  c->scan_pending = false;
  c->have_distmap = false;
  c->have_scan = false;
  c->export_epoch = 0;
}
size_t csamp_plane(const gz_ctx* c) {   // bytes of one chroma sample plane of a 4:2:0 frame
  return (size_t)((c->w + 15) / 16 * 8) * (size_t)((c->h + 15) / 16 * 8);
}

float* take_plane(gz_ctx* c) {
  float* p = c->free_planes.back();
  c->free_planes.pop_back();
  return p;
}
void alloc_psycho(gz_ctx* c, Psycho* p) {
  for (int i = 0; i < 3; ++i) p->lfv[i] = take_plane(c);
  for (int i = 0; i < 2; ++i) p->mf[i] = take_plane(c);
  for (int i = 0; i < 2; ++i) p->hf[i] = take_plane(c);
  for (int i = 0; i < 2; ++i) p->uhf[i] = take_plane(c);
}

// ------------------------------------------------------------- blur dispatch helpers --
// Radius < 16: one fused launch per blur (k_blur2d); radius >= 16: a row pass and a column pass.
// Two measured crossovers pick the instantiation (both knobs are read per call, so that the tests
// run every one of them on images small enough for the emulation):
//  * GZ_BLUR_PK -- row-pair / column-pair passes with packed arithmetic (k_blur_h_pk, k_blur_v_pk:
//    twice the outputs per thread, half the LDS reads and address computations per output, half
//    the workgroups) from 4 MPix on, the one-output-row kernels (k_blur_h, k_blur_v_compact) below
//    (profiles/r02_packed_blur_ab.log, r03_chain_kernel_experiments.log);
//  * GZ_TILE_ROWS -- 64 x 32 tiles from 1.5 MPix on, 64 x 16 below: twice the workgroups for
//    256 CUs (720p: 0.290 -> 0.257 ms per Compare; no gain at 1080p, a small loss at 4K).
// What round 4 removed after it had lost every A/B of rounds 2 and 3: the unrolled (non-compact)
// column pass and fused kernels, 64-row tiles, the epilogue without 16-byte accesses and the row
// pass with LDS bank conflicts (GZ_BLUR_OPT), the three-plane LF passes, the unpaired mask blurs.
static bool packed_blur(const gz_ctx* c) {
  const char* e = getenv("GZ_BLUR_PK");
  if (e) return atoi(e) != 0;
  return (size_t)c->w * c->h >= 4000000;
}
constexpr int kTileRows = 32;
constexpr int kSmallTileRows = 16;
static bool small_tiles(const gz_ctx* c) {
  const char* e = getenv("GZ_TILE_ROWS");
  if (e && atoi(e) == 16) return true;
  if (e && atoi(e) == 32) return false;
  return (size_t)c->w * c->h < 1500000;
}

template <int R, class Src, int NC>
int blur_h(gz_ctx* c, const SrcPack<Src, NC>& src, const PlanePack<NC>& dst,
           const BlurCfg& cfg) {
  if (cfg.r != R) { c->err = "blur radius mismatch"; return GZ_E_STATE; }
  const Taps<R> tp = taps_of<R>(cfg);
  const BorderScale bs = cfg.bx;
  const int w = c->w, h = c->h, pitch = c->pitch;
  if (packed_blur(c)) {
    dim3 grid(gz_div_up(c->w, HW), gz_div_up(c->h, HP), NC);
    GZ_LAUNCH((k_blur_h_pk<R, Src, NC>), grid, dim3(256), c->stream, src, dst, w, h, pitch, tp, bs, tp, bs);
  } else {
    dim3 grid(gz_div_up(c->w, HW), gz_div_up(c->h, HH), NC);
    GZ_LAUNCH((k_blur_h<R, Src, NC>), grid, dim3(256), c->stream, src, dst, w, h, pitch, tp, bs, tp, bs);
  }
  KCHK(c);
  return GZ_OK;
}

template <int R, int NC, class Post>
int blur_v(gz_ctx* c, const CPlanePack<NC>& src, const Post& post, const BlurCfg& cfg) {
  if (cfg.r != R) { c->err = "blur radius mismatch"; return GZ_E_STATE; }
  const Taps<R> tp = taps_of<R>(cfg);
  const BorderScale bs = cfg.by;
  const int w = c->w, h = c->h, pitch = c->pitch;
  const bool small = small_tiles(c);
  dim3 grid(gz_div_up(c->w, VW), gz_div_up(c->h, small ? kSmallTileRows : kTileRows));
  if (packed_blur(c)) {
    if (small) GZ_LAUNCH((k_blur_v_pk<R, NC, Post, kSmallTileRows>), grid, dim3(256), c->stream, src, post, w, h, pitch, tp, bs, tp, bs);
    else GZ_LAUNCH((k_blur_v_pk<R, NC, Post, kTileRows>), grid, dim3(256), c->stream, src, post, w, h, pitch, tp, bs, tp, bs);
  } else {
    if (small) GZ_LAUNCH((k_blur_v_compact<R, NC, Post, kSmallTileRows>), grid, dim3(256), c->stream, src, post, w, h, pitch, tp, bs, tp, bs);
    else GZ_LAUNCH((k_blur_v_compact<R, NC, Post, kTileRows>), grid, dim3(256), c->stream, src, post, w, h, pitch, tp, bs, tp, bs);
  }
  KCHK(c);
  return GZ_OK;
}

// Two blurs of equal radius and different sigma on two independent planes as ONE launch per
// pass (grid z = plane): the mask's radius-20 pair (butteraugli.cc:1780-1790).
template <int R, class Src>
int blur_h_pair(gz_ctx* c, const SrcPack<Src, 2>& src, const PlanePack<2>& dst, const BlurCfg& cfg0,
                const BlurCfg& cfg1) {
  if (cfg0.r != R || cfg1.r != R) { c->err = "blur radius mismatch"; return GZ_E_STATE; }
  const Taps<R> t0 = taps_of<R>(cfg0), t1 = taps_of<R>(cfg1);
  const BorderScale b0 = cfg0.bx, b1 = cfg1.bx;
  const int w = c->w, h = c->h, pitch = c->pitch;
  if (packed_blur(c)) {
    dim3 grid(gz_div_up(c->w, HW), gz_div_up(c->h, HP), 2);
    GZ_LAUNCH((k_blur_h_pk<R, Src, 2, true>), grid, dim3(256), c->stream, src, dst, w, h, pitch, t0, b0, t1, b1);
  } else {
    dim3 grid(gz_div_up(c->w, HW), gz_div_up(c->h, HH), 2);
    GZ_LAUNCH((k_blur_h<R, Src, 2, true>), grid, dim3(256), c->stream, src, dst, w, h, pitch, t0, b0, t1, b1);
  }
  KCHK(c);
  return GZ_OK;
}
template <int R>
int blur_v_pair(gz_ctx* c, const CPlanePack<2>& src, const PostStore<2>& post, const BlurCfg& cfg0,
                const BlurCfg& cfg1) {
  if (cfg0.r != R || cfg1.r != R) { c->err = "blur radius mismatch"; return GZ_E_STATE; }
  const Taps<R> t0 = taps_of<R>(cfg0), t1 = taps_of<R>(cfg1);
  const BorderScale b0 = cfg0.by, b1 = cfg1.by;
  const int w = c->w, h = c->h, pitch = c->pitch;
  const bool small = small_tiles(c);
  dim3 grid(gz_div_up(c->w, VW), gz_div_up(c->h, small ? kSmallTileRows : kTileRows), 2);
  if (packed_blur(c)) {
    if (small) GZ_LAUNCH((k_blur_v_pk<R, 2, PostStore<2>, kSmallTileRows, true>), grid, dim3(256), c->stream, src, post, w, h, pitch, t0, b0, t1, b1);
    else GZ_LAUNCH((k_blur_v_pk<R, 2, PostStore<2>, kTileRows, true>), grid, dim3(256), c->stream, src, post, w, h, pitch, t0, b0, t1, b1);
  } else {
    if (small) GZ_LAUNCH((k_blur_v_compact<R, 2, PostStore<2>, kSmallTileRows, true>), grid, dim3(256), c->stream, src, post, w, h, pitch, t0, b0, t1, b1);
    else GZ_LAUNCH((k_blur_v_compact<R, 2, PostStore<2>, kTileRows, true>), grid, dim3(256), c->stream, src, post, w, h, pitch, t0, b0, t1, b1);
  }
  KCHK(c);
  return GZ_OK;
}

// BM = true (the chain's last blur): the Post functor's results are also reduced to the per-block
// maxima and the image maximum; that kernel keeps its results in registers (always 32-row tiles).
template <int R, int NC, class Src, class Post, bool BM = false>
int blur2d(gz_ctx* c, const SrcPack<Src, NC>& src, const Post& post, const BlurCfg& cfg,
           BlockMaxOut bm = BlockMaxOut{nullptr, nullptr, 0}) {
  if (cfg.r != R) { c->err = "blur radius mismatch"; return GZ_E_STATE; }
  const Taps<R> tp = taps_of<R>(cfg);
  const BorderScale bx = cfg.bx, by = cfg.by;
  const int w = c->w, h = c->h, pitch = c->pitch;
  if (!BM && small_tiles(c)) {
    dim3 grid(gz_div_up(c->w, T2), gz_div_up(c->h, kSmallTileRows));
    GZ_LAUNCH((k_blur2d<R, NC, Src, Post, false, kSmallTileRows>), grid, dim3(256), c->stream, src, post, w,
              h, pitch, tp, bx, by, bm);
  } else {
    dim3 grid(gz_div_up(c->w, T2), gz_div_up(c->h, kTileRows));
    GZ_LAUNCH((k_blur2d<R, NC, Src, Post, BM, kTileRows>), grid, dim3(256), c->stream, src, post, w,
              h, pitch, tp, bx, by, bm);
  }
  KCHK(c);
  return GZ_OK;
}

#define TRY(x) do { int rc_ = (x); if (rc_ != GZ_OK) return rc_; } while (0)

// Reserves `bytes` of the staging buffer (waiting for its previous upload if that is still
// running) and returns it; stage_sent() marks the upload that was just enqueued on `stream`.
static int stage_reserve(gz_ctx* c, HostStage* st, size_t bytes, void** out) {
  if (!st->ev) HIPCHK(c, pool_event_create(&st->ev));
  if (st->busy) {
    HIPCHK(c, hipEventSynchronize(st->ev));
    st->busy = false;
  }
  if (bytes > st->cap) {
    if (st->h) (void)pool_host_free(st->h);
    st->h = nullptr;
    st->cap = 0;
    const size_t cap = bytes + bytes / 2 + 4096;
    HIPCHK(c, pool_host_malloc(&st->h, cap));
    st->cap = cap;
  }
  *out = st->h;
  return GZ_OK;
}
static int stage_sent(gz_ctx* c, HostStage* st, hipStream_t stream) {
  HIPCHK(c, hipEventRecord(st->ev, stream));
  st->busy = true;
  return GZ_OK;
}
static int result_buffer(gz_ctx* c, size_t bytes, void** out) {
  if (bytes > c->h_res_cap) {
    if (c->h_res) (void)pool_host_free(c->h_res);
    c->h_res = nullptr;
    c->h_res_cap = 0;
    const size_t cap = std::max<size_t>(bytes + bytes / 2, 1 << 16);
    HIPCHK(c, pool_host_malloc(&c->h_res, cap));
    c->h_res_cap = cap;
  }
  *out = c->h_res;
  return GZ_OK;
}
static void stage_free(HostStage* st) {
  if (st->ev) { (void)hipEventSynchronize(st->ev); pool_event_destroy(st->ev); }
  if (st->h) (void)pool_host_free(st->h);
  st->h = nullptr; st->ev = nullptr; st->cap = 0; st->busy = false;
}

int setup_blur_cfg(gz_ctx* c, BlurCfg* cfg, float sigma, float border_ratio) {
  make_taps_host(sigma, cfg);
  cfg->border_ratio = border_ratio;
  std::vector<float> xl, xh, yl, yh;
  border_scales_host(*cfg, c->w, &xl, &xh);
  border_scales_host(*cfg, c->h, &yl, &yh);
  const int r = cfg->r;
  if (cfg->d_scale == nullptr) HIPCHK(c, pool_malloc((void**)&cfg->d_scale, sizeof(float) * 4 * r));
  std::vector<float> all;
  all.insert(all.end(), xl.begin(), xl.end());
  all.insert(all.end(), xh.begin(), xh.end());
  all.insert(all.end(), yl.begin(), yl.end());
  all.insert(all.end(), yh.begin(), yh.end());
  HIPCHK(c, hipMemcpy(cfg->d_scale, all.data(), sizeof(float) * 4 * r, hipMemcpyHostToDevice));
  cfg->bx.lo = cfg->d_scale;
  cfg->bx.hi = cfg->d_scale + r;
  cfg->by.lo = cfg->d_scale + 2 * r;
  cfg->by.hi = cfg->d_scale + 3 * r;
  return GZ_OK;
}

// --------------------------------------------------------------- pipeline stages ------
// OpsinDynamicsImage: lin[3] -> xyb[3]
int stage_opsin(gz_ctx* c) {
  SrcPack<SrcPlain, 3> s;
  for (int i = 0; i < 3; ++i) s.s[i].p = c->lin[i];
  PostOpsin post;
  for (int i = 0; i < 3; ++i) { post.lin[i] = c->lin[i]; post.xyb[i] = c->xyb[i]; }
  TRY((blur2d<2, 3, SrcPlain, PostOpsin>(c, s, post, c->blur[B_OPSIN])));
  return GZ_OK;
}

// SeparateFrequencies: xyb[3] -> Psycho planes
// The LF blur (radius 16) runs as X / Y (two planes, PostLFxy) and B (one plane, PostLFb: its
// XybLowFreqToVals mixes in the raw LF of Y the first wrote, butteraugli.cc:386-389).  side_b (the
// candidate's chain, unless single-stream): B -- which only k_combine reads -- goes to side stream
// 2, beside the MF / HF bands instead of in front of them; the caller joins that stream before
// k_combine (join_mask_branch).  Its row-pass result goes through the distance-map plane, which
// nothing else touches before the chain's last kernel.
int stage_separate(gz_ctx* c, Psycho* ps, bool side_b = false) {
  hipStream_t main_stream = c->stream;
  hipStream_t b_stream = side_b ? c->side_stream2 : c->stream;
  int rc = GZ_OK;
  if (side_b) {
    HIPCHK(c, hipEventRecord(c->ev_xyb, c->stream));
    HIPCHK(c, hipStreamWaitEvent(b_stream, c->ev_xyb, 0));
  }
  {
    c->stream = b_stream;
    SrcPack<SrcPlain, 1> s; PlanePack<1> t;
    s.s[0].p = c->xyb[2]; t.p[0] = c->distmap;
    rc = blur_h<16, SrcPlain, 1>(c, s, t, c->blur[B_LF]);
    c->stream = main_stream;
    TRY(rc);
  }
  {
    SrcPack<SrcPlain, 2> s; PlanePack<2> t; CPlanePack<2> ct;
    for (int i = 0; i < 2; ++i) { s.s[i].p = c->xyb[i]; t.p[i] = c->tmp[i]; ct.p[i] = c->tmp[i]; }
    TRY((blur_h<16, SrcPlain, 2>(c, s, t, c->blur[B_LF])));
    PostLFxy post;
    for (int i = 0; i < 2; ++i) { post.lf_raw[i] = c->lf_raw[i]; post.lf_vals[i] = ps->lfv[i]; }
    TRY((blur_v<16, 2, PostLFxy>(c, ct, post, c->blur[B_LF])));
  }
  if (side_b) {
    HIPCHK(c, hipEventRecord(c->ev_lfy, c->stream));
    HIPCHK(c, hipStreamWaitEvent(b_stream, c->ev_lfy, 0));
  }
  {
    c->stream = b_stream;
    CPlanePack<1> ct; ct.p[0] = c->distmap;
    PostLFb post; post.lf_raw_y = c->lf_raw[1]; post.lf_vals_b = ps->lfv[2];
    rc = blur_v<16, 1, PostLFb>(c, ct, post, c->blur[B_LF]);
    c->stream = main_stream;
    TRY(rc);
  }
  {  // MF (X, Y)
    SrcPack<SrcDiff, 2> s;
    for (int i = 0; i < 2; ++i) {
      s.s[i].a = c->xyb[i];
      s.s[i].b = c->lf_raw[i];
    }
    PostMF post;
    for (int i = 0; i < 2; ++i) {
      post.xyb[i] = c->xyb[i];
      post.lf_raw[i] = c->lf_raw[i];
      post.mf[i] = ps->mf[i];
      post.hf_pre[i] = c->hfp[i];
    }
    TRY((blur2d<8, 2, SrcDiff, PostMF>(c, s, post, c->blur[B_MF])));
  }
  {  // HF / UHF
    SrcPack<SrcPlain, 2> s;
    for (int i = 0; i < 2; ++i) s.s[i].p = c->hfp[i];
    PostHF post;
    for (int i = 0; i < 2; ++i) {
      post.hf_pre[i] = c->hfp[i];
      post.hf[i] = ps->hf[i];
      post.uhf[i] = ps->uhf[i];
    }
    post.lf_raw_y = c->lf_raw[1];
    TRY((blur2d<4, 2, SrcPlain, PostHF>(c, s, post, c->blur[B_HF])));
  }
  return GZ_OK;
}

// Mask first half: DiffPrecompute + three blurs -> mxb, myb1, myb2.  The three blurs only share
// their input: the two of radius 20 (X: sigma r2 = 9.24; Y second: sigma r1 = 9.04 -- separate
// taps) are one launch per pass (grid z = plane); with `other` given, the small one (radius 5)
// goes behind whatever is queued there (the SameNoise blur, the shorter of the two side branches).
int stage_mask_blurs(gz_ctx* c, const MaskPrePack& pk, hipStream_t other = nullptr) {
  dim3 grid(gz_div_up(c->w, 1024), c->h, 2);
  GZ_LAUNCH(k_mask_pre, grid, dim3(256), c->stream, pk, c->w, c->h, c->pitch);
  KCHK(c);
  if (other) {
    HIPCHK(c, hipEventRecord(c->ev_mask_pre, c->stream));
    HIPCHK(c, hipStreamWaitEvent(other, c->ev_mask_pre, 0));
  }
  {
    SrcPack<SrcPlain, 2> s; PlanePack<2> t; CPlanePack<2> ct;
    s.s[0].p = c->diffx; s.s[1].p = c->diffy;
    t.p[0] = c->tmp[1]; t.p[1] = c->tmp[2];
    ct.p[0] = c->tmp[1]; ct.p[1] = c->tmp[2];
    TRY((blur_h_pair<20, SrcPlain>(c, s, t, c->blur[B_MASKX], c->blur[B_MASKY1])));
    PostStore<2> post; post.out[0] = c->mxb; post.out[1] = c->myb2;
    TRY((blur_v_pair<20>(c, ct, post, c->blur[B_MASKX], c->blur[B_MASKY1])));
  }
  {
    SrcPack<SrcPlain, 1> s; s.s[0].p = c->diffy;
    PostStore<1> post; post.out[0] = c->myb1;
    hipStream_t here = c->stream;
    if (other) c->stream = other;
    const int rc = blur2d<5, 1, SrcPlain, PostStore<1>>(c, s, post, c->blur[B_MASKY0]);
    c->stream = here;
    TRY(rc);
  }
  return GZ_OK;
}

// MaskPsychoImage's inputs (butteraugli.cc:753-782): a * uhf + b * hf of a PsychoImage, X and Y.
static void mask_in_psycho(const Psycho& p, MaskIn in[2]) {
  const double muls[4] = {0, 1.64178305129, 0.831081703362, 3.23680933546};   // (:759-764)
  for (int i = 0; i < 2; ++i) {
    in[i].a = muls[2 * i];
    in[i].b = muls[2 * i + 1];
    in[i].plain = 0;
    in[i].hf = p.hf[i];
    in[i].uhf = muls[2 * i] == 0 ? nullptr : p.uhf[i];
  }
}
// The original's half, once per image (gz_set_rgb): c->sup0.
int stage_mask_sup(gz_ctx* c, const MaskIn in[2], float* const out[2]) {
  MaskSupPack pk;
  for (int i = 0; i < 2; ++i) { pk.in[i] = in[i]; pk.out[i] = out[i]; }
  dim3 grid(gz_div_up(c->w, 1024), c->h, 2);
  GZ_LAUNCH(k_mask_sup, grid, dim3(256), c->stream, pk, c->w, c->h, c->pitch);
  KCHK(c);
  return GZ_OK;
}
MaskPrePack mask_pack_psycho(gz_ctx* c, const Psycho& b) {
  MaskPrePack pk;
  mask_in_psycho(b, pk.in1);
  pk.sup0[0] = c->sup0[0];
  pk.sup0[1] = c->sup0[1];
  pk.out[0] = c->diffx;
  pk.out[1] = c->diffy;
  return pk;
}
int ensure_pip(gz_ctx* c);
// Mask(xyb0, xyb1) on raw planes (StartBlockComparisons' mask of the original with itself, the
// stage probe): image 0's half goes through two scratch planes of the probe arena.
int mask_pack_plain(gz_ctx* c, const float* const a[2], const float* const b[2], MaskPrePack* pk) {
  MaskIn in0[2];
  for (int i = 0; i < 2; ++i) {
    in0[i] = {nullptr, a[i], 0.0, 1.0, 1};
    pk->in1[i] = {nullptr, b[i], 0.0, 1.0, 1};
  }
  TRY(ensure_pip(c));   // (sup_scratch)
  TRY(stage_mask_sup(c, in0, c->sup_scratch));
  pk->sup0[0] = c->sup_scratch[0];
  pk->sup0[1] = c->sup_scratch[1];
  pk->out[0] = c->diffx;
  pk->out[1] = c->diffy;
  return GZ_OK;
}

// The SameNoise blur and the mask branch (DiffPrecompute + three blurs; scratch planes
// tmp[0..2], snb, diffx, diffy, mxb, myb1, myb2) read only the two PsychoImages, so they run
// on the side stream while the main stream does Malta; k_combine needs both.  At 1080p a launch is
// ~1000 workgroups for 256 CUs and the kernels are latency-bound: the overlap is worth ~10 %.
static bool single_stream() {   // GZ_SINGLE_STREAM=1: no overlap, for per-kernel profiling
  static const char* e = getenv("GZ_SINGLE_STREAM");
  return e && atoi(e) != 0;
}
int fork_side_branch(gz_ctx* c, const Psycho& p0, const Psycho& p1) {
  if (single_stream()) {
    SrcPack<SrcSameNoise, 1> s; PlanePack<1> t; CPlanePack<1> ct;
    s.s[0].a = p0.hf[1]; s.s[0].b = p1.hf[1];
    t.p[0] = c->tmp[0]; ct.p[0] = c->tmp[0];
    TRY((blur_h<23, SrcSameNoise, 1>(c, s, t, c->blur[B_SN])));
    PostStore<1> post; post.out[0] = c->snb;
    TRY((blur_v<23, 1, PostStore<1>>(c, ct, post, c->blur[B_SN])));
    return stage_mask_blurs(c, mask_pack_psycho(c, p1));
  }
  HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
  HIPCHK(c, hipStreamWaitEvent(c->side_stream, c->ev_fork, 0));
  HIPCHK(c, hipStreamWaitEvent(c->side_stream2, c->ev_fork, 0));
  hipStream_t main_stream = c->stream;
  int rc = GZ_OK;
  c->stream = c->side_stream2;
  {  // SameNoiseLevels blur input + blur (sigma 10.67)
    SrcPack<SrcSameNoise, 1> s; PlanePack<1> t; CPlanePack<1> ct;
    s.s[0].a = p0.hf[1]; s.s[0].b = p1.hf[1];
    t.p[0] = c->tmp[0]; ct.p[0] = c->tmp[0];
    rc = blur_h<23, SrcSameNoise, 1>(c, s, t, c->blur[B_SN]);
    PostStore<1> post; post.out[0] = c->snb;
    if (rc == GZ_OK) rc = blur_v<23, 1, PostStore<1>>(c, ct, post, c->blur[B_SN]);
  }
  c->stream = c->side_stream;
  if (rc == GZ_OK) rc = stage_mask_blurs(c, mask_pack_psycho(c, p1), c->side_stream2);
  c->stream = main_stream;
  TRY(rc);
  HIPCHK(c, hipEventRecord(c->ev_join, c->side_stream));
  HIPCHK(c, hipEventRecord(c->ev_join2, c->side_stream2));
  return GZ_OK;
}
int join_mask_branch(gz_ctx* c) {
  if (single_stream()) return GZ_OK;
  HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
  HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join2, 0));
  return GZ_OK;
}

// DiffmapPsychoImage (butteraugli.cc:817-908) + score: p0 = original, p1 = candidate.
int stage_diffmap(gz_ctx* c, const Psycho& p0, const Psycho& p1, bool want_block_max,
                  bool max_cleared = false) {
  const float hf_asymmetry_ = 0.8f;
  // side stream: SameNoise blur + the mask branch; main stream: Malta
  TRY(fork_side_branch(c, p0, p1));
  MaltaSpec ms[2][3];
  malta_specs(ms);
  dim3 mgrid(gz_div_up(c->w, MW), gz_div_up(c->h, MH), 2);
  MaltaArgs<3> ay, ax;
  for (int ch = 0; ch < 2; ++ch) {   // X, Y; passes in the reference's order: UHF, HF, MF
    MaltaArgs<3>& a = ch ? ay : ax;
    a.pass[0] = {p0.uhf[ch], p1.uhf[ch], ms[ch][0].nm, ms[ch][0].lf};
    a.pass[1] = {p0.hf[ch], p1.hf[ch], ms[ch][1].nm, ms[ch][1].lf};
    a.pass[2] = {p0.mf[ch], p1.mf[ch], ms[ch][2].nm, ms[ch][2].lf};
    a.out = c->ac[ch];
  }
  GZ_LAUNCH((k_malta_rolled<3>), mgrid, dim3(256), c->stream, ay, ax, c->w, c->h, c->pitch);
  KCHK(c);
  TRY(join_mask_branch(c));
  {
    CombineArgs a;
    a.mask_x_blur = c->mxb; a.mask_y_blur1 = c->myb1; a.mask_y_blur2 = c->myb2;
    a.ac0 = c->ac[0]; a.ac1 = c->ac[1];
    a.lf0_x = p0.lfv[0]; a.lf1_x = p1.lfv[0];
    a.lf0_b = p0.lfv[2]; a.lf1_b = p1.lfv[2];
    a.luts = c->d_mask_luts;
    const double wmul1 = 32.4449876135;
    a.sn_blur = c->snb;
    a.hf0_y = p0.hf[1];
    a.hf1_y = p1.hf[1];
    a.w_sn = 884.809801415;
    a.w_0gt1 = (wmul1 * hf_asymmetry_) * 0.8;   // L2DiffAsymmetric: w *= 0.8 (:678-679)
    a.w_0lt1 = (wmul1 / hf_asymmetry_) * 0.8;
    a.out = c->dsq;
    for (int i = 0; i < 3; ++i) a.mask_out[i] = a.mask_dc_out[i] = nullptr;
    dim3 grid(gz_div_up(c->w, 1024), c->h);   // (4 pixels per thread)
    GZ_LAUNCH(k_combine, grid, dim3(256), c->stream, a, c->w, c->h, c->pitch);
    KCHK(c);
  }
  {  // CalculateDiffmap second half: blur(sigma 1.725, border_ratio 1.0) + mix
    SrcPack<SrcPlain, 1> s; s.s[0].p = c->dsq;
    PostDiffmapMix post; post.d = c->dsq; post.out = c->distmap;
    if (!max_cleared) HIPCHK(c, hipMemsetAsync(c->d_max_bits, 0, sizeof(unsigned), c->stream));
    BlockMaxOut bm{want_block_max ? c->d_block_max : nullptr, c->d_max_bits, c->bw};
    TRY((blur2d<3, 1, SrcPlain, PostDiffmapMix, true>(c, s, post, c->blur[B_FINAL], bm)));
  }
  return GZ_OK;
}

// Exclusive 64-bit prefix sums of n 32-bit values on `stream` (which: 0 = the main stream's
// scratch, 1 = the entropy stream's).
int enqueue_scan_offsets(gz_ctx* c, int which, hipStream_t stream, const unsigned* d_bits, int n,
                         unsigned long long* d_off) {
  const int max_tiles = gz_div_up(c->nb, kScanTile) + 1;
  const size_t bytes = (size_t)max_tiles * (8 + 8 + 4) + 64;
  if (!c->d_scan_state[which]) {
    HIPCHK(c, pool_malloc(&c->d_scan_state[which], bytes));
    HIPCHK(c, hipMemsetAsync(c->d_scan_state[which], 0, bytes, stream));   // ticket 0, no epoch yet
    c->scan_epoch[which] = 0;
  }
  char* base = (char*)c->d_scan_state[which];
  ScanState st;
  st.agg = (unsigned long long*)base;
  st.incl = st.agg + max_tiles;
  st.status = (unsigned*)(st.incl + max_tiles);
  st.ticket = st.status + max_tiles;
  unsigned ep = ++c->scan_epoch[which];
  if (ep >= 0x3fffffffu) {   // the epoch field of the flags would wrap: start over
    HIPCHK(c, hipMemsetAsync(c->d_scan_state[which], 0, bytes, stream));
    c->scan_epoch[which] = ep = 1;
  }
  if (n > max_tiles * kScanTile) { c->err = "scan larger than its scratch"; return GZ_E_STATE; }
  GZ_LAUNCH(k_scan_offsets, dim3(std::max(1, gz_div_up(n, kScanTile))), dim3(256), stream, d_bits, n, d_off, st, ep);
  KCHK(c);
  return GZ_OK;
}

int stage_chroma_samples(gz_ctx* c, const int16_t* d_coeffs) {
  if (!c->d_csamp) HIPCHK(c, pool_malloc((void**)&c->d_csamp, 2 * csamp_plane(c)));
  GZ_LAUNCH(k_chroma_samples, dim3(gz_div_up(c->nbc, kBlocksPerWG)), dim3(256), c->stream,
            d_coeffs + (size_t)c->coff[1] * 64, d_coeffs + (size_t)c->coff[2] * 64, c->cbw, c->nbc,
            c->d_csamp);
  KCHK(c);
  return GZ_OK;
}

int stage_reconstruct(gz_ctx* c, const int16_t* d_coeffs, float* lin0, uint8_t* srgb,
                      unsigned* clear_word = nullptr) {
  if (c->cfac == 2) {
    TRY(stage_chroma_samples(c, d_coeffs));
    GZ_LAUNCH(k_reconstruct420, dim3(c->bh * gz_div_up(c->bw, 8)), dim3(256), c->stream,
              d_coeffs, (const uint8_t*)c->d_csamp, c->w, c->h, c->bw, c->nb, c->cbw, c->cbh,
              c->pitch, c->plane, c->d_srgb_lut, lin0, srgb, clear_word);
    KCHK(c);
    return GZ_OK;
  }
  // strips of 8 blocks per workgroup: as many (up to 4) as leave the chip ~2000 workgroups (8 per CU)
  const int strips = gz_div_up(c->bw, kReconBlocks);
  int per = 4;
  while (per > 1 && (long)c->bh * gz_div_up(strips, per) < 2000) per >>= 1;
#ifdef GZ_EMU
  if (const char* e = getenv("GZ_EMU_RECON_STRIPS")) per = std::max(1, atoi(e));   // (the strip loop on images the emulation can afford)
#endif
  GZ_LAUNCH(k_reconstruct, dim3(c->bh * gz_div_up(strips, per)), dim3(256), c->stream,
            d_coeffs, c->w, c->h, c->bw, c->nb, c->pitch, c->plane, c->d_srgb_lut, lin0,
            srgb, clear_word, per);
  KCHK(c);
  return GZ_OK;
}

// One full Compare of the current candidate, everything on the stream.
int enqueue_compare(gz_ctx* c, bool want_block_max) {
  TRY(stage_reconstruct(c, c->d_cand, c->lin[0], nullptr, c->d_max_bits));
  TRY(stage_opsin(c));
  TRY(stage_separate(c, &c->pi1, !single_stream()));
  TRY(stage_diffmap(c, c->pi0, c->pi1, want_block_max, true));
  return GZ_OK;
}

int upload_planes(gz_ctx* c, const float* host, float* const* dev, int n) {
  for (int i = 0; i < n; ++i)
    HIPCHK(c, hipMemcpyAsync(dev[i], host + (size_t)i * c->w * c->h,
                             sizeof(float) * c->w * c->h, hipMemcpyHostToDevice, c->stream));
  return GZ_OK;
}
int download_plane(gz_ctx* c, const float* dev, float* host) {
  HIPCHK(c, hipMemcpyAsync(host, dev, sizeof(float) * c->w * c->h, hipMemcpyDeviceToHost,
                           c->stream));
  return GZ_OK;
}

int ensure_pip(gz_ctx* c) {
  if (c->have_pip) return GZ_OK;
  HIPCHK(c, pool_malloc((void**)&c->extra_arena, sizeof(float) * c->plane * 17));
  for (int i = 0; i < 17; ++i) c->free_planes.push_back(c->extra_arena + (size_t)i * c->plane);
  alloc_psycho(c, &c->pip);
  for (int i = 0; i < 3; ++i) { c->mask_out[i] = take_plane(c); c->mask_dc_out[i] = take_plane(c); }
  for (int i = 0; i < 2; ++i) c->sup_scratch[i] = take_plane(c);
  c->have_pip = true;
  return GZ_OK;
}


// StartBlockComparisons (butteraugli_comparator.cc:415-421): mask_xyz_ =
// Mask(opsin(orig), opsin(orig)).mask; only the values at block corners are ever read
// (CompareBlock, :484-486).
int ensure_block_mask(gz_ctx* c) {
  if (c->have_block_mask) return GZ_OK;
  TRY(ensure_pip(c));
  if (!c->d_block_mask) HIPCHK(c, pool_malloc((void**)&c->d_block_mask, sizeof(float) * 3 * c->nb));
  dim3 grid(gz_div_up(c->w, 256), c->h);
  GZ_LAUNCH(k_linear_from_rgb8, grid, dim3(256), c->stream, c->d_rgb, c->w, c->h, c->pitch,
            c->plane, c->d_srgb_lut, c->lin[0]);
  KCHK(c);
  TRY(stage_opsin(c));
  MaskPrePack pk;
  {
    const float* const x2[2] = {c->xyb[0], c->xyb[1]};
    TRY(mask_pack_plain(c, x2, x2, &pk));
  }
  TRY(stage_mask_blurs(c, pk));
  CombineArgs ca;
  memset(&ca, 0, sizeof(ca));
  ca.mask_x_blur = c->mxb; ca.mask_y_blur1 = c->myb1; ca.mask_y_blur2 = c->myb2;
  ca.luts = c->d_mask_luts;
  for (int i = 0; i < 3; ++i) { ca.mask_out[i] = c->mask_out[i]; ca.mask_dc_out[i] = nullptr; }
  GZ_LAUNCH(k_combine, dim3(gz_div_up(c->w, 1024), c->h), dim3(256), c->stream, ca, c->w, c->h, c->pitch);   // (4 pixels per thread)
  KCHK(c);
  GZ_LAUNCH(k_gather_block_corners, dim3(gz_div_up(c->nb, 256)), dim3(256), c->stream,
            (const float*)c->mask_out[0], (const float*)c->mask_out[1],
            (const float*)c->mask_out[2], c->pitch, c->bw, c->nb, c->d_block_mask);
  KCHK(c);
  c->have_block_mask = true;
  return GZ_OK;
}

// input_order of ComputeBlockZeroingOrder (processor.cc:381-400) for blocks [b0, b1):
// score = |orig| * csf + bias (order.inc), std::sort ascending on the score -- done with
// libstdc++'s std::sort on the same sequence the reference builds, because the order of
// equal scores is implementation-defined and feeds the JPEG bytes.
void rank_blocks(const int16_t* coeffs, const int16_t* orig, int nb, int new_model, int b0,
                 int b1, uint8_t* cnt, uint8_t* idx /* [nb][192] */) {
  static const uint8_t oldCsf[64] = {
      10, 10, 20, 40, 60, 70, 80, 90, 10, 20, 30, 60, 70, 80, 90, 90,
      20, 30, 60, 70, 80, 90, 90, 90, 40, 60, 70, 80, 90, 90, 90, 90,
      60, 70, 80, 90, 90, 90, 90, 90, 70, 80, 90, 90, 90, 90, 90, 90,
      80, 90, 90, 90, 90, 90, 90, 90, 90, 90, 90, 90, 90, 90, 90, 90};
  static const int zigzag[64] = {   // kJPEGZigZagOrder, jpeg_data.h:75-84
      0, 1, 5, 6, 14, 15, 27, 28, 2, 4, 7, 13, 16, 26, 29, 42,
      3, 8, 12, 17, 25, 30, 41, 43, 9, 11, 18, 24, 31, 40, 44, 53,
      10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60,
      21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};
  static const double kWeight[3] = {1.0, 0.22, 0.20};
  std::vector<std::pair<int, float> > order;
  order.reserve(192);
  for (int b = b0; b < b1; ++b) {
    order.clear();
    for (int ch = 0; ch < 3; ++ch) {
      const int16_t* blk = coeffs + ((size_t)ch * nb + b) * 64;
      const int16_t* ob = orig + ((size_t)ch * nb + b) * 64;
      for (int k = 1; k < 64; ++k) {
        if (blk[k] == 0) continue;
        const int i = ch * 64 + k;
        float score;
        if (new_model)
          score = abs((int)ob[k]) * kOrderCsf[i] + kOrderBias[i];
        else
          score = static_cast<float>((abs((int)ob[k]) - zigzag[k] / 64.0) * kWeight[ch] / oldCsf[k]);
        order.push_back(std::make_pair(i, score));
      }
    }
    std::sort(order.begin(), order.end(),
              [](const std::pair<int, float>& x, const std::pair<int, float>& y) {
                return x.second < y.second; });
    cnt[b] = (uint8_t)order.size();
    for (size_t i = 0; i < order.size(); ++i) idx[(size_t)b * 192 + i] = (uint8_t)order[i].first;
  }
}

void rank_all(const int16_t* coeffs, const int16_t* orig, int nb, int new_model,
              std::vector<int32_t>* off, std::vector<uint8_t>* idx) {
  std::vector<uint8_t> cnt(nb), wide((size_t)nb * 192);
  // threads from the cores this PROCESS may run on (a rank of a multi-GPU job is bound to its share
  // of the host: bench.py Env.bind_cpus), not from the machine's
  unsigned nt = std::thread::hardware_concurrency();
#if defined(__linux__)
  {
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) nt = (unsigned)CPU_COUNT(&set);
  }
#endif
  nt = std::max(1u, std::min(nt, 32u));
  if (nb < 4096) nt = 1;
  std::vector<std::thread> th;
  const int per = (nb + (int)nt - 1) / (int)nt;
  for (unsigned t = 0; t < nt; ++t) {
    const int b0 = (int)t * per, b1 = std::min(nb, b0 + per);
    if (b0 >= b1) break;
    th.emplace_back(rank_blocks, coeffs, orig, nb, new_model, b0, b1, cnt.data(), wide.data());
  }
  for (auto& t : th) t.join();
  off->resize(nb + 1);
  int total = 0;
  for (int b = 0; b < nb; ++b) { (*off)[b] = total; total += cnt[b]; }
  (*off)[nb] = total;
  idx->resize(total);
  for (int b = 0; b < nb; ++b)
    memcpy(idx->data() + (*off)[b], wide.data() + (size_t)b * 192, cnt[b]);
}

}  // namespace

// ===================================================================== C surface ======
extern "C" {

int gz_abi_version(void) { return 3; }

int gz_trim_pool(void) {
  {
    MemPool& p = dev_pool();
    std::lock_guard<std::mutex> lk(p.mu);
    pool_release_idle(p, false, -1);
  }
  {
    MemPool& h = host_pool();
    std::lock_guard<std::mutex> lk(h.mu);
    pool_release_idle(h, true, -1);
  }
  HandlePool& hp = handle_pool();
  std::lock_guard<std::mutex> lk(hp.mu);
  for (auto& kv : hp.streams) (void)hipStreamDestroy(kv.second);
  for (auto& kv : hp.events) (void)hipEventDestroy(kv.second);
  hp.streams.clear();
  hp.events.clear();
  return GZ_OK;
}

int gz_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return GZ_E_NO_DEVICE;
  return n;
}

const char* gz_strerror(int code) {
  switch (code) {
    case GZ_OK: return "ok";
    case GZ_E_ARG: return "invalid argument";
    case GZ_E_NO_DEVICE: return "no usable HIP device";
    case GZ_E_HIP: return "HIP runtime error";
    case GZ_E_STATE: return "invalid call sequence";
    case GZ_E_NOMEM: return "out of memory";
    default: return "unknown error";
  }
}

const char* gz_last_error(const gz_ctx* ctx) { return ctx ? ctx->err.c_str() : ""; }

static gz_ctx* create_context(int device, int w, int h, const uint8_t* rgb, float target, int* err);
gz_ctx* gz_create(int device, int w, int h, const uint8_t* rgb, float target, int* err) {
  int prev = -1;
  if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
  gz_ctx* c = create_context(device, w, h, rgb, target, err);
  if (prev >= 0 && prev != device) (void)hipSetDevice(prev);   // the caller's device stays current
  return c;
}
static gz_ctx* create_context(int device, int w, int h, const uint8_t* rgb, float target, int* err) {
  int dummy;
  if (!err) err = &dummy;
  *err = GZ_OK;
  if (!rgb || w < 8 || h < 8 || w >= (1 << 16) || h >= (1 << 16)) { *err = GZ_E_ARG; return nullptr; }
  // coefficient positions (3 x blocks x 64) and candidate offsets (blocks x 189) are 32-bit
  // on both sides of the ABI: 11.18 M blocks = 715 MPix is the largest image (tested: 268 MPix)
  if ((uint64_t)((w + 7) / 8) * (uint64_t)((h + 7) / 8) * 192u > 0x7fffffffull) { *err = GZ_E_ARG; return nullptr; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev ||
      hipSetDevice(device) != hipSuccess) {
    *err = GZ_E_NO_DEVICE;
    return nullptr;
  }
  gz_ctx* c = new gz_ctx;
  c->device = device;
  c->w = w; c->h = h;
  c->bw = (w + 7) / 8; c->bh = (h + 7) / 8; c->nb = c->bw * c->bh;
  c->pitch = w;
  c->plane = (size_t)c->pitch * h;
  c->target = target;
  set_frame(c, 1);
  auto fail = [&](int code) { *err = code; gz_destroy(c); return (gz_ctx*)nullptr; };
#define CHK0(call) do { if ((call) != hipSuccess) { return fail(GZ_E_HIP); } } while (0)
  // The chain's main stream takes the device's highest priority, so that the dispatcher serves
  // its workgroups before those of the entropy coder that runs beside it (1080p encode 0.144 ->
  // 0.140 s) -- but only for a context that has the device to itself when it is created, and no
  // stream ever goes BELOW the default: with several images in flight priorities invert (an
  // image's low-priority entropy coder starves behind the other images' chains while its host
  // thread waits for it: 16 x 1080p, 8 in flight, 21.8 -> 7.5-13.5 MPix/s with main = highest and
  // entropy = lowest on every context; profiles/r03_stream_priorities.log).
  c->prio_streams = live_contexts(device, +1) == 0;
  c->counted_live = true;
  CHK0(pool_stream_create(&c->own_stream, c->prio_streams ? 1 : 0));
  c->stream = c->own_stream;
  CHK0(pool_stream_create(&c->side_stream));
  CHK0(pool_stream_create(&c->side_stream2));
  CHK0(pool_stream_create(&c->entropy_stream, 0));   // (never below default: see above)
  CHK0(pool_event_create(&c->ev_candidate));
  CHK0(pool_event_create(&c->ev_fork));
  CHK0(pool_event_create(&c->ev_join));
  CHK0(pool_event_create(&c->ev_join2));
  CHK0(pool_event_create(&c->ev_mask_pre));
  CHK0(pool_event_create(&c->ev_next_cand));
  CHK0(pool_event_create(&c->ev_xyb));
  CHK0(pool_event_create(&c->ev_lfy));
  const size_t ncoef = (size_t)3 * c->nb * 64;
  CHK0(pool_malloc((void**)&c->d_rgb, (size_t)3 * w * h));
  CHK0(pool_malloc((void**)&c->d_orig, ncoef * 2));
  CHK0(pool_malloc((void**)&c->d_cand, ncoef * 2));
  CHK0(pool_malloc((void**)&c->d_q, sizeof(int) * 192));
  CHK0(pool_malloc((void**)&c->d_srgb_lut, sizeof(float) * 256));
  CHK0(pool_malloc((void**)&c->d_mask_luts, sizeof(double) * 2048));
  CHK0(pool_malloc((void**)&c->d_block_max, sizeof(float) * c->nb));
  CHK0(pool_malloc((void**)&c->d_max_bits, sizeof(unsigned)));
  CHK0(pool_malloc((void**)&c->d_srgb_out, (size_t)3 * w * h));
  CHK0(pool_malloc((void**)&c->arena, sizeof(float) * c->plane * kNumPlanes));
  for (int i = kNumPlanes - 1; i >= 0; --i) c->free_planes.push_back(c->arena + (size_t)i * c->plane);
  alloc_psycho(c, &c->pi0);
  alloc_psycho(c, &c->pi1);
  for (int i = 0; i < 3; ++i) { c->lin[i] = take_plane(c); }
  for (int i = 0; i < 3; ++i) { c->tmp[i] = take_plane(c); }
  for (int i = 0; i < 3; ++i) { c->xyb[i] = take_plane(c); }
  for (int i = 0; i < 2; ++i) { c->lf_raw[i] = take_plane(c); c->hfp[i] = take_plane(c); }
  c->snb = take_plane(c); c->diffx = take_plane(c); c->diffy = take_plane(c);
  c->mxb = take_plane(c); c->myb1 = take_plane(c); c->myb2 = take_plane(c);
  c->ac[0] = take_plane(c); c->ac[1] = take_plane(c);
  c->dsq = take_plane(c); c->distmap = take_plane(c);
  c->sup0[0] = take_plane(c); c->sup0[1] = take_plane(c);
  // lin planes must be contiguous for k_reconstruct / k_linear_from_rgb8 (plane stride)
  if (c->lin[1] != c->lin[0] + c->plane || c->lin[2] != c->lin[0] + 2 * c->plane) return fail(GZ_E_STATE);

  // tables
  {
    // Srgb8ToLinearTable (gamma_correct.cc:23-38), then float() as LinearRgb /
    // ToLinearRGB store it (butteraugli_comparator.cc:42, output_image.cc:434).
    float lut[256];
    int i = 0;
    for (; i < 11; ++i) lut[i] = (float)(i / 12.92);
    for (; i < 256; ++i) lut[i] = (float)(255.0 * pow(((i / 255.0) + 0.055) / 1.055, 2.4));
    CHK0(hipMemcpy(c->d_srgb_lut, lut, sizeof(lut), hipMemcpyHostToDevice));
    std::vector<double> ml(2048);
    make_mask_lut(2.59885507073, 3.08805636789, 5.62939030582, 0.315424196682, 16.2770141832, &ml[0]);
    make_mask_lut(0.9613705131, -0.581933100068, 6.64307621174, 1.00846207765, 2.2342321176, &ml[512]);
    make_mask_lut(10.0470705878, 3.18472654033, 0.373092999662, 0.0551512255218, 70.0, &ml[1024]);
    make_mask_lut(0.0115640939227, 45.9483175519, 2.52611324247, 0.0142290066313, 5.0, &ml[1536]);
    CHK0(hipMemcpy(c->d_mask_luts, ml.data(), sizeof(double) * 2048, hipMemcpyHostToDevice));
  }
  for (int b = 0; b < B_COUNT; ++b) {
    // Blur(in, float sigma, float border_ratio): both narrowed to float at the call.
    int rc = setup_blur_cfg(c, &c->blur[b], (float)kBlurSpecs[b].sigma, (float)kBlurSpecs[b].border);
    if (rc != GZ_OK) return fail(rc);
    if (c->blur[b].r != kBlurSpecs[b].r) return fail(GZ_E_STATE);
  }
  if (gz_set_rgb(c, rgb) != GZ_OK) return fail(GZ_E_HIP);
#undef CHK0
  return c;
}

int gz_set_rgb(gz_ctx* c, const uint8_t* rgb) {
  DeviceScope ds_(c);
  if (!c || !rgb) return GZ_E_ARG;
  HIPCHK(c, hipMemcpyAsync(c->d_rgb, rgb, (size_t)3 * c->w * c->h, hipMemcpyHostToDevice, c->stream));
  // pi0_ = SeparateFrequencies(OpsinDynamicsImage(LinearRgb(rgb)))
  dim3 grid(gz_div_up(c->w, 256), c->h);
  GZ_LAUNCH(k_linear_from_rgb8, grid, dim3(256), c->stream, c->d_rgb, c->w, c->h, c->pitch,
            c->plane, c->d_srgb_lut, c->lin[0]);
  KCHK(c);
  TRY(stage_opsin(c));
  TRY(stage_separate(c, &c->pi0));
  {  // the original's half of every Compare's DiffPrecompute
    MaskIn in0[2];
    mask_in_psycho(c->pi0, in0);
    TRY(stage_mask_sup(c, in0, c->sup0));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->have_block_mask = false;   // StartBlockComparisons' mask belongs to the old original
  c->have_distmap = false;
  return GZ_OK;
}

void gz_destroy(gz_ctx* c) {
  if (!c) return;
  DeviceScope ds_(c);   // the pools file what comes back under the current device
  // everything must be idle before the memory goes back to the pool (another context may get
  // it at once; hipFree would have waited, the pool does not)
  if (c->stream && c->stream != c->own_stream) (void)hipStreamSynchronize(c->stream);
  if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
  if (c->side_stream) (void)hipStreamSynchronize(c->side_stream);
  if (c->side_stream2) (void)hipStreamSynchronize(c->side_stream2);
  if (c->entropy_stream) (void)hipStreamSynchronize(c->entropy_stream);
  (void)pool_free(c->d_rgb); (void)pool_free(c->d_orig); (void)pool_free(c->d_cand); (void)pool_free(c->d_q);
  (void)pool_free(c->d_srgb_lut); (void)pool_free(c->d_mask_luts); (void)pool_free(c->d_block_max);
  (void)pool_free(c->d_max_bits); (void)pool_free(c->d_srgb_out); (void)pool_free(c->arena);
  (void)pool_free(c->d_blkidx); (void)pool_free(c->d_blkdata);
  (void)pool_free(c->extra_arena);
  (void)pool_free(c->d_block_mask); (void)pool_free(c->d_rank_cnt); (void)pool_free(c->d_rank_tables); (void)pool_free(c->d_rank_idx);
  (void)pool_free(c->d_out_cnt); (void)pool_free(c->d_out_idx); (void)pool_free(c->d_out_err);
  (void)pool_free(c->d_step_delta); (void)pool_free(c->d_csamp); (void)pool_free(c->d_gmax);
  (void)pool_free(c->d_scan_state[0]); (void)pool_free(c->d_scan_state[1]);
  (void)pool_free(c->d_jq); (void)pool_free(c->d_hist); (void)pool_free(c->d_code_depth); (void)pool_free(c->d_code_bits);
  (void)pool_free(c->d_mcu_bits); (void)pool_free(c->d_mcu_off); (void)pool_free(c->d_ff_count);
  (void)pool_free(c->d_words); (void)pool_free(c->d_words_kept);
  (void)pool_free(c->d_order); (void)pool_free(c->d_pos_l); (void)pool_free(c->d_pos_r); (void)pool_free(c->d_chunk);
  (void)pool_free(c->d_part); (void)pool_free(c->d_order_nb); (void)pool_free(c->d_order_off);
  (void)pool_free(c->d_order_groups);
  if (c->h_order_pending) (void)pool_host_free(c->h_order_pending);
  if (c->h_order_mirror) (void)pool_host_free(c->h_order_mirror);
  if (c->h_desc) (void)pool_host_free(c->h_desc);
  if (c->h_scan_result) (void)pool_host_free(c->h_scan_result);
  (void)pool_free(c->d_cmp_stage);
  (void)pool_free(c->d_desc_st); (void)pool_free(c->d_desc_pv);
  (void)pool_free(c->d_order_counters); (void)pool_free(c->d_next_cand); (void)pool_free(c->d_weight);
  (void)pool_free(c->d_max_err); (void)pool_free(c->d_wflag); (void)pool_free(c->d_edit_pos); (void)pool_free(c->d_edit_val);
  for (int b = 0; b < B_COUNT; ++b) (void)pool_free(c->blur[b].d_scale);
  if (c->side_stream) { (void)hipStreamSynchronize(c->side_stream); pool_stream_destroy(c->side_stream); }
  if (c->side_stream2) { (void)hipStreamSynchronize(c->side_stream2); pool_stream_destroy(c->side_stream2); }
  if (c->entropy_stream) { (void)hipStreamSynchronize(c->entropy_stream); pool_stream_destroy(c->entropy_stream, 0); }
  pool_event_destroy(c->ev_candidate);
  stage_free(&c->stage_main);
  stage_free(&c->stage_entropy);
  stage_free(&c->stage_edits);
  if (c->h_res) (void)pool_host_free(c->h_res);
  pool_event_destroy(c->ev_join2);
  pool_event_destroy(c->ev_mask_pre);
  pool_event_destroy(c->ev_next_cand);
  pool_event_destroy(c->ev_xyb);
  pool_event_destroy(c->ev_lfy);
  pool_event_destroy(c->ev_fork);
  pool_event_destroy(c->ev_join);
  pool_stream_destroy(c->own_stream, c->prio_streams ? 1 : 0);   // synchronised at the top of gz_destroy
  if (c->counted_live) (void)live_contexts(c->device, -1);
  delete c;
}

int gz_synchronize(gz_ctx* c) {
  DeviceScope ds_(c);
  if (!c) return GZ_E_ARG;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GZ_OK;
}

int gz_set_stream(gz_ctx* c, void* s) {
  DeviceScope ds_(c);
  if (!c) return GZ_E_ARG;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->stream = s ? (hipStream_t)s : c->own_stream;
  return GZ_OK;
}

int gz_encode_rgb(gz_ctx* c, int16_t* coeffs_out) {
  DeviceScope ds_(c);
  if (!c) return GZ_E_ARG;
  if (c->cfac != 1) {   // back to 4:4:4: the candidate and the search belonged to the other frame
    set_frame(c, 1);
    c->have_cand = false;
  }
  GZ_LAUNCH(k_encode_rgb, dim3(gz_div_up(c->nb, kBlocksPerWG)), dim3(256), c->stream, c->d_rgb,
            c->w, c->h, c->bw, c->nb, c->d_orig);
  KCHK(c);
  c->have_orig = true;
  if (coeffs_out) {
    HIPCHK(c, hipMemcpyAsync(coeffs_out, c->d_orig, (size_t)3 * c->nb * 128,
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return GZ_OK;
}

static int set_orig(gz_ctx* c, const int16_t* coeffs, int factor) {
  if (!c || !coeffs) return GZ_E_ARG;
  if (c->cfac != factor) {   // the candidate and the search belonged to the other frame
    set_frame(c, factor);
    c->have_cand = false;
  }
  HIPCHK(c, hipMemcpyAsync(c->d_orig, coeffs, (size_t)c->nblk * 128, hipMemcpyHostToDevice,
                           c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->have_orig = true;
  return GZ_OK;
}
int gz_set_orig_coeffs(gz_ctx* c, const int16_t* coeffs) {
  DeviceScope ds_(c);
  return set_orig(c, coeffs, 1);
}
int gz_set_orig_coeffs_420(gz_ctx* c, const int16_t* coeffs) {
  DeviceScope ds_(c);
  return set_orig(c, coeffs, 2);
}

int gz_frame_layout(gz_ctx* c, int* chroma_factor, int* luma_blocks, int* chroma_blocks) {
  DeviceScope ds_(c);
  if (!c) return GZ_E_ARG;
  if (chroma_factor) *chroma_factor = c->cfac;
  if (luma_blocks) *luma_blocks = c->nb;
  if (chroma_blocks) *chroma_blocks = c->nbc;
  return GZ_OK;
}

int gz_quantize(gz_ctx* c, const int* q, int16_t* coeffs_out) {
  DeviceScope ds_(c);
  if (!c) return GZ_E_ARG;
  if (!c->have_orig) { c->err = "no original coefficients"; return GZ_E_STATE; }
  int ones[192];
  if (!q) { for (int i = 0; i < 192; ++i) ones[i] = 1; q = ones; }
  for (int i = 0; i < 192; ++i) if (q[i] <= 0) return GZ_E_ARG;
  HIPCHK(c, hipMemcpyAsync(c->d_q, q, sizeof(int) * 192, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));   // q may live on the caller's stack
  const size_t total = (size_t)c->nblk * 64;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
  GZ_LAUNCH(k_quantize, dim3(blocks), dim3(256), c->stream, c->d_orig, c->d_cand, c->coff[1],
            c->coff[2], c->nblk, c->d_q);
  KCHK(c);
  c->have_cand = true;
  if (coeffs_out) {
    HIPCHK(c, hipMemcpyAsync(coeffs_out, c->d_cand, total * 2, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return GZ_OK;
}

int gz_set_coeffs(gz_ctx* c, const int16_t* coeffs) {
  DeviceScope ds_(c);
  if (!c || !coeffs) return GZ_E_ARG;
  HIPCHK(c, hipMemcpyAsync(c->d_cand, coeffs, (size_t)c->nblk * 128, hipMemcpyHostToDevice,
                           c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->have_cand = true;
  return GZ_OK;
}

int gz_set_coeff_blocks(gz_ctx* c, const int32_t* block_index, int n, const int16_t* blocks) {
  DeviceScope ds_(c);
  if (!c || n < 0 || (n > 0 && (!block_index || !blocks))) return GZ_E_ARG;
  if (!c->have_cand) { c->err = "no candidate coefficients"; return GZ_E_STATE; }
  if (c->cfac != 1) { c->err = "gz_set_coeff_blocks needs a 4:4:4 frame"; return GZ_E_STATE; }
  if (n == 0) return GZ_OK;
  for (int i = 0; i < n; ++i)
    if (block_index[i] < 0 || block_index[i] >= c->nb) return GZ_E_ARG;
  if ((size_t)n > c->blkidx_cap) {
    HIPCHK(c, hipStreamSynchronize(c->stream));   // the pool hands memory on without waiting
    (void)pool_free(c->d_blkidx); (void)pool_free(c->d_blkdata);
    c->d_blkidx = nullptr; c->d_blkdata = nullptr;
    c->blkidx_cap = std::max<size_t>((size_t)n, std::min<size_t>((size_t)c->nb, 2 * c->blkidx_cap + 1024));
    HIPCHK(c, pool_malloc((void**)&c->d_blkidx, sizeof(int32_t) * c->blkidx_cap));
    HIPCHK(c, pool_malloc((void**)&c->d_blkdata, c->blkidx_cap * 384));
  }
  HIPCHK(c, hipMemcpyAsync(c->d_blkidx, block_index, sizeof(int32_t) * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_blkdata, blocks, (size_t)n * 384, hipMemcpyHostToDevice, c->stream));
  GZ_LAUNCH(k_scatter_blocks, dim3(gz_div_up(n, kBlocksPerWG)), dim3(256), c->stream,
            (const int32_t*)c->d_blkidx, (const int16_t*)c->d_blkdata, n, c->nb, c->d_cand);
  KCHK(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));   // the caller may reuse its buffers
  return GZ_OK;
}

int gz_get_coeffs(gz_ctx* c, int16_t* out) {
  DeviceScope ds_(c);
  if (!c || !out) return GZ_E_ARG;
  if (!c->have_cand) { c->err = "no candidate coefficients"; return GZ_E_STATE; }
  HIPCHK(c, hipMemcpyAsync(out, c->d_cand, (size_t)c->nblk * 128, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GZ_OK;
}

int gz_reconstruct(gz_ctx* c, uint8_t* srgb, float* linear) {
  DeviceScope ds_(c);
  if (!c) return GZ_E_ARG;
  if (!c->have_cand) { c->err = "no candidate coefficients"; return GZ_E_STATE; }
  TRY(stage_reconstruct(c, c->d_cand, linear ? c->lin[0] : nullptr, srgb ? c->d_srgb_out : nullptr));
  if (srgb) HIPCHK(c, hipMemcpyAsync(srgb, c->d_srgb_out, (size_t)3 * c->w * c->h, hipMemcpyDeviceToHost, c->stream));
  if (linear) for (int i = 0; i < 3; ++i) TRY(download_plane(c, c->lin[i], linear + (size_t)i * c->w * c->h));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GZ_OK;
}

int gz_compare(gz_ctx* c, float* distance, float* distmap, float* block_max) {
  DeviceScope ds_(c);
  if (!c || !distance) return GZ_E_ARG;
  if (!c->have_cand) { c->err = "no candidate coefficients"; return GZ_E_STATE; }
  TRY(enqueue_compare(c, true));
  void* res = nullptr;
  TRY(result_buffer(c, 4, &res));
  HIPCHK(c, hipMemcpyAsync(res, c->d_max_bits, 4, hipMemcpyDeviceToHost, c->stream));
  if (distmap) TRY(download_plane(c, c->distmap, distmap));
  // the per-block maxima stay on the device (phase B's weights are computed there); they
  // come to the host only when asked for, here or by gz_block_weights
  c->h_block_max_valid = false;
  if (block_max) {
    c->h_block_max.resize(c->nb);
    HIPCHK(c, hipMemcpyAsync(c->h_block_max.data(), c->d_block_max, sizeof(float) * c->nb,
                             hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  memcpy(&c->last_distance, res, 4);
  *distance = c->last_distance;
  if (block_max) {
    memcpy(block_max, c->h_block_max.data(), sizeof(float) * c->nb);
    c->h_block_max_valid = true;
  }
  c->have_distmap = true;
  return GZ_OK;
}

int gz_compare_begin(gz_ctx* c) {
  DeviceScope ds_(c);
  if (!c) return GZ_E_ARG;
  if (!c->have_cand) { c->err = "no candidate coefficients"; return GZ_E_STATE; }
  HIPCHK(c, hipEventRecord(c->ev_candidate, c->stream));   // gz_jpeg_scan waits for this only
  TRY(enqueue_compare(c, true));
  c->h_block_max_valid = false;
  c->compare_pending = true;
  c->distance_in_desc = false;
  return GZ_OK;
}

int gz_compare_end(gz_ctx* c, float* distance) {
  DeviceScope ds_(c);
  if (!c || !distance) return GZ_E_ARG;
  if (!c->compare_pending) { c->err = "gz_compare_begin must precede gz_compare_end"; return GZ_E_STATE; }
  if (c->distance_in_desc) {
    // gz_order_build_auto_descend_begin behind this evaluation: the distance comes with its results
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const DescState& p = c->h_desc[kDescMaxLevels + 1];
    if (p.epoch != c->results_epoch || p.depth != 1) { c->err = "the distance did not arrive with the descent"; return GZ_E_STATE; }
    const unsigned bits = (unsigned)p.cut;
    memcpy(&c->last_distance, &bits, 4);
    c->distance_in_desc = false;
  } else {
    void* res = nullptr;
    TRY(result_buffer(c, 4, &res));
    HIPCHK(c, hipMemcpyAsync(res, c->d_max_bits, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    memcpy(&c->last_distance, res, 4);
  }
  *distance = c->last_distance;
  c->have_distmap = true;
  c->compare_pending = false;
  return GZ_OK;
}

int gz_compare_enqueue(gz_ctx* c, int iters) {
  DeviceScope ds_(c);
  if (!c || iters < 0) return GZ_E_ARG;
  if (!c->have_cand) { c->err = "no candidate coefficients"; return GZ_E_STATE; }
  for (int i = 0; i < iters; ++i) TRY(enqueue_compare(c, true));
  return GZ_OK;
}

int gz_last_distance(gz_ctx* c, float* distance) {
  DeviceScope ds_(c);
  if (!c || !distance) return GZ_E_ARG;
  unsigned bits = 0;
  HIPCHK(c, hipMemcpyAsync(&bits, c->d_max_bits, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  memcpy(distance, &bits, 4);
  return GZ_OK;
}

int gz_time_compare(gz_ctx* c, int iters, float* total_ms) {
  DeviceScope ds_(c);
  if (!c || iters <= 0 || !total_ms) return GZ_E_ARG;
  if (!c->have_cand) { c->err = "no candidate coefficients"; return GZ_E_STATE; }
  hipEvent_t e0, e1;
  HIPCHK(c, hipEventCreate(&e0));
  HIPCHK(c, hipEventCreate(&e1));
  HIPCHK(c, hipEventRecord(e0, c->stream));
  for (int i = 0; i < iters; ++i) TRY(enqueue_compare(c, true));
  HIPCHK(c, hipEventRecord(e1, c->stream));
  HIPCHK(c, hipEventSynchronize(e1));
  HIPCHK(c, hipEventElapsedTime(total_ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return GZ_OK;
}

// ComputeBlockErrorAdjustmentWeights, butteraugli_comparator.cc:521-557 (the per-block
// maxima of :505-520 come out of the final blur kernel).  O(nb) host work on nb floats.
int gz_block_weights(gz_ctx* c, int direction, int max_block_dist, double target_mul,
                     int use_distmap, float* block_weight) {
  return gz_block_weights_factor(c, direction, max_block_dist, target_mul, use_distmap, 1, block_weight);
}

int gz_block_weights_factor(gz_ctx* c, int direction, int max_block_dist, double target_mul,
                            int use_distmap, int factor, float* block_weight) {
  DeviceScope ds_(c);
  if (!c || !block_weight || max_block_dist < 0 || (factor != 1 && factor != 2)) return GZ_E_ARG;
  if (use_distmap && !c->have_distmap) { c->err = "no distance map yet"; return GZ_E_STATE; }
  std::vector<float> zero;
  if (use_distmap && !c->h_block_max_valid) {
    c->h_block_max.resize(c->nb);
    HIPCHK(c, hipMemcpyAsync(c->h_block_max.data(), c->d_block_max, sizeof(float) * c->nb,
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->h_block_max_valid = true;
  }
  const float* bmax = c->h_block_max.data();
  if (!use_distmap) { zero.assign(c->nb, 0.0f); bmax = zero.data(); }
  int gw = c->bw, gh = c->bh;
  std::vector<float> grouped;
  if (factor == 2) {   // maxima over 16x16 areas (butteraugli_comparator.cc:502-520)
    gw = (c->w + 15) / 16; gh = (c->h + 15) / 16;
    grouped.assign((size_t)gw * gh, 0.0f);
    for (int by = 0; by < c->bh; ++by)
      for (int bx = 0; bx < c->bw; ++bx) {
        float& m = grouped[(size_t)(by / 2) * gw + bx / 2];
        m = std::max(m, bmax[(size_t)by * c->bw + bx]);
      }
    bmax = grouped.data();
  }
  block_weights_host(bmax, gw, gh, c->target, direction, max_block_dist, target_mul,
                     block_weight);
  return GZ_OK;
}


// ------------------------------------------------- global candidate order (phase B) ----
static int ensure_order_capacity(gz_ctx* c, size_t n) {
  if (!c->d_part) {
    HIPCHK(c, pool_malloc((void**)&c->d_part, sizeof(PartScalars)));
    HIPCHK(c, pool_malloc((void**)&c->d_order_counters, sizeof(unsigned) * 2));
  }
  if (n <= c->order_cap) return GZ_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream));   // the pool hands memory on without waiting
  (void)pool_free(c->d_order); (void)pool_free(c->d_pos_l); (void)pool_free(c->d_pos_r); (void)pool_free(c->d_chunk);
  c->d_order = nullptr; c->d_pos_l = nullptr; c->d_pos_r = nullptr; c->d_chunk = nullptr;
  c->order_cap = 0;
  const size_t cap = n + n / 8 + 4096;
  HIPCHK(c, pool_malloc((void**)&c->d_order, sizeof(OrderEntry) * cap));
  c->chunk_cap = cap / kPartChunk + 2;
  // (gz_order_partition records at most cap / 2 swapped pairs per side; the descent's per-chunk
  // stopper lists need a full chunk's worth per chunk)
  HIPCHK(c, pool_malloc((void**)&c->d_pos_l, sizeof(unsigned) * c->chunk_cap * kPartChunk));
  HIPCHK(c, pool_malloc((void**)&c->d_pos_r, sizeof(unsigned) * c->chunk_cap * kPartChunk));
  HIPCHK(c, pool_malloc((void**)&c->d_chunk, sizeof(unsigned) * 4 * c->chunk_cap));
  c->order_cap = cap;
  return GZ_OK;
}

static int ensure_order_block_arrays(gz_ctx* c) {
  if (c->d_order_nb) return GZ_OK;
  const int nb = c->nb;
  HIPCHK(c, pool_malloc((void**)&c->d_order_nb, sizeof(unsigned) * nb));
  HIPCHK(c, pool_malloc((void**)&c->d_order_off, sizeof(unsigned long long) * (nb + 1)));
  HIPCHK(c, pool_malloc((void**)&c->d_order_groups, sizeof(unsigned) * 2 * gz_div_up(nb, kOrderGroup)));
  HIPCHK(c, pool_malloc((void**)&c->d_next_cand, sizeof(int) * nb));
  HIPCHK(c, pool_malloc((void**)&c->d_weight, sizeof(float) * nb));
  HIPCHK(c, pool_malloc((void**)&c->d_max_err, sizeof(float) * nb));
  HIPCHK(c, pool_malloc((void**)&c->d_wflag, nb));
  HIPCHK(c, hipMemsetAsync(c->d_max_err, 0, sizeof(float) * nb, c->stream));
  return GZ_OK;
}

// d_next_cand / d_weight / d_max_err are in place: sizes, offsets, entries, counters.
static int order_build_enqueue(gz_ctx* c, int direction, int count_below, float limit, bool sizes_done) {
  const int nb = c->sg_n;
  // An order never has more entries than phase A produced candidates: sized once, so that the
  // construction runs through without a host round trip between counting and filling.
  TRY(ensure_order_capacity(c, std::max<size_t>(c->search_total, 1)));
  if ((unsigned long long)c->search_total >= (1ull << 31)) { c->err = "order beyond 2^31 entries"; return GZ_E_STATE; }
  if (!sizes_done) {   // (gz_order_build_auto's weight kernels have done both already)
    HIPCHK(c, hipMemsetAsync(c->d_order_counters, 0, sizeof(unsigned) * 2, c->stream));
    GZ_LAUNCH(k_order_sizes, dim3(gz_div_up(nb, kOrderGroup)), dim3(kOrderGroup), c->stream,
              (const int*)c->d_out_cnt, (const int*)c->d_next_cand, (const float*)c->d_weight,
              direction, nb, c->d_order_nb, c->d_order_groups, (unsigned*)c->d_order_off);
    KCHK(c);
  }
  // (no scan of the counts: k_order_fill's workgroups find their offsets from the group sums)
  GZ_LAUNCH(k_order_fill, dim3(gz_div_up(nb, kFillBlocks)), dim3(256), c->stream,
            (const float*)c->d_out_err, (const int*)c->d_next_cand, (const float*)c->d_weight,
            (const float*)c->d_max_err, (const unsigned*)c->d_order_nb, (const unsigned*)c->d_order_groups,
            (const unsigned*)c->d_order_off /* the blocks' offsets inside their groups: the first 4 nb bytes */,
            direction, nb, count_below ? 1 : 0, limit, c->d_order, c->d_order_off + nb, c->d_order_counters);
  KCHK(c);
  return GZ_OK;
}

static int order_build_device(gz_ctx* c, int direction, int count_below, float limit,
                              uint64_t* total, int32_t* blocks_to_change, uint64_t* below,
                              bool sizes_done = false) {
  const int nb = c->sg_n;
  TRY(order_build_enqueue(c, direction, count_below, limit, sizes_done));
  void* res = nullptr;
  TRY(result_buffer(c, 16, &res));
  HIPCHK(c, hipMemcpyAsync(res, c->d_order_off + nb, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync((char*)res + 8, c->d_order_counters, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  unsigned long long n = 0;
  unsigned counters[2] = {0, 0};
  memcpy(&n, res, 8);
  memcpy(counters, (char*)res + 8, 8);
  if (n > c->order_cap) { c->err = "order larger than the candidate count"; return GZ_E_STATE; }
  c->order_n = (size_t)n;
  *total = n;
  *blocks_to_change = (int32_t)counters[0];
  if (below) *below = counters[1];
  return GZ_OK;
}

int gz_order_build(gz_ctx* c, int direction, const int32_t* next_cand,
                   const float* max_block_error, const float* block_weight, int count_below,
                   float limit, uint64_t* total, int32_t* blocks_to_change, uint64_t* below) {
  DeviceScope ds_(c);
  if (!c || !next_cand || !max_block_error || !block_weight || !total || !blocks_to_change ||
      (direction != 1 && direction != -1) || (count_below && !below))
    return GZ_E_ARG;
  if (!c->have_search) { c->err = "gz_block_zeroing_orders must precede gz_order_build"; return GZ_E_STATE; }
  c->order_pending = false;
  c->results_in_desc = false;
  const int nb = c->sg_n;
  TRY(ensure_order_block_arrays(c));
  HIPCHK(c, hipMemcpyAsync(c->d_next_cand, next_cand, sizeof(int) * nb, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_weight, block_weight, sizeof(float) * nb, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_max_err, max_block_error, sizeof(float) * nb, hipMemcpyHostToDevice, c->stream));
  return order_build_device(c, direction, count_below, limit, total, blocks_to_change, below);
}

int gz_order_reset(gz_ctx* c) {
  DeviceScope ds_(c);
  if (!c) return GZ_E_ARG;
  TRY(ensure_order_block_arrays(c));
  HIPCHK(c, hipMemsetAsync(c->d_max_err, 0, sizeof(float) * c->nb, c->stream));
  return GZ_OK;
}

// The weights and per-block sizes of gz_order_build_auto on the stream (everything up to the
// offsets scan).
static int order_auto_enqueue(gz_ctx* c, int direction, int max_block_dist, double target_mul,
                              int use_distmap, const int32_t* next_cand) {
  if (!c->have_search) { c->err = "gz_block_zeroing_orders must precede gz_order_build_auto"; return GZ_E_STATE; }
  // (a comparison that is still in flight on the same stream delivers the map in time)
  if (use_distmap && !c->have_distmap && !c->compare_pending) { c->err = "no distance map yet"; return GZ_E_STATE; }
  const int nb = c->sg_n;
  TRY(ensure_order_block_arrays(c));
  TRY(ensure_order_capacity(c, std::max<size_t>(c->search_total, 1)));   // also: the counters
  {
    // With a Compare chain in flight on the main stream (gz_order_build_auto_begin) the upload
    // takes side stream 1 -- behind the chain's short SameNoise / radius-5 branch there, long before
    // Malta ends on the main stream -- and the main stream waits for its event: the copy then runs
    // beside the chain instead of between its last kernel and the order's first (15-20 us of the
    // critical path of every phase-B iteration).  NOT the entropy stream: the driver queues the
    // candidate's whole scan there (gz_jpeg_scan_begin) before it asks for the order, and the
    // order, the descent and the distance that arrives with them would wait for the coder
    // (ADVICE r3).  Nothing on the main stream reads d_next_cand before the order's kernels.
    hipStream_t up = c->compare_pending ? c->side_stream : c->stream;
    void* h = nullptr;
    TRY(stage_reserve(c, &c->stage_main, sizeof(int) * nb, &h));
    memcpy(h, next_cand, sizeof(int) * nb);
    // (behind everything the main stream did before the chain -- the bulk steps read the old values)
    if (up != c->stream) HIPCHK(c, hipStreamWaitEvent(up, c->ev_candidate, 0));
    HIPCHK(c, hipMemcpyAsync(c->d_next_cand, h, sizeof(int) * nb, hipMemcpyHostToDevice, up));
    TRY(stage_sent(c, &c->stage_main, up));
    if (up != c->stream) {
      HIPCHK(c, hipEventRecord(c->ev_next_cand, up));
      HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_next_cand, 0));
    }
  }
  const int bw = c->sg_w, bh = c->sg_h;
  const float target = c->target;
  const float* d_bmax = c->d_block_max;
  if (c->sg_factor == 2 && use_distmap) {   // search grid of 16x16 areas: group the 8x8 maxima
    if (!c->d_gmax) HIPCHK(c, pool_malloc((void**)&c->d_gmax, sizeof(float) * ((c->w + 15) / 16) * ((c->h + 15) / 16)));
    GZ_LAUNCH(k_block_max_group, dim3(gz_div_up(nb, 256)), dim3(256), c->stream,
              (const float*)c->d_block_max, c->bw, c->bh, bw, bh, 2, c->d_gmax);
    KCHK(c);
    d_bmax = c->d_gmax;
  }
  GZ_LAUNCH(k_weights_flag, dim3(gz_div_up(nb, 256)), dim3(256), c->stream,
            d_bmax, use_distmap ? 1 : 0, bw, bh, target, target_mul,
            direction, max_block_dist, c->d_wflag, c->d_order_counters);
  KCHK(c);
  GZ_LAUNCH(k_weights_gather, dim3(gz_div_up(nb, kOrderGroup)), dim3(kOrderGroup), c->stream,
            (const unsigned char*)c->d_wflag, bw, bh, direction, max_block_dist, c->d_weight,
            (const int*)c->d_out_cnt, (const int*)c->d_next_cand, c->d_order_nb, c->d_order_groups,
            (unsigned*)c->d_order_off);
  KCHK(c);
  return GZ_OK;
}

int gz_order_build_auto(gz_ctx* c, int direction, int max_block_dist, double target_mul,
                        int use_distmap, const int32_t* next_cand, int count_below, float limit,
                        uint64_t* total, int32_t* blocks_to_change, uint64_t* below) {
  DeviceScope ds_(c);
  if (!c || !next_cand || !total || !blocks_to_change || (direction != 1 && direction != -1) ||
      max_block_dist < 0 || (count_below && !below))
    return GZ_E_ARG;
  c->order_pending = false;
  c->results_in_desc = false;
  c->desc_pending = false;
  TRY(order_auto_enqueue(c, direction, max_block_dist, target_mul, use_distmap, next_cand));
  return order_build_device(c, direction, count_below, limit, total, blocks_to_change, below, true);
}

int gz_order_build_auto_begin(gz_ctx* c, int direction, int max_block_dist, double target_mul,
                              int use_distmap, const int32_t* next_cand, int count_below, float limit) {
  DeviceScope ds_(c);
  if (!c || !next_cand || (direction != 1 && direction != -1) || max_block_dist < 0) return GZ_E_ARG;
  c->order_pending = false;
  c->results_in_desc = false;
  c->desc_pending = false;
  TRY(order_auto_enqueue(c, direction, max_block_dist, target_mul, use_distmap, next_cand));
  TRY(order_build_enqueue(c, direction, count_below, limit, true));
  if (!c->h_order_pending) HIPCHK(c, pool_host_malloc((void**)&c->h_order_pending, sizeof(*c->h_order_pending)));
  HIPCHK(c, hipMemcpyAsync(&c->h_order_pending->total, c->d_order_off + c->sg_n, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->h_order_pending->counters, c->d_order_counters, 8, hipMemcpyDeviceToHost, c->stream));
  c->order_pending = true;
  return GZ_OK;
}

int gz_order_build_auto_end(gz_ctx* c, uint64_t* total, int32_t* blocks_to_change, uint64_t* below) {
  DeviceScope ds_(c);
  if (!c || !total || !blocks_to_change || !below) return GZ_E_ARG;
  if (!c->order_pending) { c->err = "gz_order_build_auto_begin must precede gz_order_build_auto_end"; return GZ_E_STATE; }
  c->order_pending = false;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  gz_ctx::OrderPending r;
  if (c->results_in_desc) {   // (gz_order_build_auto_descend_begin: with the descent's state)
    const DescState& p = c->h_desc[kDescMaxLevels + 1];
    if (p.epoch != c->results_epoch || p.depth != 1) { c->err = "the order's results did not arrive with the descent"; return GZ_E_STATE; }
    r.total = p.lo;
    r.counters[0] = (unsigned)p.hi;
    r.counters[1] = (unsigned)p.last;
    c->results_in_desc = false;
  } else {
    r = *c->h_order_pending;
  }
  if (r.total > c->order_cap) { c->err = "order larger than the candidate count"; return GZ_E_STATE; }
  c->order_n = (size_t)r.total;
  *total = r.total;
  *blocks_to_change = (int32_t)r.counters[0];
  *below = r.counters[1];
  return GZ_OK;
}

static int descend_enqueue(gz_ctx* c, int derive, uint64_t n0, uint64_t last0, float per_block,
                           uint64_t threshold, int max_levels, size_t n_bound, bool publish);

// gz_order_build_auto_begin + gz_order_descend_begin in one call, with ONE transfer of results.
int gz_order_build_auto_descend_begin(gz_ctx* c, int direction, int max_block_dist, double target_mul,
                                      int use_distmap, const int32_t* next_cand, int count_below,
                                      float limit, float per_block, uint64_t threshold, int max_levels) {
  DeviceScope ds_(c);
  if (!c || !next_cand || (direction != 1 && direction != -1) || max_block_dist < 0 || max_levels < 0)
    return GZ_E_ARG;
  c->order_pending = false;
  c->results_in_desc = false;
  c->desc_pending = false;
  c->results_in_desc = false;
  c->distance_in_desc = false;
  TRY(order_auto_enqueue(c, direction, max_block_dist, target_mul, use_distmap, next_cand));
  TRY(order_build_enqueue(c, direction, count_below, limit, true));
  TRY(descend_enqueue(c, 1, 0, 0, per_block, threshold, max_levels, std::max<size_t>(c->search_total, 1), true));
  if (!c->results_in_desc) {   // (no level was launched: the order is too large for the descent's tables)
    if (!c->h_order_pending) HIPCHK(c, pool_host_malloc((void**)&c->h_order_pending, sizeof(*c->h_order_pending)));
    HIPCHK(c, hipMemcpyAsync(&c->h_order_pending->total, c->d_order_off + c->sg_n, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_order_pending->counters, c->d_order_counters, 8, hipMemcpyDeviceToHost, c->stream));
  }
  c->order_pending = true;
  return GZ_OK;
}

int gz_order_advance(gz_ctx* c, float val_threshold, int direction) {
  DeviceScope ds_(c);
  if (!c || (direction != 1 && direction != -1)) return GZ_E_ARG;
  if (!c->d_weight) { c->err = "gz_order_build_auto must precede gz_order_advance"; return GZ_E_STATE; }
  GZ_LAUNCH(k_order_advance, dim3(gz_div_up(c->sg_n, 256)), dim3(256), c->stream, c->d_max_err,
            (const float*)c->d_weight, val_threshold, direction, c->sg_n);
  KCHK(c);
  return GZ_OK;
}

int gz_apply_candidate_steps(gz_ctx* c, int direction, const int32_t* blocks,
                             const int32_t* counts, int n) {
  DeviceScope ds_(c);
  if (!c || n < 0 || (n > 0 && (!blocks || !counts)) || (direction != 1 && direction != -1))
    return GZ_E_ARG;
  if (!c->have_search || !c->d_next_cand || !c->have_cand || !c->have_orig) {
    c->err = "gz_order_build must precede gz_apply_candidate_steps";
    return GZ_E_STATE;
  }
  if (n == 0) return GZ_OK;
  for (int i = 0; i < n; ++i)
    if (blocks[i] < 0 || blocks[i] >= c->sg_n || counts[i] < 0 || counts[i] > 192) return GZ_E_ARG;
  if ((size_t)2 * n > c->edit_cap) {   // the edit buffers double as (blocks, counts) staging
    HIPCHK(c, hipStreamSynchronize(c->stream));   // the pool hands memory on without waiting
    (void)pool_free(c->d_edit_pos); (void)pool_free(c->d_edit_val);
    c->d_edit_pos = nullptr; c->d_edit_val = nullptr;
    c->edit_cap = (size_t)2 * n + (size_t)n + 4096;
    HIPCHK(c, pool_malloc((void**)&c->d_edit_pos, sizeof(int) * c->edit_cap));
    HIPCHK(c, pool_malloc((void**)&c->d_edit_val, sizeof(short) * c->edit_cap));
  }
  int* d_blocks = c->d_edit_pos;
  int* d_counts = c->d_edit_pos + n;
  {
    void* h = nullptr;
    TRY(stage_reserve(c, &c->stage_main, sizeof(int) * 2 * n, &h));
    memcpy(h, blocks, sizeof(int) * n);
    memcpy((int*)h + n, counts, sizeof(int) * n);
    HIPCHK(c, hipMemcpyAsync(d_blocks, h, sizeof(int) * 2 * n, hipMemcpyHostToDevice, c->stream));
    TRY(stage_sent(c, &c->stage_main, c->stream));
  }
  StepGeom sg;
  for (int i = 0; i < 3; ++i) sg.coff[i] = c->coff[i];
  sg.comp_mask = c->sg_mask;
  c->have_step_delta = false;
  if (c->have_jq) {
    // with the symbol statistics' quantiser known, the steps also report what they do to the
    // AC histograms (gz_steps_histogram_delta)
    if (!c->d_step_delta) HIPCHK(c, pool_malloc((void**)&c->d_step_delta, sizeof(unsigned) * 768 * kStepDeltaCopies));
    HIPCHK(c, hipMemsetAsync(c->d_step_delta, 0, sizeof(unsigned) * 768 * kStepDeltaCopies, c->stream));
    // (persistent workgroups: four per CU's worth at most, each wavefront taking several blocks)
    GZ_LAUNCH(k_apply_steps_hist, dim3(std::min(gz_div_up(n, 4), kStepHistGrid)), dim3(256), c->stream, (const int*)d_blocks,
              (const int*)d_counts, n, direction, (const int*)c->d_next_cand,
              (const unsigned char*)c->d_out_idx, (const short*)c->d_orig, (short*)c->d_cand,
              (const int*)c->d_q, (const int*)c->d_jq, sg, c->d_step_delta);
    KCHK(c);
    c->have_step_delta = true;
    return GZ_OK;
  }
  GZ_LAUNCH(k_apply_steps, dim3(gz_div_up(n, 4)), dim3(256), c->stream, (const int*)d_blocks,
            (const int*)d_counts, n, direction, (const int*)c->d_next_cand,
            (const unsigned char*)c->d_out_idx, (const short*)c->d_orig, (short*)c->d_cand,
            (const int*)c->d_q, sg);
  KCHK(c);
  return GZ_OK;   // the caller's buffers were copied to the staging buffer: no wait
}

int gz_steps_histogram_delta(gz_ctx* c, int32_t* ac_delta) {
  DeviceScope ds_(c);
  if (!c || !ac_delta) return GZ_E_ARG;
  if (!c->have_step_delta) {
    c->err = "gz_apply_candidate_steps (after gz_jpeg_histograms) must precede gz_steps_histogram_delta";
    return GZ_E_STATE;
  }
  void* res = nullptr;
  TRY(result_buffer(c, sizeof(unsigned) * 768 * kStepDeltaCopies, &res));
  HIPCHK(c, hipMemcpyAsync(res, c->d_step_delta, sizeof(unsigned) * 768 * kStepDeltaCopies, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // the workgroups' changes went to kStepDeltaCopies copies of the counters (k_apply_steps_hist)
  const unsigned* part = static_cast<const unsigned*>(res);
  for (int k = 0; k < 768; ++k) {
    unsigned sum = 0;
    for (int r = 0; r < kStepDeltaCopies; ++r) sum += part[r * 768 + k];
    ac_delta[k] = (int32_t)sum;
  }
  return GZ_OK;
}

int gz_apply_coeff_edits(gz_ctx* c, const int32_t* pos, const int16_t* val, int n) {
  DeviceScope ds_(c);
  if (!c || n < 0 || (n > 0 && (!pos || !val))) return GZ_E_ARG;
  if (!c->have_cand) { c->err = "no candidate coefficients"; return GZ_E_STATE; }
  if (n == 0) return GZ_OK;
  const int limit = c->nblk * 64;
  for (int i = 0; i < n; ++i)
    if (pos[i] < 0 || pos[i] >= limit) return GZ_E_ARG;
  // A few hundred edits per iteration, between the host's last step and the chain's first kernel:
  // the kernel reads them straight from the page-locked staging buffer (two copies of a few KB on
  // the stream cost more than the bytes' trip over the bus).  Bulk edits go through device memory.
  const bool direct = n <= 4096;
  if (!direct && (size_t)n > c->edit_cap) {
    HIPCHK(c, hipStreamSynchronize(c->stream));   // the pool hands memory on without waiting
    (void)pool_free(c->d_edit_pos); (void)pool_free(c->d_edit_val);
    c->d_edit_pos = nullptr; c->d_edit_val = nullptr;
    c->edit_cap = (size_t)n + (size_t)n / 2 + 4096;
    HIPCHK(c, pool_malloc((void**)&c->d_edit_pos, sizeof(int) * c->edit_cap));
    HIPCHK(c, pool_malloc((void**)&c->d_edit_val, sizeof(short) * c->edit_cap));
  }
  void* h = nullptr;
  TRY(stage_reserve(c, &c->stage_edits, (sizeof(int) + sizeof(short)) * n, &h));
  memcpy(h, pos, sizeof(int) * n);
  memcpy((int*)h + n, val, sizeof(short) * n);
  const int* k_pos = (const int*)h;
  const short* k_val = (const short*)((int*)h + n);
  if (!direct) {
    HIPCHK(c, hipMemcpyAsync(c->d_edit_pos, h, sizeof(int) * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_edit_val, (int*)h + n, sizeof(short) * n, hipMemcpyHostToDevice, c->stream));
    k_pos = c->d_edit_pos;
    k_val = c->d_edit_val;
  }
  GZ_LAUNCH(k_apply_coeff_edits, dim3(gz_div_up(n, 256)), dim3(256), c->stream, k_pos, k_val, n, c->d_cand);
  KCHK(c);
  TRY(stage_sent(c, &c->stage_edits, c->stream));   // (the staging buffer is free again behind the kernel)
  return GZ_OK;   // the caller's buffers were copied to the staging buffer: no wait
}

int gz_order_upload(gz_ctx* c, const void* entries, uint64_t n) {
  DeviceScope ds_(c);
  if (!c || (n > 0 && !entries)) return GZ_E_ARG;
  c->order_pending = false;
  c->results_in_desc = false;
  TRY(ensure_order_capacity(c, (size_t)n));
  if (n > 0)
    HIPCHK(c, hipMemcpyAsync(c->d_order, entries, sizeof(OrderEntry) * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->order_n = (size_t)n;
  return GZ_OK;
}

int gz_order_partition(gz_ctx* c, uint64_t lo, uint64_t hi, uint64_t* cut) {
  DeviceScope ds_(c);
  if (!c || !cut) return GZ_E_ARG;
  if (hi > c->order_n || lo >= hi || hi - lo <= 3 || hi - lo > 0xfffffff0ull) return GZ_E_ARG;
  const size_t first = (size_t)lo + 1;
  const unsigned n = (unsigned)(hi - first);
  const int nchunks = (int)((n + kPartChunk - 1) / kPartChunk);
  unsigned* cnt_l = c->d_chunk;
  unsigned* cnt_r = c->d_chunk + c->chunk_cap;
  unsigned* base_l = c->d_chunk + 2 * c->chunk_cap;
  unsigned* base_r = c->d_chunk + 3 * c->chunk_cap;
  OrderEntry* a = c->d_order;
  PartScalars* ps = c->d_part;
  unsigned* pos_l = c->d_pos_l;
  unsigned* pos_r = c->d_pos_r;
  const size_t lo_s = (size_t)lo, hi_s = (size_t)hi;
  GZ_LAUNCH(k_part_median, dim3(1), dim3(1), c->stream, a, lo_s, hi_s, ps);
  KCHK(c);
  GZ_LAUNCH(k_part_count, dim3(nchunks), dim3(256), c->stream, (const OrderEntry*)a, first, n,
            (const PartScalars*)ps, cnt_l, cnt_r);
  KCHK(c);
  GZ_LAUNCH(k_part_scan, dim3(1), dim3(1024), c->stream, (const unsigned*)cnt_l,
            (const unsigned*)cnt_r, nchunks, base_l, base_r);
  KCHK(c);
  GZ_LAUNCH(k_part_scatter, dim3(nchunks), dim3(256), c->stream, (const OrderEntry*)a, first, n,
            ps, (const unsigned*)base_l, (const unsigned*)base_r, pos_l, pos_r);
  KCHK(c);
  GZ_LAUNCH(k_part_swap, dim3(gz_div_up((int)(n / 2 + 1), 256)), dim3(256), c->stream, a, first,
            (const PartScalars*)ps, (const unsigned*)pos_l, (const unsigned*)pos_r);
  KCHK(c);
  PartScalars h;
  void* res = nullptr;
  TRY(result_buffer(c, sizeof(h), &res));
  HIPCHK(c, hipMemcpyAsync(res, ps, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  memcpy(&h, res, sizeof(h));
  uint64_t r = hi;
  if (h.cut_l != 0xffffffffu) r = std::min<uint64_t>(r, first + h.cut_l);
  if (h.cut_r != 0xffffffffu) r = std::min<uint64_t>(r, first + h.cut_r);
  *cut = r;
  return GZ_OK;
}

int gz_order_host_mirror(gz_ctx* c, uint64_t entries, void** out) {
  DeviceScope ds_(c);
  if (!c || !out) return GZ_E_ARG;
  if (entries > c->order_mirror_cap) {
    HIPCHK(c, hipStreamSynchronize(c->stream));   // (no transfer into the old one is in flight)
    if (c->h_order_mirror) (void)pool_host_free(c->h_order_mirror);
    c->h_order_mirror = nullptr;
    c->order_mirror_cap = 0;
    c->export_epoch = 0;   // (what k_desc_export wrote went with the old array)
    const size_t cap = (size_t)entries + (size_t)entries / 8 + 4096;
    HIPCHK(c, pool_host_malloc(&c->h_order_mirror, sizeof(OrderEntry) * cap));
    c->order_mirror_cap = cap;
  }
  *out = c->h_order_mirror;
  return GZ_OK;
}

int gz_order_fetch(gz_ctx* c, uint64_t lo, uint64_t hi, void* out) {
  DeviceScope ds_(c);
  if (!c || !out || lo > hi || hi > c->order_n) return GZ_E_ARG;
  const size_t bytes = sizeof(OrderEntry) * (size_t)(hi - lo);
  const char* mirror = (const char*)c->h_order_mirror;
  if (bytes > 0 && mirror && (const char*)out >= mirror &&
      (const char*)out + bytes <= mirror + sizeof(OrderEntry) * c->order_mirror_cap) {
    // into the context's pinned mirror: no landing area, no second copy
    HIPCHK(c, hipMemcpyAsync(out, c->d_order + lo, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GZ_OK;
  }
  if (bytes > 0 && bytes <= ((size_t)4 << 20)) {   // the usual case: through the pinned landing area
    void* res = nullptr;
    TRY(result_buffer(c, bytes, &res));
    HIPCHK(c, hipMemcpyAsync(res, c->d_order + lo, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    memcpy(out, res, bytes);
    return GZ_OK;
  }
  if (hi > lo)
    HIPCHK(c, hipMemcpyAsync(out, c->d_order + lo, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GZ_OK;
}


// ---- quick-select descent decided on the device (gz_kernels_order.h: k_desc_count / k_desc_swap)
static int descend_enqueue(gz_ctx* c, int derive, uint64_t n0, uint64_t last0, float per_block,
                           uint64_t threshold, int max_levels, size_t n_bound, bool publish = false) {
  if (!c->d_desc_st) {
    HIPCHK(c, pool_malloc((void**)&c->d_desc_st, sizeof(DescState) * (kDescMaxLevels + 3)));
    HIPCHK(c, pool_malloc((void**)&c->d_desc_pv, sizeof(DescPivot) * kDescMaxLevels));
    HIPCHK(c, pool_host_malloc((void**)&c->h_desc, sizeof(DescState) * (kDescMaxLevels + 3)));
    HIPCHK(c, hipMemsetAsync(c->d_desc_st, 0, sizeof(DescState) * (kDescMaxLevels + 3), c->stream));
    c->desc_epoch = 0;
  }
  if (++c->desc_epoch == 0) c->desc_epoch = 1;
  int levels = std::max(0, std::min(max_levels, kDescMaxLevels));
  const size_t nchunks = std::max<size_t>(1, (n_bound + kPartChunk - 1) / kPartChunk);
  bool big = nchunks > (size_t)kDescMaxChunks;   // (orders beyond 8.4 M entries: the instantiation with the larger tables)
#ifdef GZ_EMU
  if (getenv("GZ_EMU_DESC_BIG")) big = true;
#endif
  if (nchunks > (size_t)kDescMaxChunksBig || nchunks > c->chunk_cap) levels = 0;   // the host drives these
  DescArgs A;
  A.a = c->d_order;
  A.st = c->d_desc_st;
  A.pv = c->d_desc_pv;
  A.cnt_l = c->d_chunk;
  A.cnt_r = c->d_chunk + c->chunk_cap;
  A.lpos = c->d_pos_l;
  A.rpos = c->d_pos_r;
  A.max_chunks = big ? kDescMaxChunksBig : kDescMaxChunks;
  A.epoch = c->desc_epoch;
  A.threshold = threshold < 16 ? 16 : threshold;
  A.derive = derive;
  A.n0 = n0;
  A.last0 = last0;
  A.total = c->d_order_off ? c->d_order_off + c->sg_n : nullptr;
  A.counters = c->d_order_counters;
  A.per_block = per_block;
  A.publish = publish && levels > 0 ? 1 : 0;
  A.max_bits = c->d_max_bits;
  int swap_groups = std::min<int>((int)((n_bound + 1 + kPartChunk - 1) / kPartChunk), kDescSwapGrid);
  int count_groups = (int)std::min<size_t>(nchunks, (size_t)kDescCountGrid);
#ifdef GZ_EMU
  if (const char* e = getenv("GZ_EMU_DESC_SWAP_GRID")) {   // (the loops over groups on orders the emulation can afford)
    swap_groups = std::max(1, atoi(e));
    count_groups = std::max(1, atoi(e));
  }
#endif
  for (int l = 0; l < levels; ++l) {
    GZ_LAUNCH(k_desc_count, dim3((unsigned)count_groups), dim3(256), c->stream, A, l);
    KCHK(c);
    if (big) GZ_LAUNCH(k_desc_swap<kDescMaxChunksBig>, dim3((unsigned)std::min(swap_groups, 256)), dim3(256), c->stream, A, l);
    else GZ_LAUNCH(k_desc_swap<kDescMaxChunks>, dim3((unsigned)swap_groups), dim3(256), c->stream, A, l);
    KCHK(c);
  }
  // gz_order_build_auto_descend_begin: the prefix the driver fetches next goes to its host mirror
  // behind the last level (the driver's own bound on such a fetch: 2^19 entries)
  c->export_epoch = 0;
  if (publish && levels > 0 && c->h_order_mirror) {
    const unsigned long long max_entries = std::min<unsigned long long>(c->order_mirror_cap, 1ull << 19);
    // (its first workgroup also writes the descent's state into c->h_desc: no copy on the stream)
    GZ_LAUNCH(k_desc_export, dim3(128), dim3(256), c->stream, A, levels, (OrderEntry*)c->h_order_mirror, max_entries,
              (DescState*)c->h_desc);
    KCHK(c);
    c->export_epoch = c->desc_epoch;
  } else {
    HIPCHK(c, hipMemcpyAsync(c->h_desc, c->d_desc_st, sizeof(DescState) * (kDescMaxLevels + 3),
                             hipMemcpyDeviceToHost, c->stream));
  }
  c->desc_levels = levels;
  c->desc_pending = true;
  if (A.publish) {
    c->results_in_desc = true;
    c->distance_in_desc = c->compare_pending;
    c->results_epoch = c->desc_epoch;
  }
  return GZ_OK;
}

static int descend_collect(gz_ctx* c, uint64_t* log, int cap_levels, int* levels) {
  int n = 0;
  for (int l = 0; l < c->desc_levels && n < cap_levels; ++l) {
    const DescState& before = c->h_desc[l];
    const DescState& after = c->h_desc[l + 1];
    if (after.epoch != c->desc_epoch || before.epoch != c->desc_epoch) break;
    if (!(after.cut > before.lo && after.cut <= before.hi)) { c->err = "descent: cut outside its range"; return GZ_E_STATE; }
    log[3 * n + 0] = before.lo;
    log[3 * n + 1] = before.hi;
    log[3 * n + 2] = after.cut;
    ++n;
  }
  *levels = n;
  return GZ_OK;
}

int gz_order_descend(gz_ctx* c, uint64_t last, uint64_t threshold, int max_levels, uint64_t* log,
                     int* levels) {
  DeviceScope ds_(c);
  if (!c || !log || !levels || max_levels < 0) return GZ_E_ARG;
  *levels = 0;
  if (c->order_n == 0 || last >= c->order_n) return c->order_n == 0 ? GZ_OK : GZ_E_ARG;
  TRY(descend_enqueue(c, 0, c->order_n, last, 0.0f, threshold, max_levels, c->order_n));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->desc_pending = false;
  return descend_collect(c, log, max_levels, levels);
}

int gz_order_descend_begin(gz_ctx* c, float per_block, uint64_t threshold, int max_levels) {
  DeviceScope ds_(c);
  if (!c || max_levels < 0) return GZ_E_ARG;
  if (!c->order_pending) { c->err = "gz_order_build_auto_begin must precede gz_order_descend_begin"; return GZ_E_STATE; }
  return descend_enqueue(c, 1, 0, 0, per_block, threshold, max_levels, std::max<size_t>(c->search_total, 1));
}

int gz_order_exported(gz_ctx* c, uint64_t* entries) {
  if (!c || !entries) return GZ_E_ARG;
  *entries = 0;
  if (c->desc_pending || c->order_pending) { c->err = "gz_order_descend_end must precede gz_order_exported"; return GZ_E_STATE; }
  if (c->export_epoch == 0 || c->export_epoch != c->desc_epoch || !c->h_desc) return GZ_OK;
  const DescState& p = c->h_desc[kDescMaxLevels + 2];
  if (p.epoch == c->export_epoch && p.depth == 2 && p.lo <= c->order_n) *entries = p.lo;
  return GZ_OK;
}

int gz_order_descend_end(gz_ctx* c, uint64_t* log, int cap_levels, int* levels, uint64_t* last) {
  DeviceScope ds_(c);
  if (!c || !log || !levels || !last || cap_levels < 0) return GZ_E_ARG;
  *levels = 0;
  *last = 0;
  if (!c->desc_pending) return GZ_OK;   // nothing was enqueued (or another build took its place)
  if (c->order_pending) { c->err = "gz_order_build_auto_end must precede gz_order_descend_end"; return GZ_E_STATE; }
  c->desc_pending = false;
  TRY(descend_collect(c, log, cap_levels, levels));
  if (*levels > 0) *last = c->h_desc[0].last;
#ifdef GZ_EMU
  // test hook of the emulation build only (tests/test_host_encoder.py): a device that derived
  // another position than the host -- the driver's guard must refuse the rearranged order
  if (*levels > 0 && getenv("GZ_EMU_SKEW_DESCENT")) *last += 10;
#endif
  return GZ_OK;
}

// ------------------------------------------------------------- device entropy coder ----
static int ensure_entropy_buffers(gz_ctx* c) {
  if (c->d_jq) return GZ_OK;
  HIPCHK(c, pool_malloc((void**)&c->d_jq, sizeof(int) * 192));
  HIPCHK(c, pool_malloc((void**)&c->d_hist, sizeof(unsigned) * 1536));
  HIPCHK(c, pool_malloc((void**)&c->d_code_depth, 1536));
  HIPCHK(c, pool_malloc((void**)&c->d_code_bits, sizeof(unsigned short) * 1536));
  HIPCHK(c, pool_malloc((void**)&c->d_mcu_bits, sizeof(unsigned) * c->nb));
  HIPCHK(c, pool_malloc((void**)&c->d_mcu_off, sizeof(unsigned long long) * (c->nb + 1)));
  HIPCHK(c, pool_malloc((void**)&c->d_ff_count, sizeof(unsigned long long)));
  return GZ_OK;
}

// The frame as the JPEG sees it: ncomp == 3: the current layout with its MCUs; ncomp == 1: the
// luma component alone, one block per MCU, no padding (SaveToJpegData writes a single
// component when both chroma components are entirely zero, output_image.cc:357-365).
static FrameGeom frame_geom(const gz_ctx* c, int ncomp) {
  FrameGeom g;
  g.ncomp = ncomp;
  for (int i = 0; i < 3; ++i) {
    g.bw[i] = i == 0 ? c->bw : c->cbw;
    g.bh[i] = i == 0 ? c->bh : c->cbh;
    g.coff[i] = c->coff[i];
    g.samp[i] = (i == 0 && ncomp == 3) ? c->cfac : 1;
  }
  g.mcu_cols = ncomp == 3 ? c->cbw : c->bw;
  g.mcu_rows = ncomp == 3 ? c->cbh : c->bh;
  return g;
}

int gz_jpeg_histograms(gz_ctx* c, const int* q, uint32_t* counts) {
  return gz_jpeg_histograms_ncomp(c, q, 3, counts);
}

int gz_jpeg_histograms_ncomp(gz_ctx* c, const int* q, int ncomp, uint32_t* counts) {
  DeviceScope ds_(c);
  if (!c || !q || !counts || (ncomp != 1 && ncomp != 3)) return GZ_E_ARG;
  if (!c->have_cand) { c->err = "no candidate coefficients"; return GZ_E_STATE; }
  for (int i = 0; i < 192; ++i) if (q[i] <= 0) return GZ_E_ARG;
  TRY(ensure_entropy_buffers(c));
  if (!c->have_jq || memcmp(c->h_jq, q, sizeof(c->h_jq)) != 0) {
    memcpy(c->h_jq, q, sizeof(c->h_jq));
    HIPCHK(c, hipMemcpyAsync(c->d_jq, c->h_jq, sizeof(int) * 192, hipMemcpyHostToDevice, c->stream));
  }
  HIPCHK(c, hipMemsetAsync(c->d_hist, 0, sizeof(unsigned) * 1536, c->stream));
  const FrameGeom geom = frame_geom(c, ncomp);
  const int grid = std::min(gz_div_up(geom.mcu_cols * geom.mcu_rows, kHistWaves), 1024);
  GZ_LAUNCH(k_jpeg_histograms, dim3(grid), dim3(64 * kHistWaves), c->stream, (const int16_t*)c->d_cand,
            (const int*)c->d_jq, geom, c->d_hist);
  KCHK(c);
  void* res = nullptr;
  TRY(result_buffer(c, sizeof(unsigned) * 1536, &res));
  HIPCHK(c, hipMemcpyAsync(res, c->d_hist, sizeof(unsigned) * 1536, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  memcpy(counts, res, sizeof(unsigned) * 1536);
  c->have_jq = true;
  return GZ_OK;
}

int gz_jpeg_scan_begin(gz_ctx* c, int ncomp, const uint8_t* depth, const uint16_t* code) {
  DeviceScope ds_(c);
  if (!c || !depth || !code || (ncomp != 1 && ncomp != 3)) return GZ_E_ARG;
  c->scan_pending = false;
  if (!c->have_cand || !c->have_jq) { c->err = "gz_jpeg_histograms must precede gz_jpeg_scan"; return GZ_E_STATE; }
  // Upper bound of a scan: per coefficient a code of at most 16 bits and at most 16 extra
  // bits (int16 magnitudes), plus an end-of-block per block, plus the final padding.  Sized
  // once, so that no host round trip is needed between counting the bits and writing them.
  const size_t cap_words = (size_t)c->nb * 3 * (64 + 1) + 8;
  if (cap_words > c->words_cap) {
    HIPCHK(c, hipStreamSynchronize(c->entropy_stream));   // the pool hands memory on without waiting
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)pool_free(c->d_words);
    c->d_words = nullptr;
    c->words_cap = 0;
    HIPCHK(c, pool_malloc((void**)&c->d_words, sizeof(unsigned) * cap_words));
    c->words_cap = cap_words;
  }
  // own stream, behind the candidate (not behind a Compare that gz_compare_begin enqueued)
  hipStream_t es = c->entropy_stream;
  if (!c->compare_pending) HIPCHK(c, hipEventRecord(c->ev_candidate, c->stream));
  HIPCHK(c, hipStreamWaitEvent(es, c->ev_candidate, 0));
  {
    void* h = nullptr;
    TRY(stage_reserve(c, &c->stage_entropy, 1536 + sizeof(unsigned short) * 1536, &h));
    memcpy(h, depth, 1536);
    memcpy((uint8_t*)h + 1536, code, sizeof(unsigned short) * 1536);
    HIPCHK(c, hipMemcpyAsync(c->d_code_depth, h, 1536, hipMemcpyHostToDevice, es));
    HIPCHK(c, hipMemcpyAsync(c->d_code_bits, (uint8_t*)h + 1536, sizeof(unsigned short) * 1536, hipMemcpyHostToDevice, es));
    TRY(stage_sent(c, &c->stage_entropy, es));
  }
  JpegCodes codes{c->d_code_depth, c->d_code_bits};
  const FrameGeom geom = frame_geom(c, ncomp);
  const int nmcu = geom.mcu_cols * geom.mcu_rows;
  const int upm = ncomp == 1 ? 1 : (c->cfac == 2 ? 6 : 3);   // blocks per MCU
  const dim3 egrid(gz_div_up(nmcu, kMcuWaves * kMcuPerWave)), eblock(64 * kMcuWaves);
  if (upm == 3)
    GZ_LAUNCH((k_jpeg_block_bits<3, kMcuWaves>), egrid, eblock, es, (const int16_t*)c->d_cand,
              (const int*)c->d_jq, geom, codes, c->d_mcu_bits);
  else if (upm == 6)
    GZ_LAUNCH((k_jpeg_block_bits<6, kMcuWaves>), egrid, eblock, es, (const int16_t*)c->d_cand,
              (const int*)c->d_jq, geom, codes, c->d_mcu_bits);
  else
    GZ_LAUNCH((k_jpeg_block_bits<1, kMcuWaves>), egrid, eblock, es, (const int16_t*)c->d_cand,
              (const int*)c->d_jq, geom, codes, c->d_mcu_bits);
  KCHK(c);
  TRY(enqueue_scan_offsets(c, 1, es, (const unsigned*)c->d_mcu_bits, nmcu, c->d_mcu_off));
  const unsigned long long* d_total = c->d_mcu_off + nmcu;
  const int cgrid = (int)std::min<size_t>(512, (cap_words + 255) / 256);
  GZ_LAUNCH(k_jpeg_clear_words, dim3(cgrid), dim3(256), es, c->d_words, d_total,
            (unsigned long long)c->words_cap, c->d_ff_count);
  KCHK(c);
  if (upm == 3)
    GZ_LAUNCH((k_jpeg_emit<3, kMcuWaves>), egrid, eblock, es, (const int16_t*)c->d_cand, (const int*)c->d_jq,
              geom, codes, (const unsigned long long*)c->d_mcu_off, c->d_words, (unsigned long long)c->words_cap);
  else if (upm == 6)
    GZ_LAUNCH((k_jpeg_emit<6, kMcuWaves>), egrid, eblock, es, (const int16_t*)c->d_cand, (const int*)c->d_jq,
              geom, codes, (const unsigned long long*)c->d_mcu_off, c->d_words, (unsigned long long)c->words_cap);
  else
    GZ_LAUNCH((k_jpeg_emit<1, kMcuWaves>), egrid, eblock, es, (const int16_t*)c->d_cand, (const int*)c->d_jq,
              geom, codes, (const unsigned long long*)c->d_mcu_off, c->d_words, (unsigned long long)c->words_cap);
  KCHK(c);
  GZ_LAUNCH(k_jpeg_count_ff, dim3(cgrid), dim3(256), es, (const unsigned*)c->d_words, d_total,
            c->d_ff_count);
  KCHK(c);
  // (a buffer of its own: the calls allowed between the two halves use result_buffer)
  if (!c->h_scan_result) HIPCHK(c, pool_host_malloc(&c->h_scan_result, 16));
  HIPCHK(c, hipMemcpyAsync(c->h_scan_result, d_total, 8, hipMemcpyDeviceToHost, es));
  HIPCHK(c, hipMemcpyAsync((char*)c->h_scan_result + 8, c->d_ff_count, 8, hipMemcpyDeviceToHost, es));
  c->have_scan = false;
  c->scan_pending = true;
  return GZ_OK;
}

int gz_jpeg_scan_end(gz_ctx* c, uint64_t* scan_bytes) {
  DeviceScope ds_(c);
  if (!c || !scan_bytes) return GZ_E_ARG;
  if (!c->scan_pending) { c->err = "gz_jpeg_scan_begin must precede gz_jpeg_scan_end"; return GZ_E_STATE; }
  c->scan_pending = false;
  unsigned long long total_bits = 0, ff = 0;
  HIPCHK(c, hipStreamSynchronize(c->entropy_stream));
  memcpy(&total_bits, c->h_scan_result, 8);
  memcpy(&ff, (char*)c->h_scan_result + 8, 8);
  const unsigned long long nbytes = (total_bits + 7) / 8;
  if (nbytes / 4 + 4 > c->words_cap) { c->err = "scan larger than its bound (code lengths above 16?)"; return GZ_E_ARG; }
  c->scan_bits = total_bits;
  c->scan_ff = ff;
  c->have_scan = true;
  *scan_bytes = nbytes + ff;
  return GZ_OK;
}

int gz_jpeg_scan(gz_ctx* c, int ncomp, const uint8_t* depth, const uint16_t* code,
                 uint64_t* scan_bytes) {
  if (!scan_bytes) return GZ_E_ARG;
  TRY(gz_jpeg_scan_begin(c, ncomp, depth, code));
  return gz_jpeg_scan_end(c, scan_bytes);
}

int gz_jpeg_scan_bits(gz_ctx* c, uint64_t* bits, uint64_t* stuffed) {
  DeviceScope ds_(c);
  if (!c || !bits || !stuffed) return GZ_E_ARG;
  if (!c->have_scan) { c->err = "no scan yet"; return GZ_E_STATE; }
  *bits = c->scan_bits;
  *stuffed = c->scan_ff;
  return GZ_OK;
}

int gz_jpeg_scan_keep(gz_ctx* c) {
  DeviceScope ds_(c);
  if (!c) return GZ_E_ARG;
  if (!c->have_scan) { c->err = "no scan to keep"; return GZ_E_STATE; }
  const size_t need_words = (size_t)((c->scan_bits + 7) / 8 / 4 + 4);
  if (need_words > c->words_kept_cap) {
    HIPCHK(c, hipStreamSynchronize(c->stream));   // the pool hands memory on without waiting
    (void)pool_free(c->d_words_kept);
    c->d_words_kept = nullptr;
    c->words_kept_cap = need_words + need_words / 4 + 1024;
    HIPCHK(c, pool_malloc((void**)&c->d_words_kept, sizeof(unsigned) * c->words_kept_cap));
  }
  HIPCHK(c, hipMemcpyAsync(c->d_words_kept, c->d_words, sizeof(unsigned) * need_words,
                           hipMemcpyDeviceToDevice, c->stream));
  c->kept_bits = c->scan_bits;
  c->kept_ff = c->scan_ff;
  c->have_kept = true;
  return GZ_OK;
}

int gz_jpeg_scan_bytes(gz_ctx* c, int kept, uint8_t* out, size_t cap, size_t* n) {
  DeviceScope ds_(c);
  if (!c || !out || !n) return GZ_E_ARG;
  if (kept ? !c->have_kept : !c->have_scan) { c->err = "no scan"; return GZ_E_STATE; }
  const unsigned long long bits = kept ? c->kept_bits : c->scan_bits;
  const unsigned long long ff = kept ? c->kept_ff : c->scan_ff;
  const size_t nbytes = (size_t)((bits + 7) / 8);
  *n = nbytes + (size_t)ff;
  if (*n > cap) return GZ_E_ARG;
  std::vector<unsigned> w(nbytes / 4 + 1);
  HIPCHK(c, hipMemcpyAsync(w.data(), kept ? c->d_words_kept : c->d_words, sizeof(unsigned) * w.size(),
                           hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // byte stuffing (BitWriter::EmitByte, jpeg_bit_writer.h:66-72): 0x00 after every 0xFF
  size_t o = 0;
  for (size_t j = 0; j < nbytes; ++j) {
    const uint8_t b = (uint8_t)(w[j >> 2] >> (24 - 8 * (j & 3)));
    out[o++] = b;
    if (b == 0xff) out[o++] = 0;
  }
  if (o != *n) { c->err = "stuffed size mismatch"; return GZ_E_STATE; }
  return GZ_OK;
}

// ------------------------------------------------------------------- stage probes -----
int gz_probe_blur(gz_ctx* c, const float* in, float sigma, float border_ratio, float* out) {
  DeviceScope ds_(c);
  if (!c || !in || !out) return GZ_E_ARG;
  BlurCfg cfg;
  TRY(setup_blur_cfg(c, &cfg, sigma, border_ratio));
  float* src = c->xyb[0];
  TRY(upload_planes(c, in, &src, 1));
  SrcPack<SrcPlain, 1> s;
  s.s[0].p = src;
  PostStore<1> post; post.out[0] = c->xyb[1];
  int rc = GZ_OK;
  // the same kernels gz_compare uses for each radius: fused below 16, two passes from 16 up
  PlanePack<1> t; CPlanePack<1> ct;
  t.p[0] = c->tmp[0]; ct.p[0] = c->tmp[0];
#define GZ_BLUR_CASE(R)                                                     \
  case R:                                                                   \
    rc = blur2d<R, 1, SrcPlain, PostStore<1>>(c, s, post, cfg);             \
    break;
#define GZ_BLUR_CASE2(R)                                                    \
  case R:                                                                   \
    rc = blur_h<R, SrcPlain, 1>(c, s, t, cfg);                              \
    if (rc == GZ_OK) rc = blur_v<R, 1, PostStore<1>>(c, ct, post, cfg);     \
    break;
  switch (cfg.r) {
    GZ_BLUR_CASE(2) GZ_BLUR_CASE(3) GZ_BLUR_CASE(4) GZ_BLUR_CASE(5) GZ_BLUR_CASE(8)
    GZ_BLUR_CASE2(16) GZ_BLUR_CASE2(20) GZ_BLUR_CASE2(23)
    default: c->err = "unsupported blur radius"; rc = GZ_E_ARG;
  }
#undef GZ_BLUR_CASE
#undef GZ_BLUR_CASE2
  if (rc == GZ_OK) rc = download_plane(c, c->xyb[1], out);
  (void)hipStreamSynchronize(c->stream);
  (void)pool_free(cfg.d_scale);
  return rc;
}

int gz_probe_opsin(gz_ctx* c, const float* rgb3, float* xyb3) {
  DeviceScope ds_(c);
  if (!c || !rgb3 || !xyb3) return GZ_E_ARG;
  TRY(upload_planes(c, rgb3, c->lin, 3));
  TRY(stage_opsin(c));
  for (int i = 0; i < 3; ++i) TRY(download_plane(c, c->xyb[i], xyb3 + (size_t)i * c->w * c->h));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GZ_OK;
}

int gz_probe_separate_frequencies(gz_ctx* c, const float* xyb3, float* out10) {
  DeviceScope ds_(c);
  if (!c || !xyb3 || !out10) return GZ_E_ARG;
  TRY(ensure_pip(c));
  TRY(upload_planes(c, xyb3, c->xyb, 3));
  TRY(stage_separate(c, &c->pip));
  const size_t n = (size_t)c->w * c->h;
  for (int i = 0; i < 3; ++i) TRY(download_plane(c, c->pip.lfv[i], out10 + i * n));
  for (int i = 0; i < 2; ++i) TRY(download_plane(c, c->pip.mf[i], out10 + (3 + i) * n));
  memset(out10 + 5 * n, 0, sizeof(float) * n);   // mf[2]: dead in the reference, not computed
  for (int i = 0; i < 2; ++i) TRY(download_plane(c, c->pip.hf[i], out10 + (6 + i) * n));
  for (int i = 0; i < 2; ++i) TRY(download_plane(c, c->pip.uhf[i], out10 + (8 + i) * n));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GZ_OK;
}

int gz_probe_diffmap(gz_ctx* c, const float* rgb0, const float* rgb1, float* diffmap,
                     float* score) {
  DeviceScope ds_(c);
  if (!c || !rgb0 || !rgb1) return GZ_E_ARG;
  TRY(ensure_pip(c));
  TRY(upload_planes(c, rgb0, c->lin, 3));
  TRY(stage_opsin(c));
  TRY(stage_separate(c, &c->pip));
  TRY(upload_planes(c, rgb1, c->lin, 3));
  TRY(stage_opsin(c));
  TRY(stage_separate(c, &c->pi1));
  TRY(stage_diffmap(c, c->pip, c->pi1, false));
  if (diffmap) TRY(download_plane(c, c->distmap, diffmap));
  unsigned bits = 0;
  HIPCHK(c, hipMemcpyAsync(&bits, c->d_max_bits, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (score) memcpy(score, &bits, 4);
  return GZ_OK;
}

int gz_probe_mask(gz_ctx* c, const float* xyb0, const float* xyb1, float* mask3,
                  float* mask_dc3) {
  DeviceScope ds_(c);
  if (!c || !xyb0 || !xyb1 || !mask3) return GZ_E_ARG;
  TRY(ensure_pip(c));
  // Mask(xyb0, xyb1) reads planes 0 and 1 of each image unchanged (butteraugli.cc:1765,1777)
  float* a[2] = {c->pip.hf[0], c->pip.hf[1]};
  float* b[2] = {c->pi1.hf[0], c->pi1.hf[1]};
  TRY(upload_planes(c, xyb0, a, 2));
  TRY(upload_planes(c, xyb1, b, 2));
  const float* const ca2[2] = {a[0], a[1]};
  const float* const cb2[2] = {b[0], b[1]};
  MaskPrePack pk;
  TRY(mask_pack_plain(c, ca2, cb2, &pk));
  TRY(stage_mask_blurs(c, pk));
  CombineArgs ca;
  memset(&ca, 0, sizeof(ca));
  ca.mask_x_blur = c->mxb; ca.mask_y_blur1 = c->myb1; ca.mask_y_blur2 = c->myb2;
  ca.luts = c->d_mask_luts;
  ca.out = nullptr;
  for (int i = 0; i < 3; ++i) { ca.mask_out[i] = c->mask_out[i]; ca.mask_dc_out[i] = c->mask_dc_out[i]; }
  dim3 grid(gz_div_up(c->w, 1024), c->h);   // (4 pixels per thread)
  GZ_LAUNCH(k_combine, grid, dim3(256), c->stream, ca, c->w, c->h, c->pitch);
  KCHK(c);
  const size_t n = (size_t)c->w * c->h;
  for (int i = 0; i < 3; ++i) {
    TRY(download_plane(c, c->mask_out[i], mask3 + i * n));
    if (mask_dc3) TRY(download_plane(c, c->mask_dc_out[i], mask_dc3 + i * n));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GZ_OK;
}

static int probe_device(int device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev ||
      hipSetDevice(device) != hipSuccess)
    return GZ_E_NO_DEVICE;
  return GZ_OK;
}

int gz_probe_idct_blocks(int device, const int16_t* blocks, int n, uint8_t* out) {
  if (!blocks || !out || n <= 0) return GZ_E_ARG;
  if (probe_device(device) != GZ_OK) return GZ_E_NO_DEVICE;
  int16_t* d_in = nullptr; uint8_t* d_out = nullptr;
  if (hipMalloc((void**)&d_in, (size_t)n * 128) != hipSuccess) return GZ_E_HIP;
  if (hipMalloc((void**)&d_out, (size_t)n * 64) != hipSuccess) { (void)hipFree(d_in); return GZ_E_HIP; }
  if (hipMemcpy(d_in, blocks, (size_t)n * 128, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d_in); (void)hipFree(d_out); return GZ_E_HIP; }
  GZ_LAUNCH(k_idct_blocks, dim3(gz_div_up(n, kBlocksPerWG)), dim3(256), (hipStream_t)0, d_in, n, d_out);
  int rc = hipGetLastError() == hipSuccess ? GZ_OK : GZ_E_HIP;
  if (hipMemcpy(out, d_out, (size_t)n * 64, hipMemcpyDeviceToHost) != hipSuccess) rc = GZ_E_HIP;
  (void)hipFree(d_in); (void)hipFree(d_out);
  return rc;
}

int gz_probe_fdct_blocks(int device, int16_t* blocks, int n) {
  if (!blocks || n <= 0) return GZ_E_ARG;
  if (probe_device(device) != GZ_OK) return GZ_E_NO_DEVICE;
  int16_t* d = nullptr;
  if (hipMalloc((void**)&d, (size_t)n * 128) != hipSuccess) return GZ_E_HIP;
  if (hipMemcpy(d, blocks, (size_t)n * 128, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return GZ_E_HIP; }
  GZ_LAUNCH(k_fdct_blocks, dim3(gz_div_up(n, kBlocksPerWG)), dim3(256), (hipStream_t)0, d, n);
  int rc = hipGetLastError() == hipSuccess ? GZ_OK : GZ_E_HIP;
  if (hipMemcpy(blocks, d, (size_t)n * 128, hipMemcpyDeviceToHost) != hipSuccess) rc = GZ_E_HIP;
  (void)hipFree(d);
  return rc;
}

// ----------------------------------------------------------- double-precision DCT ----
namespace {
struct DevBuf {   // scoped device allocation for the context-free entry points
  void* p = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) (void)hipFree(p); }
  bool alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1) == hipSuccess; }
};
}  // namespace

int gz_encode_rgb_only(int device, const uint8_t* rgb, int w, int h, int16_t* coeffs_out) {
  if (!rgb || !coeffs_out || w <= 0 || h <= 0 || w >= (1 << 16) || h >= (1 << 16)) return GZ_E_ARG;
  if (probe_device(device) != GZ_OK) return GZ_E_NO_DEVICE;
  const int bw = (w + 7) / 8, bh = (h + 7) / 8, nb = bw * bh;
  DevBuf drgb, dco;
  if (!drgb.alloc((size_t)3 * w * h) || !dco.alloc((size_t)3 * nb * 128)) return GZ_E_NOMEM;
  if (hipMemcpy(drgb.p, rgb, (size_t)3 * w * h, hipMemcpyHostToDevice) != hipSuccess) return GZ_E_HIP;
  const uint8_t* d_rgb = (const uint8_t*)drgb.p;
  int16_t* d_co = (int16_t*)dco.p;
  GZ_LAUNCH(k_encode_rgb, dim3(gz_div_up(nb, kBlocksPerWG)), dim3(256), (hipStream_t)0, d_rgb, w, h,
            bw, nb, d_co);
  if (hipGetLastError() != hipSuccess) return GZ_E_HIP;
  if (hipMemcpy(coeffs_out, dco.p, (size_t)3 * nb * 128, hipMemcpyDeviceToHost) != hipSuccess)
    return GZ_E_HIP;
  return GZ_OK;
}

int gz_dct_double_blocks(int device, double* blocks, int n, int inverse) {
  if (!blocks || n <= 0) return GZ_E_ARG;
  if (probe_device(device) != GZ_OK) return GZ_E_NO_DEVICE;
  DevBuf d;
  const size_t bytes = (size_t)n * 64 * sizeof(double);
  if (!d.alloc(bytes)) return GZ_E_NOMEM;
  if (hipMemcpy(d.p, blocks, bytes, hipMemcpyHostToDevice) != hipSuccess) return GZ_E_HIP;
  double* dblk = (double*)d.p;
  if (inverse) {
    GZ_LAUNCH((k_dctd_blocks<true>), dim3(gz_div_up(n, kBlocksPerWG)), dim3(256), (hipStream_t)0,
              dblk, n);
  } else {
    GZ_LAUNCH((k_dctd_blocks<false>), dim3(gz_div_up(n, kBlocksPerWG)), dim3(256), (hipStream_t)0,
              dblk, n);
  }
  if (hipGetLastError() != hipSuccess) return GZ_E_HIP;
  if (hipMemcpy(blocks, d.p, bytes, hipMemcpyDeviceToHost) != hipSuccess) return GZ_E_HIP;
  return GZ_OK;
}

int gz_component_to_float_pixels(int device, const int16_t* coeffs, int w, int h, float* out) {
  if (!coeffs || !out || w <= 0 || h <= 0 || w >= (1 << 16) || h >= (1 << 16)) return GZ_E_ARG;
  if (probe_device(device) != GZ_OK) return GZ_E_NO_DEVICE;
  const int bw = (w + 7) / 8, bh = (h + 7) / 8, nb = bw * bh;
  DevBuf dc, dp;
  if (!dc.alloc((size_t)nb * 128) || !dp.alloc((size_t)w * h * sizeof(float))) return GZ_E_NOMEM;
  if (hipMemcpy(dc.p, coeffs, (size_t)nb * 128, hipMemcpyHostToDevice) != hipSuccess) return GZ_E_HIP;
  const int16_t* dcoef = (const int16_t*)dc.p;
  float* dpix = (float*)dp.p;
  GZ_LAUNCH(k_to_float_pixels, dim3(gz_div_up(nb, kBlocksPerWG)), dim3(256), (hipStream_t)0,
            dcoef, w, h, bw, nb, dpix);
  if (hipGetLastError() != hipSuccess) return GZ_E_HIP;
  if (hipMemcpy(out, dp.p, (size_t)w * h * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
    return GZ_E_HIP;
  return GZ_OK;
}

int gz_component_set_downsampled(int device, const float* pixels, int w, int h, int fx, int fy,
                                 int16_t* coeffs_out) {
  if (!pixels || !coeffs_out || w <= 0 || h <= 0 || w >= (1 << 16) || h >= (1 << 16) ||
      fx < 1 || fy < 1 || fx > 4 || fy > 4)
    return GZ_E_ARG;
  if (probe_device(device) != GZ_OK) return GZ_E_NO_DEVICE;
  const int bw = (w + 8 * fx - 1) / (8 * fx), bh = (h + 8 * fy - 1) / (8 * fy), nb = bw * bh;
  DevBuf dc, dp;
  if (!dc.alloc((size_t)nb * 128) || !dp.alloc((size_t)w * h * sizeof(float))) return GZ_E_NOMEM;
  if (hipMemcpy(dp.p, pixels, (size_t)w * h * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
    return GZ_E_HIP;
  const float* dpix = (const float*)dp.p;
  int16_t* dcoef = (int16_t*)dc.p;
  GZ_LAUNCH(k_set_downsampled_coeffs, dim3(gz_div_up(nb, kBlocksPerWG)), dim3(256), (hipStream_t)0,
            dpix, w, h, fx, fy, bw, nb, dcoef);
  if (hipGetLastError() != hipSuccess) return GZ_E_HIP;
  if (hipMemcpy(coeffs_out, dc.p, (size_t)nb * 128, hipMemcpyDeviceToHost) != hipSuccess)
    return GZ_E_HIP;
  return GZ_OK;
}

int gz_probe_rank_sort(int device, const float* keys, const int32_t* cnt, int narr, uint8_t* perm) {
  if (!keys || !cnt || !perm || narr <= 0) return GZ_E_ARG;
  for (int i = 0; i < narr; ++i) if (cnt[i] < 0 || cnt[i] > 192) return GZ_E_ARG;
  if (probe_device(device) != GZ_OK) return GZ_E_NO_DEVICE;
  DevBuf dk, dc, dp;
  if (!dk.alloc(sizeof(float) * narr * 192) || !dc.alloc(sizeof(int32_t) * narr) || !dp.alloc((size_t)narr * 192))
    return GZ_E_NOMEM;
  if (hipMemcpy(dk.p, keys, sizeof(float) * narr * 192, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(dc.p, cnt, sizeof(int32_t) * narr, hipMemcpyHostToDevice) != hipSuccess)
    return GZ_E_HIP;
  const float* pk = (const float*)dk.p; const int32_t* pc = (const int32_t*)dc.p; uint8_t* pp = (uint8_t*)dp.p;
  GZ_LAUNCH(k_probe_rank_sort, dim3(gz_div_up(narr, kRankLanes)), dim3(kRankLanes), (hipStream_t)0, pk, pc, narr, pp);
  if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return GZ_E_HIP;
  if (hipMemcpy(perm, dp.p, (size_t)narr * 192, hipMemcpyDeviceToHost) != hipSuccess) return GZ_E_HIP;
  return GZ_OK;
}

int gz_probe_arith(int device, int op, const void* a, const void* b, const void* c,
                   void* out, int n) {
  if (!a || !out || n <= 0 || op < 0 || op > 6) return GZ_E_ARG;
  if (probe_device(device) != GZ_OK) return GZ_E_NO_DEVICE;
  const size_t es = (op == 2 || op == 3 || op == 5 || op == 6) ? 8 : 4;
  const size_t os = (op == 2 || op == 3 || op == 5) ? 8 : 4;
  void *da = nullptr, *db = nullptr, *dc = nullptr, *dout = nullptr;
  bool ok = hipMalloc(&da, es * n) == hipSuccess && hipMalloc(&db, es * n) == hipSuccess &&
            hipMalloc(&dc, es * n) == hipSuccess && hipMalloc(&dout, os * n) == hipSuccess;
  ok = ok && hipMemcpy(da, a, es * n, hipMemcpyHostToDevice) == hipSuccess;
  if (ok && b) ok = hipMemcpy(db, b, es * n, hipMemcpyHostToDevice) == hipSuccess;
  if (ok && c) ok = hipMemcpy(dc, c, es * n, hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) {
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dc); (void)hipFree(dout);
    return GZ_E_HIP;
  }
  GZ_LAUNCH(k_probe_arith, dim3(gz_div_up(n, 256)), dim3(256), (hipStream_t)0, op,
            (const void*)da, (const void*)db, (const void*)dc, dout, n);
  int rc = hipGetLastError() == hipSuccess ? GZ_OK : GZ_E_HIP;
  if (hipMemcpy(out, dout, os * n, hipMemcpyDeviceToHost) != hipSuccess) rc = GZ_E_HIP;
  (void)hipFree(da); (void)hipFree(db); (void)hipFree(dc); (void)hipFree(dout);
  return rc;
}


int gz_rank_zeroing_candidates(const int16_t* coeffs, const int16_t* orig, int nb,
                               int new_model, int32_t* offsets, uint8_t* idx) {
  if (!coeffs || !orig || !offsets || !idx || nb <= 0) return GZ_E_ARG;
  std::vector<int32_t> off;
  std::vector<uint8_t> ix;
  rank_all(coeffs, orig, nb, new_model, &off, &ix);
  memcpy(offsets, off.data(), sizeof(int32_t) * (nb + 1));
  memcpy(idx, ix.data(), ix.size());
  return GZ_OK;
}

int gz_block_zeroing_orders(gz_ctx* c, int lookahead, int new_model, int32_t* offsets,
                            uint8_t* idx, float* err, int cap) {
  return gz_block_zeroing_orders_masked(c, 7, lookahead, new_model, offsets, idx, err, cap);
}

int gz_block_zeroing_orders_masked(gz_ctx* c, int comp_mask, int lookahead, int new_model,
                                   int32_t* offsets, uint8_t* idx, float* err, int cap) {
  DeviceScope ds_(c);
  if (!c || !offsets || !idx || lookahead < 1 || cap < 0 || comp_mask < 1 || comp_mask > 7) return GZ_E_ARG;
  if (!c->have_cand || !c->have_orig) { c->err = "needs original and candidate coefficients"; return GZ_E_STATE; }
  // SelectFrequencyMasking's grid (processor.cc:546-552) is that of the mask's last component
  int mode = 0;
  if (c->cfac == 2) {
    if (comp_mask == 1) mode = 1;
    else if (comp_mask == 6) mode = 2;
    else { c->err = "a 4:2:0 frame is searched with component mask 1 or 6"; return GZ_E_ARG; }
  }
  TRY(ensure_block_mask(c));
  c->order_pending = false;   // a new search grid: a pending order of the old one is void
  c->results_in_desc = false;
  const int nb = c->nb;   // capacity of the per-block arrays: the luma grid
  const int gn = mode == 2 ? c->nbc : c->nb;
  c->sg_w = mode == 2 ? c->cbw : c->bw;
  c->sg_h = mode == 2 ? c->cbh : c->bh;
  c->sg_n = gn;
  c->sg_factor = mode == 2 ? 2 : 1;
  c->sg_mask = comp_mask;
  if (!c->d_rank_cnt) {
    HIPCHK(c, pool_malloc((void**)&c->d_rank_cnt, sizeof(int32_t) * nb));
    HIPCHK(c, pool_malloc((void**)&c->d_rank_idx, (size_t)nb * 192));
    HIPCHK(c, pool_malloc((void**)&c->d_rank_tables, sizeof(float) * 384));
    HIPCHK(c, pool_malloc((void**)&c->d_out_cnt, sizeof(int32_t) * nb));
    HIPCHK(c, pool_malloc((void**)&c->d_out_idx, (size_t)nb * 192));
    HIPCHK(c, pool_malloc((void**)&c->d_out_err, sizeof(float) * nb * 192));
    HIPCHK(c, hipMemcpyAsync(c->d_rank_tables, kOrderCsf, sizeof(float) * 192, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_rank_tables + 192, kOrderBias, sizeof(float) * 192, hipMemcpyHostToDevice, c->stream));
  }
  {  // input_order of every block, ranked on the device with std::sort's permutation
    RankArgs r;
    r.coeffs = c->d_cand; r.orig = c->d_orig;
    r.csf = c->d_rank_tables; r.bias = c->d_rank_tables + 192;
    r.nb = gn; r.new_model = new_model; r.comp_mask = comp_mask;
    for (int i = 0; i < 3; ++i) r.coff[i] = c->coff[i];
    r.cnt = c->d_rank_cnt; r.idx = c->d_rank_idx;
    GZ_LAUNCH(k_rank_candidates, dim3(gz_div_up(gn, kRankLanes)), dim3(kRankLanes), c->stream, r);
    KCHK(c);
  }
  SearchArgs a;
  a.coeffs = c->d_cand; a.rank_cnt = c->d_rank_cnt; a.rank_idx = c->d_rank_idx;
  a.rgb = c->d_rgb; a.srgb_lut = c->d_srgb_lut; a.block_mask = c->d_block_mask;
  a.w = c->w; a.h = c->h; a.bw = c->bw; a.nb = nb;
  for (int i = 0; i < 3; ++i) a.coff[i] = c->coff[i];
  a.cbw = c->cbw;
  a.samples = nullptr;
  if (mode != 0) {   // the chroma samples of the image as it stands
    TRY(stage_chroma_samples(c, c->d_cand));
    a.samples = c->d_csamp;
  }
  a.lookahead = lookahead;
  a.limit = c->target;
  {
    // 8x8 OpsinDynamicsImage: Blur(sigma 1.2, border_ratio 0) on an 8x8 image
    BlurCfg cfg;
    make_taps_host((float)kBlurSpecs[B_OPSIN].sigma, &cfg);
    cfg.border_ratio = 0.0f;
    a.taps = taps_of<2>(cfg);
    std::vector<float> lo, hi;
    border_scales_host(cfg, 8, &lo, &hi);
    a.scale_lo[0] = lo[0]; a.scale_lo[1] = lo[1];
    a.scale_hi[0] = hi[0]; a.scale_hi[1] = hi[1];
  }
  a.out_cnt = c->d_out_cnt; a.out_idx = c->d_out_idx; a.out_err = c->d_out_err;
  if (mode == 0) GZ_LAUNCH(k_block_search<0>, dim3(gn), dim3(64), c->stream, a);
  else if (mode == 1) GZ_LAUNCH(k_block_search<1>, dim3(gn), dim3(64), c->stream, a);
  else GZ_LAUNCH(k_block_search<2>, dim3(gn), dim3(256), c->stream, a);
  KCHK(c);
  c->have_search = true;
  c->search_total = 0;   // set below, once the counts are on the host
  std::vector<int32_t> cnt(gn), rcnt(gn);
  HIPCHK(c, hipMemcpyAsync(rcnt.data(), c->d_rank_cnt, sizeof(int32_t) * gn, hipMemcpyDeviceToHost, c->stream));
  std::vector<uint8_t> widx((size_t)gn * 192);
  std::vector<float> werr(err ? (size_t)gn * 192 : 0);   // the errors stay on the device for gz_order_build
  HIPCHK(c, hipMemcpyAsync(cnt.data(), c->d_out_cnt, sizeof(int32_t) * gn, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(widx.data(), c->d_out_idx, (size_t)gn * 192, hipMemcpyDeviceToHost, c->stream));
  if (err)
    HIPCHK(c, hipMemcpyAsync(werr.data(), c->d_out_err, sizeof(float) * gn * 192, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  long total = 0;
  for (int b = 0; b < gn; ++b) total += cnt[b];
  c->search_total = (size_t)total;
  // evaluations: step s of a block with n candidates compares min(lookahead, n - s) of them, on
  // every 8x8 block of its area that lies inside the image
  c->search_evaluations = 0;
  for (int b = 0; b < gn; ++b) {
    unsigned long long e = 0;
    for (int s2 = 0; s2 < rcnt[b]; ++s2) e += (unsigned long long)std::min(lookahead, rcnt[b] - s2);
    int sub = 1;
    if (mode == 2) {
      const int bx = b % c->cbw, by = b / c->cbw;
      sub = ((16 * bx + 8 < c->w) ? 2 : 1) * ((16 * by + 8 < c->h) ? 2 : 1);
    }
    c->search_evaluations += e * sub;
  }
  if (total > cap) { c->err = "candidate capacity too small, need " + std::to_string(total); offsets[gn] = (int32_t)total; return GZ_E_ARG; }
  int t = 0;
  for (int b = 0; b < gn; ++b) {
    offsets[b] = t;
    memcpy(idx + t, widx.data() + (size_t)b * 192, cnt[b]);
    if (err) memcpy(err + t, werr.data() + (size_t)b * 192, sizeof(float) * cnt[b]);
    t += cnt[b];
  }
  offsets[gn] = t;
  return GZ_OK;
}

static void search_args_common(gz_ctx* c, SearchArgs* a) {
  a->coeffs = c->d_cand; a->rank_cnt = c->d_rank_cnt; a->rank_idx = c->d_rank_idx;
  a->rgb = c->d_rgb; a->srgb_lut = c->d_srgb_lut; a->block_mask = c->d_block_mask;
  a->w = c->w; a->h = c->h; a->bw = c->bw; a->nb = c->nb;
  for (int i = 0; i < 3; ++i) a->coff[i] = c->coff[i];
  a->cbw = c->cbw;
  a->samples = nullptr;
  a->lookahead = 3;
  a->limit = c->target;
  // 8x8 OpsinDynamicsImage: Blur(sigma 1.2, border_ratio 0) on an 8x8 image
  BlurCfg cfg;
  make_taps_host((float)kBlurSpecs[B_OPSIN].sigma, &cfg);
  cfg.border_ratio = 0.0f;
  a->taps = taps_of<2>(cfg);
  std::vector<float> lo, hi;
  border_scales_host(cfg, 8, &lo, &hi);
  a->scale_lo[0] = lo[0]; a->scale_lo[1] = lo[1];
  a->scale_hi[0] = hi[0]; a->scale_hi[1] = hi[1];
  a->out_cnt = c->d_out_cnt; a->out_idx = c->d_out_idx; a->out_err = c->d_out_err;
}

int gz_compare_blocks(gz_ctx* c, int n, const int32_t* block_xy, const int16_t* coeffs, double* out) {
  DeviceScope ds_(c);
  if (!c || n < 0 || (n > 0 && (!block_xy || !coeffs || !out))) return GZ_E_ARG;
  if (n == 0) return GZ_OK;
  for (int i = 0; i < n; ++i)
    if (block_xy[2 * i] < 0 || block_xy[2 * i] >= c->bw || block_xy[2 * i + 1] < 0 || block_xy[2 * i + 1] >= c->bh)
      return GZ_E_ARG;
  if (c->cfac != 1) { c->err = "gz_compare_blocks takes coefficient blocks of a 4:4:4 frame (gz_compare_block_pixels serves any frame)"; return GZ_E_STATE; }
  TRY(ensure_block_mask(c));
  // staging: positions, coefficients and results share one device block kept by the context
  const size_t need = (size_t)n * (8 + 384 + 8);
  if (need > c->cmp_stage_cap) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)pool_free(c->d_cmp_stage);
    c->d_cmp_stage = nullptr;
    c->cmp_stage_cap = 0;
    const size_t cap = std::max<size_t>(need, 4096);
    HIPCHK(c, pool_malloc(&c->d_cmp_stage, cap));
    c->cmp_stage_cap = cap;
  }
  int32_t* d_xy = (int32_t*)c->d_cmp_stage;
  double* d_out = (double*)((char*)c->d_cmp_stage + (size_t)n * 8);
  int16_t* d_blk = (int16_t*)((char*)c->d_cmp_stage + (size_t)n * 16);
  HIPCHK(c, hipMemcpyAsync(d_xy, block_xy, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_blk, coeffs, (size_t)n * 384, hipMemcpyHostToDevice, c->stream));
  SearchArgs a;
  search_args_common(c, &a);
  GZ_LAUNCH(k_compare_blocks, dim3(n), dim3(64), c->stream, a, (const int32_t*)d_xy, (const int16_t*)d_blk, n, d_out);
  KCHK(c);
  HIPCHK(c, hipMemcpyAsync(out, d_out, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GZ_OK;
}

int gz_compare_block_pixels(gz_ctx* c, int n, const int32_t* block_xy, const uint8_t* ycc, double* out) {
  DeviceScope ds_(c);
  if (!c || n < 0 || (n > 0 && (!block_xy || !ycc || !out))) return GZ_E_ARG;
  if (n == 0) return GZ_OK;
  for (int i = 0; i < n; ++i)
    if (block_xy[2 * i] < 0 || block_xy[2 * i] >= c->bw || block_xy[2 * i + 1] < 0 || block_xy[2 * i + 1] >= c->bh)
      return GZ_E_ARG;
  TRY(ensure_block_mask(c));
  // staging: positions, pixels and results share one device block kept by the context
  const size_t need = (size_t)n * (8 + 8 + 192);
  if (need > c->cmp_stage_cap) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)pool_free(c->d_cmp_stage);
    c->d_cmp_stage = nullptr;
    c->cmp_stage_cap = 0;
    const size_t cap = std::max<size_t>(need, 4096);
    HIPCHK(c, pool_malloc(&c->d_cmp_stage, cap));
    c->cmp_stage_cap = cap;
  }
  int32_t* d_xy = (int32_t*)c->d_cmp_stage;
  double* d_out = (double*)((char*)c->d_cmp_stage + (size_t)n * 8);
  uint8_t* d_px = (uint8_t*)c->d_cmp_stage + (size_t)n * 16;
  HIPCHK(c, hipMemcpyAsync(d_xy, block_xy, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_px, ycc, (size_t)n * 192, hipMemcpyHostToDevice, c->stream));
  SearchArgs a;
  search_args_common(c, &a);
  GZ_LAUNCH(k_compare_block_pixels, dim3(n), dim3(64), c->stream, a, (const int32_t*)d_xy,
            (const uint8_t*)d_px, n, d_out);
  KCHK(c);
  HIPCHK(c, hipMemcpyAsync(out, d_out, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GZ_OK;
}

int gz_set_frame(gz_ctx* c, int chroma_factor) {
  DeviceScope ds_(c);
  if (!c || (chroma_factor != 1 && chroma_factor != 2)) return GZ_E_ARG;
  if (c->cfac == chroma_factor) return GZ_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  set_frame(c, chroma_factor);
  c->have_cand = false;
  c->have_orig = false;   // the original coefficients on the device belonged to the other frame
  return GZ_OK;
}

int gz_search_evaluations(gz_ctx* c, uint64_t* evaluations) {
  DeviceScope ds_(c);
  if (!c || !evaluations) return GZ_E_ARG;
  *evaluations = c->search_evaluations;
  return GZ_OK;
}

// OutputImage::Downsample (output_image.cc:304-340), cfg defaults of Processor::DownsampleImage
// (processor.cc:97-104) without the silver-screen option, on the ORIGINAL coefficients of a
// 4:4:4 frame: ToFloatPixels of the three components, PreProcessChannel on V then on U
// (preprocess_downsample.cc:157-279), SetDownsampledCoefficients of U and V by 2 x 2.
static void normal_taps(double sigma, float k[5], float* mul) {   // Normal(), :85-88; Sharpen / Blur :92-100,138-146
  double kernel[5], sum = 0;
  for (size_t i = 0; i < 5; ++i) {
    const double x = 1.0 * i - 5 / 2;
    static const double kInvSqrt2Pi = 0.3989422804014327;
    kernel[i] = exp(-x * x / (2 * sigma * sigma)) * kInvSqrt2Pi / sigma;
  }
  for (size_t i = 0; i < 5; ++i) sum += kernel[i];
  for (size_t i = 0; i < 5; ++i) k[i] = static_cast<float>(kernel[i]);
  *mul = static_cast<float>(1.0 / sum);
}

int gz_downsample(gz_ctx* c, int16_t* coeffs_out) {
  DeviceScope ds_(c);
  if (!c) return GZ_E_ARG;
  if (!c->have_orig || c->cfac != 1) { c->err = "gz_downsample needs the original coefficients of a 4:4:4 frame"; return GZ_E_STATE; }
  const int w = c->w, h = c->h;
  const size_t n = (size_t)w * h;
  // scratch: planes of the candidate's evaluation (nothing of it is in flight here)
  float* yuv[3] = {c->xyb[0], c->xyb[1], c->xyb[2]};
  float* tmp_s = c->tmp[0];
  float* tmp_b = c->tmp[1];
  // byte planes: four per float plane
  uint8_t* bp0 = reinterpret_cast<uint8_t*>(c->tmp[2]);
  uint8_t* bp1 = reinterpret_cast<uint8_t*>(c->lf_raw[0]);
  uint8_t* dark_a = bp0; uint8_t* dark_b = bp0 + n; uint8_t* red_a = bp0 + 2 * n; uint8_t* red_b = bp0 + 3 * n;
  uint8_t* sharpen = bp1; uint8_t* blurm = bp1 + n; uint8_t* blur_t = bp1 + 2 * n;
  for (int i = 0; i < 3; ++i) {
    GZ_LAUNCH(k_to_float_pixels, dim3(gz_div_up(c->nb, kBlocksPerWG)), dim3(256), c->stream,
              (const int16_t*)(c->d_orig + (size_t)c->coff[i] * 64), w, h, c->bw, c->nb, yuv[i]);
    KCHK(c);
  }
  PPTaps taps;
  normal_taps((double)1.3f, taps.ks, &taps.mul_s);   // Sharpen(sigma = 1.3f)
  normal_taps(1.3, taps.kb, &taps.mul_b);            // Blur: kSigma = 1.3
  const dim3 g1(gz_div_up((int)std::min<size_t>(n, 0x7fffffff), 256)), g2(gz_div_up(w, 256), h);
  const int channels[2] = {2, 1};   // :326-329
  for (int pass = 0; pass < 2; ++pass) {
    const int channel = channels[pass];
    GZ_LAUNCH(k_pp_normalize, g1, dim3(256), c->stream, yuv[0], yuv[1], yuv[2], n);
    KCHK(c);
    GZ_LAUNCH(k_pp_maps, g1, dim3(256), c->stream, (const float*)yuv[0], (const float*)yuv[1],
              (const float*)yuv[2], n, channel, dark_a, red_a);
    KCHK(c);
    // Erode x3 (darkmap, :194-196): a -> b -> a -> b; Dilate x3 (redmap, :217-219) likewise
    for (int i = 0; i < 3; ++i) {
      GZ_LAUNCH(k_pp_morph, g2, dim3(256), c->stream, (const uint8_t*)(i & 1 ? dark_b : dark_a),
                i & 1 ? dark_a : dark_b, w, h, 1);
      KCHK(c);
      GZ_LAUNCH(k_pp_morph, g2, dim3(256), c->stream, (const uint8_t*)(i & 1 ? red_b : red_a),
                i & 1 ? red_a : red_b, w, h, 0);
      KCHK(c);
    }
    const double threshold = (channel == 2 ? 0.02 : 1.0) * 127.5;
    GZ_LAUNCH(k_pp_edge_maps, g2, dim3(256), c->stream, (const float*)yuv[channel], (const float*)yuv[1],
              (const float*)yuv[2], (const uint8_t*)dark_b, (const uint8_t*)red_b, w, h,
              threshold, sharpen, blurm);
    KCHK(c);
    // Erode x2 (blurmap, :254-255)
    GZ_LAUNCH(k_pp_morph, g2, dim3(256), c->stream, (const uint8_t*)blurm, blur_t, w, h, 1);
    KCHK(c);
    GZ_LAUNCH(k_pp_morph, g2, dim3(256), c->stream, (const uint8_t*)blur_t, blurm, w, h, 1);
    KCHK(c);
    GZ_LAUNCH(k_pp_conv_h, g2, dim3(256), c->stream, (const float*)yuv[channel], w, h, taps, tmp_s, tmp_b);
    KCHK(c);
    GZ_LAUNCH(k_pp_conv_v_select, g2, dim3(256), c->stream, yuv[channel], (const float*)tmp_s,
              (const float*)tmp_b, (const uint8_t*)sharpen, (const uint8_t*)blurm, w, h, taps, 0.5f, 1, 1);
    KCHK(c);
    GZ_LAUNCH(k_pp_denormalize, g1, dim3(256), c->stream, yuv[0], yuv[1], yuv[2], n);
    KCHK(c);
  }
  // the two chroma components, 2 x 2 subsampled, replace the 4:4:4 ones (luma is kept as it is)
  const int cbw = (w + 15) / 16, cbh = (h + 15) / 16, nbc = cbw * cbh;
  for (int i = 1; i < 3; ++i) {
    int16_t* dst = c->d_orig + ((size_t)c->nb + (size_t)(i - 1) * nbc) * 64;
    GZ_LAUNCH(k_set_downsampled_coeffs, dim3(gz_div_up(nbc, kBlocksPerWG)), dim3(256), c->stream,
              (const float*)yuv[i], w, h, 2, 2, cbw, nbc, dst);
    KCHK(c);
  }
  set_frame(c, 2);
  c->have_cand = false;
  c->have_distmap = false;
  if (coeffs_out)
    HIPCHK(c, hipMemcpyAsync(coeffs_out, c->d_orig, (size_t)c->nblk * 128, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GZ_OK;
}

int gz_downsample_planes(gz_ctx* c, const float* y, const float* u, const float* v, int16_t* coeffs_out) {
  DeviceScope ds_(c);
  if (!c || !y || !u || !v) return GZ_E_ARG;
  if (!c->have_orig || c->cfac != 1) { c->err = "gz_downsample_planes needs the original coefficients of a 4:4:4 frame"; return GZ_E_STATE; }
  const int w = c->w, h = c->h;
  const size_t n = (size_t)w * h;
  const float* src[3] = {y, u, v};
  for (int i = 0; i < 3; ++i)
    HIPCHK(c, hipMemcpyAsync(c->xyb[i], src[i], n * sizeof(float), hipMemcpyHostToDevice, c->stream));
  // output_image.cc:314-316: every component from its plane, luma included (factor 1), the
  // chroma blocks packed behind the nb luma blocks
  const int cbw = (w + 15) / 16, cbh = (h + 15) / 16, nbc = cbw * cbh;
  GZ_LAUNCH(k_set_downsampled_coeffs, dim3(gz_div_up(c->nb, kBlocksPerWG)), dim3(256), c->stream,
            (const float*)c->xyb[0], w, h, 1, 1, c->bw, c->nb, c->d_orig);
  KCHK(c);
  for (int i = 1; i < 3; ++i) {
    int16_t* dst = c->d_orig + ((size_t)c->nb + (size_t)(i - 1) * nbc) * 64;
    GZ_LAUNCH(k_set_downsampled_coeffs, dim3(gz_div_up(nbc, kBlocksPerWG)), dim3(256), c->stream,
              (const float*)c->xyb[i], w, h, 2, 2, cbw, nbc, dst);
    KCHK(c);
  }
  set_frame(c, 2);
  c->have_cand = false;
  c->have_distmap = false;
  if (coeffs_out)
    HIPCHK(c, hipMemcpyAsync(coeffs_out, c->d_orig, (size_t)c->nblk * 128, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GZ_OK;
}

}  // extern "C"
