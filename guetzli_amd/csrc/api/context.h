// The caching allocator and the stream / event pools, pinned staging, gz_ctx (one image on one GPU), DeviceScope, the error macros, the plane arena.
// (part of the one translation unit gz_api.hip, which includes these files in order; split by
// concern in round 5 -- no declaration here is visible outside libguetzli_amd.so but the C ABI)
#pragma once


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
  std::multimap<int, hipEvent_t> events;
};
HandlePool& handle_pool() { static HandlePool p; return p; }
// prio: 0 = default, +1 = the device's highest priority, -1 = its lowest.  Who takes which:
// create_context.
static bool stream_priorities() {   // GZ_STREAM_PRIO=0: no priority stream even for a lone context (A/B)
  static const bool v = [] { const char* e = getenv("GZ_STREAM_PRIO"); return e ? atoi(e) != 0 : true; }();
  return v;
}
// Contexts alive per device: adds `delta`, returns the count before.
static int live_contexts(int device, int delta) {
  static std::mutex mu;
  static std::map<int, int> live;
  std::lock_guard<std::mutex> lk(mu);
  const int before = live[device];
  live[device] = before + delta;
  return before;
}
// A context's four streams are created, pooled and reused TOGETHER.  The HIP runtime spreads streams
// over its hardware queues (GPU_MAX_HW_QUEUES, 4 by default) in creation order, so four streams made
// back to back sit on four queues; pooled one by one (rounds 1-4), the streams that came back from the
// contexts of a batch were handed out in whatever order the threads had destroyed them, and a later
// context could get its main and side streams on ONE queue: its chain then ran serialised -- 1.58
// instead of 0.96 ms per Compare at 4K after a four-in-flight batch in the same process, a 4K
// quality-84 encode 0.177 instead of 0.150 s (profiles/r05_chain_experiments.log, section 10).
struct StreamSet {
  hipStream_t own = nullptr, side = nullptr, side2 = nullptr, entropy = nullptr;
  int rot = 0;   // position of `own` in the set's creation order = the hardware queue (mod 4) its main stream sits on
};
struct StreamSetPool {
  std::mutex mu;
  std::multimap<int, StreamSet> sets;   // stream_set_key(): device, CU class, main stream at the highest priority
};
inline StreamSetPool& stream_set_pool() { static StreamSetPool p; return p; }

// CU-partitioned stream sets (round 6; GZ_CU_PARTITION = P in {2, 4, 8}, default off): the contexts alive on a
// device take SLOTS (the lowest free one), and the four streams of the context in slot s are created with
// hipExtStreamCreateWithCUMask on the (s mod P)-th P-th of the device's CUs, so that P images in flight
// run on disjoint CUs instead of time-sharing all of them.  Which CUs a run of mask bits names: the
// kernel driver deals the bits round-robin over the XCDs and, inside an XCD, over its shader engines
// (bit i -> XCD i mod 8, engine (i / 8) mod 4), so a contiguous P-th of the 256 bits is 32 / P CUs of
// EVERY XCD, spread over its engines -- every stream still sees all eight L2s and a dispatch's
// round-robin of workgroups over the XCDs finds CUs on each (tools/ubench/cumask.hip prints the map).
// GZ_CU_MAIN / GZ_CU_SIDE = "lo:hi" (bit ranges; experiments on ONE image: the chain's main stream and
// its two side streams on separate CUs).  CU class 0 = no mask.
struct CuPlan { int parts = 0; int main_lo = -1, main_hi = -1, side_lo = -1, side_hi = -1; };
inline const CuPlan& cu_plan() {
  static const CuPlan plan = [] {
    CuPlan p;
    if (const char* e = getenv("GZ_CU_PARTITION")) { const int v = atoi(e); if (v == 2 || v == 4 || v == 8) p.parts = v; }
    if (const char* e = getenv("GZ_CU_MAIN")) (void)sscanf(e, "%d:%d", &p.main_lo, &p.main_hi);
    if (const char* e = getenv("GZ_CU_SIDE")) (void)sscanf(e, "%d:%d", &p.side_lo, &p.side_hi);
    return p;
  }();
  return plan;
}
// Slots of the contexts alive per device (lowest free first).
struct CuSlots {
  std::mutex mu;
  std::map<int, std::vector<bool> > used;
};
inline CuSlots& cu_slots() { static CuSlots s; return s; }
inline int cu_slot_take(int device) {
  CuSlots& cs = cu_slots();
  std::lock_guard<std::mutex> lk(cs.mu);
  std::vector<bool>& u = cs.used[device];
  for (size_t i = 0; i < u.size(); ++i) if (!u[i]) { u[i] = true; return (int)i; }
  u.push_back(true);
  return (int)u.size() - 1;
}
inline void cu_slot_release(int device, int slot) {
  CuSlots& cs = cu_slots();
  std::lock_guard<std::mutex> lk(cs.mu);
  std::vector<bool>& u = cs.used[device];
  if (slot >= 0 && (size_t)slot < u.size()) u[(size_t)slot] = false;
}
#ifndef GZ_EMU
static hipError_t create_stream_on_cus(hipStream_t* out, int lo, int hi) {
  hipDeviceProp_t prop;
  int device = 0;
  (void)hipGetDevice(&device);
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) return e;
  const int ncu = prop.multiProcessorCount;
  lo = std::max(0, std::min(lo, ncu)); hi = std::max(lo, std::min(hi, ncu));
  if (hi - lo <= 0 || hi - lo >= ncu) return hipStreamCreate(out);
  std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
  for (int i = lo; i < hi; ++i) mask[(size_t)i >> 5] |= 1u << (i & 31);
  return hipExtStreamCreateWithCUMask(out, (uint32_t)mask.size(), mask.data());
}
#endif
inline int stream_set_key(int device, bool prio_main, int cu_class) {
  return (device * 64 + cu_class) * 2 + (prio_main && stream_priorities() ? 1 : 0);
}

// want_rot >= 0 (the default; GZ_SET_SLOT=0 switches it off): a set whose main stream sits on that hardware queue -- the contexts alive on a
// device take slots, slot s asks for rotation s mod 4, so that four images in flight have their main streams on four
// different queues whatever order the images finished in.
hipError_t pool_stream_set_create(StreamSet* out, bool prio_main, int cu_class, int want_rot = -1) {
#ifndef GZ_EMU
  int device = 0;
  (void)hipGetDevice(&device);
  const int key = stream_set_key(device, prio_main, cu_class);
  StreamSetPool& p = stream_set_pool();
  std::lock_guard<std::mutex> lk(p.mu);   // (also keeps two threads' creations from interleaving)
  {
    auto range = p.sets.equal_range(key);
    for (auto it = range.first; it != range.second; ++it)
      if (want_rot < 0 || it->second.rot == want_rot) { *out = it->second; p.sets.erase(it); return hipSuccess; }
  }
  // The k-th set is created in an order rotated by k, so that the k-th context's MAIN stream sits on
  // hardware queue k mod 4 and its three other streams on the three other queues: with every set made
  // in the same order the main streams of the four images in flight all shared one queue, and a batch
  // of eight 4K images fell from 39 to 32 MPix/s (section 10 of the experiment log).
  static int serial = 0;
  const int rot = want_rot >= 0 ? (want_rot & 3) : ((serial++) & 3);
  out->rot = rot;
  hipStream_t* slot[4] = {&out->own, &out->side, &out->side2, &out->entropy};
  int least = 0, greatest = 0;
  const bool prio = (key & 1) && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest;
  const CuPlan& cp = cu_plan();
  hipError_t e = hipSuccess;
  for (int j = 0; j < 4 && e == hipSuccess; ++j) {
    const int which = (j + 4 - rot) & 3;   // position j of the creation order takes stream `which`: own at position rot
    if (cu_class > 0 && cp.parts > 0) {        // a batch's image: all four streams on its share of the CUs
      int ncu = 256;
      (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device);
      const int per = ncu / cp.parts, s = (cu_class - 1) % cp.parts;
      e = create_stream_on_cus(slot[which], s * per, (s + 1) * per);
    } else if (cu_class > 0 && which == 0 && cp.main_lo >= 0) {
      e = create_stream_on_cus(slot[which], cp.main_lo, cp.main_hi);
    } else if (cu_class > 0 && (which == 1 || which == 2) && cp.side_lo >= 0) {
      e = create_stream_on_cus(slot[which], cp.side_lo, cp.side_hi);
    } else {
      e = which == 0 && prio ? hipStreamCreateWithPriority(slot[which], hipStreamDefault, greatest)
                             : hipStreamCreate(slot[which]);
    }
  }
  return e;
#else
  hipError_t e = hipStreamCreate(&out->own);
  if (e == hipSuccess) e = hipStreamCreate(&out->side);
  if (e == hipSuccess) e = hipStreamCreate(&out->side2);
  if (e == hipSuccess) e = hipStreamCreate(&out->entropy);
  return e;
#endif
}
void pool_stream_set_destroy(const StreamSet& s_, bool prio_main, int cu_class) {
#ifndef GZ_EMU
  if (s_.own && s_.side && s_.side2 && s_.entropy && pool_limit_bytes() != 0) {
    int device = 0;
    (void)hipGetDevice(&device);
    const int key = stream_set_key(device, prio_main, cu_class);
    StreamSetPool& p = stream_set_pool();
    std::lock_guard<std::mutex> lk(p.mu);
    if (p.sets.count(key) < 16) { p.sets.insert(std::make_pair(key, s_)); return; }
  }
#endif
  if (s_.own) (void)hipStreamDestroy(s_.own);
  if (s_.side) (void)hipStreamDestroy(s_.side);
  if (s_.side2) (void)hipStreamDestroy(s_.side2);
  if (s_.entropy) (void)hipStreamDestroy(s_.entropy);
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
  bool single_now = false;   // (chain.h choose_streams: this Compare's kernels on the main stream only)
  bool side_small = false;   // (chain.h: set while the side branches' launches are made, cfg.side_small)
  gz_config cfg;             // run-time configuration (include/guetzli_amd.h): the environment's, read once at gz_create
  int set_rot = 0;           // the stream set's rotation (goes back to the pool with it)
  int cu_slot = -1, cu_class = 0;   // CU-partitioned stream sets (cu_plan): the context's slot; 0 = unmasked streams
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
  int32_t* d_csr_off = nullptr;   // the search's CSR offsets (k_csr_offsets); the packed indices reuse d_rank_idx

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
  bool adv_pending = false; float adv_threshold = 0.0f; int adv_direction = 0;         // gz_order_advance's update, still due
  bool have_search = false;
  unsigned char* d_wflag = nullptr;                               // [nb]
  int* d_edit_pos = nullptr; short* d_edit_val = nullptr; size_t edit_cap = 0;

  bool have_orig = false, have_cand = false, have_distmap = false;
  // lin[] holds the reconstruction of d_cand as it is now (4:4:4 frames; cfg.patch_reconstruct): set by the Compare
  // chain's full reconstruction, kept by the mutators that transform the block positions they change
  // (gz_apply_candidate_steps, gz_apply_coeff_edits), dropped by everything else that writes d_cand or lin[]
  bool lin_is_cand = false;
  // ... and xyb[] the opsin image of those planes (cfg.opsin_ahead): the Compare chain's second kernel enqueued AHEAD,
  // behind the bulk steps' patches, while the host takes its serial steps; their edits then cost the opsin tiles around
  // the edited blocks (gz_apply_coeff_edits), and the Compare starts at its third kernel.  Consumed by that Compare.
  bool xyb_is_cand = false;
  std::vector<unsigned char> tile_mark;   // opsin tiles already listed (gz_apply_coeff_edits)
  std::vector<float> h_block_max;
  bool h_block_max_valid = false;
  bool compare_pending = false;
  int h_jq[192] = {0};       // the matrix d_jq holds
  unsigned* d_step_delta = nullptr; bool have_step_delta = false;   // AC statistics change of the last bulk steps
  hipEvent_t ev_steps = nullptr; bool step_delta_event = false;     // ... are in place (recorded when more work follows them on the stream)
  void* h_step_delta = nullptr;   // ... as k_steps_hist_sum writes it: 768 ints, page-locked + mapped
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
      return e_ == hipErrorOutOfMemory ? GZ_E_NOMEM : GZ_E_HIP;                      \
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

}  // namespace
