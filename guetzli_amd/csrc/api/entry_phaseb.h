// C ABI: phase B of SelectFrequencyMasking on the device -- global candidate order, libstdc++ partitions, the quick-select descent, bulk coefficient steps with their symbol statistics, coefficient edits.
// (part of the one translation unit gz_api.hip, which includes these files in order; split by
// concern in round 5 -- no declaration here is visible outside libguetzli_amd.so but the C ABI)
#pragma once

extern "C" {


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

// gz_order_advance's update of max_block_error, if it is still due (it normally rides on the next order's
// k_weights_gather): made now, for whoever reads or replaces d_max_err / d_weight some other way.
static int flush_order_advance(gz_ctx* c) {
  if (!c->adv_pending) return GZ_OK;
  c->adv_pending = false;
  GZ_LAUNCH(k_order_advance, dim3(gz_div_up(c->sg_n, 256)), dim3(256), c->stream, c->d_max_err,
            (const float*)c->d_weight, c->adv_threshold, c->adv_direction, c->sg_n);
  KCHK(c);
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
  c->adv_pending = false;   // (the caller's own max_block_error and weights replace the device's)
  HIPCHK(c, hipMemcpyAsync(c->d_next_cand, next_cand, sizeof(int) * nb, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_weight, block_weight, sizeof(float) * nb, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_max_err, max_block_error, sizeof(float) * nb, hipMemcpyHostToDevice, c->stream));
  return order_build_device(c, direction, count_below, limit, total, blocks_to_change, below);
}

int gz_order_reset(gz_ctx* c) {
  DeviceScope ds_(c);
  if (!c) return GZ_E_ARG;
  TRY(ensure_order_block_arrays(c));
  c->adv_pending = false;
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
    // In batch mode -- the evaluation in flight was put on ONE stream because other contexts are alive (chain.h,
    // choose_streams) -- the copy stays on the main stream too: on the side stream it shares a hardware queue with
    // ANOTHER image's main stream and waits there behind that image's kernels, and the order with it (8 x 4K
    // 40.7 -> 43.0 MPix/s, 16 x 1080p 34.6 -> 36.2, 64 x 1 MPix 29.6 -> 32.6: r06_chain_experiments.log, section 11).
    const bool beside = c->compare_pending && !c->single_now;
    hipStream_t up = beside ? c->side_stream : c->stream;
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
            (unsigned*)c->d_order_off, c->d_max_err, c->adv_threshold, c->adv_pending ? c->adv_direction : 0);
  c->adv_pending = false;   // (gz_order_advance's update was made by this launch, with the weights it replaced)
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
  // max_block_error[i] += block_weight[i] * val_threshold * direction -- due, not launched: the next order's
  // k_weights_gather makes the update with the weights it is about to replace (nothing reads max_block_error before
  // it; whoever else touches it or the weights calls flush_order_advance first)
  TRY(flush_order_advance(c));   // (two advances in a row: the first is made now)
  c->adv_pending = true;
  c->adv_threshold = val_threshold;
  c->adv_direction = direction;
  return GZ_OK;
}

// Does the call that changes n block positions of the candidate keep its linear planes current (transforming those
// positions again), or leave them to the next Compare's full reconstruction?  (Decides, and drops the planes' claim.)
// (Below half a megapixel a full reconstruction costs what a patch launch costs, and an iteration is made of launches:
// such images keep it, unless the tests' checking mode asks for patches at every size.)
static bool patch_wanted(gz_ctx* c, int n) {
  const bool patch = c->cfg.patch_reconstruct != 0 && c->lin_is_cand && c->cfac == 1 && (long)n * 2 <= (long)c->nb &&
                     (c->nb >= 8192 || c->cfg.patch_reconstruct == 2);
  if (!patch) c->lin_is_cand = c->xyb_is_cand = false;
  return patch;
}
static PatchPlanes patch_planes(const gz_ctx* c, bool on) {
  PatchPlanes pp;
  pp.bw = c->bw; pp.w = c->w; pp.h = c->h; pp.pitch = c->pitch;
  pp.pstride = c->plane;
  pp.srgb_lut = c->d_srgb_lut;
  pp.lin = on ? c->lin[0] : nullptr;
  return pp;
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
  {   // (26 000 pairs per iteration of a 4K encode: as a reduction the check vectorises, ~2 us instead of ~10)
    const unsigned sg_n = (unsigned)c->sg_n;
    unsigned bad = 0;
    for (int i = 0; i < n; ++i) bad |= (unsigned)((unsigned)blocks[i] >= sg_n) | (unsigned)((unsigned)counts[i] > 192u);
    if (bad) return GZ_E_ARG;
  }
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
  void* h = nullptr;
  TRY(stage_reserve(c, &c->stage_main, sizeof(int) * 2 * n, &h));
  memcpy(h, blocks, sizeof(int) * n);
  memcpy((int*)h + n, counts, sizeof(int) * n);
  StepGeom sg;
  for (int i = 0; i < 3; ++i) sg.coff[i] = c->coff[i];
  sg.comp_mask = c->sg_mask;
  c->have_step_delta = false;
  if (c->have_jq) {
    // with the symbol statistics' quantiser known, the steps also report what they do to the
    // AC histograms (gz_steps_histogram_delta)
    // (zeroed once: k_steps_hist_sum leaves the counters zeroed)
    if (!c->d_step_delta) {
      HIPCHK(c, pool_malloc((void**)&c->d_step_delta, sizeof(unsigned) * 768 * kStepDeltaCopies));
      HIPCHK(c, hipMemsetAsync(c->d_step_delta, 0, sizeof(unsigned) * 768 * kStepDeltaCopies, c->stream));
    }
    if (!c->h_step_delta) HIPCHK(c, pool_host_malloc(&c->h_step_delta, sizeof(int) * 768));
    // (persistent workgroups: four per CU's worth at most, each wavefront taking several blocks)
    // (the kernel reads the staging buffer itself -- page-locked and mapped: a copy command in front of
    // it costs more in hand-overs between commands than the 8 bytes per block cost over the bus)
#ifdef GZ_STEPS_COPY_IN   // (A/B: the copy command in front of the kernel)
    HIPCHK(c, hipMemcpyAsync(d_blocks, h, sizeof(int) * 2 * n, hipMemcpyHostToDevice, c->stream));
    h = d_blocks;
#endif
    // the touched block positions of the candidate's linear planes are transformed again behind the statistics
    // (chain.h, enqueue_compare), while they are a minority: beyond that the next Compare reconstructs the image
    const bool patch = patch_wanted(c, n);
    GZ_LAUNCH(k_apply_steps_hist, dim3(std::min(gz_div_up(n, 4), kStepHistGrid)), dim3(256), c->stream, (const int*)h,
              (const int*)h + n, n, direction, (const int*)c->d_next_cand,
              (const unsigned char*)c->d_out_idx, (const short*)c->d_orig, (short*)c->d_cand,
              (const int*)c->d_q, (const int*)c->d_jq, sg, c->d_step_delta, patch ? d_blocks : (int*)nullptr);
    // the staging buffer is free again behind THIS kernel (the only reader): marked before anything can fail, so
    // that no path returns with the kernel still reading a buffer the next stage_reserve hands out
    TRY(stage_sent(c, &c->stage_main, c->stream));
    KCHK(c);
    GZ_LAUNCH(k_steps_hist_sum, dim3(1), dim3(256), c->stream, c->d_step_delta, (int*)c->h_step_delta);
    KCHK(c);
    c->have_step_delta = true;
    c->step_delta_event = false;
    if (patch) {
      // gz_steps_histogram_delta waits for the sums, not for the stream: the patches run while the host reads them
      if (!c->ev_steps) HIPCHK(c, pool_event_create(&c->ev_steps));
      HIPCHK(c, hipEventRecord(c->ev_steps, c->stream));
      c->step_delta_event = true;
      GZ_LAUNCH((k_reconstruct_listed<false>), dim3(gz_div_up(n, kBlocksPerWG)), dim3(256), c->stream, (const int*)d_blocks, n,
                (const int16_t*)c->d_cand, c->nb, patch_planes(c, true));
      KCHK(c);
      // ... and, for a context that has the device to itself, the next Compare's opsin blur of the planes as they
      // are now: 100 us (4K) of the chain's critical path that run while the host takes its serial steps, whose
      // edits then cost the tiles around them (gz_apply_coeff_edits).  Not in a batch: the other images use the time.
      c->xyb_is_cand = false;
      if (c->cfg.opsin_ahead != 0 && !single_stream_wanted(c)) {
        TRY(stage_opsin(c));
        c->xyb_is_cand = true;
      }
    }
    return GZ_OK;
  }
  c->lin_is_cand = c->xyb_is_cand = false;
  HIPCHK(c, hipMemcpyAsync(d_blocks, h, sizeof(int) * 2 * n, hipMemcpyHostToDevice, c->stream));
  TRY(stage_sent(c, &c->stage_main, c->stream));
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
  // k_steps_hist_sum has written the sums into the context's page-locked buffer
  if (c->step_delta_event) HIPCHK(c, hipEventSynchronize(c->ev_steps));
  else HIPCHK(c, hipStreamSynchronize(c->stream));
  memcpy(ac_delta, c->h_step_delta, sizeof(int32_t) * 768);
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
  // (+ the list of opsin tiles around the edited blocks: at most four per edit, behind the values)
  const size_t tiles_at = ((sizeof(int) + sizeof(short)) * (size_t)n + 15) & ~(size_t)15;
  TRY(stage_reserve(c, &c->stage_edits, tiles_at + (direct ? sizeof(int) * 4 * (size_t)n : 0), &h));
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
  int rc_tiles = GZ_OK;   // (reported behind stage_sent: no path returns with a kernel still reading the staging buffer)
  if (patch_wanted(c, n)) {   // the edited block positions' pixels, behind the edits (one wavefront per edit)
    GZ_LAUNCH((k_reconstruct_listed<true>), dim3(gz_div_up(n, kBlocksPerWG)), dim3(256), c->stream, k_pos, n,
              (const int16_t*)c->d_cand, c->nb, patch_planes(c, true));
    if (c->xyb_is_cand) {
      // the opsin image enqueued ahead (gz_apply_candidate_steps): a pixel's value depends on the linear planes
      // within 2 pixels of it (radius-2 blur), so the tiles that meet an edited block grown by 2 are computed again
      // -- or the image, when the edits are many (the last iterations of a search)
      const int th = opsin_tile_rows(c), gx = gz_div_up(c->w, T2), gy = gz_div_up(c->h, th);
      int nt = 0;
      int* tl = direct ? (int*)((char*)h + tiles_at) : nullptr;
      if (tl) {
        if (c->tile_mark.size() != (size_t)gx * gy) c->tile_mark.assign((size_t)gx * gy, 0);
        for (int i = 0; i < n; ++i) {
          const int b = (pos[i] >> 6) % c->nb, bx = b % c->bw, by = b / c->bw;
          const int tx0 = std::max(0, 8 * bx - 2) / T2, tx1 = std::min(c->w - 1, 8 * bx + 9) / T2;
          const int ty0 = std::max(0, 8 * by - 2) / th, ty1 = std::min(c->h - 1, 8 * by + 9) / th;
          for (int ty = ty0; ty <= ty1; ++ty)
            for (int tx = tx0; tx <= tx1; ++tx) {
              unsigned char& m = c->tile_mark[(size_t)ty * gx + tx];
              if (!m) { m = 1; tl[nt++] = ty * gx + tx; }
            }
        }
        for (int i = 0; i < nt; ++i) c->tile_mark[(size_t)tl[i]] = 0;
      }
      rc_tiles = tl && (long)nt * 4 <= (long)gx * gy ? stage_opsin(c, tl, nt) : stage_opsin(c);
    }
  }
  TRY(stage_sent(c, &c->stage_edits, c->stream));   // (the staging buffer is free again behind the kernels)
  KCHK(c);
  TRY(rc_tiles);
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

}  // extern "C"
