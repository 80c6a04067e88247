// C ABI: library and context life cycle (gz_create / gz_set_rgb / gz_destroy, pools, streams).
// (part of the one translation unit gz_api.hip, which includes these files in order; split by
// concern in round 5 -- no declaration here is visible outside libguetzli_amd.so but the C ABI)
#pragma once

extern "C" {


int gz_abi_version(void) { return 5; }

int gz_config_from_environment(gz_config* out) {
  if (!out) return GZ_E_ARG;
  gz_config c;
  memset(&c, 0, sizeof(c));
  c.struct_size = (int)sizeof(gz_config);
  c.blur_packed = -1;
  c.single_stream = -1;
  if (const char* e = getenv("GZ_BLUR_PK")) c.blur_packed = atoi(e) != 0;
  if (const char* e = getenv("GZ_TILE_ROWS")) { const int v = atoi(e); if (v == 16 || v == 32) c.tile_rows = v; }
  if (const char* e = getenv("GZ_SINGLE_STREAM")) c.single_stream = atoi(e) != 0;
  if (const char* e = getenv("GZ_STORE_DISTMAP")) c.store_distmap = atoi(e) != 0;
  if (const char* e = getenv("GZ_SIDE_SMALL")) c.side_small = atoi(e) != 0;
  if (const char* e = getenv("GZ_MALTA_PAD")) c.malta_pad_bytes = std::max(0, std::min(64 << 10, atoi(e)));
  c.patch_reconstruct = 1;
  if (const char* e = getenv("GZ_PATCH_RECON")) { const int v = atoi(e); if (v >= 0 && v <= 2) c.patch_reconstruct = v; }
  c.opsin_ahead = 1;
  if (const char* e = getenv("GZ_OPSIN_AHEAD")) c.opsin_ahead = atoi(e) != 0;
  *out = c;
  return GZ_OK;
}
int gz_get_config(const gz_ctx* c, gz_config* out) {
  if (!c || !out) return GZ_E_ARG;
  *out = c->cfg;
  return GZ_OK;
}
int gz_set_config(gz_ctx* c, const gz_config* in) {
  if (!c || !in || in->struct_size != (int)sizeof(gz_config)) return GZ_E_ARG;
  if (in->blur_packed < -1 || in->blur_packed > 1 || in->single_stream < -1 || in->single_stream > 1 || (in->tile_rows != 0 && in->tile_rows != 16 && in->tile_rows != 32) ||
      in->malta_pad_bytes < 0 || in->malta_pad_bytes > (64 << 10) || in->patch_reconstruct < 0 || in->patch_reconstruct > 2)
    return GZ_E_ARG;
  if (c->compare_pending || c->scan_pending || c->order_pending || c->desc_pending) {
    c->err = "gz_set_config while work of the context is in flight";
    return GZ_E_STATE;
  }
  c->cfg = *in;
  c->lin_is_cand = c->xyb_is_cand = false;
  return GZ_OK;
}

static std::atomic<int>& images_in_flight_hint() { static std::atomic<int> v{0}; return v; }
void gz_hint_images_in_flight(int n) {
  images_in_flight_hint().store(n, std::memory_order_relaxed);
  if (n > 1) {
    // company is coming: idle stream sets with a priority main stream go (a priority stream is one more hardware
    // queue for the runtime to multiplex, even while nobody uses it)
    StreamSetPool& sp = stream_set_pool();
    std::lock_guard<std::mutex> lk(sp.mu);
    for (auto it = sp.sets.begin(); it != sp.sets.end();) {
      if (it->first & 1) {
        (void)hipStreamDestroy(it->second.own);
        (void)hipStreamDestroy(it->second.side);
        (void)hipStreamDestroy(it->second.side2);
        (void)hipStreamDestroy(it->second.entropy);
        it = sp.sets.erase(it);
      } else {
        ++it;
      }
    }
  }
}

int gz_device_pci_bus_id(int device, char* out, int cap) {
  if (!out || cap < 16) return GZ_E_ARG;
#ifdef GZ_EMU
  (void)device;
  snprintf(out, (size_t)cap, "0000:00:00.0");
  return GZ_OK;
#else
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { (void)hipGetLastError(); return GZ_E_NO_DEVICE; }
  if (hipDeviceGetPCIBusId(out, cap, device) != hipSuccess) { (void)hipGetLastError(); return GZ_E_HIP; }
  return GZ_OK;
#endif
}

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
  {
    StreamSetPool& sp = stream_set_pool();
    std::lock_guard<std::mutex> lk(sp.mu);
    for (auto& kv : sp.sets) {
      (void)hipStreamDestroy(kv.second.own);
      (void)hipStreamDestroy(kv.second.side);
      (void)hipStreamDestroy(kv.second.side2);
      (void)hipStreamDestroy(kv.second.entropy);
    }
    sp.sets.clear();
  }
  HandlePool& hp = handle_pool();
  std::lock_guard<std::mutex> lk(hp.mu);
  for (auto& kv : hp.events) (void)hipEventDestroy(kv.second);
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
  (void)gz_config_from_environment(&c->cfg);
  set_frame(c, 1);
  auto fail = [&](int code) { *err = code; gz_destroy(c); return (gz_ctx*)nullptr; };
#define CHK0(call) do { const hipError_t e0_ = (call); if (e0_ != hipSuccess) { (void)hipGetLastError(); return fail(e0_ == hipErrorOutOfMemory ? GZ_E_NOMEM : GZ_E_HIP); } } while (0)
  // The chain's main stream takes the device's highest priority, so that the dispatcher serves
  // its workgroups before those of the entropy coder that runs beside it (1080p encode 0.144 ->
  // 0.140 s) -- but only for a context that has the device to itself when it is created, and no
  // stream ever goes BELOW the default: with several images in flight priorities invert (an
  // image's low-priority entropy coder starves behind the other images' chains while its host
  // thread waits for it: 16 x 1080p, 8 in flight, 21.8 -> 7.5-13.5 MPix/s with main = highest and
  // entropy = lowest on every context; profiles/r03_stream_priorities.log).
  c->prio_streams = live_contexts(device, +1) == 0 && images_in_flight_hint().load(std::memory_order_relaxed) <= 1;
  c->counted_live = true;
  {
    const CuPlan& cp = cu_plan();
    if (cp.parts > 0) {
      c->cu_slot = cu_slot_take(device);
      c->cu_class = 1 + c->cu_slot % cp.parts;
    } else if (cp.main_lo >= 0 || cp.side_lo >= 0) {
      c->cu_class = 1;
    }
    StreamSet ss;
    // The contexts alive on a device take slots (lowest free first) and slot s gets a stream set whose MAIN stream
    // sits on hardware queue s mod 4: four images in flight then have their main streams on four different queues
    // whatever order the images before them finished in (handed out by availability, two mains could share a queue:
    // the "40 or 44-46 MPix/s" of a 4K batch from process to process; 16 x 1080p 35.4 -> 39.7 MPix/s, 12 x 1440p 38.9 ->
    // 40.2, 8 x 4K 42.4 -> 43.6, nothing at 1 MPix and below: r06_chain_experiments.log, section 12).  GZ_SET_SLOT=0: as before.
    static const bool set_slot = [] { const char* e = getenv("GZ_SET_SLOT"); return e ? atoi(e) != 0 : true; }();
    if (set_slot && c->cu_slot < 0) c->cu_slot = cu_slot_take(device);
    const hipError_t se = pool_stream_set_create(&ss, c->prio_streams, c->cu_class, set_slot ? c->cu_slot % 4 : -1);
    c->set_rot = ss.rot;
    c->own_stream = ss.own; c->side_stream = ss.side; c->side_stream2 = ss.side2; c->entropy_stream = ss.entropy;
    c->stream = c->own_stream;
    CHK0(se);
  }
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
  {
    const int rc = gz_set_rgb(c, rgb);
    if (rc != GZ_OK) return fail(rc);
  }
#undef CHK0
  return c;
}

int gz_set_rgb(gz_ctx* c, const uint8_t* rgb) {
  DeviceScope ds_(c);
  if (!c || !rgb) return GZ_E_ARG;
  HIPCHK(c, hipMemcpyAsync(c->d_rgb, rgb, (size_t)3 * c->w * c->h, hipMemcpyHostToDevice, c->stream));
  // pi0_ = SeparateFrequencies(OpsinDynamicsImage(LinearRgb(rgb)))
  c->lin_is_cand = c->xyb_is_cand = false;   // (lin[] takes the original)
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
  (void)pool_free(c->d_out_cnt); (void)pool_free(c->d_out_idx); (void)pool_free(c->d_out_err); (void)pool_free(c->d_csr_off);
  if (c->h_step_delta) (void)pool_host_free(c->h_step_delta);
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
  if (c->side_stream) (void)hipStreamSynchronize(c->side_stream);
  if (c->side_stream2) (void)hipStreamSynchronize(c->side_stream2);
  if (c->entropy_stream) (void)hipStreamSynchronize(c->entropy_stream);
  pool_event_destroy(c->ev_candidate);
  if (c->ev_steps) pool_event_destroy(c->ev_steps);
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
  {   // the four streams go back as the set they were made as (own_stream: synchronised at the top of gz_destroy)
    StreamSet ss;
    ss.own = c->own_stream; ss.side = c->side_stream; ss.side2 = c->side_stream2; ss.entropy = c->entropy_stream;
    ss.rot = c->set_rot;
    pool_stream_set_destroy(ss, c->prio_streams, c->cu_class);
  }
  if (c->cu_slot >= 0) cu_slot_release(c->device, c->cu_slot);
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


}  // extern "C"
