// C ABI: phase A -- candidate ranking, the per-block zeroing search for every component mask and frame, CompareBlock for independent blocks (the Comparator seam).
// (part of the one translation unit gz_api.hip, which includes these files in order; split by
// concern in round 5 -- no declaration here is visible outside libguetzli_amd.so but the C ABI)
#pragma once

extern "C" {


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
  TRY(flush_order_advance(c));   // (an update of max_block_error that is still due belongs to the old grid)
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
    HIPCHK(c, pool_malloc((void**)&c->d_csr_off, sizeof(int32_t) * ((size_t)nb + 1)));
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
  c->search_total = 0;   // set below, once the offsets are on the host
  // The CSR arrays are made on the device (offsets = scan of the counts, indices packed into the
  // ranked lists' buffer, which the search is done with): the host copies what the caller takes.
  GZ_LAUNCH(k_csr_offsets, dim3(1), dim3(1024), c->stream, (const int32_t*)c->d_out_cnt, gn, c->d_csr_off);
  KCHK(c);
  GZ_LAUNCH(k_csr_pack, dim3(gz_div_up(gn, 4)), dim3(256), c->stream, (const int32_t*)c->d_out_cnt,
            (const int32_t*)c->d_csr_off, (const uint8_t*)c->d_out_idx, gn, c->d_rank_idx);
  KCHK(c);
  std::vector<int32_t> rcnt(gn);
  HIPCHK(c, hipMemcpyAsync(rcnt.data(), c->d_rank_cnt, sizeof(int32_t) * gn, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(offsets, c->d_csr_off, sizeof(int32_t) * ((size_t)gn + 1), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const long total = offsets[gn];
  c->search_total = (size_t)total;
  // evaluations: step s of a block with n candidates compares min(lookahead, n - s) of them (summed in
  // closed form), on every 8x8 block of its area that lies inside the image
  c->search_evaluations = 0;
  for (int b = 0; b < gn; ++b) {
    const unsigned long long n = (unsigned long long)rcnt[b], la = (unsigned long long)lookahead;
    const unsigned long long e = n >= la ? la * (n - la + 1) + la * (la - 1) / 2 : n * (n + 1) / 2;
    int sub = 1;
    if (mode == 2) {
      const int bx = b % c->cbw, by = b / c->cbw;
      sub = ((16 * bx + 8 < c->w) ? 2 : 1) * ((16 * by + 8 < c->h) ? 2 : 1);
    }
    c->search_evaluations += e * sub;
  }
  if (total > cap) { c->err = "candidate capacity too small, need " + std::to_string(total); return GZ_E_ARG; }
  if (total > 0)
    HIPCHK(c, hipMemcpyAsync(idx, c->d_rank_idx, (size_t)total, hipMemcpyDeviceToHost, c->stream));
  if (err) {   // (tests and the stand-alone ABI; the encoder leaves the errors on the device)
    std::vector<float> werr((size_t)gn * 192);
    HIPCHK(c, hipMemcpyAsync(werr.data(), c->d_out_err, sizeof(float) * gn * 192, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int b = 0; b < gn; ++b)
      memcpy(err + offsets[b], werr.data() + (size_t)b * 192, sizeof(float) * (size_t)(offsets[b + 1] - offsets[b]));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
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


int gz_compare_counters(uint64_t out[5]) {
  if (!out) return GZ_E_ARG;
  out[0] = g_compares_patched.load();
  out[1] = g_patch_checks.load();
  out[2] = g_compares.load();
  out[3] = g_compares_ahead.load();
  out[4] = g_ahead_checks.load();
  return GZ_OK;
}

int gz_search_evaluations(gz_ctx* c, uint64_t* evaluations) {
  DeviceScope ds_(c);
  if (!c || !evaluations) return GZ_E_ARG;
  *evaluations = c->search_evaluations;
  return GZ_OK;
}

}  // extern "C"
