// C ABI: the block path (gz_encode_rgb, gz_quantize, coefficient access, gz_reconstruct) and ButteraugliComparator::Compare (gz_compare*, gz_time_compare, gz_block_weights*).
// (part of the one translation unit gz_api.hip, which includes these files in order; split by
// concern in round 5 -- no declaration here is visible outside libguetzli_amd.so but the C ABI)
#pragma once

extern "C" {

int gz_encode_rgb(gz_ctx* c, int16_t* coeffs_out) {
  DeviceScope ds_(c);
  if (!c) return GZ_E_ARG;
  if (c->cfac != 1) {   // back to 4:4:4: the candidate and the search belonged to the other frame
    set_frame(c, 1);
    c->have_cand = false;
    c->lin_is_cand = c->xyb_is_cand = false;
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
    c->lin_is_cand = c->xyb_is_cand = false;
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
  c->lin_is_cand = c->xyb_is_cand = false;
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
  c->lin_is_cand = c->xyb_is_cand = false;
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
  c->lin_is_cand = c->xyb_is_cand = false;
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
  if (linear) c->lin_is_cand = c->xyb_is_cand = false;   // (rewritten as a whole: nothing to keep track of)
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
  TRY(enqueue_compare(c, true, distmap != nullptr));   // (the map is stored only for a caller that takes it)
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
  for (int i = 0; i < iters; ++i) TRY(enqueue_compare(c, true, false, true));   // (always the whole chain, as gz_time_compare)
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
  for (int i = 0; i < iters; ++i) TRY(enqueue_compare(c, true, false, true));   // (always the whole chain)
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


}  // extern "C"
