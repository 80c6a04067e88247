// C ABI: the device JPEG entropy coder (histograms, scan, kept best scan).
// (part of the one translation unit gz_api.hip, which includes these files in order; split by
// concern in round 5 -- no declaration here is visible outside libguetzli_amd.so but the C ABI)
#pragma once

extern "C" {


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
  {
    const int upm = ncomp == 1 ? 1 : (c->cfac == 2 ? 6 : 3);   // blocks per MCU
    static const bool generic = getenv("GZ_HIST_GENERIC") != nullptr;  // (the tests: the run-time-geometry kernel; read once)
    if (generic)
      GZ_LAUNCH(k_jpeg_histograms, dim3(grid), dim3(64 * kHistWaves), c->stream, (const int16_t*)c->d_cand,
                (const int*)c->d_jq, geom, c->d_hist);
    else if (upm == 3)
      GZ_LAUNCH((k_jpeg_histograms_t<3>), dim3(grid), dim3(64 * kHistWaves), c->stream, (const int16_t*)c->d_cand,
                (const int*)c->d_jq, geom, c->d_hist);
    else if (upm == 6)
      GZ_LAUNCH((k_jpeg_histograms_t<6>), dim3(grid), dim3(64 * kHistWaves), c->stream, (const int16_t*)c->d_cand,
                (const int*)c->d_jq, geom, c->d_hist);
    else
      GZ_LAUNCH((k_jpeg_histograms_t<1>), dim3(grid), dim3(64 * kHistWaves), c->stream, (const int16_t*)c->d_cand,
                (const int*)c->d_jq, geom, c->d_hist);
  }
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

}  // extern "C"
