// C ABI: gz_probe_* -- stage probes for the tests (localise a divergence to one stage of the chain; the device's IEEE arithmetic and std::sort restatements).  Diagnostics, not part of any encode: -DGZ_NO_PROBES leaves them out of a deployment build.
// (part of the one translation unit gz_api.hip, which includes these files in order; split by
// concern in round 5 -- no declaration here is visible outside libguetzli_amd.so but the C ABI)
#pragma once

extern "C" {

#ifndef GZ_NO_PROBES

// ------------------------------------------------------------------- stage probes -----
int gz_probe_blur(gz_ctx* c, const float* in, float sigma, float border_ratio, float* out) {
  DeviceScope ds_(c);
  if (!c || !in || !out) return GZ_E_ARG;
  BlurCfg cfg;
  TRY(setup_blur_cfg(c, &cfg, sigma, border_ratio));
  float* src = c->xyb[0];
  c->xyb_is_cand = false;   // (xyb[] as scratch)
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
  c->lin_is_cand = c->xyb_is_cand = false;
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
  c->xyb_is_cand = false;
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
  c->lin_is_cand = c->xyb_is_cand = false;
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

#endif  // GZ_NO_PROBES

}  // extern "C"
