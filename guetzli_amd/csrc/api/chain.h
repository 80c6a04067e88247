// The Compare chain: blur dispatch (which instantiation for which size), the pipeline stages (opsin, SeparateFrequencies, mask branch, Malta, combine, final blur) on three streams, reconstruction, the block mask, candidate ranking on the host.
// (part of the one translation unit gz_api.hip, which includes these files in order; split by
// concern in round 5 -- no declaration here is visible outside libguetzli_amd.so but the C ABI)
#pragma once

namespace {
// ------------------------------------------------------------- blur dispatch helpers --
// Radius < 16: one fused launch per blur (k_blur2d); radius >= 16: a row pass and a column pass.
// Two measured crossovers pick the instantiation (both can be forced through the context's gz_config --
// from GZ_BLUR_PK / GZ_TILE_ROWS at gz_create -- so that the tests run every one of them on images small enough
// for the emulation):
//  * blur_packed -- row-pair / column-pair passes with packed arithmetic (k_blur_h_pk, k_blur_v_pk:
//    twice the outputs per thread, half the LDS reads and address computations per output, half
//    the workgroups) from 1.5 MPix on (round 5; 4 MPix until then), the one-output-row kernels (k_blur_h, k_blur_v_compact) below
//    (profiles/r02_packed_blur_ab.log, r03_chain_kernel_experiments.log);
//  * tile_rows -- 64 x 32 tiles from 7 MPix on, 64 x 16 below: twice the workgroups for
//    256 CUs (720p: 0.290 -> 0.257 ms per Compare; round 5: 1080p 0.343 -> 0.330, equal at 4K).
// What round 4 removed after it had lost every A/B of rounds 2 and 3: the unrolled (non-compact)
// column pass and fused kernels, 64-row tiles, the epilogue without 16-byte accesses and the row
// pass with LDS bank conflicts (GZ_BLUR_OPT), the three-plane LF passes, the unpaired mask blurs.
// cfg.side_small (GZ_SIDE_SMALL, experiment, round 6): the kernels of the two side branches (SameNoise blur, mask
// blurs) in the forms whose workgroups fit into the hole ONE retiring Malta workgroup leaves on a CU (<= 64 VGPRs,
// <= 19.7 KB of LDS): unpaired row / column passes, 16-row tiles -- c->side_small is set while those launches are made.
static bool packed_blur(const gz_ctx* c) {
  if (c->side_small) return false;
  if (c->cfg.blur_packed >= 0) return c->cfg.blur_packed != 0;
  // (round 5, with 16-row tiles: 1080p 0.3356 -> 0.3290 ms, 2560 x 1440 0.487 -> 0.466, 3200 x 1800 0.719 -> 0.684;
  // 720p equal -- profiles/r05_chain_experiments.log, section 8)
  return (size_t)c->w * c->h >= 1500000;
}
constexpr int kTileRows = 32;
constexpr int kSmallTileRows = 16;
static bool small_tiles(const gz_ctx* c) {
  if (c->side_small) return true;
  if (c->cfg.tile_rows == 16) return true;
  if (c->cfg.tile_rows == 32) return false;
  // (round 5: 16-row tiles win or tie up to 3200 x 1800 -- 1080p chain 0.343 -> 0.330 ms, sixteen 1080p images
  // four in flight 27.7 -> 29.5 MPix/s; equal at 3840 x 2160: profiles/r05_chain_experiments.log, section 8)
  return (size_t)c->w * c->h < 7000000;
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
// bm.tiles + n_tiles: the listed tiles only (tile ids of the grid this function would launch).
template <int R, int NC, class Src, class Post, bool BM = false>
int blur2d(gz_ctx* c, const SrcPack<Src, NC>& src, const Post& post, const BlurCfg& cfg,
           BlockMaxOut bm = BlockMaxOut{nullptr, nullptr, 0, nullptr}, int n_tiles = 0) {
  if (cfg.r != R) { c->err = "blur radius mismatch"; return GZ_E_STATE; }
  const Taps<R> tp = taps_of<R>(cfg);
  const BorderScale bx = cfg.bx, by = cfg.by;
  const int w = c->w, h = c->h, pitch = c->pitch;
  // (the chain's last blur, with block maxima: 16-row tiles below 1.5 MPix only -- 720p chain 0.224 -> 0.213 ms,
  // nothing at 1080p, +3 % at 2560 x 1440: profiles/r05_chain_experiments.log, section 8)
  // (a tile list: always the 16-row tiles -- a handful of workgroups, whose time is one workgroup's latency)
  if (bm.tiles || (BM ? small_tiles(c) && (size_t)c->w * c->h < 1500000 : small_tiles(c))) {
    dim3 grid(gz_div_up(c->w, T2), gz_div_up(c->h, kSmallTileRows));
    if (bm.tiles) grid = dim3(n_tiles);
    GZ_LAUNCH((k_blur2d<R, NC, Src, Post, BM, kSmallTileRows>), grid, dim3(256), c->stream, src, post, w,
              h, pitch, tp, bx, by, bm);
  } else {
    dim3 grid(gz_div_up(c->w, T2), gz_div_up(c->h, kTileRows));
    if (bm.tiles) grid = dim3(n_tiles);
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
// tiles (optional): only the listed tiles of the launch's grid (opsin_tile_rows() high), n_tiles of them.
int stage_opsin(gz_ctx* c, const int* tiles = nullptr, int n_tiles = 0) {
  SrcPack<SrcPlain, 3> s;
  for (int i = 0; i < 3; ++i) s.s[i].p = c->lin[i];
  PostOpsin post;
  for (int i = 0; i < 3; ++i) { post.lin[i] = c->lin[i]; post.xyb[i] = c->xyb[i]; }
  TRY((blur2d<2, 3, SrcPlain, PostOpsin>(c, s, post, c->blur[B_OPSIN], BlockMaxOut{nullptr, nullptr, 0, tiles}, n_tiles)));
  return GZ_OK;
}
static int opsin_tile_rows(const gz_ctx*) { return kSmallTileRows; }   // (of a tile LIST: blur2d)

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
// cfg.single_stream: 1 = the whole Compare on the main stream (per-kernel profiling; no fork / join events),
// 0 = the three-stream chain, -1 (default) = by company: three streams for a context that has the device to itself,
// ONE when other contexts are alive on it (a batch's images in flight).  The side streams buy a lone chain its
// overlap (1080p 0.326 against 0.371 ms); several images in flight overlap each other instead, and every fork /
// join costs host time in a runtime four threads are calling into: one stream per image is +65 % at 512 x 512,
// +12-25 % at 1 MPix, +10 % at 1080p, +4 % at 4K (profiles/r06_chain_experiments.log, section 6).  The choice is
// made per Compare and changes no result: every Compare joins its side streams before it ends.
static bool single_stream_wanted(const gz_ctx* c) {
  if (c->cfg.single_stream >= 0) return c->cfg.single_stream != 0;
  return live_contexts(c->device, 0) > 1;
}
// ... decided ONCE per Compare (choose_streams, at the top of every function that enqueues a diffmap stage) and kept in
// the context for its stages: contexts come and go on other threads while a Compare is being enqueued, and a fork
// made for three streams must not meet a join that thinks there was one.
static void choose_streams(gz_ctx* c) { c->single_now = single_stream_wanted(c); }
static bool single_stream(const gz_ctx* c) { return c->single_now; }
int fork_side_branch(gz_ctx* c, const Psycho& p0, const Psycho& p1) {
  if (single_stream(c)) {
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
  c->side_small = c->cfg.side_small != 0;
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
  c->side_small = false;
  c->stream = main_stream;
  TRY(rc);
  HIPCHK(c, hipEventRecord(c->ev_join, c->side_stream));
  HIPCHK(c, hipEventRecord(c->ev_join2, c->side_stream2));
  return GZ_OK;
}
int join_mask_branch(gz_ctx* c) {
  if (single_stream(c)) return GZ_OK;
  HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
  HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join2, 0));
  return GZ_OK;
}

// DiffmapPsychoImage (butteraugli.cc:817-908) + score: p0 = original, p1 = candidate.
int stage_diffmap(gz_ctx* c, const Psycho& p0, const Psycho& p1, bool want_block_max,
                  bool max_cleared = false, bool want_distmap = true, bool streams_chosen = false,
                  bool clear_in_combine = false) {
  if (!streams_chosen) choose_streams(c);
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
#ifndef GZ_EMU
  // cfg.malta_pad_bytes (GZ_MALTA_PAD) = bytes of (unused) dynamic LDS per workgroup (experiment, round 6): 19.7 KB + pad bounds the
  // workgroups a CU holds (8 without; 3 KB: 7, 7 KB: 6, 13 KB: 4) and leaves wave slots and registers to others
  const int malta_pad = c->cfg.malta_pad_bytes;
  if (malta_pad > 0) {
    hipLaunchKernelGGL((k_malta_rolled<3>), mgrid, dim3(256), (size_t)malta_pad, c->stream, ay, ax, c->w, c->h, c->pitch);
  } else
#endif
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
    a.clear_word = clear_in_combine ? c->d_max_bits : nullptr;
    dim3 grid(gz_div_up(c->w, 1024), c->h);   // (4 pixels per thread)
    GZ_LAUNCH(k_combine, grid, dim3(256), c->stream, a, c->w, c->h, c->pitch);
    KCHK(c);
  }
  {  // CalculateDiffmap second half: blur(sigma 1.725, border_ratio 1.0) + mix
    SrcPack<SrcPlain, 1> s; s.s[0].p = c->dsq;
    PostDiffmapMix post; post.d = c->dsq; post.out = want_distmap ? c->distmap : nullptr;
    if (!max_cleared && !clear_in_combine) HIPCHK(c, hipMemsetAsync(c->d_max_bits, 0, sizeof(unsigned), c->stream));
    BlockMaxOut bm{want_block_max ? c->d_block_max : nullptr, c->d_max_bits, c->bw, nullptr};
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
// want_distmap = false (the search loop, gz_time_compare): the last kernel leaves the per-block maxima and the
// image maximum only; c->distmap then holds no distance map (have_distmap_plane).
// The candidate's linear planes: reconstructed here, unless every change of the candidate since the last full
// reconstruction was patched into them by the call that made it (c->lin_is_cand; the search loop's steady state:
// a fifth of a 4K image's blocks change per iteration).  force_full: gz_time_compare, whose repetitions change
// nothing and must not measure a chain without its first kernel.
static std::atomic<unsigned long long> g_compares{0}, g_compares_patched{0}, g_patch_checks{0};   // gz_compare_counters
static std::atomic<unsigned long long> g_compares_ahead{0}, g_ahead_checks{0};
static int check_patched_planes(gz_ctx* c) {   // (gz_config.patch_reconstruct == 2)
  float* full = nullptr;
  unsigned* d_bad = nullptr;
  HIPCHK(c, pool_malloc((void**)&full, sizeof(float) * c->plane * 3));
  HIPCHK(c, pool_malloc((void**)&d_bad, sizeof(unsigned)));
  int rc = GZ_OK;
  unsigned bad = 0;
  do {
    if (hipMemsetAsync(d_bad, 0, sizeof(unsigned), c->stream) != hipSuccess) { rc = GZ_E_HIP; break; }
    const int strips = gz_div_up(c->bw, kReconBlocks);
    GZ_LAUNCH(k_reconstruct, dim3(c->bh * strips), dim3(256), c->stream, (const int16_t*)c->d_cand, c->w, c->h, c->bw, c->nb,
              c->pitch, c->plane, (const float*)c->d_srgb_lut, full, (uint8_t*)nullptr, (unsigned*)nullptr, 1);
    GZ_LAUNCH(k_count_differing_words, dim3(1024), dim3(256), c->stream, (const unsigned*)c->lin[0], (const unsigned*)full,
              (size_t)c->plane * 3, d_bad);
    if (hipMemcpyAsync(&bad, d_bad, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) { rc = GZ_E_HIP; break; }
  } while (0);
  (void)pool_free(full);
  (void)pool_free(d_bad);
  TRY(rc);
  ++g_patch_checks;
  if (bad) { c->err = "patched linear planes differ from a full reconstruction"; return GZ_E_STATE; }
  return GZ_OK;
}

// (gz_config.patch_reconstruct == 2) xyb[] as the calls kept it against the opsin blur of lin[] as a whole.
static int check_opsin_ahead(gz_ctx* c) {
  float* kept = nullptr;
  unsigned* d_bad = nullptr;
  HIPCHK(c, pool_malloc((void**)&kept, sizeof(float) * c->plane * 3));
  HIPCHK(c, pool_malloc((void**)&d_bad, sizeof(unsigned)));
  int rc = GZ_OK;
  unsigned bad = 0;
  do {
    if (hipMemsetAsync(d_bad, 0, sizeof(unsigned), c->stream) != hipSuccess) { rc = GZ_E_HIP; break; }
    for (int i = 0; i < 3 && rc == GZ_OK; ++i)
      if (hipMemcpyAsync(kept + (size_t)i * c->plane, c->xyb[i], sizeof(float) * c->plane, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) rc = GZ_E_HIP;
    if (rc != GZ_OK) break;
    rc = stage_opsin(c);
    if (rc != GZ_OK) break;
    for (int i = 0; i < 3; ++i)
      GZ_LAUNCH(k_count_differing_words, dim3(1024), dim3(256), c->stream, (const unsigned*)c->xyb[i],
                (const unsigned*)(kept + (size_t)i * c->plane), (size_t)c->plane, d_bad);
    if (hipMemcpyAsync(&bad, d_bad, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) { rc = GZ_E_HIP; break; }
  } while (0);
  (void)pool_free(kept);
  (void)pool_free(d_bad);
  TRY(rc);
  ++g_ahead_checks;
  if (bad) { c->err = "the opsin image kept ahead differs from the opsin blur of the whole planes"; return GZ_E_STATE; }
  return GZ_OK;
}

int enqueue_compare(gz_ctx* c, bool want_block_max, bool want_distmap = false, bool force_full = false) {
  want_distmap = want_distmap || c->cfg.store_distmap != 0;   // (1: the chain as it was until round 5, A/B)
  choose_streams(c);
  ++g_compares;
  const bool patched = !force_full && c->cfg.patch_reconstruct != 0 && c->lin_is_cand && c->cfac == 1;
  if (patched) {
    if (c->cfg.patch_reconstruct == 2) TRY(check_patched_planes(c));
    ++g_compares_patched;
  } else {
    TRY(stage_reconstruct(c, c->d_cand, c->lin[0], nullptr, c->d_max_bits));
    c->lin_is_cand = c->cfac == 1 && c->cfg.patch_reconstruct != 0 && (c->nb >= 8192 || c->cfg.patch_reconstruct == 2);
  }
  // ... and their opsin image, when the same calls have kept that current too (xyb_is_cand; consumed here)
  const bool ahead = patched && c->xyb_is_cand;
  c->xyb_is_cand = false;
  if (ahead) {
    if (c->cfg.patch_reconstruct == 2) TRY(check_opsin_ahead(c));
    ++g_compares_ahead;
  } else {
    TRY(stage_opsin(c));
  }
  TRY(stage_separate(c, &c->pi1, !single_stream(c)));
  TRY(stage_diffmap(c, c->pi0, c->pi1, want_block_max, !patched, want_distmap, true, patched));
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
  c->lin_is_cand = c->xyb_is_cand = false;   // (lin[] takes the original)
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
