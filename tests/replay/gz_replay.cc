// TEST INFRASTRUCTURE ONLY -- record / replay shim of the C ABI (include/guetzli_amd.h)
// for the entry points the host search driver (guetzli_amd/host) calls.
//
// Purpose: the host driver's decisions (quant-matrix bisection, global coefficient order,
// entropy-size model, JPEG bytes) can be exercised at REAL image sizes on a machine
// without a GPU.  On the GPU box the shim runs in RECORD mode: every call is forwarded to
// the real gfx950 library (dlopen) and its device-computed results are appended to a log.
// In REPLAY mode the log stands in for the device: the host driver linked against this
// shim must reproduce the reference's JPEG byte for byte from it.  Nothing under
// guetzli_amd/ links or loads this file; it is never a fallback of the product.
//
//   GZ_REPLAY_MODE = record | replay
//   GZ_REPLAY_FILE = log path
//   GZ_REPLAY_REAL = path of the real libguetzli_amd.so (record mode)
//
// Replayed results: original coefficients (gz_encode_rgb), distance + per-block maxima of
// every gz_compare, the CSR arrays of gz_block_zeroing_orders.  Integer work that is a
// pure function of replayed data (gz_quantize) is recomputed with the reference formula
// (quantize.h:24-29); gz_block_weights runs the product's own header.
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <utility>
#include <algorithm>
#include <vector>

#include "../../guetzli_amd/csrc/gz_host_weights.h"
#include "../../include/guetzli_amd.h"

namespace {

struct Real {
  void* h = nullptr;
  decltype(&gz_create) create;
  decltype(&gz_destroy) destroy;
  decltype(&gz_encode_rgb) encode_rgb;
  decltype(&gz_quantize) quantize;
  decltype(&gz_compare) compare;
  decltype(&gz_compare_begin) compare_begin;
  decltype(&gz_compare_end) compare_end;
  decltype(&gz_block_weights) block_weights;
  decltype(&gz_block_zeroing_orders) block_zeroing_orders;
  decltype(&gz_set_coeff_blocks) set_coeff_blocks;
  decltype(&gz_get_coeffs) get_coeffs;
  decltype(&gz_jpeg_histograms) jpeg_histograms;
  decltype(&gz_jpeg_scan) jpeg_scan;
  decltype(&gz_jpeg_scan_keep) jpeg_scan_keep;
  decltype(&gz_jpeg_scan_bytes) jpeg_scan_bytes;
  decltype(&gz_order_build) order_build;
  decltype(&gz_order_reset) order_reset;
  decltype(&gz_order_build_auto) order_build_auto;
  decltype(&gz_order_advance) order_advance;
  decltype(&gz_apply_coeff_edits) apply_coeff_edits;
  decltype(&gz_apply_candidate_steps) apply_candidate_steps;
  decltype(&gz_steps_histogram_delta) steps_histogram_delta;
  decltype(&gz_set_rgb) set_rgb;
  decltype(&gz_set_orig_coeffs) set_orig_coeffs;
  decltype(&gz_reconstruct) reconstruct;
  decltype(&gz_encode_rgb_only) encode_rgb_only;
  decltype(&gz_order_partition) order_partition;
  decltype(&gz_order_fetch) order_fetch;
  decltype(&gz_strerror) strerror_;
  decltype(&gz_last_error) last_error;
};

bool recording() {
  const char* m = getenv("GZ_REPLAY_MODE");
  return m && strcmp(m, "record") == 0;
}

Real* real() {
  static Real r;
  if (r.h) return &r;
  const char* p = getenv("GZ_REPLAY_REAL");
  r.h = dlopen(p ? p : "libguetzli_amd.so", RTLD_NOW | RTLD_LOCAL);
  if (!r.h) {
    fprintf(stderr, "gz_replay: cannot open real library: %s\n", dlerror());
    abort();
  }
#define SYM(field, name) r.field = (decltype(r.field))dlsym(r.h, name); if (!r.field) abort();
  SYM(create, "gz_create") SYM(destroy, "gz_destroy") SYM(encode_rgb, "gz_encode_rgb")
  SYM(quantize, "gz_quantize") SYM(compare, "gz_compare")
  SYM(compare_begin, "gz_compare_begin") SYM(compare_end, "gz_compare_end") SYM(block_weights, "gz_block_weights")
  SYM(block_zeroing_orders, "gz_block_zeroing_orders")
  SYM(set_coeff_blocks, "gz_set_coeff_blocks") SYM(strerror_, "gz_strerror")
  SYM(last_error, "gz_last_error") SYM(get_coeffs, "gz_get_coeffs")
  SYM(jpeg_histograms, "gz_jpeg_histograms") SYM(jpeg_scan, "gz_jpeg_scan")
  SYM(jpeg_scan_keep, "gz_jpeg_scan_keep") SYM(jpeg_scan_bytes, "gz_jpeg_scan_bytes")
  SYM(order_build, "gz_order_build") SYM(order_partition, "gz_order_partition")
  SYM(order_fetch, "gz_order_fetch") SYM(order_reset, "gz_order_reset")
  SYM(order_build_auto, "gz_order_build_auto") SYM(order_advance, "gz_order_advance")
  SYM(apply_coeff_edits, "gz_apply_coeff_edits")
  SYM(apply_candidate_steps, "gz_apply_candidate_steps") SYM(steps_histogram_delta, "gz_steps_histogram_delta")
  SYM(set_rgb, "gz_set_rgb") SYM(set_orig_coeffs, "gz_set_orig_coeffs")
  SYM(reconstruct, "gz_reconstruct") SYM(encode_rgb_only, "gz_encode_rgb_only")
#undef SYM
  return &r;
}

enum Tag : int32_t { T_CREATE = 1, T_ORIG = 2, T_COMPARE = 3, T_ORDERS = 4, T_HISTO = 5, T_SCAN = 6,
                     T_BYTES = 7, T_DELTA = 8 };

}  // namespace

struct gz_ctx {
  gz_ctx* inner = nullptr;   // record mode: the real context
  FILE* f = nullptr;
  int w = 0, h = 0, bw = 0, bh = 0, nb = 0;
  float target = 0;
  std::vector<int16_t> orig;
  std::vector<float> bmax;
  bool have_bmax = false;
  // replay mode: phase A's CSR arrays and the global candidate order, which are pure
  // functions of replayed data and are recomputed here with the serial algorithms
  std::vector<int32_t> cand_off;
  std::vector<float> cand_err;
  std::vector<std::pair<int, float> > order;
  std::vector<float> max_err, weight;
  std::string err;
  // GZ_REPLAY_DIRTY_LOG: blocks whose coefficients changed since the last evaluation (analysis of
  // an incremental Compare, tools/dirty_tiles.py)
  std::vector<unsigned char> dirty;
  bool all_dirty = true;
  int evals = 0;
  // A log recorded by a driver that entropy-coded EVERY candidate, replayed by one that codes only
  // the candidates that can win (round 4): a scan nobody asked for is set aside when another entry
  // is expected, and handed out if the driver asks for the scan after all.
  bool held_scan = false;
  uint64_t held_scan_bytes = 0, last_scan_bytes = 0;
};

namespace {

void put(gz_ctx* c, const void* p, size_t n) {
  if (fwrite(p, 1, n, c->f) != n) { fprintf(stderr, "gz_replay: write failed\n"); abort(); }
}
void get(gz_ctx* c, void* p, size_t n) {
  if (fread(p, 1, n, c->f) != n) { fprintf(stderr, "gz_replay: log exhausted\n"); abort(); }
}
void put_tag(gz_ctx* c, int32_t t) { put(c, &t, 4); }
void expect_tag(gz_ctx* c, int32_t t) {
  int32_t g = 0;
  get(c, &g, 4);
  static thread_local long seen[16] = {0};
  while (g == 6 /* T_SCAN */ && t != 6) {   // a scan of the recording driver that this one skips
    get(c, &c->held_scan_bytes, 8);
    c->held_scan = true;
    get(c, &g, 4);
  }
  if (g != t) {
    fprintf(stderr, "gz_replay: log out of sync (want %d got %d) after", t, g);
    for (int i = 1; i < 9; ++i) fprintf(stderr, " tag%d x %ld", i, seen[i]);
    fprintf(stderr, "\n");
    abort();
  }
  if (g > 0 && g < 16) ++seen[g];
}

}  // namespace

extern "C" {

int gz_abi_version(void) { return 1; }
const char* gz_strerror(int code) { return recording() ? real()->strerror_(code) : "replay error"; }
const char* gz_last_error(const gz_ctx* c) {
  if (!c) return "";
  return c->inner ? real()->last_error(c->inner) : c->err.c_str();
}

gz_ctx* gz_create(int device, int w, int h, const uint8_t* rgb, float target, int* err) {
  const char* path = getenv("GZ_REPLAY_FILE");
  if (!path) { fprintf(stderr, "gz_replay: GZ_REPLAY_FILE not set\n"); abort(); }
  gz_ctx* c = new gz_ctx;
  c->w = w; c->h = h; c->bw = (w + 7) / 8; c->bh = (h + 7) / 8; c->nb = c->bw * c->bh;
  c->target = target;
  if (err) *err = GZ_OK;
  if (recording()) {
    c->inner = real()->create(device, w, h, rgb, target, err);
    if (!c->inner) { delete c; return nullptr; }
    c->f = fopen(path, "wb");
    if (!c->f) abort();
    put_tag(c, T_CREATE);
    const int32_t hdr[2] = {w, h};
    put(c, hdr, 8);
    put(c, &target, 4);
  } else {
    c->f = fopen(path, "rb");
    if (!c->f) { fprintf(stderr, "gz_replay: cannot open %s\n", path); abort(); }
    expect_tag(c, T_CREATE);
    int32_t hdr[2];
    float t;
    get(c, hdr, 8);
    get(c, &t, 4);
    if (hdr[0] != w || hdr[1] != h || t != target) {
      fprintf(stderr, "gz_replay: log is for %dx%d target %g\n", hdr[0], hdr[1], t);
      abort();
    }
  }
  return c;
}

void gz_destroy(gz_ctx* c) {
  if (!c) return;
  if (c->inner) real()->destroy(c->inner);
  if (c->f) fclose(c->f);
  delete c;
}

int gz_encode_rgb(gz_ctx* c, int16_t* coeffs_out) {
  const size_t n = (size_t)3 * c->nb * 64;
  c->orig.resize(n);
  if (c->inner) {
    const int rc = real()->encode_rgb(c->inner, c->orig.data());
    if (rc != GZ_OK) return rc;
    put_tag(c, T_ORIG);
    put(c, c->orig.data(), n * 2);
  } else {
    expect_tag(c, T_ORIG);
    get(c, c->orig.data(), n * 2);
  }
  if (coeffs_out) memcpy(coeffs_out, c->orig.data(), n * 2);
  return GZ_OK;
}

int gz_quantize(gz_ctx* c, const int* q, int16_t* coeffs_out) {
  c->all_dirty = true;
  if (c->inner) return real()->quantize(c->inner, q, coeffs_out);
  if (!coeffs_out) return GZ_OK;
  const size_t per = (size_t)c->nb * 64;
  for (int ch = 0; ch < 3; ++ch)
    for (size_t i = 0; i < per; ++i) {
      const int quant = q ? q[ch * 64 + (int)(i & 63)] : 1;
      const int raw = c->orig[ch * per + i];
      const int r = raw % quant;
      const int delta = 2 * r > quant ? quant - r : ((-2) * r > quant ? -quant - r : -r);
      coeffs_out[ch * per + i] = (int16_t)(raw + delta);
    }
  return GZ_OK;
}


// ---- dirty-tile statistics (GZ_REPLAY_DIRTY_LOG=path): what fraction of the image a Compare that
// recomputed only what a changed block can reach would have to touch.  A block reaches the pixels
// within HALO of it: opsin blur 2 + LF 16 + MF 8 + HF 4 + mask pre 1 + mask blur 20 (SameNoise: 23)
// + final blur 3 = 54..56 (butteraugli.cc:184-233,489-622,1699-1817).
static void dirty_mark(gz_ctx* c, int block) {
  if (c->dirty.empty()) c->dirty.assign(c->nb, 0);
  if (block >= 0 && block < c->nb) c->dirty[block] = 1;
}
static void dirty_report(gz_ctx* c) {
  const char* path = getenv("GZ_REPLAY_DIRTY_LOG");
  if (!path) return;
  if (c->dirty.empty()) c->dirty.assign(c->nb, 0);
  const int HALO = 56, TW = 64, TH = 32;
  const int tw = (c->w + TW - 1) / TW, th = (c->h + TH - 1) / TH;
  std::vector<unsigned char> tile(tw * th, 0);
  long nd = 0;
  for (int by = 0; by < c->bh; ++by)
    for (int bx = 0; bx < c->bw; ++bx) {
      if (!c->all_dirty && !c->dirty[by * c->bw + bx]) continue;
      ++nd;
      const int x0 = std::max(0, bx * 8 - HALO) / TW, x1 = std::min(c->w - 1, bx * 8 + 7 + HALO) / TW;
      const int y0 = std::max(0, by * 8 - HALO) / TH, y1 = std::min(c->h - 1, by * 8 + 7 + HALO) / TH;
      for (int ty = y0; ty <= y1; ++ty) memset(&tile[ty * tw + x0], 1, x1 - x0 + 1);
    }
  long nt = 0;
  for (unsigned char t : tile) nt += t;
  FILE* f = fopen(path, "a");
  if (f) {
    fprintf(f, "%d %ld %d %ld %d\n", c->evals, nd, c->nb, nt, tw * th);
    fclose(f);
  }
  ++c->evals;
  c->all_dirty = false;
  std::fill(c->dirty.begin(), c->dirty.end(), 0);
}

int gz_compare(gz_ctx* c, float* distance, float* distmap, float* block_max) {
  dirty_report(c);
  if (distmap) { fprintf(stderr, "gz_replay: distmap download is not logged\n"); abort(); }
  c->bmax.resize(c->nb);
  if (c->inner) {
    const int rc = real()->compare(c->inner, distance, nullptr, c->bmax.data());
    if (rc != GZ_OK) return rc;
    put_tag(c, T_COMPARE);
    put(c, distance, 4);
    put(c, c->bmax.data(), sizeof(float) * c->nb);
  } else {
    expect_tag(c, T_COMPARE);
    get(c, distance, 4);
    get(c, c->bmax.data(), sizeof(float) * c->nb);
  }
  c->have_bmax = true;
  if (block_max) memcpy(block_max, c->bmax.data(), sizeof(float) * c->nb);
  return GZ_OK;
}

// The split form: in record mode the evaluation really runs between the two calls; the log
// entry (distance + block maxima, as for gz_compare) is written / read at _end.
int gz_compare_begin(gz_ctx* c) {
  dirty_report(c);
  return c->inner ? real()->compare_begin(c->inner) : GZ_OK;
}
int gz_compare_end(gz_ctx* c, float* distance) {
  c->bmax.resize(c->nb);
  if (c->inner) {
    int rc = real()->compare_end(c->inner, distance);
    if (rc != GZ_OK) return rc;
    // the block maxima for the log (replay builds the global order from them): the evaluation
    // once more in its one-call form, which hands them over -- same candidate, same map
    float again = 0;
    rc = real()->compare(c->inner, &again, nullptr, c->bmax.data());
    if (rc != GZ_OK) return rc;
    if (again != *distance) { fprintf(stderr, "gz_replay: the repeated evaluation differs\n"); abort(); }
    put_tag(c, T_COMPARE);
    put(c, distance, 4);
    put(c, c->bmax.data(), sizeof(float) * c->nb);
  } else {
    expect_tag(c, T_COMPARE);
    get(c, distance, 4);
    get(c, c->bmax.data(), sizeof(float) * c->nb);
  }
  c->have_bmax = true;
  return GZ_OK;
}

int gz_block_weights(gz_ctx* c, int direction, int max_block_dist, double target_mul,
                     int use_distmap, float* block_weight) {
  if (c->inner)
    return real()->block_weights(c->inner, direction, max_block_dist, target_mul, use_distmap,
                                 block_weight);
  std::vector<float> zero;
  const float* bmax = c->bmax.data();
  if (!use_distmap) { zero.assign(c->nb, 0.0f); bmax = zero.data(); }
  gz::block_weights_host(bmax, c->bw, c->bh, c->target, direction, max_block_dist, target_mul,
                         block_weight);
  return GZ_OK;
}

int gz_block_zeroing_orders(gz_ctx* c, int lookahead, int new_model, int32_t* offsets,
                            uint8_t* idx, float* err, int cap) {
  // the driver does not ask for the errors (they stay on the device); the log needs them
  std::vector<float> own;
  if (!err) { own.resize((size_t)std::max(cap, 0)); err = own.data(); }
  if (c->inner) {
    const int rc = real()->block_zeroing_orders(c->inner, lookahead, new_model, offsets, idx, err, cap);
    if (rc != GZ_OK) return rc;
    put_tag(c, T_ORDERS);
    put(c, offsets, sizeof(int32_t) * (c->nb + 1));
    put(c, idx, offsets[c->nb]);
    put(c, err, sizeof(float) * offsets[c->nb]);
    return GZ_OK;
  }
  expect_tag(c, T_ORDERS);
  get(c, offsets, sizeof(int32_t) * (c->nb + 1));
  if (offsets[c->nb] > cap) return GZ_E_ARG;
  get(c, idx, offsets[c->nb]);
  get(c, err, sizeof(float) * offsets[c->nb]);
  c->cand_off.assign(offsets, offsets + c->nb + 1);
  c->cand_err.assign(err, err + offsets[c->nb]);
  return GZ_OK;
}

// The global candidate order: forwarded in record mode (nothing to log: it is a function of
// the logged CSR arrays and the caller's inputs); in replay mode the reference's serial
// construction (processor.cc:636-663) and libstdc++'s serial partition step.
int gz_order_build(gz_ctx* c, int direction, const int32_t* next_cand,
                   const float* max_block_error, const float* block_weight, int count_below,
                   float limit, uint64_t* total, int32_t* blocks_to_change, uint64_t* below) {
  if (c->inner)
    return real()->order_build(c->inner, direction, next_cand, max_block_error, block_weight,
                               count_below, limit, total, blocks_to_change, below);
  c->order.clear();
  int btc = 0;
  uint64_t nbelow = 0;
  for (int b = 0; b < c->nb; ++b) {
    if (block_weight[b] == 0) continue;
    const int at = next_cand[b], off = c->cand_off[b], count = c->cand_off[b + 1] - off;
    const float* errs = &c->cand_err[off];
    if (direction > 0) {
      for (int i = at; i < count; ++i)
        c->order.push_back(std::make_pair(b, (errs[i] - max_block_error[b]) / block_weight[b]));
      btc += at < count ? 1 : 0;
    } else {
      for (int i = at - 1; i >= 0; --i)
        c->order.push_back(std::make_pair(b, (max_block_error[b] - errs[i]) / block_weight[b]));
      btc += at > 0 ? 1 : 0;
    }
  }
  if (count_below)
    for (size_t i = 0; i < c->order.size(); ++i) nbelow += c->order[i].second < limit ? 1 : 0;
  *total = c->order.size();
  *blocks_to_change = btc;
  if (below) *below = nbelow;
  return GZ_OK;
}

int gz_order_reset(gz_ctx* c) {
  if (c->inner) return real()->order_reset(c->inner);
  c->max_err.assign(c->nb, 0.0f);
  return GZ_OK;
}

int gz_order_build_auto(gz_ctx* c, int direction, int max_block_dist, double target_mul,
                        int use_distmap, const int32_t* next_cand, int count_below, float limit,
                        uint64_t* total, int32_t* blocks_to_change, uint64_t* below) {
  if (c->inner)
    return real()->order_build_auto(c->inner, direction, max_block_dist, target_mul, use_distmap,
                                    next_cand, count_below, limit, total, blocks_to_change, below);
  c->weight.assign(c->nb, 0.0f);
  if (c->max_err.empty()) c->max_err.assign(c->nb, 0.0f);
  gz_block_weights(c, direction, max_block_dist, target_mul, use_distmap, c->weight.data());
  return gz_order_build(c, direction, next_cand, c->max_err.data(), c->weight.data(), count_below,
                        limit, total, blocks_to_change, below);
}

int gz_order_advance(gz_ctx* c, float val_threshold, int direction) {
  if (c->inner) return real()->order_advance(c->inner, val_threshold, direction);
  for (int b = 0; b < c->nb; ++b) c->max_err[b] += c->weight[b] * val_threshold * direction;
  return GZ_OK;
}

int gz_apply_coeff_edits(gz_ctx* c, const int32_t* pos, const int16_t* val, int n) {
  for (int i = 0; i < n; ++i) dirty_mark(c, (pos[i] / 64) % c->nb);
  if (c->inner) return real()->apply_coeff_edits(c->inner, pos, val, n);
  for (int i = 0; i < n; ++i)
    if (pos[i] < 0 || pos[i] >= 3 * c->nb * 64) return GZ_E_ARG;
  return GZ_OK;
}

// Entry points of the JPEG-input and small-image paths: forwarded when recording; those
// paths are not logged, so they cannot be replayed.
static int not_logged(const char* what) {
  fprintf(stderr, "gz_replay: %s is not logged\n", what);
  abort();
}
int gz_set_rgb(gz_ctx* c, const uint8_t* rgb) {
  return c->inner ? real()->set_rgb(c->inner, rgb) : not_logged("gz_set_rgb");
}
int gz_set_orig_coeffs(gz_ctx* c, const int16_t* coeffs) {
  return c->inner ? real()->set_orig_coeffs(c->inner, coeffs) : not_logged("gz_set_orig_coeffs");
}
int gz_reconstruct(gz_ctx* c, uint8_t* srgb, float* linear) {
  return c->inner ? real()->reconstruct(c->inner, srgb, linear) : not_logged("gz_reconstruct");
}
int gz_encode_rgb_only(int device, const uint8_t* rgb, int w, int h, int16_t* coeffs_out) {
  return recording() ? real()->encode_rgb_only(device, rgb, w, h, coeffs_out)
                     : not_logged("gz_encode_rgb_only");
}

int gz_apply_candidate_steps(gz_ctx* c, int direction, const int32_t* blocks,
                             const int32_t* counts, int n) {
  for (int i = 0; i < n; ++i) if (counts[i] > 0) dirty_mark(c, blocks[i]);
  if (c->inner) return real()->apply_candidate_steps(c->inner, direction, blocks, counts, n);
  return GZ_OK;   // the host driver keeps its own mirror of the image; nothing to replay
}

int gz_steps_histogram_delta(gz_ctx* c, int32_t* ac_delta) {
  if (c->inner) {
    const int rc = real()->steps_histogram_delta(c->inner, ac_delta);
    if (rc != GZ_OK) return rc;
    put_tag(c, T_DELTA);
    put(c, ac_delta, sizeof(int32_t) * 768);
    return GZ_OK;
  }
  expect_tag(c, T_DELTA);
  get(c, ac_delta, sizeof(int32_t) * 768);
  return GZ_OK;
}

int gz_order_partition(gz_ctx* c, uint64_t lo, uint64_t hi, uint64_t* cut) {
  if (c->inner) return real()->order_partition(c->inner, lo, hi, cut);
  if (hi > c->order.size() || hi - lo <= 3) return GZ_E_ARG;
  auto less = [](const std::pair<int, float>& a, const std::pair<int, float>& b) {
    return a.second < b.second; };
  std::pair<int, float>* a = c->order.data();
  const size_t r = lo, x = lo + 1, y = lo + (hi - lo) / 2, z = hi - 1;
  if (less(a[x], a[y])) {
    if (less(a[y], a[z])) std::swap(a[r], a[y]);
    else if (less(a[x], a[z])) std::swap(a[r], a[z]);
    else std::swap(a[r], a[x]);
  } else if (less(a[x], a[z])) {
    std::swap(a[r], a[x]);
  } else if (less(a[y], a[z])) {
    std::swap(a[r], a[z]);
  } else {
    std::swap(a[r], a[y]);
  }
  size_t first = lo + 1, last = hi;
  for (;;) {
    while (less(a[first], a[lo])) ++first;
    --last;
    while (less(a[lo], a[last])) --last;
    if (!(first < last)) break;
    std::swap(a[first], a[last]);
    ++first;
  }
  *cut = first;
  return GZ_OK;
}

int gz_order_fetch(gz_ctx* c, uint64_t lo, uint64_t hi, void* out) {
  if (c->inner) return real()->order_fetch(c->inner, lo, hi, out);
  if (lo > hi || hi > c->order.size()) return GZ_E_ARG;
  memcpy(out, c->order.data() + lo, (hi - lo) * sizeof(std::pair<int, float>));
  return GZ_OK;
}

int gz_set_coeff_blocks(gz_ctx* c, const int32_t* block_index, int n, const int16_t* blocks) {
  for (int i = 0; i < n; ++i) dirty_mark(c, block_index[i]);
  if (c->inner) return real()->set_coeff_blocks(c->inner, block_index, n, blocks);
  for (int i = 0; i < n; ++i)
    if (block_index[i] < 0 || block_index[i] >= c->nb) return GZ_E_ARG;
  return GZ_OK;
}

int gz_get_coeffs(gz_ctx* c, int16_t* out) {
  if (c->inner) return real()->get_coeffs(c->inner, out);
  fprintf(stderr, "gz_replay: gz_get_coeffs is not logged\n");
  abort();
}

int gz_jpeg_histograms(gz_ctx* c, const int* q, uint32_t* counts) {
  if (c->inner) {
    const int rc = real()->jpeg_histograms(c->inner, q, counts);
    if (rc != GZ_OK) return rc;
    put_tag(c, T_HISTO);
    put(c, counts, sizeof(uint32_t) * 1536);
    return GZ_OK;
  }
  expect_tag(c, T_HISTO);
  get(c, counts, sizeof(uint32_t) * 1536);
  return GZ_OK;
}

int gz_jpeg_scan(gz_ctx* c, int ncomp, const uint8_t* depth, const uint16_t* code,
                 uint64_t* scan_bytes) {
  if (c->inner) {
    const int rc = real()->jpeg_scan(c->inner, ncomp, depth, code, scan_bytes);
    if (rc != GZ_OK) return rc;
    put_tag(c, T_SCAN);
    put(c, scan_bytes, 8);
    return GZ_OK;
  }
  if (c->held_scan) {   // (set aside when the evaluation's entry was read first)
    c->held_scan = false;
    *scan_bytes = c->held_scan_bytes;
  } else {
    expect_tag(c, T_SCAN);
    get(c, scan_bytes, 8);
  }
  c->last_scan_bytes = *scan_bytes;
  return GZ_OK;
}

int gz_jpeg_scan_keep(gz_ctx* c) { return c->inner ? real()->jpeg_scan_keep(c->inner) : GZ_OK; }

int gz_jpeg_scan_bytes(gz_ctx* c, int kept, uint8_t* out, size_t cap, size_t* n) {
  if (c->inner) {
    const int rc = real()->jpeg_scan_bytes(c->inner, kept, out, cap, n);
    if (rc != GZ_OK) return rc;
    put_tag(c, T_BYTES);
    const uint64_t n64 = *n;
    put(c, &n64, 8);
    put(c, out, *n);
    return GZ_OK;
  }
  expect_tag(c, T_BYTES);
  uint64_t n64 = 0;
  get(c, &n64, 8);
  *n = (size_t)n64;
  if (*n > cap) return GZ_E_ARG;
  get(c, out, *n);
  return GZ_OK;
}

}  // extern "C"

// ---- entry points added to the C ABI after the first version of this shim ------------------
// Record mode forwards them (dlsym by name); replay mode answers from the logged data the way
// the older entry points do.  The quick-select descents the device makes ahead are a pure
// function of the order: replay reports "no levels made", and the driver asks for the
// partitions one by one (gz_order_partition above) -- another path to the same order.
namespace {
template <class F>
F real_sym(const char* name) {
  real();
  F f = (F)dlsym(real()->h, name);
  if (!f) { fprintf(stderr, "gz_replay: %s missing in the real library\n", name); abort(); }
  return f;
}
struct Pending {
  bool order = false, scan = false;
  int direction = 0, max_block_dist = 0, use_distmap = 0, count_below = 0;
  double target_mul = 0;
  float limit = 0;
  std::vector<int32_t> next_cand;
  int ncomp = 0;
  std::vector<uint8_t> depth;
  std::vector<uint16_t> code;
  std::vector<std::pair<int, float> > mirror;
};
Pending& pending_of(gz_ctx* c) {
  static thread_local Pending p;   // (one encode per thread in this test infrastructure)
  (void)c;
  return p;
}
}  // namespace

extern "C" {

int gz_block_zeroing_orders_masked(gz_ctx* c, int comp_mask, int lookahead, int new_model,
                                   int32_t* offsets, uint8_t* idx, float* err, int cap) {
  if (comp_mask != 7) return not_logged("gz_block_zeroing_orders_masked (comp_mask != 7)");
  return gz_block_zeroing_orders(c, lookahead, new_model, offsets, idx, err, cap);
}
int gz_search_evaluations(gz_ctx* c, uint64_t* evaluations) {
  if (c->inner) return real_sym<decltype(&gz_search_evaluations)>("gz_search_evaluations")(c->inner, evaluations);
  *evaluations = 0;
  return GZ_OK;
}
// Bits and stuffed bytes of the last scan.  Not logged: replay answers with "unknown" (bits = 0),
// which the driver's GZ_VERIFY_ENTROPY check -- the only caller -- does not run on a replay.
int gz_jpeg_scan_bits(gz_ctx* c, uint64_t* bits, uint64_t* ff) {
  if (c->inner) return real_sym<decltype(&gz_jpeg_scan_bits)>("gz_jpeg_scan_bits")(c->inner, bits, ff);
  *bits = 0;
  *ff = 0;
  return GZ_OK;
}

int gz_jpeg_histograms_ncomp(gz_ctx* c, const int* q, int ncomp, uint32_t* counts) {
  if (ncomp != 3) return not_logged("gz_jpeg_histograms_ncomp (ncomp != 3)");
  return gz_jpeg_histograms(c, q, counts);
}
int gz_jpeg_scan_begin(gz_ctx* c, int ncomp, const uint8_t* depth, const uint16_t* code) {
  Pending& p = pending_of(c);
  p.scan = true;
  p.ncomp = ncomp;
  p.depth.assign(depth, depth + 2 * 3 * 256);   // (the tables are not read in replay mode; record
  p.code.assign(code, code + 2 * 3 * 256);      //  mode hands them to the real scan at _end)
  return GZ_OK;
}
int gz_jpeg_scan_end(gz_ctx* c, uint64_t* scan_bytes) {
  Pending& p = pending_of(c);
  if (!p.scan) return GZ_E_STATE;
  p.scan = false;
  return gz_jpeg_scan(c, p.ncomp, p.depth.data(), p.code.data(), scan_bytes);
}
int gz_order_build_auto_begin(gz_ctx* c, int direction, int max_block_dist, double target_mul,
                              int use_distmap, const int32_t* next_cand, int count_below, float limit) {
  Pending& p = pending_of(c);
  p.order = true;
  p.direction = direction; p.max_block_dist = max_block_dist; p.target_mul = target_mul;
  p.use_distmap = use_distmap; p.count_below = count_below; p.limit = limit;
  p.next_cand.assign(next_cand, next_cand + c->nb);
  return GZ_OK;
}
int gz_order_build_auto_end(gz_ctx* c, uint64_t* total, int32_t* blocks_to_change, uint64_t* below) {
  Pending& p = pending_of(c);
  if (!p.order) return GZ_E_STATE;
  p.order = false;
  // (the evaluation that was in flight at _begin has been fetched by gz_compare_end since)
  return gz_order_build_auto(c, p.direction, p.max_block_dist, p.target_mul, p.use_distmap,
                             p.next_cand.data(), p.count_below, p.limit, total, blocks_to_change, below);
}
int gz_order_descend(gz_ctx* c, uint64_t last, uint64_t threshold, int max_levels, uint64_t* log, int* levels) {
  (void)c; (void)last; (void)threshold; (void)max_levels; (void)log;
  *levels = 0;
  return GZ_OK;
}
int gz_order_descend_begin(gz_ctx* c, float per_block, uint64_t threshold, int max_levels) {
  (void)c; (void)per_block; (void)threshold; (void)max_levels;
  return GZ_OK;
}
int gz_order_build_auto_descend_begin(gz_ctx* c, int direction, int max_block_dist, double target_mul,
                                      int use_distmap, const int32_t* next_cand, int count_below,
                                      float limit, float per_block, uint64_t threshold, int max_levels) {
  (void)per_block; (void)threshold; (void)max_levels;
  return gz_order_build_auto_begin(c, direction, max_block_dist, target_mul, use_distmap, next_cand,
                                   count_below, limit);
}
int gz_order_descend_end(gz_ctx* c, uint64_t* log, int cap_levels, int* levels, uint64_t* last) {
  (void)c; (void)log; (void)cap_levels;
  *levels = 0;
  *last = 0;
  return GZ_OK;
}
int gz_order_host_mirror(gz_ctx* c, uint64_t entries, void** out) {
  Pending& p = pending_of(c);
  if (p.mirror.size() < entries) p.mirror.resize(entries);
  *out = p.mirror.data();
  return GZ_OK;
}
int gz_order_exported(gz_ctx* c, uint64_t* entries) {
  (void)c;
  *entries = 0;
  return GZ_OK;
}
int gz_set_orig_coeffs_420(gz_ctx* c, const int16_t* coeffs) { (void)c; (void)coeffs; return not_logged("gz_set_orig_coeffs_420"); }
int gz_downsample(gz_ctx* c, int16_t* coeffs_out) { (void)c; (void)coeffs_out; return not_logged("gz_downsample"); }
int gz_downsample_planes(gz_ctx* c, const float* y, const float* u, const float* v, int16_t* coeffs_out) {
  (void)c; (void)y; (void)u; (void)v; (void)coeffs_out;
  return not_logged("gz_downsample_planes");
}

}  // extern "C"
