/* guetzli_amd.h -- C ABI of the MI355X (gfx950) implementation of Guetzli's
 * block-parallel hot path: per-8x8-block FDCT / quantise / IDCT / colour transform and
 * the butteraugli distance map evaluated for every candidate by
 * Processor::TryQuantMatrix and Processor::SelectFrequencyMasking.
 *
 * This is the drop-in boundary.  Plain pointers and sizes only; no C++,
 * HIP or torch types.  A maintainer of the reference binds it from a
 * `guetzli::Comparator` subclass (INTEGRATION.md shows the stub); every entry point
 * cites the reference interface it replaces as path:line under /root/reference.
 *
 * Conventions
 *   - one gz_ctx per (image, GPU); calls on one context are serialised by the caller
 *     (the reference objects are single-threaded too: comparator.h:29-96); contexts are
 *     independent of each other.
 *   - every function returns GZ_OK (0) or a negative GZ_E_* code; nothing throws, nothing
 *     prints.  gz_last_error(ctx) gives a static description of the last failure.
 *   - host pointers are ordinary pageable memory unless named `dev_*`.
 *   - layouts
 *       rgb     : uint8 packed, rgb[(y*w + x)*3 + c]                (guetzli.cc ReadPNG)
 *       coeffs  : int16 DEQUANTISED DCT coefficients, component-major then block-major
 *                 (= OutputImageComponent::coeffs_, output_image.h:33-40, one after the
 *                 other for c = 0,1,2).  4:4:4 frame: coeffs[(c*nb + by*bw + bx)*64 + k],
 *                 bw=ceil(w/8), bh=ceil(h/8), nb=bw*bh.  4:2:0 frame (the two chroma
 *                 components Reset(2, 2), output_image.cc:40-49): nb luma blocks, then
 *                 nbc = ceil(w/16)*ceil(h/16) blocks of Cb, then nbc of Cr -- no MCU padding
 *                 (SaveToJpegData's padding blocks, :386-404, are implied).  A context starts
 *                 in 4:4:4; gz_downsample / gz_set_orig_coeffs_420 switch it to 4:2:0,
 *                 gz_encode_rgb / gz_set_orig_coeffs back; gz_frame_layout tells.
 *       q       : int[3][64] quantisation matrices in natural (row-major) order
 *       planes  : float, plane[y*w + x] (no row padding on the host side)
 */
#ifndef GUETZLI_AMD_H_
#define GUETZLI_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GZ_OK 0
#define GZ_E_ARG (-1)        /* bad argument (null, size out of range, w/h < 8 ...) */
#define GZ_E_NO_DEVICE (-2)  /* no usable gfx950 device / HIP runtime failure at init */
#define GZ_E_HIP (-3)        /* a HIP call failed; see gz_last_error */
#define GZ_E_STATE (-4)      /* call sequence violated (e.g. compare before coefficients) */
#define GZ_E_NOMEM (-5)

typedef struct gz_ctx gz_ctx;

/* Library / device ------------------------------------------------------------ */
int gz_abi_version(void);                 /* currently 5 (4 + gz_config.patch_reconstruct, .opsin_ahead, gz_compare_counters) */
/* Device and pinned host memory of destroyed contexts is kept (per device, exact sizes, at
 * most GZ_POOL_MB megabytes of device memory, default 16384) for the next context of the same
 * image size: a batch of same-sized images allocates once.  gz_trim_pool releases everything
 * that is cached; GZ_POOL_MB=0 turns the caching off. */
int gz_trim_pool(void);
int gz_device_count(void);                /* number of visible HIP devices, <0 on error */
const char* gz_strerror(int code);
const char* gz_last_error(const gz_ctx* ctx);
/* A caller that runs several images at once in this process (one thread + one context each) says so before it starts
 * them: n = images in flight (<= 1: lone images again).  Only a hint, process-wide, no result depends on it: a context
 * created while the device is empty takes a highest-priority main stream (worth 2 % of a lone 4K chain) unless the hint
 * says that company is coming -- a priority stream is one more hardware queue for the runtime to multiplex and costs a
 * batch 1-4 %; with n > 1 the idle priority stream sets of the pool are destroyed as well. */
void gz_hint_images_in_flight(int n);
/* PCI bus id of a HIP device ("0000:05:00.0"; hipDeviceGetPCIBusId) -- for NUMA-aware placement of the
 * process that feeds it (bench.py / guetzli_amd/affinity.py); out needs >= 16 bytes. */
int gz_device_pci_bus_id(int device, char* out, int cap);

/* Run-time configuration of a context -------------------------------------------
 * What used to be read from GZ_* environment variables on every call (rounds 1-5) is a struct of the context:
 * gz_create fills it ONCE from the environment (gz_config_from_environment: the variable behind every field is
 * named below), gz_get_config / gz_set_config read and replace it (between calls, never while work of the
 * context is in flight).  None of the fields changes a result bit: they choose kernel instantiations and
 * stream use, and exist for the tests (every instantiation on small images) and for A/B measurements.
 * Process-wide settings stay in the environment, read once: GZ_POOL_MB (cache of freed device memory),
 * GZ_CU_PARTITION / GZ_CU_MAIN / GZ_CU_SIDE (CU-masked stream sets, experiments). */
typedef struct gz_config {
  int struct_size;      /* sizeof(gz_config) as the caller compiled it (checked by gz_set_config) */
  int blur_packed;      /* GZ_BLUR_PK      -1: by image size (row / column pairs from 1.5 MPix on); 0, 1: forced */
  int tile_rows;        /* GZ_TILE_ROWS     0: by image size (16-row blur tiles below 7 MPix); 16, 32: forced */
  int single_stream;    /* GZ_SINGLE_STREAM -1: one stream when other contexts are alive on the device (a batch's images in
                                            flight), three for a lone context; 0: always three; 1: always one */
  int store_distmap;    /* GZ_STORE_DISTMAP 1: every Compare stores the distance map (default: only gz_compare with
                                            distmap != NULL and the stage probes do) */
  int side_small;       /* GZ_SIDE_SMALL    1: side-branch blurs in Malta-sized forms (experiment, round 6) */
  int malta_pad_bytes;  /* GZ_MALTA_PAD     unused dynamic LDS per Malta workgroup (experiment, round 6) */
  int patch_reconstruct;/* GZ_PATCH_RECON   1 (default): gz_apply_candidate_steps / gz_apply_coeff_edits transform the block
                                            positions they change again (4:4:4 frames of half a megapixel and more, while
                                            fewer than half of the blocks change) and the Compare behind them skips its
                                            full reconstruction; 0: every Compare reconstructs the whole image; 2: as 1 at
                                            every image size, and every such Compare checks the patched planes against a
                                            full reconstruction first (GZ_E_STATE on a difference; tests).
                                            gz_time_compare / gz_compare_enqueue always run the whole chain. */
  int opsin_ahead;      /* GZ_OPSIN_AHEAD   1 (default): with patch_reconstruct, a context that has the device to itself runs
                                            the opsin blur of the next Compare right behind gz_apply_candidate_steps'
                                            patches -- while the host takes its serial steps -- and gz_apply_coeff_edits
                                            recomputes the opsin tiles around the blocks it edits; 0: every Compare runs it */
} gz_config;
int gz_config_from_environment(gz_config* out);
int gz_get_config(const gz_ctx* ctx, gz_config* out);
int gz_set_config(gz_ctx* ctx, const gz_config* in);

/* Context ---------------------------------------------------------------------
 * gz_create: replaces guetzli::ButteraugliComparator::ButteraugliComparator
 * (butteraugli_comparator.cc:51-61) -- uploads the original sRGB image, converts it to
 * linear RGB (LinearRgb, :33-47) and precomputes the original's PsychoImage pi0_
 * (butteraugli.cc:784-791).  `target` is Params::butteraugli_target
 * (processor.h:30).  Requires w,h >= 8 (butteraugli) -- Process() itself only builds a
 * comparator when w,h >= 32 (processor.cc:940) -- and w,h < 65536 (JPEG) with at most
 * 11.18 M 8x8 blocks (715 MPix: coefficient positions are 32-bit), else GZ_E_ARG.  Returns
 * NULL on failure, *err set.
 * Every entry point that takes a context runs on the context's device and leaves the calling
 * thread's current HIP device as it found it (a thread may own contexts on several GPUs). */
gz_ctx* gz_create(int device, int w, int h, const uint8_t* rgb, float target, int* err);
void gz_destroy(gz_ctx* ctx);
/* Replace the original image of an existing context (same w, h): what constructing a new
 * ButteraugliComparator on other pixels does.  Used by the JPEG-input path, whose original is
 * DecodeJpegToRGB(jpg) (jpeg_data_decoder.cc:45-54) -- an IDCT of the input's coefficients
 * that the context itself computes (gz_set_orig_coeffs, gz_quantize(NULL), gz_reconstruct). */
int gz_set_rgb(gz_ctx* ctx, const uint8_t* rgb);
int gz_synchronize(gz_ctx* ctx);
/* Run subsequent work of this context on an externally owned hipStream_t (e.g. torch's
 * current stream, so that torch.cuda.Event timing sees the kernels).  NULL restores the
 * context's own stream. */
int gz_set_stream(gz_ctx* ctx, void* hip_stream);

/* Block path ------------------------------------------------------------------
 * gz_encode_rgb: replaces EncodeRGBToJpeg (jpeg_data_encoder.cc:66-117) for the
 * all-ones quantisation: RGBToYUV16 (:40-48, edge-clamped gather :88-98) ->
 * ComputeBlockDCT (fdct.cc:230) -> (v*65537 + 0x80000) >> 20 (:33-35).  The result is
 * kept on the device as the context's ORIGINAL coefficients and, if coeffs_out != NULL,
 * copied to the host. */
int gz_encode_rgb(gz_ctx* ctx, int16_t* coeffs_out);

/* The same transform without a context, for images too small for butteraugli (w or h < 32:
 * Process() emits the unquantised JPEG, processor.cc:832-838, and gz_create needs >= 8):
 * rgb w*h*3 -> coeffs [3][nb][64].  0 < w, h < 65536. */
int gz_encode_rgb_only(int device, const uint8_t* rgb, int w, int h, int16_t* coeffs_out);

/* Upload original (unquantised) coefficients computed elsewhere (JPEG input path:
 * JPEGData after RemoveOriginalQuantization, processor.cc:84-97). */
int gz_set_orig_coeffs(gz_ctx* ctx, const int16_t* coeffs);
/* The same for a YUV 4:2:0 input (jpg.Is420(), processor.cc:814-815): luma blocks on the
 * 8x8 grid, then the two chroma components on the 16x16 grid (the input's MCU padding left
 * out, as OutputImageComponent::CopyFromJpegComponent does, output_image.cc:211-230).  The
 * context's frame becomes 4:2:0. */
int gz_set_orig_coeffs_420(gz_ctx* ctx, const int16_t* coeffs);
/* OutputImage::Downsample (output_image.cc:304-340) as Processor::DownsampleImage calls it
 * (processor.cc:97-104; use_silver_screen == false) on the ORIGINAL coefficients of a 4:4:4
 * frame: ToFloatPixels (:99-121) of the three components, PreProcessChannel
 * (preprocess_downsample.cc:157-279) on V, then on U -- sharpen the channel where the image
 * is red and dark, blur it where it is smooth -- and SetDownsampledCoefficients (:265-300) of
 * U and V by 2 x 2.  Luma is kept.  The context's frame becomes 4:2:0; coeffs_out (may be NULL)
 * receives the new original, nb + 2*nbc blocks.  The caller skips the call for a greyscale
 * image, as the reference does (:305-308). */
int gz_downsample(gz_ctx* ctx, int16_t* coeffs_out);
/* The use_silver_screen branch of OutputImage::Downsample (output_image.cc:309-318): y, u, v
 * are the three w x h planes RGBToYUV420 returned (preprocess_downsample.cc:452-476; host
 * work: it is twenty rounds of libm pow(), guetzli_amd/host/silver_screen.cc restates it on
 * OutputImage::ToSRGB() = gz_quantize(NULL) + gz_reconstruct of the 4:4:4 original).  All three
 * components become SetDownsampledCoefficients of their plane -- luma by 1 x 1, chroma by
 * 2 x 2 -- and the context's frame 4:2:0, as after gz_downsample. */
int gz_downsample_planes(gz_ctx* ctx, const float* y, const float* u, const float* v,
                         int16_t* coeffs_out);
/* Current frame: chroma subsampling factor (1 or 2), luma blocks, blocks per chroma
 * component.  Any pointer may be NULL. */
int gz_frame_layout(gz_ctx* ctx, int* chroma_factor, int* luma_blocks, int* chroma_blocks);

/* Candidate := original, then OutputImage::ApplyGlobalQuantization(q)
 * (output_image.cc:232-243,342-346; Quantize quantize.h:24-29).  This is the coefficient
 * side of Processor::TryQuantMatrix (processor.cc:298-307).  q == NULL means all ones
 * (plain CopyFromJpegData).  coeffs_out (may be NULL) receives the candidate's
 * coefficients for the host JPEG writer. */
int gz_quantize(gz_ctx* ctx, const int* q, int16_t* coeffs_out);

/* Replace the whole candidate / individual candidate blocks with host data
 * (OutputImageComponent::SetCoeffBlock, output_image.cc:123-132).  block_index[i] =
 * by*bw + bx, all distinct; blocks holds n * 3 * 64 int16 (component-major per block:
 * Y,Cb,Cr). */
int gz_set_coeffs(gz_ctx* ctx, const int16_t* coeffs);
/* (4:4:4 frames only) */
int gz_set_coeff_blocks(gz_ctx* ctx, const int32_t* block_index, int n,
                        const int16_t* blocks);
int gz_get_coeffs(gz_ctx* ctx, int16_t* coeffs_out);
/* Single-coefficient form of SetCoeffBlock: candidate[pos[i]] = val[i] for n distinct
 * positions in the frame's coefficient array: pos = (first block of component c + block)*64 + k
 * (what one phase-B iteration changes). */
int gz_apply_coeff_edits(gz_ctx* ctx, const int32_t* pos, const int16_t* val, int n);

/* Candidate -> pixels, for parity checks and for callers that want the decoded image:
 * OutputImage::ToSRGB (output_image.cc:411-425) and ToLinearRGB (:427-440).
 * srgb: w*h*3 uint8 or NULL; linear: 3 planes of w*h float or NULL. */
int gz_reconstruct(gz_ctx* ctx, uint8_t* srgb, float* linear);

/* Double-precision DCT -----------------------------------------
 * gz_dct_double_blocks: ComputeBlockDCTDouble (inverse == 0) / ComputeBlockIDCTDouble
 * (inverse != 0), dct_double.cc:76-85, on n bare blocks of 64 doubles, in place.
 * gz_component_to_float_pixels: OutputImageComponent::ToFloatPixels with stride 1
 * (output_image.cc:99-121): one 4:4:4 component's coefficients [ceil(w/8)*ceil(h/8)][64]
 * -> out[y*w + x] = float(IDCTDouble + 128).
 * gz_component_set_downsampled: SetDownsampledCoefficients (output_image.cc:265-300):
 * pixels (w*h floats of the full-resolution component) averaged fx x fy, forward
 * DCTDouble, DC - 1024, round() -> int16 coefficients of the subsampled component,
 * ceil(w/(8 fx)) * ceil(h/(8 fy)) blocks (OutputImageComponent::Reset, :40-49).
 * These are the two consumers of dct_double.cc on the reference's YUV420 path
 * (OutputImage::Downsample, :304-340); all three are bit-exact (FP64, no contraction).
 * Context-free: they take a device ordinal. */
int gz_dct_double_blocks(int device, double* blocks, int n, int inverse);
int gz_component_to_float_pixels(int device, const int16_t* coeffs, int w, int h,
                                 float* out);
int gz_component_set_downsampled(int device, const float* pixels, int w, int h, int fx,
                                 int fy, int16_t* coeffs_out);

/* Whole-image distance -----------------------------------------------------------
 * gz_compare: replaces guetzli::ButteraugliComparator::Compare
 * (butteraugli_comparator.cc:63-75) on the current candidate: IDCT + colour + linear
 * (ToLinearRGB), OpsinDynamicsImage, SeparateFrequencies, Malta x6, L2Diff*,
 * SameNoiseLevels, Mask, CombineChannels, CalculateDiffmap, max
 * (butteraugli.cc:799-908,1623-1633).
 *   distance  : distmap_aggregate() (butteraugli_comparator.h:56)          [required]
 *   distmap   : distmap() (h:55), w*h floats                               [may be NULL]
 *   block_max : per-8x8-block maximum of the distance map, nb floats = the first loop
 *               of ComputeBlockErrorAdjustmentWeights (:505-520)           [may be NULL]
 * With distmap == block_max == NULL only 4 bytes cross PCIe. */
int gz_compare(gz_ctx* ctx, float* distance, float* distmap, float* block_max);

/* gz_compare in two halves, so that the caller can do host work (e.g. build the Huffman
 * codes of the same candidate) while the evaluation runs: _begin enqueues it and returns,
 * _end waits and returns distmap_aggregate().  Between them only calls that leave the
 * candidate unchanged are allowed: gz_jpeg_scan (which runs on the context's entropy stream,
 * beside the evaluation), gz_jpeg_scan_keep / _bytes, gz_get_coeffs. */
int gz_compare_begin(gz_ctx* ctx);
int gz_compare_end(gz_ctx* ctx, float* distance);

/* Enqueue `iters` back-to-back Compare evaluations of the current candidate on the
 * context's stream without any host transfer or synchronisation (for HIP-event timing
 * of the resident-in-HBM rate).  The distance of the last one is readable with
 * gz_last_distance after gz_synchronize.  Every one of them is the whole chain, reconstruction included
 * (gz_config.patch_reconstruct does not apply: nothing changes the candidate between them). */
int gz_compare_enqueue(gz_ctx* ctx, int iters);
int gz_last_distance(gz_ctx* ctx, float* distance);
/* Convenience: time `iters` enqueued Compare evaluations with hipEvents recorded on the
 * context's stream; returns total milliseconds. */
int gz_time_compare(gz_ctx* ctx, int iters, float* total_ms);

/* ComputeBlockErrorAdjustmentWeights (butteraugli_comparator.cc:494-558) from the
 * block maxima of the LAST gz_compare (or all-zero if use_distmap == 0, which is what
 * SelectFrequencyMasking does on its first "up" iteration, processor.cc:625-629).
 * block_weight: nb floats, in/out exactly like the reference's vector. */
int gz_block_weights(gz_ctx* ctx, int direction, int max_block_dist, double target_mul,
                     int use_distmap, float* block_weight);
/* The same for blocks of 8*factor x 8*factor pixels (factor_x = factor_y = factor, 1 or 2:
 * the chroma grid of a 4:2:0 frame); block_weight has ceil(w/(8 factor))*ceil(h/(8 factor))
 * entries. */
int gz_block_weights_factor(gz_ctx* ctx, int direction, int max_block_dist, double target_mul,
                            int use_distmap, int factor, float* block_weight);

/* Per-block zeroing search ---------------------------------------------------------
 * gz_block_zeroing_orders: phase A of Processor::SelectFrequencyMasking
 * (processor.cc:554-590) for comp_mask 7 on a 4:4:4 image: for every block,
 * ComputeBlockZeroingOrder (processor.cc:364-467) with CompareBlock
 * (butteraugli_comparator.cc:457-488) evaluated on the device, between the reference's
 * StartBlockComparisons / FinishBlockComparisons (:415-425).  Operates on the current
 * candidate (e.g. after gz_quantize(best_q)) and the original coefficients.
 *   lookahead = Params::zeroing_greedy_lookahead (3), new_model =
 *   Params::new_zeroing_model (processor.h:35-36).
 * Outputs are the reference's CSR arrays: candidate_coeff_offsets[nb+1],
 * candidate_coeffs[], candidate_coeff_errors[] (processor.cc:554-558).  cap = capacity of
 * idx/err in elements (nb*189 always suffices); if too small, GZ_E_ARG is returned and
 * offsets[nb] holds the required size.  err may be NULL: the errors then stay on the device
 * only, where gz_order_build reads them. */
int gz_block_zeroing_orders(gz_ctx* ctx, int lookahead, int new_model, int32_t* offsets,
                            uint8_t* idx, float* err, int cap);
/* The same for any comp_mask of SelectFrequencyMasking (processor.cc:539-590): candidates come
 * from the components in the mask only, and the grid is that of the mask's last component
 * (:546-552).  4:4:4 frame: any mask, the 8x8 grid.  4:2:0 frame: mask 1 (luma; 8x8 grid,
 * chroma pixels fixed) or mask 6 (chroma; ceil(w/16) x ceil(h/16) grid -- every candidate is
 * compared on the up to four 8x8 blocks of its 16x16 area and scored by the largest error,
 * processor.cc:420-430, with the 2x2-subsampled pixel model of
 * OutputImageComponent::UpdatePixelsForBlock, output_image.cc:146-203).  offsets has one
 * entry per block of that grid plus one.  Phase B's order functions below then work on the
 * same grid and mask. */
int gz_block_zeroing_orders_masked(gz_ctx* ctx, int comp_mask, int lookahead, int new_model,
                                   int32_t* offsets, uint8_t* idx, float* err, int cap);
/* Number of CompareBlock evaluations (butteraugli_comparator.cc:457-488) the last
 * gz_block_zeroing_orders* call made -- the unit the search's throughput is reported in
 * (bench.py: evaluations per second). */
int gz_search_evaluations(gz_ctx* ctx, uint64_t* evaluations);
/* Process-wide, since the library was loaded (tests, bench.py): out[0] = Compares whose candidate planes were kept
 * current by the calls that changed the candidate and that skipped the full reconstruction (gz_config.patch_reconstruct),
 * out[1] = those of them that were checked against a full reconstruction (patch_reconstruct == 2), out[2] = Compares in all,
 * out[3] = Compares that also found their opsin image in place (gz_config.opsin_ahead), out[4] = those of them checked. */
int gz_compare_counters(uint64_t out[5]);
/* The per-block form of the seam: Comparator::SwitchBlock + CompareBlock
 * (butteraugli_comparator.cc:427-488; factor_x = factor_y = 1) for n independent pairs of a
 * block position block_xy[i] = {block_x, block_y} and that block's candidate coefficients
 * coeffs[i][3][64] (Y, Cb, Cr; dequantised): out[i] = the distance CompareBlock returns.
 * StartBlockComparisons' mask (:415-421) is computed on first use.  This is what a
 * `guetzli::Comparator` subclass calls from its CompareBlock (tests/integration/ drives the
 * UNMODIFIED reference Processor through it); a search that wants speed batches whole blocks'
 * searches with gz_block_zeroing_orders instead -- one call per block costs a round trip. */
int gz_compare_blocks(gz_ctx* ctx, int n, const int32_t* block_xy, const int16_t* coeffs,
                      double* out);
/* The same for 8x8 windows given by their YCbCr PIXELS -- what CompareBlock reads of the image:
 * OutputImage::ToLinearRGB(xmin, ymin, 8, 8) (butteraugli_comparator.cc:467), i.e.
 * OutputImageComponent::ToPixels (output_image.cc:69-96, edge replication included) of the three
 * components.  This form serves every frame (comparator.h:50-52: SwitchBlock carries
 * factor_x, factor_y): the pixels of a 2x2-subsampled component depend on its neighbours' blocks,
 * which the caller's OutputImage holds.  block_xy: the window's position on the 8x8 luma grid
 * (block_x * factor_x + off_x, block_y * factor_y + off_y); ycc: n x 3 x 64 bytes. */
int gz_compare_block_pixels(gz_ctx* ctx, int n, const int32_t* block_xy, const uint8_t* ycc, double* out);
/* OutputImageComponent::Reset(factor, factor) of the two chroma components (output_image.cc:
 * 32-57) without touching their content: the layout gz_set_coeffs / gz_get_coeffs /
 * gz_set_orig_coeffs* use from now on (1: 4:4:4, 2: 4:2:0).  A change drops the candidate, the
 * original coefficients and the search results; the original PIXELS (gz_create) stay. */
int gz_set_frame(gz_ctx* ctx, int chroma_factor);
/* Host-only helper, exported for tests: the ranked input_order of
 * ComputeBlockZeroingOrder (processor.cc:381-400) for every block, as CSR, by std::sort
 * itself.  gz_block_zeroing_orders ranks on the device. */
int gz_rank_zeroing_candidates(const int16_t* coeffs, const int16_t* orig, int nb,
                               int new_model, int32_t* offsets, uint8_t* idx);
/* Test hook for the device-side ranking (k_rank_candidates sorts every block's candidates
 * with libstdc++'s std::sort permutation): narr arrays of cnt[a] <= 192 keys, keys[a][192];
 * perm[a][i] = index (into array a) of the element std::sort puts at position i. */
int gz_probe_rank_sort(int device, const float* keys, const int32_t* cnt, int narr, uint8_t* perm);

/* Global candidate order of phase B ------------------------------
 * Phase B of SelectFrequencyMasking builds `global_order` = (block, val) for every
 * remaining candidate of every block with a non-zero weight (processor.cc:622-663),
 * std::sort-s it by val (:675-678) and consumes a prefix.  The order lives on the device:
 *
 * gz_order_build: the construction loop (:636-663) over the CSR arrays the last
 *   gz_block_zeroing_orders left on the device.  next_cand = last_indexes (:608),
 *   max_block_error (:607), block_weight = the output of gz_block_weights, nb values each.
 *   *total = global_order.size(), *blocks_to_change as the reference counts it; if
 *   count_below, *below = number of entries with val < limit (what the partition_point of
 *   :690-696 yields on the sorted order).
 * gz_order_upload: replace the device order with n host entries instead.
 * gz_order_partition: one step of libstdc++'s introsort on [lo, hi) of the device order --
 *   std::__unguarded_partition_pivot: median of {lo+1, mid, hi-1} to lo, unguarded Hoare
 *   partition of [lo+1, hi) around it -- with exactly the arrangement and *cut the serial
 *   algorithm gives (ties across blocks are not stable under std::sort and feed the JPEG
 *   bytes).  Requires hi - lo > 3.
 * gz_order_fetch: entries [lo, hi) to the host, 8 bytes each: {int32 block; float val}
 *   (the layout of std::pair<int, float>).
 * The search driver (guetzli_amd/host/lazy_sort.h) refines the leading ranges on the
 * device until they are small, fetches them, and finishes them on the host. */
int gz_order_build(gz_ctx* ctx, int direction, const int32_t* next_cand,
                   const float* max_block_error, const float* block_weight, int count_below,
                   float limit, uint64_t* total, int32_t* blocks_to_change, uint64_t* below);
/* The same with the per-block state kept on the device across iterations:
 * gz_order_reset: max_block_error := 0 (processor.cc:607).
 * gz_order_build_auto: block_weight = ComputeBlockErrorAdjustmentWeights(direction,
 *   max_block_dist, target_mul) of the last gz_compare (all-zero distance map if
 *   use_distmap == 0) computed on the device (butteraugli_comparator.cc:494-558), then the
 *   construction loop as gz_order_build with the device-resident max_block_error.
 * gz_order_advance: max_block_error[i] += block_weight[i] * val_threshold * direction
 *   (processor.cc:754-756) with the weights of the last gz_order_build_auto. */
int gz_order_reset(gz_ctx* ctx);
int gz_order_build_auto(gz_ctx* ctx, int direction, int max_block_dist, double target_mul,
                        int use_distmap, const int32_t* next_cand, int count_below, float limit,
                        uint64_t* total, int32_t* blocks_to_change, uint64_t* below);
/* gz_order_build_auto in two halves, so that the order of the NEXT iteration is constructed on
 * the device right behind the evaluation of the candidate it depends on (gz_compare_begin on
 * the same context: use_distmap is then allowed before gz_compare_end), with no host round
 * trip in between -- processor.cc:607-663 run back to back with :767.  _begin enqueues and
 * returns (next_cand is copied); _end waits and reports what gz_order_build_auto reports.
 * Any other gz_order_build* call drops a pending _begin. */
int gz_order_build_auto_begin(gz_ctx* ctx, int direction, int max_block_dist, double target_mul,
                              int use_distmap, const int32_t* next_cand, int count_below, float limit);
int gz_order_build_auto_end(gz_ctx* ctx, uint64_t* total, int32_t* blocks_to_change, uint64_t* below);
int gz_order_advance(gz_ctx* ctx, float val_threshold, int direction);
/* gz_apply_candidate_steps: the coefficient side of the global loop's steps
 * (processor.cc:704-736) for whole blocks: block blocks[i] advances by counts[i] steps in
 * `direction` from the next_cand the last gz_order_build* call uploaded -- each step zeroes
 * ("up") or restores ("down") the block's next candidate coefficient unless it is
 * "precious" (:722-733).  blocks distinct; the device copy of next_cand is left as uploaded
 * (the next gz_order_build* call brings the caller's). */
int gz_apply_candidate_steps(gz_ctx* ctx, int direction, const int32_t* blocks,
                             const int32_t* counts, int n);
/* What the last gz_apply_candidate_steps did to the AC symbol statistics: ac_delta[3][256] =
 * BuildACHistograms (jpeg_data_writer.cc:255-275) of the image after the steps minus before,
 * raw occurrence counts under the quantiser of the last gz_jpeg_histograms (which must have
 * been called before the steps).  Computed from the touched blocks only, so that the caller's
 * size model (processor.cc:471-525) does not need a recount of the whole image. */
int gz_steps_histogram_delta(gz_ctx* ctx, int32_t* ac_delta);
int gz_order_upload(gz_ctx* ctx, const void* entries, uint64_t n);
int gz_order_partition(gz_ctx* ctx, uint64_t lo, uint64_t hi, uint64_t* cut);
int gz_order_fetch(gz_ctx* ctx, uint64_t lo, uint64_t hi, void* out);
/* A host array of at least `entries` order entries (8 bytes each), page-locked and owned by the
 * context: gz_order_fetch into it (at any offset) is one DMA transfer without the landing copy a
 * pageable destination needs.  The search driver keeps its host copy of the order there.  A call
 * that has to grow the array invalidates the pointer of the call before; gz_destroy frees it. */
int gz_order_host_mirror(gz_ctx* ctx, uint64_t entries, void** out);
/* After gz_order_build_auto_descend_begin ... gz_order_descend_end: the number of leading entries
 * of the order that the device has already written into the host mirror behind the descent (the
 * prefix up to the end of the range the descent ended in, when that range is within `threshold`
 * and the prefix within 2^19 entries; else 0).  They are what gz_order_fetch(ctx, 0, n, mirror)
 * would deliver at that moment: the driver skips that fetch. */
int gz_order_exported(gz_ctx* ctx, uint64_t* entries);
/* The quick-select descent of that sort, decided on the device.  What the global loop needs
 * before its stopping rule can fire (processor.cc:743-746: not before min_coeffs_to_change
 * steps) is the SET of the leading entries of the sorted order; std::sort's introsort reaches
 * it by partitioning the whole order, then the part that holds position `last`, and so on.
 * gz_order_descend runs up to max_levels of exactly those gz_order_partition steps back to back
 * -- on the range [0, n), then on whichever side of the cut holds `last`, while the range has
 * more than `threshold` entries and introsort's depth budget (2 floor(lg n)) lasts -- without
 * the host in between, and reports them: log[3*i] = lo, log[3*i+1] = hi, log[3*i+2] = cut of
 * step i, *levels of them.  Same arrangement, same cuts as the same sequence of
 * gz_order_partition calls.
 * gz_order_descend_begin: the same enqueued behind gz_order_build_auto_begin, before the order's
 *   size is known to the host: `last` is derived on the device as the search driver derives it,
 *   min_coeffs = (int)(per_block * blocks_to_change) (:685-687), capped at n - 1, rounded down
 *   to the entropy-code refresh interval of 10 (:739-741), minus one (0 if that is 0).
 * gz_order_descend_end: its log, after gz_order_build_auto_end, and the `last` it was made
 *   for (meaningful when *levels > 0): a caller that derives another position must not replay it. */
int gz_order_descend(gz_ctx* ctx, uint64_t last, uint64_t threshold, int max_levels,
                     uint64_t* log, int* levels);
int gz_order_descend_begin(gz_ctx* ctx, float per_block, uint64_t threshold, int max_levels);
/* gz_order_build_auto_begin followed by gz_order_descend_begin as ONE call: the same kernels, and
 * one device-to-host transfer for everything the caller waits for afterwards -- the order's size
 * and counters (gz_order_build_auto_end), the descent's log (gz_order_descend_end) and, when the
 * call is made between gz_compare_begin and gz_compare_end, that evaluation's distance
 * (gz_compare_end then only waits): three small copies less on the stream, each a dispatch of its
 * own between the evaluation and the host. */
int gz_order_build_auto_descend_begin(gz_ctx* ctx, int direction, int max_block_dist, double target_mul,
                                      int use_distmap, const int32_t* next_cand, int count_below,
                                      float limit, float per_block, uint64_t threshold, int max_levels);
int gz_order_descend_end(gz_ctx* ctx, uint64_t* log, int cap_levels, int* levels, uint64_t* last);

/* Entropy coding of the candidate -----------------------------------------------------
 * The search needs the exact size of every candidate's JPEG (ScoreJPEG) and the bytes of the
 * winner only.  The candidate's coefficients are resident, so the symbol statistics and the
 * scan are produced on the device; Huffman code construction (ClusterHistograms,
 * BuildHuffmanCode, jpeg_data_writer.cc:158-342) and the marker segments stay on the host.
 *
 * gz_jpeg_histograms: BuildDCHistograms + BuildACHistograms (jpeg_data_writer.cc:241-275)
 * of the candidate quantised by q (int[3][64]; value = coeff / q as in
 * OutputImage::SaveToJpegData, output_image.cc:348-409).  counts: uint32 [2][3][256] =
 * (DC, AC) x component x symbol, raw occurrence counts.  Remembers q for gz_jpeg_scan.
 *
 * gz_jpeg_scan: EncodeScan (:499-536) of the same frame with ncomp (1 or 3) components
 * under the codes depth / code ([2][3][256] each, per component, already resolved through
 * the DC/AC table indexes).  The bit stream stays on the device; *scan_bytes receives its
 * exact length in bytes including the 0x00 stuffed after every 0xFF and the final
 * 1-padding (jpeg_bit_writer.h:31-108).  Codes of used symbols longer than 16 bits can overflow the scan's bound: GZ_E_ARG.  The
 * kernels run on the context's entropy stream, ordered behind whatever put the candidate in
 * place; the call returns after its single synchronisation of that stream.
 *
 * gz_jpeg_scan_begin / _end are the same call in two halves: _begin enqueues the scan on the
 * entropy stream and returns, _end waits for it and returns the length.  Between them the
 * caller may enqueue what gz_compare_begin's contract allows (the evaluation itself, the next
 * order's construction: gz_order_build_auto_begin / gz_order_descend_begin) -- the entropy
 * coder then runs beside the evaluation instead of behind the host code that enqueues it.
 *
 * gz_jpeg_scan_keep snapshots the last scan on the device (the caller's "best so far",
 * processor.cc:139-148); gz_jpeg_scan_bytes downloads the last (kept = 0) or the kept
 * (kept = 1) scan as stuffed bytes. */
int gz_jpeg_histograms(gz_ctx* ctx, const int* q, uint32_t* counts);
/* In a 4:2:0 frame the luma DC statistics depend on whether the chroma components are
 * written (MCU order and padding blocks, output_image.cc:357-404) or not (ncomp == 1: luma
 * alone in raster order): this variant says which.  gz_jpeg_histograms == ncomp 3. */
int gz_jpeg_histograms_ncomp(gz_ctx* ctx, const int* q, int ncomp, uint32_t* counts);
int gz_jpeg_scan(gz_ctx* ctx, int ncomp, const uint8_t* depth, const uint16_t* code,
                 uint64_t* scan_bytes);
int gz_jpeg_scan_begin(gz_ctx* ctx, int ncomp, const uint8_t* depth, const uint16_t* code);
int gz_jpeg_scan_end(gz_ctx* ctx, uint64_t* scan_bytes);
int gz_jpeg_scan_keep(gz_ctx* ctx);
/* Of the last scan: its length in bits (before byte stuffing and padding) and the number of 0xFF
 * bytes it holds, i.e. of 0x00 bytes stuffed: scan_bytes = ceil(bits / 8) + stuffed.  The bit
 * count is a function of the symbol statistics and the code lengths alone (every occurrence costs
 * its code plus its extra bits, jpeg_data_writer.cc:446-497); the host driver derives it that way
 * to bound a candidate's size without coding it, and checks itself against this. */
int gz_jpeg_scan_bits(gz_ctx* ctx, uint64_t* bits, uint64_t* stuffed);
int gz_jpeg_scan_bytes(gz_ctx* ctx, int kept, uint8_t* out, size_t cap, size_t* n);

/* Stage probes (parity tests) -----------------------------------------------------
 * Each runs ONE stage of the pipeline on host-provided input through the same kernels
 * gz_compare uses.  They exist so that tests can localise a divergence; they are not on
 * the hot path.  All planes are w*h floats of a context created for that w,h. */
int gz_probe_blur(gz_ctx* ctx, const float* in, float sigma, float border_ratio,
                  float* out);                                  /* Blur, butteraugli.cc:229 */
int gz_probe_opsin(gz_ctx* ctx, const float* rgb3, float* xyb3); /* :324-366 */
int gz_probe_separate_frequencies(gz_ctx* ctx, const float* xyb3,
                                  float* out10);                /* :489-622; lf3,mf3,hf2,uhf2 */
int gz_probe_diffmap(gz_ctx* ctx, const float* rgb0_3, const float* rgb1_3,
                     float* diffmap, float* score);             /* :784-908 both sides */
int gz_probe_mask(gz_ctx* ctx, const float* xyb0_3, const float* xyb1_3, float* mask3,
                  float* mask_dc3);                             /* Mask, :1741-1817 */
/* Block kernels on bare 64-element blocks (n blocks each). */
int gz_probe_idct_blocks(int device, const int16_t* blocks, int n, uint8_t* out);
int gz_probe_fdct_blocks(int device, int16_t* blocks, int n);
/* Arithmetic self-check of the device: IEEE float/double divide & sqrt, conversions and
 * the absence of FMA contraction, on host-supplied operands.  op: 0 a/b (f32),
 * 1 sqrt(a) (f32), 2 a/b (f64), 3 sqrt(a) (f64), 4 a*b+c unfused (f32), 5 a*b+c unfused
 * (f64), 6 (float)a for double a.  Arrays hold n elements of the operand type
 * (out of (float)a is float). */
int gz_probe_arith(int device, int op, const void* a, const void* b, const void* c,
                   void* out, int n);

#ifdef __cplusplus
}
#endif
#endif /* GUETZLI_AMD_H_ */
