#include "processor.h"

#include <math.h>
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <chrono>
#include <cstdlib>
#include <exception>
#include <utility>

#include "../../include/guetzli_amd.h"
#include "jpeg_reader.h"
#include "reader_dump.h"
#include "png_reader.h"
#include "silver_screen.h"
#include "jpeg_writer.h"
#include "code_refresh.h"
#include "lazy_sort.h"
#include "parallel.h"

namespace guetzli_amd {

// ------------------------------------------------------------- quality / score tables --
namespace {
const int kLowestQuality = 70;
const int kHighestQuality = 110;
// Median butteraugli scores of libjpeg-turbo output per quality level 70..111
// (kScoreForQuality, quality.cc:31-74) -- data.
const double kQualityScore[] = {
  2.810761, 2.729300, 2.689687, 2.636811, 2.547863, 2.525400, 2.473416, 2.366133, 2.338078,
  2.318654, 2.201674, 2.145517, 2.087322, 2.009328, 1.945456, 1.900112, 1.805701, 1.750194,
  1.644175, 1.562165, 1.473608, 1.382021, 1.294298, 1.185402, 1.066781, 0.971769, 0.852901,
  0.724544, 0.611302, 0.443185, 0.211578, 0.209462, 0.207346, 0.205230, 0.203114, 0.200999,
  0.198883, 0.196767, 0.194651, 0.192535, 0.190420, 0.190420,
};
}  // namespace

double ButteraugliScoreForQuality(double quality) {
  if (quality < kLowestQuality) quality = kLowestQuality;
  if (quality > kHighestQuality) quality = kHighestQuality;
  const int index = static_cast<int>(quality);
  const double mix = quality - index;
  return kQualityScore[index - kLowestQuality] * (1 - mix) +
         kQualityScore[index - kLowestQuality + 1] * mix;
}

double ScoreJPEG(double butteraugli_distance, int size, double butteraugli_target) {
  const double kScale = 50, kMaxExponent = 10, kLargeSize = 1e30;
  const double diff = butteraugli_distance - butteraugli_target;
  if (diff <= 0.0) return size;
  const double exponent = kScale * diff;
  if (exponent > kMaxExponent) return kLargeSize * std::exp(kMaxExponent) * diff + size;
  return std::exp(exponent) * size;
}

namespace {

typedef int QuantMatrix[3][64];

struct Stopwatch {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double lap() {
    const auto t1 = std::chrono::steady_clock::now();
    const double s = std::chrono::duration<double>(t1 - t0).count();
    t0 = t1;
    return s;
  }
};

// ------------------------------------------------------------ quant matrix bisection --
// QuantMatrixGenerator (processor.cc:194-296): a 1-D family of matrices indexed by a
// "heuristic score"; bracket a passing (a) and a failing (b) score, then bisect.
double Csf(int k) { return 1.0 / (1.0 + kZigZagOrder[k] / 2.0); }

double HeuristicScore(const QuantMatrix q) {   // QuantMatrixHeuristicScore, :182-190
  double score = 0.0;
  for (int c = 0; c < 3; ++c)
    for (int k = 0; k < 64; ++k) score += 0.5 * (q[c][k] - 1.0) * Csf(k);
  return score;
}

bool SameMatrix(const QuantMatrix a, const QuantMatrix b) {
  return memcmp(a, b, sizeof(QuantMatrix)) == 0;
}

struct Trial {
  QuantMatrix q;
  size_t jpg_size;
  bool dist_ok;
};

class MatrixSearch {
 public:
  explicit MatrixSearch(bool downsample) : downsample_(downsample), lo_(-1.0), hi_(-1.0), total_(0.0) {
    for (int k = 0; k < 64; ++k) total_ += 3.0 * Csf(k);
  }
  bool Next(QuantMatrix q) {
    for (int guard = 0; guard < 1000; ++guard) {
      double h;
      if (hi_ == -1.0) {
        if (lo_ == -1.0) {
          h = downsample_ ? 0.0 : total_;
        } else {
          h = lo_ < 5.0 * total_ ? lo_ + total_ : 2 * (lo_ + total_);
        }
        if (h > 100 * total_) return false;   // nothing creates enough error
      } else if (hi_ == 0.0) {
        return false;
      } else if (lo_ == -1.0) {
        h = 0.0;
      } else {
        QuantMatrix lower, upper;
        const double eps = 0.05;
        FromScore((1 - eps) * lo_ + eps * 0.5 * (lo_ + hi_), lower);
        FromScore((1 - eps) * hi_ + eps * 0.5 * (lo_ + hi_), upper);
        if (SameMatrix(lower, upper)) return false;
        h = (lo_ + hi_) * 0.5;
      }
      FromScore(h, q);
      bool seen = false;
      for (size_t i = 0; i < tried_.size(); ++i) {
        if (SameMatrix(q, tried_[i].q)) {
          if (tried_[i].dist_ok) lo_ = h; else hi_ = h;
          seen = true;
          break;
        }
      }
      if (!seen) return true;
    }
    return false;
  }
  void Add(const Trial& t) {
    tried_.push_back(t);
    const double h = HeuristicScore(t.q);
    if (t.dist_ok) lo_ = std::max(lo_, h);
    else hi_ = hi_ == -1.0 ? h : std::min(hi_, h);
  }

 private:
  void FromScore(double score, QuantMatrix q) const {   // :269-279
    const int level = static_cast<int>(score / total_);
    score -= level * total_;
    for (int k = 63; k >= 0; --k) {
      const int nat = kNaturalOrder[k];
      for (int c = 0; c < 3; ++c) q[c][nat] = 2 * level + (score > 0.0 ? 3 : 1);
      score -= 3.0 * Csf(nat);
    }
  }
  const bool downsample_;
  double lo_, hi_, total_;
  std::vector<Trial> tried_;
};

inline int16_t QuantizeCoeff(int16_t raw, int quant) {   // quantize.h:24-29
  const int r = raw % quant;
  const int16_t delta = (int16_t)(2 * r > quant ? quant - r : (-2) * r > quant ? -quant - r : -r);
  return (int16_t)(raw + delta);
}

// "precious" coefficients are never zeroed (processor.cc:722-733): (0,1) and (1,0) of a
// block whose original value is at least 4 (8 when the block has much high-frequency energy).
inline bool IsPrecious(const int16_t* orig_blk, int k) {
  if (k != 1 && k != 8) return false;
  double sum_of_hf = 0;
  for (int ii = 3; ii < 64; ++ii) {
    if ((ii & 7) < 3 && ii < 3 * 8) continue;
    sum_of_hf += std::abs(orig_blk[ii]);
  }
  const int limit = sum_of_hf < 60 ? 4 : 8;
  return std::abs(orig_blk[k]) >= limit;
}

// The device-resident global candidate order as LazySorted's back end.
struct DeviceOrder : RangeDevice {
  explicit DeviceOrder(gz_ctx* c) : ctx(c) {}
  // Partitions the device has already made on its own (gz_order_descend*: the quick-select
  // descent towards the position phase B needs, enqueued behind the order's construction), in
  // the order LazySorted is going to ask for them: (lo, hi, cut) triples.
  uint64_t log[3 * 12];
  int log_n = 0, log_next = 0;
  bool Partition(size_t lo, size_t hi, size_t* cut) override {
    if (log_next < log_n && log[3 * log_next] == lo && log[3 * log_next + 1] == hi) {
      *cut = (size_t)log[3 * log_next + 2];
      ++log_next;
      ++n_replayed;
      return true;
    }
    if (log_next < log_n) {   // the device went another way than the host: its array is not what we think
      rc = GZ_E_STATE;
      return false;
    }
    Stopwatch w;
    uint64_t c64 = 0;
    rc = gz_order_partition(ctx, lo, hi, &c64);
    *cut = (size_t)c64;
    t_partition += w.lap();
    ++n_partition;
    return rc == GZ_OK;
  }
  // After a descent everything below the end of the range that holds the wanted position is
  // going to be fetched, range by range; one copy brings it all (to where Fetch would put it).
  bool Prefetch(size_t hi, void* base) {
    Stopwatch w;
    rc = gz_order_fetch(ctx, 0, hi, base);
    t_fetch += w.lap();
    n_fetched += (long)hi;
    if (rc == GZ_OK) have_hi = hi;
    return rc == GZ_OK;
  }
  size_t have_hi = 0;   // entries [0, have_hi) are on the host already
  bool Fetch(size_t lo, size_t hi, void* dst) override {
    if (hi <= have_hi) return true;
    Stopwatch w;
    rc = gz_order_fetch(ctx, lo, hi, dst);
    t_fetch += w.lap();
    n_fetched += (long)(hi - lo);
    return rc == GZ_OK;
  }
  gz_ctx* ctx;
  int rc = GZ_OK;
  double t_partition = 0, t_fetch = 0;   // seconds inside the device calls (round trips included)
  long n_partition = 0, n_fetched = 0, n_replayed = 0;
};

// ------------------------------------------------------------------- the encoder ------
// The large host arrays of one encode (60 MB at 1080p, 240 MB at 4K).  Fresh std::vectors of
// this size come from mmap and are paid for in page faults (~15 ms per 1080p encode, measured
// as the gap between the phase timers and the wall clock of a step): a thread keeps them
// between encodes instead, and an Encoder borrows them for its lifetime.
struct HostScratch {
  std::vector<int16_t> orig, img;
  std::vector<uint8_t> cand_idx;
  std::vector<int32_t> cand_off;
  std::vector<std::pair<int, float> > order;
  std::vector<uint8_t> scan;
};
static HostScratch& ThreadScratch() {
  static thread_local HostScratch s;
  return s;
}

// Helper threads for the size model's code refreshes (code_refresh.h) of ONE encode: GZ_CODE_THREADS,
// else by the cores the process may run on per encode in flight (batch mode: one Encoder per image in
// flight): 3 from eight cores on (4K: 0.271 s with none, 0.267 with 1, 0.258 with 2, 0.254 with 3;
// profiles/r05_chain_experiments.log, section 12), else one per core beyond the encode's own, at most 2.
static std::atomic<int> g_live_encoders{0};

// The host driver's switches, read from the environment ONCE per encode (Encoder's constructor) -- never inside
// the search loops, which run on several encoder threads at once in batch mode.  All of them are test / A-B
// switches: the defaults are the product.
struct HostKnobs {
  int code_threads = -1;              // GZ_CODE_THREADS: helper threads of the code refreshes (-1: by the cores)
  size_t parallel_count_min = (size_t)1 << 20;   // GZ_PARALLEL_COUNT_MIN: step counts on the worker pool from this many entries on
  long code_serial_steps = 30;        // GZ_CODE_SERIAL_STEPS: serial steps before the helpers are called in (0: at once)
  bool step_prefetch = true;          // GZ_STEP_PREFETCH=0: no cache-line prefetches ahead of the serial steps
  bool check_mirror = false;          // GZ_CHECK_MIRROR: host mirror against the device image after every search
  int verify_level = 0;               // GZ_VERIFY_ENTROPY=1|2: every candidate also through the host writer
  long order_device_threshold = -1;   // GZ_ORDER_DEVICE_THRESHOLD: ranges above this are partitioned on the device
  static HostKnobs FromEnvironment() {
    HostKnobs k;
    if (const char* e = getenv("GZ_CODE_THREADS")) k.code_threads = std::max(0, std::min(4, atoi(e)));
    if (const char* e = getenv("GZ_PARALLEL_COUNT_MIN")) k.parallel_count_min = (size_t)atol(e);
    if (const char* e = getenv("GZ_CODE_SERIAL_STEPS")) k.code_serial_steps = atol(e) / 10 * 10;
    if (const char* e = getenv("GZ_STEP_PREFETCH")) k.step_prefetch = atoi(e) != 0;
    k.check_mirror = getenv("GZ_CHECK_MIRROR") != nullptr;
    if (const char* e = getenv("GZ_VERIFY_ENTROPY")) k.verify_level = std::max(1, atoi(e));
    if (const char* e = getenv("GZ_ORDER_DEVICE_THRESHOLD")) k.order_device_threshold = std::max(16L, atol(e));
    return k;
  }
};

// Helper threads for ONE encode's code refreshes, by the cores the process may use per encode in flight
// (evaluated at every search: in batch mode the first encoder of a batch is alone for a moment).
static int CodeRefreshThreads(const HostKnobs& knobs) {
  if (knobs.code_threads >= 0) return knobs.code_threads;
  const int per_encode = WorkerPool::AllowedCpus() / std::max(1, g_live_encoders.load());
  if (per_encode >= 8) return 3;
  return std::max(0, std::min(2, per_encode - 1));
}

// phase B's order entries compare by key alone (processor.cc:675-678); lazy_sort.h's AVX2 pass knows the layout
struct OrderKeyLess {
  enum { float_second_key = 1 };
  bool operator()(const std::pair<int, float>& a, const std::pair<int, float>& b) const {
    return a.second < b.second;
  }
};
typedef LazySorted<std::pair<int, float>, OrderKeyLess> SortedOrder;

class Encoder {
 public:
  Encoder(const Params& p, ProcessStats* s) : params_(p), stats_(s), knobs_(HostKnobs::FromEnvironment()) {
    ++g_live_encoders;
    HostScratch& sc = ThreadScratch();
    orig_.swap(sc.orig);
    img_.swap(sc.img);
    cand_idx_.swap(sc.cand_idx);
    cand_off_.swap(sc.cand_off);
    order_.swap(sc.order);
    scan_.swap(sc.scan);
  }
  ~Encoder() {
    refreshers_.reset();
    --g_live_encoders;
    if (ctx_) gz_destroy(ctx_);
    HostScratch& sc = ThreadScratch();
    orig_.swap(sc.orig);
    img_.swap(sc.img);
    cand_idx_.swap(sc.cand_idx);
    cand_off_.swap(sc.cand_off);
    order_.swap(sc.order);
    scan_.swap(sc.scan);
  }
  bool Run(const std::vector<uint8_t>& rgb, int w, int h, std::string* out);
  bool RunJpeg(const std::string& jpeg_data, std::string* out);

 private:
  void Log(const char* fmt, ...) __attribute__((format(printf, 2, 3)));
  void LogMatrix(const QuantMatrix q);
  bool Fail(const char* what, int rc);
  // SaveToJpegData + WriteJpeg of the current image: marker segments and Huffman codes
  // on the host from the symbol statistics, the scan on the device.  *size = jpg.size().
  bool DeviceHistograms(const QuantMatrix q, SymbolHistogram* dc, SymbolHistogram* ac, int ncomp = 3);
  bool SetFrame(int factor);
  const char* FrameStr() const { return fac_ == 2 ? "f112222" : "f111111"; }   // OutputImage::FrameTypeStr
  size_t Pos(int c, int block, int k) const { return ((size_t)coff_[c] + block) * 64 + k; }
  bool Serialize(const int (*q)[64], const SymbolHistogram* dc, const SymbolHistogram* ac,
                 size_t* size);
  bool SerializeBegin(const int (*q)[64], const SymbolHistogram* dc, const SymbolHistogram* ac);
  bool SerializeEnd(const int (*q)[64], size_t* size);
  bool PrepareHead(const int (*q)[64], const SymbolHistogram* dc, const SymbolHistogram* ac);
  bool ScanBegin();
  size_t SizeLowerBound() const;
  bool CompareBegin();
  bool CompareCurrent();
  bool MaybeOutput(size_t size);
  bool VerifyAgainstHostWriter(const int (*q)[64], size_t size);
  bool DistanceOK(double target_mul) const { return distance_ <= target_mul * params_.butteraugli_target; }
  bool TryMatrix(float target_mul, const QuantMatrix q, Trial* t);
  bool SelectMatrix(QuantMatrix best, bool downsample, bool* dist_ok);
  bool SelectFrequencyMasking(int comp_mask, double target_mul, bool stop_early, bool last_search_of_round);
  // ... and its pieces (see the comment above SelectFrequencyMasking's definition)
  struct MaskSearch;
  struct Iteration;
  bool SearchBlocks(MaskSearch* ms, Stopwatch* sw);
  void RecountRawBits(MaskSearch* ms);
  void SettleBlock(MaskSearch* ms, int b);
  void SettleAll(MaskSearch* ms);
  bool AcquireOrder(MaskSearch* ms, Iteration* it);
  bool BulkSteps(MaskSearch* ms, Iteration* it, SortedOrder* sorted, DeviceOrder* dev_order, size_t fast_until);
  void ApplyStep(MaskSearch* ms, Iteration* it, SortedOrder* sorted, size_t i);
  bool StepsAsTheReference(MaskSearch* ms, Iteration* it, SortedOrder* sorted, size_t from, size_t to);
  void StepsWithHelpers(MaskSearch* ms, Iteration* it, SortedOrder* sorted, size_t base);
  bool EvaluateCandidate(MaskSearch* ms, Iteration* it, Stopwatch* sw);
  bool SetImageFromQuantization(const QuantMatrix q, bool download);
  // The tables of a frame of this image: quant matrices q, or null for the "original" (the
  // q = 1 frame of EncodeRGBToJpeg, or the input JPEG's own tables), plus the metadata a JPEG
  // input carries into every output.
  void Tables(const int (*q)[64], int ncomp, Frame* f) const;
  bool Search(const QuantMatrix first_q, std::string* out);   // ProcessJpegData from :826 on

  Params params_;
  ProcessStats* stats_;
  const HostKnobs knobs_;   // the environment's switches, read once
  gz_ctx* ctx_ = nullptr;
  int w_ = 0, h_ = 0, bw_ = 0, bh_ = 0, nb_ = 0;
  // the frame (OutputImage's component layout): chroma factor 1 (4:4:4) or 2 (4:2:0), blocks
  // per chroma component, first block of every component in orig_ / img_, blocks in total
  int fac_ = 1, cbw_ = 0, cbh_ = 0, nbc_ = 0, coff_[3] = {0, 0, 0}, nblk_ = 0;
  int jpg_ncomp_ = 3;            // jpg.components.size() of the round (1: greyscale after Downsample)
  size_t best_size_ = 0;         // final_output_->jpeg_data.size()
  std::string best_full_;        // a best candidate written on the host (the 4:2:0 input as read)
  bool best_on_host_ = false;
  std::vector<int16_t> orig_;    // unquantised coefficients (JPEGData of EncodeRGBToJpeg)
  std::vector<int16_t> img_;     // coefficients of the working image (OutputImage::coeffs_)
  // phase A's CSR arrays, the fetched part of phase B's order, the winner's scan (borrowed
  // from the thread's HostScratch like orig_ / img_; every element that is read was written
  // by this encode)
  std::vector<uint8_t> cand_idx_;
  std::vector<int32_t> cand_off_;
  std::vector<std::pair<int, float> > order_;
  std::vector<uint8_t> scan_;
  QuantMatrix quant_;            // its quant matrices
  float distance_ = 0.0f;        // ButteraugliComparator::distance_
  JpegHead head_;                // marker segments + codes of the last Serialize
  std::string best_head_;        // GuetzliOutput: head of the best candidate; its scan is
                                 // kept on the device (gz_jpeg_scan_keep)
  bool jpeg_input_ = false;      // Process(jpeg_data): tables / metadata of the input below
  FrameMeta meta_;
  Frame in_frame_;               // a 4:2:0 input as read (its own padding blocks)
  std::vector<QuantTable> in_quant_;
  int in_quant_idx_[3] = {0, 0, 0};
  int in_comp_id_[3] = {0, 1, 2};
  QuantMatrix q_in_;             // the input's quantisation per component (processor.cc:84-97)
  bool verify_ = false;
  int verify_level_ = 0;   // GZ_VERIFY_ENTROPY's value: 2 = check the size bound's own (late-scan) path
  bool mirror_valid_ = false;    // img_ mirrors the device image (phase B)          // GZ_VERIFY_ENTROPY=1: cross-check against the host writer
  double best_score_ = -1;
  double t_write_ = 0, t_compare_ = 0, t_quant_ = 0, t_blocksearch_ = 0, t_phaseb_ = 0,
         t_upload_ = 0;
  double t_pb_ensure_ = 0, t_pb_fast_ = 0, t_pb_dev_partition_ = 0, t_pb_dev_fetch_ = 0;
  double t_fs_count_ = 0, t_fs_apply_ = 0, t_fs_mirror_ = 0, t_fs_delta_ = 0, t_fs_rest_ = 0;
  long n_dev_partitions_ = 0, n_dev_fetched_ = 0, n_dev_replayed_ = 0;
  double t_pb_descend_ = 0;
  long n_dev_exported_ = 0;
  int descend_levels_ = 6;      // levels to enqueue per descent (follows what the orders need)
  long n_fast_ = 0;
  long n_scans_ = 0, n_scans_skipped_ = 0;   // candidates entropy-coded / known to lose without it
  uint64_t head_bits_ = 0;                   // scan bits of the candidate head_ was built for
  double t_pb_order_ = 0, t_pb_sort_ = 0, t_pb_loop_ = 0, t_pb_codes_ = 0;
  long n_steps_ = 0, n_order_ = 0, n_evaluations_ = 0;
  // where the host's time goes at the end of an iteration (stats_->timers)
  double t_head_ = 0, t_cmp_begin_ = 0, t_cmp_end_ = 0, t_scan_begin_ = 0, t_scan_end_ = 0, t_ahead_begin_ = 0;
  size_t device_threshold_ = 1 << 16;   // ranges above this are partitioned on the device (32-64 K measured best at 1080p and 4K)
  // the size model's code refreshes on helper threads (code_refresh.h); null: on this thread
  std::unique_ptr<CodeRefreshers> refreshers_;
  // one serial step of phase B as it was taken, so that it can be priced later and undone
  struct SlowStep {
    int32_t b, pos;
    float val;
    int16_t old_val;
    uint8_t changed, first_touch, comp, nsym;
    int16_t sym[kMaxCoeffACSymbolChanges];
  };
  std::vector<SlowStep> slow_log_;
  long slow_steps_last_ = 0;           // serial steps of the previous iteration of phase B
  std::vector<int32_t> bulk_counts_;   // (kept between iterations: no allocation on the host's path)
  long n_steps_undone_ = 0;
};

// What one SelectFrequencyMasking search keeps between its iterations.
struct Encoder::MaskSearch {
  int comp_mask = 7;
  double target_mul = 1.0;
  int factor = 1, nb = 0, ncomp = 3;   // the grid of the mask's last component (:548-552), the frame's components
  // the size model (:592-604): symbol statistics, the parts of the estimate, the AC codes' depths and the
  // raw bits HistogramRawBits gives for them (kept incrementally between refreshes)
  SymbolHistogram dc_histo[3], ac_histo[3];
  int header_size = 0, dc_size = 0, ac_header = 0, base_size = 0, prev_size = 0;
  std::vector<uint8_t> ac_depths;
  int64_t ac_raw_bits[3] = {0, 0, 0};
  std::vector<int> next_cand;     // last_indexes: how far every block has advanced
  std::vector<int> mirror_cand;   // ... and up to where the host mirror img_ has followed (SettleBlock)
  std::vector<int32_t> edit_pos;  // coefficient changes of one iteration
  std::vector<int16_t> edit_val;
  std::vector<char> touched;      // blocks changed in this iteration (listed in `dirty`)
  std::vector<int32_t> dirty, first_touch;
  std::vector<int> step_count;
  bool first_up = true;
  // the next iteration's order, constructed on the device behind the evaluation (EvaluateCandidate): the
  // direction it was built for (0: none in flight) and the partitions the device made behind it
  int ahead = 0;
  uint64_t ahead_log[3 * 12];
  int ahead_levels = 0;
  uint64_t ahead_last = 0;
};

// One pass of the reference's loop body (:607-772).
struct Encoder::Iteration {
  int direction = 1;
  float below_limit = 0.0f, per_block = 0.0f;
  uint64_t total = 0, below = 0;        // the order's size; its keys below the limit (first "up" iteration)
  int blocks_to_change = 0;
  bool have_ahead_log = false;          // the device's descent came with the order
  std::pair<int, float>* order = nullptr;   // the context's page-locked mirror of the order's fetched ranges
  size_t n_order = 0;
  double min_size_delta = 0.0;
  int min_coeffs_to_change = 0;
  float val_threshold = 0.0f;
  int changed_coeffs = 0;
  int est_size = 0;
  bool verify_failed = false;
};

void Encoder::Log(const char* fmt, ...) {   // GUETZLI_LOG / PrintDebug, debug_print.h
  if (!stats_->debug_output && !stats_->debug_output_file) return;
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (stats_->debug_output) stats_->debug_output->append(buf);
  if (stats_->debug_output_file) fputs(buf, stats_->debug_output_file);
}

void Encoder::LogMatrix(const QuantMatrix q) {   // GUETZLI_LOG_QUANT
  for (int y = 0; y < 8; ++y) {
    for (int c = 0; c < 3; ++c) {
      for (int x = 0; x < 8; ++x) Log(" %2d", q[c][8 * y + x]);
      Log("   ");
    }
    Log("\n");
  }
}

bool Encoder::Fail(const char* what, int rc) {
  fprintf(stderr, "guetzli_amd: %s failed: %s (%s)\n", what, gz_strerror(rc),
          ctx_ ? gz_last_error(ctx_) : "");
  return false;
}

bool Encoder::SetFrame(int factor) {
  fac_ = factor;
  cbw_ = (w_ + 8 * factor - 1) / (8 * factor);
  cbh_ = (h_ + 8 * factor - 1) / (8 * factor);
  nbc_ = cbw_ * cbh_;
  coff_[0] = 0;
  coff_[1] = nb_;
  coff_[2] = nb_ + nbc_;
  nblk_ = nb_ + 2 * nbc_;
  return true;
}

void Encoder::Tables(const int (*q)[64], int ncomp, Frame* f) const {
  FrameTablesFactor(q, w_, h_, ncomp, fac_, f);
  if (!jpeg_input_) return;
  f->meta = &meta_;
  if (q == nullptr) {   // jpg_in as read: its own DQT tables in file order, its component ids
    f->quant = in_quant_;
    for (int c = 0; c < 3; ++c) {
      f->quant_idx[c] = in_quant_idx_[c];
      f->comp_id[c] = in_comp_id_[c];
    }
  }
}

static bool ChromaAllZero(const SymbolHistogram* dc, const SymbolHistogram* ac) {
  // all DC differences zero (so every DC is zero) and nothing but end-of-block in AC
  for (int c = 1; c < 3; ++c)
    for (int i = 1; i + 1 < kHistoSize; ++i)
      if (dc[c].counts[i] || ac[c].counts[i]) return false;
  return true;
}

// BuildDCHistograms + BuildACHistograms of the frame SaveToJpegData would write.  In a 4:2:0
// frame the luma statistics depend on whether chroma is written at all (MCU order and padding
// blocks, or luma alone in raster order): asked for three components first, and again for one
// if both chroma components turn out to be all zero.
bool Encoder::DeviceHistograms(const QuantMatrix q, SymbolHistogram* dc, SymbolHistogram* ac, int ncomp) {
  std::vector<uint32_t> counts(2 * 3 * 256);
  const int rc = gz_jpeg_histograms_ncomp(ctx_, &q[0][0], ncomp, counts.data());
  if (rc != GZ_OK) return Fail("gz_jpeg_histograms", rc);
  for (int c = 0; c < 3; ++c) {
    dc[c].Clear();
    ac[c].Clear();
    for (int i = 0; i < 256; ++i) {
      dc[c].Add(i, (int)counts[(0 * 3 + c) * 256 + i]);
      ac[c].Add(i, (int)counts[(1 * 3 + c) * 256 + i]);
    }
  }
  if (ncomp == 3 && fac_ == 2 && ChromaAllZero(dc, ac)) return DeviceHistograms(q, dc, ac, 1);
  return true;
}

bool Encoder::Serialize(const int (*q)[64], const SymbolHistogram* dc, const SymbolHistogram* ac,
                        size_t* size) {
  return SerializeBegin(q, dc, ac) && SerializeEnd(q, size);
}

// Serialize in two halves.  _Begin: the marker segments and Huffman codes on the host, then the
// scan enqueued on the context's entropy stream (gz_jpeg_scan_begin) -- the caller goes on
// enqueueing (the next order's construction) while the entropy coder runs beside the
// evaluation.  _End collects the scan's length.
bool Encoder::SerializeBegin(const int (*q)[64], const SymbolHistogram* dc, const SymbolHistogram* ac) {
  return PrepareHead(q, dc, ac) && ScanBegin();
}

// The host's half of a candidate's JPEG: marker segments and Huffman codes (head_), and -- from the
// same symbol statistics -- the exact number of bits of its scan: every symbol occurrence costs its
// code length plus its extra bits (the low nibble of an AC symbol, the category of a DC symbol:
// jpeg_data_writer.cc:446-497), so the scan is head_bits_ long before a single bit is written.
bool Encoder::PrepareHead(const int (*q)[64], const SymbolHistogram* dc, const SymbolHistogram* ac) {
  Stopwatch sw;
  Frame f;
  // a single component is written when both chroma planes are entirely zero
  // (OutputImage::SaveToJpegData, output_image.cc:348-409); the q=1 original always has 3
  const int nc = q && ChromaAllZero(dc, ac) ? 1 : 3;
  Tables(q, nc, &f);
  SymbolHistogram dc1[3], ac1[3];
  if (nc == 1 && fac_ == 2) {
    // luma alone is written in raster order without padding blocks: its statistics are not
    // those of the 4:2:0 MCU order the caller may hold (chroma that became all zero during
    // the search); recounted
    if (!DeviceHistograms(q, dc1, ac1, 1)) return false;
    dc = dc1;
    ac = ac1;
  }
  if (!BuildJpegHead(f, dc, ac, &head_)) return Fail("BuildJpegHead", GZ_E_STATE);
  head_bits_ = 0;
  for (int c = 0; c < head_.ncomp; ++c)
    head_bits_ += (uint64_t)HistogramRawBits(dc[c], head_.depth[0][c]) +
                  (uint64_t)HistogramRawBits(ac[c], head_.depth[1][c]);
  { const double d = sw.lap(); t_write_ += d; t_head_ += d; }
  return true;
}

bool Encoder::ScanBegin() {
  Stopwatch sw;
  const int rc = gz_jpeg_scan_begin(ctx_, head_.ncomp, &head_.depth[0][0][0], &head_.code[0][0][0]);
  { const double d = sw.lap(); t_write_ += d; t_scan_begin_ += d; }
  if (rc != GZ_OK) return Fail("gz_jpeg_scan", rc);
  return true;
}

// What the candidate's JPEG weighs at least: its head, its scan's bits as bytes -- the 0x00 stuffed
// behind every 0xFF byte of the scan (jpeg_bit_writer.h:62-70) only adds to that -- and EOI.
size_t Encoder::SizeLowerBound() const {
  size_t size = head_.bytes.size() + (size_t)((head_bits_ + 7) / 8) + 2;
  if (jpeg_input_ && !meta_.strip) size += meta_.tail_data.size();
  return size;
}
bool Encoder::SerializeEnd(const int (*q)[64], size_t* size) {
  Stopwatch sw;
  uint64_t scan_bytes = 0;
  const int rc = gz_jpeg_scan_end(ctx_, &scan_bytes);
  if (rc != GZ_OK) return Fail("gz_jpeg_scan", rc);
  *size = head_.bytes.size() + (size_t)scan_bytes + 2;   // + EOI
  if (jpeg_input_ && !meta_.strip) *size += meta_.tail_data.size();
  { const double d = sw.lap(); t_write_ += d; t_scan_end_ += d; }
  ++n_scans_;
  if (verify_) {   // GZ_VERIFY_ENTROPY: the bit count derived from the statistics is the coder's
    uint64_t bits = 0, ff = 0;
    const int rb = gz_jpeg_scan_bits(ctx_, &bits, &ff);
    if (rb != GZ_OK) return Fail("gz_jpeg_scan_bits", rb);
    if (bits != head_bits_ || (bits + 7) / 8 + ff != scan_bytes || SizeLowerBound() > *size) {
      fprintf(stderr, "guetzli_amd: scan of %llu bits (+%llu stuffed bytes), the symbol statistics say %llu\n",
              (unsigned long long)bits, (unsigned long long)ff, (unsigned long long)head_bits_);
      return false;
    }
  }
  if (verify_ && !VerifyAgainstHostWriter(q, *size)) return false;
  return true;
}

// Test hook (GZ_VERIFY_ENTROPY=1): the device scan + host head must equal the serial host
// writer on the same coefficients, byte for byte.
bool Encoder::VerifyAgainstHostWriter(const int (*q)[64], size_t size) {
  std::vector<int16_t> co((size_t)nblk_ * 64);
  int rc = gz_get_coeffs(ctx_, co.data());
  if (rc != GZ_OK) return Fail("gz_get_coeffs", rc);
  if (q && mirror_valid_ && memcmp(co.data(), img_.data(), co.size() * 2) != 0) {
    size_t nd = 0, first = 0;
    for (size_t i = 0; i < co.size(); ++i)
      if (co[i] != img_[i]) { if (!nd) first = i; ++nd; }
    fprintf(stderr, "guetzli_amd: host mirror of the image differs from the device image "
            "(%zu coefficients, first at %zu: device %d host %d)\n", nd, first, co[first], img_[first]);
    return false;
  }
  Frame f;
  if (q) {
    FrameFromImageFactor(co.data(), q, w_, h_, fac_, &f);
  } else if (jpeg_input_) {   // the input as read: quantised by its own tables
    FrameFromImageFactor(co.data(), q_in_, w_, h_, fac_, &f);
    if (f.ncomp != 3) return true;   // (all-zero chroma in the input: not comparable this way)
    f.quant = in_quant_;
    for (int c = 0; c < 3; ++c) {
      f.quant_idx[c] = in_quant_idx_[c];
      f.comp_id[c] = in_comp_id_[c];
    }
  } else {
    FrameFromOriginal(co.data(), w_, h_, &f);
  }
  if (jpeg_input_) f.meta = &meta_;
  std::string ref;
  WriteJpeg(f, &ref);
  std::string got = head_.bytes;
  std::vector<uint8_t> scan(3 * co.size() + 1024);
  size_t n = 0;
  rc = gz_jpeg_scan_bytes(ctx_, 0, scan.data(), scan.size(), &n);
  if (rc != GZ_OK) return Fail("gz_jpeg_scan_bytes", rc);
  got.append((const char*)scan.data(), n);
  got.push_back((char)0xff);
  got.push_back((char)0xd9);
  if (jpeg_input_ && !meta_.strip) got.append(meta_.tail_data);
  if (got != ref || got.size() != size) {
    fprintf(stderr, "guetzli_amd: device entropy coder mismatch (device %zu/%zu bytes, host %zu)\n",
            got.size(), size, ref.size());
    return false;
  }
  return true;
}

// comparator_->Compare(*img) in two halves: the evaluation is enqueued before the candidate's
// Huffman codes are built on the host (Serialize) and collected afterwards.
bool Encoder::CompareBegin() {
  Stopwatch sw;
  const int rc = gz_compare_begin(ctx_);
  { const double d = sw.lap(); t_compare_ += d; t_cmp_begin_ += d; }
  if (rc != GZ_OK) return Fail("gz_compare_begin", rc);
  return true;
}
bool Encoder::CompareCurrent() {
  Stopwatch sw;
  const int rc = gz_compare_end(ctx_, &distance_);
  { const double d = sw.lap(); t_compare_ += d; t_cmp_end_ += d; }
  if (rc != GZ_OK) return Fail("gz_compare_end", rc);
  Log(" BA[100.00%%] D[%6.4f]", distance_);
  return true;
}

bool Encoder::MaybeOutput(size_t size) {   // processor.cc:139-148
  const double score = ScoreJPEG(distance_, (int)size, params_.butteraugli_target);
  Log(" Score[%.4f]", score);
  if (score < best_score_ || best_score_ < 0) {
    best_head_ = head_.bytes;
    const int rc = gz_jpeg_scan_keep(ctx_);
    if (rc != GZ_OK) return Fail("gz_jpeg_scan_keep", rc);
    best_score_ = score;
    best_size_ = size;
    best_on_host_ = false;
    Log(" (*)");
  }
  Log("\n");
  return true;
}

// img := orig, then ApplyGlobalQuantization(q); device and host copies.
bool Encoder::SetImageFromQuantization(const QuantMatrix q, bool download) {
  Stopwatch sw;
  const int rc = gz_quantize(ctx_, &q[0][0], download ? img_.data() : nullptr);
  t_quant_ += sw.lap();
  if (rc != GZ_OK) return Fail("gz_quantize", rc);
  memcpy(quant_, q, sizeof(QuantMatrix));
  return true;
}

bool Encoder::TryMatrix(float target_mul, const QuantMatrix q, Trial* t) {   // :298-326
  memcpy(t->q, q, sizeof(QuantMatrix));
  if (!SetImageFromQuantization(q, false)) return false;
  SymbolHistogram dc[3], ac[3];
  size_t size = 0;
  if (!DeviceHistograms(q, dc, ac) || !CompareBegin() || !Serialize(q, dc, ac, &size)) return false;
  Log("Iter %2d: %s quantization matrix:\n", stats_->counters[kNumItersCnt] + 1, FrameStr());
  LogMatrix(q);
  Log("Iter %2d: %s GQ[%5.2f] Out[%7zd]", stats_->counters[kNumItersCnt] + 1, FrameStr(),
      HeuristicScore(q), size);
  ++stats_->counters[kNumItersCnt];
  if (!CompareCurrent()) return false;
  t->dist_ok = DistanceOK(target_mul);
  t->jpg_size = size;
  return MaybeOutput(size);
}

bool Encoder::SelectMatrix(QuantMatrix best_q, bool downsample, bool* dist_ok) {   // SelectQuantMatrix, :328-360
  MatrixSearch search(downsample);
  const float target_mul_high = 0.97f, target_mul_low = 0.95f;
  Trial best;
  if (!TryMatrix(target_mul_high, best_q, &best)) return false;
  for (;;) {
    QuantMatrix next;
    if (!search.Next(next)) break;
    Trial t;
    if (!TryMatrix(target_mul_high, next, &t)) return false;
    search.Add(t);
    const bool better = t.dist_ok != best.dist_ok ? t.dist_ok : t.jpg_size < best.jpg_size;
    if (better) {
      best = t;
      if (t.dist_ok && !DistanceOK(target_mul_low)) break;
    }
  }
  memcpy(best_q, best.q, sizeof(QuantMatrix));
  Log("\n%s selected quantization matrix:\n", downsample ? "YUV420" : "YUV444");
  LogMatrix(best_q);
  *dist_ok = best.dist_ok;
  return true;
}

// ---------------------------------------------------------------------------------------------------
// SelectFrequencyMasking (processor.cc:539-780), in the pieces it is made of (round 6: one 716-line function
// until then).  MaskSearch = what one search keeps between its iterations, Iteration = one pass of the
// reference's loop body (:607-772).  Order of the calls, of the device's entry points and of every decision
// is the reference's; the --verbose trace test and the byte-exact goldens hold it there.
//   SearchBlocks            phase A on the device + the size model of the starting point
//   AcquireOrder            the iteration's global order (built ahead behind the last evaluation, or now)
//   BulkSteps               the steps no estimate can observe: device descent, prefix, per-block counts
//   StepsAsTheReference     the reference's serial loop over [from, to)
//   StepsWithHelpers        the same decisions with the code refreshes on helper threads
//   EvaluateCandidate       edits to the device, evaluation + next order enqueued, exact size only if it can win
//   SettleBlock / SettleAll the host mirror of the image catching up
// last_search_of_round: no other search of this frame follows (its host mirror of the image is not read again)
bool Encoder::SelectFrequencyMasking(int comp_mask, double target_mul, bool stop_early,
                                     bool last_search_of_round) {   // processor.cc:539-780
  Stopwatch sw;
  int last_c = 0;
  for (int c = 0; c < 3; ++c)
    if (comp_mask & (1 << c)) last_c = c;
  if (last_c >= jpg_ncomp_) return true;   // :546-547
  MaskSearch ms;
  ms.comp_mask = comp_mask;
  ms.target_mul = target_mul;
  // the grid of the mask's last component (:548-552)
  ms.factor = last_c > 0 ? fac_ : 1;
  ms.nb = ms.factor == 2 ? nbc_ : nb_;
  ms.ncomp = jpg_ncomp_;
  const int nb = ms.nb;
  if (!refreshers_) {
    const int t = CodeRefreshThreads(knobs_);
    if (t > 0) {
      try {
        refreshers_.reset(new CodeRefreshers(t));
      } catch (const std::exception&) {   // (no thread to be had: the reference's serial loop)
        refreshers_.reset();
      }
    }
  }
  if (!SearchBlocks(&ms, &sw)) return false;

  int rc = gz_order_reset(ctx_);           // max_block_error := 0, kept on the device
  if (rc != GZ_OK) return Fail("gz_order_reset", rc);
  ms.next_cand.assign(nb, 0);     // last_indexes
  ms.mirror_cand.assign(nb, 0);
  ms.touched.assign(nb, 0);
  ms.step_count.assign(nb, 0);

  for (int direction = 1; direction >= -1; direction -= 2) {
    for (;;) {
      if (stop_early && direction == -1) {
        // down-adjusting only makes the output larger (:613-621)
        if (ms.prev_size > 1.01 * (double)best_size_) break;
      }
      Iteration it;
      it.direction = direction;
      it.below_limit = 0.75f * params_.butteraugli_target;   // 0.75f * BlockErrorLimit()
      Stopwatch pw;
      if (!AcquireOrder(&ms, &it)) return false;
      t_pb_order_ += pw.lap();
      if (it.total == 0) break;
      // (an iteration takes about as many serial steps as the one before it: after a long one the helpers
      // are woken now, while the prefix is selected and the bulk steps are applied)
      if (refreshers_ && !verify_ && slow_steps_last_ >= 30) refreshers_->Activate();
      n_order_ += (long)it.total;
      {
        void* mirror = nullptr;
        rc = gz_order_host_mirror(ctx_, it.total, &mirror);
        if (rc != GZ_OK) return Fail("gz_order_host_mirror", rc);
        it.order = static_cast<std::pair<int, float>*>(mirror);
      }
      it.n_order = (size_t)it.total;

      // The reference std::sort-s `order` here (processor.cc:675-678) and then consumes a
      // prefix.  Equal keys occur across different blocks and std::sort is not stable, so
      // the permutation must be libstdc++'s; LazySorted yields exactly that permutation,
      // front first, without sorting the part the scan never reaches.  The partitions of the
      // large ranges run on the device (gz_order_partition); ranges that have become small
      // are fetched and finished here.
      DeviceOrder dev_order(ctx_);
      SortedOrder sorted(it.order, it.n_order, OrderKeyLess(), -1, 1 << 17, &dev_order, device_threshold_);
      t_pb_sort_ += pw.lap();

      double rel_size_delta = direction > 0 ? 0.01 : 0.0005;
      if (direction > 0 && DistanceOK(1.0)) rel_size_delta = 0.05;
      it.min_size_delta = ms.base_size * rel_size_delta;
      it.per_block = direction > 0 ? 2.0f : ms.factor * ms.factor * 0.2f;
      it.min_coeffs_to_change = it.per_block * it.blocks_to_change;
      if (ms.first_up) {
        // partition_point over the sorted sequence == number of keys below the limit
        it.min_coeffs_to_change = std::max<int>(it.min_coeffs_to_change, (int)it.below);
        ms.first_up = false;
      }

      t_pb_sort_ += pw.lap();
      // (touched[] and step_count[] are all zero here: whoever sets an entry records the block in
      // `dirty`, and the entries of the blocks in `dirty` are cleared before the list is)
      for (int32_t b : ms.dirty) { ms.touched[b] = 0; ms.step_count[b] = 0; }
      ms.dirty.clear();
      ms.edit_pos.clear();
      ms.edit_val.clear();
      it.val_threshold = 0.0;
      it.changed_coeffs = 0;
      it.est_size = ms.prev_size;
      // The stopping rule can only fire once changed_coeffs > min_coeffs_to_change, and the
      // size estimate of step i uses the Huffman depths refreshed at the last multiple of 10
      // not above i.  Up to that refresh point nothing the estimate produces is observable,
      // so those steps only edit coefficients ("fast steps"); the symbol statistics are
      // rebuilt once after them.
      {
        const size_t n_order = it.n_order;
        const size_t last_needed = std::min<size_t>((size_t)std::max(it.min_coeffs_to_change, 0), n_order - 1);
        const size_t fast_until = last_needed / 10 * 10;
        if (!BulkSteps(&ms, &it, &sorted, &dev_order, fast_until)) return false;
        n_steps_ += (long)fast_until;
        n_fast_ += (long)fast_until;
        const long slow_steps_before = (long)(n_steps_ - n_fast_);
        // EntropyDataSize(ac_histo, ncomp, ac_depths) after every step, without its pass over the
        // histograms: ac_raw_bits[c] follows HistogramRawBits(ac_histo[c], depths of c) through
        // ApplyStep and is recounted when the depths change
        RecountRawBits(&ms);
        // Three quarters of an encode's iterations stop within ten steps of the bulk (the first
        // refresh's estimate already differs enough): waking the helpers for those costs more than the
        // one refresh they could take over.  An iteration takes about as many serial steps as the one
        // before it: after a short one the first windows are taken as the reference takes them, and
        // the helpers are called in only if the loop goes on.
        size_t base = fast_until;   // first step of the pipelined part (a multiple of 10)
        bool stopped_early = false;
        const long serial_first = knobs_.code_serial_steps;   // (the tests: 0 = helpers from the first step on)
        if (refreshers_ && !verify_ && serial_first > 0 && slow_steps_last_ < serial_first) {
          const size_t to = std::min(n_order, fast_until + (size_t)serial_first);
          stopped_early = StepsAsTheReference(&ms, &it, &sorted, fast_until, to);
          base = to;
        }
        if (refreshers_ && !verify_ && !stopped_early && base < n_order) {
          StepsWithHelpers(&ms, &it, &sorted, base);
        } else if (!(refreshers_ && !verify_)) {
          (void)StepsAsTheReference(&ms, &it, &sorted, fast_until, n_order);
          if (it.verify_failed) return false;
        }
        slow_steps_last_ = (long)(n_steps_ - n_fast_) - slow_steps_before;
      }
      if (refreshers_) refreshers_->Deactivate();
      t_pb_loop_ += pw.lap();
      if (sorted.failed()) return Fail("gz_order_partition/fetch", dev_order.rc);
      if (dev_order.log_n > 0) {
        // as many levels next time as this order needed, plus one in reserve (an unused level
        // costs two empty launches, a missing one a round trip per partition)
        descend_levels_ = std::min(12, std::max(2, dev_order.log_n + (dev_order.n_partition > 0 ? 2 : 1)));
      }
      n_dev_replayed_ += dev_order.n_replayed;
      t_pb_dev_partition_ += dev_order.t_partition;
      t_pb_dev_fetch_ += dev_order.t_fetch;
      n_dev_partitions_ += dev_order.n_partition;
      n_dev_fetched_ += dev_order.n_fetched;
      rc = gz_order_advance(ctx_, it.val_threshold, direction);   // max_block_error += weight * ...
      if (rc != GZ_OK) return Fail("gz_order_advance", rc);

      ++stats_->counters[kNumItersCnt];
      ++stats_->counters[direction > 0 ? kNumItersUpCnt : kNumItersDownCnt];
      t_phaseb_ += sw.lap();
      if (!EvaluateCandidate(&ms, &it, &sw)) return false;
      ms.prev_size = it.est_size;
      sw.lap();
    }
    // (img_ is exact for the blocks the serial steps visited; the others catch up when they are
    // visited -- or here, if the whole mirror is going to be read)
    if (verify_ || (direction == -1 && !last_search_of_round)) SettleAll(&ms);
  }
  if (knobs_.check_mirror) {
    // self-check of the lazily maintained mirror (the tests): every block caught up now, the host's
    // image must be the device's, coefficient for coefficient
    SettleAll(&ms);
    std::vector<int16_t> co((size_t)nblk_ * 64);
    rc = gz_get_coeffs(ctx_, co.data());
    if (rc != GZ_OK) return Fail("gz_get_coeffs", rc);
    if (memcmp(co.data(), img_.data(), co.size() * sizeof(int16_t)) != 0) {
      fprintf(stderr, "guetzli_amd: the host mirror of the image differs from the device image after the search\n");
      return false;
    }
  }
  return true;
}

// Phase A on the device (ComputeBlockZeroingOrder of every block, :553-590) and the size model of the starting
// point (:592-604).
bool Encoder::SearchBlocks(MaskSearch* msp, Stopwatch* sw) {
  MaskSearch& ms = *msp;
  const int nb = ms.nb, ncomp = ms.ncomp;
  std::vector<int32_t>& cand_off = cand_off_;
  std::vector<uint8_t>& cand_idx = cand_idx_;
  if (cand_off.size() != (size_t)nb + 1) cand_off.resize((size_t)nb + 1);
  if (cand_idx.size() != (size_t)nb * 189) cand_idx.resize((size_t)nb * 189);
  // the candidates' errors stay on the device, where the global order is built from them
  int rc = gz_block_zeroing_orders_masked(ctx_, ms.comp_mask, params_.zeroing_greedy_lookahead,
                                          params_.new_zeroing_model ? 1 : 0, cand_off.data(),
                                          cand_idx.data(), nullptr, nb * 189);
  t_blocksearch_ += sw->lap();
  if (rc != GZ_OK) return Fail("gz_block_zeroing_orders", rc);
  {
    uint64_t ev = 0;
    if (gz_search_evaluations(ctx_, &ev) == GZ_OK) n_evaluations_ += (long)ev;
  }
  // ---- size model of the starting point ----
  {
    if (!DeviceHistograms(quant_, ms.dc_histo, ms.ac_histo)) return false;
    Frame f;
    Tables(quant_, ChromaAllZero(ms.dc_histo, ms.ac_histo) ? 1 : 3, &f);
    ms.header_size = (int)HeaderSize(f);
    SymbolHistogram dcs[3] = {ms.dc_histo[0], ms.dc_histo[1], ms.dc_histo[2]};
    size_t num = f.ncomp;
    int indexes[3];
    uint8_t depths[3 * kHistoSize];
    ms.dc_size = (int)ClusterHistograms(dcs, &num, indexes, depths);   // EstimateDCSize
    // BuildACHistograms fills one histogram per component SaveToJpegData wrote: with all-zero
    // chroma that is luma only, and the other entries of ac_histograms(ncomp) stay empty (:592-600)
    if (f.ncomp == 1) { ms.ac_histo[1].Clear(); ms.ac_histo[2].Clear(); }
  }
  ms.ac_depths.assign(3 * kHistoSize, 0);
  ms.ac_header = (int)EntropyCodes(ms.ac_histo, ncomp, ms.ac_depths.data());
  ms.base_size = ms.header_size + ms.dc_size + ms.ac_header +
                 (int)EntropyDataSize(ms.ac_histo, ncomp, ms.ac_depths.data());
  ms.prev_size = ms.base_size;
  return true;
}

void Encoder::RecountRawBits(MaskSearch* ms) {
  for (int c = 0; c < ms->ncomp; ++c)
    ms->ac_raw_bits[c] = HistogramRawBits(ms->ac_histo[c], &ms->ac_depths[c * kHistoSize]);
}

// The host mirror img_ follows the bulk ("fast") steps lazily: mirror_cand[b] says up to
// which candidate position block b's coefficients in img_ are current.  Only the blocks the
// slow steps touch (a hundred per iteration) need their mirror at once.  What a block looks like
// is a function of how far it has advanced, not of the way there: its candidates are distinct
// coefficients, those below next_cand are zeroed (the precious ones excepted), those from next_cand
// on hold their quantised original values -- so a block catches up in whichever direction it lags,
// also across the turn from "up" to "down", and the rest of the image is brought up to date only
// when somebody reads all of it (GZ_VERIFY_ENTROPY, a second mask's search; on the worker pool):
// 7.5 M pending steps at the turn of a 4K encode, 3.4 ms with the device idle, for blocks most of
// which the serial steps never visit.
void Encoder::SettleBlock(MaskSearch* ms, int b) {
  int m = ms->mirror_cand[b];
  const int n = ms->next_cand[b];
  for (; m < n; ++m) {   // behind: the steps up
    const int idx = cand_idx_[cand_off_[b] + m];
    const int c = idx / 64, k = idx % 64;
    if (!IsPrecious(&orig_[Pos(c, b, 0)], k)) img_[Pos(c, b, k)] = 0;
  }
  for (; m > n; --m) {   // ahead: the steps down
    const int idx = cand_idx_[cand_off_[b] + m - 1];
    const int c = idx / 64, k = idx % 64;
    const int16_t* orig_blk = &orig_[Pos(c, b, 0)];
    const int newval = QuantizeCoeff(orig_blk[k], quant_[c][k]);
    if (!(newval == 0 && IsPrecious(orig_blk, k))) img_[Pos(c, b, k)] = (int16_t)newval;
  }
  ms->mirror_cand[b] = n;
}

void Encoder::SettleAll(MaskSearch* ms) {
  WorkerPool& pool = WorkerPool::Get();
  const int nb = ms->nb;
  const int chunks = nb < 4096 ? 1 : 4 * pool.size();
  const int per = (nb + chunks - 1) / chunks;
  pool.Run(chunks, [&](int ch) {
    for (int b = ch * per; b < std::min(nb, (ch + 1) * per); ++b) SettleBlock(ms, b);
  });
}

// `order` (global_order, processor.cc:622-663) is built on the device from the CSR arrays phase A left
// there, in the reference's sequence: blocks ascending; within a block the remaining candidates ascending
// for "up", the applied ones descending for "down".  The order of the next iteration is constructed on the
// device right behind the evaluation of this iteration's candidate (EvaluateCandidate): ms->ahead says that
// such a construction is in flight, and for which direction.
bool Encoder::AcquireOrder(MaskSearch* ms, Iteration* it) {
  int rc = GZ_OK;
  for (int radius = 1; radius <= 4; ++radius) {
    // block weights (ComputeBlockErrorAdjustmentWeights) and max_block_error stay on the
    // device; the host only supplies how far each block has advanced
    int32_t btc = 0;
    if (radius == 1 && ms->ahead == it->direction && !ms->first_up) {
      rc = gz_order_build_auto_end(ctx_, &it->total, &btc, &it->below);
      if (rc == GZ_OK) {
        rc = gz_order_descend_end(ctx_, ms->ahead_log, 12, &ms->ahead_levels, &ms->ahead_last);
        it->have_ahead_log = rc == GZ_OK && ms->ahead_levels > 0;
      }
    } else {
      rc = gz_order_build_auto(ctx_, it->direction, radius, ms->target_mul, ms->first_up ? 0 : 1,
                               ms->next_cand.data(), ms->first_up ? 1 : 0, it->below_limit, &it->total, &btc,
                               &it->below);
    }
    ms->ahead = 0;
    if (rc != GZ_OK) return Fail("gz_order_build_auto", rc);
    it->blocks_to_change = btc;
    if (it->total != 0) break;
    it->have_ahead_log = false;   // (an empty order: the next radius builds another one)
  }
  return true;
}

// Steps [0, fast_until) of the sorted order: the introsort partitions that lead there on the device, the
// prefix as a set, the steps applied block by block on the device image, their effect on the AC symbol
// statistics from the device.
bool Encoder::BulkSteps(MaskSearch* msp, Iteration* itp, SortedOrder* sortedp, DeviceOrder* dev_orderp, size_t fast_until) {
  MaskSearch& ms = *msp;
  Iteration& it = *itp;
  SortedOrder& sorted = *sortedp;
  DeviceOrder& dev_order = *dev_orderp;
  const int nb = ms.nb, ncomp = ms.ncomp, direction = it.direction;
  const size_t n_order = it.n_order;
  std::pair<int, float>* order = it.order;
  int rc = GZ_OK;
  Stopwatch fw;
  // The introsort partitions that lead to position fast_until - 1, made by the device
  // without the host in between: behind the order's construction when that was enqueued
  // ahead (the device derives the position as the caller does), else now, in one call.
  if (n_order > device_threshold_) {
    const uint64_t want = fast_until ? fast_until - 1 : 0;
    if (it.have_ahead_log) {
      if (ms.ahead_last != want) return Fail("gz_order_descend: position", GZ_E_STATE);
      memcpy(dev_order.log, ms.ahead_log, sizeof(uint64_t) * 3 * ms.ahead_levels);
      dev_order.log_n = ms.ahead_levels;
    } else {
      int levels = 0;
      rc = gz_order_descend(ctx_, want, device_threshold_, descend_levels_, dev_order.log, &levels);
      if (rc != GZ_OK) return Fail("gz_order_descend", rc);
      dev_order.log_n = levels;
    }
    if (dev_order.log_n > 0) {
      // the range the descent ended in, and with it everything SelectPrefix will fetch
      uint64_t flo = 0, fhi = n_order;
      for (int l = 0; l < dev_order.log_n; ++l) {
        const uint64_t cut = dev_order.log[3 * l + 2];
        if (want < cut) fhi = cut; else flo = cut;
      }
      // (only when the descent got there: a range that is still large will be partitioned
      // further on the device, and a copy taken now would be stale)
      if (fhi - flo <= device_threshold_ && fhi <= ((size_t)1 << 19)) {
        // ... unless the device has put exactly that prefix into the mirror already, behind
        // the descent it made ahead (k_desc_export)
        uint64_t exported = 0;
        if (it.have_ahead_log) {
          rc = gz_order_exported(ctx_, &exported);
          if (rc != GZ_OK) return Fail("gz_order_exported", rc);
        }
        if (exported == fhi) {
          dev_order.have_hi = (size_t)fhi;
          ++n_dev_exported_;
        } else if (!dev_order.Prefetch((size_t)fhi, order)) {
          return Fail("gz_order_fetch", dev_order.rc);
        }
      }
    }
    t_pb_descend_ += fw.lap();
  }
  sorted.SelectPrefix(fast_until);   // the set [0, fast_until) and element fast_until - 1
  t_pb_ensure_ += fw.lap();
  // Steps [0, fast_until): only how many steps each block takes matters (the n-th step
  // of a block applies its n-th remaining candidate whatever the key), so they are
  // applied block by block: on the device image by gz_apply_candidate_steps, on the
  // host mirror by the worker pool.
  {
    // first touches go to `dirty` without a branch (which block an entry belongs to is as
    // good as random: the branch mispredicted for a third of the 63 000 entries of a 4K
    // iteration)
    if (ms.first_touch.size() != (size_t)nb + 1) ms.first_touch.resize((size_t)nb + 1);
    int32_t* dl = ms.first_touch.data();
    int* sc = ms.step_count.data();
    char* tc = ms.touched.data();
    WorkerPool& pool = WorkerPool::Get();
    const size_t parallel_from = knobs_.parallel_count_min;   // (the tests: this path on small images)
    if (fast_until >= parallel_from && pool.size() > 1) {
      // The first "up" iteration of an encode takes every candidate below the error limit at
      // once -- 7.5 M entries at 4K, 7.5 of this loop's 9 ms per encode: the entries in `parts`
      // ranges, a private count array per range (0.5 MB: it stays in the core's cache), summed
      // afterwards.  Which order the touched blocks are listed in matters to nobody (independent
      // blocks on the device, a set to be cleared here).
      const int parts = std::min(pool.size(), 8);
      std::vector<std::vector<int32_t> > part_count((size_t)parts);
      pool.Run(parts, [&](int p) {
        std::vector<int32_t>& cnt = part_count[(size_t)p];
        cnt.assign((size_t)nb, 0);
        const size_t i0 = fast_until * (size_t)p / parts, i1 = fast_until * (size_t)(p + 1) / parts;
        for (size_t i = i0; i < i1; ++i) ++cnt[(size_t)order[i].first];
      });
      size_t nd = 0;
      for (int b = 0; b < nb; ++b) {
        int n = 0;
        for (int p = 0; p < parts; ++p) n += part_count[(size_t)p][(size_t)b];
        if (n == 0) continue;
        if (sc[b] == 0) dl[nd++] = b;
        sc[b] += n;
        tc[b] = 1;
      }
      ms.dirty.assign(dl, dl + nd);
    } else {
      size_t nd = 0;
      for (size_t i = 0; i < fast_until; ++i) {
        const int b = order[i].first;
        dl[nd] = b;
        nd += sc[b] == 0;
        ++sc[b];
        tc[b] = 1;
      }
      ms.dirty.assign(dl, dl + nd);
    }
  }
  t_fs_count_ += fw.lap();
  if (fast_until > 0) {
    std::vector<int32_t>& dirty = ms.dirty;
    it.val_threshold = order[fast_until - 1].second;
    it.changed_coeffs += (int)fast_until;
    // the device applies the same steps to its image (and advances its next_cand) ...
    std::vector<int32_t>& counts = bulk_counts_;
    counts.resize(dirty.size());
    for (size_t di = 0; di < dirty.size(); ++di) counts[di] = ms.step_count[dirty[di]];
    rc = gz_apply_candidate_steps(ctx_, direction, dirty.data(), counts.data(), (int)dirty.size());
    if (rc != GZ_OK) return Fail("gz_apply_candidate_steps", rc);
    t_fs_apply_ += fw.lap();
    // ... while the host only notes how far each block has advanced; its mirror of the
    // coefficients follows when a slow step needs the block (SettleBlock)
    // (step_count holds exactly these counts and zeros elsewhere: with a fifth of the blocks
    // touched, one pass over the two arrays -- 20 us -- beats 26 000 scattered updates -- 40-120)
    if (dirty.size() * 16 > (size_t)nb) {
      int* nc = ms.next_cand.data();
      const int* scp = ms.step_count.data();
      for (int b = 0; b < nb; ++b) nc[b] += direction * scp[b];
    } else {
      for (size_t di = 0; di < dirty.size(); ++di) ms.next_cand[dirty[di]] += direction * counts[di];
    }
    if (verify_) SettleAll(&ms);   // GZ_VERIFY_ENTROPY compares the whole mirror
    // the symbol statistics of the edited image come from the device: the change the
    // steps made to BuildACHistograms, counted over the touched blocks (the host's
    // ac_histo was exact before them: the slow steps keep it so)
    t_fs_mirror_ += fw.lap();
    std::vector<int32_t> delta(3 * 256);
    rc = gz_steps_histogram_delta(ctx_, delta.data());
    t_fs_delta_ += fw.lap();
    if (rc != GZ_OK) return Fail("gz_steps_histogram_delta", rc);
    for (int c = 0; c < 3; ++c)
      for (int i = 0; i < 256; ++i)
        if (delta[c * 256 + i]) ms.ac_histo[c].Add(i, delta[c * 256 + i]);
    if (verify_) {   // GZ_VERIFY_ENTROPY=1: against a recount of the whole image
      SymbolHistogram dc_now[3], ac_now[3];
      if (!DeviceHistograms(quant_, dc_now, ac_now)) return false;
      for (int c = 0; c < ncomp; ++c)
        if (memcmp(ac_now[c].counts, ms.ac_histo[c].counts, sizeof(ac_now[c].counts)) != 0) {
          fprintf(stderr, "guetzli_amd: incremental AC statistics differ from a recount\n");
          return false;
        }
    }
  }
  t_fs_rest_ += fw.lap();
  return true;
}

// One step of the reference loop (processor.cc:704-750) without its size estimate: change one coefficient of
// block b (host mirror + edit list for the device) and keep ac_histo / ac_raw_bits current.
void Encoder::ApplyStep(MaskSearch* msp, Iteration* it, SortedOrder* sortedp, size_t i) {
  SortedOrder& sorted = *sortedp;   // (operator[] sorts lazily: not const)
  MaskSearch& ms = *msp;
  const int direction = it->direction;
  const int b = sorted[i].first;
  SettleBlock(msp, b);
  const int idx = cand_idx_[cand_off_[b] + ms.next_cand[b] + std::min(direction, 0)];
  const int c = idx / 64, k = idx % 64;
  const int* q = quant_[c];
  const int16_t* orig_blk = &orig_[Pos(c, b, 0)];
  int16_t* blk = &img_[Pos(c, b, 0)];
  const int newval = direction > 0 ? 0 : QuantizeCoeff(orig_blk[k], q[k]);
  const uint8_t* depth = &ms.ac_depths[c * kHistoSize];
  if (!(newval == 0 && IsPrecious(orig_blk, k))) {
    // UpdateACHistogram before and after the change (processor.cc:715-722): only the symbols
    // around the coefficient differ
    if (k >= 1) {
      ReplaceCoeffACSymbols(blk, q, k, newval, &ms.ac_histo[c], depth, &ms.ac_raw_bits[c]);
      blk[k] = (int16_t)newval;
    } else {
      AddBlockACSymbols(blk, q, -1, &ms.ac_histo[c], depth, &ms.ac_raw_bits[c]);
      blk[k] = (int16_t)newval;
      AddBlockACSymbols(blk, q, 1, &ms.ac_histo[c], depth, &ms.ac_raw_bits[c]);
    }
    ms.edit_pos.push_back((int32_t)Pos(c, b, k));
    ms.edit_val.push_back((int16_t)newval);
  }
  ms.next_cand[b] += direction;
  ms.mirror_cand[b] = ms.next_cand[b];
  if (!ms.touched[b]) {
    ms.touched[b] = 1;
    ms.dirty.push_back(b);
  }
  it->val_threshold = sorted[i].second;
  ++it->changed_coeffs;
}

// The reference's loop (processor.cc:704-750) over the steps [from, to): true = the stopping rule fired (at
// the last step taken).
bool Encoder::StepsAsTheReference(MaskSearch* msp, Iteration* it, SortedOrder* sorted, size_t from, size_t to) {
  MaskSearch& ms = *msp;
  const int ncomp = ms.ncomp;
  for (size_t i = from; i < to; ++i) {
    ApplyStep(msp, it, sorted, i);
    if (i % 10 == 0) {
      Stopwatch cw;
      ms.ac_header = (int)EntropyCodes(ms.ac_histo, ncomp, ms.ac_depths.data());
      RecountRawBits(msp);
      t_pb_codes_ += cw.lap();
    }
    ++n_steps_;
    size_t data_bits = 0;
    for (int c = 0; c < ncomp; ++c) data_bits += EntropyBitsFromRaw(ms.ac_raw_bits[c]);
    it->est_size = ms.header_size + ms.dc_size + ms.ac_header + (int)((data_bits + 7) / 8);
    if (verify_ && (data_bits + 7) / 8 != EntropyDataSize(ms.ac_histo, ncomp, ms.ac_depths.data())) {
      fprintf(stderr, "guetzli_amd: incremental size estimate differs from a recount\n");
      it->verify_failed = true;
      return true;
    }
    if (it->changed_coeffs > it->min_coeffs_to_change &&
        std::abs(it->est_size - ms.prev_size) > it->min_size_delta)
      return true;
  }
  return false;
}

// The serial steps from `base` (a multiple of 10) on with their code refreshes on the helper threads
// (code_refresh.h).  Window w = the steps base + 10 w .. + 9; its first step is a refresh step.  This
// thread takes the steps of up to `lag + 1` windows before it prices the oldest of them:
// the statistics right after a window's first step go to a helper, the steps' symbol changes
// are kept, and when the window's codes are there every step gets the size estimate the
// reference computes for it -- raw bits of the statistics at the refresh under the new
// depths, then the steps' changes priced with those depths (what ReplaceCoeffACSymbols adds
// to ac_raw_bits) -- and the stopping rule is applied in step order.  Steps taken beyond
// the one it fires at are undone, last first.
void Encoder::StepsWithHelpers(MaskSearch* msp, Iteration* itp, SortedOrder* sortedp, size_t base) {
  MaskSearch& ms = *msp;
  Iteration& it = *itp;
  SortedOrder& sorted = *sortedp;
  const int ncomp = ms.ncomp, direction = it.direction;
  const size_t n_order = it.n_order;
  std::vector<int32_t>& cand_off = cand_off_;
  std::vector<uint8_t>& cand_idx = cand_idx_;
  std::vector<int>& next_cand = ms.next_cand;
  std::vector<int>& mirror_cand = ms.mirror_cand;
  std::vector<char>& touched = ms.touched;
  std::vector<int32_t>& dirty = ms.dirty;
  refreshers_->Activate();
  std::vector<SlowStep>& slog = slow_log_;
  slog.clear();
  const long lag = refreshers_->threads();
  const long w0 = refreshers_->NextWindow();
  long applied_w = 0, priced_w = 0;
  size_t next_apply = base;
  bool stopped = false;
  size_t last_priced = base;   // the last step with an estimate (valid once a window is priced)
  // A step touches one block out of 130 000 at random: its entry of the per-block arrays, its
  // candidate list, its coefficient blocks in the original and in the image -- four dependent
  // cache misses, which is what a step costs.  The steps to come are known (the sorted order),
  // so their lines are asked for ahead, one dependency per stage.  (A block that advances in
  // between makes a prefetch miss its mark by a candidate; nothing depends on these.)
  const bool prefetch_ahead = knobs_.step_prefetch;
  auto prefetch_for = [&](size_t i) {
    if (i + 12 < n_order) {
      const int b1 = sorted[i + 12].first;
      __builtin_prefetch(&cand_off[b1]);
      __builtin_prefetch(&next_cand[b1]);
      __builtin_prefetch(&mirror_cand[b1]);
      __builtin_prefetch(&touched[b1]);
    }
    if (i + 8 < n_order) {
      const int b2 = sorted[i + 8].first;
      __builtin_prefetch(&cand_idx[cand_off[b2] + next_cand[b2] + std::min(direction, 0)]);
    }
    if (i + 4 < n_order) {
      const int b3 = sorted[i + 4].first;
      const int idx3 = cand_idx[cand_off[b3] + next_cand[b3] + std::min(direction, 0)];
      const size_t p3 = Pos(idx3 / 64, b3, 0);
      __builtin_prefetch(&orig_[p3]);
      __builtin_prefetch(&orig_[p3 + 32]);
      __builtin_prefetch(&img_[p3], 1);
      __builtin_prefetch(&img_[p3 + 32], 1);
    }
  };
  auto take_step = [&](size_t i) {
    if (prefetch_ahead) prefetch_for(i);
    const int b = sorted[i].first;
    SettleBlock(msp, b);
    const int idx = cand_idx[cand_off[b] + next_cand[b] + std::min(direction, 0)];
    const int c = idx / 64, k = idx % 64;
    const int* q = quant_[c];
    const int16_t* orig_blk = &orig_[Pos(c, b, 0)];
    int16_t* blk = &img_[Pos(c, b, 0)];
    const int newval = direction > 0 ? 0 : QuantizeCoeff(orig_blk[k], q[k]);
    SlowStep st;
    st.b = b;
    st.val = sorted[i].second;
    st.comp = (uint8_t)c;
    st.changed = 0;
    st.nsym = 0;
    st.pos = 0;
    st.old_val = 0;
    if (!(newval == 0 && IsPrecious(orig_blk, k))) {
      st.changed = 1;
      st.pos = (int32_t)Pos(c, b, k);
      st.old_val = blk[k];
      // (k == 0: a block's AC symbols do not depend on its DC coefficient)
      if (k >= 1) st.nsym = (uint8_t)CoeffACSymbolChanges(blk, q, k, newval, st.sym);
      for (int j = 0; j < st.nsym; ++j)
        ms.ac_histo[c].Add(std::abs(st.sym[j]) - 1, st.sym[j] > 0 ? 1 : -1);
      blk[k] = (int16_t)newval;
      ms.edit_pos.push_back(st.pos);
      ms.edit_val.push_back((int16_t)newval);
    }
    next_cand[b] += direction;
    mirror_cand[b] = next_cand[b];
    st.first_touch = !touched[b];
    if (!touched[b]) {
      touched[b] = 1;
      dirty.push_back(b);
    }
    slog.push_back(st);
  };
  auto undo_step = [&](const SlowStep& st) {
    const int b = st.b;
    if (st.first_touch) {
      touched[b] = 0;
      dirty.pop_back();
    }
    next_cand[b] -= direction;
    mirror_cand[b] = next_cand[b];
    if (st.changed) {
      ms.edit_pos.pop_back();
      ms.edit_val.pop_back();
      img_[st.pos] = st.old_val;
      for (int j = 0; j < st.nsym; ++j)
        ms.ac_histo[st.comp].Add(std::abs(st.sym[j]) - 1, st.sym[j] > 0 ? -1 : 1);
    }
  };
  for (;;) {
    if (next_apply < n_order && applied_w - priced_w <= lag) {
      // take the steps of the next window; the refresh of its first step goes out at once
      const size_t i0 = next_apply, i1 = std::min(i0 + 10, n_order);
      for (size_t i = i0; i < i1; ++i) {
        take_step(i);
        if (i == i0) {
          CodeRefresh* in = refreshers_->Input(w0 + applied_w);
          memcpy(in->histo, ms.ac_histo, sizeof(in->histo));
          in->ncomp = ncomp;
          refreshers_->Submit(w0 + applied_w);
        }
      }
      next_apply = i1;
      ++applied_w;
      continue;
    }
    if (priced_w == applied_w) break;   // every step of the order taken and priced
    Stopwatch cw;
    const CodeRefresh* r = refreshers_->Wait(w0 + priced_w);
    t_pb_codes_ += cw.lap();
    // (the helper writes the depths of the frame's components only)
    memcpy(ms.ac_depths.data(), r->depths, std::min(ms.ac_depths.size(), (size_t)ncomp * kHistoSize));
    ms.ac_header = r->ac_header;
    for (int c = 0; c < 3; ++c) ms.ac_raw_bits[c] = r->raw_bits[c];
    const size_t i0 = base + 10 * (size_t)priced_w, i1 = std::min(i0 + 10, n_order);
    for (size_t i = i0; i < i1; ++i) {
      const SlowStep& st = slog[i - base];
      if (i > i0) {   // (the refresh step's own changes are in the statistics the codes were made for)
        const uint8_t* depth = &ms.ac_depths[st.comp * kHistoSize];
        int64_t bits = 0;
        for (int j = 0; j < st.nsym; ++j) {
          const int symbol = std::abs(st.sym[j]) - 1;
          const int cost = depth[symbol] + (symbol & 0xf);
          bits += st.sym[j] > 0 ? cost : -cost;
        }
        ms.ac_raw_bits[st.comp] += bits;
      }
      size_t data_bits = 0;
      for (int c = 0; c < ncomp; ++c) data_bits += EntropyBitsFromRaw(ms.ac_raw_bits[c]);
      it.est_size = ms.header_size + ms.dc_size + ms.ac_header + (int)((data_bits + 7) / 8);
      last_priced = i;
      if ((int)i + 1 > it.min_coeffs_to_change &&
          std::abs(it.est_size - ms.prev_size) > it.min_size_delta) {
        stopped = true;
        break;
      }
    }
    ++priced_w;
    if (stopped) break;
  }
  // the steps beyond the last one the reference takes, last first; then the refreshes that
  // were asked for on their behalf (a slot is handed out again only after its window is done)
  const size_t keep = last_priced + 1;   // steps [base, keep) stay
  for (size_t i = next_apply; i > keep; --i) undo_step(slog[i - 1 - base]);
  n_steps_undone_ += (long)(next_apply - keep);
  for (long w = priced_w; w < applied_w; ++w) (void)refreshers_->Wait(w0 + w);
  it.changed_coeffs += (int)(keep - base);
  it.val_threshold = slog[keep - 1 - base].val;
  n_steps_ += (long)(keep - base);
}

// The iteration's candidate: its changed coefficients to the device image, the evaluation enqueued with the
// next iteration's order behind it, the exact size where it is observable (processor.cc:752-772).
bool Encoder::EvaluateCandidate(MaskSearch* msp, Iteration* itp, Stopwatch* sw) {
  MaskSearch& ms = *msp;
  Iteration& it = *itp;
  const int direction = it.direction;
  // push the changed coefficients to the device image (positions are distinct: a block's
  // candidates are distinct coefficients and a block advances in one direction)
  int rc = gz_apply_coeff_edits(ctx_, ms.edit_pos.data(), ms.edit_val.data(), (int)ms.edit_pos.size());
  t_upload_ += sw->lap();
  if (rc != GZ_OK) return Fail("gz_apply_coeff_edits", rc);

  size_t jpg_size = 0;
  if (!CompareBegin()) return false;
  // The candidate's exact size is observable in two places only: the --verbose trace
  // (Out[...], EstErr[...]) and MaybeOutput's comparison of scores (processor.cc:139-148,767).
  // Its head and the exact length of its scan in bits follow from the symbol statistics the
  // host holds anyway (PrepareHead); only the bytes stuffed behind 0xFF need the coder.  Without
  // a trace the candidate is therefore entropy-coded only if it can win: ScoreJPEG grows with
  // the size, so a candidate whose score at its size's LOWER bound does not beat the best so
  // far loses whatever it weighs -- 140 of the 149 candidates of a 4K encode at quality 95,
  // whose evaluation then has the device to itself (the coder's kernels took a sixth of the
  // summed kernel time, profiles/r03_bench_kernel_stats.csv).
  // GZ_VERIFY_ENTROPY=2 checks the DEFAULT path: the bound decision is taken first and the
  // candidate is coded regardless, late (behind the evaluation, as a winner is) -- the bound must
  // not exceed the coded size, and a candidate the bound rejects must lose with its real size too
  // (with =1 / --verbose every candidate takes the early-scan path instead).
  const bool verify_late = verify_ && verify_level_ >= 2 && !stats_->debug_output && !stats_->debug_output_file;
  const bool every_size = (stats_->debug_output || stats_->debug_output_file || verify_) && !verify_late;
  // the entropy coder goes to its own stream before anything else is enqueued: it runs beside
  // the evaluation, not behind the host work below.  Without it the head is built (15 us of Huffman codes) AFTER
  // the next order's launches: up to 1080p the device finishes an evaluation before the host has enqueued what
  // follows it, and whatever the host does in between is time the device idles.
  if (every_size && (!PrepareHead(quant_, ms.dc_histo, ms.ac_histo) || !ScanBegin())) return false;
  Stopwatch aw;
  {
    // the next iteration of this direction, radius 1, if it comes to that (processor.cc:
    // 622-663 behind :767): everything it reads is final -- next_cand, max_block_error
    // (gz_order_advance in the caller), and the distance map the device is about to produce -- with the
    // descent behind it, and everything the host waits for at this point (the order's size and
    // counters, the descent's cuts, the candidate's distance) in one transfer
    rc = gz_order_build_auto_descend_begin(ctx_, direction, 1, ms.target_mul, 1, ms.next_cand.data(), 0,
                                           it.below_limit, it.per_block, device_threshold_, descend_levels_);
    if (rc != GZ_OK) return Fail("gz_order_build_auto_descend_begin", rc);
    ms.ahead = direction;
  }
  t_ahead_begin_ += aw.lap();
  if (!every_size && !PrepareHead(quant_, ms.dc_histo, ms.ac_histo)) return false;
  if (every_size) {
    if (!SerializeEnd(quant_, &jpg_size)) return false;
    Log("Iter %2d: %s(%d) %s Coeffs[%d/%zd] Blocks[%zd/%d/%d] ValThres[%.4f] Out[%7zd] "
        "EstErr[%.2f%%]",
        stats_->counters[kNumItersCnt], FrameStr(), ms.comp_mask, direction > 0 ? "up" : "down",
        it.changed_coeffs, it.n_order, ms.dirty.size(), it.blocks_to_change, ms.nb, it.val_threshold,
        jpg_size, 100.0 - (100.0 * it.est_size) / jpg_size);
    if (!CompareCurrent()) return false;
    if (!MaybeOutput(jpg_size)) return false;
  } else {
    if (!CompareCurrent()) return false;
    const bool may_win = best_score_ < 0 ||
        ScoreJPEG(distance_, (int)SizeLowerBound(), params_.butteraugli_target) < best_score_;
    if (may_win) {
      if (!ScanBegin() || !SerializeEnd(quant_, &jpg_size) || !MaybeOutput(jpg_size)) return false;
    } else if (verify_late) {
      const double best_before = best_score_;
      if (!ScanBegin() || !SerializeEnd(quant_, &jpg_size)) return false;   // (checks bound <= size itself)
      if (ScoreJPEG(distance_, (int)jpg_size, params_.butteraugli_target) < best_before) {
        fprintf(stderr, "guetzli_amd: a candidate rejected on its size bound would have won\n");
        return false;
      }
      ++n_scans_skipped_;
    } else {
      ++n_scans_skipped_;
    }
  }
  return true;
}

bool Encoder::Search(const QuantMatrix first_q, std::string* out) {
  Stopwatch sw;
  int rc = GZ_OK;
  const int w = w_, h = h_;
  // the original as the fallback output (processor.cc:826-846)
  verify_ = knobs_.verify_level > 0;
  verify_level_ = knobs_.verify_level;
  if (knobs_.order_device_threshold > 0) device_threshold_ = (size_t)knobs_.order_device_threshold;
  best_score_ = -1;
  QuantMatrix ones;
  for (int c = 0; c < 3; ++c)
    for (int k = 0; k < 64; ++k) ones[c][k] = 1;
  if (!SetImageFromQuantization(ones, false)) return false;
  if (jpeg_input_ && fac_ == 2) {
    // OutputJpeg(jpg_in) of a 4:2:0 input: the file is written from the input's own blocks,
    // MCU padding included (the device frame leaves the padding out): on the host, once
    if (!WriteJpeg(in_frame_, &best_full_)) return Fail("WriteJpeg", GZ_E_STATE);
    const size_t size = best_full_.size();
    Log("Original Out[%7zd]", size);
    if (!CompareBegin() || !CompareCurrent()) return false;
    const double score = ScoreJPEG(distance_, (int)size, params_.butteraugli_target);
    Log(" Score[%.4f]", score);
    best_score_ = score;   // final_output_->score < 0: always taken (:142)
    best_size_ = size;
    best_on_host_ = true;
    Log(" (*)\n");
  } else {
    // symbols of the original: coefficient / its quantiser (1 for RGB input, the input's own
    // tables for a JPEG input, whose coefficients are held dequantised)
    SymbolHistogram dc[3], ac[3];
    size_t size = 0;
    if (!DeviceHistograms(jpeg_input_ ? q_in_ : ones, dc, ac) || !CompareBegin() ||
        !Serialize(nullptr, dc, ac, &size))
      return false;
    Log("Original Out[%7zd]", size);
    if (!CompareCurrent()) return false;
    if (!MaybeOutput(size)) return false;
  }

  // ProcessJpegData's loop over the sampling modes (:847-878)
  bool grey = true;   // IsGrayscale(jpg_in), :782-790
  for (size_t i = (size_t)coff_[1] * 64; i < (size_t)nblk_ * 64 && grey; ++i) grey = orig_[i] == 0;
  const bool input_is_420 = fac_ == 2;
  const int try_420 = (input_is_420 || params_.force_420 || (params_.try_420 && !grey)) ? 1 : 0;
  const int force_420 = (input_is_420 || params_.force_420) ? 1 : 0;
  for (int downsample = force_420; downsample <= try_420; ++downsample) {
    // SaveToJpegData writes ONE component when both chroma components are all zero
    // (output_image.cc:357) -- whatever the frame: an RGB / 4:4:4 input whose chroma is zero
    // (Downsample does nothing, :305-308) and a 4:2:0 JPEG input with zero chroma alike.  The
    // round then runs with ymul = 1.0, without the chroma search and with one AC histogram.
    jpg_ncomp_ = (downsample && grey) ? 1 : 3;
    mirror_valid_ = false;   // img_ follows the device image from SetImageFromQuantization(best_q) on
    if (downsample && fac_ == 1) {   // DownsampleImage (:97-104) + SaveToJpegData
      if (!grey) {
        Stopwatch dw;
        if (params_.use_silver_screen) {
          // output_image.cc:309-318: ToSRGB() of the unquantised image -> RGBToYUV420 (host:
          // silver_screen.cc) -> all three components from the planes it returns
          std::vector<uint8_t> srgb((size_t)3 * w_ * h_);
          rc = gz_quantize(ctx_, nullptr, nullptr);
          if (rc == GZ_OK) rc = gz_reconstruct(ctx_, srgb.data(), nullptr);
          if (rc != GZ_OK) return Fail("gz_reconstruct", rc);
          std::vector<float> py, pu, pv;
          SilverScreenYUV420(srgb.data(), w_, h_, &py, &pu, &pv);
          rc = gz_downsample_planes(ctx_, py.data(), pu.data(), pv.data(), orig_.data());
          if (rc != GZ_OK) return Fail("gz_downsample_planes", rc);
        } else {
          rc = gz_downsample(ctx_, orig_.data());
          if (rc != GZ_OK) return Fail("gz_downsample", rc);
        }
        SetFrame(2);
        stats_->timers["downsample"] = dw.lap();
      }
    }
    QuantMatrix best_q;
    memcpy(best_q, first_q, sizeof(best_q));
    bool dist_ok = false;
    if (!SelectMatrix(best_q, downsample != 0, &dist_ok)) return false;
    if (!dist_ok)
      for (int c = 0; c < 3; ++c)
        for (int k = 0; k < 64; ++k) best_q[c][k] = 1;
    stats_->timers["select_quant_matrix"] += sw.lap();
    if (!SetImageFromQuantization(best_q, true)) return false;
    mirror_valid_ = true;
    if (!downsample) {
      if (!SelectFrequencyMasking(7, 1.0, false, true)) return false;
    } else {
      const float ymul = jpg_ncomp_ == 1 ? 1.0f : 0.97f;
      if (!SelectFrequencyMasking(1, ymul, false, false)) return false;
      if (!SelectFrequencyMasking(6, 1.0, true, true)) return false;
    }
    stats_->timers["select_frequency_masking"] += sw.lap();
  }
  stats_->timers["jpeg_write"] = t_write_;
  stats_->timers["compare"] = t_compare_;
  stats_->timers["quantize"] = t_quant_;
  stats_->timers["block_search"] = t_blocksearch_;
  stats_->timers["phase_b_host"] = t_phaseb_;
  stats_->timers["block_upload"] = t_upload_;
  stats_->timers["jpeg_head"] = t_head_;
  stats_->timers["compare_begin"] = t_cmp_begin_;
  stats_->timers["compare_end"] = t_cmp_end_;
  stats_->timers["jpeg_scan_begin"] = t_scan_begin_;
  stats_->timers["jpeg_scan_end"] = t_scan_end_;
  stats_->timers["pb_order_ahead_begin"] = t_ahead_begin_;
  stats_->timers["pb_order"] = t_pb_order_;
  stats_->timers["pb_sort"] = t_pb_sort_;
  stats_->timers["pb_loop"] = t_pb_loop_;
  stats_->timers["pb_loop_codes"] = t_pb_codes_;
  stats_->timers["pb_loop_ensure_sorted"] = t_pb_ensure_;
  stats_->timers["pb_fast_count"] = t_fs_count_;
  stats_->timers["pb_fast_apply"] = t_fs_apply_;
  stats_->timers["pb_fast_mirror"] = t_fs_mirror_;
  stats_->timers["pb_fast_delta"] = t_fs_delta_;
  stats_->timers["pb_device_partitions"] = t_pb_dev_partition_;
  stats_->timers["pb_device_fetches"] = t_pb_dev_fetch_;
  stats_->timers["pb_device_descents"] = t_pb_descend_;
  stats_->counters["phase B prefixes exported by the device"] = (int)n_dev_exported_;
  stats_->counters["phase B partitions made ahead"] = (int)n_dev_replayed_;
  stats_->counters["phase B device partitions"] = (int)n_dev_partitions_;
  stats_->counters["phase B entries fetched"] = (int)std::min<long>(n_dev_fetched_, 2147483647L);
  t_pb_fast_ = t_fs_count_ + t_fs_apply_ + t_fs_mirror_ + t_fs_delta_ + t_fs_rest_;
  stats_->timers["pb_loop_fast_steps"] = t_pb_fast_;
  stats_->counters["block search evaluations"] = (int)std::min<long>(n_evaluations_, 2000000000L);
  stats_->counters["phase B fast steps"] = (int)n_fast_;
  stats_->counters["candidates entropy-coded"] = (int)n_scans_;
  stats_->counters["candidates rejected on their size bound"] = (int)n_scans_skipped_;
  stats_->counters["phase B coefficient steps"] = (int)n_steps_;
  stats_->counters["phase B steps taken ahead and undone"] = (int)n_steps_undone_;
  stats_->counters["phase B code refresh threads"] = refreshers_ ? refreshers_->threads() : 0;
  stats_->counters["phase B order entries"] = (int)std::min<long>(n_order_, 2000000000L);
  if (best_on_host_) {
    *out = best_full_;
  } else {  // the winner: its head from the host, its scan from the device
    std::vector<uint8_t>& scan = scan_;
    if (scan.size() < (size_t)6 * w * h + 4096) scan.resize((size_t)6 * w * h + 4096);
    size_t n = 0;
    rc = gz_jpeg_scan_bytes(ctx_, 1, scan.data(), scan.size(), &n);
    if (rc != GZ_OK) return Fail("gz_jpeg_scan_bytes", rc);
    *out = best_head_;
    out->append((const char*)scan.data(), n);
    out->push_back((char)0xff);
    out->push_back((char)0xd9);
    if (jpeg_input_ && !meta_.strip) out->append(meta_.tail_data);
  }
  return true;
}

// libjpeg's colour-space guess for three-component files (jpeg_data_decoder.cc:23-43): JFIF
// means YCbCr, an Adobe marker decides by its transform byte, else the ids 'R','G','B' mean RGB.
static bool HasYCbCrColorSpace(const JpegInput& jpg) {
  bool adobe = false;
  uint8_t transform = 0;
  for (const std::string& app : jpg.app_data) {
    if ((uint8_t)app[0] == 0xe0) return true;
    if ((uint8_t)app[0] == 0xee && app.size() >= 15) {
      adobe = true;
      transform = (uint8_t)app[14];
    }
  }
  if (adobe) return transform != 0;
  return jpg.components[0].id != 'R' || jpg.components[1].id != 'G' || jpg.components[2].id != 'B';
}

// guetzli::Process(params, stats, jpeg_data, &out) (processor.cc:890-924) for YUV 4:4:4 input.
bool Encoder::RunJpeg(const std::string& data, std::string* out) {
  Stopwatch total, sw;
  JpegInput jpg;
  if (!ReadJpeg((const uint8_t*)data.data(), data.size(), &jpg)) {
    fprintf(stderr, "Can't read jpg data from input file\n");
    return false;
  }
  for (const JpegComponentIn& comp : jpg.components) {   // CheckJpegSanity, :117-131
    const int* q = jpg.quant[comp.quant_idx].values;
    for (size_t i = 0; i < comp.coeffs.size(); ++i)
      if (std::abs((int64_t)comp.coeffs[i] * q[i % 64]) > (1 << 12)) {
        fprintf(stderr, "Unsupported input JPEG (unexpectedly large coefficient values).\n");
        return false;
      }
  }
  const size_t ncomp = jpg.components.size();
  const bool ycbcr3 = ncomp == 3 && HasYCbCrColorSpace(jpg);
  if (!(ncomp == 1 || (ycbcr3 && (jpg.Is420() || jpg.Is444())))) {   // DecodeJpegToRGB is empty
    fprintf(stderr, "Unsupported input JPEG file (e.g. unsupported downsampling mode).\n"
                    "Please provide the input image as a PNG file.\n");
    return false;
  }
  if (params_.butteraugli_target > 2.0f) {   // ProcessJpegData, :800-806
    fprintf(stderr,
            "Guetzli should be called with quality >= 84, otherwise the\n"
            "output will have noticeable artifacts. If you want to\n"
            "proceed anyway, please edit the source code.\n");
    return false;
  }
  if (!ycbcr3) {
    fprintf(stderr, "Only YUV color space input jpeg is supported\n");
    return false;
  }
  if (!jpg.Is444() && !jpg.Is420()) {   // ProcessJpegData, :811-824
    fprintf(stderr, "Unsupported sampling factors:");
    for (const JpegComponentIn& comp : jpg.components) fprintf(stderr, " %dx%d", comp.h_samp, comp.v_samp);
    fprintf(stderr, "\n");
    return false;
  }
  const int w = jpg.width, h = jpg.height;
  w_ = w; h_ = h;
  bw_ = (w + 7) / 8; bh_ = (h + 7) / 8; nb_ = bw_ * bh_;
  SetFrame(jpg.Is420() ? 2 : 1);
  jpeg_input_ = true;
  meta_.strip = params_.clear_metadata;
  meta_.app_data = jpg.app_data;
  meta_.com_data = jpg.com_data;
  meta_.tail_data = jpg.tail_data;
  in_quant_.clear();
  for (const JpegQuant& t : jpg.quant) {
    QuantTable q;
    memcpy(q.values, t.values, sizeof(q.values));
    q.precision = t.precision;
    q.index = t.index;
    in_quant_.push_back(q);
  }
  // RemoveOriginalQuantization (:84-97): coefficients are held dequantised; of a 4:2:0
  // input's blocks those inside the image (CopyFromJpegComponent, output_image.cc:211-230)
  orig_.resize((size_t)3 * nb_ * 64);
  for (int c = 0; c < 3; ++c) {
    const JpegComponentIn& comp = jpg.components[c];
    in_comp_id_[c] = comp.id;
    in_quant_idx_[c] = comp.quant_idx;
    memcpy(q_in_[c], jpg.quant[comp.quant_idx].values, sizeof(q_in_[c]));
    const int rw = c == 0 ? bw_ : cbw_, rh = c == 0 ? bh_ : cbh_;
    if (comp.width_in_blocks < rw || comp.height_in_blocks < rh) return Fail("block grid", GZ_E_STATE);
    int16_t* dst = &orig_[(size_t)coff_[c] * 64];
    for (int by = 0; by < rh; ++by)
      for (int bx = 0; bx < rw; ++bx, dst += 64) {
        const int16_t* src = &comp.coeffs[((size_t)by * comp.width_in_blocks + bx) * 64];
        for (int k = 0; k < 64; ++k) dst[k] = (int16_t)(src[k] * q_in_[c][k]);
      }
  }
  if (fac_ == 2) {   // the input as read, for OutputJpeg(jpg_in)
    Frame& f = in_frame_;
    f.width = w; f.height = h; f.bw = bw_; f.bh = bh_;
    f.ncomp = 3;
    f.mcu_cols = jpg.mcu_cols; f.mcu_rows = jpg.mcu_rows;
    for (int c = 0; c < 3; ++c) {
      const JpegComponentIn& comp = jpg.components[c];
      f.samp[c] = comp.h_samp;
      f.cw[c] = comp.width_in_blocks;
      f.ch[c] = comp.height_in_blocks;
      f.coeffs[c] = comp.coeffs;
      f.quant_idx[c] = in_quant_idx_[c];
      f.comp_id[c] = in_comp_id_[c];
    }
    f.quant = in_quant_;
    f.meta = &meta_;
  }
  if (w < 32 || h < 32) {
    // no butteraugli (:832-838): the input re-written with optimised Huffman codes
    Frame f444;
    if (fac_ == 1) {
      FrameFromImage(orig_.data(), q_in_, w, h, &f444);
      f444.ncomp = 3;
      f444.quant = in_quant_;
      for (int c = 0; c < 3; ++c) {
        f444.quant_idx[c] = in_quant_idx_[c];
        f444.comp_id[c] = in_comp_id_[c];
      }
      f444.meta = &meta_;
    }
    const Frame& f = fac_ == 2 ? in_frame_ : f444;
    if (!WriteJpeg(f, out)) return Fail("WriteJpeg", GZ_E_STATE);
    Log("Original Out[%7zd]", out->size());
    Log(" <image too small for Butteraugli>\n");
    return true;
  }
  // The comparator's original is DecodeJpegToRGB(jpg) (jpeg_data_decoder.cc:45-54): the
  // integer IDCT of the input, computed by the context itself.
  int err = 0;
  {
    std::vector<uint8_t> blank((size_t)3 * w * h, 0);
    ctx_ = gz_create(params_.device, w, h, blank.data(), params_.butteraugli_target, &err);
    if (!ctx_) return Fail("gz_create", err);
    int rc = fac_ == 2 ? gz_set_orig_coeffs_420(ctx_, orig_.data()) : gz_set_orig_coeffs(ctx_, orig_.data());
    if (rc == GZ_OK) rc = gz_quantize(ctx_, nullptr, nullptr);
    if (rc == GZ_OK) rc = gz_reconstruct(ctx_, blank.data(), nullptr);
    if (rc == GZ_OK) rc = gz_set_rgb(ctx_, blank.data());
    if (rc != GZ_OK) return Fail("decode of the input JPEG", rc);
  }
  img_.resize(orig_.size());
  stats_->timers["create+encode"] = sw.lap();
  if (!Search(q_in_, out)) return false;
  stats_->timers["total"] = total.lap();
  return true;
}

bool Encoder::Run(const std::vector<uint8_t>& rgb, int w, int h, std::string* out) {
  Stopwatch total, sw;
  if (params_.butteraugli_target > 2.0f) {   // processor.cc:800-806
    fprintf(stderr,
            "Guetzli should be called with quality >= 84, otherwise the\n"
            "output will have noticeable artifacts. If you want to\n"
            "proceed anyway, please edit the source code.\n");
    return false;
  }
  if (w < 0 || w >= 1 << 16 || h < 0 || h >= 1 << 16 || rgb.size() != (size_t)3 * w * h) {
    fprintf(stderr, "Could not create jpg data from rgb pixels\n");   // EncodeRGBToJpeg failed
    return false;
  }
  w_ = w; h_ = h;
  bw_ = (w + 7) / 8; bh_ = (h + 7) / 8; nb_ = bw_ * bh_;
  SetFrame(1);
  if (w < 32 || h < 32) {
    // "image too small for Butteraugli" (processor.cc:832-838, :940): the reference emits
    // the unquantised JPEG of EncodeRGBToJpeg; the forward transform runs on the device.
    if (w < 1 || h < 1) {
      fprintf(stderr, "Could not create jpg data from rgb pixels\n");
      return false;
    }
    orig_.resize((size_t)3 * nb_ * 64);
    const int rc0 = gz_encode_rgb_only(params_.device, rgb.data(), w, h, orig_.data());
    if (rc0 != GZ_OK) return Fail("gz_encode_rgb_only", rc0);
    Frame f;
    FrameFromOriginal(orig_.data(), w, h, &f);
    if (!WriteJpeg(f, out)) return Fail("WriteJpeg", GZ_E_STATE);
    Log("Original Out[%7zd]", out->size());
    Log(" <image too small for Butteraugli>\n");
    return true;
  }
  int err = 0;
  ctx_ = gz_create(params_.device, w, h, rgb.data(), params_.butteraugli_target, &err);
  if (!ctx_) return Fail("gz_create", err);
  orig_.resize((size_t)3 * nb_ * 64);
  img_.resize(orig_.size());
  int rc = gz_encode_rgb(ctx_, orig_.data());
  if (rc != GZ_OK) return Fail("gz_encode_rgb", rc);
  stats_->timers["create+encode"] = sw.lap();

  QuantMatrix ones;
  for (int c = 0; c < 3; ++c)
    for (int k = 0; k < 64; ++k) ones[c][k] = 1;
  if (!Search(ones, out)) return false;
  stats_->timers["total"] = total.lap();
  return true;
}

}  // namespace

bool Process(const Params& params, ProcessStats* stats, const std::vector<uint8_t>& rgb, int w,
             int h, std::string* out) {
  ProcessStats dummy;
  if (stats == nullptr) stats = &dummy;
  Encoder enc(params, stats);
  return enc.Run(rgb, w, h, out);
}

bool Process(const Params& params, ProcessStats* stats, const std::string& jpeg_data,
             std::string* out) {
  ProcessStats dummy;
  if (stats == nullptr) stats = &dummy;
  Encoder enc(params, stats);
  return enc.RunJpeg(jpeg_data, out);
}

}  // namespace guetzli_amd

// ---------------------------------------------------------------- C wrapper (ctypes) ---
extern "C" {

long gzh_write_jpeg_factor(const int16_t* coeffs, int w, int h, const int* q, int original,
                           int factor, uint8_t* out, long cap);
long gzh_jpeg_head_factor(const uint32_t* counts, const int* q, int w, int h, int ncomp, int factor,
                          uint8_t* head_out, long cap, uint8_t* depth, uint16_t* code);

// Every entry point below catches what the C++ underneath may throw (std::bad_alloc on a huge
// declared image, ...): nothing propagates through the C boundary.  Return values: >= 0 the
// size of the result (copied only if it fits the caller's buffer: a larger size asks for a
// retry with that much room), -1 failure (message on stderr), -2 exception.
#define GZH_GUARD_BEGIN try {
#define GZH_GUARD_END                                                        \
  } catch (const std::exception& e) {                                        \
    fprintf(stderr, "guetzli_amd: %s\n", e.what());                          \
    return -2;                                                               \
  } catch (...) {                                                            \
    fprintf(stderr, "guetzli_amd: unknown exception\n");                     \
    return -2;                                                               \
  }

static void CopyText(const std::string& s, char* dst, long cap) {
  if (!dst || cap <= 0) return;
  const size_t n = std::min<size_t>(s.size(), (size_t)cap - 1);
  memcpy(dst, s.data(), n);
  dst[n] = 0;
}

// guetzli::Process with every field of Params.  jpeg_len < 0: `data` is packed RGB of w x h,
// otherwise JPEG bytes.  quality < 0: `target` is the butteraugli target directly.
// iparams: device, clear_metadata, try_420, force_420, use_silver_screen,
// zeroing_greedy_lookahead, new_zeroing_model.
long gzh_process_params(const uint8_t* data, long jpeg_len, int w, int h, double quality,
                        float target, const int* iparams, uint8_t* out, long cap, char* trace,
                        long trace_cap, char* timers, long timers_cap) {
  GZH_GUARD_BEGIN
  guetzli_amd::Params params;
  params.butteraugli_target =
      quality >= 0 ? (float)guetzli_amd::ButteraugliScoreForQuality(quality) : target;
  params.device = iparams[0];
  params.clear_metadata = iparams[1] != 0;
  params.try_420 = iparams[2] != 0;
  params.force_420 = iparams[3] != 0;
  params.use_silver_screen = iparams[4] != 0;
  params.zeroing_greedy_lookahead = iparams[5];
  params.new_zeroing_model = iparams[6] != 0;
  guetzli_amd::ProcessStats stats;
  std::string dbg;
  if (trace) stats.debug_output = &dbg;
  std::string jpg;
  bool ok;
  if (jpeg_len < 0) {
    static thread_local std::vector<uint8_t> v;   // Process takes a vector, as the reference's does
    v.assign(data, data + (size_t)3 * w * h);
    ok = guetzli_amd::Process(params, &stats, v, w, h, &jpg);
  } else {
    std::string in((const char*)data, (size_t)jpeg_len);
    ok = guetzli_amd::Process(params, &stats, in, &jpg);
  }
  if (!ok) return -1;
  if ((long)jpg.size() <= cap) memcpy(out, jpg.data(), jpg.size());
  CopyText(dbg, trace, trace_cap);
  if (timers && timers_cap > 0) {
    std::string t;
    for (const auto& kv : stats.timers) {
      char buf[128];
      snprintf(buf, sizeof(buf), "%s=%.6f;", kv.first.c_str(), kv.second);
      t += buf;
    }
    for (const auto& kv : stats.counters) {
      char buf[128];
      snprintf(buf, sizeof(buf), "#%s=%d;", kv.first.c_str(), kv.second);
      t += buf;
    }
    CopyText(t, timers, timers_cap);
  }
  return (long)jpg.size();
  GZH_GUARD_END
}

long gzh_process(const uint8_t* rgb, int w, int h, double quality, float target, int device,
                 uint8_t* out, long cap, char* trace, long trace_cap, char* timers,
                 long timers_cap) {
  const int ip[7] = {device, 1, 0, 0, 0, 3, 1};
  return gzh_process_params(rgb, -1, w, h, quality, target, ip, out, cap, trace, trace_cap, timers,
                            timers_cap);
}

// Process(params, stats, jpeg_data, &out); clear_metadata as Params::clear_metadata.
long gzh_process_jpeg(const uint8_t* data, long len, double quality, float target, int device,
                      int clear_metadata, uint8_t* out, long cap, char* trace, long trace_cap) {
  const int ip[7] = {device, clear_metadata, 0, 0, 0, 3, 1};
  return gzh_process_params(data, len, 0, 0, quality, target, ip, out, cap, trace, trace_cap, nullptr, 0);
}

double gzh_butteraugli_score_for_quality(double q) {
  return guetzli_amd::ButteraugliScoreForQuality(q);
}

// Length-limited Huffman depths of a 257-entry histogram (CreateHuffmanTree, entropy_encode.cc:
// 73-145), the way phase B's size model calls it (stream: see jpeg_writer.h).  Test hook.
void gzh_huffman_depths(const uint32_t* counts, int tree_limit, uint8_t* depth, int stream) {
  guetzli_amd::HuffmanDepths(counts, (size_t)guetzli_amd::kHistoSize, tree_limit, depth, stream);
}

// Threads of the driver's worker pool (the calling thread included): min(16, cores the process may
// run on), GZ_HOST_THREADS overrides.  Test hook for the per-rank core share of a multi-GPU run.
int gzh_worker_pool_size() { return guetzli_amd::WorkerPool::Get().size(); }

// ReadJpeg as a canonical dump (test hook; the format: reader_dump.h).  Returns the dump size (copied if it
// fits), or -1 if the stream is rejected.
long gzh_read_jpeg(const uint8_t* data, long len, uint8_t* out, long cap) {
  GZH_GUARD_BEGIN
  guetzli_amd::JpegInput jpg;
  std::string err;
  if (!guetzli_amd::ReadJpeg(data, (size_t)len, &jpg, &err)) return -1;
  const std::string d = guetzli_amd::DumpJpegInput(jpg);
  if ((long)d.size() <= cap) memcpy(out, d.data(), d.size());
  return (long)d.size();
  GZH_GUARD_END
}

// ReadPNG (guetzli.cc:47-152): PNG bytes -> packed RGB with alpha blended on black.  Returns
// 3*w*h (copied to out if it fits) and the dimensions in wh[0..1], or -1 if the stream is
// rejected (message on stderr).
long gzh_read_png(const uint8_t* data, long len, int* wh, uint8_t* out, long cap) {
  GZH_GUARD_BEGIN
  std::vector<uint8_t> rgb;
  std::string err;
  int w = 0, h = 0;
  if (!guetzli_amd::ReadPng(data, (size_t)len, &w, &h, &rgb, &err)) {
    fprintf(stderr, "Error reading PNG data from input file: %s\n", err.c_str());
    return -1;
  }
  wh[0] = w;
  wh[1] = h;
  if ((long)rgb.size() <= cap) memcpy(out, rgb.data(), rgb.size());
  return (long)rgb.size();
  GZH_GUARD_END
}

// WriteJpeg of an image given by dequantised coefficients + quant matrices (test hook).
long gzh_write_jpeg(const int16_t* coeffs, int w, int h, const int* q, int original,
                    uint8_t* out, long cap) {
  return gzh_write_jpeg_factor(coeffs, w, h, q, original, 1, out, cap);
}

// The same for a frame with chroma subsampling factor 1 or 2 (coefficients in the frame layout
// of include/guetzli_amd.h).
long gzh_write_jpeg_factor(const int16_t* coeffs, int w, int h, const int* q, int original,
                           int factor, uint8_t* out, long cap) {
  GZH_GUARD_BEGIN
  guetzli_amd::Frame f;
  if (original) {
    guetzli_amd::FrameFromOriginal(coeffs, w, h, &f);
  } else {
    int qq[3][64];
    memcpy(qq, q, sizeof(qq));
    guetzli_amd::FrameFromImageFactor(coeffs, qq, w, h, factor, &f);
  }
  std::string s;
  if (!guetzli_amd::WriteJpeg(f, &s)) return -1;
  if ((long)s.size() <= cap) memcpy(out, s.data(), s.size());
  return (long)s.size();
  GZH_GUARD_END
}

// Marker segments + Huffman codes from symbol counts (test hook of BuildJpegHead):
// counts uint32 [2][3][256] as gz_jpeg_histograms returns them; q null = the q=1 original.
// Returns the head length (bytes copied to head_out if they fit) or -1.
long gzh_jpeg_head(const uint32_t* counts, const int* q, int w, int h, int ncomp,
                   uint8_t* head_out, long cap, uint8_t* depth /*[2][3][256]*/,
                   uint16_t* code /*[2][3][256]*/) {
  return gzh_jpeg_head_factor(counts, q, w, h, ncomp, 1, head_out, cap, depth, code);
}

long gzh_jpeg_head_factor(const uint32_t* counts, const int* q, int w, int h, int ncomp, int factor,
                          uint8_t* head_out, long cap, uint8_t* depth /*[2][3][256]*/,
                          uint16_t* code /*[2][3][256]*/) {
  GZH_GUARD_BEGIN
  guetzli_amd::SymbolHistogram dc[3], ac[3];
  for (int c = 0; c < 3; ++c)
    for (int i = 0; i < 256; ++i) {
      dc[c].Add(i, (int)counts[(0 * 3 + c) * 256 + i]);
      ac[c].Add(i, (int)counts[(1 * 3 + c) * 256 + i]);
    }
  guetzli_amd::Frame f;
  int qq[3][64];
  if (q) memcpy(qq, q, sizeof(qq));
  guetzli_amd::FrameTablesFactor(q ? qq : nullptr, w, h, ncomp, factor, &f);
  guetzli_amd::JpegHead head;
  if (!guetzli_amd::BuildJpegHead(f, dc, ac, &head)) return -1;
  if ((long)head.bytes.size() <= cap) memcpy(head_out, head.bytes.data(), head.bytes.size());
  memcpy(depth, head.depth, sizeof(head.depth));
  memcpy(code, head.code, sizeof(head.code));
  return (long)head.bytes.size();
  GZH_GUARD_END
}

}  // extern "C"
