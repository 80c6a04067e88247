// The Huffman codes of phase B's size model, computed one refresh ahead of the slow steps.
//
// The reference refreshes the code lengths every 10th coefficient step (ComputeEntropyCodes,
// processor.cc:741-743) from the symbol statistics as they are after that step.  Those
// statistics are a function of the order's next entries alone -- which block steps, and to
// which coefficient -- so while the driver thread walks steps i + 1 .. i + 10 (each with its
// size estimate and the stopping rule, in the reference's sequence) a helper thread replays the
// same ten steps on private copies of the blocks and the statistics and has the codes of step
// i + 10 ready when the driver gets there.  Nothing speculative is ever observed: a result is
// only taken for exactly the step it was computed for, and one computed beyond the step the
// stopping rule fires at is dropped.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>

#include "jpeg_writer.h"

namespace guetzli_amd {

class CodesAhead {
 public:
  static const int kMaxSteps = 10;
  struct Step {
    int slot;         // which private block
    int c, k;         // component, coefficient
    int16_t newval;
    bool keep;        // a "precious" coefficient that would become 0 keeps its value
  };
  struct Job {
    // in
    size_t (*codes)(const SymbolHistogram*, int, uint8_t*) = nullptr;   // EntropyCodes of processor.cc
    SymbolHistogram histo[3];          // statistics after the step the driver is at
    int ncomp = 0;
    const int* q[3] = {nullptr, nullptr, nullptr};
    int nsteps = 0;
    Step steps[kMaxSteps];
    int nblocks = 0;
    int16_t blocks[kMaxSteps][64];     // private copies of the blocks the steps touch
    // out
    uint8_t depths[3 * kHistoSize];
    int header = 0;
    int64_t raw_bits[3] = {0, 0, 0};   // HistogramRawBits of the replayed statistics under `depths`
  };

  CodesAhead() {}
  ~CodesAhead() {
    if (!thread_.joinable()) return;
    {
      std::lock_guard<std::mutex> lk(mu_);
      quit_.store(true, std::memory_order_release);
      armed_.store(true, std::memory_order_release);
    }
    cv_.notify_all();
    thread_.join();
  }
  CodesAhead(const CodesAhead&) = delete;
  CodesAhead& operator=(const CodesAhead&) = delete;

  Job& job() { return job_; }   // (the driver fills it between Wait / Drop and Post)

  // The helper leaves its sleep and polls for jobs until Rest(): called a phase ahead of the
  // first Post, so that the wake-up (tens of microseconds) is never waited for.
  void Arm() {
    if (!thread_.joinable()) thread_ = std::thread([this] { Loop(); });
    if (armed_.load(std::memory_order_acquire)) return;
    {
      std::lock_guard<std::mutex> lk(mu_);
      armed_.store(true, std::memory_order_release);
    }
    cv_.notify_one();
  }
  void Rest() {
    Drop();
    armed_.store(false, std::memory_order_release);
  }
  void Post() {
    pending_ = true;
    done_.store(false, std::memory_order_relaxed);
    posted_.store(true, std::memory_order_release);
  }
  bool pending() const { return pending_; }
  // The posted job's outputs are valid after this.
  void Wait() {
    if (!pending_) return;
    while (!done_.load(std::memory_order_acquire)) Pause();
    pending_ = false;
  }
  void Drop() { Wait(); }   // (a job in flight owns job(): it has to finish before the next one is filled)

 private:
  static void Pause() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
  }
  void Loop() {
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return armed_.load(std::memory_order_acquire); });
        if (quit_.load(std::memory_order_acquire)) return;
      }
      unsigned idle = 0;
      while (armed_.load(std::memory_order_acquire) && !quit_.load(std::memory_order_acquire)) {
        if (!posted_.load(std::memory_order_acquire)) {
          // (jobs follow each other within microseconds while the driver is in its slow steps;
          // during its device calls the helper lets other threads have the core)
          if (++idle < 4096) Pause(); else std::this_thread::yield();
          continue;
        }
        idle = 0;
        posted_.store(false, std::memory_order_relaxed);
        Work();
        done_.store(true, std::memory_order_release);
      }
      // (Rest() has waited for the last job: nothing is posted while the helper sleeps)
    }
  }
  void Work() {
    Job& j = job_;
    for (int s = 0; s < j.nsteps; ++s) {
      const Step& st = j.steps[s];
      int16_t* blk = j.blocks[st.slot];
      AddBlockACSymbols(blk, j.q[st.c], -1, &j.histo[st.c]);
      if (!st.keep) blk[st.k] = st.newval;
      AddBlockACSymbols(blk, j.q[st.c], 1, &j.histo[st.c]);
    }
    j.header = (int)j.codes(j.histo, j.ncomp, j.depths);
    for (int c = 0; c < j.ncomp; ++c) j.raw_bits[c] = HistogramRawBits(j.histo[c], &j.depths[c * kHistoSize]);
  }

  Job job_;
  std::thread thread_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::atomic<bool> armed_{false}, posted_{false}, done_{false}, quit_{false};
  bool pending_ = false;   // driver thread only
};

}  // namespace guetzli_amd
