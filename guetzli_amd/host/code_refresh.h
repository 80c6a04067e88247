// Phase B's size model refreshes its Huffman codes after every tenth coefficient step
// (ComputeEntropyCodes, processor.cc:739-741 / :497-525): three to five length-limited code
// constructions, 4 us, between two steps that cost 0.2 us each -- two thirds of the serial steps'
// time at 4K.  A refresh is a PURE function of the AC symbol statistics at its step, so it does not
// have to be computed by the thread that takes the steps: the search driver takes the steps of the
// next windows while helper threads construct the codes of the windows behind, and evaluates every
// step's size -- in order, on its own thread, with the arithmetic it always used -- when the
// window's codes arrive.  The steps it took beyond the stopping point are undone.  Nothing the
// helpers compute depends on timing or on which thread computes it; every decision stays on the
// driver's thread.
//
// Hand-over without system calls on the fast path: a ring of slots, one atomic sequence number per
// direction (release / acquire); helper t serves the windows t, t + T, t + 2T, ... of the encode's
// window sequence.  Between two iterations of phase B the helpers sleep on a condition variable
// (the driver wakes them when the next iteration's order arrives, a few hundred microseconds before
// its first refresh) and poll only while an iteration's steps are being taken.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "jpeg_writer.h"

namespace guetzli_amd {

struct CodeRefresh {
  SymbolHistogram histo[3];          // in: the statistics right after the refresh step
  int ncomp;                         // in
  uint8_t depths[3 * kHistoSize];    // out: EntropyCodes
  int ac_header;                     // out: its return value
  int64_t raw_bits[3];               // out: HistogramRawBits(histo[c], depths of c), c < ncomp
};

class CodeRefreshers {
 public:
  static const int kSlots = 8;       // windows in flight: at most threads() + 1

  explicit CodeRefreshers(int threads) {
    for (int s = 0; s < kSlots; ++s) {
      slot_[s].submitted.store(0, std::memory_order_relaxed);
      slot_[s].done.store(0, std::memory_order_relaxed);
    }
    try {
      for (int t = 0; t < threads; ++t) threads_.emplace_back([this, t, threads] { Loop(t, threads); });
    } catch (...) {   // (a thread could not be started: stop the ones that were, then let the caller know)
      Stop();
      throw;
    }
  }
  ~CodeRefreshers() { Stop(); }
  CodeRefreshers(const CodeRefreshers&) = delete;
  CodeRefreshers& operator=(const CodeRefreshers&) = delete;

  int threads() const { return (int)threads_.size(); }

  // The helpers poll from here on / sleep again (every submitted window has been waited for).
  void Activate() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      active_.store(true, std::memory_order_release);
    }
    cv_.notify_all();
  }
  void Deactivate() { active_.store(false, std::memory_order_release); }

  // Window numbers run through the whole encode: 0, 1, 2, ... in the order of submission.
  long NextWindow() const { return next_; }
  CodeRefresh* Input(long w) { return &slot_[w % kSlots].r; }   // to be filled before Submit(w)
  void Submit(long w) {
    if (!active_.load(std::memory_order_relaxed)) Activate();   // (a sleeping helper would never see it)
    slot_[w % kSlots].submitted.store(w + 1, std::memory_order_release);
    next_ = w + 1;
  }
  // The finished refresh of window w (every submitted window must be waited for, in any order).
  const CodeRefresh* Wait(long w) {
    Slot& s = slot_[w % kSlots];
    // (a helper that lost its core to another thread of an oversubscribed machine gets it back
    // sooner when the waiting side gives its own up: after ~10 us of polling)
    for (unsigned spins = 0; s.done.load(std::memory_order_acquire) != w + 1; ++spins) {
      if (spins < 1024) Pause(); else std::this_thread::yield();
    }
    return &s.r;
  }

 private:
  void Stop() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_.store(true, std::memory_order_release);
    }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
    threads_.clear();
  }
  struct alignas(64) Slot {
    std::atomic<long> submitted;   // window number + 1 whose input the slot holds
    std::atomic<long> done;        // window number + 1 whose output the slot holds
    CodeRefresh r;
  };
  static void Pause() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
  }
  void Loop(int t, int stride) {
    long w = t;
    unsigned idle = 0;
    for (;;) {
      Slot& s = slot_[w % kSlots];
      if (s.submitted.load(std::memory_order_acquire) == w + 1) {
        CodeRefresh& r = s.r;
        r.ac_header = (int)EntropyCodes(r.histo, r.ncomp, r.depths);
        for (int c = 0; c < 3; ++c)
          r.raw_bits[c] = c < r.ncomp ? HistogramRawBits(r.histo[c], &r.depths[c * kHistoSize]) : 0;
        s.done.store(w + 1, std::memory_order_release);
        w += stride;
        idle = 0;
        continue;
      }
      if (stop_.load(std::memory_order_acquire)) return;
      if (active_.load(std::memory_order_acquire)) {
        if (++idle < 65536) Pause(); else std::this_thread::yield();
        continue;
      }
      std::unique_lock<std::mutex> lk(mu_);
      cv_.wait(lk, [&] {
        return stop_.load(std::memory_order_acquire) || active_.load(std::memory_order_acquire);
      });
    }
  }

  Slot slot_[kSlots];
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::atomic<bool> active_{false};
  std::atomic<bool> stop_{false};
  long next_ = 0;
};

}  // namespace guetzli_amd
