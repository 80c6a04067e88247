// A small persistent worker pool for the host search driver's O(candidates) passes (building
// and partitioning the global candidate order of phase B).  The reference is single-threaded;
// every pass parallelised here produces exactly the sequence its serial form produces.
// GZ_HOST_THREADS overrides the worker count (default: min(16, the cores this PROCESS may run on:
// sched_getaffinity, not hardware_concurrency -- one process per GPU keeps to its share of the
// host cores, bench.py's Env.bind_cpus, and a pool sized for the whole machine would put 8 x 16
// workers on a node with few cores per GPU).
#pragma once
#include <sched.h>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace guetzli_amd {

class WorkerPool {
 public:
  static WorkerPool& Get() {
    static WorkerPool pool;
    return pool;
  }
  int size() const { return (int)workers_.size() + 1; }

  // Runs fn(i) for i in [0, n); returns when all are done.  The calling thread takes part.
  void Run(int n, const std::function<void(int)>& fn) {
    if (n <= 0) return;
    if (n == 1 || workers_.empty()) {
      for (int i = 0; i < n; ++i) fn(i);
      return;
    }
    // One job at a time: when several images are encoded concurrently (one host thread per
    // image, batch mode) a caller that finds the pool busy simply runs its job itself.
    std::unique_lock<std::mutex> job(job_mu_, std::try_to_lock);
    if (!job.owns_lock()) {
      for (int i = 0; i < n; ++i) fn(i);
      return;
    }
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &fn;
      next_ = 0;
      total_ = n;
      pending_ = n;
      ++generation_;
    }
    cv_.notify_all();
    Drain();
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return pending_ == 0; });
    fn_ = nullptr;
  }

  // Cores the calling process may be scheduled on (its affinity mask at the pool's creation).
  static int AllowedCpus() {
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
      int n = CPU_COUNT(&set);
      if (n > 0) return n;
    }
    return (int)std::thread::hardware_concurrency();
  }

 private:
  WorkerPool() {
    int n = AllowedCpus();
    if (n <= 0) n = 1;
    if (n > 16) n = 16;
    if (const char* e = getenv("GZ_HOST_THREADS")) n = atoi(e) > 0 ? atoi(e) : n;
    for (int i = 1; i < n; ++i) workers_.emplace_back([this] { Loop(); });
  }
  ~WorkerPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }
  void Drain() {
    for (;;) {
      int i;
      const std::function<void(int)>* fn;
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (fn_ == nullptr || next_ >= total_) return;
        i = next_++;
        fn = fn_;
      }
      (*fn)(i);
      std::lock_guard<std::mutex> lk(mu_);
      if (--pending_ == 0) done_cv_.notify_all();
    }
  }
  void Loop() {
    unsigned long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || generation_ != seen; });
        if (stop_) return;
        seen = generation_;
      }
      Drain();
    }
  }

  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::mutex job_mu_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(int)>* fn_ = nullptr;
  int next_ = 0, total_ = 0, pending_ = 0;
  unsigned long generation_ = 0;
  bool stop_ = false;
};

}  // namespace guetzli_amd
