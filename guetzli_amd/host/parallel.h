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
#include <cstdio>
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

  // Cores' worth of CPU time the calling process may use: its affinity mask, capped by the container's CPU
  // quota (cgroup v2 cpu.max / v1 cfs quota).  The MI355X boxes of round 6 show 256 logical CPUs in the mask
  // and a quota of 16: threads sized by the mask alone (three spinning code-refresh helpers for each of four
  // images in flight, a 16-thread pool) run into the quota's throttling.
  static int AllowedCpus() {
    int n = 0;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = (int)std::thread::hardware_concurrency();
    static const int quota = CpuQuota();
    if (quota > 0 && quota < n) n = quota;
    return n;
  }
  // ceil(quota / period) of the cgroup this process runs in; 0 = no quota (or unreadable)
  static int CpuQuota() {
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[32] = {0};
      long period = 0;
      const int got = fscanf(f, "%31s %ld", q, &period);
      fclose(f);
      if (got == 2 && period > 0 && q[0] != 'm') {
        const long v = atol(q);
        if (v > 0) return (int)((v + period - 1) / period);
      }
      return 0;
    }
    long q = 0, p = 0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(f, "%ld", &q) != 1) q = 0; fclose(f); }
    if (FILE* f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(f, "%ld", &p) != 1) p = 0; fclose(f); }
    return q > 0 && p > 0 ? (int)((q + p - 1) / p) : 0;
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
