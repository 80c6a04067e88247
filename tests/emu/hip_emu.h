// TEST INFRASTRUCTURE ONLY.
//
// A minimal single-process emulation of the HIP execution model, so that the VERY SAME
// kernel sources under guetzli_amd/csrc/ can be compiled with g++ and run on the CPU in
// the `-m "not gpu"` test suite (tests/emu/build_emu.py -> tests/emu/_build/*.so).
// Purpose: catch indexing / tiling / accumulation-order bugs against the oracle before
// spending GPU minutes.  It is NOT a CPU fallback: the product library is only ever
// built by hipcc for gfx950 and fails loudly without a GPU; nothing under guetzli_amd/
// references this file, and the emulated library has a different name and lives under
// tests/.
//
// Model: one OS thread; each workgroup's threads are fibers scheduled round-robin between
// __syncthreads() barriers; __shared__ is function-static storage (one workgroup runs at a
// time); atomics are plain operations.  The fiber switch is a dozen instructions of our own on
// x86-64 (glibc's swapcontext makes two rt_sigprocmask system calls per switch, and a workgroup
// of 256 threads switches thousands of times per launch: the whole CPU suite ran three times as
// long with it); other hosts keep ucontext.
#pragma once
#if !defined(__x86_64__)
#include <ucontext.h>
#endif

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace hipemu {
#if defined(__x86_64__)
// gz_emu_switch(&save, load): pushes the callee-saved registers, stores the stack pointer in
// *save, continues on the stack `load` (whose top holds six register values and a return address).
extern "C" void gz_emu_switch(void** save_sp, void* load_sp);
__asm__(
    ".text\n"
    ".p2align 4\n"
    ".weak gz_emu_switch\n"
    ".hidden gz_emu_switch\n"
    ".type gz_emu_switch,@function\n"
    "gz_emu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size gz_emu_switch,.-gz_emu_switch\n");
struct Fiber {
  void* sp = nullptr;
};
#else
typedef ucontext_t Fiber;
#endif
struct State {
  dim3 threadIdx, blockIdx, blockDim, gridDim;
  Fiber sched;
  std::vector<Fiber> fibers;
  std::vector<char*> stacks;
  std::vector<char> done;
  const std::function<void()>* body = nullptr;
  int current = -1;
  static const size_t kStack = 256 * 1024;
};
inline State& st() {
  static State s;
  return s;
}
// the running fiber <-> the scheduler
inline void to_sched() {
  State& s = st();
#if defined(__x86_64__)
  gz_emu_switch(&s.fibers[s.current].sp, s.sched.sp);
#else
  swapcontext(&s.fibers[s.current], &s.sched);
#endif
}
inline void to_fiber(int t) {
  State& s = st();
#if defined(__x86_64__)
  gz_emu_switch(&s.sched.sp, s.fibers[t].sp);
#else
  swapcontext(&s.sched, &s.fibers[t]);
#endif
}
inline void fiber_entry() {
  State& s = st();
  (*s.body)();
  s.done[s.current] = 1;
  to_sched();
  abort();   // (a finished fiber is never resumed)
}
inline void fiber_init(int t) {
  State& s = st();
#if defined(__x86_64__)
  // top of the stack, 16-byte aligned: six zeroed registers, the entry as return address, and
  // a slot that leaves the stack pointer where a call would have left it at fiber_entry
  uintptr_t top = ((uintptr_t)s.stacks[t] + State::kStack) & ~(uintptr_t)15;
  void** sp = (void**)(top - 64);
  for (int i = 0; i < 6; ++i) sp[i] = nullptr;
  sp[6] = (void*)&fiber_entry;
  sp[7] = nullptr;
  s.fibers[t].sp = sp;
#else
  getcontext(&s.fibers[t]);
  s.fibers[t].uc_stack.ss_sp = s.stacks[t];
  s.fibers[t].uc_stack.ss_size = State::kStack;
  s.fibers[t].uc_link = nullptr;
  makecontext(&s.fibers[t], (void (*)())fiber_entry, 0);
#endif
}
inline void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  State& s = st();
  const int nt = (int)(block.x * block.y * block.z);
  if ((int)s.fibers.size() < nt) {
    s.fibers.resize(nt);
    s.done.resize(nt);
    while ((int)s.stacks.size() < nt) s.stacks.push_back((char*)malloc(State::kStack));
  }
  s.body = &body;
  s.blockDim = block;
  s.gridDim = grid;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        s.blockIdx = dim3(bx, by, bz);
        for (int t = 0; t < nt; ++t) {
          fiber_init(t);
          s.done[t] = 0;
        }
        int alive = nt;
        while (alive > 0) {
          alive = 0;
          for (int t = 0; t < nt; ++t) {
            if (s.done[t]) continue;
            s.current = t;
            s.threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            to_fiber(t);
            if (!s.done[t]) ++alive;
          }
        }
      }
  s.body = nullptr;
}
inline int linear_tid() {
  State& s = st();
  return (int)(s.threadIdx.x + s.blockDim.x * (s.threadIdx.y + s.blockDim.y * s.threadIdx.z));
}
inline int num_threads() {
  State& s = st();
  return (int)(s.blockDim.x * s.blockDim.y * s.blockDim.z);
}
}  // namespace hipemu

#define threadIdx (hipemu::st().threadIdx)
#define blockIdx (hipemu::st().blockIdx)
#define blockDim (hipemu::st().blockDim)
#define gridDim (hipemu::st().gridDim)

inline void __syncthreads() { hipemu::to_sched(); }
inline void __threadfence() {}   // (one fiber runs at a time: memory is always consistent)

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __constant__ static

// ---- atomics (single-threaded) ----
inline unsigned atomicMax(unsigned* p, unsigned v) {
  unsigned o = *p;
  if (v > o) *p = v;
  return o;
}
inline unsigned atomicMin(unsigned* p, unsigned v) {
  unsigned o = *p;
  if (v < o) *p = v;
  return o;
}
inline int atomicAdd(int* p, int v) {
  int o = *p;
  *p += v;
  return o;
}
inline unsigned atomicAdd(unsigned* p, unsigned v) {
  unsigned o = *p;
  *p += v;
  return o;
}
inline unsigned __float_as_uint(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  return u;
}
inline float __uint_as_float(unsigned u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}

inline unsigned atomicOr(unsigned* p, unsigned v) {
  unsigned o = *p;
  *p |= v;
  return o;
}
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) {
  unsigned long long o = *p;
  *p += v;
  return o;
}

// ---- wavefront operations (wave64) ----
// Emulated with two scheduler round trips (every fiber publishes, every fiber reads), so
// they behave like the hardware's convergent lane exchange PROVIDED all threads of the
// workgroup execute them together -- which is how the product kernels use them.
namespace hipemu {
inline long long* wave_slots() {
  static long long slots[1024];
  return slots;
}
inline void yield() { to_sched(); }
}  // namespace hipemu
inline unsigned long long __ballot(int pred) {
  const int tid = hipemu::linear_tid();
  hipemu::wave_slots()[tid] = pred ? 1 : 0;
  hipemu::yield();
  const int base = tid & ~63;
  const int nt = hipemu::num_threads();
  unsigned long long m = 0;
  for (int i = 0; i < 64 && base + i < nt; ++i)
    if (hipemu::wave_slots()[base + i]) m |= 1ull << i;
  hipemu::yield();
  return m;
}
inline int __shfl(int v, int src_lane) {
  const int tid = hipemu::linear_tid();
  hipemu::wave_slots()[tid] = v;
  hipemu::yield();
  const int r = (int)hipemu::wave_slots()[(tid & ~63) + (src_lane & 63)];
  hipemu::yield();
  return r;
}
inline int __shfl_up(int v, unsigned delta) {
  const int tid = hipemu::linear_tid();
  hipemu::wave_slots()[tid] = v;
  hipemu::yield();
  const int lane = tid & 63;
  const int r = lane >= (int)delta ? (int)hipemu::wave_slots()[tid - (int)delta] : v;
  hipemu::yield();
  return r;
}
inline int __clzll(long long x) { return x == 0 ? 64 : __builtin_clzll((unsigned long long)x); }
inline int __clz(int x) { return x == 0 ? 32 : __builtin_clz((unsigned)x); }

// ---- runtime shims ----
typedef int hipError_t;
typedef void* hipStream_t;
struct hipEmuEvent { std::chrono::steady_clock::time_point t; };
typedef hipEmuEvent* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice,
                     hipMemcpyHostToHost, hipMemcpyDefault };
inline const char* hipGetErrorString(hipError_t e) { return e == hipErrorOutOfMemory ? "out of memory" : e ? "emu error" : "success"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
// Allocation bookkeeping and fault injection (tests/test_hip_failures.py): every device / page-locked allocation
// is counted and listed with its size; gz_emu_fail_alloc(n) makes the (n + 1)-th allocation from now on fail with
// hipErrorOutOfMemory (once), gz_emu_live() reports what is still allocated -- a failed gz_create or a failed
// encode must leave nothing behind.
namespace hipemu {
struct AllocBook {
  std::mutex mu;
  std::map<void*, size_t> device, host;
  long calls = 0, fail_at = -1, events = 0;
};
inline AllocBook& alloc_book() { static AllocBook b; return b; }
inline bool alloc_should_fail() {   // (the caller holds the lock)
  AllocBook& b = alloc_book();
  const long k = b.calls++;
  if (b.fail_at >= 0 && k == b.fail_at) { b.fail_at = -1; return true; }
  return false;
}
}  // namespace hipemu
extern "C" __attribute__((used, visibility("default"))) inline void gz_emu_fail_alloc(long n) {
  hipemu::AllocBook& b = hipemu::alloc_book();
  std::lock_guard<std::mutex> lk(b.mu);
  b.fail_at = n < 0 ? -1 : b.calls + n;
}
extern "C" __attribute__((used, visibility("default"))) inline long gz_emu_alloc_calls(void) {
  hipemu::AllocBook& b = hipemu::alloc_book();
  std::lock_guard<std::mutex> lk(b.mu);
  return b.calls;
}
extern "C" __attribute__((used, visibility("default"))) inline void gz_emu_live(long* device_bytes, long* host_bytes, long* blocks,
                                                                                long* events) {
  hipemu::AllocBook& b = hipemu::alloc_book();
  std::lock_guard<std::mutex> lk(b.mu);
  long d = 0, h = 0;
  for (auto& kv : b.device) d += (long)kv.second;
  for (auto& kv : b.host) h += (long)kv.second;
  if (device_bytes) *device_bytes = d;
  if (host_bytes) *host_bytes = h;
  if (blocks) *blocks = (long)(b.device.size() + b.host.size());
  if (events) *events = b.events;
}
inline hipError_t hipMalloc(void** p, size_t n) {
  hipemu::AllocBook& b = hipemu::alloc_book();
  std::lock_guard<std::mutex> lk(b.mu);
  *p = nullptr;
  if (hipemu::alloc_should_fail()) return hipErrorOutOfMemory;
  *p = malloc(n ? n : 1);
  // poison so that reads of never-written device memory are visible in tests
  if (*p) { memset(*p, 0xCD, n); b.device[*p] = n; }
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) {
  if (p) { hipemu::AllocBook& b = hipemu::alloc_book(); std::lock_guard<std::mutex> lk(b.mu); b.device.erase(p); }
  free(p);
  return hipSuccess;
}
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) {
  hipemu::AllocBook& b = hipemu::alloc_book();
  std::lock_guard<std::mutex> lk(b.mu);
  *p = nullptr;
  if (hipemu::alloc_should_fail()) return hipErrorOutOfMemory;
  *p = malloc(n ? n : 1);
  if (*p) b.host[*p] = n;
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
template <class T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipHostMalloc((void**)p, n, f); }
inline hipError_t hipHostFree(void* p) {
  if (p) { hipemu::AllocBook& b = hipemu::alloc_book(); std::lock_guard<std::mutex> lk(b.mu); b.host.erase(p); }
  free(p);
  return hipSuccess;
}
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline void hipemu_count_event(int d) { hipemu::AllocBook& b = hipemu::alloc_book(); std::lock_guard<std::mutex> lk(b.mu); b.events += d; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipEmuEvent; hipemu_count_event(+1); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { if (e) hipemu_count_event(-1); delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hipEmuEvent; hipemu_count_event(+1); return hipSuccess; }
enum { hipEventDisableTiming = 2 };
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
