// Round-trip latency of "tiny kernel -> 8 bytes to the host -> wait", the pattern phase B's
// host driver repeats ~10 times per iteration.  Variants: pageable vs pinned destination,
// copy vs kernel writing to mapped host memory, stream vs event wait, blocking vs spin flags.
//   hipcc --offload-arch=gfx950 -O2 -o sync sync.hip ; ./sync [flags: 0 auto, 1 spin, 2 yield, 4 blocking]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void k_tiny(unsigned long long* out, unsigned long long v) { out[0] = v; }

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
  const unsigned flags = argc > 1 ? (unsigned)atoi(argv[1]) : 0;
  if (flags) hipSetDeviceFlags(flags);
  hipStream_t s;
  hipStreamCreate(&s);
  unsigned long long* d;
  hipMalloc(&d, 64);
  unsigned long long* pinned;
  hipHostMalloc(&pinned, 64, hipHostMallocMapped);
  unsigned long long* dev_view;
  hipHostGetDevicePointer((void**)&dev_view, pinned, 0);
  unsigned long long pageable = 0;
  hipEvent_t ev;
  hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  const int N = 2000;
  for (int variant = 0; variant < 6; ++variant) {
    for (int rep = 0; rep < 2; ++rep) {   // rep 0 warms up
      const double t0 = now();
      for (int i = 0; i < N; ++i) {
        switch (variant) {
          case 0:   // copy to pageable, stream wait
            hipLaunchKernelGGL(k_tiny, dim3(1), dim3(1), 0, s, d, (unsigned long long)i);
            hipMemcpyAsync(&pageable, d, 8, hipMemcpyDeviceToHost, s);
            hipStreamSynchronize(s);
            break;
          case 1:   // copy to pinned, stream wait
            hipLaunchKernelGGL(k_tiny, dim3(1), dim3(1), 0, s, d, (unsigned long long)i);
            hipMemcpyAsync(pinned, d, 8, hipMemcpyDeviceToHost, s);
            hipStreamSynchronize(s);
            break;
          case 2:   // kernel writes mapped host memory, stream wait
            hipLaunchKernelGGL(k_tiny, dim3(1), dim3(1), 0, s, dev_view, (unsigned long long)i);
            hipStreamSynchronize(s);
            break;
          case 3:   // kernel writes mapped host memory, event wait
            hipLaunchKernelGGL(k_tiny, dim3(1), dim3(1), 0, s, dev_view, (unsigned long long)i);
            hipEventRecord(ev, s);
            hipEventSynchronize(ev);
            break;
          case 4: { // kernel writes mapped host memory, host polls the value
            hipLaunchKernelGGL(k_tiny, dim3(1), dim3(1), 0, s, dev_view, (unsigned long long)(i + 1) << 32 | rep);
            const unsigned long long want = (unsigned long long)(i + 1) << 32 | rep;
            while (*(volatile unsigned long long*)pinned != want) {}
            break;
          }
          default:  // kernel only, stream wait (no result)
            hipLaunchKernelGGL(k_tiny, dim3(1), dim3(1), 0, s, d, (unsigned long long)i);
            hipStreamSynchronize(s);
            break;
        }
      }
      const double dt = now() - t0;
      static const char* names[] = {"kernel + copy to pageable + stream wait", "kernel + copy to pinned + stream wait",
                                    "kernel writes mapped host + stream wait", "kernel writes mapped host + event wait",
                                    "kernel writes mapped host + host polls", "kernel + stream wait (no result)"};
      if (rep) printf("flags %u: %-44s %7.1f us per round trip\n", flags, names[variant], dt / N * 1e6);
    }
  }
  return 0;
}
