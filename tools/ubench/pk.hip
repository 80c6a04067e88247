// Issue rate and dependent latency of packed f32 arithmetic (v_pk_mul_f32 / v_pk_add_f32) against
// the scalar v_mul_f32 / v_add_f32 on gfx950.  Every kernel runs NOUT accumulators (scalar) or
// NOUT/2 accumulator pairs (packed) through `iters` x 32 multiply-then-add steps without memory
// traffic; ILP = number of independent chains a wave interleaves.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -mllvm -vectorize-slp=false pk.hip -o pk
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f2 __attribute__((vector_size(8)));
struct K { float k[32]; };

template <int ILP>
__global__ __launch_bounds__(256) void k_scalar(float* out, K kk, int iters) {
  float acc[ILP], x[ILP];
  for (int i = 0; i < ILP; ++i) { acc[i] = 0.0f; x[i] = 1.0f + threadIdx.x * 1e-3f + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 32; ++j)
#pragma unroll
      for (int i = 0; i < ILP; ++i) acc[i] = acc[i] + x[i] * kk.k[j];
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = acc[i] * 1e-3f;
  }
  float s = 0;
  for (int i = 0; i < ILP; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int ILP>   // ILP accumulator PAIRS
__global__ __launch_bounds__(256) void k_packed(float* out, K kk, int iters) {
  f2 acc[ILP], x[ILP];
  for (int i = 0; i < ILP; ++i) { acc[i] = f2{0.0f, 0.0f}; x[i] = f2{1.0f + threadIdx.x * 1e-3f + i, 2.0f + i}; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 32; ++j)
#pragma unroll
      for (int i = 0; i < ILP; ++i) acc[i] = acc[i] + x[i] * f2{kk.k[j], kk.k[j]};
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = acc[i] * f2{1e-3f, 1e-3f};
  }
  f2 s = {0, 0};
  for (int i = 0; i < ILP; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1];
}
template <class F>
static double time_ms(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  K kk; for (int i = 0; i < 32; ++i) kk.k[i] = 0.01f * (i + 1);
  const int iters = 2000;
  for (int wg : {256, 1024, 2048}) {
    // lane-operations: wg*256 threads * iters * 32 steps * 2 ops * chains
#define RUN(name, kern, chains)                                                       \
    { double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(wg), dim3(256), 0, 0, out, kk, iters); }); \
      double ops = (double)wg * 256 * iters * 32 * 2 * (chains);                     \
      printf("%-22s wg %4d: %8.3f ms  %7.2f T lane-ops/s\n", name, wg, ms, ops / ms * 1e-9); }
    RUN("scalar ILP1", (k_scalar<1>), 1)
    RUN("scalar ILP4", (k_scalar<4>), 4)
    RUN("scalar ILP8", (k_scalar<8>), 8)
    RUN("packed ILP1 (2 ch)", (k_packed<1>), 2)
    RUN("packed ILP2 (4 ch)", (k_packed<2>), 4)
    RUN("packed ILP4 (8 ch)", (k_packed<4>), 8)
    RUN("packed ILP8 (16 ch)", (k_packed<8>), 16)
  }
  return 0;
}
