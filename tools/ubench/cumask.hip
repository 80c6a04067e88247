// Which CUs a hipExtStreamCreateWithCUMask bit names on this part, and what a masked stream costs.
//   ./cumask map LO HI      workgroups of a launch on a stream masked to bits [LO, HI): per-XCD counts,
//                           distinct (XCD, SE, CU) seen, and the time of a streaming copy on it
//   ./cumask mod8 K         the same with the bits i = K (mod 8) -- one XCD only under the driver's
//                           round-robin deal of the bits; MAY HANG if a dispatch still sends workgroups
//                           to the XCDs that have no CU: run under `timeout`, last
//   ./cumask pair LO1 HI1 LO2 HI2   two streams on disjoint CU ranges running the same VALU-bound kernel
//                           at once vs one after the other vs both unmasked (does partitioning isolate?)
// Background: guetzli_amd/csrc/api/context.h (cu_plan): a batch's images on disjoint CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_where(unsigned* out, int spin) {
  // HW_REG_HW_ID = 4 (gfx9: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 ...), HW_REG_XCC_ID = 20
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
  const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
  float a = (float)threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc | (a == 12345.0f ? 0x80000000u : 0u);
  }
}
__global__ void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_valu(float* out, int iters) {
  float a = (float)threadIdx.x, b = 1.0f;
  for (int i = 0; i < iters; ++i) { a = a * 1.0001f; b = b + a; a = a + 0.25f; b = b * 0.9999f; }
  if (a + b == 12345.0f) out[0] = a;
}

static hipError_t masked_stream(hipStream_t* s, const std::vector<int>& bits, int ncu) {
  std::vector<uint32_t> m((size_t)(ncu + 31) / 32, 0u);
  for (int i : bits) if (i >= 0 && i < ncu) m[(size_t)i >> 5] |= 1u << (i & 31);
  return hipExtStreamCreateWithCUMask(s, (uint32_t)m.size(), m.data());
}

static int report(hipStream_t s, const char* what) {
  const int nwg = 4096;
  unsigned* d;
  CK(hipMalloc(&d, sizeof(unsigned) * 2 * nwg));
  hipLaunchKernelGGL(k_where, dim3(nwg), dim3(256), 0, s, d, 2000);
  CK(hipStreamSynchronize(s));
  std::vector<unsigned> h(2 * nwg);
  CK(hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * nwg, hipMemcpyDeviceToHost));
  int per_xcc[16] = {0};
  std::set<unsigned> cus;
  int off_mod8 = 0;
  for (int i = 0; i < nwg; ++i) {
    const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 15;
    per_xcc[xcc]++;
    if ((int)xcc != i % 8) ++off_mod8;
    cus.insert((xcc << 16) | (hw & 0xff00));   // se 15:13, sh 12, cu 11:8
  }
  printf("%s: %d workgroups; per XCD:", what, nwg);
  for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
  printf("; distinct (XCD, SE, SH, CU): %zu; workgroups not on XCD (index mod 8): %d\n", cus.size(), off_mod8);
  int per_xcd_cus[8] = {0};
  for (unsigned c : cus) per_xcd_cus[(c >> 16) & 7]++;
  printf("   CUs seen per XCD:");
  for (int x = 0; x < 8; ++x) printf(" %d", per_xcd_cus[x]);
  printf("\n");
  // streaming copy
  const size_t n = (size_t)64 << 20;   // 64 M float4 = 1 GiB
  float4 *a, *b;
  CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16));
  CK(hipMemset(a, 1, n * 16));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, s, a, b, n);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) printf("   copy of 1 GiB: %.3f ms = %.2f TB/s (read + write)\n", ms / 5, 2.0 * n * 16 / (ms / 5 * 1e-3) / 1e12);
  }
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(k_valu, dim3(8192), dim3(256), 0, s, (float*)b, 20000);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) printf("   VALU kernel (8192 x 256 threads x 80 K dependent f32 ops): %.3f ms\n", ms);
  }
  CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(d));
  return 0;
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("%s: %d CUs\n", prop.name, ncu);
  if (argc < 2) { hipStream_t s; CK(hipStreamCreate(&s)); return report(s, "unmasked"); }
  if (!strcmp(argv[1], "map") && argc >= 4) {
    std::vector<int> bits;
    for (int i = atoi(argv[2]); i < atoi(argv[3]); ++i) bits.push_back(i);
    hipStream_t s; CK(masked_stream(&s, bits, ncu));
    char what[64]; snprintf(what, sizeof what, "bits [%d, %d)", atoi(argv[2]), atoi(argv[3]));
    return report(s, what);
  }
  if (!strcmp(argv[1], "mod8") && argc >= 3) {
    std::vector<int> bits;
    for (int i = atoi(argv[2]); i < ncu; i += 8) bits.push_back(i);
    hipStream_t s; CK(masked_stream(&s, bits, ncu));
    char what[64]; snprintf(what, sizeof what, "bits = %d (mod 8)", atoi(argv[2]));
    printf("(about to launch on a stream whose mask names one XCD's CUs)\n"); fflush(stdout);
    return report(s, what);
  }
  if (!strcmp(argv[1], "pair") && argc >= 6) {
    std::vector<int> b1, b2;
    for (int i = atoi(argv[2]); i < atoi(argv[3]); ++i) b1.push_back(i);
    for (int i = atoi(argv[4]); i < atoi(argv[5]); ++i) b2.push_back(i);
    hipStream_t m1, m2, u1, u2;
    CK(masked_stream(&m1, b1, ncu)); CK(masked_stream(&m2, b2, ncu));
    CK(hipStreamCreate(&u1)); CK(hipStreamCreate(&u2));
    float* out; CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1, ej;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ej));
    auto both = [&](hipStream_t a, hipStream_t b, const char* what) -> int {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, a));
        CK(hipStreamWaitEvent(b, e0, 0));
        hipLaunchKernelGGL(k_valu, dim3(4096), dim3(256), 0, a, out, 20000);
        hipLaunchKernelGGL(k_valu, dim3(4096), dim3(256), 0, b, out, 20000);
        CK(hipEventRecord(ej, b));
        CK(hipStreamWaitEvent(a, ej, 0));
        CK(hipEventRecord(e1, a));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("  %s: %.3f ms for two launches of 4096 workgroups\n", what, ms);
      }
      return 0;
    };
    if (both(u1, u2, "two unmasked streams")) return 1;
    if (both(m1, m2, "two masked streams on disjoint CUs")) return 1;
    if (both(m1, m1, "one masked stream, both launches")) return 1;
    if (both(u1, u1, "one unmasked stream, both launches")) return 1;
    return 0;
  }
  printf("usage: cumask [map LO HI | mod8 K | pair LO1 HI1 LO2 HI2]\n");
  return 2;
}
