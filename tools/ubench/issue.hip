// Issue cost of the instruction classes the Compare chain is made of, per SIMD, on gfx950: one
// kernel per class, 8 independent register chains per lane, no memory traffic (classes 0..15), and
// the Malta line-sum pattern -- LDS taps at constant offsets feeding f32 adds -- with 4-, 8- and
// 16-byte LDS reads (classes 16..19).  Prints cycles per wave-instruction and SIMD at the
// nominal 2.4 GHz for 1, 2, 4 and 8 resident wavefronts per SIMD.  (Round 5: the SQ counters count
// VALU activity in units of four cycles, so they cannot tell a 2-cycle from a 4-cycle instruction;
// DESIGN.md section 6 prices the chain's VALU floor with the numbers this prints.)
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -mllvm -vectorize-slp=false issue.hip -o issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(256) void k_issue(float* out, int iters, float seed) {
  float a[8], b = seed + 1.0f;
  double d[8], e = (double)seed + 1.0;
  int n[8], m = (int)seed + 3;
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; d[i] = a[i]; n[i] = threadIdx.x + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#define F32(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define M32(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
#define MAX32(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define I32(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[i]) : "v"(m));
#define SHL(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(n[i]));
#define MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(n[i]) : "v"(m));
#define CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[i]) : "v"(m) : "vcc");
#define CMP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
#define RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define CVT(i) asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(d[i]) : "v"(a[i]));
#define CVTB(i) asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(a[i]) : "v"(d[i]));
#define A64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e));
#define M64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e));
#define FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(e));
#define PKA(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(e));
#define MUL24(i) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(n[i]) : "v"(m));
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(n[i]) : "v"(m));
      if (OP == 0) { REP8(F32) }
      if (OP == 1) { REP8(M32) }
      if (OP == 2) { REP8(FMA32) }
      if (OP == 3) { REP8(MAX32) }
      if (OP == 4) { REP8(I32) }
      if (OP == 5) { REP8(SHL) }
      if (OP == 6) { REP8(MOV) }
      if (OP == 7) { REP8(CND) }
      if (OP == 8) { REP8(CMP) }
      if (OP == 9) { REP8(RCP) }
      if (OP == 10) { REP8(CVT) }
      if (OP == 11) { REP8(CVTB) }
      if (OP == 12) { REP8(A64) }
      if (OP == 13) { REP8(M64) }
      if (OP == 14) { REP8(FMA64) }
      if (OP == 15) { REP8(PKA) }
      if (OP == 20) { REP8(MUL24) }
      if (OP == 21) { REP8(MULLO) }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + (float)d[i] + (float)n[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// The Malta pattern: 64 taps at constant offsets around a per-lane centre, summed into four
// accumulators.  W = floats per LDS read (1: dword reads as in k_malta_rolled, 2 / 4: aligned
// 8- / 16-byte reads, the lane then owns W adjacent centres and every read feeds W adds).
template <int W>
__global__ __launch_bounds__(256) void k_taps(float* out, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) float t[40 * 72 * W];
  for (int i = threadIdx.x; i < 40 * 72 * W; i += 256) t[i] = seed + i * 1e-6f;
  __syncthreads();
  float acc[4 * W];
  for (int i = 0; i < 4 * W; ++i) acc[i] = 0.0f;
  const int lane = threadIdx.x & 63, row = threadIdx.x >> 6;
  for (int it = 0; it < iters; ++it) {
    const float* c = t + ((row * 8 + (it & 7) + 4) * 72 + 4) * W + lane * W;
#pragma unroll
    for (int k = 0; k < 64; ++k) {
      const int dy = k / 9 - 3, dx = k % 9 - 4;   // 64 distinct taps of a 9 x 8 window
      const float* p = c + (dy * 72 + dx) * W;
      if (W == 1) {
        acc[k & 3] += *p;
      } else if (W == 2) {
        const float2 v = *reinterpret_cast<const float2*>(p);
        acc[(k & 3) * 2] += v.x; acc[(k & 3) * 2 + 1] += v.y;
      } else {
        const float4 v = *reinterpret_cast<const float4*>(p);
        acc[(k & 3) * 4] += v.x; acc[(k & 3) * 4 + 1] += v.y; acc[(k & 3) * 4 + 2] += v.z; acc[(k & 3) * 4 + 3] += v.w;
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 4 * W; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <class F>
static double time_ms(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < 3; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 3;
}

int main(int argc, char** argv) {
  float* out; hipMalloc(&out, 8192 * 256 * 4);
  const int iters = 400;
  const char* names[] = {"v_add_f32", "v_mul_f32", "v_fma_f32", "v_max_f32", "v_add_u32", "v_lshlrev_b32",
                         "v_mov_b32", "v_cndmask_b32", "v_cmp_lt_f32", "v_rcp_f32", "v_cvt_f64_f32",
                         "v_cvt_f32_f64", "v_add_f64", "v_mul_f64", "v_fma_f64", "v_pk_add_f32",
                         "", "", "", "", "v_mul_i32_i24", "v_mul_lo_u32"};
  printf("cycles per wave-instruction and SIMD at 2.4 GHz (256 CUs x 4 SIMDs); columns = wavefronts per SIMD\n");
  printf("%-16s %8s %8s %8s %8s\n", "instruction", "1", "2", "4", "8");
#define ROW(OP)                                                                                       \
  {                                                                                                   \
    printf("%-16s", names[OP]);                                                                       \
    for (int wps : {1, 2, 4, 8}) {                                                                    \
      const int wg = 256 * wps; /* 256-thread workgroups: 4 waves = one per SIMD; wps per CU */       \
      double ms = time_ms([&] { hipLaunchKernelGGL((k_issue<OP>), dim3(wg), dim3(256), 0, 0, out, iters, 0.5f); }); \
      double winst = (double)wg * 4 * iters * 64;                                                     \
      printf(" %8.2f", ms * 1e-3 * 2.4e9 * 1024 / winst);                                             \
    }                                                                                                 \
    printf("\n");                                                                                     \
  }
  ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10) ROW(11) ROW(12) ROW(13) ROW(14) ROW(15) ROW(20) ROW(21)
  printf("\nMalta line-sum pattern: 64 LDS taps -> f32 adds; cycles per TAP-ADD (one add of one lane group) and SIMD\n");
  printf("%-16s %8s %8s %8s %8s\n", "LDS read width", "1", "2", "4", "8");
#define TROW(W, label)                                                                                \
  {                                                                                                   \
    printf("%-16s", label);                                                                           \
    for (int wps : {1, 2, 4, 8}) {                                                                    \
      if (W * 40 * 72 * 4 * wps > 160 * 1024) { printf(" %8s", "-"); continue; }                      \
      const int wg = 256 * wps;                                                                       \
      double ms = time_ms([&] { hipLaunchKernelGGL((k_taps<W>), dim3(wg), dim3(256), 0, 0, out, iters, 0.5f); }); \
      double adds = (double)wg * 4 * iters * 64 * W;                                                  \
      printf(" %8.2f", ms * 1e-3 * 2.4e9 * 1024 / adds);                                              \
    }                                                                                                 \
    printf("\n");                                                                                     \
  }
  TROW(1, "4 B (b32)") TROW(2, "8 B (b64)") TROW(4, "16 B (b128)")
  return 0;
}
