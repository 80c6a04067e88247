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
  const unsigned long long mask = 0x5555aaaa3333ccccull ^ (unsigned long long)iters;
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
// selects: vcc read without telling the compiler it is touched (no s_nop between the statements),
// the VOP3 form with the mask in an SGPR pair, and the usual compare + select pair
#define CNDV(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[i]) : "v"(m));
#define CNDS(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(n[i]) : "v"(m), "s"(mask));
#define CMPCND(i) asm volatile("v_cmp_lt_f32 vcc, %0, %2\n\tv_cndmask_b32 %1, %1, %3, vcc" : : "v"(a[i]), "v"(n[i]), "v"(b), "v"(m) : "vcc");
#define ADDNOP(i) asm volatile("v_add_f32 %0, %0, %1\n\ts_nop 0" : "+v"(a[i]) : "v"(b));
#define SUB32(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define FMAC(i) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[i]) : "v"(b));
#define MIN32(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(n[i]) : "v"(m));
#define AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(n[i]) : "v"(m));
#define BFI(i) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(n[i]) : "v"(m));
#define SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
#define DSCALE(i) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[i]) : "v"(b) : "vcc");
#define DFMAS(i) asm volatile("v_div_fmas_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
#define DFIX(i) asm volatile("v_div_fixup_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
#define LSHLADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(d[i]) : "v"(e));
#define MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[i]) : "v"(n[i]), "v"(m) : "vcc");
#define CVTI(i) asm volatile("v_cvt_f32_i32 %0, %1" : "+v"(a[i]) : "v"(n[i]));
#define RCP64(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
#define PKM(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(e));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i]) : "v"(e));
#define ADDSG(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "s"(seed));
#define MULSG(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(seed));
#define PKMS(i) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(d[i]) : "s"(mask));
#define PKAS(i) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(d[i]) : "s"(mask));
#define MULLIT(i) asm volatile("v_mul_f32 %0, 0x3e824ab0, %0" : "+v"(a[i]));
#define ADDINL(i) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(a[i]));
#define MUL64LIT(i) asm volatile("v_mul_f64 %0, %0, 0.5" : "+v"(d[i]));
#define ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(n[i]) : "v"(m));
#define ABSADD(i) asm volatile("v_add_f32_e64 %0, |%0|, %1" : "+v"(a[i]) : "v"(b));
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
      if (OP == 22) { REP8(CNDV) }
      if (OP == 23) { REP8(CNDS) }
      if (OP == 24) { REP8(CMPCND) }
      if (OP == 25) { REP8(ADDNOP) }
      if (OP == 26) { REP8(SUB32) }
      if (OP == 27) { REP8(FMAC) }
      if (OP == 28) { REP8(MIN32) }
      if (OP == 29) { REP8(XOR) }
      if (OP == 30) { REP8(AND) }
      if (OP == 31) { REP8(BFI) }
      if (OP == 32) { REP8(SQRT) }
      if (OP == 33) { REP8(DSCALE) }
      if (OP == 34) { REP8(DFMAS) }
      if (OP == 35) { REP8(DFIX) }
      if (OP == 36) { REP8(LSHLADD64) }
      if (OP == 37) { REP8(MAD64) }
      if (OP == 38) { REP8(CVTI) }
      if (OP == 39) { REP8(RCP64) }
      if (OP == 40) { REP8(PKM) }
      if (OP == 41) { REP8(PKFMA) }
      if (OP == 42) { REP8(ADDSG) }
      if (OP == 43) { REP8(MULSG) }
      if (OP == 44) { REP8(ADD3) }
      if (OP == 45) { REP8(ABSADD) }
      if (OP == 46) { REP8(PKMS) }
      if (OP == 47) { REP8(PKAS) }
      if (OP == 48) { REP8(MULLIT) }
      if (OP == 49) { REP8(ADDINL) }
      if (OP == 50) { REP8(MUL64LIT) }
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

// The same pattern with the LDS reads as single instructions (the compiler pairs neighbouring 8-byte
// reads into ds_read2_b64, which moves 128 B/clk like ds_read2_b32; ds_read_b64 / ds_read_b128 move
// 256): 16 taps per round at literal offsets, one s_waitcnt, W adds per tap.
template <int W, int MIS = 0>   // MIS: floats by which every read is off its natural alignment
__global__ __launch_bounds__(256) void k_taps_single(float* out, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) float t[40 * 72 * W];
  for (int i = threadIdx.x; i < 40 * 72 * W; i += 256) t[i] = seed + i * 1e-6f;
  __syncthreads();
  float acc[4 * W];
  for (int i = 0; i < 4 * W; ++i) acc[i] = 0.0f;
  const int lane = threadIdx.x & 63, row = threadIdx.x >> 6;
  typedef float vt __attribute__((ext_vector_type(W)));
  for (int it = 0; it < iters; ++it) {
    const unsigned addr = (unsigned)(size_t)(t + ((row * 8 + (it & 7)) * 72) * W + lane * W + MIS) & 0xffffu;
    vt v[16];
#define RD(i, off)                                                                               \
    if constexpr (W == 1) { float x; asm volatile("ds_read_b32 %0, %1 offset:" #off : "=v"(x) : "v"(addr)); v[i][0] = x; } \
    if constexpr (W == 2) asm volatile("ds_read_b64 %0, %1 offset:2*" #off : "=v"(v[i]) : "v"(addr));          \
    if constexpr (W == 4) asm volatile("ds_read_b128 %0, %1 offset:4*" #off : "=v"(v[i]) : "v"(addr));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      RD(0, 0) RD(1, 4) RD(2, 8) RD(3, 12) RD(4, 288) RD(5, 292) RD(6, 296) RD(7, 300)
      RD(8, 576) RD(9, 580) RD(10, 584) RD(11, 588) RD(12, 864) RD(13, 868) RD(14, 872) RD(15, 876)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int e = 0; e < W; ++e) {
          acc[(i & 3) * W + e] += v[i][e];
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
                         "", "", "", "", "v_mul_i32_i24", "v_mul_lo_u32", "v_cndmask vcc", "v_cndmask sgpr",
                         "v_cmp+v_cndmask", "v_add_f32+s_nop", "v_sub_f32", "v_fmac_f32", "v_min_f32", "v_xor_b32",
                         "v_and_b32", "v_bfi_b32", "v_sqrt_f32", "v_div_scale_f32", "v_div_fmas_f32",
                         "v_div_fixup_f32", "v_lshl_add_u64", "v_mad_u64_u32", "v_cvt_f32_i32", "v_rcp_f64",
                         "v_pk_mul_f32", "v_pk_fma_f32", "v_add_f32 sgpr", "v_mul_f32 sgpr", "v_add3_u32",
                         "v_add_f32 |abs|", "v_pk_mul_f32 sgpr", "v_pk_add_f32 sgpr",
                         "v_mul_f32 literal", "v_add_f32 inline 1.0", "v_mul_f64 inline 0.5"};
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
  ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10) ROW(11) ROW(12) ROW(13) ROW(14) ROW(15) ROW(20) ROW(21) ROW(22) ROW(23) ROW(24) ROW(25) ROW(26) ROW(27) ROW(28) ROW(29) ROW(30) ROW(31) ROW(32)
  ROW(33) ROW(34) ROW(35) ROW(36) ROW(37) ROW(38) ROW(39) ROW(40) ROW(41) ROW(42) ROW(43) ROW(44) ROW(45) ROW(46) ROW(47) ROW(48) ROW(49) ROW(50)
  printf("(v_cmp+v_cndmask and v_add_f32+s_nop: per PAIR of instructions)\n");
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
  TROW(1, "4 B (read2_b32)") TROW(2, "8 B (read2_b64)") TROW(4, "16 B (b128)")
#define SROW(W, label)                                                                                \
  {                                                                                                   \
    printf("%-16s", label);                                                                           \
    for (int wps : {1, 2, 4, 8}) {                                                                    \
      if (W * 40 * 72 * 4 * wps > 160 * 1024) { printf(" %8s", "-"); continue; }                      \
      const int wg = 256 * wps;                                                                       \
      double ms = time_ms([&] { hipLaunchKernelGGL((k_taps_single<W>), dim3(wg), dim3(256), 0, 0, out, iters, 0.5f); }); \
      double adds = (double)wg * 4 * iters * 64 * W;                                                  \
      printf(" %8.2f", ms * 1e-3 * 2.4e9 * 1024 / adds);                                              \
    }                                                                                                 \
    printf("\n");                                                                                     \
  }
  SROW(1, "ds_read_b32") SROW(2, "ds_read_b64") SROW(4, "ds_read_b128")
  // the same reads one float off their natural alignment: speed, and whether the bytes are the right ones
  {
    float* h = (float*)malloc(256 * 4);
    double ms_a = time_ms([&] { hipLaunchKernelGGL((k_taps_single<2, 0>), dim3(1024), dim3(256), 0, 0, out, iters, 0.5f); });
    double ms_m = time_ms([&] { hipLaunchKernelGGL((k_taps_single<2, 1>), dim3(1024), dim3(256), 0, 0, out, iters, 0.5f); });
    hipMemcpy(h, out, 256 * 4, hipMemcpyDeviceToHost);
    float got = h[5];
    // lane 5 of row 0 with MIS = 1 reads what lane 5 reads with MIS = 0 shifted by one float: compare
    // against the 4-byte reads of the same addresses (k_taps_single<1> started one float further)
    hipLaunchKernelGGL((k_taps_single<1, 0>), dim3(1), dim3(256), 0, 0, out, 1, 0.5f);
    hipDeviceSynchronize();
    printf("ds_read_b64 misaligned by 4 bytes: %.2f vs %.2f cycles per tap-add aligned (4 waves/SIMD); lane checksum %.6f\n",
           ms_m * 1e-3 * 2.4e9 * 1024 / ((double)1024 * 4 * iters * 64 * 2), ms_a * 1e-3 * 2.4e9 * 1024 / ((double)1024 * 4 * iters * 64 * 2), got);
    double ms4a = time_ms([&] { hipLaunchKernelGGL((k_taps_single<4, 0>), dim3(512), dim3(256), 0, 0, out, iters, 0.5f); });
    double ms4m = time_ms([&] { hipLaunchKernelGGL((k_taps_single<4, 1>), dim3(512), dim3(256), 0, 0, out, iters, 0.5f); });
    printf("ds_read_b128 misaligned by 4 bytes: %.2f vs %.2f (2 waves/SIMD)\n",
           ms4m * 1e-3 * 2.4e9 * 1024 / ((double)512 * 4 * iters * 64 * 4), ms4a * 1e-3 * 2.4e9 * 1024 / ((double)512 * 4 * iters * 64 * 4));
  }
  return 0;
}
