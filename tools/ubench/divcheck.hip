// Device check of gz::div2_shared (gz_math.h): for the twelve numerators of the six Malta
// normalisations and EVERY float denominator in [2^-40, 2^40] (80 x 2^23 values), the two
// quotients from one v_rcp_f32 equal the compiler's IEEE divisions bit for bit.  Also times
// malta_diff against malta_diff_plain on band-like data.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../guetzli_amd/csrc divcheck.hip -o divcheck
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include "gz_common.h"
#include "gz_math.h"
using gz::MaltaNorm;

static MaltaNorm norm_of(bool lf, double w_0gt1, double w_0lt1, double norm1) {
  const double len = 3.75;
  const double mulli = lf ? 0.405371989604 : 0.354191303559;
  const float kWeight0 = 0.5;
  const float kWeight1 = 0.33;
  const double w_pre0gt1 = mulli * sqrt(kWeight0 * w_0gt1) / (len * 2 + 1);
  const double w_pre0lt1 = mulli * sqrt(kWeight1 * w_0lt1) / (len * 2 + 1);
  MaltaNorm n;
  n.norm2_0gt1 = w_pre0gt1 * norm1;
  n.norm2_0lt1 = w_pre0lt1 * norm1;
  n.norm1f = static_cast<float>(norm1);
  n.fast_div = 1;
  return n;
}
struct Norms { MaltaNorm n[6]; };

__global__ void k_check(Norms nn, unsigned long long* bad) {
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;   // 80 * 2^23 denominators
  const unsigned e = idx >> 23, m = idx & 0x7fffffu;
  const float d = __uint_as_float(((e + 127u - 40u) << 23) | m);
  unsigned b = 0;
  for (int k = 0; k < 6; ++k) {
    float q0, q1;
    gz::div2_shared(nn.n[k].norm2_0gt1, nn.n[k].norm2_0lt1, d, &q0, &q1);
    b += __float_as_uint(q0) != __float_as_uint(nn.n[k].norm2_0gt1 / d);
    b += __float_as_uint(q1) != __float_as_uint(nn.n[k].norm2_0lt1 / d);
  }
  if (b) atomicAdd(bad, (unsigned long long)b);
}
template <bool FAST>
__global__ void k_time(const float* a, const float* b, float* o, MaltaNorm nm, int n, unsigned long long* bad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float r = FAST ? gz::malta_diff(a[i], b[i], nm) : gz::malta_diff_plain(a[i], b[i], nm);
  if (FAST && __float_as_uint(r) != __float_as_uint(gz::malta_diff_plain(a[i], b[i], nm))) atomicAdd(bad, 1ull);
  o[i] = r;
}
template <bool FAST>
__global__ void k_time_only(const float* a, const float* b, float* o, MaltaNorm nm, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  o[i] = FAST ? gz::malta_diff(a[i], b[i], nm) : gz::malta_diff_plain(a[i], b[i], nm);
}
template <bool FAST, int REP>
__global__ void k_valu(const float* a, const float* b, float* o, Norms nn, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float acc = 0.0f;
  const float x = a[i], y = b[i];
#pragma unroll 1
  for (int r = 0; r < REP; ++r)
#pragma unroll
    for (int k = 0; k < 6; ++k)
      acc += FAST ? gz::malta_diff(x + (float)r, y, nn.n[k]) : gz::malta_diff_plain(x + (float)r, y, nn.n[k]);
  o[i] = acc;
}
int main() {
  const float asym = 0.8f, sq = sqrtf(asym);
  Norms nn;
  nn.n[0] = norm_of(false, 5.1409625726 * asym, 5.1409625726 / asym, 58.5001247061);
  nn.n[1] = norm_of(true, 153.671655716 * sq, 153.671655716 / sq, 83150785.9592);
  nn.n[2] = norm_of(true, 6841.81248144, 6841.81248144, 0.0135134962487);
  nn.n[3] = norm_of(false, 4.91743441556 * asym, 4.91743441556 / asym, 687196.39002);
  nn.n[4] = norm_of(true, 668.358918152 * sq, 668.358918152 / sq, 0.882954368025);
  nn.n[5] = norm_of(true, 813.901703816, 813.901703816, 16792.9322251);
  unsigned long long* bad; hipMalloc(&bad, 16); hipMemset(bad, 0, 16);
  hipLaunchKernelGGL(k_check, dim3(80u * (1u << 23) / 256u), dim3(256), 0, 0, nn, bad);
  unsigned long long hb[2] = {0, 0};
  hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost);
  printf("div2_shared on the device: %llu quotients, %llu mismatches\n", 12ull * 80 * (1u << 23), hb[0]);
  const int n = 1 << 24;
  float *a, *b, *o; hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&o, n * 4);
  float* h = (float*)malloc(n * 4);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; };
  for (int i = 0; i < n; ++i) h[i] = (rnd() - 0.5f) * 20.0f;
  hipMemcpy(a, h, n * 4, hipMemcpyHostToDevice);
  for (int i = 0; i < n; ++i) h[i] = h[i] * (0.3f + 1.2f * rnd()) * (rnd() < 0.05f ? -1.0f : 1.0f);
  hipMemcpy(b, h, n * 4, hipMemcpyHostToDevice);
  for (int k = 0; k < 6; ++k) hipLaunchKernelGGL(k_time<true>, dim3(n / 256), dim3(256), 0, 0, a, b, o, nn.n[k], n, bad + 1);
  hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost);
  printf("malta_diff vs malta_diff_plain on the device: %d x 6 pairs, %llu mismatches\n", n, hb[1]);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int fast = 0; fast < 2; ++fast) {
    hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) {
      if (fast) hipLaunchKernelGGL(k_time_only<true>, dim3(n / 256), dim3(256), 0, 0, a, b, o, nn.n[r % 6], n);
      else hipLaunchKernelGGL(k_time_only<false>, dim3(n / 256), dim3(256), 0, 0, a, b, o, nn.n[r % 6], n);
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.1f us per 16.8 M values (3 plane passes = 201 MB)\n", fast ? "malta_diff      " : "malta_diff_plain", ms / 20 * 1000);
  }
  for (int fast = 0; fast < 2; ++fast) {
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) {
      if (fast) hipLaunchKernelGGL((k_valu<true, 8>), dim3(n / 256), dim3(256), 0, 0, a, b, o, nn, n);
      else hipLaunchKernelGGL((k_valu<false, 8>), dim3(n / 256), dim3(256), 0, 0, a, b, o, nn, n);
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s, 48 evaluations per loaded pair: %.1f us per 16.8 M pairs = %.3f ns per evaluation chip-wide\n",
           fast ? "malta_diff      " : "malta_diff_plain", ms / 5 * 1000, ms / 5 * 1e6 / (48.0 * n));
  }
  return hb[0] || hb[1] ? 1 : 0;
}
