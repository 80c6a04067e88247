// Micro-benchmark: streaming-copy bandwidth on one float plane set for the access shapes the
// Compare kernels use (dword vs dwordx4 per lane, 2-D tile staging through LDS).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void copy1(const float* __restrict__ a, float* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) b[i] = a[i];
}
__global__ __launch_bounds__(256) void copy4(const float4* __restrict__ a, float4* __restrict__ b, size_t n4) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) b[i] = a[i];
}
// 4 float4 per thread, grid-stride-free (each WG 4096 floats)
__global__ __launch_bounds__(256) void copy4x4(const float4* __restrict__ a, float4* __restrict__ b, size_t n4) {
  size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
  float4 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) if (base + 256 * k < n4) v[k] = a[base + 256 * k];
#pragma unroll
  for (int k = 0; k < 4; ++k) if (base + 256 * k < n4) b[base + 256 * k] = v[k];
}
// 2-D tile like k_blur_h: 256 x 4 outputs, staged through LDS with dword accesses, halo R
template <int R>
__global__ __launch_bounds__(256) void tile_h1(const float* __restrict__ a, float* __restrict__ b, int w, int h) {
  __shared__ float t[4][256 + 2 * R];
  const int x0 = blockIdx.x * 256, y0 = blockIdx.y * 4, tid = threadIdx.x;
  for (int i = tid; i < 4 * (256 + 2 * R); i += 256) {
    const int ry = i / (256 + 2 * R), rx = i - ry * (256 + 2 * R);
    const int x = x0 - R + rx, y = y0 + ry;
    t[ry][rx] = (x >= 0 && x < w && y < h) ? a[(size_t)y * w + x] : 0.0f;
  }
  __syncthreads();
  const int x = x0 + tid;
  if (x >= w) return;
  for (int ry = 0; ry < 4; ++ry) if (y0 + ry < h) b[(size_t)(y0 + ry) * w + x] = t[ry][tid + R] + t[ry][tid];
}
// same tile, float4 staging (aligned superset) and float4 stores
template <int R>
__global__ __launch_bounds__(256) void tile_h4(const float* __restrict__ a, float* __restrict__ b, int w, int h) {
  constexpr int RA = (R + 3) & ~3;
  constexpr int TW = 256 + 2 * RA;
  __shared__ __attribute__((aligned(16))) float t[4][TW];
  const int x0 = blockIdx.x * 256, y0 = blockIdx.y * 4, tid = threadIdx.x;
  for (int i = tid; i < 4 * (TW / 4); i += 256) {
    const int ry = i / (TW / 4), q = i - ry * (TW / 4);
    const int x = x0 - RA + 4 * q, y = y0 + ry;
    float4 v = make_float4(0, 0, 0, 0);
    if (x >= 0 && x + 3 < w && y < h) v = *reinterpret_cast<const float4*>(a + (size_t)y * w + x);
    *reinterpret_cast<float4*>(&t[ry][4 * q]) = v;
  }
  __syncthreads();
  const int ry = tid >> 6, xq = (tid & 63) * 4;
  if (x0 + xq + 3 < w && y0 + ry < h) {
    float4 o;
    o.x = t[ry][xq + RA] + t[ry][xq + RA - R];
    o.y = t[ry][xq + RA + 1] + t[ry][xq + RA - R + 1];
    o.z = t[ry][xq + RA + 2] + t[ry][xq + RA - R + 2];
    o.w = t[ry][xq + RA + 3] + t[ry][xq + RA - R + 3];
    *reinterpret_cast<float4*>(b + (size_t)(y0 + ry) * w + x0 + xq) = o;
  }
}
// column tile like k_blur_v: 64 x 64 outputs + 2R halo rows; dword vs float4 staging and stores
template <int R, bool V4>
__global__ __launch_bounds__(256) void tile_v(const float* __restrict__ a, float* __restrict__ b, int w, int h) {
  __shared__ __attribute__((aligned(16))) float t[64 + 2 * R][64];
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 64, tid = threadIdx.x;
  if (!V4) {
    const int tx = tid & 63, tg = tid >> 6;
    for (int ry = tg; ry < 64 + 2 * R; ry += 4) {
      const int y = y0 - R + ry, x = x0 + tx;
      t[ry][tx] = (x < w && y >= 0 && y < h) ? a[(size_t)y * w + x] : 0.0f;
    }
    __syncthreads();
    for (int i = 0; i < 16; ++i) {
      const int y = y0 + tg * 16 + i, x = x0 + tx;
      if (x < w && y < h) b[(size_t)y * w + x] = t[tg * 16 + i][tx] + t[tg * 16 + i + 2 * R][tx];
    }
  } else {
    const int q = tid & 15, rg = tid >> 4;
    for (int ry = rg; ry < 64 + 2 * R; ry += 16) {
      const int y = y0 - R + ry, x = x0 + 4 * q;
      float4 v = make_float4(0, 0, 0, 0);
      if (x + 3 < w && y >= 0 && y < h) v = *reinterpret_cast<const float4*>(a + (size_t)y * w + x);
      *reinterpret_cast<float4*>(&t[ry][4 * q]) = v;
    }
    __syncthreads();
    for (int i = 0; i < 4; ++i) {
      const int ly = rg * 4 + i, y = y0 + ly, x = x0 + 4 * q;
      if (x + 3 < w && y < h) {
        const float4 u = *reinterpret_cast<const float4*>(&t[ly][4 * q]);
        const float4 d = *reinterpret_cast<const float4*>(&t[ly + 2 * R][4 * q]);
        *reinterpret_cast<float4*>(b + (size_t)y * w + x) = make_float4(u.x + d.x, u.y + d.y, u.z + d.z, u.w + d.w);
      }
    }
  }
}

int main() {
  const int w = 3840, h = 2160, reps = 30, nbuf = 12;   // 12 planes of 33 MB = 400 MB > infinity cache
  const size_t n = (size_t)w * h;
  float* buf;
  CHECK(hipMalloc((void**)&buf, n * 4 * nbuf * 2));
  CHECK(hipMemset(buf, 0, n * 4 * nbuf * 2));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch(buf, buf + n * nbuf);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) {
      const int k = r % nbuf;
      launch(buf + n * k, buf + n * (nbuf + k));
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %7.1f us/plane  %6.2f TB/s (read+write)\n", name, ms * 1000 / reps, 2.0 * n * 4 / (ms / reps * 1e-3) / 1e12);
  };
  run("copy dword", [&](float* a, float* b) { hipLaunchKernelGGL(copy1, dim3((n + 255) / 256), dim3(256), 0, 0, a, b, n); });
  run("copy dwordx4", [&](float* a, float* b) { hipLaunchKernelGGL(copy4, dim3((n / 4 + 255) / 256), dim3(256), 0, 0, (const float4*)a, (float4*)b, n / 4); });
  run("copy 4 x dwordx4", [&](float* a, float* b) { hipLaunchKernelGGL(copy4x4, dim3((n / 4 + 1023) / 1024), dim3(256), 0, 0, (const float4*)a, (float4*)b, n / 4); });
  run("tile_h dword R=16", [&](float* a, float* b) { hipLaunchKernelGGL(tile_h1<16>, dim3((w + 255) / 256, (h + 3) / 4), dim3(256), 0, 0, a, b, w, h); });
  run("tile_h dwordx4 R=16", [&](float* a, float* b) { hipLaunchKernelGGL(tile_h4<16>, dim3((w + 255) / 256, (h + 3) / 4), dim3(256), 0, 0, a, b, w, h); });
  run("tile_h dwordx4 R=23", [&](float* a, float* b) { hipLaunchKernelGGL(tile_h4<23>, dim3((w + 255) / 256, (h + 3) / 4), dim3(256), 0, 0, a, b, w, h); });
  run("tile_v dword R=16", [&](float* a, float* b) { hipLaunchKernelGGL((tile_v<16, false>), dim3((w + 63) / 64, (h + 63) / 64), dim3(256), 0, 0, a, b, w, h); });
  run("tile_v dwordx4 R=16", [&](float* a, float* b) { hipLaunchKernelGGL((tile_v<16, true>), dim3((w + 63) / 64, (h + 63) / 64), dim3(256), 0, 0, a, b, w, h); });
  run("tile_v dword R=4", [&](float* a, float* b) { hipLaunchKernelGGL((tile_v<4, false>), dim3((w + 63) / 64, (h + 63) / 64), dim3(256), 0, 0, a, b, w, h); });
  run("tile_v dwordx4 R=4", [&](float* a, float* b) { hipLaunchKernelGGL((tile_v<4, true>), dim3((w + 63) / 64, (h + 63) / 64), dim3(256), 0, 0, a, b, w, h); });
  return 0;
}
