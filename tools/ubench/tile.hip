// Why are the LDS-tiled kernels 1.3-1.9x slower on some boxes while a linear copy is not?
// Hypothesis: the page-table fragments behind hipMalloc are small there, and a 64x72-row tile
// touches 72 rows that are a row pitch (7.7 / 15 KB) apart -- one translation per row.
// This tool times (a) a linear plane copy and (b) a column-pass-shaped tile read (64 columns
// x (32 + 2*20) rows per workgroup) on planes from hipMalloc and on planes mapped through the
// virtual-memory API (hipMemCreate + hipMemMap) with the recommended granularity.
//   hipcc --offload-arch=gfx950 -O3 -o tile tile.hip ; ./tile
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int R = 20, TW = 64, TH = 32;

__global__ __launch_bounds__(256) void k_tile(const float* __restrict__ in, float* __restrict__ out,
                                              int w, int h, int pitch) {
  __shared__ float tile[TH + 2 * R][TW];
  const int tx = threadIdx.x & 63, tg = threadIdx.x >> 6;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
  for (int r = tg; r < TH + 2 * R; r += 4) {
    int y = y0 - R + r;
    y = y < 0 ? 0 : (y >= h ? h - 1 : y);
    const int x = x0 + tx < w ? x0 + tx : w - 1;
    tile[r][tx] = in[(size_t)y * pitch + x];
  }
  __syncthreads();
  for (int i = 0; i < TH / 4; ++i) {
    const int ly = tg * (TH / 4) + i;
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k <= 2 * R; k += 4) s += tile[ly + k][tx];
    const int x = x0 + tx, y = y0 + ly;
    if (x < w && y < h) out[(size_t)y * pitch + x] = s;
  }
}

__global__ void k_copy(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = in[i];
}

// Pure-VALU and pure-LDS kernels: no global traffic to speak of, so their time follows the
// shader clock (and the LDS pipeline) only.
__global__ __launch_bounds__(256) void k_valu(float* out, int iters) {
  float a = threadIdx.x * 0.001f, b = 1.0001f, c = 0.5f, d = 0.25f;
  for (int i = 0; i < iters; ++i) {
    a = a * b + c; b = b * 0.99999f + d; c = c * a + 0.1f; d = d * 1.00001f + a;
  }
  if (a + b + c + d == 12345.678f) out[0] = a;
}
__global__ __launch_bounds__(256) void k_lds(float* out, int iters) {
  __shared__ float s[64 * 72];
  for (int i = threadIdx.x; i < 64 * 72; i += 256) s[i] = (float)i;
  __syncthreads();
  float acc = 0.0f;
  const int tx = threadIdx.x & 63, tg = threadIdx.x >> 6;
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int k = 0; k < 40; ++k) acc += s[((tg * 8 + k + (i & 7)) % 72) * 64 + tx];
  if (acc == 12345.678f) out[0] = acc;
}
// The same 8192 dependent multiply-adds once as straight-line code (~64 KB of instructions,
// every wavefront streams through them once -- like the fully unrolled blur kernels) and once
// as a loop (a few hundred bytes): if only the former is slow on a box, instruction fetch is.
#define GZ_R8(x) x x x x x x x x
#define GZ_R64(x) GZ_R8(GZ_R8(x))
#define GZ_R512(x) GZ_R8(GZ_R64(x))
__global__ __launch_bounds__(256) void k_straight(float* out, float b, float c) {
  float a0 = threadIdx.x * 0.001f, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f;
  GZ_R512(a0 = a0 * b + c; a1 = a1 * b + c; a2 = a2 * b + c; a3 = a3 * b + c;)
  GZ_R512(a0 = a0 * c + b; a1 = a1 * c + b; a2 = a2 * c + b; a3 = a3 * c + b;)
  GZ_R512(a0 = a0 * b + c; a1 = a1 * b + c; a2 = a2 * b + c; a3 = a3 * b + c;)
  GZ_R512(a0 = a0 * c + b; a1 = a1 * c + b; a2 = a2 * c + b; a3 = a3 * c + b;)
  if (a0 + a1 + a2 + a3 == 12345.678f) out[0] = a0;
}
__global__ __launch_bounds__(256) void k_looped(float* out, float b, float c) {
  float a0 = threadIdx.x * 0.001f, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f;
#pragma unroll 1
  for (int i = 0; i < 1024; ++i) {
    a0 = a0 * b + c; a1 = a1 * b + c; a2 = a2 * b + c; a3 = a3 * b + c;
    a0 = a0 * c + b; a1 = a1 * c + b; a2 = a2 * c + b; a3 = a3 * c + b;
  }
  if (a0 + a1 + a2 + a3 == 12345.678f) out[0] = a0;
}

// 512 bytes of per-launch constants read by every wavefront: from the kernel-argument segment
// (where the blur taps and border scales of the product's kernels live) or from device memory.
struct BigArgs { float v[128]; };
__global__ __launch_bounds__(256) void k_kernarg(float* out, BigArgs a) {
  float s = threadIdx.x * 0.5f;
#pragma unroll
  for (int i = 0; i < 128; ++i) s = s * 0.999f + a.v[i];
  if (s == 12345.678f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_devarg(float* out, const float* __restrict__ a) {
  float s = threadIdx.x * 0.5f;
#pragma unroll
  for (int i = 0; i < 128; ++i) s = s * 0.999f + a[i];
  if (s == 12345.678f) out[0] = s;
}

static int run_compute(float* scratch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  BigArgs big;
  for (int i = 0; i < 128; ++i) big.v[i] = 0.01f * i;
  float* dev_args = nullptr;
  CK(hipMalloc((void**)&dev_args, sizeof(big)));
  CK(hipMemcpy(dev_args, &big, sizeof(big), hipMemcpyHostToDevice));
  for (int kind = 0; kind < 6; ++kind)
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 20; ++i) {
        if (kind == 0) hipLaunchKernelGGL(k_valu, dim3(4096), dim3(256), 0, 0, scratch, 2000);
        else if (kind == 1) hipLaunchKernelGGL(k_lds, dim3(4096), dim3(256), 0, 0, scratch, 100);
        else if (kind == 2) hipLaunchKernelGGL(k_straight, dim3(1024), dim3(256), 0, 0, scratch, 1.0001f, 0.5f);
        else if (kind == 3) hipLaunchKernelGGL(k_looped, dim3(1024), dim3(256), 0, 0, scratch, 1.0001f, 0.5f);
        else if (kind == 4) hipLaunchKernelGGL(k_kernarg, dim3(1024), dim3(256), 0, 0, scratch, big);
        else hipLaunchKernelGGL(k_devarg, dim3(1024), dim3(256), 0, 0, scratch, (const float*)dev_args);
      }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      static const char* names[] = {"valu-only", "lds-only", "straight-line 8192 fma", "looped 8192 fma",
                                    "512 B from kernarg", "512 B from device mem"};
      if (rep) printf("%-24s %7.1f us per launch\n", names[kind], ms / 20 * 1e3);
    }
  return 0;
}

static int run(const char* what, float* base, int w, int h, int nplanes, int pitch) {
  const size_t plane = (size_t)pitch * h;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 64;
  for (int kind = 0; kind < 2; ++kind) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < iters; ++i) {
        const float* in = base + (size_t)((2 * i) % nplanes) * plane;
        float* out = base + (size_t)((2 * i + 1) % nplanes) * plane;
        if (kind == 0)
          hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, (const float4*)in, (float4*)out, plane / 4);
        else
          hipLaunchKernelGGL(k_tile, dim3((w + TW - 1) / TW, (h + TH - 1) / TH), dim3(256), 0, 0, in, out, w, h, pitch);
      }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("%-22s %dx%d pitch %d %-10s %7.1f us per launch\n", what, w, h, pitch, kind ? "tile-read" : "copy", ms / iters * 1e3);
    }
  }
  return 0;
}

int main() {
  const int nplanes = 16;
  {
    float* scratch = nullptr;
    CK(hipMalloc((void**)&scratch, 4096));
    if (run_compute(scratch)) return 1;
    CK(hipFree(scratch));
  }
  for (int size = 0; size < 2; ++size) {
    const int w = size ? 3840 : 1920, h = size ? 2160 : 1080;
    const size_t bytes = (size_t)(w + 64) * h * sizeof(float) * nplanes;
    float* a = nullptr;
    CK(hipMalloc((void**)&a, bytes));
    CK(hipMemset(a, 0, bytes));
    for (int pad : {0, 16, 64})
      if (run("hipMalloc", a, w, h, nplanes, w + pad)) return 1;
    CK(hipFree(a));

    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    if (!size) printf("VMM granularity: minimum %zu, recommended %zu\n", gmin, grec);
    const size_t g = grec > (2u << 20) ? grec : (2u << 20);
    const size_t padded = (bytes + g - 1) / g * g;
    hipMemGenericAllocationHandle_t handle;
    CK(hipMemCreate(&handle, padded, &prop, 0));
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, padded, g, nullptr, 0));
    CK(hipMemMap(va, padded, 0, handle, 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, padded, &acc, 1));
    CK(hipMemset(va, 0, bytes));
    if (run("hipMemCreate+hipMemMap", (float*)va, w, h, nplanes, w)) return 1;
    CK(hipMemUnmap(va, padded));
    CK(hipMemRelease(handle));
    CK(hipMemAddressFree(va, padded));
  }
  return 0;
}
