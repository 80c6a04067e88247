// Common definitions for the gfx950 kernels of the Guetzli hot path.
//
// The product library is built ONLY by hipcc for gfx950 (see guetzli_amd/build.py).
// GZ_EMU is defined solely by the test-suite's CPU emulation build (tests/emu/), which
// compiles these same sources with g++ to check indexing and accumulation order against
// the oracle without a GPU; it is never shipped and nothing here falls back to it.
#pragma once

#ifdef GZ_EMU
#include "hip_emu.h"
#define GZ_LAUNCH(kern, grid, block, stream, ...) \
  hipemu::launch((grid), (block), [=]() { kern(__VA_ARGS__); })
#else
#include <hip/hip_runtime.h>
#define GZ_LAUNCH(kern, grid, block, stream, ...) \
  hipLaunchKernelGGL(kern, (grid), (block), 0, (stream), __VA_ARGS__)
#endif

#include <stdint.h>

#define GZ_DEVFN __device__ __forceinline__

// A value that is the same in every lane of the wavefront (derived from threadIdx.x >> 6):
// telling the compiler so turns the branches on it into scalar branches.
#ifdef GZ_EMU
#define GZ_WAVE_UNIFORM(x) (x)
#else
#define GZ_WAVE_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif

static inline int gz_div_up(int a, int b) { return (a + b - 1) / b; }
