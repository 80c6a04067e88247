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

// Load through a pointer that is known to point to device global memory (a pointer picked
// from the kernel arguments with a run-time index is otherwise treated as generic, and the
// load becomes a flat load).
#ifdef GZ_EMU
#define GZ_LDG(p, i) ((p)[i])
#else
#define GZ_LDG(p, i) (((const __attribute__((address_space(1))) float*)(p))[i])
#endif

// Four consecutive floats moved as one 16-byte access (global_load/store_dwordx4,
// ds_read/write_b128): a dword per lane reaches about two thirds of the streaming rate of
// 16 bytes per lane on gfx950 (tools/ubench/bw.hip).
struct alignas(16) gz_f4 {
  float v[4];
};
#ifdef GZ_EMU
#define GZ_LDG4(p, i) (*reinterpret_cast<const gz_f4*>((p) + (i)))
#else
typedef float gz_vec4 __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ gz_f4 gz_f4_from(gz_vec4 x) {
  gz_f4 r;
  r.v[0] = x.x; r.v[1] = x.y; r.v[2] = x.z; r.v[3] = x.w;
  return r;
}
#define GZ_LDG4(p, i)                                                          \
  gz_f4_from(*reinterpret_cast<const __attribute__((address_space(1))) gz_vec4*>( \
      (const __attribute__((address_space(1))) float*)(p) + (i)))
#endif

#ifdef GZ_EMU
#define GZ_STG4(p, i, val) (*reinterpret_cast<gz_f4*>((p) + (i)) = (val))
#else
static __device__ __forceinline__ void gz_stg4(float* p, size_t i, const gz_f4& x) {
  gz_vec4 t;
  t.x = x.v[0]; t.y = x.v[1]; t.z = x.v[2]; t.w = x.v[3];
  *reinterpret_cast<__attribute__((address_space(1))) gz_vec4*>(
      (__attribute__((address_space(1))) float*)(p) + i) = t;
}
#define GZ_STG4(p, i, val) gz_stg4((p), (i), (val))
#endif

// Two floats that one packed instruction works on (v_pk_mul_f32 / v_pk_add_f32: two IEEE
// f32 operations per lane and issue slot, each rounded like its scalar counterpart; the
// build has contraction off, so a multiply and an add stay a multiply and an add).
typedef float gz_f2 __attribute__((vector_size(8)));
GZ_DEVFN gz_f2 gz_f2_splat(float x) {
  gz_f2 r = {x, x};
  return r;
}

// Two consecutive floats as one 8-byte store (the index is even).
#ifdef GZ_EMU
#define GZ_STG2(p, i, val) (*reinterpret_cast<gz_f2*>((p) + (i)) = (val))
#else
#define GZ_STG2(p, i, val) \
  (*reinterpret_cast<__attribute__((address_space(1))) gz_f2*>((__attribute__((address_space(1))) float*)(p) + (i)) = (val))
#endif

#ifdef GZ_EMU
#define GZ_STG(p, i, val) ((p)[i] = (val))
#else
#define GZ_STG(p, i, val) (((__attribute__((address_space(1))) float*)(p))[i] = (val))
#endif

// A value that is the same in every lane of the wavefront (derived from threadIdx.x >> 6):
// telling the compiler so turns the branches on it into scalar branches.
#ifdef GZ_EMU
#define GZ_WAVE_UNIFORM(x) (x)
#else
#define GZ_WAVE_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif

// A point where the lanes of a wavefront must have executed everything before it (loads
// before a store to the same LDS row by a neighbouring lane).  On the GPU a wavefront has one
// instruction stream, so no instruction is needed -- but the COMPILER must keep every LDS access
// on its side of the point (a load it can prove not to alias an earlier store of the same thread
// may otherwise sink below it, and a neighbouring lane would read the overwritten value: ADVICE
// r3): a wavefront-scope fence + wave barrier, which emit nothing.  The emulation's threads are
// fibers that run one after the other and yield here (every thread of the workgroup must pass the
// same number of these points).
#ifdef GZ_EMU
#define GZ_WAVE_LOCKSTEP() hipemu::yield()
#else
#define GZ_WAVE_LOCKSTEP()                                  \
  do {                                                      \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                        \
  } while (0)
#endif

// Lanes of ONE wavefront exchanging data through LDS (a wavefront's LDS operations execute in
// program order): nothing to wait for, but the compiler must keep the accesses on their sides of
// this point.  (The emulation yields, as for GZ_WAVE_LOCKSTEP.)
#ifdef GZ_EMU
#define GZ_WAVE_SYNC() hipemu::yield()
#else
#define GZ_WAVE_SYNC()                                      \
  do {                                                      \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                        \
  } while (0)
#endif

// Release / acquire accesses at device scope for flags that workgroups of one launch pass to
// each other through global memory (the decoupled look-back of k_scan_offsets).
#ifdef GZ_EMU
#define GZ_STORE_RELEASE(p, v) (*(p) = (v))
#define GZ_LOAD_ACQUIRE(p) (*(p))
#define GZ_LOAD_L2(p) (*(p))
#else
#define GZ_STORE_RELEASE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT)
#define GZ_LOAD_ACQUIRE(p) __hip_atomic_load((p), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
// a load served by L2 (where other workgroups' atomics land), without an ordering of its own: loads of
// this kind issue back to back, behind one fence
#define GZ_LOAD_L2(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif

// a * b for operands that fit 24 bits signed (IDCT: 16-bit coefficients x 14-bit matrix
// entries; colour conversion: 17-bit constants x 9-bit samples): v_mul_i32_i24 issues at full
// rate, the general 32-bit v_mul_lo_u32 at a quarter of it.  Same low 32 bits of the product.
#ifdef GZ_EMU
#define GZ_MUL24(a, b) ((a) * (b))
#else
#define GZ_MUL24(a, b) __mul24((a), (b))
#endif

// A value kept in a vector register.  Kernel arguments live in scalar registers, and on gfx950 a
// v_mul_f32 / v_add_f32 with a scalar-register operand issues in 4 cycles where the all-VGPR form
// takes 2.5 (tools/ubench/issue.hip, profiles/r05_issue_cost.log): a blur's taps, multiplied into
// every sample, are worth their 2R + 1 vector registers.  No instruction beyond the one-off copy.
#ifdef GZ_EMU
#define GZ_IN_VGPR(x) (x)
#else
static __device__ __forceinline__ float gz_in_vgpr(float x) {
  asm volatile("" : "+v"(x));
  return x;
}
#define GZ_IN_VGPR(x) gz_in_vgpr(x)
#endif

// All four components of a 16-byte vector just read from LDS count as used.  A window whose first or
// last component no tap touches (a blur radius that is not a multiple of 4: R = 23 leaves one at either
// end) otherwise has that read narrowed to 8 bytes, and the compiler then pairs ALL the window's reads
// as ds_read2_b64 at offsets 8 bytes off the 16-byte grid: half the LDS rate of ds_read_b128 and a
// different bank pattern (k_blur_h_pk<23>: SQ_LDS_IDX_ACTIVE 2.8x k_blur_h_pk<16>'s, bank conflicts 0.38
// of it, profiles/r05_compare_4k_sq_counters.csv).  No instruction of its own.
#if defined(GZ_EMU) || defined(GZ_NO_KEEP_F4)   // (GZ_NO_KEEP_F4: the A/B build without it)
#define GZ_KEEP_F4(v) ((void)0)
#else
#define GZ_KEEP_F4(v) asm volatile("" : "+v"((v).v[0]), "+v"((v).v[1]), "+v"((v).v[2]), "+v"((v).v[3]))
#endif

// "Does any active lane of this wavefront need the rare path?"  A wavefront-uniform condition: the
// compiler branches around the path instead of computing it for every lane and selecting (what it
// does with a per-lane condition and a handful of instructions: malta_diff's FP64 form of absval
// cost six 4-cycle instructions per sample that way).  The emulation just evaluates the lane's own.
// GZ_RARE_PATH() at the head of the guarded block: an empty volatile asm, which the compiler may not
// execute speculatively -- without it the guarded instructions are hoisted out and selected again.
#ifdef GZ_EMU
#define GZ_ANY_LANE(c) (c)
#define GZ_RARE_PATH() ((void)0)
#else
#define GZ_ANY_LANE(c) (__builtin_amdgcn_ballot_w64(c) != 0ull)
#define GZ_RARE_PATH() asm volatile("; rare path")
#endif

// The value lane + D holds (D = -15..15, a constant), +0.0f where lane + D falls outside this lane's
// row of 16 lanes: the DPP row-shift operand modifier, which the compiler folds into the consuming
// v_mul_f32 -- a neighbour's value without LDS and without an instruction of its own.  (The emulation
// exchanges through its wavefront slots; every thread of the workgroup must get here together.)
template <int D>
GZ_DEVFN float gz_row16_shift(float v) {
  static_assert(D != 0 && D > -16 && D < 16, "gz_row16_shift: a shift inside a row of 16 lanes");
#ifdef GZ_EMU
  const int lane = (int)(threadIdx.x & 63), src = lane + D;
  const int got = __shfl(__builtin_bit_cast(int, v), src & 63);
  return src >= 0 && (src >> 4) == (lane >> 4) ? __builtin_bit_cast(float, got) : 0.0f;
#else
  // dpp_ctrl: row_shl:n = 0x100 + n (lane i takes lane i + n), row_shr:n = 0x110 + n (lane i - n);
  // bound_ctrl: lanes without a source take 0
  constexpr int ctrl = D > 0 ? 0x100 + D : 0x110 - D;
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true));
#endif
}

// Number of set bits of a wavefront mask (__ballot), and the mask of the lanes below this one.
#ifdef GZ_EMU
#define GZ_POPC64(x) __builtin_popcountll((unsigned long long)(x))
#else
#define GZ_POPC64(x) __popcll((unsigned long long)(x))
#endif
#define GZ_LANES_BELOW(lane) ((1ull << (lane)) - 1ull)

static inline int gz_div_up(int a, int b) { return (a + b - 1) / b; }

// ---- XCD-aware tile order ---------------------------------------------------------------
// MI355X has 8 XCDs with a private 4 MB L2 each, and the dispatcher is observed to place block
// b on XCD b % 8 (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement"): neighbouring tiles
// -- which share their halo rows / columns -- would land on eight different L2s and fetch the
// shared samples eight times.  gz_xcd_tile() maps the hardware block id to a tile id such that
// XCD k works through the k-th contiguous eighth of the (z, y, x) tile sequence: a band of
// consecutive tile rows per XCD.  A bijection for every grid size, so placement is a matter
// of speed only (nothing depends on which XCD a block really runs on).
struct GzTile { int x, y, z; };
GZ_DEVFN GzTile gz_xcd_tile() {
  const int gx = (int)gridDim.x, gy = (int)gridDim.y, n = gx * gy * (int)gridDim.z;
  const int b = (int)blockIdx.x + gx * ((int)blockIdx.y + gy * (int)blockIdx.z);
  const int xcd = b & 7, idx = b >> 3;
  const int q = n >> 3, r = n & 7;
  const int t = xcd * q + (xcd < r ? xcd : r) + idx;
  GzTile o;
  o.x = t % gx;
  o.y = (t / gx) % gy;
  o.z = t / (gx * gy);
  return o;
}
