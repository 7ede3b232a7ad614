// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels.
//
// Everything here assumes 64-lane wavefronts and the gfx950 MFMA / LDS-DMA
// instruction set; there is deliberately no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define DTFA_WAVE 64

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// Async 16-byte-per-lane global -> LDS copy (LDS-DMA).  The LDS destination is
// wave-uniform base + lane*16; the global source is per lane.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc, (LDS_AS void*)lds_wave_base, 16, 0, 0);
}

// The same copy issued from inline assembly.  The compiler does not see it as an LDS write, so it inserts
// no conservative "s_waitcnt vmcnt(0)" in front of LDS reads it cannot disambiguate from the copy (the
// transposed ds_read_b64_tr_b16 reads are such reads); every wait for these copies is then the caller's
// explicit counted s_waitcnt vmcnt(N) followed by a barrier.
__device__ __forceinline__ void glds16_untracked(const void* gsrc, void* lds_wave_base) {
  unsigned keep;
  const unsigned dst = (unsigned)(uintptr_t)(LDS_AS char*)lds_wave_base;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(dst)
      : "memory");
}

__device__ __forceinline__ void wait_vmcnt0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Wait until at most N vector-memory operations (LDS-DMA loads included) are outstanding.
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ float bf16_to_f32(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f32_to_bf16(float v) { return (bf16)v; }

__device__ __forceinline__ float bf16bits_to_f32(unsigned short b) {
  return __uint_as_float(((unsigned)b) << 16);
}

// Wave-wide sum (64 lanes) through DPP/shuffles.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// Bijective XCD-aware remap of a linear block id: blocks that share an XCD
// (id % 8 equal under round-robin dispatch) receive a contiguous range of
// logical tiles so that neighbouring tiles share that XCD's L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned nx = 8;
  if (nwg < nx * 2) return bid;
  unsigned q = nwg / nx, r = nwg % nx;
  unsigned xcd = bid % nx, idx = bid / nx;
  unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

#define HIP_CHECK_LAUNCH() (int)hipGetLastError()

// Geometry of a pooling window over an NHWC tensor (shared by pool.hip and bn_pool.hip).
struct PoolArgs {
  int B, IH, IW, C, OH, OW;
  int kh, kw, sy, sx, pt, pl;  // kernel, stride, pad top/left
  int count_include_pad;
};

