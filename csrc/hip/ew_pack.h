// Packed bf16 <-> fp32 helpers of the elementwise passes: two fp32 per instruction (v_pk_fma_f32 /
// v_pk_add_f32), ReLU and sign bits on the packed bf16 result (v_pk_max_i16).
#pragma once
#include "common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x2 unpack_bf16x2(unsigned w) {
  f32x2 r;
  r.x = __uint_as_float(w << 16);
  r.y = __uint_as_float(w & 0xffff0000u);
  return r;
}
__device__ __forceinline__ unsigned pack_bf16x2(f32x2 v) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
}
// max(x, 0) on two packed bf16: a negative bf16 is a negative int16
__device__ __forceinline__ unsigned relu_bf16x2(unsigned w) {
  const i16x2 zero = {0, 0};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, w), zero));
}
// (x > 0) of two packed non-negative bf16 as bits 0 and 1
__device__ __forceinline__ unsigned positive_bits_bf16x2(unsigned w) {
  return ((w & 0xffffu) != 0 ? 1u : 0u) | ((w >> 16) != 0 ? 2u : 0u);
}

