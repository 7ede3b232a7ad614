// y[M][N] = act(x[M][K] . w[N][K]^T + bias) for a handful of rows (dense layers at small batch, serving).
//
// With M <= 128 the implicit-GEMM kernel has N / 64 workgroups walking the whole reduction one k-tile after the
// other: 2048 -> 1024 at batch 32 is 16 workgroups x 32 dependent k-steps, ~29 us for 4 MB of weights.  Here the
// reduction is what is spread out: a workgroup owns a 32 x 32 output tile, its 16 waves each take every 16th
// 64-channel piece of K (one 128-byte line per operand row), feed v_mfma_f32_32x32x16_bf16 straight from global
// loads -- a lane's fragment is 16 contiguous bytes of one row, no LDS staging needed -- and the 16 partial tiles
// are summed through LDS.  N / 32 workgroups x 16 waves keep enough loads in flight to stream the weights once.
#include "common.h"

struct SkinnyArgs {
  const bf16* x;      // [M][K]
  const bf16* w;      // [N][K]
  const float* bias;  // [N] or null
  bf16* y;            // [M][N]
  int M, N, K;        // K % 16 == 0, N % 8 == 0
  int relu;
};

#define SKINNY_WAVES 16

__global__ __launch_bounds__(64 * SKINNY_WAVES) void dense_skinny_kernel(SkinnyArgs p) {
  __shared__ float red[SKINNY_WAVES][1024];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int row = m0 + (lane & 31), col = n0 + (lane & 31), kh = (lane >> 5) * 8;
  const bool rok = row < p.M, cok = col < p.N;
  const bf16* xp = p.x + (size_t)(rok ? row : 0) * p.K + kh;
  const bf16* wp = p.w + (size_t)(cok ? col : 0) * p.K + kh;
  const bf16x8 zero = {};
  f32x16 acc = {};
  const int steps = p.K / 16;
  // pieces of four k-steps (64 channels = one line per row), dealt round-robin to the waves
  for (int s0 = wave * 4; s0 < steps; s0 += SKINNY_WAVES * 4) {
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool live = s0 + i < steps;
      a[i] = (live && rok) ? *(const bf16x8*)(xp + (s0 + i) * 16) : zero;
      b[i] = (live && cok) ? *(const bf16x8*)(wp + (s0 + i) * 16) : zero;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[i], acc, 0, 0, 0);
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) red[wave][j * 64 + lane] = acc[j];
  __syncthreads();
  // 1024 outputs, one per thread: element e = j * 64 + l is row (j&3) + 8 (j>>2) + 4 (l>>5), column l & 31
  const int e = threadIdx.x;
  float v = 0.f;
#pragma unroll
  for (int w = 0; w < SKINNY_WAVES; ++w) v += red[w][e];
  const int j = e >> 6, l = e & 63;
  const int r = m0 + (j & 3) + 8 * (j >> 2) + 4 * (l >> 5), c = n0 + (l & 31);
  if (r < p.M && c < p.N) {
    if (p.bias != nullptr) v += p.bias[c];
    if (p.relu) v = fmaxf(v, 0.f);
    p.y[(size_t)r * p.N + c] = (bf16)v;
  }
}

extern "C" int dtfa_dense_skinny(const SkinnyArgs* a, hipStream_t st) {
  if (a->M <= 0 || a->N <= 0) return 0;
  if (a->K % 16 != 0 || a->K <= 0 || a->N % 8 != 0) return -1;
  hipLaunchKernelGGL(dense_skinny_kernel, dim3((a->N + 31) / 32, (a->M + 31) / 32), dim3(64 * SKINNY_WAVES), 0, st, *a);
  return (int)hipGetLastError();
}
