// Batch-norm (+ReLU, +residual) kernels for NHWC bf16 tensors viewed as [M][C].
//
// Forward statistics normally come for free from the convolution epilogue
// (partial per-channel sum / sum-of-squares rows); bn_stats_partials produces
// the same partial rows for a tensor that did not come out of that kernel.
// partial_reduce + bn_finalize turn partial rows into mean / invstd /
// scale / shift (and update the running statistics), bn_apply is the
// single-pass normalise(+residual)(+ReLU) kernel, and the backward is the
// classic two-pass form: bn_bwd_partials (per-channel sum(dy), sum(dy*xhat)),
// bn_bwd_finalize, bn_bwd_apply.
//
// All elementwise traffic is 16 bytes per lane (8 bf16 channels); C % 8 == 0.
#include "common.h"

// vectors per thread and sweep / grid cap of the elementwise kernels (overridable for tuning builds)
#ifndef BN_APPLY_U
#define BN_APPLY_U 2
#endif
#ifndef BN_BWD_U
#define BN_BWD_U 2
#endif
#ifndef BN_MAX_BLOCKS
#define BN_MAX_BLOCKS (1u << 20)  // in effect one sweep per thread: the hardware dispatcher balances better than a stride loop
#endif

// ----------------------------------------------------------------------------
// [R][NV][C] partial rows -> [RS][NV][C]; each output row sums a contiguous
// slice of input rows in a fixed order (deterministic).
// ----------------------------------------------------------------------------
__global__ void partial_reduce_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int NVC,
                                      int rows_per_split) {
  int col = blockIdx.x * blockDim.x + threadIdx.x;  // over NV*C
  int rs = blockIdx.y;
  if (col >= NVC) return;
  int r0 = rs * rows_per_split, r1 = r0 + rows_per_split;
  if (r1 > R) r1 = R;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int r = r0;
  for (; r + 4 <= r1; r += 4) {
    s0 += in[(size_t)r * NVC + col];
    s1 += in[(size_t)(r + 1) * NVC + col];
    s2 += in[(size_t)(r + 2) * NVC + col];
    s3 += in[(size_t)(r + 3) * NVC + col];
  }
  for (; r < r1; ++r) s0 += in[(size_t)r * NVC + col];
  out[(size_t)rs * NVC + col] = (s0 + s1) + (s2 + s3);
}

extern "C" int dtfa_partial_reduce(const float* in, float* out, int R, int NV, int C, int RS, hipStream_t st) {
  int NVC = NV * C;
  int rps = (R + RS - 1) / RS;
  dim3 grid((NVC + 255) / 256, RS);
  hipLaunchKernelGGL(partial_reduce_kernel, grid, dim3(256), 0, st, in, out, R, NVC, rps);
  return (int)hipGetLastError();
}

// ----------------------------------------------------------------------------
// Finalize forward statistics.  part: [R][2][C] (sum, sumsq).  Writes
// mean/invstd (saved for backward) and the fused scale/shift:
//   scale = gamma * invstd, shift = beta - mean * scale   (gamma/beta optional)
// and updates running_mean / running_var in place:
//   running = running * (1 - momentum) + batch * momentum  (unbiased variance)
// ----------------------------------------------------------------------------
// Sum the two value planes of part[R][2][C] for channel c with RL row lanes (block = 32 channels x RL
// lanes); lanes are combined through LDS in a fixed order, so the result depends only on (R, RL).
// Returns true for the one thread per channel that holds the totals.
template <int RL>
__device__ inline bool reduce_part_rows(const float* __restrict__ part, int R, int C, int c, double& s, double& ss) {
  __shared__ double red[2][RL][32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
  if (c < C) {
    // the kernel is a chain of dependent-latency loads: keep eight in flight per thread
    int r = ty;
    for (; r + 3 * RL < R; r += 4 * RL) {
      float p[4], q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        p[u] = part[((size_t)(r + u * RL) * 2) * C + c];
        q[u] = part[((size_t)(r + u * RL) * 2 + 1) * C + c];
      }
      a0 += (double)p[0];
      b0 += (double)q[0];
      a1 += (double)p[1];
      b1 += (double)q[1];
      a0 += (double)p[2];
      b0 += (double)q[2];
      a1 += (double)p[3];
      b1 += (double)q[3];
    }
    for (; r < R; r += RL) {
      a0 += (double)part[((size_t)r * 2) * C + c];
      b0 += (double)part[((size_t)r * 2 + 1) * C + c];
    }
  }
  red[0][ty][tx] = a0 + a1;
  red[1][ty][tx] = b0 + b1;
  __syncthreads();
  if (ty != 0 || c >= C) return false;
  s = 0.0;
  ss = 0.0;
  for (int t = 0; t < RL; ++t) {
    s += red[0][t][tx];
    ss += red[1][t][tx];
  }
  return true;
}

// Row-lane count of the finalize kernels: 8 lanes up to 64 partial rows, 32 lanes (1024 threads) above.
static inline int finalize_lanes(int R) { return R <= 64 ? 8 : 32; }

// mean / invstd / scale / shift of one channel from its sums, running statistics updated in place
__device__ inline void finalize_channel(int c, double s, double ss, float count, float eps,
                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                        float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale,
                                        float* __restrict__ shift, float* __restrict__ run_mean,
                                        float* __restrict__ run_var, float momentum) {
  double mu = s / count;
  double var = ss / count - mu * mu;
  if (var < 0.0) var = 0.0;
  float is = (float)(1.0 / sqrt(var + (double)eps));
  float g = gamma ? gamma[c] : 1.f;
  float b = beta ? beta[c] : 0.f;
  float sc = g * is;
  mean[c] = (float)mu;
  invstd[c] = is;
  scale[c] = sc;
  shift[c] = b - (float)mu * sc;
  if (run_mean) {
    double unb = count > 1.f ? var * (double)count / ((double)count - 1.0) : var;
    run_mean[c] = run_mean[c] * (1.f - momentum) + (float)mu * momentum;
    run_var[c] = run_var[c] * (1.f - momentum) + (float)unb * momentum;
  }
}

template <int RL>
__global__ __launch_bounds__(32 * RL) void bn_finalize_kernel(
    const float* __restrict__ part, int R, int C, float count, float eps, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale,
    float* __restrict__ shift, float* __restrict__ run_mean, float* __restrict__ run_var, float momentum) {
  const int c = blockIdx.x * 32 + (threadIdx.x & 31);
  double s, ss;
  if (!reduce_part_rows<RL>(part, R, C, c, s, ss)) return;
  finalize_channel(c, s, ss, count, eps, gamma, beta, mean, invstd, scale, shift, run_mean, run_var, momentum);
}

// Many partial rows (a layer of a few hundred row tiles sits between two convolutions of a dependent chain, and this
// kernel is nothing but the latency of its loads): 16 channels x 256 row lanes per block, each thread four channels
// per load and eight loads in flight, so 2048 rows are two trips to memory instead of sixteen.  The row lanes of a
// wave are combined by shuffles, the 16 waves through LDS, both in a fixed order.  Needs C % 4 == 0.
#ifndef BN_FINALIZE_WAVES
#define BN_FINALIZE_WAVES 4
#endif
template <int NW>
__global__ __launch_bounds__(64 * NW) void bn_finalize_wide_kernel(
    const float* __restrict__ part, int R, int C, float count, float eps, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale,
    float* __restrict__ shift, float* __restrict__ run_mean, float* __restrict__ run_var, float momentum) {
  __shared__ double red[2][NW][16];   // [plane][wave][channel of the block]
  constexpr int RL = 16 * NW;
  const int q = threadIdx.x & 3, ty = threadIdx.x >> 2;
  const int c4 = blockIdx.x * 16 + q * 4;
  double s[4] = {0.0, 0.0, 0.0, 0.0}, ss[4] = {0.0, 0.0, 0.0, 0.0};
  if (c4 < C) {
    int r = ty;
    for (; r + 3 * RL < R; r += 4 * RL) {
      f32x4 pv[4], qv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        pv[u] = *(const f32x4*)(part + ((size_t)(r + u * RL) * 2) * C + c4);
        qv[u] = *(const f32x4*)(part + ((size_t)(r + u * RL) * 2 + 1) * C + c4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          s[i] += (double)pv[u][i];
          ss[i] += (double)qv[u][i];
        }
    }
    for (; r < R; r += RL) {
      const f32x4 pv = *(const f32x4*)(part + ((size_t)r * 2) * C + c4);
      const f32x4 qv = *(const f32x4*)(part + ((size_t)r * 2 + 1) * C + c4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s[i] += (double)pv[i];
        ss[i] += (double)qv[i];
      }
    }
  }
  // lanes of a wave with the same channel quad differ in bits 2..5 of the lane number
#pragma unroll
  for (int off = 4; off < 64; off <<= 1)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s[i] += __shfl_xor(s[i], off);
      ss[i] += __shfl_xor(ss[i], off);
    }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane < 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      red[0][wave][lane * 4 + i] = s[i];
      red[1][wave][lane * 4 + i] = ss[i];
    }
  }
  __syncthreads();
  const int c = blockIdx.x * 16 + threadIdx.x;
  if (threadIdx.x >= 16 || c >= C) return;
  double ts = 0.0, tss = 0.0;
  for (int w = 0; w < NW; ++w) {
    ts += red[0][w][threadIdx.x];
    tss += red[1][w][threadIdx.x];
  }
  finalize_channel(c, ts, tss, count, eps, gamma, beta, mean, invstd, scale, shift, run_mean, run_var, momentum);
}

extern "C" int dtfa_bn_finalize(const float* part, int R, int C, float count, float eps, const float* gamma,
                                const float* beta, float* mean, float* invstd, float* scale, float* shift,
                                float* run_mean, float* run_var, float momentum, hipStream_t st) {
  if (finalize_lanes(R) == 8)
    hipLaunchKernelGGL(bn_finalize_kernel<8>, dim3((C + 31) / 32), dim3(256), 0, st, part, R, C, count, eps, gamma,
                       beta, mean, invstd, scale, shift, run_mean, run_var, momentum);
  else if (C % 4 == 0 && ((size_t)part & 15) == 0)
    hipLaunchKernelGGL(bn_finalize_wide_kernel<BN_FINALIZE_WAVES>, dim3((C + 15) / 16), dim3(64 * BN_FINALIZE_WAVES), 0, st, part, R, C, count, eps,
                       gamma, beta, mean, invstd, scale, shift, run_mean, run_var, momentum);
  else
    hipLaunchKernelGGL(bn_finalize_kernel<32>, dim3((C + 31) / 32), dim3(1024), 0, st, part, R, C, count, eps,
                       gamma, beta, mean, invstd, scale, shift, run_mean, run_var, momentum);
  return (int)hipGetLastError();
}

// Inference-mode scale/shift from running statistics.
__global__ void bn_eval_coeffs_kernel(int C, float eps, const float* __restrict__ gamma,
                                      const float* __restrict__ beta, const float* __restrict__ run_mean,
                                      const float* __restrict__ run_var, float* __restrict__ scale,
                                      float* __restrict__ shift) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float is = rsqrtf(run_var[c] + eps);
  float sc = (gamma ? gamma[c] : 1.f) * is;
  scale[c] = sc;
  shift[c] = (beta ? beta[c] : 0.f) - run_mean[c] * sc;
}

extern "C" int dtfa_bn_eval_coeffs(int C, float eps, const float* gamma, const float* beta, const float* rm,
                                   const float* rv, float* scale, float* shift, hipStream_t st) {
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 127) / 128), dim3(128), 0, st, C, eps, gamma, beta, rm, rv,
                     scale, shift);
  return (int)hipGetLastError();
}

// ----------------------------------------------------------------------------
// Elementwise passes.  At ~5 TB/s a CU moves one 16-byte vector every other clock, which leaves ~100 VALU
// lane-operations per vector: written element by element (bf16 unpack, fma, max, convert, 64-bit index
// division) these kernels were bound by the vector ALU, not by HBM.  Hence: two fp32 per instruction
// (v_pk_fma_f32 / v_pk_add_f32), ReLU on the packed bf16 result (v_pk_max_i16), and a vector index that is
// divided once per thread and then advanced by additions.
// ----------------------------------------------------------------------------
#include "ew_pack.h"

#ifndef BN_NT
#define BN_NT 3
#endif
// Streaming accesses of the elementwise passes (BN_NT: bit 0 = non-temporal loads, bit 1 = non-temporal
// stores).  These tensors are far larger than the caches at training batch sizes; marking them non-temporal
// took bn_apply from 5.2 to 5.9 TB/s and the ResNet-50 step from 85.5 to 84.0 ms.
__device__ __forceinline__ u32x4 stream_load(const void* p) {
#if (BN_NT & 1)
  return __builtin_nontemporal_load((const u32x4*)p);
#else
  return *(const u32x4*)p;
#endif
}
__device__ __forceinline__ void stream_store(void* p, u32x4 v) {
#if (BN_NT & 2)
  __builtin_nontemporal_store(v, (u32x4*)p);
#else
  *(u32x4*)p = v;
#endif
}

// Position of a thread's vectors in the [rows][C / 8] view, advanced without divisions.
struct VecCursor {
  unsigned i, row, cv;          // vector index, its row and its column (in vectors)
  unsigned cvec, step, step_r, step_c;
  __device__ __forceinline__ VecCursor() {}
  __device__ __forceinline__ VecCursor(unsigned first, unsigned stride, unsigned cvec_) {
    cvec = cvec_;
    i = first;
    row = first / cvec;
    cv = first - row * cvec;
    step = stride;
    step_r = stride / cvec;
    step_c = stride - step_r * cvec;
  }
  __device__ __forceinline__ void advance() {
    i += step;
    row += step_r;
    cv += step_c;
    if (cv >= cvec) {
      cv -= cvec;
      row += 1;
    }
  }
};

// ----------------------------------------------------------------------------
// z = act(y * scale[c] + shift[c] (+ res))      16 B per lane, U vectors per thread and sweep.
// ----------------------------------------------------------------------------
template <bool RELU, bool RES, int U>
__global__ __launch_bounds__(256) void bn_apply_kernel(const bf16* __restrict__ y, const bf16* __restrict__ res,
                                                        bf16* __restrict__ z, const float* __restrict__ scale,
                                                        const float* __restrict__ shift,
                                                        unsigned char* __restrict__ mask, unsigned nvec, int C,
                                                        int ldz) {
  // ldz: elements between consecutive output rows (== C for a dense output, larger when z is a channel
  // slice of a wider NHWC tensor, which is how branch outputs land in a concatenation without a copy)
  VecCursor cur[U];
#pragma unroll
  for (int u = 0; u < U; ++u)
    cur[u] = VecCursor((blockIdx.x * U + u) * 256u + threadIdx.x, gridDim.x * (256u * U), (unsigned)C >> 3);
  while (cur[0].i < nvec) {
    u32x4 v[U], r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = cur[u].i < nvec;
      const size_t at = ok ? (size_t)cur[u].i * 8 : 0;
      v[u] = stream_load(y + at);
      if (RES) r[u] = stream_load(res + at);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (cur[u].i < nvec) {
        const int c0 = (int)cur[u].cv << 3;
        u32x4 o;
        unsigned bits = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x2 sc = *(const f32x2*)(scale + c0 + 2 * j), sh = *(const f32x2*)(shift + c0 + 2 * j);
          f32x2 f = __builtin_elementwise_fma(unpack_bf16x2(v[u][j]), sc, sh);
          if (RES) f += unpack_bf16x2(r[u][j]);
          unsigned w = pack_bf16x2(f);
          if (RELU) {
            w = relu_bf16x2(w);
            bits |= positive_bits_bf16x2(w) << (2 * j);
          }
          o[j] = w;
        }
        stream_store(z + (size_t)cur[u].row * ldz + c0, o);
        // one bit per element, (z > 0): the backward pass reads this instead of the whole activation
        if (RELU && mask != nullptr) mask[cur[u].i] = (unsigned char)bits;
      }
      cur[u].advance();
    }
  }
}

extern "C" int dtfa_bn_apply(const void* y, const void* res, void* z, const float* scale, const float* shift,
                             void* mask, long long M, int C, int ldz, int relu, hipStream_t st) {
  if (C % 8 || ldz % 8 || ldz < C) return -1;
  if (mask != nullptr && !relu) return -2;
  size_t nvec = (size_t)M * C / 8;
  if (nvec == 0) return 0;
  if (nvec >= (1ull << 31)) return -3;  // 32-bit vector indices
  constexpr int U = BN_APPLY_U;
  unsigned blocks = (unsigned)((nvec + 256 * U - 1) / (256 * U));
  if (blocks > BN_MAX_BLOCKS) blocks = BN_MAX_BLOCKS;
#define L(R_, S_)                                                                                          \
  hipLaunchKernelGGL((bn_apply_kernel<R_, S_, U>), dim3(blocks), dim3(256), 0, st, (const bf16*)y,         \
                     (const bf16*)res, (bf16*)z, scale, shift, (unsigned char*)mask, (unsigned)nvec, C, ldz)
  if (relu && res) L(true, true);
  else if (relu) L(true, false);
  else if (res) L(false, true);
  else L(false, false);
#undef L
  return (int)hipGetLastError();
}

// ----------------------------------------------------------------------------
// z = act(y * scale[c] + shift[c]) with the coefficients made on the spot from the fp64 column sums a statistics
// epilogue accumulated (ConvArgs::stat_acc): every block derives scale / shift of all C channels into LDS first, block
// 0 also updates the running statistics.  A forward-only layer of few tiles is then two launches, convolution and
// this one -- at 10 .. 20 us per convolution the separate finalize kernel was a tenth of the step.
// ----------------------------------------------------------------------------
#define BN_SUMS_MAX_C 4096
template <bool RELU, int U>
__global__ __launch_bounds__(256) void bn_apply_sums_kernel(const bf16* __restrict__ y, bf16* __restrict__ z,
                                                             const double* __restrict__ sums, float count, float eps,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             float* __restrict__ run_mean, float* __restrict__ run_var,
                                                             float momentum, unsigned nvec, int C, int ldz, int rep) {
  extern __shared__ float coef[];  // [2][C]
  for (int c = threadIdx.x; c < C; c += 256) {
    // (the accumulators may come in ``rep`` replicas of 2 C values, see ConvArgs::stat_rep)
    double s1 = 0.0, s2 = 0.0;
    for (int r = 0; r < rep; ++r) {
      s1 += sums[(size_t)r * 2 * C + c];
      s2 += sums[(size_t)r * 2 * C + C + c];
    }
    const double mu = s1 / (double)count;
    double var = s2 / (double)count - mu * mu;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = (gamma ? gamma[c] : 1.f) * is;
    coef[c] = sc;
    coef[C + c] = (beta ? beta[c] : 0.f) - (float)mu * sc;
    if (blockIdx.x == 0 && run_mean != nullptr) {
      const double unb = count > 1.f ? var * (double)count / ((double)count - 1.0) : var;
      run_mean[c] = run_mean[c] * (1.f - momentum) + (float)mu * momentum;
      run_var[c] = run_var[c] * (1.f - momentum) + (float)unb * momentum;
    }
  }
  __syncthreads();
  VecCursor cur[U];
#pragma unroll
  for (int u = 0; u < U; ++u)
    cur[u] = VecCursor((blockIdx.x * U + u) * 256u + threadIdx.x, gridDim.x * (256u * U), (unsigned)C >> 3);
  while (cur[0].i < nvec) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = *(const u32x4*)(y + (cur[u].i < nvec ? (size_t)cur[u].i * 8 : 0));
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (cur[u].i < nvec) {
        const int c0 = (int)cur[u].cv << 3;
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x2 sc = *(const f32x2*)(coef + c0 + 2 * j), sh = *(const f32x2*)(coef + C + c0 + 2 * j);
          unsigned w = pack_bf16x2(__builtin_elementwise_fma(unpack_bf16x2(v[u][j]), sc, sh));
          if (RELU) w = relu_bf16x2(w);
          o[j] = w;
        }
        *(u32x4*)(z + (size_t)cur[u].row * ldz + c0) = o;
      }
      cur[u].advance();
    }
  }
}

extern "C" int dtfa_bn_apply_sums(const void* y, void* z, const double* sums, float count, float eps,
                                  const float* gamma, const float* beta, float* run_mean, float* run_var,
                                  float momentum, long long M, int C, int ldz, int relu, int rep, hipStream_t st) {
  if (C % 8 || ldz % 8 || ldz < C || C > BN_SUMS_MAX_C || rep < 1 || rep > 64) return -1;
  size_t nvec = (size_t)M * C / 8;
  if (nvec == 0) return 0;
  if (nvec >= (1ull << 31)) return -3;
  constexpr int U = 2;
  unsigned blocks = (unsigned)((nvec + 256 * U - 1) / (256 * U));
  if (blocks > BN_MAX_BLOCKS) blocks = BN_MAX_BLOCKS;
  const size_t lds = (size_t)2 * C * sizeof(float);
  if (relu)
    hipLaunchKernelGGL((bn_apply_sums_kernel<true, U>), dim3(blocks), dim3(256), lds, st, (const bf16*)y, (bf16*)z, sums,
                       count, eps, gamma, beta, run_mean, run_var, momentum, (unsigned)nvec, C, ldz, rep);
  else
    hipLaunchKernelGGL((bn_apply_sums_kernel<false, U>), dim3(blocks), dim3(256), lds, st, (const bf16*)y, (bf16*)z, sums,
                       count, eps, gamma, beta, run_mean, run_var, momentum, (unsigned)nvec, C, ldz, rep);
  return (int)hipGetLastError();
}

// ----------------------------------------------------------------------------
// Column partial sums.  Thread block = TX channel-groups (8 channels each) x
// TY row lanes.  MODE 0: x -> (sum x, sum x^2).  MODE 1 (backward):
//   g = dz * mask,  mask = (zout > 0) [relu 1], (y*scale+shift > 0) [relu 2], a stored bitmask with one
//   bit per element passed through ``zout`` [relu 3], else 1
//   -> (sum g, sum g * xhat),  xhat = (y - mean) * invstd
// Output rows: [gridDim.y][2][C].
// ----------------------------------------------------------------------------
template <int MODE>
__global__ void col_partials_kernel(const bf16* __restrict__ a, const bf16* __restrict__ yv,
                                    const bf16* __restrict__ zout, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, const float* __restrict__ scale,
                                    const float* __restrict__ shift, float* __restrict__ out, long long M, int C,
                                    int TX, int rows_per_block, int relu) {
  extern __shared__ float red[];  // [TY][TX][16]
  const int TY = blockDim.x / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int cg = blockIdx.x * TX + tx;
  const int cvec = C >> 3;
  // sums kept as fp32 pairs (two channels per register pair): the loop body is v_pk_fma_f32 / v_pk_add_f32
  f32x2 s1p[4], s2p[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) s1p[j] = s2p[j] = f32x2{0.f, 0.f};
  if (cg < cvec) {
    // xhat = y * is + xc with xc = -mean * is
    f32x2 is2[4], xc2[4], sc2[4], sh2[4];
    if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x2 mu = *(const f32x2*)(mean + cg * 8 + 2 * j);
        is2[j] = *(const f32x2*)(invstd + cg * 8 + 2 * j);
        xc2[j] = -mu * is2[j];
        sc2[j] = relu == 2 ? *(const f32x2*)(scale + cg * 8 + 2 * j) : f32x2{0.f, 0.f};
        sh2[j] = relu == 2 ? *(const f32x2*)(shift + cg * 8 + 2 * j) : f32x2{0.f, 0.f};
      }
    }
    long long r0 = (long long)blockIdx.y * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    // U rows per iteration: all loads of an iteration are issued before any of them is consumed
    constexpr int U = 4;
    for (long long r = r0 + ty; r < r1; r += (long long)TY * U) {
      u32x4 v[U], yy[U], zz[U];
      unsigned mb[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        long long rr = r + (long long)u * TY;
        ok[u] = rr < r1;
        size_t off = (size_t)(ok[u] ? rr : r) * C + (size_t)cg * 8;
        v[u] = *(const u32x4*)(a + off);
        if (MODE == 1) {
          yy[u] = *(const u32x4*)(yv + off);
          if (relu == 1) zz[u] = *(const u32x4*)(zout + off);
          if (relu == 3) mb[u] = ((const unsigned char*)zout)[off >> 3];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!ok[u]) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (MODE == 0) {
            const f32x2 f = unpack_bf16x2(v[u][j]);
            s1p[j] += f;
            s2p[j] = __builtin_elementwise_fma(f, f, s2p[j]);
          } else {
            const f32x2 yf = unpack_bf16x2(yy[u][j]);
            unsigned gw = v[u][j];
            if (relu == 1) {
              const unsigned zw = zz[u][j];  // z >= 0 after the ReLU: clipped <=> the half is zero
              gw &= ((zw & 0xffffu) != 0 ? 0xffffu : 0u) | ((zw >> 16) != 0 ? 0xffff0000u : 0u);
            } else if (relu == 2) {
              const f32x2 pre = __builtin_elementwise_fma(yf, sc2[j], sh2[j]);
              gw &= (pre.x > 0.f ? 0xffffu : 0u) | (pre.y > 0.f ? 0xffff0000u : 0u);
            } else if (relu == 3) {
              const unsigned m2 = mb[u] >> (2 * j);
              gw &= ((m2 & 1u) ? 0xffffu : 0u) | ((m2 & 2u) ? 0xffff0000u : 0u);
            }
            const f32x2 g = unpack_bf16x2(gw);
            const f32x2 xh = __builtin_elementwise_fma(yf, is2[j], xc2[j]);
            s1p[j] += g;
            s2p[j] = __builtin_elementwise_fma(g, xh, s2p[j]);
          }
        }
      }
    }
  }
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    s1[2 * j] = s1p[j].x;
    s1[2 * j + 1] = s1p[j].y;
    s2[2 * j] = s2p[j].x;
    s2[2 * j + 1] = s2p[j].y;
  }
  float* mine = red + ((size_t)ty * TX + tx) * 16;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    mine[j] = s1[j];
    mine[8 + j] = s2[j];
  }
  __syncthreads();
  if (ty == 0 && cg < cvec) {
    for (int t = 1; t < TY; ++t) {
      const float* o = red + ((size_t)t * TX + tx) * 16;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        s1[j] += o[j];
        s2[j] += o[8 + j];
      }
    }
    float* dst = out + (size_t)blockIdx.y * 2 * C + (size_t)cg * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      dst[j] = s1[j];
      dst[C + j] = s2[j];
    }
  }
}

static void col_partials_geometry(long long M, int C, int* TX, int* gx, int* gy, int* rpb) {
  int cvec = C / 8;
  int tx = 1;
  while (tx * 2 <= cvec && tx < 64) tx *= 2;
  *TX = tx;
  *gx = (cvec + tx - 1) / tx;
  long long want = 2048 / *gx;
  if (want < 1) want = 1;
  long long rows = (M + want - 1) / want;
  int ty = 256 / tx;
  if (rows < ty * 4) rows = ty * 4;
  *rpb = (int)rows;
  *gy = (int)((M + rows - 1) / rows);
}

// Rows of partials produced for a given problem (so the caller can size buffers).
extern "C" int dtfa_col_partials_rows(long long M, int C) {
  int TX, gx, gy, rpb;
  col_partials_geometry(M, C, &TX, &gx, &gy, &rpb);
  return gy;
}

extern "C" int dtfa_bn_stats_partials(const void* x, float* out, long long M, int C, hipStream_t st) {
  if (C % 8) return -1;
  int TX, gx, gy, rpb;
  col_partials_geometry(M, C, &TX, &gx, &gy, &rpb);
  hipLaunchKernelGGL((col_partials_kernel<0>), dim3(gx, gy), dim3(256), 256 * 16 * sizeof(float), st,
                     (const bf16*)x, (const bf16*)nullptr, (const bf16*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, out, M, C, TX, rpb, 0);
  return (int)hipGetLastError();
}

// relu: 0 = no activation, 1 = mask from the stored activation output zout,
//       2 = mask recomputed as (y*scale + shift > 0) (no residual), saving the zout read.
extern "C" int dtfa_bn_bwd_partials(const void* dz, const void* y, const void* zout, const float* mean,
                                    const float* invstd, const float* scale, const float* shift, float* out,
                                    long long M, int C, int relu, hipStream_t st) {
  if (C % 8) return -1;
  if ((relu == 1 && !zout) || (relu == 2 && (!scale || !shift))) return -2;
  int TX, gx, gy, rpb;
  col_partials_geometry(M, C, &TX, &gx, &gy, &rpb);
  hipLaunchKernelGGL((col_partials_kernel<1>), dim3(gx, gy), dim3(256), 256 * 16 * sizeof(float), st,
                     (const bf16*)dz, (const bf16*)y, (const bf16*)zout, mean, invstd, scale, shift, out, M, C, TX,
                     rpb, relu);
  return (int)hipGetLastError();
}

// ----------------------------------------------------------------------------
// Backward finalize: part [R][2][C] = (sum g, sum g*xhat).
//   dbeta = sum g ; dgamma = sum g*xhat
// and the three per-channel coefficients of the elementwise pass
//   dy = gscale * (g - c1 - xhat*c2),  c1 = sum g / M, c2 = sum g*xhat / M
//      = ca * g + cb * y + cc
// ----------------------------------------------------------------------------
template <int RL>
__global__ __launch_bounds__(32 * RL) void bn_bwd_finalize_kernel(
    const float* __restrict__ part, int R, int C, float count, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ dgamma,
    float* __restrict__ dbeta, float* __restrict__ ca, float* __restrict__ cb, float* __restrict__ cc,
    int accumulate) {
  const int c = blockIdx.x * 32 + (threadIdx.x & 31);
  double s, sx;
  if (!reduce_part_rows<RL>(part, R, C, c, s, sx)) return;
  if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)sx;
  if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s;
  float c1 = (float)(s / count), c2 = (float)(sx / count);
  float is = invstd[c];
  float gs = (gamma ? gamma[c] : 1.f) * is;
  ca[c] = gs;
  cb[c] = -gs * is * c2;
  cc[c] = -gs * c1 + gs * is * c2 * mean[c];
}

extern "C" int dtfa_bn_bwd_finalize(const float* part, int R, int C, float count, const float* gamma,
                                    const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                    float* ca, float* cb, float* cc, int accumulate, hipStream_t st) {
  if (finalize_lanes(R) == 8)
    hipLaunchKernelGGL(bn_bwd_finalize_kernel<8>, dim3((C + 31) / 32), dim3(256), 0, st, part, R, C, count, gamma,
                       mean, invstd, dgamma, dbeta, ca, cb, cc, accumulate);
  else
    hipLaunchKernelGGL(bn_bwd_finalize_kernel<32>, dim3((C + 31) / 32), dim3(1024), 0, st, part, R, C, count, gamma,
                       mean, invstd, dgamma, dbeta, ca, cb, cc, accumulate);
  return (int)hipGetLastError();
}

// ----------------------------------------------------------------------------
// dy = ca*g + cb*y + cc,  g = dz * mask ; optionally dres = g.
// ----------------------------------------------------------------------------
template <int RELU, bool DRES, int U>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const bf16* __restrict__ dz, const bf16* __restrict__ y, const bf16* __restrict__ zout, bf16* __restrict__ dy,
    bf16* __restrict__ dres, const float* __restrict__ ca, const float* __restrict__ cb,
    const float* __restrict__ cc, const float* __restrict__ scale, const float* __restrict__ shift, unsigned nvec,
    int C) {
  VecCursor cur[U];
#pragma unroll
  for (int u = 0; u < U; ++u)
    cur[u] = VecCursor((blockIdx.x * U + u) * 256u + threadIdx.x, gridDim.x * (256u * U), (unsigned)C >> 3);
  while (cur[0].i < nvec) {
    u32x4 g4[U], y4[U], z4[U];
    unsigned mb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned i = cur[u].i < nvec ? cur[u].i : 0;
      g4[u] = stream_load(dz + (size_t)i * 8);
      y4[u] = stream_load(y + (size_t)i * 8);
      if (RELU == 1) z4[u] = stream_load(zout + (size_t)i * 8);
      if (RELU == 3) mb[u] = ((const unsigned char*)zout)[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (cur[u].i < nvec) {
        const int c0 = (int)cur[u].cv << 3;
        u32x4 o, r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x2 yv = unpack_bf16x2(y4[u][j]);
          unsigned gw = g4[u][j];
          // zero the gradient where the forward ReLU clipped: keep masks per 16-bit half
          unsigned keep = 0xffffffffu;
          if (RELU == 1) {
            const unsigned zw = z4[u][j];  // z >= 0 after ReLU: clipped <=> the half is zero
            keep = ((zw & 0xffffu) != 0 ? 0xffffu : 0u) | ((zw >> 16) != 0 ? 0xffff0000u : 0u);
          }
          if (RELU == 2) {
            const f32x2 sc = *(const f32x2*)(scale + c0 + 2 * j), sh = *(const f32x2*)(shift + c0 + 2 * j);
            const f32x2 pre = __builtin_elementwise_fma(yv, sc, sh);
            // the forward pass rounded y*scale+shift to bf16 before the ReLU; a positive fp32 stays positive
            keep = (pre.x > 0.f ? 0xffffu : 0u) | (pre.y > 0.f ? 0xffff0000u : 0u);
          }
          if (RELU == 3) {
            const unsigned m2 = mb[u] >> (2 * j);
            keep = ((m2 & 1u) ? 0xffffu : 0u) | ((m2 & 2u) ? 0xffff0000u : 0u);
          }
          if (RELU != 0) gw &= keep;
          const f32x2 g = unpack_bf16x2(gw);
          const f32x2 a = *(const f32x2*)(ca + c0 + 2 * j), bq = *(const f32x2*)(cb + c0 + 2 * j),
                      cq = *(const f32x2*)(cc + c0 + 2 * j);
          o[j] = pack_bf16x2(__builtin_elementwise_fma(a, g, __builtin_elementwise_fma(bq, yv, cq)));
          if (DRES) r[j] = gw;
        }
        stream_store(dy + (size_t)cur[u].i * 8, o);
        if (DRES) stream_store(dres + (size_t)cur[u].i * 8, r);
      }
      cur[u].advance();
    }
  }
}

extern "C" int dtfa_bn_bwd_apply(const void* dz, const void* y, const void* zout, void* dy, void* dres,
                                 const float* ca, const float* cb, const float* cc, const float* scale,
                                 const float* shift, long long M, int C, int relu, hipStream_t st) {
  if (C % 8) return -1;
  if (((relu == 1 || relu == 3) && !zout) || (relu == 2 && (!scale || !shift))) return -2;
  size_t nvec = (size_t)M * C / 8;
  if (nvec == 0) return 0;
  if (nvec >= (1ull << 31)) return -3;
  constexpr int U = BN_BWD_U;
  unsigned blocks = (unsigned)((nvec + 256 * U - 1) / (256 * U));
  if (blocks > BN_MAX_BLOCKS) blocks = BN_MAX_BLOCKS;
#define L(R_, D_)                                                                                            \
  hipLaunchKernelGGL((bn_bwd_apply_kernel<R_, D_, U>), dim3(blocks), dim3(256), 0, st, (const bf16*)dz,      \
                     (const bf16*)y, (const bf16*)zout, (bf16*)dy, (bf16*)dres, ca, cb, cc, scale, shift,   \
                     (unsigned)nvec, C)
  if (relu == 1 && dres) L(1, true);
  else if (relu == 1) L(1, false);
  else if (relu == 2 && dres) L(2, true);
  else if (relu == 2) L(2, false);
  else if (relu == 3 && dres) L(3, true);
  else if (relu == 3) L(3, false);
  else if (dres) L(0, true);
  else L(0, false);
#undef L
  return (int)hipGetLastError();
}

// dx = dz where the bit is set, else 0 (the masked gradient as a tensor, for consumers without a fused path).
__global__ void mask_mul_kernel(const bf16* __restrict__ dz, const unsigned char* __restrict__ mask,
                                bf16* __restrict__ dx, size_t nvec) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    bf16x8 g = *(const bf16x8*)(dz + i * 8);
    unsigned mb = mask[i];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (!((mb >> j) & 1u)) g[j] = (bf16)0.f;
    *(bf16x8*)(dx + i * 8) = g;
  }
}

extern "C" int dtfa_mask_mul(const void* dz, const void* mask, void* dx, long long n, hipStream_t st) {
  if (n % 8) return -1;
  size_t nvec = (size_t)n / 8;
  if (!nvec) return 0;
  unsigned blocks = (unsigned)((nvec + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(mask_mul_kernel, dim3(blocks), dim3(256), 0, st, (const bf16*)dz, (const unsigned char*)mask,
                     (bf16*)dx, nvec);
  return (int)hipGetLastError();
}

// ----------------------------------------------------------------------------
// Plain ReLU backward / elementwise add for bf16 (used where no BN is fused).
// ----------------------------------------------------------------------------
__global__ void relu_bwd_kernel(const bf16* __restrict__ dz, const bf16* __restrict__ z, bf16* __restrict__ dx,
                                size_t nvec) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    bf16x8 g = *(const bf16x8*)(dz + i * 8);
    bf16x8 zz = *(const bf16x8*)(z + i * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (!((float)zz[j] > 0.f)) g[j] = (bf16)0.f;
    *(bf16x8*)(dx + i * 8) = g;
  }
}

extern "C" int dtfa_relu_bwd(const void* dz, const void* z, void* dx, long long n, hipStream_t st) {
  if (n % 8) return -1;
  size_t nvec = (size_t)n / 8;
  if (!nvec) return 0;
  unsigned blocks = (unsigned)((nvec + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(blocks), dim3(256), 0, st, (const bf16*)dz, (const bf16*)z, (bf16*)dx,
                     nvec);
  return (int)hipGetLastError();
}

__global__ void add_bf16_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b, bf16* __restrict__ o,
                                size_t nvec) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    bf16x8 x = *(const bf16x8*)(a + i * 8);
    bf16x8 y = *(const bf16x8*)(b + i * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (bf16)((float)x[j] + (float)y[j]);
    *(bf16x8*)(o + i * 8) = x;
  }
}

extern "C" int dtfa_add_bf16(const void* a, const void* b, void* o, long long n, hipStream_t st) {
  if (n % 8) return -1;
  size_t nvec = (size_t)n / 8;
  if (!nvec) return 0;
  unsigned blocks = (unsigned)((nvec + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(add_bf16_kernel, dim3(blocks), dim3(256), 0, st, (const bf16*)a, (const bf16*)b, (bf16*)o,
                     nvec);
  return (int)hipGetLastError();
}
