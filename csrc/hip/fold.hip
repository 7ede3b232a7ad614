// Per-channel algebra of a batch-norm folded through a 1x1 convolution  y = x W^T  (x [M][C], W [N][C]):
// everything the normalisation needs of the N-wide y follows from C-wide moments of x, see
// ops/functional.py:_ExpandBnResFn.  These kernels are the small dense part of that algebra, one wavefront per
// output channel n (a row of W), so that a layer costs two launches here instead of some forty elementwise ones.
//
//   forward   mean_n = w_n . s / M,   E[y^2]_n = w_n . t_n / M   (s = x^T 1,  T = W S,  S = x^T x)
//             -> invstd, scale = gamma * invstd, shift = beta - mean * scale, running statistics
//   backward  with P = g^T x and sum_g = column sums of the gated gradient g:
//             sum_m g y = w_n . p_n  ->  dgamma, dbeta and the coefficients of  dy = c0 g + c1 y + c2
//             dW_n = c0 p_n + c1 t_n + c2 s
//             operands of the data-gradient GEMM over [g | x]:  wd[c][n] = c0 w_n[c] (bf16, transposed),
//             W1_n = c1 w_n (fp32; Q = W^T W1 is a library GEMM), c2
#include "common.h"

__global__ __launch_bounds__(256) void fold_fwd_kernel(
    const bf16* __restrict__ W, const float* __restrict__ T, const float* __restrict__ s,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ mean,
    float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ run_mean,
    float* __restrict__ run_var, int N, int C, float M, float eps, float momentum) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const bf16* w = W + (size_t)n * C;
  const float* t = T + (size_t)n * C;
  float a = 0.f, b = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float wv = (float)w[c];
    a += wv * s[c];
    b += wv * t[c];
  }
  a = wave_sum(a);
  b = wave_sum(b);
  if (lane != 0) return;
  const double mu = (double)a / M;
  double var = (double)b / M - mu * mu;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = (gamma ? gamma[n] : 1.f) * is;
  mean[n] = (float)mu;
  invstd[n] = is;
  scale[n] = sc;
  shift[n] = (beta ? beta[n] : 0.f) - (float)mu * sc;
  if (run_mean) {
    const double unb = M > 1.f ? var * (double)M / ((double)M - 1.0) : var;
    run_mean[n] = run_mean[n] * (1.f - momentum) + (float)mu * momentum;
    run_var[n] = run_var[n] * (1.f - momentum) + (float)unb * momentum;
  }
}

extern "C" int dtfa_fold_fwd(const void* W, const float* T, const float* s, const float* gamma, const float* beta,
                             float* mean, float* invstd, float* scale, float* shift, float* run_mean, float* run_var,
                             int N, int C, float M, float eps, float momentum, hipStream_t st) {
  if (N <= 0 || C <= 0) return -1;
  hipLaunchKernelGGL(fold_fwd_kernel, dim3((N + 3) / 4), dim3(256), 0, st, (const bf16*)W, T, s, gamma, beta, mean,
                     invstd, scale, shift, run_mean, run_var, N, C, M, eps, momentum);
  return (int)hipGetLastError();
}

__global__ __launch_bounds__(256) void fold_bwd_kernel(
    const bf16* __restrict__ W, const float* __restrict__ P, const float* __restrict__ T, const float* __restrict__ s,
    const float* __restrict__ sum_g, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ scale, float M, float* __restrict__ dgamma, float* __restrict__ dbeta,
    float* __restrict__ dW, bf16* __restrict__ wd, float* __restrict__ W1, float* __restrict__ c2_out, int N, int C,
    int ldwd, int acc_gamma, int acc_beta, int acc_w) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const bf16* w = W + (size_t)n * C;
  const float* p = P + (size_t)n * C;
  const float* t = T + (size_t)n * C;
  float a = 0.f;
  for (int c = lane; c < C; c += 64) a += (float)w[c] * p[c];
  a = wave_sum(a);
  const float sg = sum_g[n], is = invstd[n], mu = mean[n], c0 = scale[n];
  const float dg = is * (a - mu * sg);
  const float c1 = -c0 * is * (dg / M);
  const float c2 = -c0 * (sg / M) - c1 * mu;
  for (int c = lane; c < C; c += 64) {
    const float wv = (float)w[c];
    if (dW) {
      const float v = c0 * p[c] + c1 * t[c] + c2 * s[c];
      dW[(size_t)n * C + c] = acc_w ? dW[(size_t)n * C + c] + v : v;
    }
    if (wd) wd[(size_t)c * ldwd + n] = (bf16)(c0 * wv);
    if (W1) W1[(size_t)n * C + c] = c1 * wv;
  }
  if (lane == 0) {
    if (dgamma) dgamma[n] = acc_gamma ? dgamma[n] + dg : dg;
    if (dbeta) dbeta[n] = acc_beta ? dbeta[n] + sg : sg;
    if (c2_out) c2_out[n] = c2;
  }
}

extern "C" int dtfa_fold_bwd(const void* W, const float* P, const float* T, const float* s, const float* sum_g,
                             const float* mean, const float* invstd, const float* scale, float M, float* dgamma,
                             float* dbeta, float* dW, void* wd, float* W1, float* c2_out, int N, int C, int ldwd,
                             int acc_gamma, int acc_beta, int acc_w, hipStream_t st) {
  if (N <= 0 || C <= 0 || (wd != nullptr && ldwd < N)) return -1;
  hipLaunchKernelGGL(fold_bwd_kernel, dim3((N + 3) / 4), dim3(256), 0, st, (const bf16*)W, P, T, s, sum_g, mean,
                     invstd, scale, M, dgamma, dbeta, dW, (bf16*)wd, W1, c2_out, N, C, ldwd, acc_gamma, acc_beta, acc_w);
  return (int)hipGetLastError();
}
