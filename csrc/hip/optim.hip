// Fused optimizer updates over flat parameter shards (gfx950).
//
// Parameters live in one flat fp32 "master" buffer per group together with
// optimizer slots of the same length; gradients arrive as fp32 or bf16 flat
// buffers (the output of a reduce-scatter or all-reduce).  One pass reads the
// gradient and slots, updates the master weights and writes the bf16 copy the
// compute kernels consume, so no separate cast kernel runs.
//
// Adam follows TensorFlow-1 ApplyAdam:
//   lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t)
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ; w -= lr_t * m / (sqrt(v) + eps)
// Momentum SGD:
//   g' = g*gscale + wd*w ; buf = mu*buf + g' ; w -= lr * (nesterov ? g' + mu*buf : buf)
//
// Step-dependent scalars are read from a small device array so a captured
// hipGraph can be replayed with new values:  hyper[0] = lr (or lr_t), hyper[1]
// = gradient scale.
#include "common.h"
#include "opt_update.h"

template <typename G>
__device__ __forceinline__ float4 load_grad4(const G* g, size_t i);
template <>
__device__ __forceinline__ float4 load_grad4<float>(const float* g, size_t i) {
  return *(const float4*)(g + i);
}
template <>
__device__ __forceinline__ float4 load_grad4<bf16>(const bf16* g, size_t i) {
  bf16x4 v = *(const bf16x4*)(g + i);
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}

template <typename G>
__global__ void adam_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                            const G* __restrict__ g, bf16* __restrict__ w16, size_t n,
                            const float* __restrict__ hyper, float b1, float b2, float eps) {
  const float lr_t = hyper[0], gs = hyper[1];
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n;
       i += (size_t)gridDim.x * blockDim.x * 4) {
    if (i + 4 <= n) {
      float4 gg = load_grad4<G>(g, i);
      float4 ww = *(float4*)(w + i), mm = *(float4*)(m + i), vv = *(float4*)(v + i);
      float ga[4] = {gg.x, gg.y, gg.z, gg.w};
      float wa[4] = {ww.x, ww.y, ww.z, ww.w}, ma[4] = {mm.x, mm.y, mm.z, mm.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) adam_update(wa[j], ma[j], va[j], ga[j], gs, lr_t, b1, b2, eps);
      *(float4*)(w + i) = make_float4(wa[0], wa[1], wa[2], wa[3]);
      *(float4*)(m + i) = make_float4(ma[0], ma[1], ma[2], ma[3]);
      *(float4*)(v + i) = make_float4(va[0], va[1], va[2], va[3]);
      if (w16) {
        bf16x4 o = {(bf16)wa[0], (bf16)wa[1], (bf16)wa[2], (bf16)wa[3]};
        *(bf16x4*)(w16 + i) = o;
      }
    } else {
      for (size_t k = i; k < n; ++k) {
        float mk = m[k], vk = v[k], wk = w[k];
        adam_update(wk, mk, vk, (float)g[k], gs, lr_t, b1, b2, eps);
        m[k] = mk; v[k] = vk; w[k] = wk;
        if (w16) w16[k] = (bf16)wk;
      }
    }
  }
}

template <typename G>
__global__ void momentum_kernel(float* __restrict__ w, float* __restrict__ buf, const G* __restrict__ g,
                                bf16* __restrict__ w16, size_t n, const float* __restrict__ hyper, float mu,
                                float wd, int nesterov) {
  const float lr = hyper[0], gs = hyper[1];
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n;
       i += (size_t)gridDim.x * blockDim.x * 4) {
    if (i + 4 <= n) {
      float4 gg = load_grad4<G>(g, i);
      float4 ww = *(float4*)(w + i), bb = *(float4*)(buf + i);
      float ga[4] = {gg.x, gg.y, gg.z, gg.w};
      float wa[4] = {ww.x, ww.y, ww.z, ww.w}, ba[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) momentum_update(wa[j], ba[j], ga[j], gs, lr, mu, wd, nesterov);
      *(float4*)(w + i) = make_float4(wa[0], wa[1], wa[2], wa[3]);
      *(float4*)(buf + i) = make_float4(ba[0], ba[1], ba[2], ba[3]);
      if (w16) {
        bf16x4 o = {(bf16)wa[0], (bf16)wa[1], (bf16)wa[2], (bf16)wa[3]};
        *(bf16x4*)(w16 + i) = o;
      }
    } else {
      for (size_t k = i; k < n; ++k) {
        float bk = buf[k], wk = w[k];
        momentum_update(wk, bk, (float)g[k], gs, lr, mu, wd, nesterov);
        buf[k] = bk; w[k] = wk;
        if (w16) w16[k] = (bf16)wk;
      }
    }
  }
}

static unsigned opt_grid(size_t n) {
  size_t b = (n / 4 + 255) / 256;
  return (unsigned)(b > 4096 ? 4096 : (b ? b : 1));
}

extern "C" int dtfa_adam(float* w, float* m, float* v, const void* g, int g_is_bf16, void* w16, long long n,
                         const float* hyper, float b1, float b2, float eps, hipStream_t st) {
  if (n <= 0) return 0;
  if (g_is_bf16)
    hipLaunchKernelGGL((adam_kernel<bf16>), dim3(opt_grid(n)), dim3(256), 0, st, w, m, v, (const bf16*)g, (bf16*)w16,
                       (size_t)n, hyper, b1, b2, eps);
  else
    hipLaunchKernelGGL((adam_kernel<float>), dim3(opt_grid(n)), dim3(256), 0, st, w, m, v, (const float*)g,
                       (bf16*)w16, (size_t)n, hyper, b1, b2, eps);
  return (int)hipGetLastError();
}

extern "C" int dtfa_momentum(float* w, float* buf, const void* g, int g_is_bf16, void* w16, long long n,
                             const float* hyper, float mu, float wd, int nesterov, hipStream_t st) {
  if (n <= 0) return 0;
  if (g_is_bf16)
    hipLaunchKernelGGL((momentum_kernel<bf16>), dim3(opt_grid(n)), dim3(256), 0, st, w, buf, (const bf16*)g,
                       (bf16*)w16, (size_t)n, hyper, mu, wd, nesterov);
  else
    hipLaunchKernelGGL((momentum_kernel<float>), dim3(opt_grid(n)), dim3(256), 0, st, w, buf, (const float*)g,
                       (bf16*)w16, (size_t)n, hyper, mu, wd, nesterov);
  return (int)hipGetLastError();
}

// fp32 -> bf16 cast of a flat buffer (initial shadow copy, all-gather staging).
__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, bf16* __restrict__ dst, size_t n) {
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n;
       i += (size_t)gridDim.x * blockDim.x * 4) {
    if (i + 4 <= n) {
      float4 v = *(const float4*)(src + i);
      bf16x4 o = {(bf16)v.x, (bf16)v.y, (bf16)v.z, (bf16)v.w};
      *(bf16x4*)(dst + i) = o;
    } else {
      for (size_t k = i; k < n; ++k) dst[k] = (bf16)src[k];
    }
  }
}

extern "C" int dtfa_cast_f32_bf16(const float* src, void* dst, long long n, hipStream_t st) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(opt_grid(n)), dim3(256), 0, st, src, (bf16*)dst, (size_t)n);
  return (int)hipGetLastError();
}

// Upload the two per-step optimizer scalars (step size, gradient scale).  Kernel arguments are captured
// at launch, so the upload is ordered in the stream without any host buffer that has to stay alive.
__global__ void set_hyper_kernel(float* __restrict__ dst, float a, float b) {
  if (threadIdx.x == 0) dst[0] = a;
  if (threadIdx.x == 1) dst[1] = b;
}

extern "C" int dtfa_set_hyper(float* dst, float a, float b, hipStream_t st) {
  hipLaunchKernelGGL(set_hyper_kernel, dim3(1), dim3(64), 0, st, dst, a, b);
  return (int)hipGetLastError();
}
