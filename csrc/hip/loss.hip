// Fused sparse softmax cross-entropy for small class counts (gfx950).
//
// One 64-lane wave per row.  A single pass produces:
//   loss[row]   = logsumexp(logits) - logits[label] (+ 0.5*act_l2*sum(logits^2))
//   dlogits     = (softmax - onehot) * grad_scale + act_l2*grad_scale*logits
//   top1[row]   = argmax index, score[row] = raw max logit,
//   correct[row]= (argmax == label)
// which covers the loss, its gradient, top-k(1), in-top-k(1) and the optional
// L2 activity regulariser on the logits in one read of the logits.
// Logits are bf16 with row stride ld (>= n_classes; padded columns get zero
// gradient).
#include "common.h"

__global__ void softmax_xent_kernel(const bf16* __restrict__ logits, const int* __restrict__ labels,
                                    float* __restrict__ loss, bf16* __restrict__ dlogits, int* __restrict__ top1,
                                    float* __restrict__ score, int* __restrict__ correct, int B, int n, int ld,
                                    float grad_scale, float act_l2) {
  int row = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= B) return;
  const bf16* lr = logits + (size_t)row * ld;
  float mx = -INFINITY;
  int amax = 0;
  for (int c = lane; c < n; c += 64) {
    float v = (float)lr[c];
    if (v > mx) { mx = v; amax = c; }
  }
  // wave arg-max (lowest index wins ties)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    float om = __shfl_xor(mx, off, 64);
    int oa = __shfl_xor(amax, off, 64);
    if (om > mx || (om == mx && oa < amax)) { mx = om; amax = oa; }
  }
  float se = 0.f, sq = 0.f;
  for (int c = lane; c < n; c += 64) {
    float v = (float)lr[c];
    se += __expf(v - mx);
    sq += v * v;
  }
  se = wave_sum(se);
  sq = wave_sum(sq);
  int lab = labels[row];
  float lse = mx + __logf(se);
  if (lane == 0) {
    float ll = (lab >= 0 && lab < n) ? (float)lr[lab] : 0.f;
    if (loss) loss[row] = lse - ll + 0.5f * act_l2 * sq;
    if (top1) top1[row] = amax;
    if (score) score[row] = mx;
    if (correct) correct[row] = (amax == lab) ? 1 : 0;
  }
  if (dlogits) {
    bf16* dr = dlogits + (size_t)row * ld;
    float inv = 1.f / se;
    for (int c = lane; c < ld; c += 64) {
      float g = 0.f;
      if (c < n) {
        float v = (float)lr[c];
        g = (__expf(v - mx) * inv - (c == lab ? 1.f : 0.f)) * grad_scale + act_l2 * grad_scale * v;
      }
      dr[c] = (bf16)g;
    }
  }
}

extern "C" int dtfa_softmax_xent(const void* logits, const int* labels, float* loss, void* dlogits, int* top1,
                                 float* score, int* correct, int B, int n, int ld, float grad_scale, float act_l2,
                                 hipStream_t st) {
  if (B <= 0) return 0;
  if (n < 1 || ld < n) return -1;
  int rows_per_block = 4;
  hipLaunchKernelGGL(softmax_xent_kernel, dim3((B + rows_per_block - 1) / rows_per_block), dim3(64 * rows_per_block),
                     0, st, (const bf16*)logits, labels, loss, (bf16*)dlogits, top1, score, correct, B, n, ld,
                     grad_scale, act_l2);
  return (int)hipGetLastError();
}

// mean of a small fp32 vector into out[0] (single block, deterministic).
__global__ void mean_f32_kernel(const float* __restrict__ v, float* __restrict__ out, int n, float extra_scale) {
  __shared__ float part[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += v[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += part[w];
    out[0] = t / (float)n * extra_scale;
  }
}

extern "C" int dtfa_mean_f32(const float* v, float* out, int n, float extra_scale, hipStream_t st) {
  if (n <= 0) return -1;
  hipLaunchKernelGGL(mean_f32_kernel, dim3(1), dim3(1024), 0, st, v, out, n, extra_scale);
  return (int)hipGetLastError();
}

// L2 activity regulariser on a bf16 activation (the reference applies l2_regularizer(0.0005) to the head's hidden
// layer, recognition_flower/single/flower_train.py:77):  reg = coef/2 * sum(h^2).
// Forward: up to 256 blocks write one partial each (fixed grid -> deterministic), block 0 of a second tiny launch
// is avoided by letting mean_f32_kernel's caller fold the partials (n = number of blocks, extra_scale = n).
__global__ void act_l2_partials_kernel(const bf16* __restrict__ h, long long n8, float half_coef, float* __restrict__ part) {
  __shared__ float wsum[16];
  float s = 0.f;
  const bf16x8* hv = (const bf16x8*)h;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    bf16x8 v = hv[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float f = (float)v[j];
      s += f * f;
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += wsum[w];
    part[blockIdx.x] = t * half_coef;
  }
}

// Backward, fused with the gradient that arrives from the layer above:  dh = dh_in + (coef * g) * h, where g is
// the (device-resident) gradient of the loss with respect to the regulariser scalar.
__global__ void act_l2_bwd_kernel(const bf16* __restrict__ h, const bf16* __restrict__ dh_in, const float* __restrict__ g,
                                  float coef, bf16* __restrict__ dh, long long n8) {
  const float c = coef * g[0];
  const bf16x8* hv = (const bf16x8*)h;
  const bf16x8* dv = (const bf16x8*)dh_in;
  bf16x8* ov = (bf16x8*)dh;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    bf16x8 a = hv[i], o;
    if (dh_in != nullptr) {
      bf16x8 d = dv[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)d[j] + c * (float)a[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16)(c * (float)a[j]);
    }
    ov[i] = o;
  }
}

extern "C" int dtfa_act_l2_blocks(long long n) {
  long long n8 = n / 8;
  long long b = (n8 + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 256 ? 256 : b));
}

// part must hold dtfa_act_l2_blocks(n) floats; out[0] receives the regulariser.  n % 8 == 0.
extern "C" int dtfa_act_l2_fwd(const void* h, long long n, float coef, float* part, float* out, hipStream_t st) {
  if (n <= 0 || n % 8 != 0) return -1;
  int blocks = dtfa_act_l2_blocks(n);
  hipLaunchKernelGGL(act_l2_partials_kernel, dim3(blocks), dim3(256), 0, st, (const bf16*)h, n / 8, 0.5f * coef, part);
  hipLaunchKernelGGL(mean_f32_kernel, dim3(1), dim3(256), 0, st, (const float*)part, out, blocks, (float)blocks);
  return (int)hipGetLastError();
}

extern "C" int dtfa_act_l2_bwd(const void* h, const void* dh_in, const float* g, float coef, void* dh, long long n,
                               hipStream_t st) {
  if (n <= 0 || n % 8 != 0) return -1;
  long long n8 = n / 8;
  long long blocks = (n8 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(act_l2_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16*)h, (const bf16*)dh_in, g,
                     coef, (bf16*)dh, n8);
  return (int)hipGetLastError();
}
