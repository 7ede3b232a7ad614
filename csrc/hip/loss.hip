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
