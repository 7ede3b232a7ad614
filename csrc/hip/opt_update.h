// Per-element optimizer updates shared by the fused optimizer kernels (optim.hip) and the parameter-server
// apply kernel (xps.hip).  Floating-point contraction is switched off inside them, so every kernel that
// inlines them performs the same IEEE operations in the same order: a parameter updated by an owner shard
// and the same parameter updated by a replicated optimizer come out with identical bits.
#pragma once

// TensorFlow-1 ApplyAdam with the bias correction folded into lr_t:
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ; w -= lr_t * m / (sqrt(v) + eps)
__device__ __forceinline__ void adam_update(float& w, float& m, float& v, float g, float gscale, float lr_t, float b1,
                                            float b2, float eps) {
#pragma clang fp contract(off)
  const float gs = g * gscale;
  m = b1 * m + (1.f - b1) * gs;
  v = b2 * v + (1.f - b2) * gs * gs;
  w = w - lr_t * m / (sqrtf(v) + eps);
}

// Momentum SGD:  g' = g*gscale + wd*w ; buf = mu*buf + g' ; w -= lr * (nesterov ? g' + mu*buf : buf)
__device__ __forceinline__ void momentum_update(float& w, float& buf, float g, float gscale, float lr, float mu, float wd,
                                                int nesterov) {
#pragma clang fp contract(off)
  const float gp = g * gscale + wd * w;
  buf = mu * buf + gp;
  w = w - lr * (nesterov ? gp + mu * buf : buf);
}
