// Pooling kernels for NHWC bf16 tensors on gfx950.  Every thread owns 8
// consecutive channels (one 16-byte access) of one pixel, so all global
// traffic is 16 B per lane and coalesced along C.  C % 8 == 0.
//
//  * max pool: forward stores the winning tap index (uint8) next to the
//    output; backward is a deterministic gather over the <= ceil(k/s)^2
//    windows that cover an input pixel (no atomics).
//  * average pool: divisor either the full window (count_include_pad) or the
//    number of in-image taps (TensorFlow "SAME" semantics).
//  * global average pool and its backward.
#include "common.h"


__global__ void maxpool_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, unsigned char* __restrict__ idx,
                                   PoolArgs p, size_t nvec) {
  const int cvec = p.C >> 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    int cg = (int)(i % cvec);
    size_t pix = i / cvec;
    int ox = (int)(pix % p.OW);
    size_t t = pix / p.OW;
    int oy = (int)(t % p.OH);
    int b = (int)(t / p.OH);
    float best[8];
    unsigned char bi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; bi[j] = 255; }
    for (int ky = 0; ky < p.kh; ++ky) {
      int iy = oy * p.sy + ky - p.pt;
      if ((unsigned)iy >= (unsigned)p.IH) continue;
      for (int kx = 0; kx < p.kw; ++kx) {
        int ix = ox * p.sx + kx - p.pl;
        if ((unsigned)ix >= (unsigned)p.IW) continue;
        bf16x8 v = *(const bf16x8*)(x + (((size_t)b * p.IH + iy) * p.IW + ix) * p.C + cg * 8);
        unsigned char tap = (unsigned char)(ky * p.kw + kx);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float f = (float)v[j];
          if (f > best[j] || bi[j] == 255) { best[j] = f; bi[j] = tap; }
        }
      }
    }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)best[j];
    *(bf16x8*)(y + i * 8) = o;
    if (idx) {
      uint2 packed;
      packed.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | ((unsigned)bi[3] << 24);
      packed.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | ((unsigned)bi[7] << 24);
      *(uint2*)(idx + i * 8) = packed;
    }
  }
}

__global__ void maxpool_bwd_kernel(const bf16* __restrict__ dy, const unsigned char* __restrict__ idx,
                                   bf16* __restrict__ dx, PoolArgs p, size_t nvec) {
  const int cvec = p.C >> 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    int cg = (int)(i % cvec);
    size_t pix = i / cvec;
    int ix = (int)(pix % p.IW);
    size_t t = pix / p.IW;
    int iy = (int)(t % p.IH);
    int b = (int)(t / p.IH);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    // windows oy with oy*sy - pt <= iy <= oy*sy - pt + kh - 1
    int ylo = iy + p.pt - p.kh + 1;
    int oy0 = ylo <= 0 ? 0 : (ylo + p.sy - 1) / p.sy;
    int oy1 = (iy + p.pt) / p.sy;
    if (oy1 > p.OH - 1) oy1 = p.OH - 1;
    int xlo = ix + p.pl - p.kw + 1;
    int ox0 = xlo <= 0 ? 0 : (xlo + p.sx - 1) / p.sx;
    int ox1 = (ix + p.pl) / p.sx;
    if (ox1 > p.OW - 1) ox1 = p.OW - 1;
    for (int oy = oy0; oy <= oy1; ++oy) {
      int ky = iy + p.pt - oy * p.sy;
      for (int ox = ox0; ox <= ox1; ++ox) {
        int kx = ix + p.pl - ox * p.sx;
        unsigned tap = (unsigned)(ky * p.kw + kx);
        size_t o = (((size_t)b * p.OH + oy) * p.OW + ox) * p.C + cg * 8;
        uint2 pk = *(const uint2*)(idx + o);
        bf16x8 g = *(const bf16x8*)(dy + o);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          unsigned w = j < 4 ? pk.x : pk.y;
          unsigned bj = (w >> (8 * (j & 3))) & 0xff;
          if (bj == tap) acc[j] += (float)g[j];
        }
      }
    }
    bf16x8 o8;
#pragma unroll
    for (int j = 0; j < 8; ++j) o8[j] = (bf16)acc[j];
    *(bf16x8*)(dx + i * 8) = o8;
  }
}

__global__ void avgpool_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, PoolArgs p, size_t nvec) {
  const int cvec = p.C >> 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    int cg = (int)(i % cvec);
    size_t pix = i / cvec;
    int ox = (int)(pix % p.OW);
    size_t t = pix / p.OW;
    int oy = (int)(t % p.OH);
    int b = (int)(t / p.OH);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    int cnt = 0;
    for (int ky = 0; ky < p.kh; ++ky) {
      int iy = oy * p.sy + ky - p.pt;
      if ((unsigned)iy >= (unsigned)p.IH) continue;
      for (int kx = 0; kx < p.kw; ++kx) {
        int ix = ox * p.sx + kx - p.pl;
        if ((unsigned)ix >= (unsigned)p.IW) continue;
        bf16x8 v = *(const bf16x8*)(x + (((size_t)b * p.IH + iy) * p.IW + ix) * p.C + cg * 8);
        ++cnt;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
      }
    }
    float inv = 1.f / (float)(p.count_include_pad ? p.kh * p.kw : (cnt > 0 ? cnt : 1));
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)(acc[j] * inv);
    *(bf16x8*)(y + i * 8) = o;
  }
}

// Fixed-window form: one output vector per thread, 32-bit index arithmetic, every tap's load requested before
// the first is consumed, two fp32 sums per instruction (v_pk_add_f32).  Taps are added in the same (ky, kx)
// order as the generic kernel, so the results are identical.
typedef float pool_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int pool_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 pool_bf16x2 __attribute__((ext_vector_type(2)));

template <int KH, int KW>
__global__ __launch_bounds__(256) void avgpool_fwd_fixed_kernel(const bf16* __restrict__ x, bf16* __restrict__ y,
                                                                PoolArgs p, unsigned nvec) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= nvec) return;
  const unsigned cvec = (unsigned)p.C >> 3;
  const unsigned pix = i / cvec;
  const int cg = (int)(i - pix * cvec);
  const unsigned t = pix / (unsigned)p.OW;
  const int ox = (int)(pix - t * p.OW);
  const int b = (int)(t / (unsigned)p.OH);
  const int oy = (int)(t - (unsigned)b * p.OH);
  const int iy0 = oy * p.sy - p.pt, ix0 = ox * p.sx - p.pl;
  const bf16* base = x + (size_t)b * p.IH * p.IW * p.C + cg * 8;
  pool_u32x4 v[KH * KW];
  bool ok[KH * KW];
#pragma unroll
  for (int ky = 0; ky < KH; ++ky)
#pragma unroll
    for (int kx = 0; kx < KW; ++kx) {
      const int iy = iy0 + ky, ix = ix0 + kx;
      ok[ky * KW + kx] = (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
      v[ky * KW + kx] = *(const pool_u32x4*)(base + (ok[ky * KW + kx] ? ((size_t)iy * p.IW + ix) * p.C : 0));
    }
  pool_f32x2 acc[4] = {};
  int cnt = 0;
#pragma unroll
  for (int tp = 0; tp < KH * KW; ++tp) {
    if (!ok[tp]) continue;
    ++cnt;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      pool_f32x2 f;
      f.x = __uint_as_float(v[tp][j] << 16);
      f.y = __uint_as_float(v[tp][j] & 0xffff0000u);
      acc[j] += f;
    }
  }
  const float inv = 1.f / (float)(p.count_include_pad ? KH * KW : (cnt > 0 ? cnt : 1));
  pool_u32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    o[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(acc[j] * inv, pool_bf16x2));
  *(pool_u32x4*)(y + (size_t)i * 8) = o;
}

__device__ __forceinline__ int window_count(int o, int s, int k, int pad, int n) {
  int lo = o * s - pad, hi = lo + k;
  if (lo < 0) lo = 0;
  if (hi > n) hi = n;
  return hi - lo;
}

__global__ void avgpool_bwd_kernel(const bf16* __restrict__ dy, bf16* __restrict__ dx, PoolArgs p, size_t nvec) {
  const int cvec = p.C >> 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    int cg = (int)(i % cvec);
    size_t pix = i / cvec;
    int ix = (int)(pix % p.IW);
    size_t t = pix / p.IW;
    int iy = (int)(t % p.IH);
    int b = (int)(t / p.IH);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    int ylo = iy + p.pt - p.kh + 1;
    int oy0 = ylo <= 0 ? 0 : (ylo + p.sy - 1) / p.sy;
    int oy1 = (iy + p.pt) / p.sy;
    if (oy1 > p.OH - 1) oy1 = p.OH - 1;
    int xlo = ix + p.pl - p.kw + 1;
    int ox0 = xlo <= 0 ? 0 : (xlo + p.sx - 1) / p.sx;
    int ox1 = (ix + p.pl) / p.sx;
    if (ox1 > p.OW - 1) ox1 = p.OW - 1;
    for (int oy = oy0; oy <= oy1; ++oy) {
      int cy = window_count(oy, p.sy, p.kh, p.pt, p.IH);
      for (int ox = ox0; ox <= ox1; ++ox) {
        int cx = window_count(ox, p.sx, p.kw, p.pl, p.IW);
        float inv = 1.f / (float)(p.count_include_pad ? p.kh * p.kw : cy * cx);
        bf16x8 g = *(const bf16x8*)(dy + (((size_t)b * p.OH + oy) * p.OW + ox) * p.C + cg * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (float)g[j] * inv;
      }
    }
    bf16x8 o8;
#pragma unroll
    for (int j = 0; j < 8; ++j) o8[j] = (bf16)acc[j];
    *(bf16x8*)(dx + i * 8) = o8;
  }
}

// [B, HW, C] -> [B, C]  (mean over HW).  One workgroup per (image, tile of up to 256 channel groups): the
// pixels are dealt over S = 256 / (channel groups) thread rows, every thread keeps four independent 16-byte loads
// in flight, and the S partial sums meet in LDS.  (One thread per output walking all pixels with a single load
// in flight ran at 0.3 TB/s.)
__global__ __launch_bounds__(256) void gap_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int B, int HW,
                                                      int C, int TX, int S) {
  __shared__ float red[256 * 8];
  const int cvec = C >> 3;
  const int tx = threadIdx.x % TX, sy = threadIdx.x / TX;
  const int cg = blockIdx.y * TX + tx;
  const int b = blockIdx.x;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (cg < cvec && sy < S) {
    const bf16* src = x + (size_t)b * HW * C + cg * 8;
    int p = sy;
    for (; p + 3 * S < HW; p += 4 * S) {
      bf16x8 v0 = *(const bf16x8*)(src + (size_t)p * C);
      bf16x8 v1 = *(const bf16x8*)(src + (size_t)(p + S) * C);
      bf16x8 v2 = *(const bf16x8*)(src + (size_t)(p + 2 * S) * C);
      bf16x8 v3 = *(const bf16x8*)(src + (size_t)(p + 3 * S) * C);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += ((float)v0[j] + (float)v1[j]) + ((float)v2[j] + (float)v3[j]);
    }
    for (; p < HW; p += S) {
      bf16x8 v = *(const bf16x8*)(src + (size_t)p * C);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = acc[j];
  __syncthreads();
  if (sy == 0 && cg < cvec) {
    for (int t = 1; t < S; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += red[(t * TX + tx) * 8 + j];
    const float inv = 1.f / (float)HW;
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)(acc[j] * inv);
    *(bf16x8*)(y + ((size_t)b * cvec + cg) * 8) = o;
  }
}

__global__ void gap_bwd_kernel(const bf16* __restrict__ dy, bf16* __restrict__ dx, int B, int HW, int C,
                               size_t nvec) {
  const int cvec = C >> 3;
  float inv = 1.f / (float)HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    int cg = (int)(i % cvec);
    size_t b = i / ((size_t)cvec * HW);
    bf16x8 g = *(const bf16x8*)(dy + (b * cvec + cg) * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = (bf16)((float)g[j] * inv);
    *(bf16x8*)(dx + i * 8) = g;
  }
}

static unsigned grid_for(size_t nvec) {
  size_t b = (nvec + 255) / 256;
  return (unsigned)(b > 16384 ? 16384 : (b ? b : 1));
}

extern "C" int dtfa_maxpool_fwd(const void* x, void* y, void* idx, const PoolArgs* p, hipStream_t st) {
  if (p->C % 8 || p->kh * p->kw > 254) return -1;
  size_t nvec = (size_t)p->B * p->OH * p->OW * p->C / 8;
  if (!nvec) return 0;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(nvec)), dim3(256), 0, st, (const bf16*)x, (bf16*)y,
                     (unsigned char*)idx, *p, nvec);
  return (int)hipGetLastError();
}
extern "C" int dtfa_maxpool_bwd(const void* dy, const void* idx, void* dx, const PoolArgs* p, hipStream_t st) {
  if (p->C % 8) return -1;
  size_t nvec = (size_t)p->B * p->IH * p->IW * p->C / 8;
  if (!nvec) return 0;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(nvec)), dim3(256), 0, st, (const bf16*)dy,
                     (const unsigned char*)idx, (bf16*)dx, *p, nvec);
  return (int)hipGetLastError();
}
extern "C" int dtfa_avgpool_fwd(const void* x, void* y, const PoolArgs* p, hipStream_t st) {
  if (p->C % 8) return -1;
  size_t nvec = (size_t)p->B * p->OH * p->OW * p->C / 8;
  if (!nvec) return 0;
  if (p->kh == 3 && p->kw == 3 && nvec < (1ull << 31)) {
    hipLaunchKernelGGL((avgpool_fwd_fixed_kernel<3, 3>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, st,
                       (const bf16*)x, (bf16*)y, *p, (unsigned)nvec);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(grid_for(nvec)), dim3(256), 0, st, (const bf16*)x, (bf16*)y, *p, nvec);
  return (int)hipGetLastError();
}
extern "C" int dtfa_avgpool_bwd(const void* dy, void* dx, const PoolArgs* p, hipStream_t st) {
  if (p->C % 8) return -1;
  size_t nvec = (size_t)p->B * p->IH * p->IW * p->C / 8;
  if (!nvec) return 0;
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for(nvec)), dim3(256), 0, st, (const bf16*)dy, (bf16*)dx, *p, nvec);
  return (int)hipGetLastError();
}
extern "C" int dtfa_gap_fwd(const void* x, void* y, int B, int HW, int C, hipStream_t st) {
  if (C % 8) return -1;
  size_t n = (size_t)B * C / 8;
  if (!n) return 0;
  const int cvec = C / 8;
  int TX = 1;
  while (TX < cvec && TX < 256) TX *= 2;  // channel groups per workgroup (power of two, so TX * S == 256)
  const int S = 256 / TX;
  hipLaunchKernelGGL(gap_fwd_kernel, dim3((unsigned)B, (unsigned)((cvec + TX - 1) / TX)), dim3(256), 0, st,
                     (const bf16*)x, (bf16*)y, B, HW, C, TX, S);
  return (int)hipGetLastError();
}
extern "C" int dtfa_gap_bwd(const void* dy, void* dx, int B, int HW, int C, hipStream_t st) {
  if (C % 8) return -1;
  size_t nvec = (size_t)B * HW * C / 8;
  if (!nvec) return 0;
  hipLaunchKernelGGL(gap_bwd_kernel, dim3(grid_for(nvec)), dim3(256), 0, st, (const bf16*)dy, (bf16*)dx, B, HW, C,
                     nvec);
  return (int)hipGetLastError();
}
