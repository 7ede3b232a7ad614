// Batch-norm + ReLU + max-pool fused around the pooling window, NHWC bf16, C % 8 == 0.
//
// A convolution followed by batch-norm, ReLU and a max-pool normally writes the full-resolution
// activation z = relu(y*scale + shift) only for the pool to read it back.  Here the pool reads the
// convolution output y directly and applies the affine map and the ReLU on the fly, so z is never
// stored: the forward pass reads y once (window overlap is served by the cache) and writes the pooled
// tensor plus one winning-tap byte per output element.
//
// Backward needs, per full-resolution element, g = dz * (z > 0) with dz the pooled gradient routed to
// the winning tap of every window that covers the element; both batch-norm backward passes (the
// per-channel sums, then dy = ca*g + cb*y + cc) gather it from the pooled gradient and the tap bytes
// instead of reading a materialised dz.
#include "common.h"

__global__ void bn_relu_maxpool_fwd_kernel(const bf16* __restrict__ y, const float* __restrict__ scale,
                                           const float* __restrict__ shift, bf16* __restrict__ out,
                                           unsigned char* __restrict__ idx, bf16* __restrict__ ywin, PoolArgs p,
                                           int relu, size_t nvec) {
  const int cvec = p.C >> 3;
  // (the launcher guarantees nvec < 2^31: 32-bit index arithmetic)
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)nvec; i += gridDim.x * blockDim.x) {
    const unsigned pix = i / (unsigned)cvec;
    const int cg = (int)(i - pix * cvec);
    const unsigned t = pix / (unsigned)p.OW;
    const int ox = (int)(pix - t * p.OW);
    const int b = (int)(t / (unsigned)p.OH);
    const int oy = (int)(t - (unsigned)b * p.OH);
    float4 sa = *(const float4*)(scale + cg * 8), sb = *(const float4*)(scale + cg * 8 + 4);
    float4 ha = *(const float4*)(shift + cg * 8), hb = *(const float4*)(shift + cg * 8 + 4);
    const float sc[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
    const float sh[8] = {ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w};
    float best[8];
    unsigned char bi[8];
    bf16 braw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      best[j] = -INFINITY;
      bi[j] = 255;
      braw[j] = (bf16)0.f;
    }
    for (int ky = 0; ky < p.kh; ++ky) {
      const int iy = oy * p.sy + ky - p.pt;
      if ((unsigned)iy >= (unsigned)p.IH) continue;
      for (int kx = 0; kx < p.kw; ++kx) {
        const int ix = ox * p.sx + kx - p.pl;
        if ((unsigned)ix >= (unsigned)p.IW) continue;
        bf16x8 v = *(const bf16x8*)(y + (((size_t)b * p.IH + iy) * p.IW + ix) * p.C + cg * 8);
        const unsigned char tap = (unsigned char)(ky * p.kw + kx);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          // the value a separate batch-norm pass would have stored (rounded to bf16), so that ties and the
          // winning tap are exactly those of the unfused sequence
          float f = (float)v[j] * sc[j] + sh[j];
          if (relu) f = fmaxf(f, 0.f);
          f = (float)(bf16)f;
          if (f > best[j] || bi[j] == 255) {
            best[j] = f;
            bi[j] = tap;
            braw[j] = v[j];
          }
        }
      }
    }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)best[j];
    *(bf16x8*)(out + (size_t)i * 8) = o;
    if (idx) {
      uint2 packed;
      packed.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | ((unsigned)bi[3] << 24);
      packed.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | ((unsigned)bi[7] << 24);
      *(uint2*)(idx + (size_t)i * 8) = packed;
    }
    if (ywin) {
      bf16x8 w;
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = braw[j];
      *(bf16x8*)(ywin + (size_t)i * 8) = w;
    }
  }
}

// ReLU fast path of the forward pass.  After the ReLU every candidate is a non-negative bf16, and those
// order like 16-bit integers, so the window is reduced two values per instruction: the affine map with
// v_pk_fma_f32, one rounding (v_cvt_pk_bf16_f32), ReLU and running maximum with v_pk_max_i16.  The winning tap
// is found afterwards by walking the stored candidates backwards and keeping the tap wherever the candidate
// equals the maximum (so the first maximum wins, as in the scalar form).  Taps outside the image hold -1,
// which neither wins nor equals a maximum.
typedef float f32x2p __attribute__((ext_vector_type(2)));
typedef short i16x2p __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2p __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2p __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4p __attribute__((ext_vector_type(4)));

template <int KH, int KW>
__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_packed_kernel(
    const bf16* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
    bf16* __restrict__ out, unsigned char* __restrict__ idx, bf16* __restrict__ ywin, PoolArgs p, unsigned nvec) {
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
  const bf16* ybase = y + (size_t)b * p.IH * p.IW * p.C + cg * 8;
  constexpr int NT = KH * KW;
  u32x4p raw[NT];
  bool ok[NT];
#pragma unroll
  for (int ky = 0; ky < KH; ++ky)
#pragma unroll
    for (int kx = 0; kx < KW; ++kx) {
      const int iy = iy0 + ky, ix = ix0 + kx;
      const int tp = ky * KW + kx;
      ok[tp] = (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
      raw[tp] = *(const u32x4p*)(ybase + (ok[tp] ? ((size_t)iy * p.IW + ix) * p.C : 0));
    }
  f32x2p sc[4], sh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sc[j] = *(const f32x2p*)(scale + cg * 8 + 2 * j);
    sh[j] = *(const f32x2p*)(shift + cg * 8 + 2 * j);
  }
  const i16x2p zero = {0, 0};
  unsigned cand[NT][4], best[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int tp = 0; tp < NT; ++tp)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x2p v;
      v.x = __uint_as_float(raw[tp][j] << 16);
      v.y = __uint_as_float(raw[tp][j] & 0xffff0000u);
      const f32x2p f = __builtin_elementwise_fma(v, sc[j], sh[j]);
      unsigned w = __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2p));
      w = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2p, w), zero));
      w = ok[tp] ? w : 0xffffffffu;
      cand[tp][j] = w;
      best[j] = __builtin_bit_cast(
          unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2p, best[j]), __builtin_bit_cast(i16x2p, w)));
    }
  // winning taps, two 16-bit fields per register
  unsigned win[4] = {0u, 0u, 0u, 0u}, wraw[4] = {0u, 0u, 0u, 0u};   // (wraw: the winners' un-normalised values)
#pragma unroll
  for (int tp = NT - 1; tp >= 0; --tp)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned x = cand[tp][j] ^ best[j];
      const unsigned lo = (x & 0xffffu) == 0 ? 0xffffu : 0u, hi = (x >> 16) == 0 ? 0xffff0000u : 0u;
      const unsigned m = lo | hi;
      win[j] = (win[j] & ~m) | ((unsigned)tp * 0x00010001u & m);
      wraw[j] = (wraw[j] & ~m) | (raw[tp][j] & m);
    }
  u32x4p o = {best[0], best[1], best[2], best[3]};
  *(u32x4p*)(out + (size_t)i * 8) = o;
  if (idx) {
    uint2 packed;  // bytes in channel order: fields (lo, hi) of win[0], win[1] | win[2], win[3]
    packed.x = (win[0] & 0xffu) | ((win[0] >> 8) & 0xff00u) | ((win[1] & 0xffu) << 16) | ((win[1] >> 16) << 24);
    packed.y = (win[2] & 0xffu) | ((win[2] >> 8) & 0xff00u) | ((win[3] & 0xffu) << 16) | ((win[3] >> 16) << 24);
    *(uint2*)(idx + (size_t)i * 8) = packed;
  }
  if (ywin) {
    u32x4p w = {wraw[0], wraw[1], wraw[2], wraw[3]};
    *(u32x4p*)(ywin + (size_t)i * 8) = w;
  }
}

// Pooled gradient routed to full-resolution pixel (b, iy, ix), channels cg*8 .. cg*8+7.
__device__ __forceinline__ void gather_pool_grad(const bf16* __restrict__ dp, const unsigned char* __restrict__ idx,
                                                 const PoolArgs& p, int b, int iy, int ix, int cg, float acc[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  const int ylo = iy + p.pt - p.kh + 1;
  const int oy0 = ylo <= 0 ? 0 : (ylo + p.sy - 1) / p.sy;
  int oy1 = (iy + p.pt) / p.sy;
  if (oy1 > p.OH - 1) oy1 = p.OH - 1;
  const int xlo = ix + p.pl - p.kw + 1;
  const int ox0 = xlo <= 0 ? 0 : (xlo + p.sx - 1) / p.sx;
  int ox1 = (ix + p.pl) / p.sx;
  if (ox1 > p.OW - 1) ox1 = p.OW - 1;
  for (int oy = oy0; oy <= oy1; ++oy) {
    const int ky = iy + p.pt - oy * p.sy;
    for (int ox = ox0; ox <= ox1; ++ox) {
      const int kx = ix + p.pl - ox * p.sx;
      const unsigned tap = (unsigned)(ky * p.kw + kx);
      const size_t o = (((size_t)b * p.OH + oy) * p.OW + ox) * p.C + cg * 8;
      const uint2 pk = *(const uint2*)(idx + o);
      const bf16x8 g = *(const bf16x8*)(dp + o);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned w = j < 4 ? pk.x : pk.y;
        if (((w >> (8 * (j & 3))) & 0xff) == tap) acc[j] += (float)g[j];
      }
    }
  }
}

// Per-channel partial sums (sum g, sum g*xhat), computed from the pooled side: g is non-zero only at the
// winning element of a window, so the sums over all full-resolution elements are sums over pooled elements
// of dp * [relu passed at the winner] * (1, xhat(winner)).  Each thread takes the 8 channels of one pooled
// pixel, decodes the 8 winning taps and fetches just those 8 values of y (2-byte loads that hit the lines
// the neighbouring threads touch), instead of every full-resolution element searching its covering windows.
// Block = TX channel groups x TY pooled-pixel lanes, grid = (channel-group blocks, pixel slices); output
// rows [gridDim.y][2][C].
__global__ void bn_pool_bwd_partials_kernel(const bf16* __restrict__ dp, const unsigned char* __restrict__ idx,
                                            const bf16* __restrict__ y, const float* __restrict__ mean,
                                            const float* __restrict__ invstd, const float* __restrict__ scale,
                                            const float* __restrict__ shift, float* __restrict__ out, PoolArgs p,
                                            int relu, int TX, long long rows_per_block, unsigned kw_magic) {
  extern __shared__ float red[];  // [TY][TX][16]
  const int TY = blockDim.x / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int cg = blockIdx.x * TX + tx;
  const int cvec = p.C >> 3;
  const long long M = (long long)p.B * p.OH * p.OW;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  if (cg < cvec) {
    float mu[8], is[8], sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      mu[j] = mean[cg * 8 + j];
      is[j] = invstd[cg * 8 + j];
      sc[j] = scale[cg * 8 + j];
      sh[j] = shift[cg * 8 + j];
    }
    long long r0l = (long long)blockIdx.y * rows_per_block, r1l = r0l + rows_per_block;
    if (r1l > M) r1l = M;
    const unsigned r1 = (unsigned)r1l;  // M < 2^31 (checked by the launcher)
    constexpr int U = 2;                // pooled pixels per iteration: their loads are all in flight together
    for (unsigned r = (unsigned)r0l + ty; r < r1; r += TY * U) {
      bf16x8 g[U];
      bf16 yw[U][8];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned rr = r + u * TY;
        ok[u] = rr < r1;
        const unsigned rc = ok[u] ? rr : r;
        const unsigned t = rc / (unsigned)p.OW;
        const int ox = (int)(rc - t * p.OW);
        const int b = (int)(t / (unsigned)p.OH);
        const int oy = (int)(t - (unsigned)b * p.OH);
        const size_t o = (size_t)rc * p.C + cg * 8;
        const uint2 pk = *(const uint2*)(idx + o);
        g[u] = *(const bf16x8*)(dp + o);
        const int iy0 = oy * p.sy - p.pt, ix0 = ox * p.sx - p.pl;
        const bf16* ybase = y + (size_t)b * p.IH * p.IW * p.C + cg * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned tap = ((j < 4 ? pk.x : pk.y) >> (8 * (j & 3))) & 0xff;
          const unsigned ky = (tap * kw_magic) >> 16;  // tap / kw, exact for tap < 255 and kw <= 16
          const unsigned kx = tap - ky * p.kw;
          // the winner always lies inside the image (padding never wins); clamp only guards a corrupt byte
          int iy = iy0 + (int)ky, ix = ix0 + (int)kx;
          iy = iy < 0 ? 0 : (iy >= p.IH ? p.IH - 1 : iy);
          ix = ix < 0 ? 0 : (ix >= p.IW ? p.IW - 1 : ix);
          yw[u][j] = ybase[((size_t)iy * p.IW + ix) * p.C + j];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!ok[u]) continue;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float yf = (float)yw[u][j];
          float gj = (float)g[u][j];
          if (relu && !(yf * sc[j] + sh[j] > 0.f)) gj = 0.f;
          s1[j] += gj;
          s2[j] += gj * ((yf - mu[j]) * is[j]);
        }
      }
    }
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
    float* dst = out + (size_t)blockIdx.y * 2 * p.C + (size_t)cg * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      dst[j] = s1[j];
      dst[p.C + j] = s2[j];
    }
  }
}

// The same sums when the forward pass kept each window's winning convolution output (``ywin``, pooled shape): two
// contiguous streams, the pooled gradient and the winners' values, instead of a 2-byte gather per element into
// the full-resolution tensor (which touched most of its lines: 3.3 GB for the ResNet stem at batch 2048).
__global__ void bn_pool_bwd_partials_ywin_kernel(const bf16* __restrict__ dp, const bf16* __restrict__ ywin,
                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                 float* __restrict__ out, int C, long long M, int relu, int TX,
                                                 long long rows_per_block) {
  extern __shared__ float red[];  // [TY][TX][16]
  const int TY = blockDim.x / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int cg = blockIdx.x * TX + tx;
  const int cvec = C >> 3;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  if (cg < cvec) {
    float mu[8], is[8], sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      mu[j] = mean[cg * 8 + j];
      is[j] = invstd[cg * 8 + j];
      sc[j] = scale[cg * 8 + j];
      sh[j] = shift[cg * 8 + j];
    }
    long long r0l = (long long)blockIdx.y * rows_per_block, r1l = r0l + rows_per_block;
    if (r1l > M) r1l = M;
    constexpr int U = 4;
    for (long long r = r0l + ty; r < r1l; r += (long long)TY * U) {
      bf16x8 g[U], yw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long rr = r + (long long)u * TY;
        const size_t o = (size_t)(rr < r1l ? rr : r) * C + cg * 8;
        g[u] = __builtin_nontemporal_load((const bf16x8*)(dp + o));
        yw[u] = __builtin_nontemporal_load((const bf16x8*)(ywin + o));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (r + (long long)u * TY >= r1l) continue;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float yf = (float)yw[u][j];
          float gj = (float)g[u][j];
          if (relu && !(yf * sc[j] + sh[j] > 0.f)) gj = 0.f;
          s1[j] += gj;
          s2[j] += gj * ((yf - mu[j]) * is[j]);
        }
      }
    }
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

__global__ void bn_pool_bwd_apply_kernel(const bf16* __restrict__ dp, const unsigned char* __restrict__ idx,
                                         const bf16* __restrict__ y, bf16* __restrict__ dy,
                                         const float* __restrict__ ca, const float* __restrict__ cb,
                                         const float* __restrict__ cc, const float* __restrict__ scale,
                                         const float* __restrict__ shift, PoolArgs p, int relu, size_t nvec) {
  const int cvec = p.C >> 3;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)nvec; i += gridDim.x * blockDim.x) {
    const unsigned pix = i / (unsigned)cvec;
    const int cg = (int)(i - pix * cvec);
    const unsigned t = pix / (unsigned)p.IW;
    const int ix = (int)(pix - t * p.IW);
    const int b = (int)(t / (unsigned)p.IH);
    const int iy = (int)(t - (unsigned)b * p.IH);
    const bf16x8 yy = *(const bf16x8*)(y + (size_t)i * 8);
    float g[8];
    gather_pool_grad(dp, idx, p, b, iy, ix, cg, g);
    const int c0 = cg * 8;
    const float4 a0 = *(const float4*)(ca + c0), a1 = *(const float4*)(ca + c0 + 4);
    const float4 b0 = *(const float4*)(cb + c0), b1 = *(const float4*)(cb + c0 + 4);
    const float4 d0 = *(const float4*)(cc + c0), d1 = *(const float4*)(cc + c0 + 4);
    const float4 s0 = *(const float4*)(scale + c0), s1 = *(const float4*)(scale + c0 + 4);
    const float4 h0 = *(const float4*)(shift + c0), h1 = *(const float4*)(shift + c0 + 4);
    const float A[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const float Bc[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    const float Cc[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
    const float S[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float H[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float yf = (float)yy[j];
      float gj = g[j];
      if (relu && !(yf * S[j] + H[j] > 0.f)) gj = 0.f;
      o[j] = (bf16)(A[j] * gj + Bc[j] * yf + Cc[j]);
    }
    *(bf16x8*)(dy + (size_t)i * 8) = o;
  }
}

// Stride-2 fast path of the elementwise backward pass (windows up to 3 x 3): a thread owns the 2 x 2 cell of
// full-resolution pixels (2a .. 2a+1, 2c .. 2c+1) for 8 channels.  The cell is covered by at most 2 x 2
// windows, whose pooled gradients and tap bytes are loaded once for the four pixels (the per-pixel search
// above loads 2.25 windows per pixel and repeats the index arithmetic for each).
__global__ __launch_bounds__(256) void bn_pool_bwd_apply_s2_kernel(
    const bf16* __restrict__ dp, const unsigned char* __restrict__ idx, const bf16* __restrict__ y,
    bf16* __restrict__ dy, const float* __restrict__ ca, const float* __restrict__ cb, const float* __restrict__ cc,
    const float* __restrict__ scale, const float* __restrict__ shift, PoolArgs p, int relu, unsigned ncell) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= ncell) return;
  const unsigned cvec = (unsigned)p.C >> 3;
  const unsigned CW = ((unsigned)p.IW + 1) >> 1, CH = ((unsigned)p.IH + 1) >> 1;
  const unsigned cellpix = i / cvec;
  const int cg = (int)(i - cellpix * cvec);
  const unsigned t = cellpix / CW;
  const int cx = (int)(cellpix - t * CW);
  const int b = (int)(t / CH);
  const int cy = (int)(t - (unsigned)b * CH);
  const int iy0 = cy * 2, ix0 = cx * 2;
  // first window row / column that can cover the cell: ceil((iy0 + pt - kh + 1) / 2), argument >= -2
  const int oyb = (iy0 + p.pt - p.kh + 1 + 3) / 2 - 1;
  const int oxb = (ix0 + p.pl - p.kw + 1 + 3) / 2 - 1;

  // ---- all loads first ----
  uint2 pk[2][2];
  bf16x8 gw[2][2];
  bool wok[2][2];
#pragma unroll
  for (int wy = 0; wy < 2; ++wy)
#pragma unroll
    for (int wx = 0; wx < 2; ++wx) {
      const int oy = oyb + wy, ox = oxb + wx;
      // a window takes part if it exists and reaches the cell (a 2 x 2 window reaches only one cell row)
      wok[wy][wx] = (unsigned)oy < (unsigned)p.OH && (unsigned)ox < (unsigned)p.OW &&
                    oy * 2 - p.pt <= iy0 + 1 && ox * 2 - p.pl <= ix0 + 1;
      const size_t o = wok[wy][wx] ? (((size_t)b * p.OH + oy) * p.OW + ox) * p.C + cg * 8 : (size_t)cg * 8;
      pk[wy][wx] = *(const uint2*)(idx + o);
      gw[wy][wx] = *(const bf16x8*)(dp + o);
    }
  bf16x8 yv[2][2];
  bool pok[2][2];
  size_t poff[2][2];
#pragma unroll
  for (int py = 0; py < 2; ++py)
#pragma unroll
    for (int px = 0; px < 2; ++px) {
      pok[py][px] = iy0 + py < p.IH && ix0 + px < p.IW;
      poff[py][px] = pok[py][px] ? (((size_t)b * p.IH + iy0 + py) * p.IW + ix0 + px) * p.C + cg * 8 : (size_t)cg * 8;
      yv[py][px] = *(const bf16x8*)(y + poff[py][px]);
    }
  const int c0 = cg * 8;
  float A[8], Bc[8], Cc[8], S[8], H[8];
  {
    const float4 a0 = *(const float4*)(ca + c0), a1 = *(const float4*)(ca + c0 + 4);
    const float4 b0 = *(const float4*)(cb + c0), b1 = *(const float4*)(cb + c0 + 4);
    const float4 d0 = *(const float4*)(cc + c0), d1 = *(const float4*)(cc + c0 + 4);
    const float4 s0 = *(const float4*)(scale + c0), s1 = *(const float4*)(scale + c0 + 4);
    const float4 h0 = *(const float4*)(shift + c0), h1 = *(const float4*)(shift + c0 + 4);
    A[0] = a0.x; A[1] = a0.y; A[2] = a0.z; A[3] = a0.w; A[4] = a1.x; A[5] = a1.y; A[6] = a1.z; A[7] = a1.w;
    Bc[0] = b0.x; Bc[1] = b0.y; Bc[2] = b0.z; Bc[3] = b0.w; Bc[4] = b1.x; Bc[5] = b1.y; Bc[6] = b1.z; Bc[7] = b1.w;
    Cc[0] = d0.x; Cc[1] = d0.y; Cc[2] = d0.z; Cc[3] = d0.w; Cc[4] = d1.x; Cc[5] = d1.y; Cc[6] = d1.z; Cc[7] = d1.w;
    S[0] = s0.x; S[1] = s0.y; S[2] = s0.z; S[3] = s0.w; S[4] = s1.x; S[5] = s1.y; S[6] = s1.z; S[7] = s1.w;
    H[0] = h0.x; H[1] = h0.y; H[2] = h0.z; H[3] = h0.w; H[4] = h1.x; H[5] = h1.y; H[6] = h1.z; H[7] = h1.w;
  }

  // ---- route the pooled gradients to the cell's pixels ----
  float acc[2][2][8];
#pragma unroll
  for (int py = 0; py < 2; ++py)
#pragma unroll
    for (int px = 0; px < 2; ++px)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[py][px][j] = 0.f;
#pragma unroll
  for (int wy = 0; wy < 2; ++wy)
#pragma unroll
    for (int wx = 0; wx < 2; ++wx) {
      if (!wok[wy][wx]) continue;
      unsigned tap[8];
      float g[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        tap[j] = ((j < 4 ? pk[wy][wx].x : pk[wy][wx].y) >> (8 * (j & 3))) & 0xff;
        g[j] = (float)gw[wy][wx][j];
      }
      const int ky0 = iy0 + p.pt - (oyb + wy) * 2, kx0 = ix0 + p.pl - (oxb + wx) * 2;  // tap of cell pixel (0, 0)
#pragma unroll
      for (int py = 0; py < 2; ++py) {
        const int ky = ky0 + py;
        if ((unsigned)ky >= (unsigned)p.kh) continue;
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          const int kx = kx0 + px;
          if ((unsigned)kx >= (unsigned)p.kw) continue;
          const unsigned want = (unsigned)(ky * p.kw + kx);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[py][px][j] += tap[j] == want ? g[j] : 0.f;
        }
      }
    }

  // ---- dy = ca*g + cb*y + cc ----
#pragma unroll
  for (int py = 0; py < 2; ++py)
#pragma unroll
    for (int px = 0; px < 2; ++px) {
      if (!pok[py][px]) continue;
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float yf = (float)yv[py][px][j];
        float gj = acc[py][px][j];
        if (relu && !(yf * S[j] + H[j] > 0.f)) gj = 0.f;
        o[j] = (bf16)(A[j] * gj + Bc[j] * yf + Cc[j]);
      }
      __builtin_nontemporal_store(o, (bf16x8*)(dy + poff[py][px]));
    }
}

static bool pool_args_ok(const PoolArgs* p) {
  return p->C % 8 == 0 && p->kh * p->kw <= 254 && p->kh > 0 && p->kw > 0 && p->sy > 0 && p->sx > 0 && p->B > 0 &&
         p->OH > 0 && p->OW > 0 && (long long)p->B * p->IH * p->IW * (p->C / 8) < (1ll << 31);
}

static unsigned grid_1d(size_t nvec) {
  size_t b = (nvec + 255) / 256;
  return (unsigned)(b > 16384 ? 16384 : b);
}

extern "C" int dtfa_bn_relu_maxpool_fwd(const void* y, const float* scale, const float* shift, void* out, void* idx,
                                        void* ywin, const PoolArgs* p, int relu, hipStream_t st) {
  if (!pool_args_ok(p)) return -1;
  size_t nvec = (size_t)p->B * p->OH * p->OW * p->C / 8;
  if (relu && p->kh == p->kw && (p->kh == 3 || p->kh == 2)) {
    const dim3 grid((unsigned)((nvec + 255) / 256));
    if (p->kh == 3)
      hipLaunchKernelGGL((bn_relu_maxpool_fwd_packed_kernel<3, 3>), grid, dim3(256), 0, st, (const bf16*)y, scale,
                         shift, (bf16*)out, (unsigned char*)idx, (bf16*)ywin, *p, (unsigned)nvec);
    else
      hipLaunchKernelGGL((bn_relu_maxpool_fwd_packed_kernel<2, 2>), grid, dim3(256), 0, st, (const bf16*)y, scale,
                         shift, (bf16*)out, (unsigned char*)idx, (bf16*)ywin, *p, (unsigned)nvec);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel, dim3(grid_1d(nvec)), dim3(256), 0, st, (const bf16*)y, scale, shift,
                     (bf16*)out, (unsigned char*)idx, (bf16*)ywin, *p, relu, nvec);
  return (int)hipGetLastError();
}

static void partials_geometry(long long M, int C, int* TX, int* gx, int* gy, long long* rpb) {
  int cvec = C / 8, tx = 1;
  while (tx * 2 <= cvec && tx < 64) tx *= 2;
  *TX = tx;
  *gx = (cvec + tx - 1) / tx;
  long long want = 4096 / *gx;
  if (want < 1) want = 1;
  long long rows = (M + want - 1) / want;
  int ty = 256 / tx;
  if (rows < ty * 4) rows = ty * 4;
  *rpb = rows;
  *gy = (int)((M + rows - 1) / rows);
}

extern "C" int dtfa_bn_pool_partials_rows(const PoolArgs* p) {
  int TX, gx, gy;
  long long rpb;
  partials_geometry((long long)p->B * p->OH * p->OW, p->C, &TX, &gx, &gy, &rpb);
  return gy;
}

extern "C" int dtfa_bn_pool_bwd_partials(const void* dp, const void* idx, const void* y, const void* ywin,
                                         const float* mean, const float* invstd, const float* scale,
                                         const float* shift, float* out, const PoolArgs* p, int relu, hipStream_t st) {
  if (!pool_args_ok(p)) return -1;
  int TX, gx, gy;
  long long rpb;
  if (p->kw > 16) return -1;
  partials_geometry((long long)p->B * p->OH * p->OW, p->C, &TX, &gx, &gy, &rpb);
  if (ywin != nullptr) {
    hipLaunchKernelGGL(bn_pool_bwd_partials_ywin_kernel, dim3(gx, gy), dim3(256), 256 * 16 * sizeof(float), st,
                       (const bf16*)dp, (const bf16*)ywin, mean, invstd, scale, shift, out, p->C,
                       (long long)p->B * p->OH * p->OW, relu, TX, rpb);
    return (int)hipGetLastError();
  }
  const unsigned kw_magic = 65536u / (unsigned)p->kw + 1u;
  hipLaunchKernelGGL(bn_pool_bwd_partials_kernel, dim3(gx, gy), dim3(256), 256 * 16 * sizeof(float), st,
                     (const bf16*)dp, (const unsigned char*)idx, (const bf16*)y, mean, invstd, scale, shift, out, *p,
                     relu, TX, rpb, kw_magic);
  return (int)hipGetLastError();
}

extern "C" int dtfa_bn_pool_bwd_apply(const void* dp, const void* idx, const void* y, void* dy, const float* ca,
                                      const float* cb, const float* cc, const float* scale, const float* shift,
                                      const PoolArgs* p, int relu, hipStream_t st) {
  if (!pool_args_ok(p)) return -1;
  if (p->sy == 2 && p->sx == 2 && p->kh <= 3 && p->kw <= 3 && p->kh >= 2 && p->kw >= 2 && p->pt >= 0 && p->pt <= 1 &&
      p->pl >= 0 && p->pl <= 1) {
    const size_t ncell = (size_t)p->B * ((p->IH + 1) / 2) * ((p->IW + 1) / 2) * (p->C / 8);
    hipLaunchKernelGGL(bn_pool_bwd_apply_s2_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, st,
                       (const bf16*)dp, (const unsigned char*)idx, (const bf16*)y, (bf16*)dy, ca, cb, cc, scale, shift,
                       *p, relu, (unsigned)ncell);
    return (int)hipGetLastError();
  }
  size_t nvec = (size_t)p->B * p->IH * p->IW * p->C / 8;
  hipLaunchKernelGGL(bn_pool_bwd_apply_kernel, dim3(grid_1d(nvec)), dim3(256), 0, st, (const bf16*)dp,
                     (const unsigned char*)idx, (const bf16*)y, (bf16*)dy, ca, cb, cc, scale, shift, *p, relu, nvec);
  return (int)hipGetLastError();
}
