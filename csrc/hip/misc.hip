// Small support kernels (gfx950): weight-matrix permutation for the
// data-gradient convolution, counter-based dropout, fills.
#include "common.h"

// ----------------------------------------------------------------------------
// dst[ci][j][co] = src[co][src_tap[j]][ci]      (bf16)
// src is the forward weight matrix [Cout][T][Cin]; dst is the weight matrix of
// the data-gradient convolution restricted to the taps listed in src_tap
// (TP entries).  32x32 tiles of the (co, ci) plane are transposed through LDS so
// both the read and the write are contiguous.
// ----------------------------------------------------------------------------
__global__ void weight_permute_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst,
                                      const int* __restrict__ src_tap, int Cout, int T, int Cin, int TP) {
  __shared__ bf16 tile[32][33];
  const int j = blockIdx.z;
  const int t = src_tap[j];
  const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    int co = co0 + r, ci = ci0 + tx;
    tile[r][tx] = (co < Cout && ci < Cin) ? src[((size_t)co * T + t) * Cin + ci] : (bf16)0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    int ci = ci0 + r, co = co0 + tx;
    if (ci < Cin && co < Cout) dst[((size_t)ci * TP + j) * Cout + co] = tile[tx][r];
  }
}

extern "C" int dtfa_weight_permute(const void* src, void* dst, const int* src_tap, int Cout, int T, int Cin, int TP,
                                   hipStream_t st) {
  if (TP <= 0) return 0;
  dim3 grid((Cin + 31) / 32, (Cout + 31) / 32, TP);
  hipLaunchKernelGGL(weight_permute_kernel, grid, dim3(256), 0, st, (const bf16*)src, (bf16*)dst, src_tap, Cout, T,
                     Cin, TP);
  return (int)hipGetLastError();
}

// ----------------------------------------------------------------------------
// Dropout.  The keep decision for element i is a pure function of
// (seed, i), so the backward pass regenerates the identical mask instead of
// storing it:  y = keep(i) ? x * scale : 0.
// ----------------------------------------------------------------------------
__device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

// ``step`` (optional) points at a device-resident step counter that is folded into the seed, so a
// launch recorded once in a HIP graph still draws a fresh mask on every replay.
__global__ void dropout_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, size_t n, unsigned seed_lo,
                               unsigned seed_hi, const unsigned long long* __restrict__ step, unsigned thresh16,
                               float scale) {
  if (step != nullptr) {
    unsigned long long t = step[0];
    seed_lo ^= mix32((unsigned)t * 0x85ebca6bu + 0x27d4eb2fu);
    seed_hi += mix32((unsigned)(t >> 32) ^ 0xc2b2ae35u) + (unsigned)t;
  }
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; i < n;
       i += (size_t)gridDim.x * blockDim.x * 2) {
    unsigned h = mix32((unsigned)(i >> 1) * 0x9e3779b9u + seed_lo);
    h = mix32(h ^ seed_hi ^ (unsigned)(i >> 33));
    unsigned r0 = h & 0xffff, r1 = h >> 16;
    float a = (float)x[i];
    y[i] = (bf16)(r0 >= thresh16 ? a * scale : 0.f);
    if (i + 1 < n) {
      float b = (float)x[i + 1];
      y[i + 1] = (bf16)(r1 >= thresh16 ? b * scale : 0.f);
    }
  }
}

// rate = drop probability; the same call with dy as input is the backward.
extern "C" int dtfa_dropout(const void* x, void* y, long long n, unsigned long long seed,
                            const unsigned long long* step, float rate, hipStream_t st) {
  if (n <= 0) return 0;
  if (!(rate >= 0.f) || rate >= 1.f) return -1;
  unsigned thresh = (unsigned)(rate * 65536.f + 0.5f);
  float scale = 1.f / (1.f - (float)thresh / 65536.f);
  size_t pairs = ((size_t)n + 1) / 2;
  size_t blocks = (pairs + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16*)x, (bf16*)y, (size_t)n,
                     (unsigned)(seed & 0xffffffffu), (unsigned)(seed >> 32), step, thresh, scale);
  return (int)hipGetLastError();
}

// Library self-description, used by the Python loader to verify the build.
extern "C" const char* dtfa_hip_arch() { return "gfx950"; }
extern "C" int dtfa_hip_abi_version() { return 1; }

// ----------------------------------------------------------------------------
// y[b][oy][ox][:] = x[b][oy * sy][ox * sx][:]   (bf16 NHWC, C a multiple of 8): the block input on the pixel grid of a
// strided projection shortcut (slim resnet_v1 bottleneck, conv_1x1 shortcut with stride: the models behind
// recognition_flower/distributed/train_dataset.py:121-127).  16 bytes per lane, pixel rows contiguous on both sides;
// the generic strided copy of the tensor library ran this at 2.2 TB/s.
// ----------------------------------------------------------------------------
__global__ void subsample_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int IH, int IW, int OH, int OW,
                                 int cvec, int sy, int sx, size_t nvec) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i / cvec;
    const int c = (int)(i - pix * cvec);
    const size_t row = pix / OW;                  // b * OH + oy
    const int ox = (int)(pix - row * OW);
    const size_t b = row / OH;
    const int oy = (int)(row - b * OH);
    const size_t src = ((b * IH + (size_t)oy * sy) * IW + (size_t)ox * sx) * cvec + c;
    const bf16x8 v = __builtin_nontemporal_load((const bf16x8*)x + src);
    ((bf16x8*)y)[i] = v;
  }
}

extern "C" int dtfa_subsample(const void* x, void* y, int B, int IH, int IW, int C, int sy, int sx, hipStream_t st) {
  if (C % 8 || sy < 1 || sx < 1) return -1;
  const int OH = (IH + sy - 1) / sy, OW = (IW + sx - 1) / sx;
  const size_t nvec = (size_t)B * OH * OW * (C / 8);
  if (nvec == 0) return 0;
  size_t blocks = (nvec + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(subsample_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16*)x, (bf16*)y, IH, IW, OH, OW,
                     C / 8, sy, sx, nvec);
  return (int)hipGetLastError();
}

