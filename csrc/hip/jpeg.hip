// GPU half of the JPEG decoder (gfx950).  The host entropy decoder
// (csrc/host/jpeg_huffman.cc) delivers quantised DCT coefficients; here
//   jpeg_idct:   dequantise + 8x8 inverse DCT + level shift -> uint8 component planes
//   jpeg_color:  chroma upsampling (libjpeg's "fancy" triangle filters for 2:1 ratios, replication otherwise)
//                + YCbCr -> RGB -> packed HWC uint8
// Both kernels are batched: blockIdx.y selects the image, a descriptor table
// carries per-image geometry, so a whole batch is two launches.
#include "common.h"

struct JpegDesc {
  unsigned long long coef_offset[3];   // int16 units into the coefficient buffer
  unsigned long long plane_offset[3];  // byte offset into the plane buffer
  unsigned long long rgb_offset;       // byte offset into the output buffer
  int width, height, ncomp;
  int hs[3], vs[3], blocks_w[3], blocks_h[3];
  int hmax, vmax;
  unsigned short qt[3][64];
};

__constant__ float kCos[8][8];  // kCos[x][u] = c(u)/2 * cos((2x+1) u pi / 16)

__global__ void jpeg_idct_kernel(const short* __restrict__ coefs, const JpegDesc* __restrict__ descs,
                                 unsigned char* __restrict__ planes) {
  const JpegDesc& d = descs[blockIdx.y];
  int blk = blockIdx.x * blockDim.x + threadIdx.x;
  int c = 0;
  for (; c < d.ncomp; ++c) {
    int nb = d.blocks_w[c] * d.blocks_h[c];
    if (blk < nb) break;
    blk -= nb;
  }
  if (c >= d.ncomp) return;
  const short* src = coefs + d.coef_offset[c] + (size_t)blk * 64;
  float f[64], tmp[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) f[i] = (float)src[i] * (float)d.qt[c][i];
  // rows: tmp[v][x] = sum_u f[v][u] * kCos[x][u]
  for (int v = 0; v < 8; ++v)
    for (int x = 0; x < 8; ++x) {
      float s = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) s += f[v * 8 + u] * kCos[x][u];
      tmp[v * 8 + x] = s;
    }
  int bw = d.blocks_w[c];
  int by = blk / bw, bx = blk % bw;
  int pitch = bw * 8;
  unsigned char* dst = planes + d.plane_offset[c] + (size_t)(by * 8) * pitch + bx * 8;
  for (int y = 0; y < 8; ++y)
    for (int x = 0; x < 8; ++x) {
      float s = 0.f;
#pragma unroll
      for (int v = 0; v < 8; ++v) s += tmp[v * 8 + x] * kCos[y][v];
      float r = rintf(s) + 128.f;
      dst[(size_t)y * pitch + x] = (unsigned char)fminf(fmaxf(r, 0.f), 255.f);
    }
}

__global__ void jpeg_color_kernel(const unsigned char* __restrict__ planes, const JpegDesc* __restrict__ descs,
                                  unsigned char* __restrict__ out) {
  const JpegDesc& d = descs[blockIdx.y];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.width * d.height) return;
  int y = i / d.width, x = i % d.width;
  float comp[3];
  for (int c = 0; c < d.ncomp && c < 3; ++c) {
    const unsigned char* P = planes + d.plane_offset[c];
    const int pitch = d.blocks_w[c] * 8;
    const int hr = d.hmax / d.hs[c], vr = d.vmax / d.vs[c];
    const bool exact = hr * d.hs[c] == d.hmax && vr * d.vs[c] == d.vmax;
    // real (unpadded) size of the subsampled plane: the filters replicate its last row / column, as libjpeg does
    const int cw = (d.width * d.hs[c] + d.hmax - 1) / d.hmax, ch = (d.height * d.vs[c] + d.vmax - 1) / d.vmax;
    int v;
    if (exact && hr == 2 && vr == 2) {
      // h2v2: 3/4 nearer + 1/4 further row, then the same horizontally; rounding biases 8 / 7 alternate per column
      const int cy = y >> 1, cx = x >> 1;
      const int yn = (y & 1) ? min(cy + 1, ch - 1) : max(cy - 1, 0);
      const int xn = (x & 1) ? min(cx + 1, cw - 1) : max(cx - 1, 0);
      const int here = 3 * P[(size_t)cy * pitch + cx] + P[(size_t)yn * pitch + cx];
      const int there = 3 * P[(size_t)cy * pitch + xn] + P[(size_t)yn * pitch + xn];
      v = (3 * here + there + ((x & 1) ? 7 : 8)) >> 4;
    } else if (exact && hr == 2 && vr == 1) {
      const int cx = x >> 1;
      const int xn = (x & 1) ? min(cx + 1, cw - 1) : max(cx - 1, 0);
      v = (3 * P[(size_t)y * pitch + cx] + P[(size_t)y * pitch + xn] + ((x & 1) ? 2 : 1)) >> 2;
    } else if (exact && hr == 1 && vr == 2) {
      const int cy = y >> 1;
      const int yn = (y & 1) ? min(cy + 1, ch - 1) : max(cy - 1, 0);
      v = (3 * P[(size_t)cy * pitch + x] + P[(size_t)yn * pitch + x] + ((y & 1) ? 2 : 1)) >> 2;
    } else {
      const int cy = y * d.vs[c] / d.vmax, cx = x * d.hs[c] / d.hmax;
      v = P[(size_t)cy * pitch + cx];
    }
    comp[c] = (float)v;
  }
  unsigned char* o = out + d.rgb_offset + (size_t)i * 3;
  if (d.ncomp == 1) {
    o[0] = o[1] = o[2] = (unsigned char)comp[0];
    return;
  }
  float Y = comp[0], cb = comp[1] - 128.f, cr = comp[2] - 128.f;
  float r = Y + 1.402f * cr;
  float g = Y - 0.344136f * cb - 0.714136f * cr;
  float b = Y + 1.772f * cb;
  o[0] = (unsigned char)fminf(fmaxf(rintf(r), 0.f), 255.f);
  o[1] = (unsigned char)fminf(fmaxf(rintf(g), 0.f), 255.f);
  o[2] = (unsigned char)fminf(fmaxf(rintf(b), 0.f), 255.f);
}

static bool g_cos_ready[64] = {};   // the cosine table is a per-device symbol

extern "C" int dtfa_jpeg_decode_batch(const void* coefs, const void* descs, void* planes, void* out, int n_images,
                                      int max_blocks, int max_pixels, hipStream_t st) {
  if (n_images <= 0) return 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev &= 63;
  if (!g_cos_ready[dev]) {
    float h[8][8];
    for (int x = 0; x < 8; ++x)
      for (int u = 0; u < 8; ++u)
        h[x][u] = (u == 0 ? 0.35355339059327373f : 0.5f) * cosf((2 * x + 1) * u * 3.14159265358979323846f / 16.f);
    if (hipMemcpyToSymbol(HIP_SYMBOL(kCos), h, sizeof(h)) != hipSuccess) return -1;
    g_cos_ready[dev] = true;
  }
  hipLaunchKernelGGL(jpeg_idct_kernel, dim3((max_blocks + 63) / 64, n_images), dim3(64), 0, st, (const short*)coefs,
                     (const JpegDesc*)descs, (unsigned char*)planes);
  hipLaunchKernelGGL(jpeg_color_kernel, dim3((max_pixels + 255) / 256, n_images), dim3(256), 0, st,
                     (const unsigned char*)planes, (const JpegDesc*)descs, (unsigned char*)out);
  return (int)hipGetLastError();
}

extern "C" int dtfa_jpeg_desc_size() { return (int)sizeof(JpegDesc); }
