// Persistent weight gradient of a strided convolution over 8-channel pixels for gfx950 (MI355X): the pixel-pair stem of
// ResNet-50 (7 x 4 taps of 8 channels, 64 output channels; slim resnet_v1 root block as used through
// recognition_flower/distributed/train_dataset.py:121-127).
//
//   dW[co][tap][c] = sum_m dY[m][co] * X[f(m) + dy(tap) * IW + dx(tap)][c]        (pixels outside the image: zero)
//
// with f(m) the flattened input pixel of output pixel m's tap origin (conv_p8.hip: the input is stride x the output, so
// the input pixels a 128-pixel tile touches over all taps are one contiguous range).  The ordinary kernel
// (conv_wgrad.hip) gathers a 16-byte chunk per pixel and TAP and reads dY once per column tile: 700 bytes per pixel
// through the L2 -> LDS path, 18 GB per launch at batch 2048.  Here, as in conv_wp3.hip, two workgroups per CU live for
// the whole launch, each holding the WHOLE 64 x 224 gradient in registers -- a wave is one half of the output channels
// and every other 16-pixel k step, seven 32 x 32 accumulators (4 taps x 8 channels wide) -- and walking a contiguous
// range of 128-pixel tiles; per tile the 128 dY rows and the input range are staged once by LDS-DMA (256 bytes per
// pixel), one tile ahead into the other of two buffers.  Both operands are read with the transposing ds_read_b64_tr_b16 (the reduction index, the pixel, is the slow
// axis of both); a lane whose pixel's tap falls outside the image reads a pixel of zeros.  Every wave pair writes its
// partial gradient as one fp32 slab; wgrad_reduce_kernel sums the slabs in a fixed order.
#include "wgrad_args.h"

namespace {

constexpr int WP8_BM = 128;      // pixels per tile
constexpr int WP8_ROWB = 128;    // bytes per dY row (64 channels)
constexpr int WP8_DY0 = 4;       // first LDS row (128-byte units) of the dY tile; row 0 is zeros
constexpr int WP8_XB = (WP8_DY0 + WP8_BM) * WP8_ROWB;   // first byte of the input range

// XOR applied to the 16-byte chunk index of dY row ``row`` (as conv_wp3.hip: the four consecutive rows a transposed
// fragment read touches cover all banks)
__device__ __forceinline__ unsigned wp8_swz(unsigned row) { return ((row >> 1) & 1u) << 2; }

__device__ __forceinline__ s16x4 wp8_tr_read(unsigned lds_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(uintptr_t)lds_addr);
}

__device__ __forceinline__ void wp8_glds16_sbase(const void* base, unsigned lane_off, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(lane_off), "s"(base), "s"(lds_dst)
      : "memory");
}

// NB: column blocks of 32 = 4 taps each (T = 4 NB).  kh x kw: the taps are the dense grid (dy0 + i, dx0 + j),
// row-major, (dy0, dx0) = taps[0] (the caller vouches).
template <int NB>
__global__ __launch_bounds__(256, 2) void conv_wp8_kernel(WgradArgs p, int x_bytes, int tiles_per_wg, int kh, int kw) {
  constexpr int BM = WP8_BM, ROWB = WP8_ROWB, KSTEPS = BM / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(LDS_AS char*)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, ph = wid & 1;          // half of the output channels, parity of the k steps
  const unsigned mtiles = (p.M + BM - 1) / BM;
  const unsigned tile_begin = blockIdx.x * (unsigned)tiles_per_wg;
  unsigned tile_end = tile_begin + (unsigned)tiles_per_wg;
  if (tile_end > mtiles) tile_end = mtiles;

  if (tid < WP8_DY0 * ROWB / 4) ((float*)smem)[tid] = 0.f;   // the head rows (row 0: zeros)

  const int tp0 = p.taps[0];
  const int dy0 = __builtin_amdgcn_readfirstlane(tp0 >> 16), dx0 = __builtin_amdgcn_readfirstlane((int)(short)(tp0 & 0xffff));
  const int toff_min = dy0 * p.IW + dx0, toff_max = (dy0 + kh - 1) * p.IW + dx0 + kw - 1;
  const int fpitch = p.isy * p.IW;                 // input pixels per output row
  const int fwrap = fpitch - p.OW * p.isx;
  const long long npix = (long long)(p.M / (p.OH * p.OW)) * p.IH * p.IW;
  const int max_inst = x_bytes >> 10;
  const unsigned tile_bytes = BM * ROWB + (unsigned)x_bytes;   // one buffer: the dY rows, then the input range

  // ---- staging.  dY: 8 rows of 128 bytes per wave instruction, chunks permuted by the row's swizzle (it repeats every
  // 4 rows and the tile starts at a multiple of 4: one source offset per lane).  Input range: 64 pixels of 16 bytes per
  // instruction from its first pixel on, as they lie in memory. ----
  const int row_in_inst = lane >> 3;
  const unsigned lane_off = row_in_inst * ROWB + (((lane & 7) ^ wp8_swz(row_in_inst)) << 4);
  const int step_y = BM / p.OW, step_x = BM - step_y * p.OW;
  const int q127 = (BM - 1) / p.OW, r127 = BM - 1 - q127 * p.OW;
  int tg = (int)((tile_begin * BM) / (unsigned)p.OW), tx = (int)(tile_begin * BM - (unsigned)tg * p.OW);   // the tile's first pixel
  tg = __builtin_amdgcn_readfirstlane(tg);
  tx = __builtin_amdgcn_readfirstlane(tx);
  auto request_tile = [&](unsigned tile, int buf) -> int {
    const unsigned boff = (unsigned)buf * tile_bytes;
    const long long m0 = (long long)tile * BM;
    for (int j = wid; j < BM / 8; j += 4) {          // dY rows m0 .. m0 + 127
      const long long gp = m0 + j * 8 + row_in_inst;
      const bf16* src = gp < (long long)p.M ? (const bf16*)((const char*)(p.dy + (m0 + j * 8) * 64) + lane_off) : p.zero;
      glds16_untracked(src, smem + boff + (WP8_DY0 + j * 8) * ROWB);
    }
    int g1 = tg + q127, x1 = tx + r127;              // the tile's last pixel
    if (x1 >= p.OW) {
      x1 -= p.OW;
      g1 += 1;
    }
    const int f0 = tg * fpitch + tx * p.isx, f1 = g1 * fpitch + x1 * p.isx;
    const int pst = __builtin_amdgcn_readfirstlane(f0 + toff_min);
    int ninst = __builtin_amdgcn_readfirstlane((f1 + toff_max - pst + 64) >> 6);
    if (ninst > max_inst) ninst = max_inst;
    for (int j = wid; j < ninst; j += 4) {
      const long long first = (long long)pst + j * 64;    // uniform
      const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(boff + WP8_XB + j * 1024);
      if (first >= 0 && first + 64 <= npix) {
        wp8_glds16_sbase(p.x + first * 8, lane * 16, dst);
      } else {   // the tensor's first and last pixels: out-of-range pixels read zeros (their taps are masked anyway)
        const long long q = first + lane;
        const bf16* src = (q >= 0 && q < npix) ? p.x + q * 8 : p.zero;
        glds16_untracked(src, smem + dst);
      }
    }
    tx += step_x;   // on to the next tile's first pixel
    tg += step_y;
    if (tx >= p.OW) {
      tx -= p.OW;
      tg += 1;
    }
    return pst;
  };

  // ---- transposed fragment geometry (as conv_wgrad.hip): a lane addresses 4 columns of one k row ----
  const int tr_q = (lane & 15) >> 2;
  const int tr_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int tr_k = 8 * (lane >> 5) + tr_q;        // + 4 r + 16 ks
  // A (dY): columns are output channels
  const unsigned acol = wr * 32 + tr_col;
  unsigned aaddr[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const unsigned R = WP8_DY0 + tr_k + 4 * r;
    aaddr[r] = lds0 + R * ROWB + (((acol >> 3) ^ wp8_swz(R)) << 4) + (acol & 7) * 2;
  }
  // B (input): the 32 columns of block cb are taps 4 cb .. 4 cb + 3 times 8 channels; the lane addresses 4 channels of
  // tap 4 cb + tapi of its pixel
  const int tapi = tr_col >> 3;
  const unsigned chb = (tr_col & 7) * 2;
  int tl[NB];
#pragma unroll
  for (int cb = 0; cb < NB; ++cb) {
    const int tp = p.taps[4 * cb + tapi];
    const int dy = tp >> 16, dx = (int)(short)(tp & 0xffff);
    tl[cb] = (dy * p.IW + dx) * 16;
  }
  const unsigned zaddr = lds0 + chb;   // the same piece of the pixel of zeros
  unsigned rowpat = 0;   // bit i * kw for every tap row i (validity masks: bit t = i * kw + j of tap (i, j))
  for (int i = 0; i < kh; ++i) rowpat |= 1u << (i * kw);

  // ---- the lane's first pixel of its first k step of the first tile: position in the output and flattened input pixel
  // of the tap origin, stepped from there (4 pixels to the second piece of a k step, 28 to the next k step of this
  // wave's parity: 4 x 32 = a tile) ----
  int oy, ox, fl;
  {
    const unsigned m = tile_begin * BM + 16 * ph + tr_k;
    const int gy = (int)(m / (unsigned)p.OW);
    ox = (int)(m - (unsigned)gy * p.OW);
    oy = gy % p.OH;
    fl = gy * fpitch + ox * p.isx;
  }
  auto advance = [&](int n) {      // n pixels on (n <= OW)
    ox += n;
    fl += n * p.isx;
    if (ox >= p.OW) {
      ox -= p.OW;
      fl += fwrap;
      if (++oy == p.OH) oy = 0;
    }
  };

  f32x16 acc[NB];
#pragma unroll
  for (int cb = 0; cb < NB; ++cb)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[cb][j] = 0.f;

  // Two buffers: tile i + 1 is requested when every wave has left tile i - 1 (the barrier below) and lands while tile i
  // is multiplied.
  int cur = 0;
  int pst = tile_begin < tile_end ? request_tile(tile_begin, 0) : 0;
  for (unsigned tile = tile_begin; tile < tile_end; ++tile) {
    wait_vmcnt<0>();
    __syncthreads();
    int pst_next = 0;
    if (tile + 1 < tile_end) pst_next = request_tile(tile + 1, cur ^ 1);
    const unsigned boff = (unsigned)cur * tile_bytes;

    const int m_first = (int)(tile * BM) + 16 * ph + tr_k;
    const unsigned xbase = lds0 + boff + WP8_XB + chb - (unsigned)pst * 16u;
#pragma unroll 1
    for (int ks = ph; ks < KSTEPS; ks += 2) {
      // the lane's two pixels of this k step: LDS address of the tap origin, and whether every tap is inside the image
      unsigned fb[2];
      int poy[2], pox[2];
      bool inner = true;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        fb[r] = xbase + (unsigned)fl * 16u;
        poy[r] = oy;
        pox[r] = ox;
        const int ix0 = ox * p.isx + dx0, iy0 = oy * p.isy + dy0;
        inner = inner && (unsigned)ix0 <= (unsigned)(p.IW - kw) && (unsigned)iy0 <= (unsigned)(p.IH - kh) &&
                m_first + (ks - ph) * 16 + r * 4 < p.M;
        advance(r == 0 ? 4 : 28);
      }
      const unsigned koff = boff + (unsigned)(ks * 16 * ROWB);
      s16x8 af, bfr[NB];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const s16x4 v = wp8_tr_read(aaddr[r] + koff);
        af[4 * r + 0] = v[0]; af[4 * r + 1] = v[1]; af[4 * r + 2] = v[2]; af[4 * r + 3] = v[3];
      }
      if (__all(inner)) {
#pragma unroll
        for (int cb = 0; cb < NB; ++cb)
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const s16x4 v = wp8_tr_read(fb[r] + (unsigned)tl[cb]);
            bfr[cb][4 * r + 0] = v[0]; bfr[cb][4 * r + 1] = v[1]; bfr[cb][4 * r + 2] = v[2]; bfr[cb][4 * r + 3] = v[3];
          }
      } else {
        unsigned okm[2];   // validity of the pixel's taps, shifted so that bit 4 cb is the lane's tap of block cb
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int ix0 = pox[r] * p.isx + dx0, iy0 = poy[r] * p.isy + dy0;
          const int lo = max(0, -ix0), hi = min(kw, p.IW - ix0);
          const int rlo = max(0, -iy0), rhi = min(kh, p.IH - iy0);
          const unsigned colbits = hi > lo ? (((1u << hi) - 1u) >> lo) << lo : 0u;
          const unsigned rows = rhi > rlo ? (rowpat >> ((kh - (rhi - rlo)) * kw)) << (rlo * kw) : 0u;
          okm[r] = m_first + (ks - ph) * 16 + r * 4 < p.M ? (colbits * rows) >> tapi : 0u;
        }
#pragma unroll
        for (int cb = 0; cb < NB; ++cb)
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const bool ok = ((okm[r] >> (4 * cb)) & 1u) != 0;
            const s16x4 v = wp8_tr_read(ok ? fb[r] + (unsigned)tl[cb] : zaddr);
            bfr[cb][4 * r + 0] = v[0]; bfr[cb][4 * r + 1] = v[1]; bfr[cb][4 * r + 2] = v[2]; bfr[cb][4 * r + 3] = v[3];
          }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int cb = 0; cb < NB; ++cb)
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bfr[cb]),
                                                          acc[cb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    pst = pst_next;
    cur ^= 1;
  }

  // ---- the wave pair's partial gradient: slab[2 blockIdx.x + ph][co][tap * 8 + c] ----
  float* slab = p.slab + ((size_t)blockIdx.x * 2 + ph) * p.slab_stride;
  const int colq = lane & 31, khalf = lane >> 5;
#pragma unroll
  for (int cb = 0; cb < NB; ++cb)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int co = wr * 32 + (j & 3) + 8 * (j >> 2) + 4 * khalf;
      slab[(size_t)co * p.K + cb * 32 + colq] = acc[cb][j];
    }
}

int wp8_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus = n;
  }
  return cus;
}

unsigned wp8_wgs(int M) {
  const unsigned mtiles = ((unsigned)M + WP8_BM - 1) / WP8_BM;
  const unsigned slots = (unsigned)wp8_cus() * 2u;
  const unsigned per = (mtiles + slots - 1) / slots;
  return (mtiles + per - 1) / per;
}

// bytes of the input range of a tile (see dtfa_conv_p8_buffer), or 0 where two workgroups per CU do not fit
int wp8_x_bytes(int iw, int ow, int isy, int isx, int kh, int kw) {
  const int crossed = (WP8_BM - 1 + ow - 1) / ow;
  const long long span = (long long)(WP8_BM - 1) * isx + (long long)crossed * (isy * iw - ow * isx) + (long long)(kh - 1) * iw + kw;
  const long long bytes = ((span + 63) / 64 + 1) * 1024;
  return WP8_XB + WP8_BM * WP8_ROWB + 2 * bytes <= 80 * 1024 ? (int)bytes : 0;
}

}  // namespace

// Slabs the caller sums for a launch of ``M`` pixels: two (one per k-step parity) per resident workgroup.
extern "C" int dtfa_conv_wp8_slabs(int M) { return (int)(2u * wp8_wgs(M)); }

// a->slab: dtfa_conv_wp8_slabs(M) x [64][224] floats.  ``kh`` x ``kw``: the taps are the dense grid (dy0 + i, dx0 + j)
// in row-major order (the caller vouches).  Returns -10 for shapes the kernel does not take.
extern "C" int dtfa_conv_wp8(WgradArgs* a, int kh, int kw, hipStream_t stream) {
  if (a->Cin != 8 || a->Cout != 64 || a->T != 28 || kh * kw != a->T || a->K != a->T * 8 || a->IH != a->OH * a->isy ||
      a->IW != a->OW * a->isx || a->OW < 32 || a->IW <= kw || a->colsum || a->M <= 0 || a->M % (a->OH * a->OW) != 0)
    return -10;
  const int x_bytes = wp8_x_bytes(a->IW, a->OW, a->isy, a->isx, kh, kw);
  if (x_bytes <= 0) return -10;
  const size_t lds = (size_t)WP8_DY0 * WP8_ROWB + 2 * ((size_t)WP8_BM * WP8_ROWB + x_bytes);
  static bool attr_set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev = (dev >= 0 && dev < 64) ? dev : 0;
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute((const void*)conv_wp8_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    attr_set[dev] = true;
  }
  const unsigned wgs = wp8_wgs(a->M);
  const unsigned mtiles = ((unsigned)a->M + WP8_BM - 1) / WP8_BM;
  const unsigned per = (mtiles + wgs - 1) / wgs;
  a->slab_stride = (long long)a->Cout * a->K;
  a->splits = (int)(2 * wgs);
  hipLaunchKernelGGL(conv_wp8_kernel<7>, dim3(wgs), dim3(256), lds, stream, *a, x_bytes, (int)per, kh, kw);
  return (int)hipGetLastError();
}
