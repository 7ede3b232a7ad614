// Persistent strided convolution over 8-channel pixels for gfx950 (MI355X): the image layers (the 7x7 / stride-2 stem of
// ResNet-50 over pixel pairs -- 7 x 4 taps of 8 channels, slim resnet_v1 root block as used through
// recognition_flower/distributed/train_dataset.py:121-127; the first 3x3 / stride-2 layer of the flower CNN,
// recognition_flower/single/flower_train.py:67-68).
//
// A pixel of 8 bf16 channels is one 16-byte chunk, and a tap of it is half an MFMA k-step.  The ordinary implicit-GEMM
// kernel (conv_igemm.hip) gathers those chunks once per TAP: 28 chunks per output pixel for the stem, 11.5 GB through
// the L2 -> LDS path per launch at batch 2048 for a layer that reads 0.8 GB and writes 3.3 GB.  Here, as in conv_p3.hip,
// two workgroups per CU live for the whole launch and walk contiguous ranges of 128-row tiles:
//   * the wave's weights (32 output channels x K <= 256: at most 64 registers) stay in registers;
//   * where the input is stride times the output in both directions (IH = OH * isy, IW = OW * isx), output pixel m =
//     (g, ox) -- g the output row counted over the whole batch -- reads tap (dy, dx) at the flattened input pixel
//     f(m) + dy * IW + dx with f(m) = g * isy * IW + ox * isx, so a tile touches ONE contiguous range of input pixels,
//     [f(m0) + first tap, f(m0 + 127) + last tap]: about a thousand pixels, 16 KB, brought in by LDS-DMA into one of two
//     buffers while the previous tile is multiplied, most of it found in L2 (consecutive tiles overlap);
//   * a lane's A fragment of a k-step is one ds_read_b128 at its pixel's offset plus the tap's, or at a pixel of zeros
//     where the tap falls outside the image;
//   * the epilogue (batch-norm statistics or scale / bias / ReLU, bf16 rows staged through LDS) is conv_epilogue, staged
//     in the buffer the tile has just finished with.
// Same order of accumulation as the ordinary kernel (taps in order, one accumulator chain), so the same bits.
#include "common.h"

#include "conv_epilogue.h"

namespace {

#ifndef P8_WGS
#define P8_WGS 3   // workgroups resident per CU
#endif

#ifndef P8_DBG
#define P8_DBG 0   // timing-only builds (wrong results): bit 1 drops the epilogue, bit 2 the patch loads
#endif

constexpr int P8_BM = 128, P8_BN = 64;
constexpr int P8_EP = 4 * 64 * (32 * 2 + EP_PAD) + 2 * 32 * 2 * 4;   // epilogue row tiles + statistics exchange
constexpr int P8_HEAD = 64;                                          // a pixel of zeros (16 bytes) at LDS offset 0

__device__ __forceinline__ void p8_glds16_sbase(const void* base, unsigned lane_off, unsigned lds_dst) {
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

// KST: k-steps of 16 = pairs of taps, (T + 1) / 2 (an odd tap count ends with a tap of zeros).
// kh x kw: the taps are the dense grid (dy0 + i, dx0 + j), row-major, (dy0, dx0) = taps[0] (the caller vouches).
template <int EPI, int KST>
__global__ __launch_bounds__(256, P8_WGS) void conv_p8_kernel(ConvArgs p, int buf_bytes, int tiles_per_wg, int kh, int kw) {
  constexpr int BM = P8_BM, BN = P8_BN, WM = 64, WN = 32, MT = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const unsigned mtiles = (p.M + BM - 1) / BM;
  const unsigned tile_begin = blockIdx.x * (unsigned)tiles_per_wg;
  unsigned tile_end = tile_begin + (unsigned)tiles_per_wg;
  if (tile_end > mtiles) tile_end = mtiles;
  if (tile_begin >= tile_end) return;

  if (tid < 4) ((float*)smem)[tid] = 0.f;

  const int tp0 = p.taps[0];
  const int dy0 = __builtin_amdgcn_readfirstlane(tp0 >> 16), dx0 = __builtin_amdgcn_readfirstlane((int)(short)(tp0 & 0xffff));
  const int toff_min = dy0 * p.IW + dx0, toff_max = (dy0 + kh - 1) * p.IW + dx0 + kw - 1;
  const int fpitch = p.isy * p.IW;                                     // input pixels per output row
  const long long npix = (long long)(p.M / (p.OYn * p.OXn)) * p.IH * p.IW;   // pixels of the input tensor
  const int max_inst = buf_bytes >> 10;

  // ---- the patch of a tile: 64 pixels per wave instruction, dealt over the waves; returns its first pixel ----
  // (g0, x0): output row (counted over the whole batch) and column of the tile's first pixel, wave-uniform and advanced
  // tile by tile like the lanes' positions below: no division per tile.
  const int step_y = BM / p.OXn, step_x = BM - step_y * p.OXn;
  const int q127 = (BM - 1) / p.OXn, r127 = BM - 1 - q127 * p.OXn;
  auto request_patch = [&](int g0, int x0, int buf) -> int {
    int g1 = g0 + q127, x1 = x0 + r127;   // the tile's last pixel (past the tensor's end in the last tile: clipped below)
    if (x1 >= p.OXn) {
      x1 -= p.OXn;
      g1 += 1;
    }
    const int f0 = g0 * fpitch + x0 * p.isx, f1 = g1 * fpitch + x1 * p.isx;
    const int pst = __builtin_amdgcn_readfirstlane(f0 + toff_min);
    int ninst = __builtin_amdgcn_readfirstlane((f1 + toff_max - pst + 64) >> 6);
    if (ninst > max_inst) ninst = max_inst;
    const unsigned bufoff = P8_HEAD + buf * buf_bytes;
    for (int j = wid; j < ((P8_DBG & 4) ? 0 : ninst); j += 4) {
      const long long first = (long long)pst + j * 64;    // uniform
      const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(bufoff + j * 1024);
      if (first >= 0 && first + 64 <= npix) {
        p8_glds16_sbase(p.x + first * 8, lane * 16, dst);
      } else {   // the tensor's first and last pixels: out-of-range pixels read zeros (their taps are masked anyway)
        const long long q = first + lane;
        const bf16* src = (q >= 0 && q < npix) ? p.x + q * 8 : p.zero;
        glds16_untracked(src, smem + dst);
      }
    }
    return pst;
  };
  int tg = (int)((tile_begin * BM) / (unsigned)p.OXn), tx = (int)(tile_begin * BM - (unsigned)tg * p.OXn);
  tg = __builtin_amdgcn_readfirstlane(tg);
  tx = __builtin_amdgcn_readfirstlane(tx);
  auto next_tile_start = [&]() {
    tx += step_x;
    tg += step_y;
    if (tx >= p.OXn) {
      tx -= p.OXn;
      tg += 1;
    }
  };
  int pst_next = request_patch(tg, tx, 0);

  // ---- the lane's taps: tap 2 s + (lane >> 5) of k-step s, as a byte offset inside a patch ----
  const int khalf = lane >> 5;
  const int T = p.T;
  int tl[KST];
#pragma unroll
  for (int s = 0; s < KST; ++s) {
    const int t = 2 * s + khalf;
    const int tp = p.taps[t < T ? t : 0];
    const int dy = tp >> 16, dx = (int)(short)(tp & 0xffff);
    tl[s] = (dy * p.IW + dx) * 16;
  }

  // ---- the wave's weights: B fragments of all k-steps, [Cout][T * 8] global -> registers, once ----
  const bool wave_live = wc * WN < p.Cout;
  bf16x8 bw[KST];
  {
    const int wn = wc * WN + (lane & 31);
    const bf16* wrow = p.w + (size_t)(wn < p.Cout ? wn : 0) * p.K;
#pragma unroll
    for (int s = 0; s < KST; ++s) {
      bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (wn < p.Cout && 2 * s + khalf < T) v = *(const bf16x8*)(wrow + (2 * s + khalf) * 8);
      bw[s] = v;
    }
  }

  // ---- per-lane rows: position in the output and flattened input pixel of the tap origin, advanced tile by tile ----
  const int fstep = step_y * fpitch + step_x * p.isx, fwrap = fpitch - p.OXn * p.isx;
  int rloc[MT], oy[MT], ox[MT], fl[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    rloc[mt] = wr * WM + mt * 32 + (lane & 31);
    const unsigned m = tile_begin * BM + rloc[mt];
    const int gy = (int)(m / (unsigned)p.OXn);   // output row counted over the whole batch
    ox[mt] = (int)(m - (unsigned)gy * p.OXn);
    oy[mt] = gy % p.OYn;
    fl[mt] = gy * fpitch + ox[mt] * p.isx;
  }
  // validity masks: bit t says that tap t = (i, j), t = i * kw + j, reads inside the image
  unsigned rowpat = 0;   // bit i * kw for every tap row i
  for (int i = 0; i < kh; ++i) rowpat |= 1u << (i * kw);
  const unsigned full_l = (T >= 32 ? 0xffffffffu : (1u << T) - 1u) >> khalf;

  float held[2];
  epi_hold_column<EPI>(p, wc * WN + (lane & 31), held);

  wait_vmcnt<0>();   // the first patch (and the weights)
  __syncthreads();   // ... and the pixel of zeros, of every wave

  int cur = 0;
  for (unsigned tile = tile_begin; tile < tile_end; ++tile) {
    const int m0 = (int)(tile * BM);
    // Every wave has left the previous tile's epilogue (barrier), so the buffer it was staged in may take the next
    // request.  (A raw barrier behind a wait for the wave's own LDS reads: __syncthreads would also wait for the
    // epilogue's global stores, a memory round trip per tile.)
    if (tile != tile_begin) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    const int pst = pst_next;
    if (tile + 1 < tile_end) {
      next_tile_start();
      pst_next = request_patch(tg, tx, cur ^ 1);
    }
    const unsigned bufoff = P8_HEAD + cur * buf_bytes;

    // validity of a lane's rows per tap, shifted by the lane's k half: the valid columns j of a tap row are a range
    // [lo, hi), the valid rows i a range too, and the mask is the product of the column bits and the row pattern
    unsigned maskl[MT];
    unsigned abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int ix0 = ox[mt] * p.isx + dx0, iy0 = oy[mt] * p.isy + dy0;
      const int lo = max(0, -ix0), hi = min(kw, p.IW - ix0);
      const int rlo = max(0, -iy0), rhi = min(kh, p.IH - iy0);
      const unsigned colbits = hi > lo ? (((1u << hi) - 1u) >> lo) << lo : 0u;
      const unsigned rows = rhi > rlo ? (rowpat >> ((kh - (rhi - rlo)) * kw)) << (rlo * kw) : 0u;
      maskl[mt] = m0 + rloc[mt] < p.M ? (colbits * rows) >> khalf : 0u;
      abase[mt] = bufoff + (unsigned)(fl[mt] - pst) * 16u;
    }
    bool inner = true;   // every tap of every row of the wave inside the image (most tiles): no selects
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) inner = inner && maskl[mt] == full_l;
    inner = __all(inner);

    f32x16 acc[MT][1];
    if (!wave_live) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[mt][0][j] = 0.f;
    } else {
      auto tap_loop = [&](const bool every) {
        bf16x8 af[2][MT];
        auto read_step = [&](int s, int dst) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const unsigned a = (every || ((maskl[mt] >> (2 * s)) & 1u)) ? abase[mt] + (unsigned)tl[s] : 0u;
            af[dst][mt] = *(const bf16x8*)(smem + a);
          }
        };
        read_step(0, 0);
#pragma unroll
        for (int s = 0; s < KST; ++s) {
          if (s + 1 < KST) read_step(s + 1, (s + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                af[s & 1][mt], bw[s], s == 0 ? f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0} : acc[mt][0], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      if (inner) tap_loop(true);
      else tap_loop(false);
    }
    // advance the lanes' output positions to the next tile
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      ox[mt] += step_x;
      oy[mt] += step_y;
      fl[mt] += fstep;
      if (ox[mt] >= p.OXn) {
        ox[mt] -= p.OXn;
        oy[mt] += 1;
        fl[mt] += fwrap;
      }
      while (oy[mt] >= p.OYn) oy[mt] -= p.OYn;
    }
    // The next patch has had a tile's worth of work to arrive: wait for it here, before the epilogue's stores are issued
    // (the counter is in order); the barrier publishes it and tells that every wave is done reading this tile's patch,
    // which the epilogue now overwrites.  The stores then drain under the next tile.
    wait_vmcnt<0>();
    __syncthreads();
#if (P8_DBG & 2)
    {
      float sum = 0.f;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 16; ++j) sum += acc[mt][0][j];
      if (sum == 123.456f) p.y[0] = (bf16)sum;
    }
#else
    conv_epilogue<BM, BN, EPI, false>(p, acc, smem + bufoff, wid, lane, m0, 0, tile, true, nullptr, held);
#endif
    cur ^= 1;
  }
}

template <int EPI, int KST>
int launch_p8(const ConvArgs& a, int buf_bytes, int kh, int kw, hipStream_t stream) {
  const size_t lds = P8_HEAD + 2 * (size_t)buf_bytes;
  static bool attr_set[64] = {};
  static int cus[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev = (dev >= 0 && dev < 64) ? dev : 0;
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute((const void*)conv_p8_kernel<EPI, KST>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set[dev] = true;
  }
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
  }
  const unsigned mtiles = (a.M + P8_BM - 1) / P8_BM;
  const unsigned slots = cus[dev] * (unsigned)P8_WGS;   // as many workgroups as fit on the chip at once, each a contiguous range of tiles
  const unsigned per = (mtiles + slots - 1) / slots;
  const unsigned wgs = (mtiles + per - 1) / per;
  hipLaunchKernelGGL((conv_p8_kernel<EPI, KST>), dim3(wgs), dim3(256), lds, stream, a, buf_bytes, (int)per, kh, kw);
  return (int)hipGetLastError();
}

}  // namespace

// Bytes of one patch buffer for a layer of this geometry, or 0 where P8_WGS workgroups with two buffers each do not fit
// into a CU's LDS (very wide images).
extern "C" int dtfa_conv_p8_buffer(int iw, int ow, int isy, int isx, int kh, int kw) {
  // pixels a tile can touch: 127 steps along a row, the rows it can cross, and the span of the taps
  const int crossed = (P8_BM - 1 + ow - 1) / ow;
  const long long span = (long long)(P8_BM - 1) * isx + (long long)crossed * (isy * iw - ow * isx) + (long long)(kh - 1) * iw + kw;
  long long buf_bytes = ((span + 63) / 64 + 1) * 1024;
  if (buf_bytes < P8_EP) buf_bytes = (P8_EP + 1023) & ~1023;
  return P8_HEAD + 2 * buf_bytes <= 160 * 1024 / P8_WGS - 1024 ? (int)buf_bytes : 0;
}

// ``kh`` x ``kw``: the taps of ``a`` are the dense grid (dy0 + i, dx0 + j) in row-major order (the caller vouches: the
// table lives in device memory).  -10: not a shape this kernel takes.
extern "C" int dtfa_conv_p8(const ConvArgs* ap, int kh, int kw, hipStream_t stream) {
  const ConvArgs& a = *ap;
  const bool ok = a.Cin == 8 && a.Cout <= P8_BN && kh >= 1 && kw >= 1 && a.T == kh * kw && a.T <= 32 && a.K == a.T * 8 &&
                  a.osy == 1 && a.osx == 1 && a.oy0 == 0 && a.ox0 == 0 && a.dense_out && !a.pointwise && a.x2 == nullptr &&
                  a.OYn == a.OH && a.OXn == a.OW && a.IH == a.OH * a.isy && a.IW == a.OW * a.isx && a.IW > kw &&
                  a.tile_m_base == 0 && a.tile_m_count == 0 && a.stat_row_base == 0 && a.stat_acc == nullptr &&
                  a.bn_y == nullptr && !a.bn_gate_store && a.M > 0 && a.M % (a.OH * a.OW) == 0 &&
                  (long long)(a.M / (a.OH * a.OW)) * a.IH * a.IW < (1ll << 31) - (1 << 20);
  if (!ok) return -10;
  const int buf_bytes = dtfa_conv_p8_buffer(a.IW, a.OXn, a.isy, a.isx, kh, kw);
  if (buf_bytes <= 0) return -10;
  const int epi = a.stats != nullptr ? 1 : (a.bias != nullptr || a.relu || a.oscale != nullptr || a.ldo != a.Cout) ? 2 : 0;
  const int kst = (a.T + 1) / 2;
#define P8_CASE(E, S) \
  if (epi == E && kst == S) return launch_p8<E, S>(a, buf_bytes, kh, kw, stream);
  P8_CASE(0, 14) P8_CASE(1, 14) P8_CASE(2, 14)
  P8_CASE(0, 5) P8_CASE(1, 5) P8_CASE(2, 5)
#undef P8_CASE
  return -10;
}
