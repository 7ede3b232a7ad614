// Persistent weight gradient of a same-size 3x3 convolution over 64 -> 64 channels for gfx950 (MI355X).
//
//   dW[co][tap][ci] = sum_m dY[m][co] * X[m + dy(tap) * IW + dx(tap)][ci]        (pixels outside the image: zero)
//
// The ordinary kernel (conv_wgrad.hip) cuts this GEMM -- 64 rows (co) x 576 columns (tap, ci), the pixels as the
// reduction -- into 64 x 128 output tiles, each of which gathers its own copy of the shifted X rows and re-reads dY:
// every X pixel goes through L2 -> LDS nine times and every dY pixel four and a half times (slim conv2d 3x3 under
// recognition_flower/single/flower_train.py:67-68; the 56x56 layers of ResNet-50 ran at 35 % of their byte bound).
// Here, as in conv_p3.hip for the forward pass, a workgroup lives for the whole launch and owns the WHOLE 64 x 576
// gradient in registers (nine 32 x 32 accumulators per wave: a wave is one half of the output channels times one
// half of the input channels, all taps); it walks a contiguous range of 128-pixel tiles, and per tile stages
//   * the 128 dY rows, and
//   * the flattened pixel range the tile touches over all taps, [m0 - IW - 1, m0 + 127 + IW + 1] (the "patch"),
// once, by LDS-DMA.  A tap reads its X fragments from the patch at the row offset dy * IW + dx with the transposing
// ds_read_b64_tr_b16 (the reduction index -- the pixel -- is the slow axis of both operands); a lane whose pixel's
// tap falls outside the image points at a row of zeros instead.  One buffer per workgroup, two workgroups per CU: the
// other workgroup's multiplications hide a tile's round trip.  Every workgroup writes its partial gradient as one
// fp32 slab; the slabs are summed in a fixed order by wgrad_reduce_kernel, so the result is reproducible.
#include "wgrad_args.h"

#ifndef WP3_DBG
#define WP3_DBG 0   // timing-only builds (wrong results): bit 0 drops the tile loads, bit 1 the multiplication loop
#endif

namespace {

constexpr int WP3_BM = 128;      // pixels per tile
constexpr int WP3_ROWB = 128;    // bytes per pixel row (64 channels) of both operands
constexpr int WP3_HEAD = 4;      // LDS rows ahead of the tiles; row 0 is the row of zeros
constexpr int WP3_DY0 = WP3_HEAD;
constexpr int WP3_X0 = WP3_HEAD + WP3_BM;

// XOR applied to the 16-byte chunk index of LDS row ``row`` (128-byte rows: two rows share a 256-byte bank row; the
// four consecutive rows a transposed fragment read touches then cover all banks, whatever the first row)
__device__ __forceinline__ unsigned wp3_swz(unsigned row) { return ((row >> 1) & 1u) << 2; }

__device__ __forceinline__ s16x4 wp3_tr_read(unsigned lds_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(uintptr_t)lds_addr);
}

__global__ __launch_bounds__(256, 2) void conv_wp3_kernel(WgradArgs p, int patch_rows, int tiles_per_wg) {
  constexpr int BM = WP3_BM, ROWB = WP3_ROWB, T = 9, KSTEPS = BM / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(LDS_AS char*)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;          // half of the output channels, half of the input channels
  const unsigned mtiles = (p.M + BM - 1) / BM;
  const unsigned tile_begin = blockIdx.x * (unsigned)tiles_per_wg;
  unsigned tile_end = tile_begin + (unsigned)tiles_per_wg;
  if (tile_end > mtiles) tile_end = mtiles;
  const int margin = p.IW + 1;

  if (tid < 4 * ROWB / 4) ((float*)smem)[tid] = 0.f;   // the head rows (row 0: zeros)

  // ---- staging: 8 rows of 128 bytes per wave instruction; the swizzle repeats every 4 rows and both regions start
  // at a multiple of 4, so a lane's source chunk is the same for every instruction ----
  const int row_in_inst = lane >> 3;
  const unsigned lane_off = row_in_inst * ROWB + (((lane & 7) ^ wp3_swz(row_in_inst)) << 4);
  auto request_tile = [&](unsigned tile) {
    const long long m0 = (long long)tile * BM;
    for (int j = wid; j < BM / 8; j += 4) {          // dY rows m0 .. m0 + 127
      const long long gp = m0 + j * 8 + row_in_inst;
      const bf16* src = gp < (long long)p.M ? (const bf16*)((const char*)(p.dy + (m0 + j * 8) * 64) + lane_off) : p.zero;
      glds16_untracked(src, smem + (WP3_DY0 + j * 8) * ROWB);
    }
    const long long x0 = m0 - margin;
    for (int j = wid; j < patch_rows / 8; j += 4) {   // X rows m0 - margin .. (same-size: pixel index == row of dY)
      const long long gp = x0 + j * 8 + row_in_inst;
      const bf16* src = (gp >= 0 && gp < (long long)p.M) ? (const bf16*)((const char*)(p.x + (x0 + j * 8) * 64) + lane_off) : p.zero;
      glds16_untracked(src, smem + (WP3_X0 + j * 8) * ROWB);
    }
  };

  // ---- taps in scalar registers: row offset inside the patch, bit of a pixel's 3 x 3 validity mask ----
  int toff[T], tbit[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int tp = p.taps[t];
    const int dy = tp >> 16, dx = (int)(short)(tp & 0xffff);
    toff[t] = __builtin_amdgcn_readfirstlane(dy * p.IW + dx + margin);
    tbit[t] = __builtin_amdgcn_readfirstlane((dy + 1) * 3 + dx + 1);
  }

  // ---- transposed fragment geometry (as conv_wgrad.hip): a lane addresses 4 columns of one k row ----
  const int tr_q = (lane & 15) >> 2;
  const int tr_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int tr_k = 8 * (lane >> 5) + tr_q;        // + 4 r + 16 ks
  const unsigned acol = wr * 32 + tr_col, bcol = wc * 32 + tr_col;
  const unsigned acolb = (acol & 7) * 2, bcolb = (bcol & 7) * 2, achunk = acol >> 3, bchunk = bcol >> 3;

  // ---- the image position of the lane's first pixel of the first tile, stepped from there ----
  const int opix = p.IH * p.IW;
  int oy0, ox0;
  {
    const long long m = (long long)tile_begin * BM + tr_k;
    const int rem = (int)(m % opix);
    oy0 = rem / p.IW;
    ox0 = rem - oy0 * p.IW;
  }
  auto advance = [&](int& oy, int& ox, int n) {      // n pixels on (n <= IW)
    ox += n;
    if (ox >= p.IW) {
      ox -= p.IW;
      if (++oy == p.IH) oy = 0;
    }
  };

  // ---- LDS addresses of the lane's fragment pieces at k step 0.  They do not depend on the tile, and a k step on is
  // 16 rows on, which leaves a row's swizzle (bit 1 of its index) as it is: computed once, not per read -- computed per
  // read, the address arithmetic was three times the matrix work of a tile ----
  unsigned aaddr[2], baddr[T][2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const unsigned R = WP3_DY0 + tr_k + 4 * r;
    aaddr[r] = lds0 + R * ROWB + ((achunk ^ wp3_swz(R)) << 4) + acolb;
  }
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const unsigned R = (unsigned)(WP3_X0 + 4 * r + toff[t] + tr_k);
      baddr[t][r] = lds0 + R * ROWB + ((bchunk ^ wp3_swz(R)) << 4) + bcolb;
    }
  const unsigned zaddr = lds0 + (bchunk << 4) + bcolb;   // the same piece of the row of zeros

  f32x16 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;

  for (unsigned tile = tile_begin; tile < tile_end; ++tile) {
    // every wave has finished reading the previous tile (barrier at the end of the loop body)
    if (!(WP3_DBG & 1) || tile == tile_begin) request_tile(tile);

    wait_vmcnt<0>();
    __syncthreads();

    // One k step (16 pixels) at a time, not unrolled (unrolled, a tile's 144 fragment reads are all scheduled ahead
    // and spill).  A lane addresses two pixels per k step (r = 0, 1); their positions in the image are stepped along,
    // and a tap that falls outside the image for such a pixel reads the row of zeros.  Two thirds of the k steps have
    // no pixel on an image border: those skip the per-read choice between the patch and the row of zeros.
    // All fragment reads of a k step are requested before its first multiplication, and each MFMA waits only for the
    // reads it consumes, in request order (with the reads of a tap requested one tap ahead -- a transposed read's
    // latency is several MFMAs long -- the launch took 1.10 ms; requesting step ks + 1 behind the MFMAs of step ks,
    // into the registers they release, costs the registers the kernel does not have: 0.84 ms against 0.73).
    int oy = oy0, ox = ox0;
    const int m_first = (int)(tile * BM) + tr_k;
#pragma unroll 1
    for (int ks = 0; ks < ((WP3_DBG & 2) ? 0 : KSTEPS); ++ks) {
      unsigned okm[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const unsigned colok = 2u | (ox > 0 ? 1u : 0u) | (ox < p.IW - 1 ? 4u : 0u);
        const unsigned up = oy > 0 ? colok : 0u, down = oy < p.IH - 1 ? colok << 6 : 0u;
        okm[r] = m_first + ks * 16 + r * 4 < p.M ? (up | (colok << 3) | down) : 0u;
        advance(oy, ox, r == 0 ? 4 : 12);
      }
      const unsigned koff = (unsigned)(ks * 16 * ROWB);
      s16x8 af, bfr[T];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const s16x4 v = wp3_tr_read(aaddr[r] + koff);
        af[4 * r + 0] = v[0]; af[4 * r + 1] = v[1]; af[4 * r + 2] = v[2]; af[4 * r + 3] = v[3];
      }
      if (__all((okm[0] & okm[1]) == 0x1ffu)) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const s16x4 v = wp3_tr_read(baddr[t][r] + koff);
            bfr[t][4 * r + 0] = v[0]; bfr[t][4 * r + 1] = v[1]; bfr[t][4 * r + 2] = v[2]; bfr[t][4 * r + 3] = v[3];
          }
      } else {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const bool ok = ((okm[r] >> tbit[t]) & 1u) != 0;
            const s16x4 v = wp3_tr_read(ok ? baddr[t][r] + koff : zaddr);
            bfr[t][4 * r + 0] = v[0]; bfr[t][4 * r + 1] = v[1]; bfr[t][4 * r + 2] = v[2]; bfr[t][4 * r + 3] = v[3];
          }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < T; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bfr[t]),
                                                         acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    oy0 = oy;   // 8 x 16 = 128 pixels on: the lane's first pixel of the next tile
    ox0 = ox;
    // every wave is done with the tile before the next one is requested into the same rows
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // ---- the workgroup's partial gradient: slab[blockIdx.x][co][tap * 64 + ci] ----
  float* slab = p.slab + (size_t)blockIdx.x * p.slab_stride;
  const int colq = lane & 31, khalf = lane >> 5;
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int co = wr * 32 + (j & 3) + 8 * (j >> 2) + 4 * khalf;
      slab[(size_t)co * p.K + t * 64 + wc * 32 + colq] = acc[t][j];
    }
}

}  // namespace

// Workgroups (= slabs the caller sums) the launch of ``M`` pixels uses: as many as are resident at once, each a
// contiguous range of tiles.
extern "C" int dtfa_conv_wp3_slabs(int M) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus = n;
  }
  const unsigned mtiles = ((unsigned)M + WP3_BM - 1) / WP3_BM;
  const unsigned slots = (unsigned)cus * 2u;
  const unsigned per = (mtiles + slots - 1) / slots;
  return (int)((mtiles + per - 1) / per);
}

// a->slab: dtfa_conv_wp3_slabs(M) x [64][576] floats.  Returns -10 for shapes the kernel does not take.
extern "C" int dtfa_conv_wp3(WgradArgs* a, hipStream_t stream) {
  if (a->Cin != 64 || a->Cout != 64 || a->T != 9 || a->K != 576 || a->isy != 1 || a->isx != 1 || a->IH != a->OH ||
      a->IW != a->OW || a->IW < 16 || a->colsum || a->M <= 0 || a->M % (a->IH * a->IW) != 0)
    return -10;
  // (the caller vouches that every tap offset lies within one pixel)
  int patch_rows = (WP3_BM + 2 * (a->IW + 1) + 7) & ~7;
  const size_t lds = (size_t)(WP3_X0 + patch_rows) * WP3_ROWB;
  if (lds > 80 * 1024) return -10;
  static bool attr_set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev = (dev >= 0 && dev < 64) ? dev : 0;
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute((const void*)conv_wp3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    attr_set[dev] = true;
  }
  const int wgs = dtfa_conv_wp3_slabs(a->M);
  const unsigned mtiles = ((unsigned)a->M + WP3_BM - 1) / WP3_BM;
  const unsigned per = (mtiles + wgs - 1) / wgs;
  a->slab_stride = (long long)a->Cout * a->K;
  a->splits = wgs;
  hipLaunchKernelGGL(conv_wp3_kernel, dim3(wgs), dim3(256), lds, stream, *a, patch_rows, (int)per);
  return (int)hipGetLastError();
}
