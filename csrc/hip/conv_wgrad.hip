// Weight-gradient of the tap-table convolution for gfx950 (MI355X).
//
//   dW[co][tap][ci] = sum_m dY[m][co] * X[pixel(m) + tap][ci]
//
// GEMM view: rows = cout, cols = (tap, cin), reduction = output pixels m.
// Both operands are stored pixel-major (NHWC), i.e. the reduction index is the
// *slow* memory axis, so the MFMA fragments (8 consecutive k per lane) are
// fetched from the LDS tiles with the gfx950 transposing read
// ds_read_b64_tr_b16.  Tiles are staged with LDS-DMA (global_load_lds_dwordx4);
// the im2col gather and the zero padding live in the per-lane source address,
// and the bank swizzle is applied through the source chunk choice.
//
// The reduction is split across blockIdx.y; every split writes an fp32 slab
// [split][Cout][K] and a second kernel sums the slabs in a fixed order, so the
// result is bitwise reproducible.
#include "common.h"

#include "wgrad_args.h"

// XOR applied to the 16-byte chunk index of tile row ``row`` (RB bytes per row).  A transposed fragment read
// touches 64 bytes in each of four consecutive rows: rows of 256 bytes or more all start at bank 0, so the
// swizzle moves row r by (r & 3) * 64 bytes; with 128-byte rows two rows share a bank row and pairs are moved.
template <int RB>
__device__ __forceinline__ int wswz(int row) {
  if (RB >= 256) return (row & 3) << 2;
  constexpr int RBR = 256 / RB;  // rows per 256-byte bank row
  return ((row / RBR) & (4 / RBR - 1)) << 2;
}

__device__ __forceinline__ s16x4 lds_tr_read(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)p);
}

#ifndef WGRAD_MFMA_FIRST
#define WGRAD_MFMA_FIRST 0
#endif

// ONES: the launch also produces the column sums of its X operand over the pixels, sum_m X[m][n] -- one more MFMA
// per fragment column with an all-ones row operand held in registers, in the waves of the first row of tiles.  The
// moments a folded batch-norm needs of a tensor (x^T x and x^T 1) then come from one pass over it.
// NW: waves per workgroup, 4 (2 x 2) or 8 (2 rows x 4 columns of waves: the 256 x 256 tile, each wave 128 x 64 as in the
// 256 x 128 tile, a third fewer operand bytes through L2 -> LDS per FLOP, one workgroup per CU).
template <int BMW, int BNW, int BKP, bool ONES = false, int NW = 4>
__global__ __launch_bounds__(64 * NW) void conv_wgrad_kernel(WgradArgs p) {
  constexpr int RBA = BMW * 2, RBB = BNW * 2;       // row bytes
  constexpr int CPRA = RBA / 16, CPRB = RBB / 16;   // 16-byte chunks per row
  constexpr int RPIA = 64 / CPRA, RPIB = 64 / CPRB; // rows per wave-instruction
  constexpr int NA = BKP / (NW * RPIA), NB = BKP / (NW * RPIB);
  constexpr int WCOLS = NW / 2;                     // waves side by side
  constexpr int WM = BMW / 2, WN = BNW / WCOLS;
  static_assert(NW == 4 || NW == 8, "four or eight waves");
  static_assert((NW * RPIA) % 4 == 0 && (NW * RPIB) % 4 == 0, "the staging swizzle repeats every instruction round");
  constexpr int MT = WM / 32, NT = WN / 32;
  constexpr int KS = BKP / 16;
  constexpr int STAGE = BKP * (RBA + RBB);
  constexpr int NST = BKP == 32 ? 3 : 2;            // LDS stages
  static_assert(NA >= 1 && NB >= 1, "tile too small");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* lds_taps = (int*)smem;
  char* lds_tiles = smem + MAX_TAPS * 4;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid / WCOLS, wc = wid % WCOLS;

  // 1-D grid ordered so that all output tiles of one pixel split share blockIdx % 8, i.e. (under the
  // round-robin block->XCD dispatch) one XCD and its L2: they re-read the same dY / X rows.
  //   id = ((split / 8) * tiles + tile) * 8 + split % 8
  const unsigned ntiles = (p.K + BNW - 1) / BNW;
  const unsigned mtiles = (p.Cout + BMW - 1) / BMW;
  const unsigned tiles = ntiles * mtiles;
  const unsigned grp = blockIdx.x / 8, lane8 = blockIdx.x % 8;
  const unsigned tile = grp % tiles;
  const int split = (int)((grp / tiles) * 8 + lane8);
  if (split >= p.splits) return;
  const unsigned tile_m = tile / ntiles, tile_n = tile % ntiles;
  const int co0 = tile_m * BMW, n0 = tile_n * BNW;

  if (tid < p.T) lds_taps[tid] = p.taps[tid];
  __syncthreads();

  const int kt_begin = split * p.ktiles_per_split;
  int kt_end = kt_begin + p.ktiles_per_split;
  const int kt_total = (p.M + BKP - 1) / BKP;
  if (kt_end > kt_total) kt_end = kt_total;

  // ---- A' (dY) staging rows ----
  const int a_row = lane / CPRA;                                 // row within instruction
  // logical chunk: tile row of instruction i is (i * NW + wid) * RPIA + a_row, and NW * RPIA rows repeat the swizzle
  const int a_chunk = (lane % CPRA) ^ wswz<RBA>(wid * RPIA + a_row);
  const int a_co = co0 + a_chunk * 8;
  const bool a_ok = a_co < p.Cout;

  // ---- B' (X gather) staging rows ----
  const int b_row = lane / CPRB;
  const int b_chunk = (lane % CPRB) ^ wswz<RBB>(wid * RPIB + b_row);
  const int b_n = n0 + b_chunk * 8;
  const bool b_kok = b_n < p.K;
  int b_tap = (int)__umulhi((unsigned)b_n, p.cin_magic);
  const int b_ci = b_n - b_tap * p.Cin;
  const int tp = b_kok ? lds_taps[b_tap] : 0;
  const int b_dy = tp >> 16, b_dx = (int)(short)(tp & 0xffff);

  const int opix = p.OH * p.OW;
  int boy[NB], box[NB], bpix[NB], bm[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    int m = kt_begin * BKP + (i * NW + wid) * RPIB + b_row;
    bm[i] = m;
    int b = m / opix;
    int rem = m - b * opix;
    boy[i] = rem / p.OW;
    box[i] = rem - boy[i] * p.OW;
    bpix[i] = b * p.IH * p.IW;
  }
  const int img_pix = p.IH * p.IW;

  auto stage = [&](int kt, int buf) {
    char* sa = lds_tiles + buf * STAGE;
    char* sb = sa + BKP * RBA;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int m = kt * BKP + (i * NW + wid) * RPIA + a_row;
      const bf16* src = (a_ok && m < p.M) ? p.dy + (size_t)m * p.Cout + a_co : p.zero;
      glds16_untracked(src, sa + (i * NW + wid) * 1024);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int iy = boy[i] * p.isy + b_dy, ix = box[i] * p.isx + b_dx;
      bool ok = b_kok && bm[i] < p.M && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
      const bf16* src = ok ? p.x + ((size_t)(bpix[i] + iy * p.IW + ix) * p.Cin + b_ci) : p.zero;
      glds16_untracked(src, sb + (i * NW + wid) * 1024);
      // advance this row by one k-tile (BKP pixels)
      bm[i] += BKP;
      box[i] += p.step_x;
      if (box[i] >= p.OW) { box[i] -= p.OW; boy[i] += 1; }
      boy[i] += p.step_y;
      if (boy[i] >= p.OH) { boy[i] -= p.OH; bpix[i] += img_pix; }
      bpix[i] += p.step_b * img_pix;
    }
  };

  // ---- transposed fragment read offsets ----
  // lane -> (k row within 16-step, column) of the 4x16 block it addresses
  const int tr_q = (lane & 15) >> 2;            // block row supplied by this lane
  const int tr_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int tr_k = 8 * (lane >> 5) + tr_q;      // + 4*r for the second read, + 16*ks
  int acol[MT], bcol[NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acol[mt] = wr * WM + mt * 32 + tr_col;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bcol[nt] = wc * WN + nt * 32 + tr_col;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[mt][nt][j] = 0.f;
  f32x16 ones_acc[ONES ? NT : 1];
  const bool sum_cols = ONES && tile_m == 0 && wr == 0;   // wave-uniform
  bf16x8 ones_frag;
  if (ONES) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int j = 0; j < 16; ++j) ones_acc[nt][j] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones_frag[j] = (bf16)1.0f;
  }

  // NST LDS stages: while k-tile kt is multiplied, the tiles up to kt+NST-1 are in flight (counted vmcnt,
  // raw barrier: __syncthreads() would drain every outstanding LDS-DMA load).  Fragments of k-step ks+1 are
  // read while the MFMAs of ks run; the first fragments of tile kt+1 are read behind the barrier, ahead of
  // the last MFMAs of tile kt.
  constexpr int LPS = NA + NB;
  static_assert(KS % 2 == 0, "fragment double-buffering assumes an even number of k-steps per tile");
  s16x8 af[2][MT], bfr[2][NT];
  auto load_frags = [&](int buf, int ks, int dst) {
    const char* sa = lds_tiles + buf * STAGE;
    const char* sb = sa + BKP * RBA;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int krow = ks * 16 + tr_k + 4 * r;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        int ch = (acol[mt] >> 3) ^ wswz<RBA>(krow);
        s16x4 v = lds_tr_read(sa + krow * RBA + ch * 16 + (acol[mt] & 7) * 2);
        af[dst][mt][4 * r + 0] = v[0]; af[dst][mt][4 * r + 1] = v[1];
        af[dst][mt][4 * r + 2] = v[2]; af[dst][mt][4 * r + 3] = v[3];
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        int ch = (bcol[nt] >> 3) ^ wswz<RBB>(krow);
        s16x4 v = lds_tr_read(sb + krow * RBB + ch * 16 + (bcol[nt] & 7) * 2);
        bfr[dst][nt][4 * r + 0] = v[0]; bfr[dst][nt][4 * r + 1] = v[1];
        bfr[dst][nt][4 * r + 2] = v[2]; bfr[dst][nt][4 * r + 3] = v[3];
      }
    }
  };
  auto mfma_step = [&](int src) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
            __builtin_bit_cast(bf16x8, af[src][mt]), __builtin_bit_cast(bf16x8, bfr[src][nt]), acc[mt][nt], 0, 0, 0);
    if (ONES && sum_cols) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        ones_acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones_frag, __builtin_bit_cast(bf16x8, bfr[src][nt]),
                                                              ones_acc[nt], 0, 0, 0);
    }
  };

  if (kt_begin < kt_end) {
    stage(kt_begin, 0);
    if (NST == 3 && kt_begin + 1 < kt_end) {
      stage(kt_begin + 1, 1);
      wait_vmcnt<LPS>();
    } else {
      wait_vmcnt<0>();
    }
    __syncthreads();
    load_frags(0, 0, 0);
    int cur = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      const int nxt = cur == NST - 1 ? 0 : cur + 1;
      // the slot requested here held tile kt-1, which every wave finished reading before the last barrier
      if (kt + NST - 1 < kt_end) stage(kt + NST - 1, NST == 2 ? nxt : (nxt == 2 ? 0 : nxt + 1));
#pragma unroll
      for (int ks = 0; ks + 1 < KS; ++ks) {
#if WGRAD_MFMA_FIRST
        // MFMAs of k-step ks, then the reads of ks+1 in their shadow (an MFMA group then waits only for
        // fragments requested a whole k-step earlier)
        mfma_step(ks & 1);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(cur, ks + 1, (ks + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#else
        load_frags(cur, ks + 1, (ks + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of the MFMAs
        mfma_step(ks & 1);
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
      if (kt + 1 < kt_end) {
        if (NST == 3 && kt + 2 < kt_end) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#if !WGRAD_MFMA_FIRST
        load_frags(nxt, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
      mfma_step((KS - 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#if WGRAD_MFMA_FIRST
      if (kt + 1 < kt_end) load_frags(nxt, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#endif
      cur = nxt;
    }
  }

  // ---- store the fp32 partial tile ----
  float* slab = p.slab + (size_t)split * p.slab_stride;
  const int colq = lane & 31, khalf = lane >> 5;
  if (ONES && sum_cols && khalf == 0) {
    // every row of the ones product holds the column sums; row 0 is element 0 of the lanes 0..31
    float* cs = slab + (size_t)p.Cout * p.K;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n0 + wc * WN + nt * 32 + colq;
      if (n < p.K) cs[n] = ones_acc[nt][0];
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      int n = n0 + wc * WN + nt * 32 + colq;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        int co = co0 + wr * WM + mt * 32 + (j & 3) + 8 * (j >> 2) + 4 * khalf;
        if (co < p.Cout && n < p.K) slab[(size_t)co * p.K + n] = acc[mt][nt][j];
      }
    }
}

// Sum the split slabs; write (or accumulate into) the gradient.  Block = 64
// float4 columns x 16 split lanes: lane sl sums splits sl, sl+16, ... and the 16
// partial sums are combined through LDS in lane order, so the result does not
// depend on scheduling.
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ slab, int splits, size_t n,
                                                            float* __restrict__ out_f32, bf16* __restrict__ out_bf16,
                                                            int accumulate) {
  __shared__ float4 red[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const size_t i = ((size_t)blockIdx.x * 64 + tx) * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n) {
    if (i + 4 <= n) {
      for (int k = ty; k < splits; k += 16) {
        float4 v = *(const float4*)(slab + (size_t)k * n + i);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    } else {
      float* sp = &s.x;
      for (int k = ty; k < splits; k += 16)
        for (size_t e = i; e < n; ++e) sp[e - i] += slab[(size_t)k * n + e];
    }
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && i < n) {
    for (int t = 1; t < 16; ++t) {
      float4 v = red[t][tx];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float* sp = &s.x;
    for (int e = 0; e < 4 && i + e < n; ++e) {
      float r = sp[e];
      if (out_f32) {
        if (accumulate) r += out_f32[i + e];
        out_f32[i + e] = r;
      }
      if (out_bf16) out_bf16[i + e] = (bf16)r;
    }
  }
}

template <int BMW, int BNW, int BKP, bool ONES = false, int NW = 4>
static int launch_wgrad(const WgradArgs& a, int splits, hipStream_t stream) {
  size_t lds = MAX_TAPS * 4 + (BKP == 32 ? 3 : 2) * (size_t)BKP * (BMW * 2 + BNW * 2);
  unsigned mt = (a.Cout + BMW - 1) / BMW, nt = (a.K + BNW - 1) / BNW;
  // the dynamic-LDS limit is a per-device function attribute: remember it per device, not per process
  static bool attr_set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev = (dev >= 0 && dev < 64) ? dev : 0;
  if (!attr_set[dev] && lds > 48 * 1024) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<BMW, BNW, BKP, ONES, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set[dev] = true;
  }
  unsigned split_groups = (splits + 7) / 8;
  hipLaunchKernelGGL((conv_wgrad_kernel<BMW, BNW, BKP, ONES, NW>), dim3(mt * nt * split_groups * 8), dim3(64 * NW), lds, stream, a);
  return (int)hipGetLastError();
}

extern "C" int dtfa_conv_wgrad(WgradArgs* a, int bmw, int bnw, int bkp, int splits, hipStream_t stream) {
  if (a->Cin % 8 != 0 || a->Cout % 8 != 0 || a->T > MAX_TAPS || a->T < 1) return -1;
  if (a->K != a->T * a->Cin || a->K >= (1 << 16) || splits < 1) return -2;
  int kt_total = (a->M + bkp - 1) / bkp;
  a->ktiles_per_split = (kt_total + splits - 1) / splits;
  a->splits = splits;
  int opix = a->OH * a->OW;
  a->step_b = bkp / opix;
  int rem = bkp % opix;
  a->step_y = rem / a->OW;
  a->step_x = rem % a->OW;
  a->slab_stride = (long long)a->Cout * a->K + (a->colsum ? a->K : 0);
  if (a->colsum) {
    // column sums ride on the tiles used for x^T x of 64 .. 512 channels
#define CASE1(A_, B_, C_) \
  if (bmw == A_ && bnw == B_ && bkp == C_) return launch_wgrad<A_, B_, C_, true>(*a, splits, stream);
    CASE1(256, 128, 32)
    CASE1(128, 128, 32)
    CASE1(128, 128, 64)
    CASE1(64, 64, 32)
    CASE1(64, 64, 64)
#undef CASE1
    return -4;
  }
#define CASE(A_, B_, C_) \
  if (bmw == A_ && bnw == B_ && bkp == C_) return launch_wgrad<A_, B_, C_>(*a, splits, stream);
  if (bmw == 256 && bnw == 256 && bkp == 32) return launch_wgrad<256, 256, 32, false, 8>(*a, splits, stream);
  if (bmw == 256 && bnw == 256 && bkp == 64) return launch_wgrad<256, 256, 64, false, 8>(*a, splits, stream);
  CASE(256, 128, 32)
  CASE(128, 256, 32)
  CASE(64, 256, 32)
  CASE(128, 128, 32)
  CASE(64, 128, 32)
  CASE(128, 64, 32)
  CASE(64, 64, 32)
  CASE(128, 128, 64)
  CASE(64, 128, 64)
  CASE(128, 64, 64)
  CASE(64, 64, 64)
#undef CASE
  return -3;
}

extern "C" int dtfa_wgrad_reduce(const float* slab, int splits, long long n, float* out_f32, void* out_bf16,
                                 int accumulate, hipStream_t stream) {
  if (n <= 0) return 0;
  size_t vecs = ((size_t)n + 3) / 4;
  unsigned blocks = (unsigned)((vecs + 63) / 64);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(1024), 0, stream, slab, splits, (size_t)n, out_f32,
                     (bf16*)out_bf16, accumulate);
  return (int)hipGetLastError();
}
