// Implicit-GEMM convolution for gfx950 (MI355X): NHWC bf16 activations,
// [Cout][tap][Cin] bf16 weights, fp32 MFMA accumulation.
//
// One kernel serves forward convolution, data-gradient (run as a convolution
// of dY with a permuted weight matrix, one launch per output-stride phase) and
// dense layers (1x1 convolution over a [B,1,1,C] tensor).  The geometry is
// described by a *tap table*: tap t reads the input pixel
//     (oy*isy + dy[t], ox*isx + dx[t])
// for logical output pixel (oy, ox), and the result is written to physical
// output pixel (oy*osy + oy0, ox*osx + ox0).  Out-of-image taps read zeros.
//
// GEMM view:  D[m][n] = sum_k A[m][k] * W[n][k]
//     m = (b, oy, ox)      rows   -> MFMA A operand (pixels)
//     n = cout             cols   -> MFMA B operand (weights)
//     k = (tap, cin)       reduction, Cin % 8 == 0 so a 16-byte chunk never
//                          straddles a tap
//
// Structure: 256 threads = 4 waves in a 2x2 grid, each wave owns a
// (BM/2)x(BN/2) tile made of 32x32 MFMA blocks (v_mfma_f32_32x32x16_bf16).
// Operand tiles go global -> LDS with LDS-DMA (global_load_lds_dwordx4); the
// im2col gather is expressed in the per-lane *source* address, padding is a
// read from a zero page, and the LDS image is XOR-swizzled through the source
// chunk choice so that ds_read_b128 fragment reads are bank-conflict free.
// Three LDS stages: while k-tile t is multiplied, tile t+1 is landing and
// tile t+2 is being requested.  The epilogue optionally adds a bias, applies ReLU, emits
// per-channel sum / sum-of-squares partials for a following batch-norm, and
// stores bf16 rows through LDS so that global stores are 16 B per lane.
#include "common.h"

struct ConvArgs {
  const bf16* x;
  const bf16* w;
  bf16* y;
  const float* bias;   // [Cout] or null
  const bf16* addend;  // optional tensor of the output's shape, added in the epilogue (residual gradient; with the
                       // bias / ReLU epilogue: the residual of a block, added before the ReLU)
  const unsigned char* addend_mask;  // optional: one bit per element of addend, only set elements are added
  float* stats;        // [mtiles][2][Cout] partial (sum, sumsq) or null
  const bf16* zero;    // >= 16 bytes of zeros, 16-byte aligned
  const int* taps;     // T entries, (dy << 16) | (dx & 0xffff)
  int IH, IW, Cin;
  int OH, OW, Cout;
  int OYn, OXn;
  int osy, osx, oy0, ox0;
  int isy, isx;
  int T, M, K;
  unsigned cin_magic;  // ceil(2^32 / Cin)
  int relu;
  int dense_out;       // output pixel index == m (no stride/offset mapping)
  int pointwise;       // 1x1, stride 1, no padding, dense output: input pixel index == m
  // Data-gradient launches whose output is the incoming gradient dz of a batch-norm layer can emit that
  // layer's backward partial sums (sum g, sum g*xhat; g = dz gated by the ReLU sign, xhat = (y-mean)*invstd)
  // into ``stats`` while the output tile is still on chip:
  const bf16* bn_y;              // the batch-norm input (shape of the output), or null
  const float* bn_mean;          // [Cout]
  const float* bn_invstd;        // [Cout]
  const float* bn_scale;         // [Cout]  (bn_relu == 2: sign recomputed from y*scale+shift)
  const float* bn_shift;         // [Cout]
  const unsigned char* bn_mask;  // one bit per output element (bn_relu == 3)
  int bn_relu;                   // 0: no gating, 2: recompute, 3: stored bits
  // bias / ReLU epilogue only (inference with folded batch-norm, dense layers):
  int ldo;                       // channel pitch of the output rows (> Cout: y is a channel slice of a wider tensor)
  const float* oscale;           // [Cout] multiplier applied to the accumulator before the bias, or null
};

#define MAX_TAPS 64

template <int BK>
__device__ __forceinline__ int swz_of_row(int row) {
  constexpr int S = BK / 8;
  constexpr int R = 256 / (BK * 2);
  return (row / R) & (S - 1);
}

// EPI selects the epilogue at compile time: 0 = plain store (data-gradient, no statistics),
// 1 = batch-norm partial statistics, 2 = bias (+ optional ReLU), 3 = data-gradient that also emits the
// batch-norm backward partial sums of the layer its output flows into.
// Epilogue shared by the kernels below.  ``acc`` is the wave's MT x NT grid of 32x32 accumulators for the
// tile at (m0, n0); ``w4`` is the wave's index inside its 2x2 group and ``lds_tiles`` that group's LDS
// region (at least 4 * (BM/2) * EP_PITCH + BN*8 bytes), free for reuse.  ``valid`` is false for a group
// that has no tile (it still takes part in the barriers).  Bias / ReLU / batch-norm statistics / bf16 rows
// staged through LDS so that global stores are 16 B per lane.
#ifndef CONV_NT
#define CONV_NT 0
#endif
// epilogue traffic hints (CONV_NT: bit 0 = non-temporal output stores, bit 1 = non-temporal operand loads)
__device__ __forceinline__ void out_store(bf16x8* p, bf16x8 v) {
#if (CONV_NT & 1)
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
__device__ __forceinline__ bf16x8 epi_load(const bf16x8* p) {
#if (CONV_NT & 2)
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}

template <int BM, int BN, int EPI>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x16 (&acc)[BM / 64][BN / 64], char* lds_tiles,
                                              int w4, int lane, int m0, int n0, unsigned tile_m, bool valid) {
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int MT = WM / 32, NT = WN / 32;
  constexpr int EP_PITCH = WN * 2 + 16;
  const int wr = w4 >> 1, wc = w4 & 1;
  const int wid = w4;
  const int khalf = lane >> 5;
  const int opix = p.OYn * p.OXn;
  const int ldo = EPI == 2 ? p.ldo : p.Cout;  // only the bias / ReLU epilogue writes into channel slices
  // ---- epilogue: bias / relu / bn partial statistics / bf16 rows via LDS ----
  // Row readback geometry: a wave-instruction covers RPP rows x CPR 16-byte chunks; lane -> (row, chunk)
  // is the same in every pass, so a lane keeps one chunk column (8 channels).
  // CPR (chunk slots per row) is the row's chunk count rounded up to a power of two, so that lane -> (row,
  // chunk) is a shift / mask; for WN = 96 four of the 16 slots are idle.
  constexpr int CHUNKS = WN / 8;      // 16-byte chunks per row
  constexpr int CPR = CHUNKS <= 4 ? 4 : (CHUNKS <= 8 ? 8 : 16);
  constexpr int RPP = 64 / CPR;       // rows per pass
  constexpr int NP = WM / RPP;        // passes
  const bool chunk_live = (lane % CPR) < CHUNKS;
  const int chunk_co = n0 + wc * WN + (lane % CPR) * 8;
  size_t last_pixel = 0;  // physical output pixel of the latest out_offset() call
  auto out_offset = [&](int pass, bool& ok) -> size_t {
    const int m = m0 + wr * WM + pass * RPP + lane / CPR;
    ok = m < p.M && chunk_co < p.Cout && chunk_live;
    size_t opixel = 0;
    if (ok) {
      if (p.dense_out) {
        opixel = (size_t)m;
      } else {
        int b = m / opix;
        int rem = m - b * opix;
        int oy = rem / p.OXn;
        int ox = rem - oy * p.OXn;
        opixel = ((size_t)b * p.OH + (oy * p.osy + p.oy0)) * p.OW + (ox * p.osx + p.ox0);
      }
    }
    last_pixel = opixel;
    return opixel * ldo + chunk_co;
  };
  char* ep = lds_tiles + wid * (WM * EP_PITCH);
  float* xch = (float*)(lds_tiles + 4 * WM * EP_PITCH);  // [2 wave columns][WN][2] exchange area behind the row tiles
  float st1[NT], st2[NT];
  const int colq = lane & 31;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = n0 + wc * WN + nt * 32 + colq;
    float bias = 0.f, osc = 1.f;
    if (EPI == 2 && p.bias != nullptr && co < p.Cout) bias = p.bias[co];
    if (EPI == 2 && p.oscale != nullptr && co < p.Cout) osc = p.oscale[co];
    const bool relu = EPI == 2 && p.relu != 0 && p.addend == nullptr;  // with a residual the ReLU follows the add
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float v = acc[mt][nt][j];
        if (EPI == 2) {
          v = v * osc + bias;
          if (relu) v = fmaxf(v, 0.f);
        }
        if (EPI == 1) {  // statistics of the fp32 accumulators (rounding to bf16 is unbiased)
          s1 += v;
          s2 += v * v;
        }
        int r = mt * 32 + (j & 3) + 8 * (j >> 2) + 4 * khalf;
        *(bf16*)(ep + r * EP_PITCH + (nt * 32 + colq) * 2) = (bf16)v;
      }
    }
    if (EPI == 1) {
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      st1[nt] = s1;
      st2[nt] = s2;
      // the lower wave row parks its sums; the upper one adds them after the barrier (one row per tile)
      if (wr == 1 && khalf == 0) {
        xch[((wc * NT + nt) * 32 + colq) * 2] = s1;
        xch[((wc * NT + nt) * 32 + colq) * 2 + 1] = s2;
      }
    }
  }
  __syncthreads();
  if (EPI == 1 && wr == 0 && khalf == 0 && valid) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = n0 + wc * WN + nt * 32 + colq;
      if (co < p.Cout) {
        const size_t prow = (size_t)tile_m * 2;
        p.stats[prow * p.Cout + co] = st1[nt] + xch[((wc * NT + nt) * 32 + colq) * 2];
        p.stats[(prow + 1) * p.Cout + co] = st2[nt] + xch[((wc * NT + nt) * 32 + colq) * 2 + 1];
      }
    }
  }

  float bs1[8], bs2[8], bmu[8], bis[8], bsc[8], bsh[8];
  if (EPI == 3) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = chunk_co + j < p.Cout && chunk_live;
      bs1[j] = bs2[j] = 0.f;
      bmu[j] = ok ? p.bn_mean[chunk_co + j] : 0.f;
      bis[j] = ok ? p.bn_invstd[chunk_co + j] : 0.f;
      bsc[j] = (ok && p.bn_relu == 2) ? p.bn_scale[chunk_co + j] : 0.f;
      bsh[j] = (ok && p.bn_relu == 2) ? p.bn_shift[chunk_co + j] : 0.f;
    }
  }
  if (EPI == 1 || EPI == 2) {
    // forward: plain coalesced row stores
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
      bool ok;
      const size_t off = out_offset(pass, ok);
      if (ok) {
        const int r = pass * RPP + lane / CPR;
        bf16x8 v = *(const bf16x8*)(ep + r * EP_PITCH + (lane % CPR) * 16);
        if (EPI == 2 && p.addend != nullptr) {
          // inference residual block: act(conv * scale + shift + residual); the residual is dense ([.., Cout])
          const bf16x8 res = epi_load((const bf16x8*)(p.addend + last_pixel * p.Cout + chunk_co));
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float f = (float)v[j] + (float)res[j];
            v[j] = (bf16)(p.relu ? fmaxf(f, 0.f) : f);
          }
        }
        out_store((bf16x8*)(p.y + off), v);
      }
    }
  } else {
    // data gradient: the readback may add a residual gradient (optionally gated by a bitmask) and fold
    // the batch-norm backward sums.  Passes are handled in groups: all global loads of a group are
    // requested before any is consumed, so a tile pays the memory latency once per group, not per pass.
    constexpr int G = NP >= 4 ? 4 : NP;
    const bool has_add = p.addend != nullptr;
    const bool has_addm = p.addend_mask != nullptr;
#pragma unroll
    for (int g0 = 0; g0 < NP; g0 += G) {
      size_t ooff[G];
      bool ook[G];
      bf16x8 addv[G], bnyv[G];
      unsigned addm[G], bnm[G];
#pragma unroll
      for (int i = 0; i < G; ++i) {
        ooff[i] = out_offset(g0 + i, ook[i]);
        const size_t off = ook[i] ? ooff[i] : 0;  // idle lanes read an in-range dummy
        if (has_add) {
          addv[i] = epi_load((const bf16x8*)(p.addend + off));
          addm[i] = has_addm ? (unsigned)p.addend_mask[off >> 3] : 0xffu;
        }
        if (EPI == 3) {
          bnyv[i] = epi_load((const bf16x8*)(p.bn_y + off));
          bnm[i] = p.bn_relu == 3 ? (unsigned)p.bn_mask[off >> 3] : 0xffu;
        }
      }
#pragma unroll
      for (int i = 0; i < G; ++i) {
        if (!ook[i]) continue;
        const int r = (g0 + i) * RPP + lane / CPR;
        bf16x8 v = *(const bf16x8*)(ep + r * EP_PITCH + (lane % CPR) * 16);
        if (has_add) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (bf16)((float)v[j] + (((addm[i] >> j) & 1u) ? (float)addv[i][j] : 0.f));
        }
        out_store((bf16x8*)(p.y + ooff[i]), v);
        if (EPI == 3) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float yf = (float)bnyv[i][j];
            float g = (float)v[j];  // the rounded value, the one the elementwise backward pass will read
            if (p.bn_relu == 2 && !(yf * bsc[j] + bsh[j] > 0.f)) g = 0.f;
            if (!((bnm[i] >> j) & 1u)) g = 0.f;
            bs1[j] += g;
            bs2[j] += g * ((yf - bmu[j]) * bis[j]);
          }
        }
      }
    }
  }
  if (EPI == 3) {
    // lanes that share a chunk column differ in the lane bits above log2(CPR): fold those
#pragma unroll
    for (int sh = CPR; sh < 64; sh <<= 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        bs1[j] += __shfl_xor(bs1[j], sh, 64);
        bs2[j] += __shfl_xor(bs2[j], sh, 64);
      }
    }
    const int co = chunk_co;
    float* mine = xch + (wc * WN + (lane % CPR) * 8) * 2;
    if (wr == 1 && lane < CHUNKS) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        mine[j * 2] = bs1[j];
        mine[j * 2 + 1] = bs2[j];
      }
    }
    __syncthreads();
    if (wr == 0 && lane < CHUNKS && co < p.Cout && valid) {
      const size_t prow = (size_t)tile_m * 2;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        p.stats[prow * p.Cout + co + j] = bs1[j] + mine[j * 2];
        p.stats[(prow + 1) * p.Cout + co + j] = bs2[j] + mine[j * 2 + 1];
      }
    }
  }
}

// Timing-only builds for attributing main-loop time (results are wrong): -DDTFA_DBG=1 drops the global
// loads of the main loop, =2 replaces the MFMAs by one scalar FMA per fragment pair, =3 both.
#ifndef DTFA_DBG
#define DTFA_DBG 0
#endif
// main-loop order: 0 = fragment reads of k-step ks+1, then the MFMAs of ks; 1 = MFMAs first, reads behind the
// group; 2 = one read behind each MFMA of the group
#ifndef CONV_INTERLEAVE
#define CONV_INTERLEAVE 2
#endif
// workgroups per CU the narrow (<= 128 x 64) tiles are compiled for
#ifndef CONV_NARROW_OCC
#define CONV_NARROW_OCC 3
#endif

// WG = 2 doubles the workgroup: 512 threads as two groups of 2 x 2 waves, each group computing its own
// BM-row tile of a (2 * BM) x BN macro tile against ONE shared BN-column weight tile.  The kernel moves
// (WG * BM + BN) * BK operand bytes from L2 into LDS per WG * BM * BN * BK multiply-adds; that path (measured
// ~12 TB/s chip-wide with ~90 % L2 hits) is what bounds the compute-heavy layers, and sharing the weight tile
// takes a quarter of the bytes per FLOP away at 128 x 128.
template <int BM, int BN, int BK, int EPI, int NST, int WG>
__global__ __launch_bounds__(256 * WG, (WG == 2 ? (BN > 128 ? 1 : 2) : (BM * BN > 128 * 128 ? 2 : (BM * BN <= 128 * 64 ? CONV_NARROW_OCC : 3)))) void conv_igemm_kernel(ConvArgs p) {
  constexpr int S = BK / 8;            // 16-byte slots per LDS row
  constexpr int ROWB = BK * 2;         // bytes per LDS row
  constexpr int RPI = 1024 / ROWB;     // rows covered by one wave-instruction
  constexpr int NA = BM * ROWB / 4096; // LDS-DMA instructions per thread (A)
  constexpr int NB = BN * ROWB / 4096;
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int MT = WM / 32, NT = WN / 32;
  constexpr int KS = BK / 16;
  constexpr int STAGE = (WG * BM + BN) * ROWB;   // [A of group 0][A of group 1][B]
  constexpr int NBW = NB / WG;                   // weight-tile loads per wave
  constexpr int EP = 4 * (BM / 2) * ((BN / 2) * 2 + 16) + 2 * (BN / 2) * 2 * 4;  // epilogue bytes per group
  static_assert(NA >= 1 && NBW >= 1 && NB % WG == 0, "tile too small for this staging");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* lds_taps = (int*)smem;
  char* lds_tiles = smem + MAX_TAPS * 4;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = WG == 1 ? 0 : wid_all >> 2;   // wave group
  const int wid = wid_all & 3;    // wave inside its 2 x 2 group
  const int wr = wid >> 1, wc = wid & 1;

  const unsigned mtiles = (p.M + BM - 1) / BM;
  const unsigned mgroups = (mtiles + WG - 1) / WG;
  const unsigned ntiles = (p.Cout + BN - 1) / BN;
  const unsigned bid = xcd_remap(blockIdx.x, mgroups * ntiles);
  const unsigned tile_m = (bid / ntiles) * WG + grp, tile_n = bid % ntiles;
  const bool valid = tile_m < mtiles;   // the last macro tile may have a group without rows
  const int m0 = valid ? (int)(tile_m * BM) : p.M, n0 = tile_n * BN;

  // (a pointwise convolution never looks at the tap table: its workgroups skip this global round trip and
  // the barrier that publishes it)
  const bool pointwise = p.pointwise != 0;
  if (!pointwise && tid < p.T) lds_taps[tid] = p.taps[tid];

  // ---- per-thread staging rows (fixed across the K loop) ----
  const int slot = lane % S;
  const int row_in_inst = lane / S;
  const int cslot = slot ^ swz_of_row<BK>(wid * RPI + row_in_inst);  // same for every i

  const bf16* abase[NA];
  int aiy[NA], aix[NA];
  const int opix = p.OYn * p.OXn;
  // 1x1 / stride-1 / unpadded: input pixel == output pixel == m, no decode and no bounds checks
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    int m = m0 + (i * 4 + wid) * RPI + row_in_inst;
    if (pointwise) {
      aiy[i] = m < p.M ? 0 : -(1 << 24);
      aix[i] = 0;
      abase[i] = p.x + (size_t)(m < p.M ? m : 0) * p.Cin;
    } else if (m < p.M) {
      int b = m / opix;
      int rem = m - b * opix;
      int oy = rem / p.OXn;
      int ox = rem - oy * p.OXn;
      aiy[i] = oy * p.isy;
      aix[i] = ox * p.isx;
      abase[i] = p.x + (size_t)b * p.IH * p.IW * p.Cin;
    } else {
      aiy[i] = -(1 << 24);
      aix[i] = 0;
      abase[i] = p.x;
    }
  }
  // the weight tile's loads are dealt over all waves of the workgroup (the swizzle of a row repeats every
  // 16 rows, so wave wid_all stages with the same column slot as wave wid)
  const bf16* bbase[NBW];
  bool bok[NBW];
#pragma unroll
  for (int i = 0; i < NBW; ++i) {
    int n = n0 + (i * 4 * WG + wid_all) * RPI + row_in_inst;
    bok[i] = n < p.Cout;
    bbase[i] = p.w + (size_t)(bok[i] ? n : 0) * p.K;
  }

  if (!pointwise) __syncthreads();  // tap table visible

  auto stage = [&](int kt, int buf) {
    char* sa = lds_tiles + buf * STAGE + grp * (BM * ROWB);
    char* sb = lds_tiles + buf * STAGE + WG * BM * ROWB;
    const int kc8 = kt * BK + cslot * 8;
    const bool kvalid = kc8 < p.K;
    if (pointwise) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const bf16* src = (kvalid && aiy[i] == 0) ? abase[i] + kc8 : p.zero;
        glds16(src, sa + (i * 4 + wid) * 1024);
      }
    } else {
      const int tap = (int)__umulhi((unsigned)kc8, p.cin_magic);
      const int ci = kc8 - tap * p.Cin;
      const int tp = kvalid ? lds_taps[tap] : 0;
      const int dy = tp >> 16;
      const int dx = (int)(short)(tp & 0xffff);
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        int iy = aiy[i] + dy, ix = aix[i] + dx;
        bool ok = kvalid && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
        // element offset inside the image with two full-rate 24-bit multiplies (the launcher checks
        // IH * IW < 2^24 and IH * IW * Cin < 2^32); the 64-bit form costs two quarter-rate multiplies per row
        const unsigned off = __umul24(__umul24((unsigned)iy, (unsigned)p.IW) + (unsigned)ix, (unsigned)p.Cin) + (unsigned)ci;
        const bf16* src = ok ? abase[i] + off : p.zero;
        glds16(src, sa + (i * 4 + wid) * 1024);
      }
    }
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
      const bf16* src = (kvalid && bok[i]) ? bbase[i] + kc8 : p.zero;
      glds16(src, sb + (i * 4 * WG + wid_all) * 1024);
    }
  };

  // ---- fragment read offsets ----
  int aoff[MT], aswz[MT], boff[NT], bswz[NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int row = wr * WM + mt * 32 + (lane & 31);
    aoff[mt] = grp * (BM * ROWB) + row * ROWB;
    aswz[mt] = swz_of_row<BK>(row);
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int row = wc * WN + nt * 32 + (lane & 31);
    boff[nt] = WG * BM * ROWB + row * ROWB;
    bswz[nt] = swz_of_row<BK>(row);
  }
  const int khalf = lane >> 5;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[mt][nt][j] = 0.f;

  const int KT = (p.K + BK - 1) / BK;
  constexpr int LPS = NA + NBW;  // LDS-DMA loads per thread and stage
  static_assert(KS % 2 == 0, "fragment double-buffering assumes an even number of k-steps per tile");
  // NST LDS stages (2 or 3): while k-tile kt is multiplied, the tiles up to kt+NST-1 are in flight.
  // Within a tile the fragments of k-step ks+1 are read from LDS while the MFMAs of k-step ks run, and
  // the first fragments of tile kt+1 are read behind the barrier, ahead of the last MFMAs of tile kt.
  static_assert(NST == 2 || NST == 3, "two or three LDS stages");
  stage(0, 0);
  if (NST == 3 && KT > 1) {
    stage(1, 1);
    wait_vmcnt<LPS>();
  } else {
    wait_vmcnt<0>();
  }
  __syncthreads();

  bf16x8 af[2][MT], bfr[2][NT];
#if (DTFA_DBG & 4)
  bool frag_skip = false;
#endif
  // A wave whose 32/64 output columns all lie beyond Cout (a 64-wide tile on a 32-channel output, the
  // 32-column tail of 160 channels) takes part in the loads and barriers but skips its fragment reads and
  // MFMAs, leaving its SIMD's matrix pipe to the other workgroups on the CU.
  const bool wave_live = n0 + wc * WN < p.Cout;
  auto read_frags = [&](int buf, int ks, int dst) {
    if (!wave_live) return;
#if (DTFA_DBG & 4)
    if (frag_skip) return;
#endif
    const char* base = lds_tiles + buf * STAGE;
    const int sl = ks * 2 + khalf;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af[dst][mt] = *(const bf16x8*)(base + aoff[mt] + ((sl ^ aswz[mt]) << 4));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bfr[dst][nt] = *(const bf16x8*)(base + boff[nt] + ((sl ^ bswz[nt]) << 4));
  };
  auto mfma_step = [&](int src) {
    if (!wave_live) return;
#if (DTFA_DBG & 2)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt][0] += (float)af[src][mt][0] * (float)bfr[src][nt][0];
#else
    __builtin_amdgcn_s_setprio(1);  // a wave that has its operands keeps the MFMA pipe ahead of waves issuing loads
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[src][mt], bfr[src][nt], acc[mt][nt], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
#endif
  };

  // MFMAs of fragment set ``src`` with the reads of (buf, ks) into set ``dst`` issued one behind each MFMA
  auto mfma_read_step = [&](int src, int buf, int ks, int dst, bool do_read) {
    if (!wave_live) return;
#if (DTFA_DBG & 6)
    mfma_step(src);
    if (do_read) read_frags(buf, ks, dst);
#else
    const char* base = lds_tiles + buf * STAGE;
    const int sl = ks * 2 + khalf;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[src][mt], bfr[src][nt], acc[mt][nt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const int i = mt * NT + nt;
        if (do_read) {
          if (i < MT) af[dst][i] = *(const bf16x8*)(base + aoff[i] + ((sl ^ aswz[i]) << 4));
          else if (i < MT + NT) bfr[dst][i - MT] = *(const bf16x8*)(base + boff[i - MT] + ((sl ^ bswz[i - MT]) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    // tiles with fewer MFMAs than fragments (a single MFMA per k-step): the remaining reads
    if (do_read) {
#pragma unroll
      for (int i = MT * NT; i < MT + NT; ++i) {
        if (i < MT) af[dst][i] = *(const bf16x8*)(base + aoff[i] + ((sl ^ aswz[i]) << 4));
        else bfr[dst][i - MT] = *(const bf16x8*)(base + boff[i - MT] + ((sl ^ bswz[i - MT]) << 4));
      }
    }
    __builtin_amdgcn_s_setprio(0);
#endif
  };

  read_frags(0, 0, 0);
  int cur = 0;  // kt % NST
  for (int kt = 0; kt < KT; ++kt) {
    const int nxt = cur == NST - 1 ? 0 : cur + 1;
    // the slot requested here held tile kt-1, which every wave finished reading before the last barrier
    if (kt + NST - 1 < KT && !(DTFA_DBG & 1)) stage(kt + NST - 1, NST == 2 ? nxt : (nxt == 2 ? 0 : nxt + 1));
#pragma unroll
    for (int ks = 0; ks + 1 < KS; ++ks) {
#if CONV_INTERLEAVE
      // MFMAs of k-step ks first, then the LDS reads of k-step ks+1, which complete in the shadow of those
      // MFMAs: an MFMA group waits only for fragments requested a whole k-step earlier.  With the reads issued
      // BEFORE the MFMAs, the compiler's lgkmcnt(0) in front of the group waited for the fresh reads as well.
#if CONV_INTERLEAVE == 2
      mfma_read_step(ks & 1, cur, ks + 1, (ks + 1) & 1, true);
      __builtin_amdgcn_sched_barrier(0);
#else
      mfma_step(ks & 1);
      __builtin_amdgcn_sched_barrier(0);
      read_frags(cur, ks + 1, (ks + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#endif
#else
      read_frags(cur, ks + 1, (ks + 1) & 1);
      // keep the reads ahead of the MFMAs: left alone the scheduler sinks them and reuses one register set
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(ks & 1);
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    if (kt + 1 < KT) {
      // this wave's reads of tile kt are complete and everything but the stage just requested has landed;
      // after the barrier tile kt+1 is visible to all waves and tile kt's slot may be overwritten
      // (a raw barrier: __syncthreads() would add a fence that drains every outstanding LDS-DMA load)
      if (NST == 3 && kt + 2 < KT) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#if !CONV_INTERLEAVE
      read_frags(nxt, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
#if CONV_INTERLEAVE == 2
    mfma_read_step((KS - 1) & 1, nxt, 0, 0, kt + 1 < KT);
    __builtin_amdgcn_sched_barrier(0);
#else
    mfma_step((KS - 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
#if CONV_INTERLEAVE
    // the first fragments of tile kt+1, requested in the shadow of the MFMAs just issued
    if (kt + 1 < KT) read_frags(nxt, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#endif
#endif
    cur = nxt;
#if (DTFA_DBG & 4)
    frag_skip = true;
#endif
  }
  __syncthreads();  // all fragment reads done before the tiles are reused by the epilogue

#if (DTFA_DBG & 8)
  {
    float sum = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 16; ++j) sum += acc[mt][nt][j];
    if (sum == 123.456f) p.y[0] = (bf16)sum;
    return;
  }
#endif
  // each group stages its output through its own part of the (now idle) tile area
  conv_epilogue<BM, BN, EPI>(p, acc, lds_tiles + grp * EP, wid, lane, m0, n0, tile_m, valid);
}

template <int BM, int BN, int BK, int EPI, int NST, int WG>
static int launch_conv(const ConvArgs& a, hipStream_t stream) {
  constexpr int ROWB = BK * 2;
  constexpr int STAGE = (WG * BM + BN) * ROWB;
  constexpr int EP = WG * (4 * (BM / 2) * ((BN / 2) * 2 + 16) + 2 * (BN / 2) * 2 * 4);  // row tiles + statistics exchange
  size_t lds = MAX_TAPS * 4 + (size_t)((NST * STAGE > EP) ? NST * STAGE : EP);
  unsigned mtiles = (a.M + BM - 1) / BM, ntiles = (a.Cout + BN - 1) / BN;
  unsigned mgroups = (mtiles + WG - 1) / WG;
  // the dynamic-LDS limit is a per-device function attribute: remember it per device, not per process
  static bool attr_set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev = (dev >= 0 && dev < 64) ? dev : 0;
  if (!attr_set[dev] && lds > 48 * 1024) {
    (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<BM, BN, BK, EPI, NST, WG>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, EPI, NST, WG>), dim3(mgroups * ntiles), dim3(256 * WG), lds,
                     stream, a);
  return (int)hipGetLastError();
}

template <int BM, int BN, int BK, int NST, int WG = 1>
static int launch_conv_epi(const ConvArgs& a, hipStream_t stream) {
  if (a.bn_y != nullptr) return launch_conv<BM, BN, BK, 3, NST, WG>(a, stream);
  if (a.stats != nullptr) return launch_conv<BM, BN, BK, 1, NST, WG>(a, stream);
  if (a.bias != nullptr || a.relu || a.oscale != nullptr || a.ldo != a.Cout)
    return launch_conv<BM, BN, BK, 2, NST, WG>(a, stream);
  return launch_conv<BM, BN, BK, 0, NST, WG>(a, stream);
}

// Number of partial-statistics rows the kernel writes for a given M and BM (one per row of tiles; a 256-row
// macro tile is two 128-row tiles).
extern "C" int dtfa_conv_stats_rows(int M, int bm) {
  if (bm == 256) bm = 128;
  return (M + bm - 1) / bm;
}

extern "C" int dtfa_conv_igemm(const ConvArgs* a, int bm, int bn, int bk, int nst, hipStream_t stream) {
  if (a->Cin % 8 != 0 || a->Cout % 8 != 0 || a->T > MAX_TAPS || a->T < 1) return -1;
  if (a->K != a->T * a->Cin || a->K >= (1 << 16)) return -2;
  if ((long long)a->IH * a->IW >= (1ll << 24) || (long long)a->IH * a->IW * a->Cin >= (1ll << 32)) return -2;
  if (a->stats != nullptr && a->bias != nullptr) return -4;
  // a residual in the bias / ReLU (forward) epilogue: no bitmask, no statistics
  const bool fwd_residual = a->addend != nullptr && (a->bias != nullptr || a->relu || a->oscale != nullptr);
  if (fwd_residual && (a->addend_mask != nullptr || a->stats != nullptr || a->bn_y != nullptr)) return -4;
  if (a->addend != nullptr && !fwd_residual && a->stats != nullptr && a->bn_y == nullptr) return -4;
  if (a->bn_y != nullptr) {
    if (a->stats == nullptr || a->bn_mean == nullptr || a->bn_invstd == nullptr || a->bias != nullptr || a->relu) return -5;
    if (a->bn_relu == 2 && (a->bn_scale == nullptr || a->bn_shift == nullptr)) return -5;
    if (a->bn_relu == 3 && a->bn_mask == nullptr) return -5;
    if (a->bn_relu != 0 && a->bn_relu != 2 && a->bn_relu != 3) return -5;
  }
  if (a->ldo < a->Cout || a->ldo % 8 != 0) return -6;
  if ((a->oscale != nullptr || a->ldo != a->Cout) && (a->stats != nullptr || a->bn_y != nullptr)) return -6;
  if (a->ldo != a->Cout && a->addend != nullptr && !fwd_residual) return -6;
  if (a->M <= 0) return 0;
  // bm == 256: two wave groups of 128 rows sharing one weight tile
  if (bm == 256 && bn == 128 && bk == 32 && nst == 3) return launch_conv_epi<128, 128, 32, 3, 2>(*a, stream);
  if (bm == 256 && bn == 128 && bk == 32 && nst == 2) return launch_conv_epi<128, 128, 32, 2, 2>(*a, stream);
  if (bm == 256 && bn == 128 && bk == 64 && nst == 2) return launch_conv_epi<128, 128, 64, 2, 2>(*a, stream);
  if (bm == 256 && bn == 256 && bk == 32 && nst == 3) return launch_conv_epi<128, 256, 32, 3, 2>(*a, stream);
  if (bm == 256 && bn == 256 && bk == 64 && nst == 2) return launch_conv_epi<128, 256, 64, 2, 2>(*a, stream);
#define CASE(BM_, BN_, BK_)                                                                   \
  if (bm == BM_ && bn == BN_ && bk == BK_) {                                                  \
    if (nst == 2) return launch_conv_epi<BM_, BN_, BK_, 2>(*a, stream);                       \
    if (nst == 3) return launch_conv_epi<BM_, BN_, BK_, 3>(*a, stream);                       \
  }
  CASE(128, 192, 32)
  CASE(128, 192, 64)
  CASE(128, 128, 32)
  CASE(128, 64, 32)
  CASE(64, 64, 32)
  CASE(128, 128, 64)
  CASE(128, 64, 64)
  CASE(64, 64, 64)
#undef CASE
  return -3;
}
