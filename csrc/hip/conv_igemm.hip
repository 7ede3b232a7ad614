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

#include "conv_epilogue.h"


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
// 1: the LDS-DMA requests of a k-tile are issued one behind each MFMA of the tile's first k-step (the address
// arithmetic and the issue slots of a request then sit in the shadow of a running MFMA); 0: all of them in one burst
// ahead of the k-tile's first MFMA, during which neither wave of a SIMD feeds the matrix pipe.  (1: the 256 x 256
// macro tile only, 2: every tile.)  Measured at batch 2048 -- profiles/conv_pmc.md: +-3 % on the 256 x 256 tile, slower
// on the others (the requests' live state costs 10 - 19 registers, which takes a resident workgroup from the
// 256 x 128 macro tile), so the burst stays.
#ifndef CONV_SPREAD
#define CONV_SPREAD 0
#endif
// Two-stage tiles: 0 = k-tile kt+1 is requested at the top of k-tile kt (into the slot whose last reads completed
// before the previous barrier); 1 = it is requested one MFMA group earlier, right behind that barrier and ahead of
// the last MFMAs of k-tile kt-1 (every fragment of the slot is in registers by then), so the LDS-DMA path also works
// during that group and a k-tile has a third more time to land; 2 = the same with the requests spread over the
// group's MFMAs.  Measured: -3 .. 6 % on the 7x7 / 14x14 pointwise layers, +3 % on 14x14 3x3, i.e. no consistent gain.
#ifndef CONV_LATE
#define CONV_LATE 0
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
//
// KSP > 1 (with WG = 1) splits the REDUCTION over KSP groups of 2 x 2 waves inside one workgroup: a k-tile is
// KSP * BK deep, group g stages and multiplies the slice [g * BK, (g + 1) * BK) of it for the same output tile, and
// the groups' accumulators are summed through LDS before the epilogue.  A launch of few tiles with a deep reduction
// (the 8x8 / 17x17 Inception layers at batch 32, single-image serving) is a chain of k-tiles on a mostly idle chip;
// its pace is the fixed cost of a k-tile, so KSP times fewer of them is nearly KSP times faster.
template <int BM, int BN, int BK, int EPI, int NST, int WG, int KSP = 1>
__global__ __launch_bounds__(256 * WG * KSP, (KSP > 1 ? 1 : WG == 2 ? (BN > 128 ? 1 : 2) : (BM * BN > 128 * 128 ? 2 : (BM * BN <= 128 * 64 ? CONV_NARROW_OCC : 3)))) void conv_igemm_kernel(ConvArgs p) {
  constexpr int S = BK / 8;            // 16-byte slots per LDS row
  constexpr int ROWB = BK * 2;         // bytes per LDS row
  constexpr int RPI = 1024 / ROWB;     // rows covered by one wave-instruction
  constexpr int NA = BM * ROWB / 4096; // LDS-DMA instructions per thread (A)
  constexpr int NB = BN * ROWB / 4096;
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int MT = WM / 32, NT = WN / 32;
  constexpr int KS = BK / 16;
  static_assert(KSP == 1 || WG == 1, "the reduction is split over wave groups of a single-tile workgroup");
  // WG = 2: [A of group 0][A of group 1][B];  KSP > 1: [A slice 0] .. [A slice KSP-1][B slice 0] .. [B slice KSP-1]
  constexpr int STAGE = KSP > 1 ? KSP * (BM + BN) * ROWB : (WG * BM + BN) * ROWB;
  constexpr int NBW = NB / WG;                   // weight-tile loads per wave
  constexpr int EP = 4 * (BM / 2) * ((BN / 2) * 2 + EP_PAD) + 2 * (BN / 2) * 2 * 4;  // epilogue bytes per group
  constexpr int BOFF = (KSP > 1 ? KSP : WG) * BM * ROWB;   // start of the weight rows inside a stage
  static_assert(NA >= 1 && NBW >= 1 && NB % WG == 0, "tile too small for this staging");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* lds_taps = (int*)smem;
  char* lds_tiles = smem + MAX_TAPS * 4;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = (WG == 1 && KSP == 1) ? 0 : wid_all >> 2;   // wave group
  const int wid = wid_all & 3;    // wave inside its 2 x 2 group
  const int wr = wid >> 1, wc = wid & 1;

  const unsigned mtiles = (p.M + BM - 1) / BM;
  // this launch covers tile_m_count row tiles from tile_m_base on (0: all that follow)
  const unsigned mgroups = ((p.tile_m_count > 0 ? (unsigned)p.tile_m_count : mtiles - p.tile_m_base) + WG - 1) / WG;
  const unsigned ntiles = (p.Cout + BN - 1) / BN;
  const unsigned bid = xcd_remap(blockIdx.x, mgroups * ntiles);
  const unsigned tile_m = p.tile_m_base + (bid / ntiles) * WG + (KSP > 1 ? 0 : grp), tile_n = bid % ntiles;
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
  const bf16* abase2[NA];   // pointwise launches with a second operand: rows of x2
  int aiy[NA], aix[NA];
  const int K1 = p.K1, C2 = p.K - p.K1;
  const int opix = p.OYn * p.OXn;
  // 1x1 / stride-1 / unpadded: input pixel == output pixel == m, no decode and no bounds checks
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    int m = m0 + (i * 4 + wid) * RPI + row_in_inst;
    if (pointwise) {
      aiy[i] = m < p.M ? 0 : -(1 << 24);
      aix[i] = 0;
      abase[i] = p.x + (size_t)(m < p.M ? m : 0) * K1;
      abase2[i] = p.x2 != nullptr ? p.x2 + (size_t)(m < p.M ? m : 0) * C2 - K1 : abase[i];
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
    int n = n0 + (KSP > 1 ? (i * 4 + wid) : (i * 4 * WG + wid_all)) * RPI + row_in_inst;
    bok[i] = n < p.Cout;
    bbase[i] = p.w + (size_t)(bok[i] ? n : 0) * p.K;
  }

  if (!pointwise) __syncthreads();  // tap table visible

  // One k-tile's requests: ``stage_begin`` fixes the tile (destination slot, reduction chunk, tap), ``stage_piece(j)``
  // issues request j of NA + NBW (the pixel rows first, then this wave's share of the weight rows).
  char* st_sa = nullptr;
  char* st_sb = nullptr;
  int st_kc8 = 0, st_ci = 0, st_dy = 0, st_dx = 0;
  bool st_kvalid = false;
  auto stage_begin = [&](int kt, int buf) {
    st_sa = lds_tiles + buf * STAGE + grp * (BM * ROWB);
    st_sb = lds_tiles + buf * STAGE + BOFF + (KSP > 1 ? grp * (BN * ROWB) : 0);
    st_kc8 = (KSP > 1 ? (kt * KSP + grp) * BK : kt * BK) + cslot * 8;
    st_kvalid = st_kc8 < p.K;
    if (!pointwise) {
      const int tap = (int)__umulhi((unsigned)st_kc8, p.cin_magic);
      st_ci = st_kc8 - tap * p.Cin;
      const int tp = st_kvalid ? lds_taps[tap] : 0;
      st_dy = tp >> 16;
      st_dx = (int)(short)(tp & 0xffff);
    }
  };
  auto stage_piece = [&](int j) {
    if (j < NA) {
      const int i = j;
      const bf16* src;
      if (pointwise) {
        src = (st_kvalid && aiy[i] == 0) ? (st_kc8 < K1 ? abase[i] : abase2[i]) + st_kc8 : p.zero;
      } else {
        int iy = aiy[i] + st_dy, ix = aix[i] + st_dx;
        bool ok = st_kvalid && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
        // element offset inside the image with two full-rate 24-bit multiplies (the launcher checks
        // IH * IW < 2^24 and IH * IW * Cin < 2^32); the 64-bit form costs two quarter-rate multiplies per row
        const unsigned off = __umul24(__umul24((unsigned)iy, (unsigned)p.IW) + (unsigned)ix, (unsigned)p.Cin) + (unsigned)st_ci;
        src = ok ? abase[i] + off : p.zero;
      }
      glds16(src, st_sa + (i * 4 + wid) * 1024);
    } else {
      const int i = j - NA;
      const bf16* src = (st_kvalid && bok[i]) ? bbase[i] + st_kc8 : p.zero;
      glds16(src, st_sb + (KSP > 1 ? (i * 4 + wid) : (i * 4 * WG + wid_all)) * 1024);
    }
  };
  auto stage = [&](int kt, int buf) {
    stage_begin(kt, buf);
#pragma unroll
    for (int j = 0; j < NA + NBW; ++j) stage_piece(j);
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
    boff[nt] = BOFF + (KSP > 1 ? grp * (BN * ROWB) : 0) + row * ROWB;
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

  const int KT = (p.K + KSP * BK - 1) / (KSP * BK);
  // epilogue operands of the first readback passes, requested ahead of the main loop (narrow tiles only: the
  // registers come on top of the accumulators)
  constexpr bool PRE = WG == 1 && KSP == 1 && BN <= 128 && BK == 32 && EPI != 1;
  EpiPrefetch<BM, BN> pre;
  if (PRE && valid) epi_prefetch<BM, BN, EPI>(p, pre, wid, lane, m0, n0);
  constexpr int LPS = NA + NBW;  // LDS-DMA loads per thread and stage
  static_assert(KS % 2 == 0, "fragment double-buffering assumes an even number of k-steps per tile");
  // NST LDS stages (2 or 3): while k-tile kt is multiplied, the tiles up to kt+NST-1 are in flight.
  // Within a tile the fragments of k-step ks+1 are read from LDS while the MFMAs of k-step ks run, and
  // the first fragments of tile kt+1 are read behind the barrier, ahead of the last MFMAs of tile kt.
  static_assert(NST == 2 || NST == 3, "two or three LDS stages");
  constexpr bool LATE = CONV_LATE != 0 && NST == 2 && CONV_INTERLEAVE == 2;
  stage(0, 0);
  if (NST == 3 && KT > 1) {
    stage(1, 1);
    wait_vmcnt<LPS>();
  } else {
    wait_vmcnt<0>();
  }
  __syncthreads();
  if (LATE && KT > 1 && !(DTFA_DBG & 1)) stage(1, 1);

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

  // MFMAs of fragment set ``src`` with the reads of (buf, ks) into set ``dst`` issued one behind each MFMA; with
  // ``do_stage`` the requests of the k-tile prepared by ``stage_begin`` go out one behind each MFMA as well
  auto mfma_read_step = [&](int src, int buf, int ks, int dst, bool do_read, bool do_stage) {
    if (!wave_live) {
      if (do_stage) {
#pragma unroll
        for (int j = 0; j < LPS; ++j) stage_piece(j);
      }
      return;
    }
#if (DTFA_DBG & 6)
    mfma_step(src);
    if (do_read) read_frags(buf, ks, dst);
    if (do_stage) {
#pragma unroll
      for (int j = 0; j < LPS; ++j) stage_piece(j);
    }
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
        if (do_stage && i < LPS) stage_piece(i);
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
    if (do_stage) {   // ... and the remaining requests
#pragma unroll
      for (int j = MT * NT; j < LPS; ++j) stage_piece(j);
    }
    __builtin_amdgcn_s_setprio(0);
#endif
  };

  read_frags(0, 0, 0);
  int cur = 0;  // kt % NST
  for (int kt = 0; kt < KT; ++kt) {
    const int nxt = cur == NST - 1 ? 0 : cur + 1;
    // the slot requested here held tile kt-1, which every wave finished reading before the last barrier
    const bool do_stage = !LATE && kt + NST - 1 < KT && !(DTFA_DBG & 1);
    // (CONV_SPREAD 1: the 256 x 256 macro tile only -- one workgroup per CU, so nothing else feeds the matrix pipe during
    // a burst of eight requests per thread, and it has the registers for the requests' live state; 2: every tile)
    constexpr bool SPREAD = CONV_INTERLEAVE == 2 && (CONV_SPREAD == 2 || (CONV_SPREAD == 1 && WG == 2 && BN == 256));
    if (do_stage) {
      if (SPREAD) stage_begin(kt + NST - 1, NST == 2 ? nxt : (nxt == 2 ? 0 : nxt + 1));
      else stage(kt + NST - 1, NST == 2 ? nxt : (nxt == 2 ? 0 : nxt + 1));
    }
#pragma unroll
    for (int ks = 0; ks + 1 < KS; ++ks) {
#if CONV_INTERLEAVE
      // MFMAs of k-step ks first, then the LDS reads of k-step ks+1, which complete in the shadow of those
      // MFMAs: an MFMA group waits only for fragments requested a whole k-step earlier.  With the reads issued
      // BEFORE the MFMAs, the compiler's lgkmcnt(0) in front of the group waited for the fresh reads as well.
#if CONV_INTERLEAVE == 2
      mfma_read_step(ks & 1, cur, ks + 1, (ks + 1) & 1, true, SPREAD && do_stage && ks == 0);
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
    // (LATE: tile kt's slot is free behind the barrier -- its last fragments are in registers -- and takes tile kt+2)
    const bool late_stage = LATE && kt + 2 < KT && !(DTFA_DBG & 1);
    if (late_stage) {
      if (CONV_LATE == 2) stage_begin(kt + 2, cur);
      else stage(kt + 2, cur);
    }
    mfma_read_step((KS - 1) & 1, nxt, 0, 0, kt + 1 < KT, CONV_LATE == 2 && late_stage);
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
  if (KSP > 1) {
    // sum the groups' accumulators into group 0: group g parks its registers behind the epilogue areas (lane-major,
    // one float4 per store), group 0 adds them, one group after the other
    float* red = (float*)(lds_tiles + KSP * EP) + ((size_t)wid * 64 + lane) * (MT * NT * 16);
#pragma unroll
    for (int g = 1; g < KSP; ++g) {
      if (grp == g) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              *(f32x4*)(red + (mt * NT + nt) * 16 + q * 4) = f32x4{acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1],
                                                                  acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
      }
      __syncthreads();
      if (grp == 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4 v = *(const f32x4*)(red + (mt * NT + nt) * 16 + q * 4);
              acc[mt][nt][4 * q] += v[0];
              acc[mt][nt][4 * q + 1] += v[1];
              acc[mt][nt][4 * q + 2] += v[2];
              acc[mt][nt][4 * q + 3] += v[3];
            }
      }
      __syncthreads();
    }
  }
  // each group stages its output through its own part of the (now idle) tile area; with a split reduction only
  // group 0 holds the tile, the others walk through the epilogue's barriers without rows
  const bool has_tile = KSP == 1 || grp == 0;
  conv_epilogue<BM, BN, EPI, PRE>(p, acc, lds_tiles + grp * EP, wid, lane, has_tile ? m0 : p.M, n0, tile_m,
                                  valid && has_tile, &pre);
}

template <int BM, int BN, int BK, int EPI, int NST, int WG, int KSP = 1>
static int launch_conv(const ConvArgs& a, hipStream_t stream) {
  constexpr int ROWB = BK * 2;
  constexpr int STAGE = KSP > 1 ? KSP * (BM + BN) * ROWB : (WG * BM + BN) * ROWB;
  // row tiles + statistics exchange per group (+ the accumulator exchange of a split reduction)
  constexpr int EP = WG * KSP * (4 * (BM / 2) * ((BN / 2) * 2 + EP_PAD) + 2 * (BN / 2) * 2 * 4) + (KSP > 1 ? BM * BN * 4 : 0);
  size_t lds = MAX_TAPS * 4 + (size_t)((NST * STAGE > EP) ? NST * STAGE : EP);
  unsigned mtiles = a.tile_m_count > 0 ? (unsigned)a.tile_m_count : (a.M + BM - 1) / BM - a.tile_m_base;
  unsigned ntiles = (a.Cout + BN - 1) / BN;
  unsigned mgroups = (mtiles + WG - 1) / WG;
  // the dynamic-LDS limit is a per-device function attribute: remember it per device, not per process
  static bool attr_set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev = (dev >= 0 && dev < 64) ? dev : 0;
  if (!attr_set[dev] && lds > 48 * 1024) {
    (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<BM, BN, BK, EPI, NST, WG, KSP>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, EPI, NST, WG, KSP>), dim3(mgroups * ntiles), dim3(256 * WG * KSP), lds,
                     stream, a);
  return (int)hipGetLastError();
}

template <int BM, int BN, int BK, int NST, int WG = 1, int KSP = 1>
static int launch_conv_epi(const ConvArgs& a, hipStream_t stream) {
  if (a.bn_y != nullptr || a.bn_gate_store) return launch_conv<BM, BN, BK, 3, NST, WG, KSP>(a, stream);
  if (a.stats != nullptr || a.stat_acc != nullptr) return launch_conv<BM, BN, BK, 1, NST, WG, KSP>(a, stream);
  if (a.bias != nullptr || a.relu || a.oscale != nullptr || a.ldo != a.Cout)
    return launch_conv<BM, BN, BK, 2, NST, WG, KSP>(a, stream);  // (omask rides on relu)
  return launch_conv<BM, BN, BK, 0, NST, WG, KSP>(a, stream);
}

// Same-size 3x3 over 64 input channels with the weights resident on chip and the pixel operand staged once per tile:
// a persistent kernel of its own (conv_p3.hip).
int conv_p3_launch(const ConvArgs& a, hipStream_t stream);

// Number of partial-statistics rows the kernel writes for a given M and BM (one per row of tiles; a 256-row
// macro tile is two 128-row tiles).
extern "C" int dtfa_conv_stats_rows(int M, int bm) {
  if (bm == 256) bm = 128;
  return (M + bm - 1) / bm;
}

static int conv_igemm_dispatch(const ConvArgs* a, int bm, int bn, int bk, int nst, int ksp, hipStream_t stream);

extern "C" int dtfa_conv_igemm(const ConvArgs* a, int bm, int bn, int bk, int nst, hipStream_t stream) {
  return conv_igemm_dispatch(a, bm, bn, bk, nst, 1, stream);
}

// ``ksp`` > 1: the reduction of every tile split over ksp wave groups of one workgroup (the 64 x 64 x 64 tile with two
// stages: 2 or 4 groups).  ``ksp`` == -1 (with the 128 x 64 x 64 tile): the persistent 3x3 kernel of conv_p3.hip.
extern "C" int dtfa_conv_igemm_ks(const ConvArgs* a, int bm, int bn, int bk, int nst, int ksp, hipStream_t stream) {
  return conv_igemm_dispatch(a, bm, bn, bk, nst, ksp, stream);
}

static int conv_igemm_dispatch(const ConvArgs* a, int bm, int bn, int bk, int nst, int ksp, hipStream_t stream) {
  if (a->Cin % 8 != 0 || a->Cout % 8 != 0 || a->T > MAX_TAPS || a->T < 1) return -1;
  if ((a->x2 == nullptr && a->K != a->T * a->Cin) || a->K >= (1 << 16)) return -2;
  if ((long long)a->IH * a->IW >= (1ll << 24) || (long long)a->IH * a->IW * a->Cin >= (1ll << 32)) return -2;
  if (a->stats != nullptr && a->bias != nullptr) return -4;
  if (a->stat_acc != nullptr && (a->stat_rep < 0 || (a->stat_rep & (a->stat_rep - 1)) != 0)) return -4;
  if (a->stat_acc != nullptr && (a->stats != nullptr || a->bias != nullptr || a->relu || a->oscale != nullptr ||
                                 a->addend != nullptr || a->bn_y != nullptr || a->bn_gate_store || a->ldo != a->Cout))
    return -4;
  // a residual in the bias / ReLU (forward) epilogue: no bitmask, no statistics
  const bool fwd_residual = a->addend != nullptr && (a->bias != nullptr || a->relu || a->oscale != nullptr);
  if (fwd_residual && (a->addend_mask != nullptr || a->stats != nullptr || a->bn_y != nullptr)) return -4;
  if (a->addend != nullptr && !fwd_residual && a->stats != nullptr && a->bn_y == nullptr && !a->bn_gate_store) return -4;
  if (a->bn_gate_store && (a->bn_relu != 3 || a->bn_mask == nullptr || a->stats == nullptr || a->bias != nullptr || a->relu))
    return -5;
  if (a->omask != nullptr && (!a->relu || a->ldo != a->Cout || a->stats != nullptr || a->bn_y != nullptr)) return -6;
  if (a->dbias != nullptr && (a->bias != nullptr || a->relu || a->oscale != nullptr)) return -4;
  if (a->x2 != nullptr && (!a->pointwise || a->K1 <= 0 || a->K1 >= a->K || a->K1 % 8 != 0)) return -7;
  if (a->x2 == nullptr && a->K1 != a->K) return -7;
  if ((a->addend_shy | a->addend_shx) != 0 &&
      (a->addend == nullptr || fwd_residual || !a->dense_out || a->addend_mask != nullptr || a->addend_shy < 0 ||
       a->addend_shy > 4 || a->addend_shx < 0 || a->addend_shx > 4 || a->M != (long long)(a->M / (a->OH * a->OW)) * a->OH * a->OW))
    return -8;
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
  {
    const int unit = bm == 256 ? 128 : bm;   // tile_m_base counts the kernel's own row tiles (128 rows for the macro tiles)
    if (a->tile_m_base < 0 || a->tile_m_count < 0 || a->stat_row_base < 0 ||
        ((long long)a->tile_m_base + (a->tile_m_count > 0 ? a->tile_m_count - 1 : 0)) * unit >= a->M)
      return -9;
    if (a->tile_m_count > 0 && bm == 256 && (a->tile_m_count & 1)) return -9;   // whole macro tiles
  }
  if (ksp == -1) return (bm == 128 && bn == 64 && bk == 64) ? conv_p3_launch(*a, stream) : -3;
  if (ksp != 1) {
    if (bm == 64 && bn == 64 && bk == 64 && nst == 2 && ksp == 2) return launch_conv_epi<64, 64, 64, 2, 1, 2>(*a, stream);
    if (bm == 64 && bn == 64 && bk == 64 && nst == 2 && ksp == 4) return launch_conv_epi<64, 64, 64, 2, 1, 4>(*a, stream);
    return -3;
  }
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
