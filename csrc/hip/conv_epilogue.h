// Argument block and output epilogue shared by the implicit-GEMM convolution kernels (conv_igemm.hip,
// conv_p3.hip): bias / ReLU / batch-norm statistics / residual-gradient add / batch-norm backward sums, with
// bf16 rows staged through LDS so that global stores are 16 B per lane.
#pragma once
#include "common.h"
#include "ew_pack.h"

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
  int stream_out;                // the output is far larger than the caches: store it non-temporally
  unsigned char* omask;          // bias / ReLU epilogue with ReLU: one bit per output element, (out > 0), or null
  // data-gradient epilogues (EPI 0 / 3):
  const float* dbias;            // [Cout] added to the accumulator before rounding, or null
  int bn_gate_store;             // EPI 3: store the gated gradient g (not the raw sum) -- consumers then need no mask
  // pointwise launches: reduction indices >= K1 read a second operand x2 ([M][K - K1]); K1 == K: single operand
  const bf16* x2;
  int K1;
  // data-gradient epilogues with a dense output: the addend lives on a coarser pixel grid (the gradient of a strided
  // shortcut): output pixel (b, oy, ox) receives addend pixel (b, oy >> shy, ox >> shx) when oy and ox are
  // multiples of the stride (1 << shy, 1 << shx) and nothing otherwise.  0 / 0: the addend has the output's shape.
  int addend_shy, addend_shx;
  // statistics epilogue: instead of one partial row per tile in ``stats``, add the tile's column sums to
  // stat_acc[0..Cout) (sum) and stat_acc[Cout..2 Cout) (sum of squares), fp64, zeroed by the caller.  For launches of
  // few tiles, where a separate reduction kernel per layer costs as much as the layer (see bn_apply_sums).
  double* stat_acc;
  // A layer may be covered by two launches with different tiles (the last, partly filled round of large tiles
  // re-cut into small ones): this launch covers tile_m_count row tiles from tile_m_base on (in units of its own
  // tile height, 128 rows for the macro tiles; count 0 = up to the last), and its first row of partial statistics
  // is stat_row_base.
  int tile_m_base, tile_m_count, stat_row_base;
  // stat_acc: number of replicas of the accumulators (a power of two, 0 = 1), 2 * Cout doubles each; row tile t adds to
  // replica t mod stat_rep and the consumer sums the replicas.  Every tile of a column adds to the same 2 * Cout
  // addresses, and atomics on one address are served one after the other: a layer of 613 row tiles (35x35 at batch 32)
  // spent 12 of its 19 us waiting for them.
  int stat_rep;
};

#define MAX_TAPS 64
// Bytes added to the pitch of the epilogue's bf16 staging rows (see conv_epilogue).  With 0 the readback of a wave
// (64 lanes x 16 bytes) is one contiguous kilobyte, conflict-free for the lane groups of ds_read_b128, with 16 three of
// every sixteen lanes share a bank pair with another -- and the ResNet-50 step takes the same time either way (16 943 /
// 16 951 against 16 955 / 16 940 img/s, alternating on one box): these kernels wait for memory, not for LDS.
#ifndef EP_PAD
#define EP_PAD 16
#endif

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

// Readback geometry of the epilogue for a BM x BN tile computed by 2 x 2 waves (see conv_epilogue).
template <int BM, int BN>
struct EpiGeom {
  static constexpr int WM = BM / 2, WN = BN / 2;
  static constexpr int CHUNKS = WN / 8;      // 16-byte chunks per row
  static constexpr int CPR = CHUNKS <= 4 ? 4 : (CHUNKS <= 8 ? 8 : 16);
  static constexpr int RPP = 64 / CPR;       // rows per pass
  static constexpr int NP = WM / RPP;        // passes
  static constexpr int G = NP >= 4 ? 4 : NP; // passes whose global loads are in flight together
};

// Element offset of the 16-byte chunk a lane handles in readback pass ``pass`` (``ok`` false: nothing to do);
// ``pixel`` receives the physical output pixel.
template <int BM, int BN>
__device__ __forceinline__ size_t epi_out_offset(const ConvArgs& p, int ldo, int w4, int lane, int m0, int n0, int pass,
                                                 bool& ok, size_t& pixel) {
  using E = EpiGeom<BM, BN>;
  const int wr = w4 >> 1, wc = w4 & 1;
  const int chunk_co = n0 + wc * E::WN + (lane % E::CPR) * 8;
  const int m = m0 + wr * E::WM + pass * E::RPP + lane / E::CPR;
  ok = m < p.M && chunk_co < p.Cout && (lane % E::CPR) < E::CHUNKS;
  size_t opixel = 0;
  if (ok) {
    if (p.dense_out) {
      opixel = (size_t)m;
    } else {
      const int opix = p.OYn * p.OXn;
      int b = m / opix;
      int rem = m - b * opix;
      int oy = rem / p.OXn;
      int ox = rem - oy * p.OXn;
      opixel = ((size_t)b * p.OH + (oy * p.osy + p.oy0)) * p.OW + (ox * p.osx + p.ox0);
    }
  }
  pixel = opixel;
  return opixel * ldo + chunk_co;
}

// Element offset of the addend chunk for output pixel ``pixel`` when the addend is on a coarser grid (see
// ConvArgs::addend_shy); ``live`` is false for pixels between the grid points.
__device__ __forceinline__ size_t epi_coarse_addend_offset(const ConvArgs& p, size_t pixel, int chunk_co, bool& live) {
  const unsigned m = (unsigned)pixel;
  const unsigned opix = (unsigned)(p.OH * p.OW);
  const unsigned b = m / opix, rem = m - b * opix;
  const unsigned oy = rem / (unsigned)p.OW, ox = rem - oy * (unsigned)p.OW;
  const unsigned my = (1u << p.addend_shy) - 1u, mx = (1u << p.addend_shx) - 1u;
  live = ((oy & my) | (ox & mx)) == 0u;
  const unsigned ah = ((unsigned)p.OH + my) >> p.addend_shy, aw = ((unsigned)p.OW + mx) >> p.addend_shx;
  return (((size_t)b * ah + (oy >> p.addend_shy)) * aw + (ox >> p.addend_shx)) * p.Cout + chunk_co;
}

// The epilogue's operand loads of the first G readback passes (residual / residual gradient and the gate bits).  They do not depend on the GEMM, so a kernel whose reduction is short (a 1x1
// layer with few input channels is two k-tiles) issues them before its main loop: the tile then pays the
// memory latency once, not once for the operand tiles and again for the epilogue.
template <int BM, int BN>
struct EpiPrefetch {
  static constexpr int G = EpiGeom<BM, BN>::G;
  bf16x8 addv[G];
  unsigned addm[G], bnm[G];
};

template <int BM, int BN, int EPI>
__device__ __forceinline__ void epi_prefetch(const ConvArgs& p, EpiPrefetch<BM, BN>& r, int w4, int lane, int m0, int n0) {
  using E = EpiGeom<BM, BN>;
  if (EPI == 1) return;
  const int ldo = EPI == 2 ? p.ldo : p.Cout;
  const int chunk_co = n0 + (w4 & 1) * E::WN + (lane % E::CPR) * 8;
#pragma unroll
  for (int i = 0; i < E::G; ++i) {
    bool ok;
    size_t pixel;
    const size_t o = epi_out_offset<BM, BN>(p, ldo, w4, lane, m0, n0, i, ok, pixel);
    const size_t off = ok ? (EPI == 2 ? pixel * p.Cout + chunk_co : o) : 0;  // idle lanes read an in-range dummy
    if (p.addend != nullptr) {
      if (EPI != 2 && (p.addend_shy | p.addend_shx) != 0) {
        bool live;
        const size_t aoff = epi_coarse_addend_offset(p, pixel, chunk_co, live);
        r.addv[i] = (ok && live) ? epi_load((const bf16x8*)(p.addend + aoff)) : bf16x8{};
        r.addm[i] = 0xffu;
      } else {
        r.addv[i] = epi_load((const bf16x8*)(p.addend + off));
        r.addm[i] = (EPI != 2 && p.addend_mask != nullptr) ? (unsigned)p.addend_mask[off >> 3] : 0xffu;
      }
    }
    if (EPI == 3) r.bnm[i] = p.bn_relu == 3 ? (unsigned)p.bn_mask[off >> 3] : 0xffu;
  }
}

// Timing-only builds for attributing epilogue time (results are wrong): -DEPI_DBG bit 0 drops the gate-bit loads of
// a tap, bit 1 the final reduction and store of its sums, bit 2 the staging of the accumulators through LDS.
#ifndef EPI_DBG
#define EPI_DBG 0
#endif

// The (bias, scale) pair of a lane's column ``co`` as conv_epilogue loads it (see its ``held`` argument).
template <int EPI>
__device__ __forceinline__ void epi_hold_column(const ConvArgs& p, int co, float* pair) {
  pair[0] = 0.f;
  pair[1] = 1.f;
  if (co >= p.Cout) return;
  if (EPI == 2 && p.bias != nullptr) pair[0] = p.bias[co];
  if (EPI == 2 && p.oscale != nullptr) pair[1] = p.oscale[co];
  if ((EPI == 0 || EPI == 3) && p.dbias != nullptr) pair[0] = p.dbias[co];
}

template <int BM, int BN, int EPI, bool PRE = false>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x16 (&acc)[BM / 64][BN / 64], char* lds_tiles,
                                              int w4, int lane, int m0, int n0, unsigned tile_m, bool valid,
                                              const EpiPrefetch<BM, BN>* pre = nullptr, const float* held = nullptr) {
  // ``held`` (kernels that walk many tiles): the lane's per-column (bias, scale) pairs, [NT][2], loaded once by the
  // caller -- a load per tile here waits behind the previous tile's stores, a memory round trip per tile.
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int MT = WM / 32, NT = WN / 32;
  constexpr int EP_PITCH = WN * 2 + EP_PAD;
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
    return epi_out_offset<BM, BN>(p, ldo, w4, lane, m0, n0, pass, ok, last_pixel);
  };
  char* ep = lds_tiles + wid * (WM * EP_PITCH);
  float* xch = (float*)(lds_tiles + 4 * WM * EP_PITCH);  // [2 wave columns][WN][2] exchange area behind the row tiles
  float st1[NT], st2[NT];
  const int colq = lane & 31;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = n0 + wc * WN + nt * 32 + colq;
    float bias = 0.f, osc = 1.f;
    if (held != nullptr) {
      bias = held[nt * 2];
      osc = held[nt * 2 + 1];
    } else {
      if (EPI == 2 && p.bias != nullptr && co < p.Cout) bias = p.bias[co];
      if (EPI == 2 && p.oscale != nullptr && co < p.Cout) osc = p.oscale[co];
      if ((EPI == 0 || EPI == 3) && p.dbias != nullptr && co < p.Cout) bias = p.dbias[co];
    }
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
        if (EPI == 0 || EPI == 3) v += bias;
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
        const float t1 = st1[nt] + xch[((wc * NT + nt) * 32 + colq) * 2];
        const float t2 = st2[nt] + xch[((wc * NT + nt) * 32 + colq) * 2 + 1];
        if (p.stat_acc != nullptr) {
          double* acc = p.stat_acc + (p.stat_rep > 1 ? (size_t)(tile_m & (unsigned)(p.stat_rep - 1)) * (2 * p.Cout) : 0);
          unsafeAtomicAdd(acc + co, (double)t1);
          unsafeAtomicAdd(acc + p.Cout + co, (double)t2);
        } else {
          const size_t prow = ((size_t)tile_m - p.tile_m_base + p.stat_row_base) * 2;
          p.stats[prow * p.Cout + co] = t1;
          p.stats[(prow + 1) * p.Cout + co] = t2;
        }
      }
    }
  }

  float bs1[8], bs2[8], bmu[8], bis[8], bsc[8], bsh[8];
  if (EPI == 3) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = chunk_co + j < p.Cout && chunk_live;
      bs1[j] = bs2[j] = 0.f;
      bmu[j] = (ok && p.bn_mean != nullptr) ? p.bn_mean[chunk_co + j] : 0.f;
      bis[j] = (ok && p.bn_invstd != nullptr) ? p.bn_invstd[chunk_co + j] : 0.f;
      bsc[j] = (ok && p.bn_relu == 2) ? p.bn_scale[chunk_co + j] : 0.f;
      bsh[j] = (ok && p.bn_relu == 2) ? p.bn_shift[chunk_co + j] : 0.f;
    }
  }
  f32x2 bs1p_out[4], bs2p_out[4];
  // Readback in groups of G passes.  All global loads of a group (residual / residual gradient, gate bits, the
  // batch-norm input of a tap) are requested before any is consumed, and the loads of group g+1 are requested
  // before group g is consumed (two register sets; the accumulators are dead by now), so a tile pays the memory
  // latency of its epilogue operands once.  Group 0 may have been requested ahead of the main loop (PRE).
  {
    constexpr int G = EpiGeom<BM, BN>::G;
    constexpr int NG = NP / G;
    static_assert(NP % G == 0, "passes are grouped evenly");
    const bool has_add = EPI != 1 && p.addend != nullptr;
    const bool has_addm = (EPI == 0 || EPI == 3) && p.addend_mask != nullptr;
    const bool has_bny = EPI == 3 && p.bn_y != nullptr;
    const bool coarse_add = (EPI == 0 || EPI == 3) && (p.addend_shy | p.addend_shx) != 0;
    const bool gate_store = EPI == 3 && p.bn_gate_store != 0;
    struct Group {
      size_t ooff[G];
      bool ook[G];
      bf16x8 addv[G], bnyv[G];
      unsigned addm[G], bnm[G];
    };
    Group grp[2];
    auto load_group = [&](int g0, Group& q, bool from_pre) {
#pragma unroll
      for (int i = 0; i < G; ++i) {
        q.ooff[i] = out_offset(g0 + i, q.ook[i]);
        // idle lanes read an in-range dummy; the forward residual is dense ([.., Cout]) even when y is a slice
        const size_t off = q.ook[i] ? (EPI == 2 ? last_pixel * p.Cout + chunk_co : q.ooff[i]) : 0;
        if (has_add) {
          if (from_pre) {
            q.addv[i] = pre->addv[i];
            q.addm[i] = pre->addm[i];
          } else if (coarse_add) {
            bool live;
            const size_t aoff = epi_coarse_addend_offset(p, last_pixel, chunk_co, live);
            q.addv[i] = (q.ook[i] && live) ? epi_load((const bf16x8*)(p.addend + aoff)) : bf16x8{};
            q.addm[i] = 0xffu;
          } else {
            q.addv[i] = epi_load((const bf16x8*)(p.addend + off));
            q.addm[i] = has_addm ? (unsigned)p.addend_mask[off >> 3] : 0xffu;
          }
        }
        if (EPI == 3) {
          if (has_bny) q.bnyv[i] = epi_load((const bf16x8*)(p.bn_y + off));
#if (EPI_DBG & 1)
          q.bnm[i] = 0xffu;
#else
          q.bnm[i] = from_pre ? pre->bnm[i] : (p.bn_relu == 3 ? (unsigned)p.bn_mask[off >> 3] : 0xffu);
#endif
        }
      }
    };
    // two bits of a gate byte -> an AND mask over the two bf16 halves of a word
    auto keep2 = [](unsigned bits) -> unsigned { return ((bits & 1u) ? 0xffffu : 0u) | ((bits & 2u) ? 0xffff0000u : 0u); };
    f32x2 bs1p[4], bs2p[4];   // EPI 3: the sums as fp32 pairs (v_pk_add_f32 / v_pk_fma_f32)
#pragma unroll
    for (int j = 0; j < 4; ++j) bs1p[j] = bs2p[j] = f32x2{0.f, 0.f};
    // Per element the readback is a handful of operations, but a 128 x 128 tile of a short-reduction layer is
    // nothing but readback: written one element at a time it cost as much vector-ALU time as the tile's memory
    // traffic.  Hence two elements per instruction on the packed bf16 words, gates as AND masks.
    auto consume_group = [&](int g0, const Group& q) {
#pragma unroll
      for (int i = 0; i < G; ++i) {
        if (!q.ook[i]) continue;
        const int r = (g0 + i) * RPP + lane / CPR;
        u32x4 v = *(const u32x4*)(ep + r * EP_PITCH + (lane % CPR) * 16);
        const u32x4 av = __builtin_bit_cast(u32x4, q.addv[i]);
        if (EPI == 1 || EPI == 2) {
          unsigned bits = 0;
          if (has_add) {
            // residual block: act(conv * scale + shift + residual)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              unsigned w = pack_bf16x2(unpack_bf16x2(v[j]) + unpack_bf16x2(av[j]));
              if (p.relu) w = relu_bf16x2(w);
              v[j] = w;
            }
          }
          if (p.stream_out) __builtin_nontemporal_store(__builtin_bit_cast(bf16x8, v), (bf16x8*)(p.y + q.ooff[i]));
          else out_store((bf16x8*)(p.y + q.ooff[i]), __builtin_bit_cast(bf16x8, v));
          if (EPI == 2 && p.omask != nullptr) {
            // sign bits of the activation (training: all the backward pass needs of it); dense outputs only.
            // After a ReLU a value is positive exactly when its bf16 word is non-zero.
#pragma unroll
            for (int j = 0; j < 4; ++j) bits |= positive_bits_bf16x2(v[j]) << (2 * j);
            p.omask[q.ooff[i] >> 3] = (unsigned char)bits;
          }
        } else {
          // data gradient: add a residual gradient (optionally gated by a bitmask), fold the batch-norm sums
          const u32x4 yv = __builtin_bit_cast(u32x4, q.bnyv[i]);
          u32x4 gv;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            unsigned w = v[j];
            if (has_add) {
              const unsigned aw = has_addm ? (av[j] & keep2(q.addm[i] >> (2 * j))) : av[j];
              w = pack_bf16x2(unpack_bf16x2(w) + unpack_bf16x2(aw));
            }
            v[j] = w;
            if (EPI == 3) {
              // g: the rounded value, the one the elementwise backward pass will read, gated by the ReLU sign
              unsigned gw = w;
              f32x2 yf = f32x2{0.f, 0.f};
              if (has_bny) {
                yf = unpack_bf16x2(yv[j]);
                if (p.bn_relu == 2) {
                  const f32x2 pre = __builtin_elementwise_fma(yf, f32x2{bsc[2 * j], bsc[2 * j + 1]}, f32x2{bsh[2 * j], bsh[2 * j + 1]});
                  gw &= (pre.x > 0.f ? 0xffffu : 0u) | (pre.y > 0.f ? 0xffff0000u : 0u);
                }
              }
              if (p.bn_relu == 3) gw &= keep2(q.bnm[i] >> (2 * j));
              const f32x2 g = unpack_bf16x2(gw);
              bs1p[j] += g;
              if (has_bny) {
                const f32x2 xh = (yf - f32x2{bmu[2 * j], bmu[2 * j + 1]}) * f32x2{bis[2 * j], bis[2 * j + 1]};
                bs2p[j] = __builtin_elementwise_fma(g, xh, bs2p[j]);
              }
              gv[j] = gw;
            }
          }
          if (p.stream_out) __builtin_nontemporal_store(__builtin_bit_cast(bf16x8, gate_store ? gv : v), (bf16x8*)(p.y + q.ooff[i]));
          else out_store((bf16x8*)(p.y + q.ooff[i]), __builtin_bit_cast(bf16x8, gate_store ? gv : v));
        }
      }
    };
#ifndef EPI_PIPE
#define EPI_PIPE 1
#endif
    constexpr bool PIPE = EPI_PIPE == 2 || (EPI_PIPE == 1 && EPI == 2);
    if (PIPE) {
      load_group(0, grp[0], PRE);
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
        if (gi + 1 < NG) load_group((gi + 1) * G, grp[(gi + 1) & 1], false);
        consume_group(gi * G, grp[gi & 1]);
      }
    } else {
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
        load_group(gi * G, grp[0], PRE && gi == 0);
        consume_group(gi * G, grp[0]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bs1p_out[j] = bs1p[j];
      bs2p_out[j] = bs2p[j];
    }
  }
  if (EPI == 3 && !(EPI_DBG & 2)) {
    // Per-column totals of the tile: every lane parks its eight sums in its wave's (now idle) row-tile area,
    // lane-linear; after the barrier thread c of the group adds the 2 wave rows x (64 / CPR) row groups of
    // column c in a fixed order and stores one float -- a coalesced row of BN values per plane.
    const bool has_bny2 = p.bn_y != nullptr;
    float* red = (float*)ep;   // this wave's region: [lane][8] per plane
    // both planes side by side where a wave's region holds them (2 x 2 KB), else one after the other
    constexpr bool ONE_ROUND = WM * EP_PITCH >= 4096;
    const int t = w4 * 64 + lane;   // thread index inside the 2 x 2 wave group
    const bool reader = t < BN && n0 + t < p.Cout && valid;
    const int cwc = t / WN, cc = t % WN;
    const int at = (cc >> 3) * 8 + (cc & 7);
    const size_t prow = ((size_t)tile_m - p.tile_m_base + p.stat_row_base) * 2;
    auto park = [&](const f32x2 (&v)[4], int plane_off) {
      *(f32x4*)(red + plane_off + lane * 8) = f32x4{v[0].x, v[0].y, v[1].x, v[1].y};
      *(f32x4*)(red + plane_off + lane * 8 + 4) = f32x4{v[2].x, v[2].y, v[3].x, v[3].y};
    };
    auto column_total = [&](int plane_off) {
      float a = 0.f;
#pragma unroll
      for (int wrow = 0; wrow < 2; ++wrow) {
        const float* rw = (const float*)(lds_tiles + (wrow * 2 + cwc) * (WM * EP_PITCH));
#pragma unroll
        for (int rg = 0; rg < 64 / CPR; ++rg) a += rw[plane_off + rg * CPR * 8 + at];
      }
      return a;
    };
    park(bs1p_out, 0);
    if (ONE_ROUND && has_bny2) park(bs2p_out, 512);
    __syncthreads();
    if (reader) {
      p.stats[prow * p.Cout + n0 + t] = column_total(0);
      if (!has_bny2) p.stats[(prow + 1) * p.Cout + n0 + t] = 0.f;
      else if (ONE_ROUND) p.stats[(prow + 1) * p.Cout + n0 + t] = column_total(512);
    }
    if (!ONE_ROUND && has_bny2) {
      __syncthreads();
      park(bs2p_out, 0);
      __syncthreads();
      if (reader) p.stats[(prow + 1) * p.Cout + n0 + t] = column_total(0);
    }
  }
}
