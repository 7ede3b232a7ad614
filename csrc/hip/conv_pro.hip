// Pointwise (1x1, stride 1, unpadded) convolution whose pixel operand is produced on the fly by a fused
// *prologue*: the batch-norm passes that would otherwise sweep HBM on their own.
//
// A residual bottleneck spends a third of its memory traffic in two elementwise passes over its widest
// tensors: forward  z = relu(y*scale + shift + r)  (read y, read r, write z), whose result is immediately
// re-read by the next 1x1 convolution, and backward  dy = ca*g + cb*y + cc  (read dz, read y, write dy), whose
// result is immediately re-read by the data-gradient GEMM of the same layer.  Here those GEMMs load the raw
// operands themselves -- global -> registers, 16 B per lane -- apply the per-channel transform in fp32, and
// write the bf16 result both into the LDS operand tile (same XOR-swizzled image the LDS-DMA path produces, so
// the MFMA main loop is unchanged) and, from the first column of tiles only, back to global memory as a *side
// output* for the other consumers of the tensor (the residual connection and weight gradient forward, the
// weight gradient backward).  The separate pass and the re-read of its output disappear: 17.25 -> 13.25
// tensor sweeps for batch-norm + convolution of a bottleneck's input, 18.25 -> 14.25 for the gradient.
//
// The per-channel coefficients are indexed by the reduction index k (= input channel of a 1x1 convolution);
// a thread's 16-byte chunk keeps one column of 8 channels for every row it stages, so the coefficients are
// read (from a table staged in LDS) once per k-tile.  The weight tile still arrives by LDS-DMA.  Two LDS stages
// and two register stages: the raw chunks of two further k-tiles are in flight while one tile is multiplied,
// which is what a streaming kernel needs to cover HBM latency at two or three workgroups per CU.
#include "conv_epilogue.h"
#include "ew_pack.h"
#include <type_traits>

// streaming hints (PRO_NT: bit 0 = non-temporal operand loads, bit 1 = non-temporal side stores)
#ifndef PRO_NT
#define PRO_NT 0
#endif
__device__ __forceinline__ u32x4 pro_load(const bf16* p) {
#if (PRO_NT & 1)
  return __builtin_nontemporal_load((const u32x4*)p);
#else
  return *(const u32x4*)p;
#endif
}
__device__ __forceinline__ void pro_store(bf16* p, u32x4 v) {
#if (PRO_NT & 2)
  __builtin_nontemporal_store(v, (u32x4*)p);
#else
  *(u32x4*)p = v;
#endif
}

// prologue modes
#define PRO_FWD 1        // a = act(y*c0 + c1)
#define PRO_FWD_RES 2    // a = relu(y*c0 + c1 + r), sign bits out
#define PRO_BWD_RELU 3   // g = dz gated by (y*gscale + gshift > 0);  a = c0*g + c1*y + c2
#define PRO_BWD_MASK 4   // g = dz gated by stored sign bits;         a = c0*g + c1*y + c2
#define PRO_BWD_PLAIN 5  // g = dz;                                   a = c0*g + c1*y + c2

struct ProArgs {
  const bf16* src2;            // PRO_FWD_RES: the residual r;  PRO_BWD_*: the batch-norm input y   [M][K]
  const unsigned char* gate;   // PRO_BWD_MASK: one bit per element of the first operand
  const float* c0;             // [K]
  const float* c1;             // [K]
  const float* c2;             // [K] (backward modes)
  const float* gscale;         // [K] PRO_BWD_RELU: scale / shift of the forward pass
  const float* gshift;
  bf16* side;                  // [M][K]: the transformed operand written back, or null
  unsigned char* side_mask;    // forward modes with ReLU: (a > 0), one bit per element, or null
  int relu;                    // PRO_FWD: apply the ReLU
};

// workgroups per CU the kernel is compiled for: two register stages of the pixel operand come on top of the
// accumulators, so only the narrow tiles fit the 168-register budget of three
constexpr int pro_occupancy(int BN, int BK, int PRO) {
  return (BN > 64 || BK > 32) ? 2 : 3;
}
constexpr int pro_ncoef(int PRO) { return PRO <= PRO_FWD_RES ? 2 : (PRO == PRO_BWD_RELU ? 5 : 3); }

template <int BN, int BK, int EPI, int PRO>
__global__ __launch_bounds__(256, pro_occupancy(BN, BK, PRO)) void conv_pro_kernel(ConvArgs p, ProArgs q) {
  constexpr int BM = 128;
  constexpr int S = BK / 8;            // 16-byte slots per LDS row
  constexpr int ROWB = BK * 2;         // bytes per LDS row
  constexpr int RPI = 1024 / ROWB;     // rows covered by one wave-instruction
  constexpr int NA = BM * ROWB / 4096; // 16-byte chunks per thread and k-tile (pixel operand)
  constexpr int NB = BN * ROWB / 4096; // LDS-DMA instructions per thread (weights)
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int MT = WM / 32, NT = WN / 32;
  constexpr int KS = BK / 16;
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr bool TWO_SRC = PRO != PRO_FWD;
  constexpr bool BWD = PRO >= PRO_BWD_RELU;
  constexpr int NCOEF = pro_ncoef(PRO);
  constexpr int NSRC = TWO_SRC ? 2 : 1;
  static_assert(NA >= 1 && NB >= 1 && KS % 2 == 0, "tile too small for this staging");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_tiles = smem;
  const int K = p.K;
  const int KT = (K + BK - 1) / BK;
  const int KP = KT * BK;              // coefficient rows are padded to whole k-tiles
  // per-channel coefficients behind the tile area: [NCOEF][KP] floats
  constexpr int EP = 4 * (BM / 2) * ((BN / 2) * 2 + 16) + 2 * (BN / 2) * 2 * 4;
  constexpr int TILES = 2 * STAGE > EP ? 2 * STAGE : EP;
  float* lds_coef = (float*)(smem + TILES);
  unsigned char* lds_mask = (unsigned char*)(lds_coef + NCOEF * KP);   // [BM][K / 8] (forward modes with a bitmask)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;

  const unsigned mtiles = (p.M + BM - 1) / BM;
  const unsigned ntiles = (p.Cout + BN - 1) / BN;
  const unsigned bid = xcd_remap(blockIdx.x, mtiles * ntiles);
  const unsigned tile_m = bid / ntiles, tile_n = bid % ntiles;
  const int m0 = (int)(tile_m * BM), n0 = tile_n * BN;

  {
    const float* src[5] = {q.c0, q.c1, q.c2, q.gscale, q.gshift};
#pragma unroll
    for (int c = 0; c < NCOEF; ++c)
      for (int k = tid; k < KP; k += 256) lds_coef[c * KP + k] = k < K ? src[c][k] : 0.f;
  }

  // ---- per-thread staging rows (fixed across the K loop) ----
  const int slot = lane % S;
  const int row_in_inst = lane / S;
  const int cslot = slot ^ swz_of_row<BK>(wid * RPI + row_in_inst);  // same for every i
  const int mrow0 = m0 + wid * RPI + row_in_inst;                     // row of chunk i: mrow0 + i * 4 * RPI
  const bool side_tile = tile_n == 0;
  const bool want_mask = !BWD && q.side_mask != nullptr && side_tile;

  const bf16* bbase[NB];
  bool bok[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    int n = n0 + (i * 4 + wid) * RPI + row_in_inst;
    bok[i] = n < p.Cout;
    bbase[i] = p.w + (size_t)(bok[i] ? n : 0) * K;
  }

  // ---- two register stages of the pixel operand: tile t lives in set t & 1 ----
  u32x4 ra[2][NA], rb[2][NA];
  unsigned rmask[2][NA];
  auto load_a = [&](int kt, auto set_c) {
    constexpr int set = decltype(set_c)::value;
    const int kc8 = kt * BK + cslot * 8;
    const bool kvalid = kc8 < K;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int m = mrow0 + i * 4 * RPI;
      const size_t off = (kvalid && m < p.M) ? (size_t)m * K + kc8 : 0;  // idle lanes read an in-range dummy
      ra[set][i] = pro_load(p.x + off);
      if (TWO_SRC) rb[set][i] = pro_load(q.src2 + off);
      if (PRO == PRO_BWD_MASK) rmask[set][i] = q.gate[off >> 3];
    }
  };
  // transform the staged chunks, store them into the LDS tile of ``buf`` and (first column of tiles) to global
  auto put_a = [&](int kt, int buf, auto set_c) {
    constexpr int set = decltype(set_c)::value;
    char* sa = lds_tiles + buf * STAGE;
    const int kc8 = kt * BK + cslot * 8;
    const bool kvalid = kc8 < K;
    f32x4 kv[NCOEF][2];
#pragma unroll
    for (int c = 0; c < NCOEF; ++c) {
      kv[c][0] = *(const f32x4*)(lds_coef + c * KP + kc8);
      kv[c][1] = *(const f32x4*)(lds_coef + c * KP + kc8 + 4);
    }
    auto coef = [&](int c, int j) { return f32x2{kv[c][j >> 1][(j & 1) * 2], kv[c][j >> 1][(j & 1) * 2 + 1]}; };
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int m = mrow0 + i * 4 * RPI;
      const bool ok = kvalid && m < p.M;
      u32x4 o;
      unsigned bits = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (!BWD) {
          f32x2 f = __builtin_elementwise_fma(unpack_bf16x2(ra[set][i][j]), coef(0, j), coef(1, j));
          if (PRO == PRO_FWD_RES) f += unpack_bf16x2(rb[set][i][j]);
          unsigned w = pack_bf16x2(f);
          if (PRO == PRO_FWD_RES || q.relu) {
            w = relu_bf16x2(w);
            bits |= positive_bits_bf16x2(w) << (2 * j);
          }
          o[j] = w;
        } else {
          const f32x2 yv = unpack_bf16x2(rb[set][i][j]);
          unsigned gw = ra[set][i][j];
          if (PRO == PRO_BWD_RELU) {
            const f32x2 pre = __builtin_elementwise_fma(yv, coef(NCOEF - 2, j), coef(NCOEF - 1, j));
            gw &= (pre.x > 0.f ? 0xffffu : 0u) | (pre.y > 0.f ? 0xffff0000u : 0u);
          }
          if (PRO == PRO_BWD_MASK) {
            const unsigned m2 = rmask[set][i] >> (2 * j);
            gw &= ((m2 & 1u) ? 0xffffu : 0u) | ((m2 & 2u) ? 0xffff0000u : 0u);
          }
          o[j] = pack_bf16x2(__builtin_elementwise_fma(coef(0, j), unpack_bf16x2(gw),
                                                       __builtin_elementwise_fma(coef(1, j), yv, coef(2, j))));
        }
      }
      if (!ok) o = u32x4{0u, 0u, 0u, 0u};  // rows beyond M / columns beyond K multiply as zeros
      *(u32x4*)(sa + (i * 4 + wid) * 1024 + lane * 16) = o;
      if (ok && side_tile) {
        const size_t off = (size_t)m * K + kc8;
        if (q.side != nullptr) pro_store(q.side + off, o);
        // sign bits: collected in an LDS panel [rows][K / 8] and written out in one coalesced sweep at the end
        if (!BWD && want_mask) lds_mask[(m - m0) * (K >> 3) + (kc8 >> 3)] = (unsigned char)bits;
      }
    }
  };
  auto stage_b = [&](int kt, int buf) {
    char* sb = lds_tiles + buf * STAGE + BM * ROWB;
    const int kc8 = kt * BK + cslot * 8;
    const bool kvalid = kc8 < K;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const bf16* src = (kvalid && bok[i]) ? bbase[i] + kc8 : p.zero;
      glds16(src, sb + (i * 4 + wid) * 1024);
    }
  };

  // ---- fragment read offsets ----
  int aoff[MT], aswz[MT], boff[NT], bswz[NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int row = wr * WM + mt * 32 + (lane & 31);
    aoff[mt] = row * ROWB;
    aswz[mt] = swz_of_row<BK>(row);
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int row = wc * WN + nt * 32 + (lane & 31);
    boff[nt] = BM * ROWB + row * ROWB;
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

  using set0 = std::integral_constant<int, 0>;
  using set1 = std::integral_constant<int, 1>;
  // Pipeline: while tile kt is multiplied out of LDS buffer kt & 1, the weights of tile kt+1 are landing in the
  // other buffer, the pixel chunks of tile kt+1 sit in register set (kt+1) & 1 and those of tile kt+2 are in
  // flight into set kt & 1.  Issue order (oldest first) at the wait below: A(kt+1), stores(kt), B(kt+1), A(kt+2);
  // vmcnt(NA * NSRC) leaves at most the 16-byte loads of A(kt+2) outstanding (loads return in order).
  load_a(0, set0{});
  stage_b(0, 0);
  if (KT > 1) load_a(1, set1{});
  __syncthreads();  // coefficient table visible
  if (KT > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA * NSRC) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  put_a(0, 0, set0{});
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (KT > 1) stage_b(1, 1);
  if (KT > 2) load_a(2, set0{});

  bf16x8 af[2][MT], bfr[2][NT];
  const bool wave_live = n0 + wc * WN < p.Cout;
  // MFMAs of fragment set ``src`` with the reads of (buf, ks) into set ``dst`` issued one behind each MFMA
  auto mfma_read_step = [&](int src, int buf, int ks, int dst, bool do_read) {
    if (!wave_live) return;
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
    if (do_read) {
#pragma unroll
      for (int i = MT * NT; i < MT + NT; ++i) {
        if (i < MT) af[dst][i] = *(const bf16x8*)(base + aoff[i] + ((sl ^ aswz[i]) << 4));
        else bfr[dst][i - MT] = *(const bf16x8*)(base + boff[i - MT] + ((sl ^ bswz[i - MT]) << 4));
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };

  if (wave_live) {
    const char* base = lds_tiles;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af[0][mt] = *(const bf16x8*)(base + aoff[mt] + ((khalf ^ aswz[mt]) << 4));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bfr[0][nt] = *(const bf16x8*)(base + boff[nt] + ((khalf ^ bswz[nt]) << 4));
  }
  // one k-tile out of buffer ``cur`` (compile-time), preparing tile kt+1 from register set ``nset``
  auto k_tile = [&](int kt, auto cur_c) {
    constexpr int cur = decltype(cur_c)::value;
    constexpr int nxt = cur ^ 1;
    using nset = std::integral_constant<int, nxt>;
    const bool more = kt + 1 < KT;
#pragma unroll
    for (int ks = 0; ks + 1 < KS; ++ks) {
      mfma_read_step(ks & 1, cur, ks + 1, (ks + 1) & 1, true);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) {
      if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA * NSRC) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      put_a(kt + 1, nxt, nset{});
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // buffer ``cur`` and register set ``nxt`` are free: every wave has read its last fragments of tile kt
      if (kt + 2 < KT) stage_b(kt + 2, cur);
      if (kt + 3 < KT) load_a(kt + 3, nset{});
    }
    mfma_read_step((KS - 1) & 1, nxt, 0, 0, more);
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int kt = 0; kt < KT; kt += 2) {
    k_tile(kt, set0{});
    if (kt + 1 < KT) k_tile(kt + 1, set1{});
  }
  __syncthreads();  // all fragment reads done before the tiles are reused by the epilogue

  if (!BWD && want_mask) {
    // rows m0 .. of the bitmask are one contiguous run of bytes (16-byte aligned: m0 * K / 8 = tile_m * 16 * K)
    const int rows = p.M - m0 < BM ? p.M - m0 : BM;
    const int nbytes = rows * (K >> 3);
    unsigned char* dst = q.side_mask + (size_t)m0 * (K >> 3);
    for (int b = tid * 16; b + 16 <= nbytes; b += 256 * 16) *(u32x4*)(dst + b) = *(const u32x4*)(lds_mask + b);
    for (int b = (nbytes & ~15) + tid; b < nbytes; b += 256) dst[b] = lds_mask[b];
  }
  conv_epilogue<BM, BN, EPI>(p, acc, lds_tiles, wid, lane, m0, n0, tile_m, true);
}

template <int BN, int BK, int EPI, int PRO>
static int launch_pro(const ConvArgs& a, const ProArgs& q, hipStream_t stream) {
  constexpr int BM = 128;
  constexpr int STAGE = (BM + BN) * BK * 2;
  constexpr int EP = 4 * (BM / 2) * ((BN / 2) * 2 + 16) + 2 * (BN / 2) * 2 * 4;
  const int KP = (a.K + BK - 1) / BK * BK;
  size_t lds = (size_t)((2 * STAGE > EP) ? 2 * STAGE : EP) + (size_t)pro_ncoef(PRO) * KP * 4;
  if (PRO <= PRO_FWD_RES && q.side_mask != nullptr) lds += (size_t)BM * (a.K / 8);
  if (lds > 160 * 1024) return -8;
  unsigned mtiles = (a.M + BM - 1) / BM, ntiles = (a.Cout + BN - 1) / BN;
  // the dynamic-LDS limit is a per-device function attribute; remember the largest size granted per device
  static int attr_lds[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev = (dev >= 0 && dev < 64) ? dev : 0;
  if ((int)lds > attr_lds[dev] && lds > 48 * 1024) {
    (void)hipFuncSetAttribute((const void*)conv_pro_kernel<BN, BK, EPI, PRO>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    attr_lds[dev] = (int)lds;
  }
  hipLaunchKernelGGL((conv_pro_kernel<BN, BK, EPI, PRO>), dim3(mtiles * ntiles), dim3(256), lds, stream, a, q);
  return (int)hipGetLastError();
}

template <int BN, int BK, int PRO>
static int launch_pro_epi(const ConvArgs& a, const ProArgs& q, hipStream_t stream) {
  if (PRO >= PRO_BWD_RELU) {  // data gradient: plain store or batch-norm backward sums of the layer below
    if (a.bn_y != nullptr) return launch_pro<BN, BK, 3, PRO>(a, q, stream);
    return launch_pro<BN, BK, 0, PRO>(a, q, stream);
  }
  if (a.stats != nullptr) return launch_pro<BN, BK, 1, PRO>(a, q, stream);
  return launch_pro<BN, BK, 0, PRO>(a, q, stream);
}

template <int PRO>
static int launch_pro_tile(const ConvArgs& a, const ProArgs& q, int bn, int bk, hipStream_t stream) {
  if (bn == 64 && bk == 32) return launch_pro_epi<64, 32, PRO>(a, q, stream);
  if (bn == 64 && bk == 64) return launch_pro_epi<64, 64, PRO>(a, q, stream);
  if (bn == 128 && bk == 32) return launch_pro_epi<128, 32, PRO>(a, q, stream);
  if (bn == 128 && bk == 64) return launch_pro_epi<128, 64, PRO>(a, q, stream);
  return -3;
}

// Pointwise convolution  y[m][n] = sum_k a[m][k] * w[n][k]  with  a = prologue(x, src2, ...)  (see the top of
// the file).  ``a->x`` is the first operand ([M][K], K = a->Cin), the output is dense ([M][Cout]); the epilogue
// options of the data-gradient / statistics kernels apply (addend, addend_mask, stats, bn_*), not bias / ReLU.
extern "C" int dtfa_conv_pro(const ConvArgs* a, const ProArgs* q, int pro, int bn, int bk, hipStream_t stream) {
  if (a->Cin % 8 != 0 || a->Cout % 8 != 0 || a->T != 1 || a->K != a->Cin) return -1;
  if (!a->pointwise || !a->dense_out || a->ldo != a->Cout) return -2;
  if (a->tile_m_base != 0 || a->tile_m_count != 0 || a->stat_row_base != 0) return -2;   // one launch covers the whole layer here
  if (a->bias != nullptr || a->relu || a->oscale != nullptr) return -4;
  if (a->addend != nullptr && a->stats != nullptr && a->bn_y == nullptr) return -4;
  if (a->bn_y != nullptr) {
    if (a->stats == nullptr || a->bn_mean == nullptr || a->bn_invstd == nullptr) return -5;
    if (a->bn_relu == 2 && (a->bn_scale == nullptr || a->bn_shift == nullptr)) return -5;
    if (a->bn_relu == 3 && a->bn_mask == nullptr) return -5;
    if (a->bn_relu != 0 && a->bn_relu != 2 && a->bn_relu != 3) return -5;
  }
  if (q->c0 == nullptr || q->c1 == nullptr) return -7;
  if (pro >= PRO_BWD_RELU && (q->c2 == nullptr || q->src2 == nullptr)) return -7;
  if (pro == PRO_FWD_RES && q->src2 == nullptr) return -7;
  if (pro == PRO_BWD_RELU && (q->gscale == nullptr || q->gshift == nullptr)) return -7;
  if (pro == PRO_BWD_MASK && q->gate == nullptr) return -7;
  if (pro >= PRO_BWD_RELU && q->side_mask != nullptr) return -7;
  if (a->M <= 0) return 0;
  switch (pro) {
    case PRO_FWD: return launch_pro_tile<PRO_FWD>(*a, *q, bn, bk, stream);
    case PRO_FWD_RES: return launch_pro_tile<PRO_FWD_RES>(*a, *q, bn, bk, stream);
    case PRO_BWD_RELU: return launch_pro_tile<PRO_BWD_RELU>(*a, *q, bn, bk, stream);
    case PRO_BWD_MASK: return launch_pro_tile<PRO_BWD_MASK>(*a, *q, bn, bk, stream);
    case PRO_BWD_PLAIN: return launch_pro_tile<PRO_BWD_PLAIN>(*a, *q, bn, bk, stream);
  }
  return -3;
}
