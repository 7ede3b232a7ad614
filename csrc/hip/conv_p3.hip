// Persistent same-size 3x3 convolution over 64 (or 32) input channels for gfx950 (MI355X): weights resident in registers,
// the pixel operand staged in LDS once per tile and double-buffered.
//
// The ordinary implicit-GEMM kernel (conv_igemm.hip) serves a 3x3 layer by gathering a 128 x 64 pixel tile once per
// TAP: for the 56x56 64 -> 64 layers of ResNet-50 (forward, and the data gradient, which is the same convolution
// with flipped taps and permuted weights; slim conv2d 3x3 under recognition_flower/single/flower_train.py:67-68)
// that moves every input pixel nine times and the 73 KB weight matrix once per tile from L2 into LDS -- 11 GB per
// launch at batch 2048 over a path that delivers ~12 TB/s -- for a layer whose HBM traffic is 1.6 GB.  Staging the
// pixels once per tile alone did not help: with several workgroups per CU the weight fragments then come through the
// vector cache or are re-staged per tap, and cost as much as the pixel re-reads did.
//
// Here TWO workgroups per CU live for the whole launch, each walking a contiguous range of row tiles:
//   * a wave's share of the weight matrix (its 32 output channels x 576, as MFMA B fragments: 144 registers per
//     lane) is loaded once and stays in registers: the main loop reads only pixels from LDS;
//   * the flattened pixel range a 128-row tile touches over all nine taps, [m0 - IW - 1, m0 + 127 + IW + 1] (the
//     "patch", 128-byte rows), is brought in by LDS-DMA into one of two buffers; the patch of tile i + 1 is requested
//     before tile i is multiplied, so its HBM latency hides behind a whole tile of work.  Tap (dy, dx) reads its A
//     fragments at the row offset dy * IW + dx; rows whose tap falls outside the image point at a row of zeros;
//   * consecutive tiles of a workgroup overlap in 2 (IW + 1) patch rows, which the second request finds in L2:
//     HBM sees every input pixel about once;
//   * the epilogue (batch-norm statistics, residual-gradient add, batch-norm backward sums, bf16 rows staged through
//     LDS) is conv_epilogue of the ordinary kernel, staged in the patch buffer the tile has just finished with; its
//     stores drain while the next tile is multiplied, and it runs beside the other workgroup's multiplications.
// 4 waves per workgroup (2 x 2, each 64 x 32 of the 128 x 64 tile), two waves per SIMD with 256 registers each.
// Layers with more than 64 output channels (up to 256) run as slices of 64 columns, one workgroup each over the same
// rows; neighbours in the grid, so the second slice finds the patch rows in L2.
// Images too wide for two workgroups with two patch buffers each (beyond 87 pixels) run with ONE buffer per workgroup:
// a tile's patch is requested when the previous tile's epilogue has left the buffer and waited for, and what hides the
// round trip is the CU's other workgroup (two buffers at one workgroup per CU measured slower than the ordinary kernel).
#include "common.h"

#include "conv_epilogue.h"

#ifndef P3_DBG
#define P3_DBG 0   // timing-only builds (wrong results): bit 0 drops the tap loop, bit 1 the epilogue, bit 2 the patch loads
#endif

namespace {

constexpr int P3_BM = 128, P3_BN = 64;
constexpr int P3_EP = 4 * 64 * (32 * 2 + EP_PAD) + 2 * 32 * 2 * 4;           // epilogue row tiles + statistics exchange
constexpr int P3_HEAD = MAX_TAPS * 4 + 128;                              // (unused words) + a pixel row of zeros

// LDS-DMA of 16 bytes per lane from a wave-uniform base plus a 32-bit lane offset to lds_dst + lane * 16 (not seen by
// the compiler as an LDS write: waits are the caller's, see glds16_untracked).
__device__ __forceinline__ void glds16_sbase(const void* base, unsigned lane_off, unsigned lds_dst) {
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

// CIN: input channels, 64 or 32 (pixel rows of 128 or 64 bytes; a tap is four or two k-steps of 16).
// SLICED: more than 64 output channels (a template parameter: as a run-time value the slice offset costs the plain
// kernel the registers it does not have).
// NBUF: patch buffers per workgroup, 2 or 1 (see above; a template parameter for the same reason).
template <int EPI, int CIN, bool SLICED, int NBUF>
__global__ __launch_bounds__(256, 2) void conv_p3_kernel(ConvArgs p, int patch_rows, int tiles_per_wg, int nslices) {
  constexpr int BM = P3_BM, BN = P3_BN, ROWB = CIN * 2, WM = 64, WN = 32, MT = 2, NT = 1;
  constexpr int CPR = ROWB / 16;      // 16-byte chunks per pixel row
  constexpr int RPI = 64 / CPR;       // pixel rows per LDS-DMA wave instruction
  constexpr int NV = 16 / RPI;        // distinct source permutations of an instruction (the swizzle repeats every 16 rows)
  constexpr int KS = CIN / 16;        // k-steps per tap
  constexpr int HS = KS / 2;          // steps of two k-steps per tap
  static_assert(CIN == 64 || CIN == 32, "pixel rows of 128 or 64 bytes");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* zrow = smem + MAX_TAPS * 4;          // 128 bytes of zeros
  char* patch0 = smem + P3_HEAD;
  const int patch_bytes = patch_rows * ROWB;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const unsigned mtiles = (p.M + BM - 1) / BM;
  // More than 64 output channels: slices of 64 columns, each a workgroup of its own over the same range of row tiles
  // (neighbours in the grid, so that the second finds the patch rows the first brought into L2).
  const unsigned range = SLICED ? blockIdx.x / (unsigned)nslices : blockIdx.x;
  const int n0 = SLICED ? (int)(blockIdx.x % (unsigned)nslices) * BN : 0;
  const unsigned tile_begin = range * (unsigned)tiles_per_wg;
  unsigned tile_end = tile_begin + (unsigned)tiles_per_wg;
  if (tile_end > mtiles) tile_end = mtiles;
  if (tile_begin >= tile_end) return;
  const int margin = p.IW + 1;
  constexpr int T = 9;
  constexpr int ZROW = MAX_TAPS * 4 / ROWB;   // LDS row (units of a pixel row) of the zero row

  if (tid < 32) ((float*)zrow)[tid] = 0.f;

  // Rows of LDS are addressed in units of a pixel row from the start of the workgroup's allocation; the 16-byte chunks of
  // a row are stored permuted by swz_of_row of that absolute row index, so that 16 lanes reading the same logical
  // chunk of 16 consecutive rows touch every bank once.
  const int prow0 = P3_HEAD / ROWB;           // first row of patch buffer 0

  // ---- the patch of a tile: 8 (16) pixel rows per wave instruction, dealt over the waves ----
  // patch_rows is a multiple of 16 and so is the swizzle's period: a lane's source chunk for instruction j depends
  // on j modulo NV only, not on the buffer.  Source = uniform 64-bit base (first pixel row of the instruction)
  // + a per-lane 32-bit offset, one of NV.
  const int row_in_inst = lane / CPR;
  unsigned lane_off[NV];
#pragma unroll
  for (int par = 0; par < NV; ++par)
    lane_off[par] = row_in_inst * ROWB + (((lane % CPR) ^ swz_of_row<CIN>(prow0 + par * RPI + row_in_inst)) << 4);
  const int ninst = patch_rows / RPI;
  auto request_patch = [&](unsigned tile, int buf) {
    const int brow = prow0 + buf * patch_rows;
    const long long gp0 = (long long)tile * BM - margin;
    for (int j = wid; j < ((P3_DBG & 4) ? 0 : ninst); j += 4) {
      const long long first = gp0 + j * RPI;    // uniform
      const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((brow + j * RPI) * ROWB);
      if (first >= 0 && first + RPI <= (long long)p.M) {
        glds16_sbase(p.x + first * CIN, lane_off[j & (NV - 1)], dst);
      } else {   // the tensor's first and last rows: out-of-range rows read zeros (their taps are masked anyway)
        const long long gp = first + row_in_inst;
        const bf16* src = (gp >= 0 && gp < (long long)p.M) ? (const bf16*)((const char*)(p.x + first * CIN) + lane_off[j & (NV - 1)]) : p.zero;
        glds16_untracked(src, smem + dst);
      }
    }
  };
  if (NBUF == 2) request_patch(tile_begin, 0);   // (one buffer: every tile's patch is requested inside the loop)

  // ---- the taps, in scalar registers: row offset inside a patch and index into a lane's 3 x 3 validity mask ----
  int toff[T], tbit[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int tp = p.taps[t];
    const int dy = tp >> 16, dx = (int)(short)(tp & 0xffff);
    toff[t] = __builtin_amdgcn_readfirstlane(dy * p.IW + dx + margin);
    tbit[t] = __builtin_amdgcn_readfirstlane((dy + 1) * 3 + dx + 1);
  }

  // ---- the wave's weights: B fragments of all taps, [Cout][T * CIN] global -> registers, once ----
  const int khalf = lane >> 5;
  const bool wave_live = n0 + wc * WN < p.Cout;
  bf16x8 bw[T][KS];
  {
    const int wn = n0 + wc * WN + (lane & 31);
    const bf16* wrow = p.w + (size_t)(wn < p.Cout ? wn : 0) * p.K + khalf * 8;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (wn < p.Cout) v = *(const bf16x8*)(wrow + t * CIN + ks * 16);
        bw[t][ks] = v;
      }
  }

  // ---- per-lane rows: position in the image, advanced tile by tile without divisions ----
  const int opix = p.IH * p.IW;
  const int step_y = BM / p.IW, step_x = BM - step_y * p.IW;
  int rloc[MT], oy[MT], ox[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    rloc[mt] = wr * WM + mt * 32 + (lane & 31);
    const long long m = (long long)tile_begin * BM + rloc[mt];
    const int rem = (int)(m % opix);
    oy[mt] = rem / p.IW;
    ox[mt] = rem - oy[mt] * p.IW;
  }

  float held[2];
  epi_hold_column<EPI>(p, n0 + wc * WN + (lane & 31), held);

  wait_vmcnt<0>();   // the first patch (and the weights)
  __syncthreads();   // ... and the zero row, of every wave

  int cur = 0;
#if (P3_DBG & 8)
  long long clk[5] = {0, 0, 0, 0, 0};
#define P3_CLK(i, expr) { const long long c_ = __builtin_readcyclecounter(); clk[i] += c_ - cprev; cprev = c_; }
  long long cprev = __builtin_readcyclecounter();
#else
#define P3_CLK(i, expr)
#endif
  for (unsigned tile = tile_begin; tile < tile_end; ++tile) {
    const int m0 = (int)(tile * BM);
    // Every wave has left the previous tile's epilogue (barrier), so the buffer it was staged in -- the patch of the
    // previous tile -- may take the next request.  This tile's patch arrived before the previous epilogue (see below).
    // (A raw barrier behind a wait for the wave's own LDS reads: __syncthreads would also wait for the epilogue's
    // global stores, a memory round trip per tile.)
    if (tile != tile_begin) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (NBUF == 2) {
      if (tile + 1 < tile_end) request_patch(tile + 1, cur ^ 1);
    } else {
      // One buffer (images too wide for two patches per workgroup at two workgroups per CU): this tile's patch is
      // requested only now that the previous tile's epilogue has left the buffer, and waited for; what hides the round
      // trip is the CU's other workgroup.
      request_patch(tile, 0);
      wait_vmcnt<0>();
      __syncthreads();
    }
    // (One buffer: the buffer's row is made opaque per tile.  As a constant it lets the compiler hoist the 18 tap
    // addresses of a lane out of the tile loop, into registers the kernel does not have: 10 .. 13 spilled.)
    int curb = cur;
    if (NBUF == 1) asm volatile("" : "+s"(curb));
    const int brow = prow0 + curb * patch_rows;
    char* patch = smem + brow * ROWB;

    // validity of a lane's rows per tap: bit (dy + 1) * 3 + (dx + 1) says that (oy + dy, ox + dx) is inside the image
    unsigned okmask[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const unsigned colok = 2u | (ox[mt] > 0 ? 1u : 0u) | (ox[mt] < p.IW - 1 ? 4u : 0u);
      const unsigned up = oy[mt] > 0 ? colok : 0u, down = oy[mt] < p.IH - 1 ? colok << 6 : 0u;
      okmask[mt] = m0 + rloc[mt] < p.M ? (up | (colok << 3) | down) : 0u;
    }

    f32x16 acc[MT][NT];
    if (!wave_live || (P3_DBG & 1)) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[mt][0][j] = 0.f;
    }
    P3_CLK(0, "top barrier, patch request, masks")

    if (wave_live && !(P3_DBG & 1)) {
      // Steps of half a tap (2 k-slices x 2 row blocks: 4 fragment reads, 4 MFMAs); the fragments of step s + 1 are
      // requested before step s multiplies.
      bf16x8 af[2][2][MT];
      unsigned abase[MT];   // LDS byte address of the lane's chunk for k-slice 0 of the tap being read
      auto read_step = [&](int s, int dst) {
        const int t = s / HS, h = s % HS;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if (h == 0) {
            const bool ok = ((okmask[mt] >> tbit[t]) & 1u) != 0;
            const unsigned R = ok ? (unsigned)(brow + rloc[mt] + toff[t]) : (unsigned)ZROW;
            abase[mt] = R * (unsigned)ROWB + (((unsigned)swz_of_row<CIN>((int)R) ^ (unsigned)khalf) << 4);
          }
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2)
            af[dst][k2][mt] = *(const bf16x8*)(smem + (abase[mt] ^ (unsigned)((h * 2 + k2) << 5)));
        }
      };
      read_step(0, 0);
#pragma unroll
      for (int s = 0; s < HS * T; ++s) {
        if (s + 1 < HS * T) read_step(s + 1, (s + 1) & 1);
        // (keeps the scheduler from sinking the reads behind the multiplications they are meant to run beside)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                af[s & 1][k2][mt], bw[s / HS][(s % HS) * 2 + k2],
                (s == 0 && k2 == 0) ? f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0} : acc[mt][0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    P3_CLK(1, "tap loop")
    // advance the lanes' image positions to the next tile
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      ox[mt] += step_x;
      oy[mt] += step_y;
      if (ox[mt] >= p.IW) {
        ox[mt] -= p.IW;
        oy[mt] += 1;
      }
      while (oy[mt] >= p.IH) oy[mt] -= p.IH;
    }
    // The next patch has had a tile's worth of multiplication to arrive: wait for it here, before the epilogue's
    // stores are issued (the counter is in order: behind them the wait would also cover their round trip); the
    // barrier publishes it to the other waves and tells that every wave is done reading this tile's patch, which
    // the epilogue now overwrites.  The stores then drain under the next tile.
    wait_vmcnt<0>();
    P3_CLK(2, "wait for the next patch")
    __syncthreads();
    P3_CLK(3, "barrier")
#if (P3_DBG & 2)
    {
      float sum = 0.f;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 16; ++j) sum += acc[mt][0][j];
      if (sum == 123.456f) p.y[0] = (bf16)sum;
    }
#else
    conv_epilogue<BM, BN, EPI, false>(p, acc, patch, wid, lane, m0, n0, tile, true, nullptr, held);
#endif
    P3_CLK(4, "epilogue")
    if (NBUF == 2) cur ^= 1;
  }
#if (P3_DBG & 8)
  if (lane == 0)
    for (int i = 0; i < 5; ++i) ((long long*)p.oscale)[(blockIdx.x * 4 + wid) * 5 + i] = clk[i];
#endif
}

template <int EPI, int CIN, bool SLICED, int NBUF>
int launch_p3(const ConvArgs& a, int patch_rows, hipStream_t stream) {
  constexpr int ROWB = CIN * 2;
  const size_t lds = P3_HEAD + NBUF * (size_t)patch_rows * ROWB;
  static bool attr_set[64] = {};
  static int cus[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev = (dev >= 0 && dev < 64) ? dev : 0;
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute((const void*)conv_p3_kernel<EPI, CIN, SLICED, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set[dev] = true;
  }
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
  }
  const unsigned mtiles = (a.M + P3_BM - 1) / P3_BM;
  // as many workgroups as fit on the chip at once (two per CU), each a
  // contiguous range of row tiles (neighbouring tiles share patch rows through L2)
  const unsigned nslices = (unsigned)(a.Cout + P3_BN - 1) / P3_BN;
  const unsigned slots = cus[dev] * 2u / nslices;
  const unsigned per = (mtiles + slots - 1) / slots;
  const unsigned wgs = (mtiles + per - 1) / per;
  hipLaunchKernelGGL((conv_p3_kernel<EPI, CIN, SLICED, NBUF>), dim3(wgs * nslices), dim3(256), lds, stream, a, patch_rows,
                     (int)per, (int)nslices);
  return (int)hipGetLastError();
}

// The shapes the kernel takes (the caller vouches that every tap offset lies within one pixel).
bool p3_ok(const ConvArgs& a) {
  return (a.Cin == 64 || a.Cin == 32) && a.Cout <= 256 && a.T == 9 && a.K == 9 * a.Cin && a.isy == 1 && a.isx == 1 && a.osy == 1 &&
         a.osx == 1 && a.oy0 == 0 && a.ox0 == 0 && a.dense_out && !a.pointwise && a.x2 == nullptr && a.IH == a.OYn &&
         a.IW == a.OXn && a.IH == a.OH && a.IW == a.OW && a.tile_m_base == 0 && a.tile_m_count == 0 &&
         a.stat_row_base == 0 && a.stat_acc == nullptr && a.M % (a.IH * a.IW) == 0;
}

}  // namespace

int conv_p3_launch(const ConvArgs& a, hipStream_t stream) {
  if (!p3_ok(a)) return -10;
  // The data gradient that also emits batch-norm backward sums stays with the ordinary kernel: its epilogue reads
  // a second tensor, and with the weights holding 144 registers there are neither registers to keep enough of those
  // loads in flight nor a second workgroup's worth of waves to hide them (measured: 1.33 ms against 1.18 ms).
  if (a.bn_y != nullptr || a.bn_gate_store) return -10;
  const int epi = a.stats != nullptr ? 1 : (a.bias != nullptr || a.relu || a.oscale != nullptr || a.ldo != a.Cout) ? 2 : 0;
  const bool sliced = a.Cout > P3_BN;
  const int rowb = a.Cin * 2;
  int patch_rows = (P3_BM + 2 * (a.IW + 1) + 15) & ~15;   // (a multiple of 16: see request_patch)
  if (patch_rows * rowb < P3_EP) patch_rows = ((P3_EP + rowb - 1) / rowb + 15) & ~15;
  // two patch buffers where two workgroups with two each fit into a CU's LDS, else one buffer per workgroup
  const int nbuf = P3_HEAD + 2 * (size_t)patch_rows * rowb <= 80 * 1024 ? 2 : 1;
  if (P3_HEAD + nbuf * (size_t)patch_rows * rowb > 80 * 1024) return -10;
#define P3_CASE(E, C, S) \
  if (epi == E && a.Cin == C && sliced == S) \
    return nbuf == 2 ? launch_p3<E, C, S, 2>(a, patch_rows, stream) : launch_p3<E, C, S, 1>(a, patch_rows, stream);
  P3_CASE(0, 64, false) P3_CASE(1, 64, false) P3_CASE(2, 64, false)
  P3_CASE(0, 64, true) P3_CASE(1, 64, true) P3_CASE(2, 64, true)
  P3_CASE(0, 32, false) P3_CASE(1, 32, false) P3_CASE(2, 32, false)
  P3_CASE(0, 32, true) P3_CASE(1, 32, true) P3_CASE(2, 32, true)
#undef P3_CASE
  return -10;
}
