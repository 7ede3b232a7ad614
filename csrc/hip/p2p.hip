// One-shot peer-to-peer all-reduce over xGMI for small buffers.
//
// A ring collective pays 2(N-1) latency-bound steps; for a buffer of a few hundred kilobytes the
// fully connected xGMI mesh allows something simpler: every rank publishes its data in a mailbox
// that all peers have mapped (hipIpc), raises a flag in every peer's mailbox, waits for the N flags
// in its own mailbox, then reads all N mailboxes directly over the links and sums them in rank order.
// Every rank performs the same additions in the same order, so all replicas get bit-identical
// results.  One kernel, one round of flag traffic, N-1 remote reads per element.
//
// Mailbox layout (fine-grained device memory, so stores become visible to peers without waiting for
// the end of the kernel):
//   [0, 256)                     flags[2][8] : uint32 sequence numbers, indexed [parity][source rank]
//   [256, 256 + max_bytes)       data, parity 0
//   [256 + max_bytes, ...)       data, parity 1
// Calls alternate between the two parities.  A rank can start call n+2 (same parity as n) only after
// it completed call n+1, which needs every rank's n+1 flag, which a rank raises only after it has
// finished reading the data of call n -- so a data area is never overwritten while a peer reads it.
//
// The flag wait is bounded by a wall-clock timeout: a missing peer makes the kernel give up, set a
// status word and leave the output untouched instead of spinning forever.
//
// Two-shot variant for buffers of megabytes (the 8.4 MB of head gradients of the reference's model): the
// one-shot form moves (N-1) x size into every rank.  Here rank r reduces only slice r (reading that slice from
// every mailbox, one link per peer, all links at once), publishes the reduced slice in a result area of its own
// mailbox, raises a second flag at every peer, and then every rank collects the N reduced slices -- again one per
// link.  Each rank reads 2 (N-1)/N x size in total, a direct reduce-scatter followed by a direct all-gather on
// the fully connected mesh; the sums are formed in rank order by one rank per slice, so all replicas still hold
// identical bits, and they are the bits of the one-shot form.
//   [256, 512)                   flags2[2][8] : slice-reduced sequence numbers, indexed [parity][source rank]
//   behind the two data areas:   result areas, parity 0 / 1, each ceil(max_bytes / world) bytes
#include "common.h"
#include <string.h>

#define P2P_MAX_RANKS 8
#define P2P_FLAG_BYTES 512
#define P2P_FLAG2_OFF 256

struct P2PKernelArgs {
  const char* data[P2P_MAX_RANKS];   // this call's data area of every rank (own included)
  unsigned* peer_flag[P2P_MAX_RANKS];  // slot [parity][my rank] in every rank's mailbox
  const unsigned* my_flags;          // slots [parity][0..world) of the own mailbox
  unsigned* status;
  long long timeout_ticks;           // wall_clock64 ticks (100 MHz)
  size_t nvec;                       // 16-byte vectors
  int world;
  unsigned seq;
  float scale;
};

template <bool BF16>
__global__ __launch_bounds__(256) void p2p_allreduce_kernel(P2PKernelArgs a, void* __restrict__ out) {
  __shared__ int ok;
  if (threadIdx.x == 0) ok = 1;
  if (blockIdx.x == 0 && threadIdx.x < a.world)
    __hip_atomic_store(a.peer_flag[threadIdx.x], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  __syncthreads();
  if (threadIdx.x < a.world) {
    const long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(a.my_flags + threadIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - a.seq) < 0) {
      __builtin_amdgcn_s_sleep(16);
      if (wall_clock64() - t0 > a.timeout_ticks) {
        ok = 0;
        __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
  __syncthreads();
  if (!ok) return;
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.nvec; i += (size_t)gridDim.x * blockDim.x) {
    if (BF16) {
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int r = 0; r < a.world; ++r) {
        bf16x8 v = *(const bf16x8*)(a.data[r] + i * 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
      }
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16)(acc[j] * a.scale);
      *(bf16x8*)((char*)out + i * 16) = o;
    } else {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int r = 0; r < a.world; ++r) {
        float4 v = *(const float4*)(a.data[r] + i * 16);
        acc.x += v.x;
        acc.y += v.y;
        acc.z += v.z;
        acc.w += v.w;
      }
      acc.x *= a.scale;
      acc.y *= a.scale;
      acc.z *= a.scale;
      acc.w *= a.scale;
      *(float4*)((char*)out + i * 16) = acc;
    }
  }
}

struct P2PKernelArgs2 {
  const char* data[P2P_MAX_RANKS];     // this call's data area of every rank
  char* res[P2P_MAX_RANKS];            // this call's result area of every rank (slice r lives in rank r's)
  unsigned* peer_flag[P2P_MAX_RANKS];  // round 1: slot [parity][my rank] in every rank's mailbox
  unsigned* peer_flag2[P2P_MAX_RANKS]; // round 2
  const unsigned* my_flags;
  const unsigned* my_flags2;
  unsigned* status;
  unsigned* counter;                   // blocks of this launch that have written their part of the slice (zeroed by the host)
  long long timeout_ticks;
  size_t nvec;                         // 16-byte vectors in the buffer
  size_t per;                          // vectors per slice = ceil(nvec / world)
  int world, rank;
  unsigned seq;
  float scale;
};

// Wait until all ``world`` sequence words reach ``seq``; false (and the status word set) after the timeout.
__device__ __forceinline__ bool p2p_wait_flags(const unsigned* flags, int world, unsigned seq, long long timeout_ticks,
                                               unsigned* status, int* ok) {
  if (threadIdx.x < (unsigned)world) {
    const long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(flags + threadIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
      __builtin_amdgcn_s_sleep(16);
      if (wall_clock64() - t0 > timeout_ticks) {
        *ok = 0;
        __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
  __syncthreads();
  return *ok != 0;
}

template <bool BF16>
__global__ __launch_bounds__(256) void p2p_allreduce2_kernel(P2PKernelArgs2 a, void* __restrict__ out) {
  __shared__ int ok;
  __shared__ unsigned last;
  if (threadIdx.x == 0) ok = 1;
  if (blockIdx.x == 0 && threadIdx.x < a.world)
    __hip_atomic_store(a.peer_flag[threadIdx.x], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  __syncthreads();
  if (!p2p_wait_flags(a.my_flags, a.world, a.seq, a.timeout_ticks, a.status, &ok)) return;
  __atomic_thread_fence(__ATOMIC_ACQUIRE);

  // ---- reduce the slice this rank owns, in rank order, into the own result area ----
  const size_t lo = a.per * (size_t)a.rank < a.nvec ? a.per * (size_t)a.rank : a.nvec;
  const size_t hi = lo + a.per < a.nvec ? lo + a.per : a.nvec;
  char* mine = a.res[a.rank];
  for (size_t i = lo + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += (size_t)gridDim.x * blockDim.x) {
    if (BF16) {
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int r = 0; r < a.world; ++r) {
        bf16x8 v = *(const bf16x8*)(a.data[r] + i * 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
      }
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16)(acc[j] * a.scale);
      *(bf16x8*)(mine + (i - lo) * 16) = o;
    } else {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int r = 0; r < a.world; ++r) {
        float4 v = *(const float4*)(a.data[r] + i * 16);
        acc.x += v.x;
        acc.y += v.y;
        acc.z += v.z;
        acc.w += v.w;
      }
      acc.x *= a.scale;
      acc.y *= a.scale;
      acc.z *= a.scale;
      acc.w *= a.scale;
      *(float4*)(mine + (i - lo) * 16) = acc;
    }
  }
  // every block's part of the slice is visible system-wide before the block checks in; the block that checks in
  // last tells the peers
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(a.counter, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (last) {
    __threadfence_system();
    if (threadIdx.x < (unsigned)a.world)
      __hip_atomic_store(a.peer_flag2[threadIdx.x], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }

  // ---- collect the reduced slices ----
  if (!p2p_wait_flags(a.my_flags2, a.world, a.seq, a.timeout_ticks, a.status, &ok)) return;
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.nvec; i += (size_t)gridDim.x * blockDim.x) {
    const size_t owner = i / a.per;
    *(float4*)((char*)out + i * 16) = *(const float4*)(a.res[owner] + (i - owner * a.per) * 16);
  }
}

struct P2PCtx {
  int world, rank;
  size_t max_bytes;
  char* local;
  char* peers[P2P_MAX_RANKS];
  unsigned seq;
  unsigned* status;      // [0]: timeout seen; [16]: block counter of the two-shot kernel
  double timeout_s;
};

// bytes of one result area: the largest slice, ceil(max vectors / world) vectors
static size_t slice_cap(size_t max_bytes, int world) { return ((max_bytes / 16 + world - 1) / world) * 16; }
static size_t mailbox_bytes(size_t max_bytes, int world) {
  return P2P_FLAG_BYTES + 2 * max_bytes + 2 * slice_cap(max_bytes, world);
}

extern "C" int dtfa_p2p_create(int world, int rank, long long max_bytes, double timeout_s, void** out) {
  if (world < 1 || world > P2P_MAX_RANKS || rank < 0 || rank >= world || max_bytes <= 0 || max_bytes % 16) return -1;
  P2PCtx* c = new P2PCtx();
  c->world = world;
  c->rank = rank;
  c->max_bytes = (size_t)max_bytes;
  c->seq = 0;
  c->timeout_s = timeout_s > 0 ? timeout_s : 5.0;
  for (int i = 0; i < P2P_MAX_RANKS; ++i) c->peers[i] = nullptr;
  hipError_t e = hipExtMallocWithFlags((void**)&c->local, mailbox_bytes(c->max_bytes, c->world), hipDeviceMallocFinegrained);
  if (e != hipSuccess) {
    delete c;
    return (int)e;
  }
  e = hipExtMallocWithFlags((void**)&c->status, 256, hipDeviceMallocFinegrained);
  if (e != hipSuccess) {
    (void)hipFree(c->local);
    delete c;
    return (int)e;
  }
  (void)hipMemset(c->local, 0, mailbox_bytes(c->max_bytes, c->world));
  (void)hipMemset(c->status, 0, 256);
  (void)hipDeviceSynchronize();
  c->peers[rank] = c->local;
  *out = c;
  return 0;
}

extern "C" int dtfa_p2p_handle(void* ctx, char* handle64) {
  P2PCtx* c = (P2PCtx*)ctx;
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, c->local);
  if (e != hipSuccess) return (int)e;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  memcpy(handle64, &h, 64);
  return 0;
}

// handles: world x 64 bytes, in rank order (the own entry is ignored).
extern "C" int dtfa_p2p_connect(void* ctx, const char* handles) {
  P2PCtx* c = (P2PCtx*)ctx;
  for (int r = 0; r < c->world; ++r) {
    if (r == c->rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, handles + (size_t)r * 64, 64);
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return (int)e;
    c->peers[r] = (char*)p;
  }
  return 0;
}

// In-place capable (out may equal in).  dtype: 0 = fp32, 1 = bf16.  out = scale * sum over ranks.
extern "C" int dtfa_p2p_allreduce(void* ctx, const void* in, void* out, long long nbytes, int dtype, float scale,
                                  hipStream_t st) {
  P2PCtx* c = (P2PCtx*)ctx;
  if (nbytes <= 0) return 0;
  if ((size_t)nbytes > c->max_bytes || nbytes % 16 || ((uintptr_t)in | (uintptr_t)out) % 16) return -1;
  for (int r = 0; r < c->world; ++r)
    if (c->peers[r] == nullptr) return -2;
  c->seq += 1;
  const unsigned parity = c->seq & 1u;
  const size_t data_off = P2P_FLAG_BYTES + (size_t)parity * c->max_bytes;
  hipError_t e = hipMemcpyAsync(c->local + data_off, in, (size_t)nbytes, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) return (int)e;
  P2PKernelArgs a;
  for (int r = 0; r < P2P_MAX_RANKS; ++r) {
    a.data[r] = r < c->world ? c->peers[r] + data_off : nullptr;
    a.peer_flag[r] = r < c->world ? (unsigned*)c->peers[r] + parity * P2P_MAX_RANKS + c->rank : nullptr;
  }
  a.my_flags = (const unsigned*)c->local + parity * P2P_MAX_RANKS;
  a.status = c->status;
  a.timeout_ticks = (long long)(c->timeout_s * 1e8);
  a.nvec = (size_t)nbytes / 16;
  a.world = c->world;
  a.seq = c->seq;
  a.scale = scale;
  unsigned blocks = (unsigned)((a.nvec + 255) / 256);
  if (blocks > 64) blocks = 64;
  if (dtype == 1)
    hipLaunchKernelGGL(p2p_allreduce_kernel<true>, dim3(blocks), dim3(256), 0, st, a, out);
  else
    hipLaunchKernelGGL(p2p_allreduce_kernel<false>, dim3(blocks), dim3(256), 0, st, a, out);
  return (int)hipGetLastError();
}

// Two-shot form (direct reduce-scatter + direct all-gather), same contract and same result bits as
// dtfa_p2p_allreduce.
extern "C" int dtfa_p2p_allreduce2(void* ctx, const void* in, void* out, long long nbytes, int dtype, float scale,
                                   hipStream_t st) {
  P2PCtx* c = (P2PCtx*)ctx;
  if (nbytes <= 0) return 0;
  if ((size_t)nbytes > c->max_bytes || nbytes % 16 || ((uintptr_t)in | (uintptr_t)out) % 16) return -1;
  for (int r = 0; r < c->world; ++r)
    if (c->peers[r] == nullptr) return -2;
  c->seq += 1;
  const unsigned parity = c->seq & 1u;
  const size_t data_off = P2P_FLAG_BYTES + (size_t)parity * c->max_bytes;
  const size_t res_off = P2P_FLAG_BYTES + 2 * c->max_bytes + (size_t)parity * slice_cap(c->max_bytes, c->world);
  hipError_t e = hipMemcpyAsync(c->local + data_off, in, (size_t)nbytes, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) return (int)e;
  e = hipMemsetAsync(c->status + 16, 0, 4, st);
  if (e != hipSuccess) return (int)e;
  P2PKernelArgs2 a;
  for (int r = 0; r < P2P_MAX_RANKS; ++r) {
    const bool live = r < c->world;
    a.data[r] = live ? c->peers[r] + data_off : nullptr;
    a.res[r] = live ? c->peers[r] + res_off : nullptr;
    a.peer_flag[r] = live ? (unsigned*)c->peers[r] + parity * P2P_MAX_RANKS + c->rank : nullptr;
    a.peer_flag2[r] = live ? (unsigned*)(c->peers[r] + P2P_FLAG2_OFF) + parity * P2P_MAX_RANKS + c->rank : nullptr;
  }
  a.my_flags = (const unsigned*)c->local + parity * P2P_MAX_RANKS;
  a.my_flags2 = (const unsigned*)(c->local + P2P_FLAG2_OFF) + parity * P2P_MAX_RANKS;
  a.status = c->status;
  a.counter = c->status + 16;
  a.timeout_ticks = (long long)(c->timeout_s * 1e8);
  a.nvec = (size_t)nbytes / 16;
  a.per = (a.nvec + c->world - 1) / c->world;
  a.world = c->world;
  a.rank = c->rank;
  a.seq = c->seq;
  a.scale = scale;
  // every block spins on the peers' flags, so the grid stays far below what the chip holds at once
  unsigned blocks = (unsigned)((a.nvec + 1023) / 1024);
  if (blocks < 1) blocks = 1;
  if (blocks > 128) blocks = 128;
  if (dtype == 1)
    hipLaunchKernelGGL(p2p_allreduce2_kernel<true>, dim3(blocks), dim3(256), 0, st, a, out);
  else
    hipLaunchKernelGGL(p2p_allreduce2_kernel<false>, dim3(blocks), dim3(256), 0, st, a, out);
  return (int)hipGetLastError();
}

// Nonzero when a flag wait timed out since creation (synchronises the device).
extern "C" int dtfa_p2p_status(void* ctx) {
  P2PCtx* c = (P2PCtx*)ctx;
  unsigned v = 0;
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(&v, c->status, 4, hipMemcpyDeviceToHost);
  return (int)v;
}

extern "C" void dtfa_p2p_destroy(void* ctx) {
  P2PCtx* c = (P2PCtx*)ctx;
  if (c == nullptr) return;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < c->world; ++r)
    if (r != c->rank && c->peers[r] != nullptr) (void)hipIpcCloseMemHandle(c->peers[r]);
  (void)hipFree(c->local);
  (void)hipFree(c->status);
  delete c;
}
