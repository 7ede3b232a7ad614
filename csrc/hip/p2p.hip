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
#include "common.h"
#include <string.h>

#define P2P_MAX_RANKS 8
#define P2P_FLAG_BYTES 256

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

struct P2PCtx {
  int world, rank;
  size_t max_bytes;
  char* local;
  char* peers[P2P_MAX_RANKS];
  unsigned seq;
  unsigned* status;
  double timeout_s;
};

static size_t mailbox_bytes(size_t max_bytes) { return P2P_FLAG_BYTES + 2 * max_bytes; }

extern "C" int dtfa_p2p_create(int world, int rank, long long max_bytes, double timeout_s, void** out) {
  if (world < 1 || world > P2P_MAX_RANKS || rank < 0 || rank >= world || max_bytes <= 0 || max_bytes % 16) return -1;
  P2PCtx* c = new P2PCtx();
  c->world = world;
  c->rank = rank;
  c->max_bytes = (size_t)max_bytes;
  c->seq = 0;
  c->timeout_s = timeout_s > 0 ? timeout_s : 5.0;
  for (int i = 0; i < P2P_MAX_RANKS; ++i) c->peers[i] = nullptr;
  hipError_t e = hipExtMallocWithFlags((void**)&c->local, mailbox_bytes(c->max_bytes), hipDeviceMallocFinegrained);
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
  (void)hipMemset(c->local, 0, mailbox_bytes(c->max_bytes));
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
