// GPU-resident asynchronous parameter server over one-sided peer memory (xGMI), gfx950.
//
// The reference trains with TensorFlow's asynchronous parameter server: variables and ApplyAdam live on
// the ps, every worker pulls the variables, computes a gradient and pushes it, nobody waits for anybody
// (recognition_flower/distributed/train_dataset.py:141-149,162-165,172-174).  On an xGMI node the ps is not
// a process: rank r OWNS shard r (1/N) of the flat parameter space -- fp32 master, optimizer slots -- in
// fine-grained device memory that every peer has mapped through hipIpc.  There is no collective and no host
// on the data path:
//
//   push   a worker copies shard o of its gradient straight into a mailbox slot in owner o's memory
//          (one slot pair per source rank) and then raises a sequence word there;
//   apply  the owner's kernel consumes whatever has arrived -- fused Adam / momentum, one optimizer step per
//          arrived gradient (asynchronous, as ApplyAdam on a ps) or one step on the rank-ordered sum of a
//          round (lock-step) -- and publishes a new parameter version;
//   pull   a worker reads the current version of every shard directly from its owner's memory;
//   gate   a worker whose clock is more than `staleness` steps ahead of the slowest live worker waits,
//          reading the peers' clock words; a peer that does not move for `timeout` stops counting.
//
// Memory of one rank (one region per flat parameter group):
//   [0, 256)      control words (XpsCtrl)
//   params        2 x piece fp32: parameter versions v (buffer v & 1), written by the owner only
//   slots         nslot x piece fp32: Adam m, v / momentum buffer
//   mail          world x 2 x piece gradients (fp32 or bf16): [source rank][push parity]
// Sequence words only ever grow.  A parameter buffer is read under a sequence lock: the owner raises `wseq`
// before it starts writing version n into buffer n & 1 and `version` after; a reader that copied buffer
// v & 1 and then finds wseq >= v + 2 read a buffer that was being overwritten and retries.
//
// Every wait in here is bounded by a wall-clock timeout and runs in a single 64-lane workgroup.
#include "common.h"
#include "opt_update.h"
#include <string.h>

#define XPS_MAX_RANKS 8
#define XPS_SLOTS 2
#define XPS_CTRL_BYTES 256
#define XPS_MAX_PLAN 16

struct XpsCtrl {
  unsigned clock;     // steps this rank has completed as a worker
  unsigned done;      // nonzero once the rank has left the job
  unsigned version;   // parameter versions published by this owner
  unsigned wseq;      // version whose write has begun
  unsigned applied;   // optimizer steps applied to this shard
  unsigned pad[11];
  unsigned seq_in[XPS_MAX_RANKS];    // [source]: gradient pushes completed by that source
  unsigned seq_done[XPS_MAX_RANKS];  // [source]: pushes of that source consumed by the owner
};
static_assert(sizeof(XpsCtrl) <= XPS_CTRL_BYTES, "control block");

// private (not shared) device words of a rank
struct XpsPriv {
  unsigned dead[XPS_MAX_RANKS];        // peers that stopped moving
  unsigned dead_clock[XPS_MAX_RANKS];  // their clock when they were written off (a change revives them)
  unsigned skip[XPS_MAX_RANKS];        // push: owners whose mailbox slot is not free (gradient dropped)
  unsigned plan_n;                     // apply: entries planned
  unsigned plan_src[XPS_MAX_PLAN];
  unsigned plan_seq[XPS_MAX_PLAN];
  unsigned plan_steps;                 // optimizer steps the plan amounts to
  float plan_scale;                    // gradient scale of an aggregated round
  unsigned push_seq;                   // pushes this rank has made
  unsigned block_counter;              // last-block detection of the push kernel
  unsigned status;                     // bit 0: a peer timed out, bit 1: push dropped, bit 2: torn pull kept
  unsigned torn_pulls, dropped_pushes, timeouts;
};

struct XpsPtrs {
  char* base[XPS_MAX_RANKS];  // region of every rank (own included)
};

__device__ __forceinline__ unsigned ld_sys(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys(unsigned* p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ XpsCtrl* ctrl_of(const XpsPtrs& P, int r) { return (XpsCtrl*)P.base[r]; }

// A peer counts for a bound while it is not done and not written off.  A written-off peer whose clock has
// moved since is alive again.
__device__ __forceinline__ bool peer_counts(XpsPriv* pv, const XpsCtrl* c, int j, unsigned cj) {
  if (ld_sys(&c->done)) return false;
  if (pv->dead[j]) {
    if (cj == pv->dead_clock[j]) return false;
    pv->dead[j] = 0;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------- gate
// Wait while my clock is more than `staleness` ahead of a live peer.  An owner advances its clock right after it
// has published the parameter version of that step, so with staleness 0 (lock-step rounds) passing the gate also
// means that every live owner has applied the previous round.
__global__ __launch_bounds__(64) void xps_gate_kernel(XpsPtrs P, XpsPriv* pv, int world, int rank, int staleness,
                                                      long long timeout_ticks) {
  const int j = threadIdx.x;
  if (j >= world || j == rank) return;
  const unsigned my = ld_sys(&ctrl_of(P, rank)->clock);
  const XpsCtrl* c = ctrl_of(P, j);
  const long long t0 = wall_clock64();
  for (;;) {
    const unsigned cj = ld_sys(&c->clock);
    if (!peer_counts(pv, c, j, cj)) return;
    if ((int)(my - cj) <= staleness) return;
    __builtin_amdgcn_s_sleep(32);
    if (wall_clock64() - t0 > timeout_ticks) {
      pv->dead[j] = 1;
      pv->dead_clock[j] = cj;
      atomicOr(&pv->status, 1u);
      atomicAdd(&pv->timeouts, 1u);
      return;
    }
  }
}

// ---------------------------------------------------------------------------------------------------- pull
// dst (compute dtype) and / or dst32 (fp32) receive world * piece values: shard o from owner o's current version.
template <bool BF16>
__global__ __launch_bounds__(256) void xps_pull_kernel(XpsPtrs P, XpsPriv* pv, int world, size_t piece, size_t off_params,
                                                       void* __restrict__ dst, float* __restrict__ dst32,
                                                       unsigned blocks_per_owner) {
  const int o = blockIdx.x / blocks_per_owner;
  const unsigned b = blockIdx.x % blocks_per_owner;
  if (o >= world) return;
  const XpsCtrl* c = ctrl_of(P, o);
  const size_t nvec = piece / 4;
  __shared__ unsigned sv;
  for (int attempt = 0; attempt < 4; ++attempt) {
    if (threadIdx.x == 0) sv = ld_sys(&c->version);
    __syncthreads();
    const unsigned v = sv;
    const float* src = (const float*)(P.base[o] + off_params) + (size_t)(v & 1u) * piece;
    for (size_t i = (size_t)b * blockDim.x + threadIdx.x; i < nvec; i += (size_t)blocks_per_owner * blockDim.x) {
      const float4 w = *(const float4*)(src + i * 4);
      const size_t e = (size_t)o * piece + i * 4;
      if (dst32) *(float4*)(dst32 + e) = w;
      if (dst) {
        if (BF16) {
          bf16x4 q = {(bf16)w.x, (bf16)w.y, (bf16)w.z, (bf16)w.w};
          *(bf16x4*)((bf16*)dst + e) = q;
        } else {
          *(float4*)((float*)dst + e) = w;
        }
      }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    __syncthreads();
    if (threadIdx.x == 0) sv = ld_sys(&c->wseq);
    __syncthreads();
    const bool torn = (int)(sv - (v + 2u)) >= 0;
    __syncthreads();
    if (!torn) return;
    if (attempt == 3 && threadIdx.x == 0) {  // keep what we have (the reference's reads are unsynchronised too)
      atomicAdd(&pv->torn_pulls, 1u);
      atomicOr(&pv->status, 4u);
    }
  }
}

// ---------------------------------------------------------------------------------------------------- push
// Before a push: lane o waits until owner o has consumed this rank's push k - 2 (the slot push k reuses).
__global__ __launch_bounds__(64) void xps_push_wait_kernel(XpsPtrs P, XpsPriv* pv, int world, int rank,
                                                           long long timeout_ticks) {
  const int o = threadIdx.x;
  if (o >= world) return;
  const unsigned k = pv->push_seq + 1u;
  pv->skip[o] = 0;
  if (k <= XPS_SLOTS) return;
  const XpsCtrl* c = ctrl_of(P, o);
  const long long t0 = wall_clock64();
  while ((int)(ld_sys(&c->seq_done[rank]) - (k - XPS_SLOTS)) < 0) {
    // an owner that has left or was written off is not consuming: its gradient shard is dropped at once
    bool gone = o != rank && !peer_counts(pv, c, o, ld_sys(&c->clock));
    if (!gone) {
      __builtin_amdgcn_s_sleep(32);
      if (wall_clock64() - t0 > timeout_ticks) {
        pv->dead[o] = 1;
        pv->dead_clock[o] = ld_sys(&c->clock);
        atomicAdd(&pv->timeouts, 1u);
        atomicOr(&pv->status, 1u);
        gone = true;
      }
    }
    if (gone) {
      pv->skip[o] = 1;
      atomicOr(&pv->status, 2u);
      atomicAdd(&pv->dropped_pushes, 1u);
      return;
    }
  }
}

template <typename G>
__global__ __launch_bounds__(256) void xps_push_kernel(XpsPtrs P, XpsPriv* pv, int world, int rank, size_t piece,
                                                       size_t off_mail, const G* __restrict__ grad,
                                                       unsigned blocks_per_owner) {
  const int o = blockIdx.x / blocks_per_owner;
  const unsigned b = blockIdx.x % blocks_per_owner;
  const unsigned k = pv->push_seq + 1u;
  if (o < world && !pv->skip[o]) {
    constexpr int VE = 16 / sizeof(G);  // elements per 16-byte vector
    const size_t nvec = piece / VE;
    G* dst = (G*)(P.base[o] + off_mail) + ((size_t)rank * XPS_SLOTS + (k & 1u)) * piece;
    const G* src = grad + (size_t)o * piece;
    for (size_t i = (size_t)b * blockDim.x + threadIdx.x; i < nvec; i += (size_t)blocks_per_owner * blockDim.x)
      *(float4*)(dst + i * VE) = *(const float4*)(src + i * VE);
  }
  // every block's stores are complete and visible system-wide before the block checks in; the block that
  // checks in last raises the sequence words
  __threadfence_system();
  __syncthreads();
  __shared__ unsigned last;
  if (threadIdx.x == 0) last = atomicAdd(&pv->block_counter, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (!last) return;
  __threadfence_system();
  if (threadIdx.x < (unsigned)world && !pv->skip[threadIdx.x]) st_sys(&ctrl_of(P, threadIdx.x)->seq_in[rank], k);
  __syncthreads();
  if (threadIdx.x == 0) {
    pv->push_seq = k;
    pv->block_counter = 0;
  }
}

// ---------------------------------------------------------------------------------------------------- apply
// Decide what the owner applies now.
//   aggregate = 1: the gradients of round c = own clock (push number c + 1 of every live source), waited for;
//                  one optimizer step on their rank-ordered sum scaled by 1 / (sources included).
//   aggregate = 0: every push that has arrived and is not consumed yet, one optimizer step each, sources in
//                  rank order, oldest first.  No waiting.
__global__ __launch_bounds__(64) void xps_collect_kernel(XpsPtrs P, XpsPriv* pv, int world, int rank, int aggregate,
                                                         long long timeout_ticks) {
  XpsCtrl* me = ctrl_of(P, rank);
  const int j = threadIdx.x;
  __shared__ unsigned have[XPS_MAX_RANKS];
  if (j < world) {
    unsigned in = ld_sys(&me->seq_in[j]);
    if (aggregate) {
      const unsigned need = ld_sys(&me->clock) + 1u;
      const XpsCtrl* c = ctrl_of(P, j);
      const long long t0 = wall_clock64();
      bool counts = true;
      while ((int)(in - need) < 0) {
        if (j != rank && !peer_counts(pv, c, j, ld_sys(&c->clock))) { counts = false; break; }
        __builtin_amdgcn_s_sleep(32);
        if (wall_clock64() - t0 > timeout_ticks) {
          pv->dead[j] = 1;
          pv->dead_clock[j] = ld_sys(&c->clock);
          atomicOr(&pv->status, 1u);
          atomicAdd(&pv->timeouts, 1u);
          counts = false;
          break;
        }
        in = ld_sys(&me->seq_in[j]);
      }
      have[j] = (counts && (int)(in - need) >= 0) ? need : 0u;
    } else {
      have[j] = in;
    }
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  unsigned n = 0;
  if (aggregate) {
    for (int s = 0; s < world; ++s)
      if (have[s]) {
        pv->plan_src[n] = s;
        pv->plan_seq[n] = have[s];
        ++n;
      }
    pv->plan_steps = n ? 1u : 0u;
    pv->plan_scale = n ? 1.0f / (float)n : 0.f;
  } else {
    for (int s = 0; s < world; ++s) {
      const unsigned done = me->seq_done[s];
      for (unsigned q = done + 1u; (int)(have[s] - q) >= 0 && n < XPS_MAX_PLAN; ++q) {
        pv->plan_src[n] = s;
        pv->plan_seq[n] = q;
        ++n;
      }
    }
    pv->plan_steps = n;
    pv->plan_scale = 1.0f;
  }
  pv->plan_n = n;
  if (n) {
    __atomic_thread_fence(__ATOMIC_ACQUIRE);       // the gradients behind the sequence words
    st_sys(&me->wseq, me->version + 1u);           // readers of the buffer about to be overwritten retry
  }
}

struct XpsOpt {
  int kind;            // 0 = Adam (TensorFlow-1 ApplyAdam), 1 = momentum
  float b1, b2, eps;   // Adam
  float mu, wd;        // momentum
  int nesterov;
  int host_step_size;  // Adam: hyper[0] already holds lr_t for this step (lock-step rounds, one step per call)
};

// hyper[0] = learning rate (or Adam's bias-corrected step size when host_step_size), hyper[1] = gradient scale.
template <typename G>
__global__ __launch_bounds__(256) void xps_apply_kernel(char* base, const XpsPriv* pv, size_t piece, size_t off_params,
                                                        size_t off_slots, size_t off_mail, XpsOpt opt,
                                                        const float* __restrict__ hyper) {
  const unsigned n = pv->plan_n;
  if (n == 0) return;
  XpsCtrl* me = (XpsCtrl*)base;
  const unsigned cur = me->version & 1u;
  const unsigned t0 = me->applied;
  const float* wsrc = (const float*)(base + off_params) + (size_t)cur * piece;
  float* wdst = (float*)(base + off_params) + (size_t)(cur ^ 1u) * piece;
  float* s0 = (float*)(base + off_slots);
  float* s1 = s0 + piece;
  const G* mail = (const G*)(base + off_mail);
  const float lr = hyper[0], gs = hyper[1] * pv->plan_scale;
  const bool sum_round = pv->plan_steps == 1u && n > 1u;
  // per-step Adam step sizes (bias correction folded in, as ApplyAdam)
  float lr_t[XPS_MAX_PLAN];
  if (opt.kind == 0) {
    for (unsigned e = 0; e < pv->plan_steps; ++e) {
      const float t = (float)(t0 + e + 1u);
      lr_t[e] = opt.host_step_size ? lr : lr * sqrtf(1.f - powf(opt.b2, t)) / (1.f - powf(opt.b1, t));
    }
  }
  const size_t nvec = piece / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    float4 w4 = *(const float4*)(wsrc + i * 4);
    float4 a4 = *(const float4*)(s0 + i * 4);
    float4 b4 = opt.kind == 0 ? *(const float4*)(s1 + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    float w[4] = {w4.x, w4.y, w4.z, w4.w}, a[4] = {a4.x, a4.y, a4.z, a4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w};
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (unsigned e = 0; e < n; ++e) {
      const G* gp = mail + ((size_t)pv->plan_src[e] * XPS_SLOTS + (pv->plan_seq[e] & 1u)) * piece + i * 4;
      float g[4];
      if (sizeof(G) == 4) {
        const float4 q = *(const float4*)gp;
        g[0] = q.x; g[1] = q.y; g[2] = q.z; g[3] = q.w;
      } else {
        const bf16x4 q = *(const bf16x4*)gp;
        g[0] = (float)q[0]; g[1] = (float)q[1]; g[2] = (float)q[2]; g[3] = (float)q[3];
      }
      if (sum_round) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += g[k];
        if (e + 1 < n) continue;
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = acc[k];
      }
      const unsigned step = sum_round ? 0u : e;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (opt.kind == 0) adam_update(w[k], a[k], b[k], g[k], gs, lr_t[step], opt.b1, opt.b2, opt.eps);
        else momentum_update(w[k], a[k], g[k], gs, lr, opt.mu, opt.wd, opt.nesterov);
      }
    }
    *(float4*)(wdst + i * 4) = make_float4(w[0], w[1], w[2], w[3]);
    *(float4*)(s0 + i * 4) = make_float4(a[0], a[1], a[2], a[3]);
    if (opt.kind == 0) *(float4*)(s1 + i * 4) = make_float4(b[0], b[1], b[2], b[3]);
  }
}

// Publish: new version, consumed pushes acknowledged, optimizer step count; optionally the worker's clock.
__global__ __launch_bounds__(64) void xps_commit_kernel(char* base, XpsPriv* pv, int advance_clock) {
  if (threadIdx.x != 0) return;
  XpsCtrl* me = (XpsCtrl*)base;
  const unsigned n = pv->plan_n;
  __threadfence_system();
  if (n) {
    st_sys(&me->applied, me->applied + pv->plan_steps);
    st_sys(&me->version, me->version + 1u);
    for (unsigned e = 0; e < n; ++e) {
      const unsigned s = pv->plan_src[e];
      if ((int)(pv->plan_seq[e] - me->seq_done[s]) > 0) st_sys(&me->seq_done[s], pv->plan_seq[e]);
    }
    pv->plan_n = 0;
  }
  if (advance_clock) st_sys(&me->clock, me->clock + 1u);
}

__global__ void xps_set_word_kernel(unsigned* p, unsigned v) {
  if (threadIdx.x == 0) st_sys(p, v);
}

// =================================================================================================== host
struct XpsCtx {
  int world, rank, nslot, gbytes;
  size_t piece, off_params, off_slots, off_mail, total;
  char* local;
  XpsPtrs ptrs;
  XpsPriv* priv;
  double timeout_s;
};

static long long ticks_of(double s) { return (long long)(s * 1e8); }  // wall_clock64 runs at 100 MHz

extern "C" int dtfa_xps_create(int world, int rank, long long piece, int nslot, int grad_is_bf16, double timeout_s,
                               void** out) {
  if (world < 1 || world > XPS_MAX_RANKS || rank < 0 || rank >= world || piece <= 0 || piece % 8 || nslot < 1 || nslot > 2)
    return -1;
  XpsCtx* c = new XpsCtx();
  c->world = world;
  c->rank = rank;
  c->nslot = nslot;
  c->gbytes = grad_is_bf16 ? 2 : 4;
  c->piece = (size_t)piece;
  c->off_params = XPS_CTRL_BYTES;
  c->off_slots = c->off_params + 2 * c->piece * 4;
  c->off_mail = c->off_slots + (size_t)nslot * c->piece * 4;
  c->total = c->off_mail + (size_t)world * XPS_SLOTS * c->piece * c->gbytes;
  c->timeout_s = timeout_s > 0 ? timeout_s : 5.0;
  for (int i = 0; i < XPS_MAX_RANKS; ++i) c->ptrs.base[i] = nullptr;
  hipError_t e = hipExtMallocWithFlags((void**)&c->local, c->total, hipDeviceMallocFinegrained);
  if (e != hipSuccess) {
    delete c;
    return (int)e;
  }
  e = hipMalloc((void**)&c->priv, sizeof(XpsPriv));
  if (e != hipSuccess) {
    (void)hipFree(c->local);
    delete c;
    return (int)e;
  }
  (void)hipMemset(c->local, 0, c->total);
  (void)hipMemset(c->priv, 0, sizeof(XpsPriv));
  (void)hipDeviceSynchronize();
  c->ptrs.base[rank] = c->local;
  *out = c;
  return 0;
}

extern "C" int dtfa_xps_handle(void* ctx, char* handle64) {
  XpsCtx* c = (XpsCtx*)ctx;
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, c->local);
  if (e != hipSuccess) return (int)e;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  memcpy(handle64, &h, 64);
  return 0;
}

extern "C" int dtfa_xps_connect(void* ctx, const char* handles) {
  XpsCtx* c = (XpsCtx*)ctx;
  for (int r = 0; r < c->world; ++r) {
    if (r == c->rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, handles + (size_t)r * 64, 64);
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return (int)e;
    c->ptrs.base[r] = (char*)p;
  }
  return 0;
}

static int connected(const XpsCtx* c) {
  for (int r = 0; r < c->world; ++r)
    if (c->ptrs.base[r] == nullptr) return 0;
  return 1;
}

// Owner: (re)initialise the own shard -- master values, optimizer slots (null = zeros), optimizer step count.
// Version and sequence words keep growing; the new values become version + 1.
extern "C" int dtfa_xps_set_shard(void* ctx, const float* master, const float* slot0, const float* slot1,
                                  unsigned applied, hipStream_t st) {
  XpsCtx* c = (XpsCtx*)ctx;
  XpsCtrl host;
  (void)hipStreamSynchronize(st);
  (void)hipMemcpy(&host, c->local, sizeof(XpsCtrl), hipMemcpyDeviceToHost);
  const unsigned nv = host.version + 1u;
  hipLaunchKernelGGL(xps_set_word_kernel, dim3(1), dim3(64), 0, st, &((XpsCtrl*)c->local)->wseq, nv);
  float* dstw = (float*)(c->local + c->off_params) + (size_t)(nv & 1u) * c->piece;
  hipError_t e = hipMemcpyAsync(dstw, master, c->piece * 4, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) return (int)e;
  const float* slots[2] = {slot0, slot1};
  for (int k = 0; k < c->nslot; ++k) {
    float* ds = (float*)(c->local + c->off_slots) + (size_t)k * c->piece;
    e = slots[k] ? hipMemcpyAsync(ds, slots[k], c->piece * 4, hipMemcpyDeviceToDevice, st)
                 : hipMemsetAsync(ds, 0, c->piece * 4, st);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(xps_set_word_kernel, dim3(1), dim3(64), 0, st, &((XpsCtrl*)c->local)->applied, applied);
  hipLaunchKernelGGL(xps_set_word_kernel, dim3(1), dim3(64), 0, st, &((XpsCtrl*)c->local)->version, nv);
  return (int)hipGetLastError();
}

// staleness < 0: no bound, nothing to wait for.
extern "C" int dtfa_xps_gate(void* ctx, int staleness, hipStream_t st) {
  XpsCtx* c = (XpsCtx*)ctx;
  if (!connected(c)) return -2;
  if (c->world == 1 || staleness < 0) return 0;
  hipLaunchKernelGGL(xps_gate_kernel, dim3(1), dim3(64), 0, st, c->ptrs, c->priv, c->world, c->rank, staleness,
                     ticks_of(c->timeout_s));
  return (int)hipGetLastError();
}

static unsigned blocks_per_owner(size_t piece, int world) {
  size_t b = (piece / 4 + 255) / 256;
  size_t cap = (size_t)(1024 / world);
  if (b > cap) b = cap;
  return (unsigned)(b ? b : 1);
}

// dst: world * piece values of the compute dtype (may be null); dst32: the same in fp32 (may be null).
extern "C" int dtfa_xps_pull(void* ctx, void* dst, int dst_is_bf16, float* dst32, hipStream_t st) {
  XpsCtx* c = (XpsCtx*)ctx;
  if (!connected(c)) return -2;
  if (dst == nullptr && dst32 == nullptr) return -1;
  const unsigned bpo = blocks_per_owner(c->piece, c->world);
  if (dst_is_bf16)
    hipLaunchKernelGGL(xps_pull_kernel<true>, dim3(bpo * c->world), dim3(256), 0, st, c->ptrs, c->priv, c->world, c->piece,
                       c->off_params, dst, dst32, bpo);
  else
    hipLaunchKernelGGL(xps_pull_kernel<false>, dim3(bpo * c->world), dim3(256), 0, st, c->ptrs, c->priv, c->world, c->piece,
                       c->off_params, dst, dst32, bpo);
  return (int)hipGetLastError();
}

// grad: world * piece gradient values in the context's gradient dtype.
extern "C" int dtfa_xps_push(void* ctx, const void* grad, hipStream_t st) {
  XpsCtx* c = (XpsCtx*)ctx;
  if (!connected(c)) return -2;
  if (((uintptr_t)grad) % 16) return -1;
  hipLaunchKernelGGL(xps_push_wait_kernel, dim3(1), dim3(64), 0, st, c->ptrs, c->priv, c->world, c->rank,
                     ticks_of(c->timeout_s));
  const unsigned bpo = blocks_per_owner(c->piece, c->world);
  if (c->gbytes == 2)
    hipLaunchKernelGGL(xps_push_kernel<bf16>, dim3(bpo * c->world), dim3(256), 0, st, c->ptrs, c->priv, c->world, c->rank,
                       c->piece, c->off_mail, (const bf16*)grad, bpo);
  else
    hipLaunchKernelGGL(xps_push_kernel<float>, dim3(bpo * c->world), dim3(256), 0, st, c->ptrs, c->priv, c->world, c->rank,
                       c->piece, c->off_mail, (const float*)grad, bpo);
  return (int)hipGetLastError();
}

// Owner side: consume arrived gradients, publish, optionally count one worker step.
// hyper: device float[2] = (learning rate or Adam step size, gradient scale).
extern "C" int dtfa_xps_apply(void* ctx, int kind, float b1, float b2, float eps, float mu, float wd, int nesterov,
                              int host_step_size, int aggregate, int advance_clock, const float* hyper, hipStream_t st) {
  XpsCtx* c = (XpsCtx*)ctx;
  if (!connected(c)) return -2;
  if ((kind == 0 && c->nslot != 2) || (kind == 1 && c->nslot != 1) || kind < 0 || kind > 1) return -1;
  XpsOpt o;
  o.kind = kind; o.b1 = b1; o.b2 = b2; o.eps = eps; o.mu = mu; o.wd = wd; o.nesterov = nesterov;
  o.host_step_size = host_step_size;
  hipLaunchKernelGGL(xps_collect_kernel, dim3(1), dim3(64), 0, st, c->ptrs, c->priv, c->world, c->rank, aggregate,
                     ticks_of(c->timeout_s));
  size_t b = (c->piece / 4 + 255) / 256;
  if (b > 1024) b = 1024;
  if (c->gbytes == 2)
    hipLaunchKernelGGL(xps_apply_kernel<bf16>, dim3((unsigned)b), dim3(256), 0, st, c->local, c->priv, c->piece,
                       c->off_params, c->off_slots, c->off_mail, o, hyper);
  else
    hipLaunchKernelGGL(xps_apply_kernel<float>, dim3((unsigned)b), dim3(256), 0, st, c->local, c->priv, c->piece,
                       c->off_params, c->off_slots, c->off_mail, o, hyper);
  hipLaunchKernelGGL(xps_commit_kernel, dim3(1), dim3(64), 0, st, c->local, c->priv, advance_clock);
  return (int)hipGetLastError();
}

extern "C" int dtfa_xps_leave(void* ctx, hipStream_t st) {
  XpsCtx* c = (XpsCtx*)ctx;
  hipLaunchKernelGGL(xps_set_word_kernel, dim3(1), dim3(64), 0, st, &((XpsCtrl*)c->local)->done, 1u);
  return (int)hipGetLastError();
}

// Copy optimizer slot k of owner `peer` (its whole shard) into dst (device, piece floats).
extern "C" int dtfa_xps_read_slot(void* ctx, int peer, int k, float* dst, hipStream_t st) {
  XpsCtx* c = (XpsCtx*)ctx;
  if (peer < 0 || peer >= c->world || k < 0 || k >= c->nslot || c->ptrs.base[peer] == nullptr) return -1;
  const float* src = (const float*)(c->ptrs.base[peer] + c->off_slots) + (size_t)k * c->piece;
  return (int)hipMemcpyAsync(dst, src, c->piece * 4, hipMemcpyDeviceToDevice, st);
}

// out[0..15] (host): own clock, own version, own applied, applied of owner 0, status, torn pulls, dropped pushes,
// timeouts, dead mask, pushes made, done mask of the peers, clock of every rank (8 words).  Synchronises the stream.
extern "C" int dtfa_xps_state(void* ctx, unsigned* out, hipStream_t st) {
  XpsCtx* c = (XpsCtx*)ctx;
  (void)hipStreamSynchronize(st);
  XpsCtrl mine, other;
  XpsPriv pv;
  hipError_t e = hipMemcpy(&mine, c->local, sizeof(XpsCtrl), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return (int)e;
  e = hipMemcpy(&pv, c->priv, sizeof(XpsPriv), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return (int)e;
  memset(out, 0, 24 * sizeof(unsigned));
  out[0] = mine.clock;
  out[1] = mine.version;
  out[2] = mine.applied;
  out[4] = pv.status;
  out[5] = pv.torn_pulls;
  out[6] = pv.dropped_pushes;
  out[7] = pv.timeouts;
  out[9] = pv.push_seq;
  for (int r = 0; r < c->world; ++r) {
    if (pv.dead[r]) out[8] |= 1u << r;
    if (c->ptrs.base[r] == nullptr) continue;
    if (r == c->rank) {
      other = mine;
    } else {
      e = hipMemcpy(&other, c->ptrs.base[r], sizeof(XpsCtrl), hipMemcpyDeviceToHost);
      if (e != hipSuccess) return (int)e;
    }
    if (r == 0) out[3] = other.applied;
    if (other.done) out[10] |= 1u << r;
    out[16 + r] = other.clock;
  }
  return 0;
}

extern "C" void dtfa_xps_destroy(void* ctx) {
  XpsCtx* c = (XpsCtx*)ctx;
  if (c == nullptr) return;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < c->world; ++r)
    if (r != c->rank && c->ptrs.base[r] != nullptr) (void)hipIpcCloseMemHandle(c->ptrs.base[r]);
  (void)hipFree(c->local);
  (void)hipFree(c->priv);
  delete c;
}
