// GPU-resident asynchronous parameter server over one-sided peer memory (xGMI), gfx950.
//
// The reference trains with TensorFlow's asynchronous parameter server: variables and ApplyAdam live on
// ps tasks, every worker pulls the variables, computes a gradient and pushes it, nobody waits for anybody, and a
// worker that dies never touches the variables (recognition_flower/distributed/train_dataset.py:141-149,162-165,
// 172-174).  Here the variables live in GPU memory that every process of the node has mapped through hipIpc, and
// neither a collective nor the host is on the data path.
//
// Roles.  A job has `nown` OWNERS (shards) and `nwrk` WORKERS (gradient sources).  Owner o holds shard o (1/nown) of a
// flat parameter group -- fp32 master, optimizer slots, one pair of mailbox slots per worker -- in fine-grained device
// memory, its REGION.  Two deployments use the same kernels:
//   * ps tasks are the owners (a ps process runs the apply loop and nothing else), workers are pure clients that map
//     the owners' regions and may join at any time or die at any time;
//   * no ps tasks: every worker process also owns a shard.  When such a process leaves or dies its region stays
//     mapped in the other processes, and the lowest-ranked live process TAKES THE REGION OVER in place: it claims the
//     region's `adopter` word with a compare-and-swap and from then on runs collect / apply / commit for it with its
//     own kernels (peer loads and stores).  Pushers and pullers notice nothing; no shard stops training.
//
//   push   a worker copies shard o of its gradient straight into its mailbox slot in region o and then raises a
//          sequence word there;
//   apply  whoever applies region o consumes what has arrived -- fused Adam / momentum, one optimizer step per
//          arrived gradient (asynchronous, as ApplyAdam on a ps) or one step on the worker-ordered sum of a round
//          (lock step) -- and publishes a new parameter version;
//   pull   a worker reads the current version of every shard directly from its region;
//   gate   a worker whose clock is more than `staleness` steps ahead of the slowest live worker waits, reading the
//          workers' clock words (kept in region 0); a worker that does not move for `timeout` stops counting -- unless
//          it sits in a bounded wait of its own, which its beat word shows (see `beat`).
//
// Region: [0, 512) control words (XpsCtrl) | 2 x piece fp32 parameter versions (buffer v & 1) | nslot x piece fp32
// optimizer slots | nwrk x 2 x piece gradients (fp32 or bf16): [worker][push parity].
// Sequence words only ever grow.  A parameter buffer is read under a sequence lock: the applier raises `wseq`
// before it starts writing version n into buffer n & 1 and `version` after; a reader that copied buffer
// v & 1 and then finds wseq >= v + 2 read a buffer that was being overwritten and retries.
//
// Every wait in here is bounded by a wall-clock timeout and runs in a single 64-lane workgroup.
#include "common.h"
#include "opt_update.h"
#include <string.h>

#define XPS_MAX_O 16
#define XPS_MAX_W 16
#define XPS_SLOTS 2
#define XPS_CTRL_BYTES 512
#define XPS_MAX_PLAN (XPS_MAX_W * XPS_SLOTS)

struct XpsCtrl {
  unsigned version;   // parameter versions published for this shard
  unsigned wseq;      // version whose write has begun
  unsigned applied;   // optimizer steps applied to this shard
  unsigned adopter;   // 0: the region's own process applies it; p + 1: process p has taken it over
  unsigned odone;     // the region's own process has left the job
  unsigned shutdown;  // the chief asks owner-only processes to stop serving
  unsigned pad[2];
  unsigned wclock[XPS_MAX_W];    // region 0 only: steps worker w has completed
  unsigned wdone[XPS_MAX_W];     // region 0 only: worker w has left
  unsigned wjoined[XPS_MAX_W];   // region 0 only: worker w has joined at some time
  unsigned seq_in[XPS_MAX_W];    // [worker]: gradient pushes completed by that worker
  unsigned seq_done[XPS_MAX_W];  // [worker]: pushes of that worker consumed
  unsigned wbeat[XPS_MAX_W];     // region 0 only: keeps changing while worker w sits in one of its own bounded waits
};
static_assert(sizeof(XpsCtrl) <= XPS_CTRL_BYTES, "control block");

struct XpsPlan {
  unsigned n;       // mailbox slots to consume
  unsigned steps;   // optimizer steps they amount to
  float scale;      // gradient scale of an aggregated round
  unsigned src[XPS_MAX_PLAN];
  unsigned seq[XPS_MAX_PLAN];
};

// private (not shared) device words of a process
struct XpsPriv {
  unsigned dead[XPS_MAX_W];        // workers that stopped moving
  unsigned dead_clock[XPS_MAX_W];  // their clock when they were written off (a change revives them)
  unsigned odead[XPS_MAX_O];       // regions whose mailbox slot stayed full for a whole timeout
  unsigned skip[XPS_MAX_O];        // push: regions whose mailbox slot is not free (gradient dropped)
  XpsPlan plan[XPS_MAX_O];         // apply: what to consume, per region
  unsigned push_seq;               // pushes this worker has made (dropped ones included)
  unsigned oseq[XPS_MAX_O];        // pushes of this worker that ARRIVED in region o: a region's numbering has no gaps, so
                                   // "push k reuses the slot of push k - 2" holds whatever was dropped on the way
  unsigned block_counter;          // last-block detection of the push kernel
  unsigned status;                 // bit 0: a peer timed out, bit 1: push dropped, bit 2: torn pull kept, bit 3: took a region over
  unsigned torn_pulls, dropped_pushes, timeouts, adopted_mask;
};

struct XpsPtrs {
  char* base[XPS_MAX_O];  // region of every owner (own included)
};

// Words the host reads without synchronising (page-locked host memory, written by kernels with system-scope stores).
struct XpsReport {
  unsigned applied0;   // optimizer steps applied to shard 0 as of the last pull
  unsigned status;     // XpsPriv::status as of the last push
  unsigned own_applied;
  unsigned pad;
};

__device__ __forceinline__ unsigned ld_sys(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys(unsigned* p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ XpsCtrl* ctrl_of(const XpsPtrs& P, int o) { return (XpsCtrl*)P.base[o]; }

// Telling "waiting" from "gone".  A process that sits in one of its bounded waits (gate, full mailbox slot, a round's
// missing push) keeps changing its beat word; a process waiting for THAT process restarts its timer whenever the word
// has changed, up to XPS_WAIT_CAP timeouts in all.  Without this, two survivors of a dead third write each other off
// in turn: each one's wait for the dead process lasts exactly as long as the other's patience with it.
#define XPS_WAIT_CAP 4
__device__ __forceinline__ void beat(XpsCtrl* c0, int widx, unsigned it) {
  if (widx >= 0 && (it & 15u) == 0u)
    __hip_atomic_store(&c0->wbeat[widx], (unsigned)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
struct PeerWait {
  long long t_start, t0;
  unsigned last;
};
__device__ __forceinline__ PeerWait begin_wait(const XpsCtrl* c0, int peer) {
  PeerWait w;
  w.t_start = w.t0 = wall_clock64();
  w.last = peer >= 0 ? ld_sys(&c0->wbeat[peer]) : 0u;
  return w;
}
// ``peer`` < 0: nobody's beat to watch (an owner-only process never waits for anybody), the plain timeout.
__device__ __forceinline__ bool peer_timed_out(PeerWait& w, const XpsCtrl* c0, int peer, long long timeout_ticks) {
  const long long now = wall_clock64();
  if (peer >= 0) {
    const unsigned b = ld_sys(&c0->wbeat[peer]);
    if (b != w.last) {
      w.last = b;
      w.t0 = now;
    }
  }
  return now - w.t0 > timeout_ticks || now - w.t_start > XPS_WAIT_CAP * timeout_ticks;
}
// Written off, and its clock has not moved since.
__device__ __forceinline__ bool known_dead(const XpsPriv* pv, const XpsCtrl* c0, int j) {
  return pv->dead[j] && ld_sys(&c0->wclock[j]) == pv->dead_clock[j];
}

// A worker counts for a bound while it has joined, is not done and is not written off.  A written-off worker whose
// clock has moved since is alive again.
__device__ __forceinline__ bool worker_counts(XpsPriv* pv, const XpsCtrl* c0, int j, unsigned cj) {
  if (!ld_sys(&c0->wjoined[j]) || ld_sys(&c0->wdone[j])) return false;
  if (pv->dead[j]) {
    if (cj == pv->dead_clock[j]) return false;
    pv->dead[j] = 0;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------- join
// A worker enters the job (at start-up or late): its clock starts at the slowest live worker's clock (so that it is
// nobody's straggler), its push numbering continues where the regions' sequence words for it stand.
__global__ __launch_bounds__(64) void xps_join_kernel(XpsPtrs P, XpsPriv* pv, int nown, int nwrk, int widx) {
  XpsCtrl* c0 = ctrl_of(P, 0);
  const int j = threadIdx.x;
  unsigned cj = 0xffffffffu, sq = 0;
  if (j < nwrk && j != widx && ld_sys(&c0->wjoined[j]) && !ld_sys(&c0->wdone[j])) cj = ld_sys(&c0->wclock[j]);
  if (j < nown) {
    sq = ld_sys(&ctrl_of(P, j)->seq_in[widx]);
    pv->oseq[j] = sq;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    cj = min(cj, (unsigned)__shfl_xor((int)cj, off, 64));
    sq = max(sq, (unsigned)__shfl_xor((int)sq, off, 64));
  }
  if (j == 0) {
    const unsigned mine = ld_sys(&c0->wclock[widx]);
    const unsigned start = cj == 0xffffffffu ? mine : (mine > cj ? mine : cj);
    pv->push_seq = sq;
    st_sys(&c0->wclock[widx], start);
    st_sys(&c0->wdone[widx], 0u);
    st_sys(&c0->wjoined[widx], 1u);
  }
}

// ---------------------------------------------------------------------------------------------------- gate
// Wait while my clock is more than `staleness` ahead of a live worker.  In the deployment without ps tasks an owner
// advances its clock right after it has published the parameter version of that step, so with staleness 0
// (lock-step rounds) passing the gate also means that every live owner has applied the previous round.
__global__ __launch_bounds__(64) void xps_gate_kernel(XpsPtrs P, XpsPriv* pv, int nwrk, int widx, int staleness,
                                                      long long timeout_ticks) {
  const int j = threadIdx.x;
  if (j >= nwrk || j == widx) return;
  XpsCtrl* c0 = ctrl_of(P, 0);
  const unsigned my = ld_sys(&c0->wclock[widx]);
  PeerWait w = begin_wait(c0, j);
  for (unsigned it = 0;; ++it) {
    const unsigned cj = ld_sys(&c0->wclock[j]);
    if (!worker_counts(pv, c0, j, cj)) return;
    if ((int)(my - cj) <= staleness) return;
    __builtin_amdgcn_s_sleep(32);
    beat(c0, widx, it);
    if (peer_timed_out(w, c0, j, timeout_ticks)) {
      pv->dead[j] = 1;
      pv->dead_clock[j] = cj;
      atomicOr(&pv->status, 1u);
      atomicAdd(&pv->timeouts, 1u);
      return;
    }
  }
}

// ---------------------------------------------------------------------------------------------------- pull
// dst (compute dtype) and / or dst32 (fp32) receive the group's values [lo, hi) (element indices in the group, whole
// 16-byte vectors): every shard's part of that range from its region's current version.
template <bool BF16>
__global__ __launch_bounds__(256) void xps_pull_kernel(XpsPtrs P, XpsPriv* pv, int nown, size_t piece, size_t off_params,
                                                       void* __restrict__ dst, float* __restrict__ dst32,
                                                       unsigned blocks_per_owner, XpsReport* report, size_t lo, size_t hi) {
  const int o = blockIdx.x / blocks_per_owner;
  const unsigned b = blockIdx.x % blocks_per_owner;
  if (o >= nown) return;
  const XpsCtrl* c = ctrl_of(P, o);
  if (blockIdx.x == 0 && threadIdx.x == 0) st_sys(&report->applied0, ld_sys(&c->applied));
  const size_t o_lo = (size_t)o * piece, o_hi = o_lo + piece;
  const size_t a = lo > o_lo ? lo : o_lo, z = hi < o_hi ? hi : o_hi;
  if (a >= z) return;
  const size_t v0 = (a - o_lo) / 4, v1 = (z - o_lo) / 4;   // this shard's vectors [v0, v1)
  __shared__ unsigned sv;
  for (int attempt = 0; attempt < 4; ++attempt) {
    if (threadIdx.x == 0) sv = ld_sys(&c->version);
    __syncthreads();
    const unsigned v = sv;
    const float* src = (const float*)(P.base[o] + off_params) + (size_t)(v & 1u) * piece;
    for (size_t i = v0 + (size_t)b * blockDim.x + threadIdx.x; i < v1; i += (size_t)blocks_per_owner * blockDim.x) {
      const f32x4 w = __builtin_nontemporal_load((const f32x4*)(src + i * 4));
      const size_t e = o_lo + i * 4;
      if (dst32) *(f32x4*)(dst32 + e) = w;
      if (dst) {
        if (BF16) {
          bf16x4 q = {(bf16)w[0], (bf16)w[1], (bf16)w[2], (bf16)w[3]};
          *(bf16x4*)((bf16*)dst + e) = q;
        } else {
          *(f32x4*)((float*)dst + e) = w;
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
// Before a push: lane o waits until this worker's push k - 2 to region o (the slot push k reuses; k counts the pushes
// that arrived in THAT region) has been consumed there.
// A region whose slot stays full for a whole timeout is written off: this push and the following ones skip it at
// once (dropped, counted) until its slot is free again -- which it becomes as soon as somebody applies the region,
// e.g. the process that takes it over.
__global__ __launch_bounds__(64) void xps_push_wait_kernel(XpsPtrs P, XpsPriv* pv, int nown, int widx,
                                                           int owners_are_workers, long long timeout_ticks) {
  const int o = threadIdx.x;
  if (o >= nown) return;
  const unsigned k = pv->oseq[o] + 1u;
  pv->skip[o] = 0;
  if (k <= XPS_SLOTS) return;
  XpsCtrl* c0 = ctrl_of(P, 0);
  const XpsCtrl* c = ctrl_of(P, o);
  PeerWait w = begin_wait(c0, -1);
  int watched = -1;
  for (unsigned it = 0; (int)(ld_sys(&c->seq_done[widx]) - (k - XPS_SLOTS)) < 0; ++it) {
    // who is to free the slot: the region's own process or whoever has taken the region over (where the owners are
    // the workers; an owner-only process has no beat and is simply given the timeout)
    const unsigned a = ld_sys(&c->adopter);
    int applier = owners_are_workers ? (a ? (int)a - 1 : o) : -1;
    if (applier == widx) applier = -1;
    if (applier != watched) {
      watched = applier;
      w.last = applier >= 0 ? ld_sys(&c0->wbeat[applier]) : 0u;
    }
    // (a process the gate has already written off is not waited for a second time)
    bool give_up = pv->odead[o] != 0 || (applier >= 0 && known_dead(pv, c0, applier));
    if (!give_up) {
      __builtin_amdgcn_s_sleep(32);
      beat(c0, widx, it);
      if (peer_timed_out(w, c0, applier, timeout_ticks)) {
        pv->odead[o] = 1;
        if (applier >= 0) {   // ... nor does the gate wait for a process that let a slot time out
          pv->dead_clock[applier] = ld_sys(&c0->wclock[applier]);
          pv->dead[applier] = 1;
        }
        atomicAdd(&pv->timeouts, 1u);
        atomicOr(&pv->status, 1u);
        give_up = true;
      }
    }
    if (give_up) {
      pv->skip[o] = 1;
      atomicOr(&pv->status, 2u);
      atomicAdd(&pv->dropped_pushes, 1u);
      return;
    }
  }
  pv->odead[o] = 0;
}

template <typename G>
__global__ __launch_bounds__(256) void xps_push_kernel(XpsPtrs P, XpsPriv* pv, int nown, int widx, size_t piece,
                                                       size_t off_mail, const G* __restrict__ grad,
                                                       unsigned blocks_per_owner, XpsReport* report) {
  const int o = blockIdx.x / blocks_per_owner;
  const unsigned b = blockIdx.x % blocks_per_owner;
  if (o < nown && !pv->skip[o]) {
    const unsigned k = pv->oseq[o] + 1u;
    constexpr int VE = 16 / sizeof(G);  // elements per 16-byte vector
    const size_t nvec = piece / VE;
    G* dst = (G*)(P.base[o] + off_mail) + ((size_t)widx * XPS_SLOTS + (k & 1u)) * piece;
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
  if (threadIdx.x < (unsigned)nown && !pv->skip[threadIdx.x]) {
    const unsigned k = pv->oseq[threadIdx.x] + 1u;
    st_sys(&ctrl_of(P, threadIdx.x)->seq_in[widx], k);
    pv->oseq[threadIdx.x] = k;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    pv->push_seq = pv->push_seq + 1u;
    pv->block_counter = 0;
    st_sys(&report->status, pv->status);
  }
}

// ---------------------------------------------------------------------------------------------------- apply
// Whether the process that ran as owner / worker p is (as far as this process can tell) still in the job.
__device__ __forceinline__ bool proc_alive(const XpsPtrs& P, const XpsPriv* pv, const XpsCtrl* c0, int p) {
  return !ld_sys(&c0->wdone[p]) && !pv->dead[p] && !pv->odead[p] && !ld_sys(&ctrl_of(P, p)->odone);
}

// One workgroup per region: decide what is applied to region o = blockIdx.x now.
//   own region (o == self): everything below, unless another process has taken the region over (then nothing,
//     for good: this process was written off by its peers and is only a worker from now on).
//   other regions (o != self; only in the deployment where every owner is also worker o): nothing while the
//     region's own process is in the job.  Once that process has left or was written off, the lowest-ranked live
//     process claims the region (compare-and-swap on `adopter`, also over a previous adopter that has gone) and
//     plans for it like for its own.
//   aggregate = 1: the next unconsumed push of every live worker, waited for; one optimizer step on their
//                  worker-ordered sum scaled by 1 / (workers included).
//   aggregate = 0: every push that has arrived and is not consumed yet, one optimizer step each, workers in
//                  index order, oldest first.  No waiting -- except with wait_any (the serving loop of an
//                  owner-only process), which sleeps up to `idle_ticks` until something has arrived.
__global__ __launch_bounds__(64) void xps_collect_kernel(XpsPtrs P, XpsPriv* pv, int nown, int nwrk, int self,
                                                         int widx, int adopt_others, int aggregate, int wait_any,
                                                         long long timeout_ticks, long long idle_ticks) {
  const int o = blockIdx.x;
  XpsCtrl* R = ctrl_of(P, o);
  const XpsCtrl* c0 = ctrl_of(P, 0);
  XpsPlan* plan = &pv->plan[o];
  const int j = threadIdx.x;
  __shared__ unsigned have[XPS_MAX_W];
  __shared__ int go;
  if (j == 0) {
    const unsigned a = ld_sys(&R->adopter);
    if (o == self) {
      go = a == 0u || a == (unsigned)self + 1u;
    } else if (!adopt_others || self < 0) {
      go = 0;
    } else {
      const bool gone = ld_sys(&R->odone) || pv->odead[o] || pv->dead[o] || ld_sys(&c0->wdone[o]);
      if (!gone) {
        go = 0;
      } else if (a == (unsigned)self + 1u) {
        go = 1;
      } else if (a != 0u && (int)a - 1 != o && proc_alive(P, pv, c0, (int)a - 1)) {
        go = 0;   // somebody who is still here has it
      } else {
        int lowest = -1;
        for (int p = 0; p < nown && lowest < 0; ++p)
          if (p != o && (p == self || proc_alive(P, pv, c0, p))) lowest = p;
        go = 0;
        if (lowest == self) {
          unsigned expect = a;
          go = __hip_atomic_compare_exchange_strong(&R->adopter, &expect, (unsigned)self + 1u, __ATOMIC_ACQ_REL,
                                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (go) {
            atomicOr(&pv->adopted_mask, 1u << o);
            atomicOr(&pv->status, 8u);
          }
        }
      }
    }
  }
  __syncthreads();
  if (!go) {
    if (j == 0) plan->n = 0;
    return;
  }
  if (wait_any) {
    // single wave: sleep until a push has arrived, the chief asks to stop, or the idle period is over
    const long long t0 = wall_clock64();
    for (;;) {
      const bool mine = j < nwrk && ld_sys(&R->seq_in[j]) != R->seq_done[j];
      if (__any(mine) || ld_sys(&R->shutdown) || wall_clock64() - t0 > idle_ticks) break;
      __builtin_amdgcn_s_sleep(64);
    }
  }
  if (j < nwrk) {
    unsigned in = ld_sys(&R->seq_in[j]);
    if (aggregate) {
      const unsigned need = R->seq_done[j] + 1u;
      PeerWait w = begin_wait(c0, j == widx ? -1 : j);
      bool counts = true;
      for (unsigned it = 0; (int)(in - need) < 0; ++it) {
        if (!worker_counts(pv, c0, j, ld_sys(&c0->wclock[j]))) { counts = false; break; }
        __builtin_amdgcn_s_sleep(32);
        beat(ctrl_of(P, 0), widx, it);
        if (peer_timed_out(w, c0, j == widx ? -1 : j, timeout_ticks)) {
          pv->dead[j] = 1;
          pv->dead_clock[j] = ld_sys(&c0->wclock[j]);
          atomicOr(&pv->status, 1u);
          atomicAdd(&pv->timeouts, 1u);
          counts = false;
          break;
        }
        in = ld_sys(&R->seq_in[j]);
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
    for (int s = 0; s < nwrk; ++s)
      if (have[s]) {
        plan->src[n] = s;
        plan->seq[n] = have[s];
        ++n;
      }
    plan->steps = n ? 1u : 0u;
    plan->scale = n ? 1.0f / (float)n : 0.f;
  } else {
    for (int s = 0; s < nwrk; ++s) {
      const unsigned done = R->seq_done[s];
      // (at most the XPS_SLOTS newest pushes of a worker are still in its mailbox)
      unsigned q = done + 1u;
      if ((int)(have[s] - q) >= XPS_SLOTS) q = have[s] - (XPS_SLOTS - 1u);
      for (; (int)(have[s] - q) >= 0 && n < XPS_MAX_PLAN; ++q) {
        plan->src[n] = s;
        plan->seq[n] = q;
        ++n;
      }
    }
    plan->steps = n;
    plan->scale = 1.0f;
  }
  plan->n = n;
  if (n) {
    __atomic_thread_fence(__ATOMIC_ACQUIRE);       // the gradients behind the sequence words
    st_sys(&R->wseq, R->version + 1u);             // readers of the buffer about to be overwritten retry
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
// Grid: nown x blocks_per_region; the workgroups of a region without a plan return at once.
template <typename G>
__global__ __launch_bounds__(256) void xps_apply_kernel(XpsPtrs P, const XpsPriv* pv, size_t piece, size_t off_params,
                                                        size_t off_slots, size_t off_mail, XpsOpt opt,
                                                        const float* __restrict__ hyper, unsigned blocks_per_region) {
  const int o = blockIdx.x / blocks_per_region;
  const unsigned bid = blockIdx.x % blocks_per_region;
  const XpsPlan* plan = &pv->plan[o];
  const unsigned n = plan->n;
  if (n == 0) return;
  char* base = P.base[o];
  XpsCtrl* R = (XpsCtrl*)base;
  const unsigned cur = R->version & 1u;
  const unsigned t0 = R->applied;
  const float* wsrc = (const float*)(base + off_params) + (size_t)cur * piece;
  float* wdst = (float*)(base + off_params) + (size_t)(cur ^ 1u) * piece;
  float* s0 = (float*)(base + off_slots);
  float* s1 = s0 + piece;
  const G* mail = (const G*)(base + off_mail);
  const float lr = hyper[0], gs = hyper[1] * plan->scale;
  const bool sum_round = plan->steps == 1u && n > 1u;
  // per-step Adam step sizes (bias correction folded in, as ApplyAdam)
  float lr_t[XPS_MAX_PLAN];
  if (opt.kind == 0) {
    for (unsigned e = 0; e < plan->steps; ++e) {
      const float t = (float)(t0 + e + 1u);
      lr_t[e] = opt.host_step_size ? lr : lr * sqrtf(1.f - powf(opt.b2, t)) / (1.f - powf(opt.b1, t));
    }
  }
  const size_t nvec = piece / 4;
  for (size_t i = (size_t)bid * blockDim.x + threadIdx.x; i < nvec; i += (size_t)blocks_per_region * blockDim.x) {
    float4 w4 = *(const float4*)(wsrc + i * 4);
    float4 a4 = *(const float4*)(s0 + i * 4);
    float4 b4 = opt.kind == 0 ? *(const float4*)(s1 + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    float w[4] = {w4.x, w4.y, w4.z, w4.w}, a[4] = {a4.x, a4.y, a4.z, a4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w};
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (unsigned e = 0; e < n; ++e) {
      // (peer-written mailbox data: loaded past the caches' stale lines -- the acquire of the collect kernel
      // ordered this launch behind the stores, a non-temporal load also keeps the vector cache out of the way)
      const G* gp = mail + ((size_t)plan->src[e] * XPS_SLOTS + (plan->seq[e] & 1u)) * piece + i * 4;
      float g[4];
      if (sizeof(G) == 4) {
        const f32x4 q = __builtin_nontemporal_load((const f32x4*)gp);
        g[0] = q[0]; g[1] = q[1]; g[2] = q[2]; g[3] = q[3];
      } else {
        const bf16x4 q = __builtin_nontemporal_load((const bf16x4*)gp);
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

// Publish, one workgroup per region: new version, consumed pushes acknowledged, optimizer step count; block `self`
// optionally advances the worker's clock (kept in region 0).
__global__ __launch_bounds__(64) void xps_commit_kernel(XpsPtrs P, XpsPriv* pv, int self, int widx, int advance_clock,
                                                        XpsReport* report) {
  if (threadIdx.x != 0) return;
  const int o = blockIdx.x;
  XpsCtrl* R = ctrl_of(P, o);
  XpsPlan* plan = &pv->plan[o];
  const unsigned n = plan->n;
  __threadfence_system();
  if (n) {
    st_sys(&R->applied, R->applied + plan->steps);
    st_sys(&R->version, R->version + 1u);
    for (unsigned e = 0; e < n; ++e) {
      const unsigned s = plan->src[e];
      if ((int)(plan->seq[e] - R->seq_done[s]) > 0) st_sys(&R->seq_done[s], plan->seq[e]);
    }
    plan->n = 0;
  }
  if (o == self) st_sys(&report->own_applied, R->applied);
  if (advance_clock && widx >= 0 && o == (self >= 0 ? self : 0)) {
    XpsCtrl* c0 = ctrl_of(P, 0);
    st_sys(&c0->wclock[widx], c0->wclock[widx] + 1u);
  }
}

__global__ void xps_set_word_kernel(unsigned* p, unsigned v) {
  if (threadIdx.x == 0) st_sys(p, v);
}

// =================================================================================================== host
struct XpsCtx {
  int nown, nwrk, oidx, widx, nslot, gbytes;
  size_t piece, off_params, off_slots, off_mail, total;
  char* local;
  XpsPtrs ptrs;
  XpsPriv* priv;
  XpsReport* report;   // page-locked host memory, device-visible
  double timeout_s;
  double ticks_per_s;
};

static long long ticks_of(const XpsCtx* c, double s) { return (long long)(s * c->ticks_per_s); }

// oidx < 0: this process owns no shard (a pure worker); widx < 0: it pushes no gradients (an owner-only process).
extern "C" int dtfa_xps_create(int nown, int nwrk, int oidx, int widx, long long piece, int nslot, int grad_is_bf16,
                               double timeout_s, void** out) {
  if (nown < 1 || nown > XPS_MAX_O || nwrk < 1 || nwrk > XPS_MAX_W || oidx >= nown || widx >= nwrk ||
      (oidx < 0 && widx < 0) || piece <= 0 || piece % 8 || nslot < 1 || nslot > 2)
    return -1;
  XpsCtx* c = new XpsCtx();
  c->nown = nown;
  c->nwrk = nwrk;
  c->oidx = oidx;
  c->widx = widx;
  c->nslot = nslot;
  c->gbytes = grad_is_bf16 ? 2 : 4;
  c->piece = (size_t)piece;
  c->off_params = XPS_CTRL_BYTES;
  c->off_slots = c->off_params + 2 * c->piece * 4;
  c->off_mail = c->off_slots + (size_t)nslot * c->piece * 4;
  c->total = c->off_mail + (size_t)nwrk * XPS_SLOTS * c->piece * c->gbytes;
  c->timeout_s = timeout_s > 0 ? timeout_s : 5.0;
  c->local = nullptr;
  c->report = nullptr;
  for (int i = 0; i < XPS_MAX_O; ++i) c->ptrs.base[i] = nullptr;
  int dev = 0, khz = 0;
  (void)hipGetDevice(&dev);
  // wall_clock64() counts at the device's constant "wall clock" rate (100 MHz on gfx950)
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
  c->ticks_per_s = (double)khz * 1e3;
  hipError_t e = hipSuccess;
  if (oidx >= 0) {
    e = hipExtMallocWithFlags((void**)&c->local, c->total, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
      delete c;
      return (int)e;
    }
    (void)hipMemset(c->local, 0, c->total);
    c->ptrs.base[oidx] = c->local;
  }
  e = hipMalloc((void**)&c->priv, sizeof(XpsPriv));
  if (e == hipSuccess) e = hipHostMalloc((void**)&c->report, sizeof(XpsReport), hipHostMallocMapped);
  if (e != hipSuccess) {
    if (c->local) (void)hipFree(c->local);
    delete c;
    return (int)e;
  }
  memset(c->report, 0, sizeof(XpsReport));
  (void)hipMemset(c->priv, 0, sizeof(XpsPriv));
  (void)hipDeviceSynchronize();
  *out = c;
  return 0;
}

extern "C" int dtfa_xps_handle(void* ctx, char* handle64) {
  XpsCtx* c = (XpsCtx*)ctx;
  if (c->local == nullptr) return -1;
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, c->local);
  if (e != hipSuccess) return (int)e;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  memcpy(handle64, &h, 64);
  return 0;
}

// handles: nown x 64 bytes, the regions' IPC handles in owner order (the own entry is ignored).
extern "C" int dtfa_xps_connect(void* ctx, const char* handles) {
  XpsCtx* c = (XpsCtx*)ctx;
  for (int o = 0; o < c->nown; ++o) {
    if (o == c->oidx || c->ptrs.base[o] != nullptr) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, handles + (size_t)o * 64, 64);
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return (int)e;
    c->ptrs.base[o] = (char*)p;
  }
  return 0;
}

static int connected(const XpsCtx* c) {
  for (int o = 0; o < c->nown; ++o)
    if (c->ptrs.base[o] == nullptr) return 0;
  return 1;
}

// (Re)initialise shard o -- master values, optimizer slots (null = zeros), optimizer step count -- in its region,
// the own one or a peer's (the chief initialises the regions of owner-only processes).  Version and sequence words
// keep growing; the new values become version + 1.
extern "C" int dtfa_xps_set_shard(void* ctx, int o, const float* master, const float* slot0, const float* slot1,
                                  unsigned applied, hipStream_t st) {
  XpsCtx* c = (XpsCtx*)ctx;
  if (o < 0 || o >= c->nown || c->ptrs.base[o] == nullptr) return -1;
  char* base = c->ptrs.base[o];
  XpsCtrl host;
  (void)hipStreamSynchronize(st);
  hipError_t e = hipMemcpy(&host, base, sizeof(XpsCtrl), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return (int)e;
  const unsigned nv = host.version + 1u;
  hipLaunchKernelGGL(xps_set_word_kernel, dim3(1), dim3(64), 0, st, &((XpsCtrl*)base)->wseq, nv);
  float* dstw = (float*)(base + c->off_params) + (size_t)(nv & 1u) * c->piece;
  e = hipMemcpyAsync(dstw, master, c->piece * 4, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) return (int)e;
  const float* slots[2] = {slot0, slot1};
  for (int k = 0; k < c->nslot; ++k) {
    float* ds = (float*)(base + c->off_slots) + (size_t)k * c->piece;
    e = slots[k] ? hipMemcpyAsync(ds, slots[k], c->piece * 4, hipMemcpyDeviceToDevice, st)
                 : hipMemsetAsync(ds, 0, c->piece * 4, st);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(xps_set_word_kernel, dim3(1), dim3(64), 0, st, &((XpsCtrl*)base)->applied, applied);
  hipLaunchKernelGGL(xps_set_word_kernel, dim3(1), dim3(64), 0, st, &((XpsCtrl*)base)->version, nv);
  return (int)hipGetLastError();
}

extern "C" int dtfa_xps_join(void* ctx, hipStream_t st) {
  XpsCtx* c = (XpsCtx*)ctx;
  if (!connected(c)) return -2;
  if (c->widx < 0) return 0;
  hipLaunchKernelGGL(xps_join_kernel, dim3(1), dim3(64), 0, st, c->ptrs, c->priv, c->nown, c->nwrk, c->widx);
  return (int)hipGetLastError();
}

// staleness < 0: no bound, nothing to wait for.
extern "C" int dtfa_xps_gate(void* ctx, int staleness, hipStream_t st) {
  XpsCtx* c = (XpsCtx*)ctx;
  if (!connected(c)) return -2;
  if (c->widx < 0 || c->nwrk == 1 || staleness < 0) return 0;
  hipLaunchKernelGGL(xps_gate_kernel, dim3(1), dim3(64), 0, st, c->ptrs, c->priv, c->nwrk, c->widx, staleness,
                     ticks_of(c, c->timeout_s));
  return (int)hipGetLastError();
}

static unsigned blocks_per_owner(size_t piece, int nown, size_t max_blocks = 1024) {
  size_t b = (piece / 4 + 255) / 256;
  size_t cap = max_blocks / (size_t)nown;
  if (cap == 0) cap = 1;
  if (b > cap) b = cap;
  return (unsigned)(b ? b : 1);
}

// dst: the group's nown * piece values in the compute dtype (may be null); dst32: the same in fp32 (may be null).
// Only the elements [lo, hi) are refreshed (multiples of 4; hi <= 0: up to the end).
extern "C" int dtfa_xps_pull_range(void* ctx, void* dst, int dst_is_bf16, float* dst32, long long lo, long long hi,
                                   hipStream_t st) {
  XpsCtx* c = (XpsCtx*)ctx;
  if (!connected(c)) return -2;
  if (dst == nullptr && dst32 == nullptr) return -1;
  const long long total = (long long)c->piece * c->nown;
  if (hi <= 0 || hi > total) hi = total;
  if (lo < 0 || lo >= hi || (lo & 3) || (hi & 3)) return -1;
  // (sized for the range: a chunk of an eighth of the group does not need the whole group's grid)
  size_t span = (size_t)(hi - lo) < c->piece ? (size_t)(hi - lo) : c->piece;
  unsigned bpo = blocks_per_owner(span, c->nown);
  if (dst_is_bf16)
    hipLaunchKernelGGL(xps_pull_kernel<true>, dim3(bpo * c->nown), dim3(256), 0, st, c->ptrs, c->priv, c->nown, c->piece,
                       c->off_params, dst, dst32, bpo, c->report, (size_t)lo, (size_t)hi);
  else
    hipLaunchKernelGGL(xps_pull_kernel<false>, dim3(bpo * c->nown), dim3(256), 0, st, c->ptrs, c->priv, c->nown, c->piece,
                       c->off_params, dst, dst32, bpo, c->report, (size_t)lo, (size_t)hi);
  return (int)hipGetLastError();
}

extern "C" int dtfa_xps_pull(void* ctx, void* dst, int dst_is_bf16, float* dst32, hipStream_t st) {
  return dtfa_xps_pull_range(ctx, dst, dst_is_bf16, dst32, 0, 0, st);
}

// grad: nown * piece gradient values in the context's gradient dtype.
extern "C" int dtfa_xps_push(void* ctx, const void* grad, hipStream_t st) {
  XpsCtx* c = (XpsCtx*)ctx;
  if (!connected(c)) return -2;
  if (c->widx < 0 || ((uintptr_t)grad) % 16) return -1;
  hipLaunchKernelGGL(xps_push_wait_kernel, dim3(1), dim3(64), 0, st, c->ptrs, c->priv, c->nown, c->widx,
                     (c->oidx >= 0 && c->widx >= 0) ? 1 : 0, ticks_of(c, c->timeout_s));
  // (every block ends in a system-scope release, which costs the same whatever the block stored: few, fat blocks)
  static const size_t push_blocks = getenv("DTFA_XPS_PUSH_BLOCKS") ? (size_t)atoi(getenv("DTFA_XPS_PUSH_BLOCKS")) : 256;
  const unsigned bpo = blocks_per_owner(c->piece, c->nown, push_blocks);
  if (c->gbytes == 2)
    hipLaunchKernelGGL(xps_push_kernel<bf16>, dim3(bpo * c->nown), dim3(256), 0, st, c->ptrs, c->priv, c->nown, c->widx,
                       c->piece, c->off_mail, (const bf16*)grad, bpo, c->report);
  else
    hipLaunchKernelGGL(xps_push_kernel<float>, dim3(bpo * c->nown), dim3(256), 0, st, c->ptrs, c->priv, c->nown, c->widx,
                       c->piece, c->off_mail, (const float*)grad, bpo, c->report);
  return (int)hipGetLastError();
}

// Applier side: consume arrived gradients of the own region -- and, with adopt_others, of every region whose own
// process has gone and that this process has taken over (or takes over now) -- publish, optionally count one worker
// step.  A process without a region only counts the step.  wait_any: sleep up to idle_s until something has arrived
// (the serving loop of an owner-only process).  hyper: device float[2] = (learning rate or Adam step size,
// gradient scale).
extern "C" int dtfa_xps_apply(void* ctx, int kind, float b1, float b2, float eps, float mu, float wd, int nesterov,
                              int host_step_size, int aggregate, int advance_clock, int adopt_others, int wait_any,
                              double idle_s, const float* hyper, hipStream_t st) {
  XpsCtx* c = (XpsCtx*)ctx;
  if (!connected(c)) return -2;
  if ((kind == 0 && c->nslot != 2) || (kind == 1 && c->nslot != 1) || kind < 0 || kind > 1) return -1;
  if (c->oidx >= 0) {
    XpsOpt o;
    o.kind = kind; o.b1 = b1; o.b2 = b2; o.eps = eps; o.mu = mu; o.wd = wd; o.nesterov = nesterov;
    o.host_step_size = host_step_size;
    // one workgroup per region; without adopt_others only the own region's does anything
    const int self = c->oidx;
    hipLaunchKernelGGL(xps_collect_kernel, dim3(c->nown), dim3(64), 0, st, c->ptrs, c->priv, c->nown, c->nwrk, self,
                       c->widx, adopt_others, aggregate, wait_any, ticks_of(c, c->timeout_s), ticks_of(c, idle_s));
    size_t b = (c->piece / 4 + 255) / 256;
    if (b > 1024) b = 1024;
    const unsigned bpr = (unsigned)b;
    if (c->gbytes == 2)
      hipLaunchKernelGGL(xps_apply_kernel<bf16>, dim3(bpr * c->nown), dim3(256), 0, st, c->ptrs, c->priv, c->piece,
                         c->off_params, c->off_slots, c->off_mail, o, hyper, bpr);
    else
      hipLaunchKernelGGL(xps_apply_kernel<float>, dim3(bpr * c->nown), dim3(256), 0, st, c->ptrs, c->priv, c->piece,
                         c->off_params, c->off_slots, c->off_mail, o, hyper, bpr);
    hipLaunchKernelGGL(xps_commit_kernel, dim3(c->nown), dim3(64), 0, st, c->ptrs, c->priv, self, c->widx, advance_clock,
                       c->report);
  } else if (advance_clock) {
    hipLaunchKernelGGL(xps_commit_kernel, dim3(1), dim3(64), 0, st, c->ptrs, c->priv, -1, c->widx, 1, c->report);
  }
  return (int)hipGetLastError();
}

extern "C" int dtfa_xps_leave(void* ctx, hipStream_t st) {
  XpsCtx* c = (XpsCtx*)ctx;
  if (!connected(c)) return 0;
  if (c->widx >= 0)
    hipLaunchKernelGGL(xps_set_word_kernel, dim3(1), dim3(64), 0, st, &((XpsCtrl*)c->ptrs.base[0])->wdone[c->widx], 1u);
  if (c->oidx >= 0)
    hipLaunchKernelGGL(xps_set_word_kernel, dim3(1), dim3(64), 0, st, &((XpsCtrl*)c->local)->odone, 1u);
  return (int)hipGetLastError();
}

// The chief asks the owner-only processes to stop serving.
extern "C" int dtfa_xps_shutdown(void* ctx, hipStream_t st) {
  XpsCtx* c = (XpsCtx*)ctx;
  if (!connected(c)) return -2;
  for (int o = 0; o < c->nown; ++o)
    hipLaunchKernelGGL(xps_set_word_kernel, dim3(1), dim3(64), 0, st, &((XpsCtrl*)c->ptrs.base[o])->shutdown, 1u);
  return (int)hipGetLastError();
}

// Copy optimizer slot k of shard o (the whole shard) into dst (device, piece floats).
extern "C" int dtfa_xps_read_slot(void* ctx, int o, int k, float* dst, hipStream_t st) {
  XpsCtx* c = (XpsCtx*)ctx;
  if (o < 0 || o >= c->nown || k < 0 || k >= c->nslot || c->ptrs.base[o] == nullptr) return -1;
  const float* src = (const float*)(c->ptrs.base[o] + c->off_slots) + (size_t)k * c->piece;
  return (int)hipMemcpyAsync(dst, src, c->piece * 4, hipMemcpyDeviceToDevice, st);
}

// Words the kernels reported last, read without synchronising: out[0] = optimizer steps applied to shard 0 as of
// the last pull, out[1] = status as of the last push, out[2] = steps applied to the own shard as of the last commit.
extern "C" int dtfa_xps_report(void* ctx, unsigned* out) {
  XpsCtx* c = (XpsCtx*)ctx;
  const volatile XpsReport* r = c->report;
  out[0] = r->applied0;
  out[1] = r->status;
  out[2] = r->own_applied;
  return 0;
}

// out[0..95] (host), synchronises the stream:
//   0 own worker clock, 1 own version, 2 own applied, 3 applied of shard 0, 4 status, 5 torn pulls, 6 dropped pushes,
//   7 timeouts, 8 dead-worker mask, 9 pushes made, 10 done-worker mask, 11 adopted-region mask (by this process),
//   12 joined-worker mask, 13 shutdown word of the own region (or region 0), 14 written-off region mask,
//   15 max applied over the shards,
//   16..31 clock of every worker, 32..47 applied of every shard, 48..63 adopter word of every shard,
//   64..79 pushes consumed per worker in the own region (or region 0), 80..95 pushes arrived per worker there.
extern "C" int dtfa_xps_state(void* ctx, unsigned* out, hipStream_t st) {
  XpsCtx* c = (XpsCtx*)ctx;
  (void)hipStreamSynchronize(st);
  XpsCtrl r0, mine, other;
  XpsPriv* pv = new XpsPriv();
  hipError_t e = hipMemcpy(pv, c->priv, sizeof(XpsPriv), hipMemcpyDeviceToHost);
  if (e == hipSuccess && c->ptrs.base[0] != nullptr) e = hipMemcpy(&r0, c->ptrs.base[0], sizeof(XpsCtrl), hipMemcpyDeviceToHost);
  if (e != hipSuccess || c->ptrs.base[0] == nullptr) {
    delete pv;
    return e != hipSuccess ? (int)e : -2;
  }
  mine = r0;
  if (c->oidx > 0) e = hipMemcpy(&mine, c->local, sizeof(XpsCtrl), hipMemcpyDeviceToHost);
  if (e != hipSuccess) {
    delete pv;
    return (int)e;
  }
  memset(out, 0, 96 * sizeof(unsigned));
  out[0] = c->widx >= 0 ? r0.wclock[c->widx] : 0u;
  out[1] = mine.version;
  out[2] = mine.applied;
  out[3] = r0.applied;
  out[4] = pv->status;
  out[5] = pv->torn_pulls;
  out[6] = pv->dropped_pushes;
  out[7] = pv->timeouts;
  out[9] = pv->push_seq;
  out[11] = pv->adopted_mask;
  out[13] = mine.shutdown;
  for (int w = 0; w < c->nwrk; ++w) {
    if (pv->dead[w]) out[8] |= 1u << w;
    if (r0.wdone[w]) out[10] |= 1u << w;
    if (r0.wjoined[w]) out[12] |= 1u << w;
    out[16 + w] = r0.wclock[w];
    out[64 + w] = mine.seq_done[w];
    out[80 + w] = mine.seq_in[w];
  }
  for (int o = 0; o < c->nown; ++o) {
    if (pv->odead[o]) out[14] |= 1u << o;
    if (c->ptrs.base[o] == nullptr) continue;
    if (o == 0) other = r0;
    else if (o == c->oidx) other = mine;
    else {
      e = hipMemcpy(&other, c->ptrs.base[o], sizeof(XpsCtrl), hipMemcpyDeviceToHost);
      if (e != hipSuccess) {
        delete pv;
        return (int)e;
      }
    }
    out[32 + o] = other.applied;
    out[48 + o] = other.adopter;
    if (other.applied > out[15]) out[15] = other.applied;
  }
  delete pv;
  return 0;
}

extern "C" void dtfa_xps_destroy(void* ctx) {
  XpsCtx* c = (XpsCtx*)ctx;
  if (c == nullptr) return;
  (void)hipDeviceSynchronize();
  for (int o = 0; o < c->nown; ++o)
    if (o != c->oidx && c->ptrs.base[o] != nullptr) (void)hipIpcCloseMemHandle(c->ptrs.base[o]);
  if (c->local) (void)hipFree(c->local);
  (void)hipFree(c->priv);
  if (c->report) (void)hipHostFree(c->report);
  delete c;
}
