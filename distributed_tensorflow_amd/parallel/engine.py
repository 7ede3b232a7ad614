"""Collective data-parallel training engine (one process per GPU).

The reference's only distribution strategy is asynchronous parameter-server
training over TensorFlow's gRPC runtime: variables and ``ApplyAdam`` live on the
ps, workers pull variables and push gradients every step
(recognition_flower/distributed/train_dataset.py:141-165).  On an 8-GPU MI355X
node the same roles map onto RCCL collectives over xGMI:

``sync_dp``
    Replicated parameters; every gradient bucket is **all-reduced** as soon as
    the backward pass has filled it (overlapping communication with the rest of
    backward), then every rank applies the same fused optimizer step.

``sharded_ps``
    "Every GPU is the ps for 1/N of every bucket".  Push = **reduce-scatter**
    of the gradient bucket (each rank receives the sum for the slice it owns),
    apply = fused optimizer on the owned shard (master weights and optimizer
    slots exist only on the owner), pull = **all-gather** of the updated bf16
    parameters.  Same arithmetic as ``sync_dp`` in a fixed reduction order,
    1/N of the optimizer memory and work.

``async_ssp``
    Staleness-bounded asynchronous updates: the all-reduce of step *t* runs in
    the background and is applied when step *t + s* finishes, so compute never
    waits for communication of the current step and every replica sees the
    same, at most ``s``-steps-stale parameters (``s = 0`` degenerates to
    ``sync_dp``).  The bound is enforced by construction instead of by the
    reference's unbounded races.  Stale gradients narrow the stable step-size
    range, most of all under momentum (one step of delay with heavy-ball
    momentum ``mu``: ``lr * curvature < 1 - mu`` instead of ``< 2 (1 + mu)``),
    so this mode wants a correspondingly smaller rate than ``sync_dp``.  (First-order delay compensation,
    ``g + lambda g*g (w_now - w_then)``, was tried on the ResNet-50 benchmark step and does not widen that range:
    with momentum 0.9 and the sync rate the loss diverges for every lambda in 0.5 .. 8, as without it.)

Buckets are sized for the xGMI mesh: a ring collective moves ``bucket/N`` per
link per step, so ``bucket_mb`` defaults to 32 MiB (4 MiB messages at N = 8)
which keeps transfers bandwidth- rather than latency-bound while still giving
ResNet-50 several buckets to overlap with backward.  ``bucket_mb=0`` picks the size: 32 MiB for models with at
least 64 MiB of gradient, a quarter of the gradient (at least 1 MiB) below that.

The same code runs on CPU tensors with the ``gloo`` backend; that is how the
multi-process tests exercise it.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from ..optim import OptRegion, make_optimizer, regions_from_flat
from ..optim.fused import decay_spec
from .flat import Bucket, FlatGroup, FlatParams

MODES = ("sync_dp", "sharded_ps", "async_ssp")


class DataParallelEngine:
    def __init__(self, model: nn.Module, *, mode: str = "sync_dp", optimizer: str = "momentum", lr=0.1,
                 opt_kwargs: Optional[dict] = None, bucket_mb: float = 32.0, staleness: int = 1,
                 process_group=None, grad_dtype: torch.dtype = torch.float32, p2p_max_bytes: int = 0,
                 overlap_gather: bool = True):
        if mode not in MODES:
            raise ValueError("mode must be one of %s" % (MODES,))
        self.model = model
        self.mode = mode
        self.pg = process_group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(self.pg) if self.distributed else 1
        self.rank = dist.get_rank(self.pg) if self.distributed else 0
        # collectives are skipped in a world of one, unless DTFA_FORCE_COLLECTIVES=1 asks for them anyway
        # (a one-rank RCCL run then exercises the same calls as a multi-GPU job)
        self.collective = self.distributed and (self.world > 1 or os.environ.get("DTFA_FORCE_COLLECTIVES") == "1")
        self.staleness = max(0, int(staleness)) if mode == "async_ssp" else 0
        self.bucket_source = "given"
        shard_multiple = self.world if mode == "sharded_ps" else 1
        if bucket_mb <= 0:
            # automatic: 32 MiB for models with at least two such buckets of gradient; a small model (the flower CNN
            # has 8 MiB) is cut in four, so that all but the last quarter is reduced while backward still runs
            # instead of one collective after it
            esz = torch.empty((), dtype=grad_dtype).element_size()
            total_mb = sum(p.numel() for p in model.parameters() if p.requires_grad) * esz / float(1 << 20)
            bucket_mb = 32.0 if total_mb >= 64.0 else max(1.0, total_mb / 4.0)
            # ... unless this machine's table has a measured winner for this world, mode and model size
            # (tools/tune_buckets.py; parallel/rccl.py)
            from .rccl import tuned_bucket_mb

            measured = tuned_bucket_mb(self.world, mode, total_mb) if self.world > 1 else None
            self.bucket_source = "rule" if measured is None else "measured"
            if measured is not None:
                bucket_mb = measured
        self.flat = FlatParams(model, bucket_mb=bucket_mb, shard_multiple=shard_multiple, grad_dtype=grad_dtype)
        self.flat.on_bucket_ready = self._on_bucket_ready
        self._handles: List = []
        self._step = 0
        # communication accounting (``measure_comm``): device events around the point where the compute stream
        # waits for the step's collectives; their distance is the communication time NOT hidden behind backward
        self.measure_comm = False
        self._comm_events: List = []
        self._comm_calls = 0
        # sharded mode: the all-gathers of the updated parameters stay in flight after ``finish_step``; the next
        # forward waits for a bucket right before the first layer that reads it (ops.functional.param_fence), so the
        # gather of the last layers' parameters runs while the first layers compute
        self.overlap_gather = bool(overlap_gather) and mode == "sharded_ps"
        self._pending_gather: Dict[int, object] = {}
        self._gather_events: List = []
        self._gather_steps = 0
        opt_kwargs = dict(opt_kwargs or {})
        if mode == "sharded_ps":
            self._build_shards()
            regions = self._shard_regions
        else:
            regions = regions_from_flat(self.flat)
        self.optimizer = make_optimizer(optimizer, regions, lr, **opt_kwargs)
        if mode == "async_ssp" and self.staleness > 0:
            # ring of staged gradient copies: slot (t mod (s+1)) holds step t's reduced gradient
            self._ring: List[Dict[str, torch.Tensor]] = [
                {g.name: torch.zeros_like(g.grad) for g in self.flat.groups} for _ in range(self.staleness + 1)
            ]
            self._ring_handles: List[List] = [[] for _ in range(self.staleness + 1)]
            # version counter of every ring slot: the step whose gradient it holds.  DTFA_DEBUG_STALENESS=1
            # asserts at every apply that the gradient is exactly ``staleness`` steps old.
            self._ring_version: List[int] = [-1] * (self.staleness + 1)
            # first step whose gradient has not been applied yet (``drain`` applies ahead of schedule and moves
            # this forward, so that ``finish_step`` never applies a staged gradient a second time)
            self._applied_through = 0
            self._debug_staleness = os.environ.get("DTFA_DEBUG_STALENESS") == "1"
        # buckets up to ``p2p_max_bytes`` skip RCCL and use the one-shot xGMI mailbox all-reduce
        self._p2p = None
        if p2p_max_bytes > 0 and self.collective and mode == "sync_dp" \
                and self.flat.groups[0].grad.is_cuda:
            from .p2p import P2PAllReduce

            self._p2p = P2PAllReduce(p2p_max_bytes, group=self.pg)
        if self.overlap_gather:
            from ..ops import functional as F

            for m in model.modules():
                own = [self.flat._bucket_of[id(p)] for p in m._parameters.values()
                       if p is not None and id(p) in self.flat._bucket_of]
                if own:
                    m._dtfa_buckets = list({id(b): b for b in own}.values())
            F.set_param_fence(self._param_fence)
        if self.collective:
            self.broadcast_parameters()

    # ------------------------------------------------------------------ setup
    def broadcast_parameters(self, src: int = 0) -> None:
        """Make every replica start from rank ``src``'s parameters and buffers
        (the reference initialises variables once, on the ps)."""
        self._wait_all_gathers()
        for g in self.flat.groups:
            dist.broadcast(g.master, src, group=self.pg)
            g.sync_compute_from_master()
        for b in self.model.buffers():
            if b.is_floating_point():
                dist.broadcast(b.data, src, group=self.pg)
        if self.mode == "sharded_ps":
            self._load_shards_from_master()

    def _pieces(self, g: FlatGroup):
        """(bucket, shard-space offset, piece length) for every bucket of ``g``."""
        for b in g.buckets:
            piece = b.numel // self.world
            yield b, b.start // self.world, piece

    def _build_shards(self) -> None:
        self._master_shard: Dict[str, torch.Tensor] = {}
        self._grad_shard: Dict[str, torch.Tensor] = {}
        self._w16_shard: Dict[str, Optional[torch.Tensor]] = {}
        self._shard_regions: List[OptRegion] = []
        for g in self.flat.groups:
            n = g.total // self.world
            dev = g.master.device
            self._master_shard[g.name] = torch.zeros(n, dtype=torch.float32, device=dev)
            self._grad_shard[g.name] = torch.zeros(n, dtype=g.grad.dtype, device=dev)
            self._w16_shard[g.name] = (torch.zeros(n, dtype=g.compute.dtype, device=dev)
                                       if g.compute is not g.master else None)
            index = None
            if not g.master.is_cuda:
                index = torch.cat([torch.arange(b.start + self.rank * p, b.start + (self.rank + 1) * p)
                                   for b, _, p in self._pieces(g)])
            self._shard_regions.append(OptRegion(g.name, self._master_shard[g.name], self._grad_shard[g.name],
                                                 self._w16_shard[g.name], decay_spec(g, 0, g.total, index)
                                                 if index is not None else decay_spec(g, 0, g.total)))
        self._load_shards_from_master()

    def _load_shards_from_master(self) -> None:
        for g in self.flat.groups:
            ms = self._master_shard[g.name]
            for b, off, p in self._pieces(g):
                ms[off:off + p].copy_(g.master[b.start + self.rank * p: b.start + (self.rank + 1) * p])

    # ------------------------------------------------------------------ step
    def begin_step(self) -> None:
        self.flat.begin_step()
        self._handles = []

    def _join_side_streams(self) -> None:
        # weight-gradient kernels may run on a side stream (ops.functional.WGRAD_STREAM): whoever consumes the
        # flat gradient buffer -- a collective, the optimizer -- first makes the current stream wait for them
        dev = self.flat.device
        if dev.type == "cuda":
            from ..ops import functional as F

            F.join_wgrad_stream(dev)

    def _on_bucket_ready(self, b: Bucket) -> None:
        if not self.collective:
            return
        self._join_side_streams()
        g = b.group
        grad = g.grad[b.start:b.end]
        if self.mode == "sync_dp" or (self.mode == "async_ssp" and self.staleness == 0):
            if self._p2p is not None and self._p2p.fits(grad):
                self._p2p.all_reduce_(grad)  # in stream order, nothing to wait for on the host
            else:
                self._handles.append(dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        elif self.mode == "sharded_ps":
            p = b.numel // self.world
            off = b.start // self.world
            out = self._grad_shard[g.name][off:off + p]
            self._handles.append(dist.reduce_scatter_tensor(out, grad, op=dist.ReduceOp.SUM, group=self.pg,
                                                            async_op=True))
        else:  # async_ssp, s > 0: stage a copy and reduce it in the background
            slot = self._step % (self.staleness + 1)
            stage = self._ring[slot][g.name][b.start:b.end]
            stage.copy_(grad)
            self._ring_handles[slot].append(dist.all_reduce(stage, op=dist.ReduceOp.SUM, group=self.pg,
                                                            async_op=True))

    @property
    def graph_capturable(self) -> bool:
        """Every step enqueues the same device work (no step-dependent buffer rotation)."""
        # (the mailbox all-reduce passes its sequence number as a kernel argument)
        return not (self.mode == "async_ssp" and self.staleness > 0) and self._p2p is None

    def prepare_step(self) -> None:
        """Host part of the optimizer step (schedule -> device hyper-parameters).  ``finish_step``
        does this itself unless told that it was done, which is how a captured step graph keeps the
        upload outside the graph."""
        self.optimizer.prepare(1.0 / self.world)

    def advance_step(self) -> None:
        """Book-keeping after a replayed step graph (the graph holds only device work)."""
        self.optimizer.advance()
        self._step += 1

    def _wait_handles(self, handles) -> None:
        """Make the current stream wait for the collectives of this step."""
        if not handles:
            return
        timed = self.measure_comm and self.flat.device.type == "cuda"
        if timed:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        for h in handles:
            h.wait()
        if timed:
            b.record()
            self._comm_events.append((a, b))
        self._comm_calls += len(handles)

    def comm_stats(self) -> dict:
        """Bucket layout and, after a ``measure_comm`` run, the exposed collective wait per step (ms).  Call after a
        device synchronisation."""
        buckets = [b for g in self.flat.groups for b in g.buckets]
        esz = self.flat.groups[0].grad.element_size() if self.flat.groups else 4
        if self.mode == "sharded_ps":
            self._wait_all_gathers()   # (the last step's gathers have no next forward to be waited in)
            if self.flat.device.type == "cuda":
                torch.cuda.synchronize(self.flat.device)
        out = {"collective": bool(self.collective), "world": self.world, "mode": self.mode,
               "grad_dtype": str(self.flat.groups[0].grad.dtype).replace("torch.", "") if self.flat.groups else None,
               "buckets": len(buckets), "bucket_bytes": [int(b.numel * esz) for b in buckets],
               "bucket_size_from": self.bucket_source,
               "gradient_bytes": int(sum(b.numel for b in buckets) * esz),
               "collectives_waited": self._comm_calls}
        if self._comm_events:
            ms = [a.elapsed_time(b) for a, b in self._comm_events]
            out["exposed_wait_ms_per_step"] = round(sum(ms) / len(ms), 4)
            out["exposed_wait_ms_max"] = round(max(ms), 4)
        if self.mode == "sharded_ps":
            out["overlap_gather"] = self.overlap_gather
            if self._gather_events and self._gather_steps:
                # time the compute stream stood still for parameter all-gathers (inside the next forward when they
                # overlap it, at the end of the step otherwise), per step
                out["exposed_gather_ms_per_step"] = round(sum(a.elapsed_time(b) for a, b in self._gather_events)
                                                          / self._gather_steps, 4)
        return out

    def reset_comm_stats(self) -> None:
        self._comm_events, self._comm_calls = [], 0
        self._gather_events, self._gather_steps = [], 0

    def _opt_step(self, scale: float, prepared: bool) -> None:
        if not prepared:
            self.optimizer.prepare(scale)
        self.optimizer.launch()
        if not prepared:
            self.optimizer.advance()

    def finish_step(self, prepared: bool = False) -> None:
        """Call after ``loss.backward()``: completes gradient communication,
        applies the optimizer and (sharded mode) re-assembles parameters.  With ``prepared`` the
        caller has run :meth:`prepare_step` and will run :meth:`advance_step` (graph capture)."""
        if prepared and not self.graph_capturable:
            raise ValueError("async_ssp with staleness > 0 rotates buffers per step and cannot be captured")
        self._wait_all_gathers()   # (parameters no layer asked for: the optimizer is about to rewrite their source)
        self.flat.collect_autograd_grads()
        self._join_side_streams()
        scale = 1.0 / self.world
        if self.mode == "sharded_ps":
            if not self.collective:
                for g in self.flat.groups:
                    self._grad_shard[g.name].copy_(g.grad)
            self._wait_handles(self._handles)
            self._opt_step(scale, prepared)
            # (a captured step must end with its own gathers complete: the replay has no "next forward" to wait in)
            self._gather_params(defer=self.overlap_gather and not prepared)
            self._gather_steps += 1
        elif self.mode == "async_ssp" and self.staleness > 0:
            s = self.staleness
            slot = self._step % (s + 1)
            if not self.collective:
                for g in self.flat.groups:
                    self._ring[slot][g.name].copy_(g.grad)
            self._ring_version[slot] = self._step
            if self._step - s >= self._applied_through:
                apply_slot = (self._step - s) % (s + 1)
                if self._debug_staleness and self._ring_version[apply_slot] != self._step - s:
                    raise AssertionError("staleness violation: applying the gradient of step %d at step %d (bound %d)"
                                         % (self._ring_version[apply_slot], self._step, s))
                self._wait_handles(self._ring_handles[apply_slot])
                self._ring_handles[apply_slot] = []
                for r in self.optimizer.regions:
                    r.g = self._ring[apply_slot][r.name]
                self.optimizer.step(grad_scale=scale)
                self._applied_through = self._step - s + 1
        else:
            self._wait_handles(self._handles)
            self._opt_step(scale, prepared)
        self._handles = []
        if not prepared:
            self._step += 1

    def drain(self) -> None:
        """Apply every gradient still in flight (async mode), e.g. before a checkpoint.  Collective: every rank
        must call it at the same step, or the replicas apply the same gradients at different times and diverge
        (``TrainingSession`` therefore runs the checkpoint hook on all ranks in this mode)."""
        if self.mode != "async_ssp" or self.staleness == 0:
            return
        s = self.staleness
        for t in range(self._applied_through, self._step):
            slot = t % (s + 1)
            for h in self._ring_handles[slot]:
                h.wait()
            self._ring_handles[slot] = []
            for r in self.optimizer.regions:
                r.g = self._ring[slot][r.name]
            self.optimizer.step(grad_scale=1.0 / self.world)
        self._applied_through = self._step

    def _gather_params(self, defer: bool = False) -> None:
        """All-gather the updated parameter shards, bucket by bucket.  Buckets are laid out last layer first, and a
        forward pass needs them first layer first: the collectives are issued in reverse bucket order, and with
        ``defer`` their handles are kept for :meth:`_param_fence` instead of being waited for here."""
        jobs = []
        for g in self.flat.groups:
            src_all = self._w16_shard[g.name] if self._w16_shard[g.name] is not None else self._master_shard[g.name]
            for b, off, p in self._pieces(g):
                jobs.append((b, src_all[off:off + p], g.compute[b.start:b.end]))
        jobs.sort(key=lambda j: -j[0].start)   # within a group: the bucket of the first layers first
        handles = []
        for b, src, dst in jobs:
            if self.collective:
                h = dist.all_gather_into_tensor(dst, src, group=self.pg, async_op=True)
                if defer:
                    self._pending_gather[id(b)] = h
                else:
                    handles.append(h)
            else:
                dst.copy_(src)
        if handles:
            self._timed_gather_wait(handles)

    def _timed_gather_wait(self, handles) -> None:
        timed = self.measure_comm and self.flat.device.type == "cuda"
        if timed:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        for h in handles:
            h.wait()
        if timed:
            b.record()
            self._gather_events.append((a, b))
        self._comm_calls += len(handles)

    def _param_fence(self, modules) -> None:
        """Called right before ``modules``' parameters are read: wait for the gathers that cover them."""
        if not self._pending_gather:
            return
        hs = []
        for m in modules:
            for b in getattr(m, "_dtfa_buckets", ()):
                h = self._pending_gather.pop(id(b), None)
                if h is not None:
                    hs.append(h)
        if hs:
            self._timed_gather_wait(hs)

    def _wait_all_gathers(self) -> None:
        if self._pending_gather:
            hs = list(self._pending_gather.values())
            self._pending_gather.clear()
            self._timed_gather_wait(hs)

    # ------------------------------------------------------------------ export
    def full_master(self) -> Dict[str, torch.Tensor]:
        """fp32 master parameters of every flat group (gathers shards)."""
        self._wait_all_gathers()
        out = {}
        for g in self.flat.groups:
            if self.mode != "sharded_ps":
                out[g.name] = g.master
                continue
            full = torch.empty_like(g.master)
            ms = self._master_shard[g.name]
            for b, off, p in self._pieces(g):
                dst = full[b.start:b.end]
                if self.collective:
                    dist.all_gather_into_tensor(dst, ms[off:off + p].contiguous(), group=self.pg)
                else:
                    dst.copy_(ms[off:off + p])
            g.master.copy_(full)
            out[g.name] = g.master
        return out

    def full_slots(self) -> Dict[str, torch.Tensor]:
        """Optimizer slots in full (un-sharded) group layout, keyed ``<slot>/<group>``."""
        slots = self.optimizer.slots()
        if self.mode != "sharded_ps":
            return slots
        out = {}
        groups = {g.name: g for g in self.flat.groups}
        for key, t in slots.items():
            gname = key.rsplit("/", 1)[1]
            g = groups[gname]
            full = torch.zeros(g.total, dtype=t.dtype, device=t.device)
            for b, off, p in self._pieces(g):
                dst = full[b.start:b.end]
                if self.collective:
                    dist.all_gather_into_tensor(dst, t[off:off + p].contiguous(), group=self.pg)
                else:
                    dst.copy_(t[off:off + p])
            out[key] = full
        return out

    def load_full_slots(self, full: Dict[str, torch.Tensor]) -> None:
        slots = self.optimizer.slots()
        groups = {g.name: g for g in self.flat.groups}
        for key, t in slots.items():
            if key not in full:
                continue
            src = full[key].to(t.device)
            if self.mode != "sharded_ps":
                t.copy_(src)
                continue
            g = groups[key.rsplit("/", 1)[1]]
            for b, off, p in self._pieces(g):
                t[off:off + p].copy_(src[b.start + self.rank * p: b.start + (self.rank + 1) * p])

    def load_master(self) -> None:
        """Re-derive compute copies (and shards) after ``FlatGroup.master`` was overwritten (restore)."""
        self._wait_all_gathers()
        for g in self.flat.groups:
            g.sync_compute_from_master()
        if self.mode == "sharded_ps":
            self._load_shards_from_master()
