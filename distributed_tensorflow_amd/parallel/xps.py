"""GPU-resident asynchronous parameter server over one-sided peer memory (``csrc/hip/xps.hip``).

The reference's distribution strategy is asynchronous between-graph replication: variables and
``ApplyAdam`` live on a ps task, each worker pulls, computes and pushes on its own
(recognition_flower/distributed/train_dataset.py:141-149,162-165), and a worker never depends on another
worker being alive (``device_filters``, train_dataset.py:172-174).  :mod:`parallel.ps` realises that with ps
*processes* over TCP.  This engine is the same contract on one xGMI node without a ps process, a host copy
or a collective on the data path:

* every rank is a worker and **owns** shard ``r`` (1/N) of each flat parameter group: fp32 master values and
  optimizer slots live in fine-grained device memory that all peers map through ``hipIpc``;
* **push**: a worker copies shard ``o`` of its gradient into its mailbox slot in owner ``o``'s memory and
  raises a sequence word there (direct peer stores, one xGMI link per owner);
* **apply**: at the end of its own step an owner consumes whatever has arrived -- one fused Adam / momentum
  step per arrived gradient (``apply="per_push"``, TensorFlow's ps semantics) or, in lock-step rounds, one step
  on the rank-ordered sum of the round (``apply="aggregate"``, the arithmetic of synchronous data
  parallelism) -- and publishes a new parameter version;
* **pull**: a worker reads the current version of every shard straight out of its owner's memory;
* **bounded staleness**: a worker whose clock is more than ``staleness`` steps ahead of the slowest live
  worker waits on the peers' clock words (``-1`` = unbounded, ``0`` = lock step); a peer that does not move
  for ``timeout`` seconds, or has left, stops counting, so a dead worker delays the others once, by one
  timeout, and never again.

All device work of a step is kernels on the current stream that read their sequence numbers from device
memory, so a step is the same launch sequence every time (HIP-graph capturable).  The process group is used
once, to exchange the IPC handles.  The worker set is fixed at start-up; workers may finish or die at any time.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from ..ops import hipk
from ..optim.fused import decay_spec
from ..optim.schedules import Schedule, constant
from .flat import FlatParams

_SIGS = {
    "dtfa_xps_create": [C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_void_p)],
    "dtfa_xps_handle": [C.c_void_p, C.c_char_p],
    "dtfa_xps_connect": [C.c_void_p, C.c_char_p],
    "dtfa_xps_set_shard": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p],
    "dtfa_xps_gate": [C.c_void_p, C.c_int, C.c_void_p],
    "dtfa_xps_pull": [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p],
    "dtfa_xps_push": [C.c_void_p, C.c_void_p, C.c_void_p],
    "dtfa_xps_apply": [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int,
                       C.c_int, C.c_int, C.c_void_p, C.c_void_p],
    "dtfa_xps_leave": [C.c_void_p, C.c_void_p],
    "dtfa_xps_read_slot": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p],
    "dtfa_xps_state": [C.c_void_p, C.POINTER(C.c_uint), C.c_void_p],
}
_sigs_done = False


def _lib():
    global _sigs_done
    L = hipk.lib()
    if not _sigs_done:
        for name, sig in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = sig
            fn.restype = C.c_int
        L.dtfa_xps_destroy.argtypes = [C.c_void_p]
        L.dtfa_xps_destroy.restype = None
        _sigs_done = True
    return L


class _Shard:
    """The peer-visible region of one flat group on this rank (context handle + sizes)."""

    def __init__(self, group, world, rank, nslot, grad_is_bf16, timeout, pg):
        self.group = group
        self.piece = group.total // world
        self.ctx = C.c_void_p()
        L = _lib()
        hipk.check(L.dtfa_xps_create(world, rank, self.piece, nslot, int(grad_is_bf16), float(timeout), C.byref(self.ctx)),
                   "xps_create")
        buf = C.create_string_buffer(64)
        hipk.check(L.dtfa_xps_handle(self.ctx, buf), "xps_handle")
        handles = [None] * world
        if world > 1:
            dist.all_gather_object(handles, buf.raw, group=pg)
        else:
            handles = [buf.raw]
        hipk.check(L.dtfa_xps_connect(self.ctx, b"".join(handles)), "xps_connect")
        self.hyper = torch.zeros(2, dtype=torch.float32, device=group.master.device)


class XgmiPSEngine:
    """Worker + shard-owner engine with the ``begin_step`` / ``finish_step`` interface of the other engines.

    ``staleness``: -1 unbounded (the reference), 0 lock step, s > 0 stale-synchronous.  ``apply``: ``per_push`` |
    ``aggregate`` | ``auto`` (aggregate when ``staleness == 0``)."""

    def __init__(self, model: nn.Module, *, optimizer: str = "adam", lr=1e-3, opt_kwargs: Optional[dict] = None,
                 staleness: int = -1, apply: str = "auto", grad_dtype: torch.dtype = torch.float32,
                 timeout: float = 30.0, process_group=None):
        self.model = model
        self.pg = process_group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(self.pg) if self.distributed else 1
        self.rank = dist.get_rank(self.pg) if self.distributed else 0
        if self.world > 8:
            raise ValueError("the mailbox layout holds at most 8 ranks (one xGMI node)")
        self.mode = "async_ps"
        self.backend = "xgmi"
        self.staleness = int(staleness)
        if apply == "auto":
            apply = "aggregate" if self.staleness == 0 else "per_push"
        if apply not in ("per_push", "aggregate"):
            raise ValueError("apply must be per_push, aggregate or auto")
        if apply == "aggregate" and self.staleness != 0:
            raise ValueError("aggregated rounds are lock-step: they need staleness 0")
        self.apply = apply
        self.timeout = float(timeout)
        self.flat = FlatParams(model, bucket_mb=float(1 << 20), shard_multiple=self.world, grad_dtype=grad_dtype)
        if self.flat.device.type != "cuda":
            raise ValueError("the xGMI parameter server keeps parameters in GPU memory; use --ps_backend=tcp on CPUs")
        kw = dict(opt_kwargs or {})
        self.opt_name = optimizer.lower()
        if self.opt_name == "adam":
            self.kind, self.nslot = 0, 2
            self.b1, self.b2, self.eps = kw.get("beta1", 0.9), kw.get("beta2", 0.999), kw.get("eps", 1e-8)
            self.mu = self.wd = 0.0
            self.nesterov = False
        elif self.opt_name in ("momentum", "sgd"):
            self.kind, self.nslot = 1, 1
            self.b1 = self.b2 = self.eps = 0.0
            self.mu, self.wd = kw.get("momentum", 0.9), kw.get("weight_decay", 0.0)
            self.nesterov = bool(kw.get("nesterov", False))
        else:
            raise ValueError("unknown optimizer %r" % optimizer)
        self.schedule: Schedule = lr if callable(lr) else constant(float(lr))
        self.optimizer = self  # the session asks ``engine.optimizer`` for current_lr / step_count
        self._steps = 0        # steps this worker has run
        self._base_step = 0    # global step at initialisation / restore
        self.shards: List[_Shard] = []
        with torch.cuda.device(self.flat.device):
            for g in self.flat.groups:
                gb = g.grad.dtype == torch.bfloat16
                self.shards.append(_Shard(g, self.world, self.rank, self.nslot, gb, self.timeout, self.pg))
            if self.distributed and self.world > 1:
                # the chief's initial values everywhere (variables are initialised once, by the chief)
                for g in self.flat.groups:
                    dist.broadcast(g.master, 0, group=self.pg)
                for b in model.buffers():
                    if b.is_floating_point():
                        dist.broadcast(b.data, 0, group=self.pg)
            self._upload_own_shards(slots=None, applied=0)
            torch.cuda.synchronize(self.flat.device)
            if self.distributed and self.world > 1:
                dist.barrier(group=self.pg)  # every shard is initialised and mapped before the first pull
            self.pull()
        self._closed = False

    # ------------------------------------------------------------------ helpers
    def _st(self) -> int:
        return torch.cuda.current_stream(self.flat.device).cuda_stream

    def _own(self, t: torch.Tensor, sh: _Shard) -> torch.Tensor:
        return t[self.rank * sh.piece:(self.rank + 1) * sh.piece]

    def _upload_own_shards(self, slots: Optional[Dict[str, torch.Tensor]], applied: int) -> None:
        """This rank's part of ``FlatGroup.master`` (and of full-layout optimizer slots) becomes the owner state."""
        L = _lib()
        kinds = ["Adam", "Adam_1"] if self.kind == 0 else ["Momentum"]
        for sh in self.shards:
            g = sh.group
            sl = []
            for k in kinds:
                full = None if slots is None else slots.get(k + "/" + g.name)
                sl.append(None if full is None else self._own(full.to(g.master.device, torch.float32), sh).contiguous())
            while len(sl) < 2:
                sl.append(None)
            own = self._own(g.master, sh).contiguous()
            hipk.check(L.dtfa_xps_set_shard(sh.ctx, own.data_ptr(), hipk.ptr(sl[0]), hipk.ptr(sl[1]), int(applied),
                                            self._st()), "xps_set_shard")
        self._base_step = int(applied)

    def current_lr(self) -> float:
        return self.schedule(self.global_step_estimate)

    @property
    def global_step_estimate(self) -> int:
        """Host-side estimate used for the learning-rate schedule (no device synchronisation): lock-step rounds
        count one global step per round, asynchronous applies one per push of every worker."""
        per_step = 1 if self.apply == "aggregate" else self.world
        return self._base_step + self._steps * per_step

    # ------------------------------------------------------------------ protocol
    def pull(self, into_master: bool = False) -> None:
        L = _lib()
        for sh in self.shards:
            g = sh.group
            is_bf16 = g.compute.dtype == torch.bfloat16
            dst32 = g.master if (into_master or g.compute is g.master) else None
            dst = g.compute if g.compute is not g.master else None
            hipk.check(L.dtfa_xps_pull(sh.ctx, hipk.ptr(dst), int(is_bf16), hipk.ptr(dst32), self._st()), "xps_pull")

    def begin_step(self) -> None:
        self.flat.begin_step()
        L = _lib()
        for sh in self.shards:
            hipk.check(L.dtfa_xps_gate(sh.ctx, self.staleness, self._st()), "xps_gate")
        self.pull()

    def _step_size(self) -> float:
        lr = self.schedule(self.global_step_estimate)
        if self.kind == 0 and self.apply == "aggregate":
            t = self.global_step_estimate + 1  # one optimizer step per round: the host knows its number
            return lr * math.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)
        return lr

    def prepare_step(self) -> None:
        a = self._step_size()
        for sh in self.shards:
            hipk.check(hipk.lib().dtfa_set_hyper(sh.hyper.data_ptr(), float(a), 1.0, self._st()), "set_hyper")

    def advance_step(self) -> None:
        self._steps += 1

    @property
    def graph_capturable(self) -> bool:
        return True

    def finish_step(self, prepared: bool = False) -> None:
        self.flat.collect_autograd_grads()
        from ..ops import functional as F

        F.join_wgrad_stream(self.flat.device)
        if not prepared:
            self.prepare_step()
        L = _lib()
        host_lr_t = int(self.kind == 0 and self.apply == "aggregate")
        for sh in self.shards:
            hipk.check(L.dtfa_xps_push(sh.ctx, sh.group.grad.data_ptr(), self._st()), "xps_push")
        for sh in self.shards:
            wd = self.wd if decay_spec(sh.group, 0, sh.group.total) is True else 0.0
            hipk.check(L.dtfa_xps_apply(sh.ctx, self.kind, self.b1, self.b2, self.eps, self.mu, wd, int(self.nesterov),
                                        host_lr_t, int(self.apply == "aggregate"), 1, sh.hyper.data_ptr(), self._st()),
                       "xps_apply")
        if not prepared:
            self.advance_step()

    def drain(self) -> None:
        """Owner: apply what has arrived without taking a worker step (e.g. before a checkpoint or at the end)."""
        if self.apply == "aggregate":
            return
        L = _lib()
        self.prepare_step()
        for sh in self.shards:
            wd = self.wd if decay_spec(sh.group, 0, sh.group.total) is True else 0.0
            hipk.check(L.dtfa_xps_apply(sh.ctx, self.kind, self.b1, self.b2, self.eps, self.mu, wd, int(self.nesterov), 0, 0, 0,
                                        sh.hyper.data_ptr(), self._st()), "xps_apply")

    # ------------------------------------------------------------------ state
    def state(self) -> dict:
        """Counters of the first group's region (synchronises the stream)."""
        out = (C.c_uint * 24)()
        hipk.check(_lib().dtfa_xps_state(self.shards[0].ctx, out, self._st()), "xps_state")
        return {"clock": out[0], "version": out[1], "applied": out[2], "global_step": out[3], "status": out[4],
                "torn_pulls": out[5], "dropped_pushes": out[6], "timeouts": out[7], "dead_mask": out[8],
                "pushes": out[9], "done_mask": out[10], "clocks": [out[16 + r] for r in range(self.world)]}

    # -- communication accounting (same interface as DataParallelEngine; bench.py reports it) ----------------
    measure_comm = False

    def reset_comm_stats(self) -> None:
        self._stats_base = self.state()

    def comm_stats(self) -> dict:
        """Protocol counters since :meth:`reset_comm_stats` (synchronises the stream): what was pushed, what was
        dropped or torn, which peers were written off."""
        st = self.state()
        base = getattr(self, "_stats_base", None) or {}
        gbytes = sum(sh.group.grad.numel() * sh.group.grad.element_size() for sh in self.shards)
        pbytes = sum(sh.group.total * 4 for sh in self.shards)

        def since(k):
            return int(st[k]) - int(base.get(k, 0))

        return {"collective": False, "world": self.world, "mode": self.mode, "backend": self.backend,
                "apply": self.apply, "staleness": self.staleness,
                "grad_dtype": str(self.shards[0].group.grad.dtype).replace("torch.", "") if self.shards else None,
                "groups": len(self.shards), "pushed_bytes_per_step": int(gbytes), "pulled_bytes_per_step": int(pbytes),
                "pushes": since("pushes"), "applied": since("applied"), "status": int(st["status"]),
                "dropped_pushes": since("dropped_pushes"), "torn_pulls": since("torn_pulls"),
                "timeouts": since("timeouts"), "dead_mask": int(st["dead_mask"]), "done_mask": int(st["done_mask"]),
                "clocks": st["clocks"]}

    @property
    def global_step(self) -> int:
        """Optimizer steps applied to shard 0 (the analogue of ``global_step`` living on ps task 0)."""
        return int(self.state()["global_step"])

    @property
    def step_count(self) -> int:
        return self.global_step_estimate

    @step_count.setter
    def step_count(self, v: int) -> None:
        self._base_step, self._steps = int(v), 0

    # -- checkpoint support (io.saver.Saver) ------------------------------------------
    def full_master(self) -> Dict[str, torch.Tensor]:
        self.pull(into_master=True)
        torch.cuda.current_stream(self.flat.device).synchronize()
        return {g.name: g.master for g in self.flat.groups}

    def full_slots(self) -> Dict[str, torch.Tensor]:
        L = _lib()
        kinds = ["Adam", "Adam_1"] if self.kind == 0 else ["Momentum"]
        out = {}
        for sh in self.shards:
            g = sh.group
            for k, kind in enumerate(kinds):
                full = torch.zeros(g.total, dtype=torch.float32, device=g.master.device)
                for o in range(self.world):
                    hipk.check(L.dtfa_xps_read_slot(sh.ctx, o, k, full[o * sh.piece:].data_ptr(), self._st()), "xps_read_slot")
                out[kind + "/" + g.name] = full
        return out

    @property
    def fetched_global_step(self) -> Optional[int]:
        return self.global_step

    @property
    def beta1_power(self):
        return self.b1 ** (self.global_step + 1) if self.kind == 0 else None

    @property
    def beta2_power(self):
        return self.b2 ** (self.global_step + 1) if self.kind == 0 else None

    def load_master(self) -> None:
        """After a restore wrote ``FlatGroup.master`` (every rank loads the checkpoint): each owner re-publishes its
        own shard with the restored step count; slots follow in :meth:`load_full_slots`."""
        self._pending_restore = True
        self._upload_own_shards(slots=None, applied=self._base_step)
        self._resync()

    def load_full_slots(self, full: Dict[str, torch.Tensor]) -> None:
        self._upload_own_shards(slots=full, applied=self._base_step)
        self._resync()

    def _resync(self) -> None:
        torch.cuda.current_stream(self.flat.device).synchronize()
        if self.distributed and self.world > 1:
            dist.barrier(group=self.pg)
        self.pull()

    def close(self, shutdown_ps: bool = False) -> None:
        if self._closed:
            return
        self._closed = True
        L = _lib()
        for sh in self.shards:
            hipk.check(L.dtfa_xps_leave(sh.ctx, self._st()), "xps_leave")
        torch.cuda.current_stream(self.flat.device).synchronize()
        # the regions stay allocated until the process exits: peers may still be reading them
