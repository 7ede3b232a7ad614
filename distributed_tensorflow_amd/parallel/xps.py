"""GPU-resident asynchronous parameter server over one-sided peer memory (``csrc/hip/xps.hip``).

The reference's distribution strategy is asynchronous between-graph replication: variables and
``ApplyAdam`` live on ps tasks, each worker pulls, computes and pushes on its own
(recognition_flower/distributed/train_dataset.py:141-149,162-165), and a worker never depends on another
worker being alive (``device_filters``, train_dataset.py:172-174).  :mod:`parallel.ps` realises that with ps
*processes* over TCP.  This module is the same contract on one xGMI node without a host copy or a collective on
the data path: the variables live in GPU memory that every process maps through ``hipIpc``.

* an **owner** holds shard ``o`` (1/``nown``) of each flat parameter group: fp32 master values, optimizer slots
  and one pair of mailbox slots per worker, in fine-grained device memory (its *region*);
* **push**: a worker copies shard ``o`` of its gradient into its mailbox slot in region ``o`` and raises a
  sequence word there (direct peer stores, one xGMI link per owner);
* **apply**: whoever applies a region consumes what has arrived -- one fused Adam / momentum step per arrived
  gradient (``apply="per_push"``, TensorFlow's ps semantics) or, in lock-step rounds, one step on the
  worker-ordered sum of the round (``apply="aggregate"``, the arithmetic of synchronous data parallelism) -- and
  publishes a new parameter version;
* **pull**: a worker reads the current version of every shard straight out of its region;
* **bounded staleness**: a worker whose clock is more than ``staleness`` steps ahead of the slowest live
  worker waits on the workers' clock words (``-1`` = unbounded, ``0`` = lock step); a worker that does not move
  for ``timeout`` seconds, or has left, stops counting.  A process that is itself inside such a bounded wait keeps
  changing a *beat* word, and whoever waits for that process restarts its timer on every change (up to four timeouts
  in all): the survivors of a dead process do not write each other off while each of them waits for it.

Two deployments:

**ps tasks own the variables** (:class:`XgmiPSServer` + :class:`XgmiPSEngine` with a :class:`XpsCluster`): the
reference's topology.  A ps process allocates its regions, publishes their IPC handles in a key-value store hosted by
ps task 0 and then only runs the apply loop; workers are pure clients.  A worker that dies takes nothing with it,
a worker may join late (it maps the handles from the store, takes the slowest live worker's clock and pulls the
current values), and the chief stops the ps tasks when it finishes.

**no ps tasks** (:class:`XgmiPSEngine` on a process group, e.g. under ``torchrun`` / ``bench.py``): every worker
process also owns a shard.  When such a process leaves or dies, its region stays mapped in the other processes and
the lowest-ranked live process takes it over *in place* (compare-and-swap on the region's ``adopter`` word, then
collect / apply / commit with its own kernels through the peer mapping): every shard keeps training, pushers and
pullers notice nothing.  Gradients pushed to a dead owner between its death and the take-over wait in their
mailbox slots; a worker whose slot in that region stays full for a whole ``timeout`` drops that shard of its
gradient (counted in ``dropped_pushes``) instead of blocking.  Whichever wait notices the dead process first (the
staleness gate or the full slot) writes it off for the other one too, so a survivor pays one timeout and drops one
shard of one gradient per dead process.

All device work of a step is kernels on the current stream that read their sequence numbers from device
memory, so a step is the same launch sequence every time (HIP-graph capturable).  Nothing here uses a collective
on the data path; the process group (or the store) only carries IPC handles and the initial values at start-up.
"""
from __future__ import annotations

import ctypes as C
import datetime
import json
import math
import time
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from ..ops import hipk
from ..optim.fused import decay_spec
from ..optim.schedules import Schedule, constant
from .flat import FlatParams

MAX_OWNERS = 16
MAX_WORKERS = 16

_SIGS = {
    "dtfa_xps_create": [C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_double,
                        C.POINTER(C.c_void_p)],
    "dtfa_xps_handle": [C.c_void_p, C.c_char_p],
    "dtfa_xps_connect": [C.c_void_p, C.c_char_p],
    "dtfa_xps_set_shard": [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p],
    "dtfa_xps_join": [C.c_void_p, C.c_void_p],
    "dtfa_xps_gate": [C.c_void_p, C.c_int, C.c_void_p],
    "dtfa_xps_pull": [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p],
    "dtfa_xps_pull_range": [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_longlong, C.c_void_p],
    "dtfa_xps_push": [C.c_void_p, C.c_void_p, C.c_void_p],
    "dtfa_xps_apply": [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int,
                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p],
    "dtfa_xps_leave": [C.c_void_p, C.c_void_p],
    "dtfa_xps_shutdown": [C.c_void_p, C.c_void_p],
    "dtfa_xps_read_slot": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p],
    "dtfa_xps_report": [C.c_void_p, C.POINTER(C.c_uint)],
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


@dataclass
class XpsCluster:
    """ps tasks own the variables: where the rendezvous store lives (ps task 0 hosts it) and who takes part."""

    store_host: str
    store_port: int
    num_ps: int
    num_workers: int
    connect_timeout: float = 120.0

    def store(self, is_master: bool = False):
        return dist.TCPStore(self.store_host, int(self.store_port), None, is_master,
                             timeout=datetime.timedelta(seconds=self.connect_timeout), wait_for_workers=False)


def _state_dict(out, nown: int, nwrk: int) -> dict:
    return {"clock": out[0], "version": out[1], "applied": out[2], "global_step": out[15], "applied_shard0": out[3],
            "status": out[4], "torn_pulls": out[5], "dropped_pushes": out[6], "timeouts": out[7], "dead_mask": out[8],
            "pushes": out[9], "done_mask": out[10], "adopted_mask": out[11], "joined_mask": out[12],
            "shutdown": out[13], "written_off_regions": out[14],
            "clocks": [out[16 + w] for w in range(nwrk)], "applied_per_shard": [out[32 + o] for o in range(nown)],
            "adopters": [out[48 + o] for o in range(nown)], "consumed": [out[64 + w] for w in range(nwrk)],
            "arrived": [out[80 + w] for w in range(nwrk)]}


def _opt_config(optimizer: str, opt_kwargs: Optional[dict]) -> dict:
    kw = dict(opt_kwargs or {})
    name = optimizer.lower()
    if name == "adam":
        return {"kind": 0, "nslot": 2, "b1": kw.get("beta1", 0.9), "b2": kw.get("beta2", 0.999), "eps": kw.get("eps", 1e-8),
                "mu": 0.0, "wd": 0.0, "nesterov": False, "slots": ["Adam", "Adam_1"]}
    if name in ("momentum", "sgd"):
        return {"kind": 1, "nslot": 1, "b1": 0.0, "b2": 0.0, "eps": 0.0, "mu": kw.get("momentum", 0.9),
                "wd": kw.get("weight_decay", 0.0), "nesterov": bool(kw.get("nesterov", False)), "slots": ["Momentum"]}
    raise ValueError("unknown optimizer %r" % optimizer)


class _Region:
    """Native context of one flat group in this process: the own region (if this process owns a shard) and the
    mapped regions of all owners."""

    def __init__(self, name: str, total: int, grad_is_bf16: bool, nown: int, nwrk: int, oidx: int, widx: int, nslot: int,
                 timeout: float, device):
        self.name = name
        self.total = int(total)
        self.piece = self.total // nown
        if self.piece * nown != self.total or self.piece % 8:
            raise ValueError("group %s: %d values do not split into %d shards of whole 32-byte vectors" % (name, total, nown))
        self.ctx = C.c_void_p()
        hipk.check(_lib().dtfa_xps_create(nown, nwrk, oidx, widx, self.piece, nslot, int(grad_is_bf16), float(timeout),
                                          C.byref(self.ctx)), "xps_create")
        self.hyper = torch.zeros(2, dtype=torch.float32, device=device)
        self.no_decay_all = False

    def handle(self) -> bytes:
        buf = C.create_string_buffer(64)
        hipk.check(_lib().dtfa_xps_handle(self.ctx, buf), "xps_handle")
        return buf.raw

    def connect(self, handles: List[bytes]) -> None:
        hipk.check(_lib().dtfa_xps_connect(self.ctx, b"".join(handles)), "xps_connect")

    def state(self, stream: int, nown: int, nwrk: int) -> dict:
        out = (C.c_uint * 96)()
        hipk.check(_lib().dtfa_xps_state(self.ctx, out, stream), "xps_state")
        return _state_dict(out, nown, nwrk)

    def report(self):
        out = (C.c_uint * 4)()
        _lib().dtfa_xps_report(self.ctx, out)
        return out[0], out[1], out[2]


class XgmiPSEngine:
    """Worker (and, without ps tasks, shard-owner) engine with the ``begin_step`` / ``finish_step`` interface of the
    other engines.

    ``staleness``: -1 unbounded (the reference), 0 lock step, s > 0 stale-synchronous.  ``apply``: ``per_push`` |
    ``aggregate`` | ``auto`` (aggregate when ``staleness == 0`` and the workers own the shards).  ``cluster``: ps
    tasks own the variables (see :class:`XpsCluster`); ``worker_index`` is then this worker's index and
    ``defer_init`` lets a resuming chief restore a checkpoint before the ps tasks are initialised."""

    def __init__(self, model: nn.Module, *, optimizer: str = "adam", lr=1e-3, opt_kwargs: Optional[dict] = None,
                 staleness: int = -1, apply: str = "auto", grad_dtype: torch.dtype = torch.float32,
                 timeout: float = 30.0, process_group=None, cluster: Optional[XpsCluster] = None,
                 worker_index: Optional[int] = None, defer_init: bool = False, overlap_pull: bool = True,
                 pull_chunks: int = 8):
        self.model = model
        self.pg = process_group
        self.cluster = cluster
        self.mode = "async_ps"
        self.backend = "xgmi"
        self.staleness = int(staleness)
        self.timeout = float(timeout)
        if cluster is None:
            self.distributed = dist.is_available() and dist.is_initialized()
            self.nwrk = dist.get_world_size(self.pg) if self.distributed else 1
            self.widx = dist.get_rank(self.pg) if self.distributed else 0
            self.nown, self.oidx = self.nwrk, self.widx
        else:
            if worker_index is None:
                raise ValueError("a worker of a ps cluster needs its worker_index")
            self.distributed = False
            self.nwrk, self.widx = int(cluster.num_workers), int(worker_index)
            self.nown, self.oidx = int(cluster.num_ps), -1
        self.world, self.rank = self.nwrk, self.widx
        self.is_chief = self.widx == 0
        if self.nown > MAX_OWNERS or self.nwrk > MAX_WORKERS:
            raise ValueError("the region layout holds at most %d owners and %d workers (one xGMI node)"
                             % (MAX_OWNERS, MAX_WORKERS))
        if apply == "auto":
            apply = "aggregate" if (self.staleness == 0 and cluster is None) else "per_push"
        if apply not in ("per_push", "aggregate"):
            raise ValueError("apply must be per_push, aggregate or auto")
        if apply == "aggregate" and self.staleness != 0:
            raise ValueError("aggregated rounds are lock-step: they need staleness 0")
        if apply == "aggregate" and cluster is not None:
            raise ValueError("aggregated rounds need the workers to own the shards (no ps tasks)")
        self.apply = apply
        self.flat = FlatParams(model, bucket_mb=float(1 << 20), shard_multiple=self.nown, grad_dtype=grad_dtype)
        if self.flat.device.type != "cuda":
            raise ValueError("the xGMI parameter server keeps parameters in GPU memory; use --ps_backend=tcp on CPUs")
        self.opt_name = optimizer.lower()
        self.cfg = _opt_config(optimizer, opt_kwargs)
        self.kind, self.nslot = self.cfg["kind"], self.cfg["nslot"]
        self.schedule: Schedule = lr if callable(lr) else constant(float(lr))
        self.optimizer = self  # the session asks ``engine.optimizer`` for current_lr / step_count
        self._steps = 0        # steps this worker has run
        self._base_step = 0    # global step at initialisation / restore
        self._stats_base: dict = {}
        self._closed = False
        self._initialized = False
        self._store = None
        self.shards: List[_Region] = []
        # The pull of a step runs in chunks on a side stream, first layers' parameters first (they sit at the END of
        # the flat buffers); a layer waits for the chunks that cover its parameters right before it reads them
        # (ops.functional.param_fence), so the forward starts as soon as the first chunk is in, and the rest of the
        # pull -- peer reads over xGMI -- runs beside it.
        self.overlap_pull = bool(overlap_pull)
        self._pull_stream = None
        self._pull_events: Dict[tuple, object] = {}     # (group index, chunk) -> event of this step's pull, not yet waited for
        self._chunks: List[List[tuple]] = []            # per group: [(lo, hi)] in forward order
        self._pull_waits: List = []
        dev = self.flat.device
        with torch.cuda.device(dev):
            for g in self.flat.groups:
                r = _Region(g.name, g.total, g.grad.dtype == torch.bfloat16, self.nown, self.nwrk, self.oidx, self.widx,
                            self.nslot, self.timeout, dev)
                r.group = g
                self.shards.append(r)
            if cluster is None:
                self._connect_through_process_group()
            else:
                self._connect_through_store()
            if self.overlap_pull:
                self._plan_chunks(max(1, int(pull_chunks)))
            if not (defer_init and self.is_chief):
                self.initialize()

    # ------------------------------------------------------------------ start-up
    def _connect_through_process_group(self) -> None:
        for sh in self.shards:
            mine = sh.handle()
            handles = [None] * self.nown
            if self.nown > 1:
                dist.all_gather_object(handles, mine, group=self.pg)
            else:
                handles = [mine]
            sh.connect(handles)
        if self.distributed and self.nwrk > 1:
            # the chief's initial values everywhere (variables are initialised once, by the chief)
            for g in self.flat.groups:
                dist.broadcast(g.master, 0, group=self.pg)
            for b in self.model.buffers():
                if b.is_floating_point():
                    dist.broadcast(b.data, 0, group=self.pg)

    def _connect_through_store(self) -> None:
        store = self._store = self.cluster.store()
        layout = json.dumps([{"name": sh.name, "total": sh.total, "grad_bf16": sh.group.grad.dtype == torch.bfloat16,
                              "nslot": self.nslot, "decay": decay_spec(sh.group, 0, sh.group.total) is True}
                             for sh in self.shards])
        # every worker derives the same layout from the same model; the ps tasks allocate from the first one set
        if store.add("xps/layout_set", 1) == 1:
            store.set("xps/layout", layout)
        else:
            have = store.get("xps/layout").decode()
            if json.loads(have) != json.loads(layout):
                raise RuntimeError("this worker's parameter layout differs from the one the ps tasks were started with")
        for gi, sh in enumerate(self.shards):
            sh.connect([bytes(store.get("xps/%d/handle/%d" % (gi, o))) for o in range(self.nown)])

    def initialize(self) -> None:
        """Variables are initialised exactly once: the chief (or, without ps tasks, every owner for its own shard)
        uploads the initial or restored values; everybody else waits for that.  Then the worker joins and pulls."""
        if self._initialized:
            return
        dev = self.flat.device
        with torch.cuda.device(dev):
            if self.cluster is None:
                self._upload_shards(slots=getattr(self, "_pending_slots", None), applied=self._base_step)
                torch.cuda.synchronize(dev)
                if self.distributed and self.nwrk > 1:
                    dist.barrier(group=self.pg)  # every shard is initialised and mapped before the first pull
            elif self.is_chief:
                if self._store.add("xps/init_claim", 1) == 1:   # (a restarted chief finds the job initialised)
                    self._upload_shards(slots=getattr(self, "_pending_slots", None), applied=self._base_step)
                    torch.cuda.synchronize(dev)
                    self._store.set("xps/ready", "1")
                else:
                    self._store.get("xps/ready")
            else:
                self._store.get("xps/ready")   # blocks until the chief has initialised the ps tasks
            self._pending_slots = None
            for sh in self.shards:
                hipk.check(_lib().dtfa_xps_join(sh.ctx, self._st()), "xps_join")
            self._initialized = True
            self.pull()
            torch.cuda.synchronize(dev)
            self._base_step = int(self.shards[0].report()[0])

    # ------------------------------------------------------------------ overlapped pull
    def _plan_chunks(self, k: int) -> None:
        from ..ops import functional as F

        self._pull_stream = torch.cuda.Stream(device=self.flat.device)
        seg_of = {}
        for gi, g in enumerate(self.flat.groups):
            step = max(64, (g.total // k + 63) // 64 * 64)
            bounds = [(lo, min(lo + step, g.total)) for lo in range(0, g.total, step)]
            bounds.reverse()   # parameters are laid out last layer first: the forward needs the tail first
            self._chunks.append(bounds)
            for sgm in g.segments:
                seg_of[id(sgm.param)] = (gi, sgm.offset, sgm.offset + sgm.numel)
        for m in self.model.modules():
            need = set()
            for p in m._parameters.values():
                if p is None or id(p) not in seg_of:
                    continue
                gi, a, z = seg_of[id(p)]
                need.update((gi, ci) for ci, (lo, hi) in enumerate(self._chunks[gi]) if lo < z and a < hi)
            if need:
                m._dtfa_pull_chunks = sorted(need)
        F.set_param_fence(self._param_fence)

    def _param_fence(self, modules) -> None:
        if not self._pull_events:
            return
        cur = torch.cuda.current_stream(self.flat.device)
        for m in modules:
            for key in getattr(m, "_dtfa_pull_chunks", ()):
                ev = self._pull_events.pop(key, None)
                if ev is not None:
                    self._wait_pull(cur, ev)

    def _wait_pull(self, cur, ev) -> None:
        if self.measure_comm:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(cur)
            cur.wait_event(ev)
            b.record(cur)
            self._pull_waits.append((a, b))
        else:
            cur.wait_event(ev)

    def _join_pull(self) -> None:
        """The current stream waits for whatever is left of this step's pull (parameters no layer asked for)."""
        if self._pull_events:
            cur = torch.cuda.current_stream(self.flat.device)
            for ev in self._pull_events.values():
                self._wait_pull(cur, ev)
            self._pull_events.clear()

    def _pull_chunked(self) -> None:
        L = _lib()
        cur = torch.cuda.current_stream(self.flat.device)
        side = self._pull_stream
        side.wait_stream(cur)     # behind the gate, and behind the previous step's last reads of the parameters
        with torch.cuda.stream(side):
            st = side.cuda_stream
            order = sorted(((ci, gi) for gi, ch in enumerate(self._chunks) for ci in range(len(ch))))
            for ci, gi in order:   # chunk 0 of every group first: the first layers' weights and batch-norm parameters
                sh, g = self.shards[gi], self.flat.groups[gi]
                lo, hi = self._chunks[gi][ci]
                is_bf16 = g.compute.dtype == torch.bfloat16
                dst32 = g.master if g.compute is g.master else None
                dst = g.compute if g.compute is not g.master else None
                hipk.check(L.dtfa_xps_pull_range(sh.ctx, hipk.ptr(dst), int(is_bf16), hipk.ptr(dst32), lo, hi, st), "xps_pull")
                self._pull_events[(gi, ci)] = side.record_event()

    # ------------------------------------------------------------------ helpers
    def _st(self) -> int:
        return torch.cuda.current_stream(self.flat.device).cuda_stream

    def _piece(self, t: torch.Tensor, sh: _Region, o: int) -> torch.Tensor:
        return t[o * sh.piece:(o + 1) * sh.piece]

    def _upload_shards(self, slots: Optional[Dict[str, torch.Tensor]], applied: int) -> None:
        """``FlatGroup.master`` (and full-layout optimizer slots) become the owners' state: without ps tasks every
        process uploads the shard it owns, with ps tasks the chief uploads all of them through the peer mappings."""
        L = _lib()
        owners = [self.oidx] if self.cluster is None else (list(range(self.nown)) if self.is_chief else [])
        for sh in self.shards:
            g = sh.group
            for o in owners:
                sl = []
                for k in self.cfg["slots"]:
                    full = None if slots is None else slots.get(k + "/" + g.name)
                    sl.append(None if full is None else self._piece(full.to(g.master.device, torch.float32), sh, o).contiguous())
                while len(sl) < 2:
                    sl.append(None)
                own = self._piece(g.master, sh, o).contiguous()
                hipk.check(L.dtfa_xps_set_shard(sh.ctx, o, own.data_ptr(), hipk.ptr(sl[0]), hipk.ptr(sl[1]), int(applied),
                                                self._st()), "xps_set_shard")
        self._base_step = int(applied)
        self._steps = 0

    def current_lr(self) -> float:
        return self.schedule(self.global_step_estimate)

    @property
    def global_step_estimate(self) -> int:
        """The global step the learning-rate schedule sees, without a device synchronisation.  Lock-step rounds: the
        host counts them (one optimizer step per round, exact).  Asynchronous applies: the device-side count of
        optimizer steps applied to shard 0, as of this worker's last pull -- it moves with what was really applied,
        whatever the other workers' pace and however many of them are left."""
        if self.apply == "aggregate":
            return self._base_step + self._steps
        return max(int(self.shards[0].report()[0]), self._base_step) if self.shards else self._base_step

    # ------------------------------------------------------------------ protocol
    def pull(self, into_master: bool = False) -> None:
        self._join_pull()
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
        if self.overlap_pull:
            self._join_pull()
            self._pull_chunked()
        else:
            self.pull()

    def _step_size(self) -> float:
        lr = self.schedule(self.global_step_estimate)
        if self.kind == 0 and self.apply == "aggregate":
            t = self.global_step_estimate + 1  # one optimizer step per round: the host knows its number
            return lr * math.sqrt(1.0 - self.cfg["b2"] ** t) / (1.0 - self.cfg["b1"] ** t)
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

    def _apply(self, sh: _Region, advance_clock: int, aggregate: bool) -> None:
        c = self.cfg
        wd = c["wd"] if decay_spec(sh.group, 0, sh.group.total) is True else 0.0
        host_lr_t = int(self.kind == 0 and aggregate)
        # regions of departed owner-workers are taken over by the lowest live one (only where workers own the shards)
        adopt = int(self.cluster is None and self.nown > 1)
        hipk.check(_lib().dtfa_xps_apply(sh.ctx, self.kind, c["b1"], c["b2"], c["eps"], c["mu"], wd, int(c["nesterov"]),
                                         host_lr_t, int(aggregate), advance_clock, adopt, 0, 0.0, sh.hyper.data_ptr(),
                                         self._st()), "xps_apply")

    def finish_step(self, prepared: bool = False) -> None:
        self.flat.collect_autograd_grads()
        from ..ops import functional as F

        F.join_wgrad_stream(self.flat.device)
        self._join_pull()
        if not prepared:
            self.prepare_step()
        L = _lib()
        for sh in self.shards:
            hipk.check(L.dtfa_xps_push(sh.ctx, sh.group.grad.data_ptr(), self._st()), "xps_push")
        for sh in self.shards:
            self._apply(sh, 1, self.apply == "aggregate")
        if not prepared:
            self.advance_step()

    def drain(self) -> None:
        """Owner: apply what has arrived without taking a worker step (e.g. before a checkpoint or at the end)."""
        if self.apply == "aggregate" or self.oidx < 0:
            return
        self.prepare_step()
        for sh in self.shards:
            self._apply(sh, 0, False)

    # ------------------------------------------------------------------ state
    def state(self) -> dict:
        """Counters of the first group's regions (synchronises the stream)."""
        return self.shards[0].state(self._st(), self.nown, self.nwrk)

    # -- communication accounting (same interface as DataParallelEngine; bench.py reports it) ----------------
    measure_comm = False

    def reset_comm_stats(self) -> None:
        self._stats_base = self.state()
        self._stats_steps0 = self._steps
        self._pull_waits = []

    def comm_stats(self) -> dict:
        """Protocol counters since :meth:`reset_comm_stats` (synchronises the stream): what was pushed, what was
        dropped or torn, which peers were written off, which regions this process took over."""
        st = self.state()
        base = self._stats_base or {}
        gbytes = sum(sh.group.grad.numel() * sh.group.grad.element_size() for sh in self.shards)
        pbytes = sum(sh.group.total * 4 for sh in self.shards)

        def since(k):
            return int(st[k]) - int(base.get(k, 0))

        extra = {"overlap_pull": self.overlap_pull}
        steps = self._steps - getattr(self, "_stats_steps0", 0)
        if self._pull_waits and steps > 0:
            # time the compute stream stood still for chunks of the pull that had not arrived when a layer needed them
            extra["exposed_pull_ms_per_step"] = round(sum(a.elapsed_time(b) for a, b in self._pull_waits) / steps, 4)
        return {**extra, "collective": False, "world": self.nwrk, "owners": self.nown, "mode": self.mode, "backend": self.backend,
                "owned_by": "ps tasks" if self.cluster is not None else "workers",
                "apply": self.apply, "staleness": self.staleness,
                "grad_dtype": str(self.shards[0].group.grad.dtype).replace("torch.", "") if self.shards else None,
                "groups": len(self.shards), "pushed_bytes_per_step": int(gbytes), "pulled_bytes_per_step": int(pbytes),
                "pushes": since("pushes"), "applied": since("global_step"), "status": int(st["status"]),
                "dropped_pushes": since("dropped_pushes"), "torn_pulls": since("torn_pulls"),
                "timeouts": since("timeouts"), "dead_mask": int(st["dead_mask"]), "done_mask": int(st["done_mask"]),
                "adopted_mask": int(st["adopted_mask"]), "applied_per_shard": st["applied_per_shard"],
                "clocks": st["clocks"]}

    @property
    def global_step(self) -> int:
        """Optimizer steps applied to the variables: the largest count over the shards (every push reaches every
        shard, so the counts differ only by what is in flight, and a departed chief or owner does not freeze it).
        Read from the words the kernels report; no device synchronisation."""
        if self.apply == "aggregate":
            return self._base_step + self._steps
        a0, _, own = self.shards[0].report()
        return max(int(a0), int(own), self._base_step)

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
        out = {}
        for sh in self.shards:
            g = sh.group
            for k, kind in enumerate(self.cfg["slots"]):
                full = torch.zeros(g.total, dtype=torch.float32, device=g.master.device)
                for o in range(self.nown):
                    hipk.check(L.dtfa_xps_read_slot(sh.ctx, o, k, full[o * sh.piece:].data_ptr(), self._st()), "xps_read_slot")
                out[kind + "/" + g.name] = full
        return out

    @property
    def fetched_global_step(self) -> Optional[int]:
        return int(self.state()["global_step"])

    @property
    def beta1_power(self):
        return self.cfg["b1"] ** (self.fetched_global_step + 1) if self.kind == 0 else None

    @property
    def beta2_power(self):
        return self.cfg["b2"] ** (self.fetched_global_step + 1) if self.kind == 0 else None

    def load_master(self) -> None:
        """After a restore wrote ``FlatGroup.master``: the owners' shards are re-published with the restored step
        count (without ps tasks every rank loads the checkpoint and uploads its own shard; with ps tasks the chief
        uploads all of them); slots follow in :meth:`load_full_slots`."""
        if not self._initialized:
            return   # (a resuming chief: the restored values are uploaded by initialize())
        self._upload_shards(slots=None, applied=self._base_step)
        self._resync()

    def load_full_slots(self, full: Dict[str, torch.Tensor]) -> None:
        if not self._initialized:
            self._pending_slots = {k: v.detach().clone() for k, v in full.items()}
            return
        self._upload_shards(slots=full, applied=self._base_step)
        self._resync()

    def _resync(self) -> None:
        torch.cuda.current_stream(self.flat.device).synchronize()
        if self.distributed and self.nwrk > 1:
            dist.barrier(group=self.pg)
        self.pull()

    def close(self, shutdown_ps: bool = False) -> None:
        if self._closed:
            return
        self._closed = True
        L = _lib()
        for sh in self.shards:
            hipk.check(L.dtfa_xps_leave(sh.ctx, self._st()), "xps_leave")
            if shutdown_ps and self.cluster is not None:
                hipk.check(L.dtfa_xps_shutdown(sh.ctx, self._st()), "xps_shutdown")
        torch.cuda.current_stream(self.flat.device).synchronize()
        # the regions stay allocated until the process exits: peers may still be reading them, and the process
        # that takes a departed owner's region over keeps applying it in place


class XgmiPSServer:
    """One ps task of the xGMI parameter server: owns shard ``ps_index`` of every flat group and applies the
    gradients that arrive in its mailboxes until the chief asks it to stop (the counterpart of ``server.join()``,
    recognition_flower/distributed/train_dataset.py:142-145).  ps task 0 also hosts the rendezvous store.  The
    layout (group sizes, gradient dtype) is taken from the store, where the first worker publishes it, so a ps task
    never builds the model."""

    def __init__(self, cluster: XpsCluster, ps_index: int, *, optimizer: str = "adam", lr=1e-3,
                 opt_kwargs: Optional[dict] = None, timeout: float = 30.0, device=None, idle_seconds: float = 0.02):
        self.cluster = cluster
        self.ps_index = int(ps_index)
        self.cfg = _opt_config(optimizer, opt_kwargs)
        self.schedule: Schedule = lr if callable(lr) else constant(float(lr))
        self.timeout = float(timeout)
        self.idle = float(idle_seconds)
        self.device = torch.device(device if device is not None else "cuda")
        self.store = cluster.store(is_master=(self.ps_index == 0))
        self.regions: List[_Region] = []
        self.applied = 0

    def _allocate(self) -> None:
        layout = json.loads(self.store.get("xps/layout").decode())
        n, w = self.cluster.num_ps, self.cluster.num_workers
        with torch.cuda.device(self.device):
            for gi, g in enumerate(layout):
                if int(g["nslot"]) != self.cfg["nslot"]:
                    raise RuntimeError("the workers run optimizer slots %d, this ps task %d" % (g["nslot"], self.cfg["nslot"]))
                r = _Region(g["name"], g["total"], bool(g["grad_bf16"]), n, w, self.ps_index, -1, self.cfg["nslot"],
                            self.timeout, self.device)
                r.no_decay_all = not g.get("decay", True)   # (a group of 1-D parameters: batch-norm scales, biases)
                self.regions.append(r)
                self.store.set("xps/%d/handle/%d" % (gi, self.ps_index), r.handle())
            for gi, r in enumerate(self.regions):
                r.connect([bytes(self.store.get("xps/%d/handle/%d" % (gi, o))) for o in range(n)])

    def serve_forever(self, poll_seconds: float = 0.25) -> int:
        """Returns the number of optimizer steps applied to this task's shard of the first group."""
        self._allocate()
        L = _lib()
        c = self.cfg
        n, w = self.cluster.num_ps, self.cluster.num_workers
        with torch.cuda.device(self.device):
            st = torch.cuda.current_stream(self.device).cuda_stream
            last_poll = 0.0
            while True:
                lr = self.schedule(self.applied)
                for r in self.regions:
                    hipk.check(hipk.lib().dtfa_set_hyper(r.hyper.data_ptr(), float(lr), 1.0, st), "set_hyper")
                    wd = 0.0 if r.no_decay_all else c["wd"]
                    # (the collect kernel sleeps up to ``idle`` seconds on the device until a gradient has arrived)
                    hipk.check(L.dtfa_xps_apply(r.ctx, c["kind"], c["b1"], c["b2"], c["eps"], c["mu"], wd, int(c["nesterov"]),
                                                0, 0, 0, 0, 1, self.idle, r.hyper.data_ptr(), st), "xps_apply")
                torch.cuda.current_stream(self.device).synchronize()
                self.applied = int(self.regions[0].report()[2])
                now = time.monotonic()
                if now - last_poll >= poll_seconds:
                    last_poll = now
                    s = self.regions[0].state(st, n, w)
                    joined, done = s["joined_mask"], s["done_mask"]
                    if s["shutdown"] or (joined and (joined & ~done) == 0 and s["arrived"] == s["consumed"]):
                        # asked to stop, or every worker that ever joined has left and nothing is waiting
                        break
        return self.applied
