"""Parameter-server runtime: ``ps`` / ``worker`` roles over TCP.

This is the between-graph-replicated asynchronous training of the reference
(recognition_flower/distributed/train_dataset.py:131-192) without TensorFlow's
gRPC runtime:

* a **ps** task owns a subset of the trainable variables (assigned round-robin
  in creation order, like ``tf.train.replica_device_setter``), their optimizer
  slots, and -- on ps 0 -- the global step.  It applies every gradient it
  receives immediately (``ApplyAdam`` runs on the ps, next to the variable).
* a **worker** task keeps a full model replica, and per step *pulls* the current
  variables, computes gradients on its own batch and *pushes* them.  Workers
  never talk to each other, may join late, finish early or die without
  affecting the rest (the property the reference gets from ``device_filters``,
  train_dataset.py:172-174).  Only trainable variables travel; the frozen trunk
  stays on the worker (the reference re-pulls all of it every step).
* initialisation happens exactly once: the chief worker uploads its initial
  values, everybody else blocks in ``wait_init`` (the reference lets every
  worker re-run the init op, resetting the ps each time a worker joins).
* **bounded staleness** (stale-synchronous parallel): every worker carries a
  clock (its number of pushed steps); with ``staleness = s >= 0`` a pull from a
  worker whose clock is more than ``s`` ahead of the slowest *live* worker is
  held back until the others catch up.  ``s < 0`` is the reference's unbounded
  asynchrony, ``s = 0`` is lock-step.  A worker that stays silent for
  ``worker_timeout`` seconds or disconnects stops counting, so nobody waits
  for the dead.

Wire format: 32-byte header ``<4sIiiqQ`` (magic ``DTPS``, op, worker, status,
clock/step, payload bytes) followed by the payload (fp32 little-endian blobs or
UTF-8 JSON).
"""
from __future__ import annotations

import ctypes as C
import json
import os
import queue
import socket
import struct
import threading
import time
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch


MAGIC = b"DTPS"
HEADER = struct.Struct("<4sIiiqQ")

OP_HELLO, OP_INIT, OP_WAIT_INIT, OP_PULL, OP_PUSH, OP_GET_STATE, OP_SET_STATE, OP_BYE, OP_SHUTDOWN, OP_STATUS = range(1, 11)
ST_OK, ST_NOT_INIT, ST_ERROR, ST_STOPPED = 0, 1, 2, 3


class PSStopped(ConnectionError):
    """The parameter server is shutting down (the chief finished the step budget)."""


def _recv_exact(sock: socket.socket, n: int, into=None):
    """``n`` bytes from the socket, into ``into`` (a writable byte buffer of at least ``n`` bytes, e.g. pinned
    staging memory) or a new bytearray."""
    buf = bytearray(n) if into is None or len(into) < n else into
    view = memoryview(buf)[:n]
    got = 0
    while got < n:
        k = sock.recv_into(view[got:], n - got)
        if k == 0:
            raise ConnectionError("peer closed the connection")
        got += k
    return buf


def _send_msg(sock, op, worker=0, status=0, clock=0, payload=b""):
    """``payload``: bytes or any C-contiguous buffer (numpy array, memoryview); sent without an intermediate copy."""
    view = memoryview(payload).cast("B") if not isinstance(payload, (bytes, bytearray)) else payload
    sock.sendall(HEADER.pack(MAGIC, op, worker, status, clock, len(view)))
    if len(view):
        sock.sendall(view)


def _recv_msg(sock, into=None):
    magic, op, worker, status, clock, n = HEADER.unpack(bytes(_recv_exact(sock, HEADER.size)))
    if magic != MAGIC:
        raise ConnectionError("bad magic")
    payload = _recv_exact(sock, n, into) if n else bytearray()
    if into is not None and payload is into:
        payload = memoryview(into)[:n]
    return op, worker, status, clock, payload


def _big_buffers(sock) -> None:
    # variable / gradient messages are megabytes: fewer, larger kernel copies per message
    for opt in (socket.SO_SNDBUF, socket.SO_RCVBUF):
        try:
            sock.setsockopt(socket.SOL_SOCKET, opt, 4 << 20)
        except OSError:
            pass


# ----------------------------------------------------------------------------
# server
# ----------------------------------------------------------------------------
class _NativeOptimizer:
    """What the rest of the runtime asks of a ps task's optimizer (slot tensors, step count, Adam's powers); the
    update itself runs in ``csrc/host/ps_core.cc``."""

    def __init__(self, server, name: str, region: str, n: int, opt_kwargs: dict):
        self._server = server
        self.name = name.lower()
        kw = dict(opt_kwargs or {})
        if self.name == "adam":
            self.kind = 0
            self.b1, self.b2, self.eps = float(kw.get("beta1", 0.9)), float(kw.get("beta2", 0.999)), float(kw.get("eps", 1e-8))
            self.mu, self.nesterov = 0.0, False
            self._slots = {"Adam/" + region: torch.zeros(n), "Adam_1/" + region: torch.zeros(n)}
        elif self.name in ("momentum", "sgd"):
            self.kind = 1
            self.b1 = self.b2 = self.eps = 0.0
            self.mu, self.nesterov = float(kw.get("momentum", 0.9)), bool(kw.get("nesterov", False))
            self._slots = {"Momentum/" + region: torch.zeros(n)}
        else:
            raise ValueError("unknown optimizer %r" % name)

    def slots(self) -> Dict[str, torch.Tensor]:
        return self._slots

    @property
    def step_count(self) -> int:
        return self._server.global_step

    @step_count.setter
    def step_count(self, v: int) -> None:
        self._server._set_step(int(v))

    @property
    def beta1_power(self):
        return self.b1 ** (self.step_count + 1) if self.kind == 0 else None

    @property
    def beta2_power(self):
        return self.b2 ** (self.step_count + 1) if self.kind == 0 else None


class ParameterServer:
    """One ps task.  ``serve_forever`` blocks like ``server.join()``
    (recognition_flower/distributed/train_dataset.py:142-145) until a SHUTDOWN
    request arrives.

    This class is the control plane: it accepts connections and answers the rare frames (hello, initialisation,
    checkpoint state, status, goodbye, shutdown).  The frames of every worker step -- PULL and PUSH -- are served
    by the native core (``csrc/host/ps_core.cc``), which also owns the clocks, the staleness bound and the global
    step: a connection thread spends its life inside ``dtfa_ps_serve`` (no GIL) and comes back to Python only for a
    control frame."""

    def __init__(self, host: str, port: int, ps_index: int = 0, staleness: int = -1, worker_timeout: float = 30.0,
                 debug: Optional[bool] = None, apply_threads: int = 4):
        # debug mode (also DTFA_DEBUG_STALENESS=1): the allowed races are asserted on version counters --
        # a pull is never admitted beyond the staleness bound, a worker's clock advances by exactly one per
        # applied push, and the number of applies equals the sum of the clock advances.  Violations are
        # recorded in ``violations`` and answered with an error status instead of being served.
        from ..io import hostlib

        self.debug = bool(int(os.environ.get("DTFA_DEBUG_STALENESS", "0"))) if debug is None else bool(debug)
        self.addr = (host, port)
        self.ps_index = ps_index
        self.staleness = staleness
        self.worker_timeout = worker_timeout
        self.apply_threads = int(apply_threads)
        self._L = _core_lib(hostlib.lib())
        self._core = self._L.dtfa_ps_new(int(staleness), float(worker_timeout), int(self.debug))
        self.lock = threading.Condition()
        self.initialized = False
        self.params: Optional[torch.Tensor] = None
        self.optimizer: Optional[_NativeOptimizer] = None
        self._stop = threading.Event()
        self._sock = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        self._sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self._sock.bind(self.addr)
        self._sock.listen(64)
        self._sock.settimeout(0.2)
        self._threads: List[threading.Thread] = []
        self._conns: List[socket.socket] = []

    # -- state kept by the native core -----------------------------------------------
    def _status(self):
        out = (C.c_longlong * 8)()
        clocks = (C.c_longlong * 128)()
        self._L.dtfa_ps_status(self._core, out, clocks, 64)
        return out, {int(clocks[2 * i]): int(clocks[2 * i + 1]) for i in range(int(out[4]))}

    @property
    def global_step(self) -> int:
        return int(self._status()[0][0])

    @property
    def apply_count(self) -> int:
        return int(self._status()[0][1])

    @property
    def max_lag_seen(self) -> int:
        """Largest (clock - min live clock) ever admitted to pull; checked by tests."""
        return int(self._status()[0][2])

    @property
    def clocks(self) -> Dict[int, int]:
        return self._status()[1]

    @property
    def violations(self) -> List[str]:
        n = int(self._status()[0][3])
        out = []
        buf = C.create_string_buffer(512)
        for i in range(n):
            if self._L.dtfa_ps_violation(self._core, i, buf, 512) > 0:
                out.append(buf.value.decode())
        return out

    def _set_step(self, step: int) -> None:
        self._L.dtfa_ps_set_step(self._core, int(step))

    def _load_slots(self, names, blob, n: int) -> None:
        """Optimizer slots from a state message: ``names[i]`` (a slot kind such as ``Adam`` / ``Adam_1`` /
        ``Momentum``, optionally followed by ``/<region>``) is stored in ``blob[(i+1)*n:(i+2)*n]``."""
        slots = self.optimizer.slots()
        by_kind = {k.rsplit("/", 1)[0]: v for k, v in slots.items()}
        for i, k in enumerate(names):
            dst = slots.get(k)
            if dst is None:
                dst = by_kind.get(k.rsplit("/", 1)[0] if "/" in k else k)
            if dst is not None and blob.size >= (i + 2) * n:
                dst.copy_(torch.from_numpy(blob[(i + 1) * n:(i + 2) * n].copy()))

    def _initialise(self, meta: dict, blob: np.ndarray) -> None:
        nslots = len(meta.get("slots", []))
        n_par = blob.size // (1 + nslots)
        self.params = torch.from_numpy(blob[:n_par].copy())
        opt = _NativeOptimizer(self, meta.get("optimizer", "adam"), "ps%d" % self.ps_index, n_par, meta.get("opt_kwargs", {}))
        self.optimizer = opt
        if nslots:  # a resumed run: the chief hands over the restored optimizer state too
            self._load_slots(meta["slots"], blob, n_par)
        decay = meta.get("lr_decay")   # [decay_steps, decay_rate]: the schedule runs on the ps, on its own step count
        sl = list(opt.slots().values())
        self._L.dtfa_ps_set_data(self._core, self.params.data_ptr(), sl[0].data_ptr(), sl[1].data_ptr() if len(sl) > 1 else None,
                                 n_par, opt.kind, float(meta.get("lr", 1e-3)), int(decay[0]) if decay else 0,
                                 float(decay[1]) if decay else 1.0, opt.b1, opt.b2, opt.eps, opt.mu, 0.0, int(opt.nesterov),
                                 int(meta.get("global_step", 0)), self.apply_threads)
        self.initialized = True

    # -- request handling --------------------------------------------------------
    def _handle(self, conn: socket.socket) -> None:
        worker = -1
        hdr = C.create_string_buffer(HEADER.size)
        try:
            conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            _big_buffers(conn)
            while True:
                # PULL / PUSH frames are served natively; this returns with the header of the first other frame
                rc = self._L.dtfa_ps_serve(self._core, conn.fileno(), hdr)
                if rc != 1:
                    return
                magic, op, worker, _, clock, n = HEADER.unpack(hdr.raw)
                payload = _recv_exact(conn, n) if n else bytearray()
                gs = self.global_step
                if self._stop.is_set() and op in (OP_WAIT_INIT, OP_INIT):
                    # shutting down: tell the worker instead of dropping the connection
                    _send_msg(conn, op, worker, ST_STOPPED, gs)
                    continue
                if op == OP_HELLO:
                    self._L.dtfa_ps_hello(self._core, int(worker), int(clock))
                    _send_msg(conn, op, worker, ST_OK if self.initialized else ST_NOT_INIT, gs)
                elif op == OP_INIT:
                    meta_len = struct.unpack("<I", bytes(payload[:4]))[0]
                    meta = json.loads(bytes(payload[4:4 + meta_len]).decode())
                    blob = np.frombuffer(payload, dtype=np.float32, offset=4 + meta_len)
                    with self.lock:
                        if not self.initialized:  # first initialiser wins; later ones are ignored
                            self._initialise(meta, blob)
                            self.lock.notify_all()
                    _send_msg(conn, op, worker, ST_OK, self.global_step)
                elif op == OP_WAIT_INIT:
                    with self.lock:
                        while not self.initialized and not self._stop.is_set():
                            self.lock.wait(timeout=0.1)
                    _send_msg(conn, op, worker, ST_OK if self.initialized else ST_STOPPED, self.global_step)
                elif op == OP_GET_STATE:
                    if not self.initialized:
                        _send_msg(conn, op, worker, ST_NOT_INIT, 0)
                        continue
                    self._L.dtfa_ps_lock(self._core)   # no apply runs while the variables and slots are copied
                    try:
                        slots = self.optimizer.slots()
                        names = sorted(slots)
                        st, _ = self._status()
                        meta = json.dumps({"slots": names, "global_step": int(st[0]), "apply_count": int(st[1]),
                                           "beta1_power": self.optimizer.beta1_power,
                                           "beta2_power": self.optimizer.beta2_power}).encode()
                        blob = b"".join([self.params.numpy().tobytes()] + [slots[k].numpy().tobytes() for k in names])
                    finally:
                        self._L.dtfa_ps_unlock(self._core)
                    _send_msg(conn, op, worker, ST_OK, int(st[0]), struct.pack("<I", len(meta)) + meta + blob)
                elif op == OP_SET_STATE:
                    meta_len = struct.unpack("<I", bytes(payload[:4]))[0]
                    meta = json.loads(bytes(payload[4:4 + meta_len]).decode())
                    blob = np.frombuffer(payload, dtype=np.float32, offset=4 + meta_len)
                    if not self.initialized:
                        _send_msg(conn, op, worker, ST_NOT_INIT, 0)
                        continue
                    self._L.dtfa_ps_lock(self._core)
                    try:
                        n = self.params.numel()
                        self.params.copy_(torch.from_numpy(blob[:n].copy()))
                        self._load_slots(meta.get("slots", []), blob, n)
                        if "global_step" in meta:
                            self._set_step(int(meta["global_step"]))
                    finally:
                        self._L.dtfa_ps_unlock(self._core)
                    _send_msg(conn, op, worker, ST_OK, self.global_step)
                elif op == OP_STATUS:
                    st, clocks = self._status()
                    meta = json.dumps({"initialized": self.initialized, "global_step": int(st[0]),
                                       "clocks": {str(k): v for k, v in clocks.items()},
                                       "max_lag_seen": int(st[2]), "staleness": self.staleness,
                                       "debug": self.debug, "violations": self.violations}).encode()
                    _send_msg(conn, op, worker, ST_OK, int(st[0]), meta)
                elif op == OP_BYE:
                    self._L.dtfa_ps_bye(self._core, int(worker))
                    _send_msg(conn, op, worker, ST_OK, gs)
                    return
                elif op == OP_SHUTDOWN:
                    _send_msg(conn, op, worker, ST_OK, gs)
                    self.stop()
                    return
                else:
                    _send_msg(conn, op, worker, ST_ERROR, 0)
        except (ConnectionError, OSError):
            pass
        finally:
            # a vanished worker must not hold the others back
            if worker >= 0:
                self._L.dtfa_ps_bye(self._core, int(worker))
            try:
                conn.close()
            except OSError:
                pass

    def serve_forever(self) -> None:
        try:
            while not self._stop.is_set():
                try:
                    conn, _ = self._sock.accept()
                except socket.timeout:
                    continue
                conn.settimeout(None)
                t = threading.Thread(target=self._handle, args=(conn,), daemon=True)
                t.start()
                self._threads.append(t)
                self._conns.append(conn)
        finally:
            self._sock.close()
            # grace period: let the remaining workers notice ST_STOPPED and say BYE, then
            # unblock whatever is left so that every handler thread exits before we return
            deadline = time.time() + 3.0
            for t in self._threads:
                t.join(timeout=max(0.0, deadline - time.time()))
            for c in self._conns:
                try:
                    c.shutdown(socket.SHUT_RDWR)
                except OSError:
                    pass
            for t in self._threads:
                t.join(timeout=1.0)

    def start_background(self) -> threading.Thread:
        t = threading.Thread(target=self.serve_forever, daemon=True)
        t.start()
        return t

    def stop(self) -> None:
        self._stop.set()
        self._L.dtfa_ps_stop(self._core)
        with self.lock:
            self.lock.notify_all()


_core_sigs_done = False


def _core_lib(L):
    """ctypes signatures of the native ps core (csrc/host/ps_core.cc)."""
    global _core_sigs_done
    if not _core_sigs_done:
        L.dtfa_ps_new.argtypes = [C.c_int, C.c_double, C.c_int]
        L.dtfa_ps_new.restype = C.c_void_p
        L.dtfa_ps_set_data.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_double,
                                       C.c_longlong, C.c_double, C.c_double, C.c_double, C.c_float, C.c_float, C.c_float,
                                       C.c_int, C.c_longlong, C.c_int]
        L.dtfa_ps_set_data.restype = None
        for name, args in (("dtfa_ps_set_step", [C.c_void_p, C.c_longlong]), ("dtfa_ps_hello", [C.c_void_p, C.c_int, C.c_longlong]),
                           ("dtfa_ps_bye", [C.c_void_p, C.c_int]), ("dtfa_ps_stop", [C.c_void_p]), ("dtfa_ps_lock", [C.c_void_p]),
                           ("dtfa_ps_unlock", [C.c_void_p]), ("dtfa_ps_free", [C.c_void_p]),
                           ("dtfa_ps_status", [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.c_int])):
            getattr(L, name).argtypes = args
            getattr(L, name).restype = None
        L.dtfa_ps_violation.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int]
        L.dtfa_ps_violation.restype = C.c_int
        L.dtfa_ps_serve.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
        L.dtfa_ps_serve.restype = C.c_int
        _core_sigs_done = True
    return L


# ----------------------------------------------------------------------------
# client
# ----------------------------------------------------------------------------
class _Conn:
    def __init__(self, host: str, port: int, worker: int, connect_timeout: float):
        deadline = time.time() + connect_timeout
        last = None
        while True:
            try:
                self.sock = socket.create_connection((host, port), timeout=5.0)
                break
            except OSError as e:  # the ps may not be up yet
                last = e
                if time.time() > deadline:
                    raise ConnectionError("cannot reach ps at %s:%d: %s" % (host, port, last))
                time.sleep(0.1)
        self.sock.settimeout(None)
        self.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        _big_buffers(self.sock)
        self.worker = worker

    def call(self, op, clock=0, payload=b"", into=None):
        """``into``: byte buffer that receives the reply's payload when it fits (no allocation, no copy)."""
        _send_msg(self.sock, op, self.worker, 0, clock, payload)
        rop, _, status, rclock, data = _recv_msg(self.sock, into)
        if rop != op:
            raise ConnectionError("protocol error: reply %d to request %d" % (rop, op))
        return status, rclock, data

    def close(self):
        try:
            self.sock.close()
        except OSError:
            pass


class PSWorkerEngine:
    """Worker-side engine with the ``begin_step`` / ``finish_step`` interface of
    :class:`DataParallelEngine`, talking to one or more ps tasks."""

    def __init__(self, model, ps_addresses: List[Tuple[str, int]], worker_index: int, is_chief: bool, *,
                 optimizer: str = "adam", lr: float = 1e-3, opt_kwargs: Optional[dict] = None,
                 connect_timeout: float = 60.0, initial_global_step: int = 0, defer_init: bool = False,
                 lr_decay: Optional[Tuple[int, float]] = None, pipeline: bool = False):
        """``defer_init`` (chief only): connect, but leave the ps tasks uninitialised until :meth:`initialize`,
        so that a checkpoint can be restored into this engine first and the ps tasks start from the restored
        variables, optimizer slots and global step -- non-chief workers blocked in ``wait_init`` never see the
        un-restored values.  ``lr_decay`` = (decay_steps, decay_rate): exponential decay of ``lr`` over the global
        step, evaluated on the ps tasks (where the optimizer runs); the chief hands it over at initialisation.

        ``pipeline``: the network part of a step -- push of this step's gradient, then the pull for a later step --
        runs on a helper thread while the device computes the next step.  A step then starts from the variables of
        the newest pull that has COMPLETED, which contain every push up to the step before last at least: one step of
        staleness on the worker's own gradient, which asynchronous training accepts by definition (the reference
        reads variables while other workers' ApplyAdam ops are running).  At most one push / pull pair is in flight,
        so the worker runs at max(device time, network time) per step instead of their sum."""
        from .flat import FlatParams

        self.model = model
        self.worker_index = worker_index
        self.is_chief = is_chief
        self.flat = FlatParams(model, bucket_mb=1 << 20)  # one bucket per group: no collectives here
        self.lr = lr
        self.lr_decay = (int(lr_decay[0]), float(lr_decay[1])) if lr_decay else None
        self.mode = "async_ps"
        self.clock = 0
        self._prepared = False
        self.global_step = initial_global_step
        self.optimizer = self  # exposes current_lr() to the session
        self._optimizer_name, self._opt_kwargs = optimizer, dict(opt_kwargs or {})
        # round-robin variable placement over ps tasks, in creation order
        segs = []
        for g in self.flat.groups:
            for s in g.segments:
                segs.append((g, s))
        segs.reverse()  # FlatParams stores reverse registration order
        self.placement: List[List[Tuple[object, object]]] = [[] for _ in ps_addresses]
        for i, gs in enumerate(segs):
            self.placement[i % len(ps_addresses)].append(gs)
        # one host staging buffer per ps task, page-locked when the model lives on a GPU: gradients are gathered
        # into it with asynchronous device-to-host copies and sent from it, variables are received into it and
        # uploaded from it -- no intermediate byte strings
        pin = self.flat.device.type == "cuda"
        self._stage = [torch.empty(sum(s.numel for _, s in pl), dtype=torch.float32, pin_memory=pin)
                       for pl in self.placement]
        self._stage_np = [t.numpy() for t in self._stage]
        self.pipeline = bool(pipeline)
        if self.pipeline:
            # pulls land in two alternating buffers per ps task (one may still be uploading while the next arrives);
            # ``_stage`` keeps carrying the gradients
            self._pstage = [[torch.empty_like(t, pin_memory=pin) for t in self._stage] for _ in range(2)]
            self._pstage_np = [[t.numpy() for t in row] for row in self._pstage]
            self._job_done = threading.Event()
            self._job_done.set()
            self._job_error: Optional[BaseException] = None
            self._job_seq = 0            # jobs submitted
            self._pulled_seq = 0         # newest job whose pull has completed
            self._uploaded_seq = 0       # newest pull uploaded to the device
            self._jobs: "queue.Queue" = queue.Queue()
            self._io = threading.Thread(target=self._io_loop, daemon=True)
            self._io.start()
        self.conns = [_Conn(h, p, worker_index, connect_timeout) for h, p in ps_addresses]
        for c in self.conns:
            c.call(OP_HELLO, 0)
        self.initialized = False
        self._pending_slots: Dict[str, torch.Tensor] = {}
        self._cached_slots = None
        self._cached_info: dict = {}
        if not (defer_init and is_chief):
            self.initialize()

    def _slot_kinds(self) -> List[str]:
        return ["Adam", "Adam_1"] if self._optimizer_name.lower() == "adam" else ["Momentum"]

    def _pack_slot(self, ps: int, full: torch.Tensor) -> np.ndarray:
        """The part of a full-layout slot tensor (keyed by flat group) that lives on ps task ``ps``."""
        parts = [full[g.name][s.offset:s.offset + s.numel].detach().float().cpu().numpy() for g, s in self.placement[ps]]
        return np.concatenate(parts) if parts else np.zeros(0, np.float32)

    def _state_payload(self, ps: int, slots: Dict[str, torch.Tensor], extra_meta: dict) -> bytes:
        """``<meta><params><slot 0><slot 1>..`` for ps task ``ps``; ``slots`` is keyed ``<kind>/<group>``."""
        kinds = [k for k in self._slot_kinds() if all((k + "/" + g.name) in slots for g in self.flat.groups)]
        meta = dict(extra_meta, slots=kinds, global_step=int(self.global_step))
        blobs = [self._pack(ps, "master").tobytes()]
        for k in kinds:
            blobs.append(self._pack_slot(ps, {g.name: slots[k + "/" + g.name] for g in self.flat.groups}).tobytes())
        m = json.dumps(meta).encode()
        return struct.pack("<I", len(m)) + m + b"".join(blobs)

    def initialize(self) -> None:
        """Chief: upload the initial (or restored) variables, optimizer slots and global step -- exactly once per
        run; everybody else waits until that has happened.  Then the first pull."""
        if self.initialized:
            return
        if self.is_chief:
            for i, c in enumerate(self.conns):
                payload = self._state_payload(i, self._pending_slots, {"optimizer": self._optimizer_name, "lr": self.lr,
                                                                       "opt_kwargs": self._opt_kwargs,
                                                                       "lr_decay": list(self.lr_decay) if self.lr_decay else None})
                c.call(OP_INIT, 0, payload)
            self._pending_slots = {}
        else:
            for c in self.conns:
                st, _, _ = c.call(OP_WAIT_INIT, 0)
                if st != ST_OK:
                    raise ConnectionError("parameter server stopped before initialisation")
        self.initialized = True
        self.pull()

    def current_lr(self) -> float:
        if self.lr_decay:
            return self.lr * self.lr_decay[1] ** (self.global_step / self.lr_decay[0])
        return self.lr

    # -- (un)packing -------------------------------------------------------------
    def _pack(self, ps: int, what: str):
        """The segments placed on ps task ``ps`` gathered into its staging buffer; returns that buffer (numpy)."""
        stage, pos = self._stage[ps], 0
        for g, s in self.placement[ps]:
            src = g.master if what == "master" else g.grad
            stage[pos:pos + s.numel].copy_(src[s.offset:s.offset + s.numel].detach(), non_blocking=True)
            pos += s.numel
        if stage.is_pinned():
            torch.cuda.current_stream(self.flat.device).synchronize()
        return self._stage_np[ps]

    def _unpack_params(self, ps: int, data) -> None:
        """``data``: the reply payload -- a view of this ps task's staging buffer when it was received in place."""
        n = len(data) // 4
        if n != self._stage[ps].numel():
            raise ConnectionError("ps %d returned %d values, expected %d" % (ps, n, self._stage[ps].numel()))
        stage = self._stage[ps]
        if not isinstance(data, memoryview):
            stage.copy_(torch.from_numpy(np.frombuffer(data, dtype=np.float32)))
        pos = 0
        for g, s in self.placement[ps]:
            g.master[s.offset:s.offset + s.numel].copy_(stage[pos:pos + s.numel], non_blocking=True)
            pos += s.numel

    # -- protocol ---------------------------------------------------------------
    def pull(self) -> None:
        self._drain_pipeline()
        if self._stage[0].is_pinned():
            # uploads of an earlier pull may still be reading the staging buffers
            torch.cuda.current_stream(self.flat.device).synchronize()
        for i, c in enumerate(self.conns):
            st, step, data = c.call(OP_PULL, self.clock, into=memoryview(self._stage_np[i]).cast("B"))
            if st == ST_STOPPED:
                raise PSStopped("parameter server is shutting down")
            if st != ST_OK:
                raise ConnectionError("pull failed with status %d" % st)
            self._unpack_params(i, data)
            if i == 0:
                self.global_step = int(step)
        for g in self.flat.groups:
            g.sync_compute_from_master()

    def begin_step(self) -> None:
        self.flat.begin_step()
        if self._prepared:  # the pull was done by prepare_step (step-graph protocol)
            self._prepared = False
            return
        self._refresh()

    def _refresh(self) -> None:
        if self.pipeline:
            self._take_newest_pull()
        else:
            self.pull()

    # Step-graph protocol (train/graph.py): the device part of a worker step -- forward, backward, gradients
    # into the flat buffer -- is captured once and replayed; the network parts stay on the host around it:
    # ``prepare_step`` pulls the variables before the replay, ``advance_step`` pushes the gradients after it.
    @property
    def graph_capturable(self) -> bool:
        return self.flat.device.type == "cuda"

    def prepare_step(self) -> None:
        self._refresh()
        self._prepared = True

    def advance_step(self) -> int:
        self._prepared = False
        return self._submit() if self.pipeline else self._push()

    def finish_step(self, prepared: bool = False) -> int:
        self.flat.collect_autograd_grads()
        if prepared:  # inside a graph capture: the push follows the replay (advance_step)
            return self.global_step
        return self._submit() if self.pipeline else self._push()

    def _push(self) -> int:
        for i, c in enumerate(self.conns):
            st, step, _ = c.call(OP_PUSH, self.clock, self._pack(i, "grad"))
            if st == ST_STOPPED:
                raise PSStopped("parameter server is shutting down")
            if st != ST_OK:
                raise ConnectionError("push failed with status %d" % st)
            if i == 0:
                self.global_step = int(step)
        self.clock += 1
        return self.global_step

    # -- pipelined network part ---------------------------------------------------
    def _wait_job(self) -> None:
        self._job_done.wait()
        if self._job_error is not None:
            e, self._job_error = self._job_error, None
            raise e

    def _submit(self) -> int:
        """Hand this step's gradient to the helper thread: device-to-host copies are enqueued on the current stream,
        the thread waits for them, pushes, and pulls.  Waits first for the previous pair (one in flight)."""
        self._wait_job()
        for i in range(len(self.conns)):
            stage, pos = self._stage[i], 0
            for g, s in self.placement[i]:
                stage[pos:pos + s.numel].copy_(g.grad[s.offset:s.offset + s.numel].detach(), non_blocking=True)
                pos += s.numel
        ev = None
        if self.flat.device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.flat.device))
        self._job_seq += 1
        self._job_done.clear()
        # the push in flight counts: the step budget is compared with what the ps tasks WILL hold when it lands (the
        # helper thread replaces this with their answer, which other workers' pushes can only make larger)
        self.global_step += 1
        self._jobs.put((self._job_seq, ev))
        return self.global_step

    def _io_loop(self) -> None:
        while True:
            job = self._jobs.get()
            if job is None:
                return
            seq, ev = job
            try:
                if ev is not None:
                    ev.synchronize()
                for i, c in enumerate(self.conns):
                    st, step, _ = c.call(OP_PUSH, self.clock, self._stage_np[i])
                    if st == ST_STOPPED:
                        raise PSStopped("parameter server is shutting down")
                    if st != ST_OK:
                        raise ConnectionError("push failed with status %d" % st)
                    if i == 0:
                        self.global_step = int(step)
                self.clock += 1
                slot = seq & 1
                for i, c in enumerate(self.conns):
                    st, step, data = c.call(OP_PULL, self.clock, into=memoryview(self._pstage_np[slot][i]).cast("B"))
                    if st == ST_STOPPED:
                        raise PSStopped("parameter server is shutting down")
                    if st != ST_OK or len(data) != self._pstage[slot][i].numel() * 4:
                        raise ConnectionError("pull failed with status %d" % st)
                    if i == 0:
                        self.global_step = int(step)
                self._pulled_seq = seq
            except BaseException as e:  # noqa: BLE001 -- handed to the training thread at its next wait
                self._job_error = e
            finally:
                self._job_done.set()

    def _take_newest_pull(self) -> None:
        """Upload the newest completed pull (if it is newer than what the device holds).  Before the first job there
        is nothing in flight and nothing pulled: a plain pull."""
        if self._job_seq == 0:
            self.pull()
            return
        if self._job_error is not None:
            self._wait_job()
        seq = self._pulled_seq
        if seq <= self._uploaded_seq:
            if seq < self._job_seq - 1 or self._uploaded_seq == 0:
                # (not even the pair before last has come back: wait for it rather than run further ahead)
                self._wait_job()
                seq = self._pulled_seq
            if seq <= self._uploaded_seq:
                return
        slot = seq & 1
        for i in range(len(self.conns)):
            stage, pos = self._pstage[slot][i], 0
            for g, s in self.placement[i]:
                g.master[s.offset:s.offset + s.numel].copy_(stage[pos:pos + s.numel], non_blocking=True)
                pos += s.numel
        for g in self.flat.groups:
            g.sync_compute_from_master()
        self._uploaded_seq = seq

    def _drain_pipeline(self) -> None:
        if self.pipeline and self._job_seq:
            self._wait_job()

    def status(self) -> dict:
        self._drain_pipeline()
        _, _, data = self.conns[0].call(OP_STATUS, self.clock)
        return json.loads(bytes(data).decode())

    # -- checkpoint support (used by io.saver.Saver) --------------------------------
    def _fetch_state(self):
        """Full-layout fp32 masters and optimizer slots gathered from every ps."""
        self._drain_pipeline()   # (the connections carry one conversation at a time)
        masters = {g.name: g.master.clone() for g in self.flat.groups}
        slots: Dict[str, torch.Tensor] = {}
        info = {}
        for i, c in enumerate(self.conns):
            st, step, payload = c.call(OP_GET_STATE, self.clock)
            if st != ST_OK:
                raise ConnectionError("get_state failed")
            meta_len = struct.unpack("<I", bytes(payload[:4]))[0]
            meta = json.loads(bytes(payload[4:4 + meta_len]).decode())
            blob = np.frombuffer(payload, dtype=np.float32, offset=4 + meta_len)
            n = sum(s.numel for _, s in self.placement[i])
            if i == 0:
                info = meta
            for k, sname in enumerate([None] + meta["slots"]):
                arr = blob[k * n:(k + 1) * n]
                pos = 0
                for g, s in self.placement[i]:
                    if sname is None:
                        dst = masters[g.name]
                    else:
                        key = sname.rsplit("/", 1)[0] + "/" + g.name
                        if key not in slots:
                            slots[key] = torch.zeros_like(g.master)
                        dst = slots[key]
                    dst[s.offset:s.offset + s.numel].copy_(torch.from_numpy(arr[pos:pos + s.numel].copy()))
                    pos += s.numel
        return masters, slots, info

    def full_master(self):
        if not self.initialized:  # deferred initialisation: nothing on the ps yet, the local values are the state
            self._cached_slots = {k + "/" + g.name: torch.zeros_like(g.master) for k in self._slot_kinds()
                                  for g in self.flat.groups}
            self._cached_slots.update(self._pending_slots)
            self._cached_info = {}
            return {g.name: g.master for g in self.flat.groups}
        masters, slots, info = self._fetch_state()
        self._cached_slots, self._cached_info = slots, info
        for g in self.flat.groups:
            g.master.copy_(masters[g.name])
        return {g.name: g.master for g in self.flat.groups}

    def full_slots(self):
        if getattr(self, "_cached_slots", None) is None:
            self.full_master()
        s = self._cached_slots
        self._cached_slots = None
        return s

    @property
    def beta1_power(self):
        return (getattr(self, "_cached_info", {}) or {}).get("beta1_power")

    @property
    def beta2_power(self):
        return (getattr(self, "_cached_info", {}) or {}).get("beta2_power")

    @property
    def fetched_global_step(self) -> Optional[int]:
        """Global step of the state last fetched from the ps tasks (what a checkpoint of it must be labelled
        with: other workers keep stepping between this worker's update and the fetch)."""
        gs = (self._cached_info or {}).get("global_step")
        return int(gs) if gs is not None else None

    def load_master(self) -> None:
        """Push restored master values (``FlatGroup.master``) and the restored global step to the ps tasks
        (after initialisation; before it the values simply wait for :meth:`initialize`)."""
        if self.initialized:
            self._drain_pipeline()
            for i, c in enumerate(self.conns):
                c.call(OP_SET_STATE, self.clock, self._state_payload(i, {}, {}))
        for g in self.flat.groups:
            g.sync_compute_from_master()

    def load_full_slots(self, full) -> None:
        """Restored optimizer slots (full layout, keyed ``<kind>/<group>``) to their ps tasks, together with the
        variables and the global step (Adam's ``beta*_power`` follow from the step count on the ps)."""
        if not self.initialized:
            self._pending_slots = {k: v.detach().clone() for k, v in full.items()}
            return
        self._drain_pipeline()
        for i, c in enumerate(self.conns):
            c.call(OP_SET_STATE, self.clock, self._state_payload(i, full, {}))

    @property
    def step_count(self):
        return self.global_step

    @step_count.setter
    def step_count(self, v):
        self.global_step = int(v)

    def close(self, shutdown_ps: bool = False) -> None:
        if self.pipeline:
            try:
                self._drain_pipeline()
            except (ConnectionError, OSError):
                pass
            self._jobs.put(None)
        for c in self.conns:
            try:
                c.call(OP_SHUTDOWN if shutdown_ps else OP_BYE, self.clock)
            except (ConnectionError, OSError):
                pass
            c.close()
