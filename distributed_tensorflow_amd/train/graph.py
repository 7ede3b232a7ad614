"""Whole-step HIP graph capture.

A training step of the reference's own configuration (Inception-v3 trunk
forward + two dense layers, batch 32; recognition_flower/single/flower_train.py:111,177-187)
is about 700 kernel launches of a few microseconds each: enqueueing them from
Python takes several times longer than executing them.  ``StepGraph`` records
one complete step -- forward, loss, backward, gradient collectives, fused
optimizer -- into a HIP graph once and replays it with a single launch per step.

What makes a step replayable:

* inputs live in fixed device buffers that are overwritten before each replay;
* everything that changes from step to step is *data*, not a kernel argument:
  the optimizer's step size and gradient scale are read from a device array
  (:meth:`optim.fused._RegionOptimizer.prepare`), dropout folds a device-resident
  step counter into its seed (:func:`ops.kernels.rng_step`);
* gradients are written into the persistent flat buffers by the backward kernels,
  so no ``.grad`` tensors are created or freed inside the graph;
* batch-norm running statistics and parameters are updated in place.

Loss / accuracy / logits are returned through fixed output tensors.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import os
import time

import torch
import torch.distributed as dist

from ..ops import kernels as K


# Collectives recorded inside a capture are never handed to the communicator's watchdog, so they never retire: their
# record ids are remembered here and not waited for by a later capture.
_captured_records: set = set()


def _collective_records():
    """(record id, retired) of every collective the communicator's flight recorder knows, or None when the recorder is
    off (TORCH_FR_BUFFER_SIZE = 0) or its dump cannot be read."""
    try:
        import pickle

        from torch._C._distributed_c10d import _dump_nccl_trace

        entries = pickle.loads(_dump_nccl_trace(includeCollectives=True, includeStackTraces=False, onlyActive=False)).get("entries")
    except Exception:  # noqa: BLE001 -- an unreadable dump is "no information", handled by the caller
        return None
    if entries is None:
        return None
    return [(e.get("record_id"), bool(e.get("retired"))) for e in entries]


def retire_eager_collectives(timeout: float = 20.0) -> None:
    """Return once the communicator's watchdog thread holds no collective of the eager steps any more.

    The watchdog retires finished collectives by querying their events from its own thread, every 100 ms.  A
    collective of the eager warm-up steps that it has not retired when a capture starts is queried DURING the capture,
    once the communicator's stream has joined it -- and HIP refuses the query of an event whose stream is capturing
    ("operation not permitted on an event last recorded in a capturing stream"), which the watchdog turns into an abort
    of the process.  Waiting for the device is not enough: a finished collective stays in the watchdog's list until
    its next look.  The communicator's flight recorder says when the watchdog HAS looked: it marks a collective
    ``retired`` at the moment the watchdog drops it.  So the capture starts only when every recorded collective
    that was issued outside a capture is retired; with the device idle that takes at most one watchdog period, and
    no timing assumption is involved (the loop ends on the recorder's state, the timeout only bounds a defect).
    The package turns the recorder on (``distributed_tensorflow_amd/__init__.py``); should it be off nevertheless,
    this raises instead of guessing."""
    t0 = time.monotonic()
    while True:
        recs = _collective_records()
        if recs is None:
            raise RuntimeError("the communicator's flight recorder is off (TORCH_FR_BUFFER_SIZE=0): a step graph "
                               "that holds collectives cannot be captured safely; unset the variable or capture before "
                               "the process group is created")
        pending = [i for i, retired in recs if not retired and i not in _captured_records]
        if not pending:
            return
        if time.monotonic() - t0 > timeout:
            raise RuntimeError("collectives %s of the eager steps were not retired within %.0f s" % (pending[:8], timeout))
        os.sched_yield()   # (the loop ends on the recorder's state; this only gives the core away between looks)


def note_captured_collectives() -> None:
    recs = _collective_records() or []
    _captured_records.update(i for i, retired in recs if not retired)


class StepGraph:
    def __init__(self, engine, loss_fn: Callable, images: torch.Tensor, labels: torch.Tensor):
        """``loss_fn(images, labels) -> (total_loss, *extras)``; extras are tensors kept as outputs."""
        if not images.is_cuda:
            raise ValueError("HIP graphs need device tensors")
        if not engine.graph_capturable:
            raise ValueError("engine mode %r cannot be captured" % engine.mode)
        self.engine = engine
        self.images = images.clone()
        self.labels = labels.clone()
        self.graph = torch.cuda.CUDAGraph()
        self.outputs: Optional[Tuple[torch.Tensor, ...]] = None
        self._capture(loss_fn)

    def _capture(self, loss_fn) -> None:
        eng = self.engine
        torch.cuda.synchronize()
        if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
            retire_eager_collectives()
        eng.prepare_step()
        # Other threads keep issuing HIP calls while the capture runs -- RCCL's watchdog polls events of earlier
        # work, the input pipeline's helper thread uploads the next batch on its copy stream -- and
        # "thread_local" keeps those calls from invalidating the capture.
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            eng.begin_step()
            out = loss_fn(self.images, self.labels)
            total = out[0]
            total.backward()
            eng.finish_step(prepared=True)
            self.outputs = tuple(o.detach() if isinstance(o, torch.Tensor) else o for o in out)
        if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
            note_captured_collectives()

    def compatible(self, images: torch.Tensor, labels: torch.Tensor) -> bool:
        return (images.shape == self.images.shape and images.dtype == self.images.dtype
                and labels.shape == self.labels.shape)

    def run(self, images: torch.Tensor, labels: torch.Tensor) -> Tuple[torch.Tensor, ...]:
        """One training step on ``images`` / ``labels``; returns the (static) output tensors."""
        if images.data_ptr() != self.images.data_ptr():
            self.images.copy_(images, non_blocking=True)
        if labels.data_ptr() != self.labels.data_ptr():
            self.labels.copy_(labels.to(self.labels.dtype), non_blocking=True)
        self.engine.prepare_step()
        K.rng_step(self.images.device).add_(1)
        self.graph.replay()
        self.engine.advance_step()
        return self.outputs
