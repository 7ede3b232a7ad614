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

import time

import torch
import torch.distributed as dist

from ..ops import kernels as K


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
            # The communicator's watchdog thread retires finished collectives by polling their events every 100 ms.
            # A collective of the eager steps before this capture that it has not looked at yet would be polled
            # DURING the capture, once the communicator's stream has joined it -- and HIP refuses a query of an
            # event whose stream is capturing ("operation not permitted on an event last recorded in a capturing
            # stream"), which the watchdog turns into an abort of the process (seen as an intermittent SIGABRT of
            # captured sync_dp / sharded_ps runs).  The device is idle here, so a few watchdog periods empty its list.
            time.sleep(0.35)
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
