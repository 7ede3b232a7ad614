"""Device-side prefetch of training batches.

The reference's single-process trainer prepares every batch in line with the step
(``prepare_batch`` then ``sess.run``, recognition_flower/single/flower_train.py:180-185): decode, augmentation and
the host-to-device copy all sit on the critical path.  :class:`DevicePrefetcher` runs a batch producer one batch
ahead on a helper thread and a dedicated HIP stream: while step *k* executes on the compute stream, batch *k+1* is
read, entropy-decoded on the host threads, uploaded and transformed (IDCT, crop, resize) on the copy stream.
``next()`` hands the finished batch to the compute stream behind an event, so the host never waits for the device.
"""
from __future__ import annotations

import concurrent.futures as cf
from typing import Callable, Sequence

import torch


class DevicePrefetcher:
    """``produce()`` returns a tuple of device tensors (it may allocate, copy and launch kernels; everything it
    enqueues lands on the prefetcher's stream).  It is always called from the one helper thread, in order, so a
    producer drawing from seeded random generators stays deterministic."""

    def __init__(self, produce: Callable[[], Sequence[torch.Tensor]], device, depth: int = 1):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("DevicePrefetcher is for GPU devices; call the producer directly on the CPU")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._produce = produce
        self._stream = torch.cuda.Stream(device=self.device)
        self._pool = cf.ThreadPoolExecutor(1, thread_name_prefix="dtfa-batch")
        self._depth = max(1, depth)
        self._ahead = True
        self._queue = [self._pool.submit(self._run) for _ in range(self._depth)]

    def _run(self):
        torch.cuda.set_device(self.device)
        with torch.cuda.stream(self._stream):
            out = tuple(self._produce())
            return out, self._stream.record_event()

    def set_ahead(self, ahead: bool) -> None:
        """``False``: stop preparing batches ahead; the ones already queued are handed out in order, after that
        ``next()`` runs the producer in line on the caller's stream (same sequence of producer calls).  Worth it once
        a producer has nothing left to overlap -- e.g. every image comes from the device-resident cache, so a batch
        is a few kernel launches and the helper thread only adds hand-over cost.  ``True`` resumes."""
        self._ahead = bool(ahead)

    def next(self):
        """The oldest prepared batch (waits for the helper thread if it is still at work) and starts another one."""
        if not self._queue:
            if not self._ahead:
                return tuple(self._produce())
            self._queue = [self._pool.submit(self._run) for _ in range(self._depth)]
        tensors, event = self._queue.pop(0).result()
        if self._ahead:
            self._queue.append(self._pool.submit(self._run))
        main = torch.cuda.current_stream(self.device)
        main.wait_event(event)
        for t in tensors:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(main)  # allocated on the copy stream, used and freed on the compute stream
        return tensors

    __next__ = next

    def __iter__(self):
        return self

    def close(self) -> None:
        for f in self._queue:
            f.cancel()
        self._pool.shutdown(wait=True)
        self._queue = []
