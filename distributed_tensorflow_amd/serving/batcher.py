"""Dynamic request batching for the model server.

A GPU forward pass over 1 image and over 64 images cost nearly the same
(the batch-1 pass is launch- and latency-bound), so concurrent requests are
merged: callers :meth:`DynamicBatcher.submit` a list of items and get a future;
one worker thread -- the only thread that touches the device -- drains the
queue, runs ``fn`` once over up to ``max_batch`` items and hands every caller
its slice of the result.  A batch is closed when it is full or when
``timeout_ms`` have passed since its first item arrived, so an isolated request
waits at most that long.
"""
from __future__ import annotations

import collections
import threading
import time
from concurrent.futures import Future
from typing import Callable, Deque, Dict, List, Sequence, Tuple

import numpy as np


class QueueFull(RuntimeError):
    """More items are waiting than ``max_queue`` allows; the request is rejected instead of queued."""


class DynamicBatcher:
    def __init__(self, fn: Callable[[List], Dict[str, np.ndarray]], max_batch: int = 64, timeout_ms: float = 2.0,
                 max_queue: int = 4096, name: str = "batcher"):
        if max_batch < 1:
            raise ValueError("max_batch must be positive")
        self.fn = fn
        self.max_batch = int(max_batch)
        self.timeout = float(timeout_ms) / 1e3
        self.max_queue = int(max_queue)
        self._q: Deque[Tuple[List, Future, float]] = collections.deque()
        self._queued_items = 0
        self._cv = threading.Condition()
        self._stop = False
        # counters (read by the /metrics endpoint)
        self.batches = 0
        self.items = 0
        self.requests = 0
        self.rejected = 0
        self.split_batches = 0   # merged batches that failed and were re-run request by request
        self.busy_seconds = 0.0
        self.wait_seconds = 0.0
        self._thread = threading.Thread(target=self._loop, name=name, daemon=True)
        self._thread.start()

    # ------------------------------------------------------------------ client side
    def submit(self, items: Sequence) -> Future:
        """Queue ``items`` (one request); the future resolves to a dict of arrays with ``len(items)`` rows."""
        fut: Future = Future()
        items = list(items)
        if not items:
            fut.set_result(self.fn([]))
            return fut
        with self._cv:
            if self._stop:
                raise RuntimeError("batcher is closed")
            if self._queued_items + len(items) > self.max_queue:
                self.rejected += 1
                raise QueueFull("%d items queued" % self._queued_items)
            self._q.append((items, fut, time.monotonic()))
            self._queued_items += len(items)
            self.requests += 1
            self._cv.notify()
        return fut

    def __call__(self, items: Sequence) -> Dict[str, np.ndarray]:
        return self.submit(items).result()

    # ------------------------------------------------------------------ worker
    def _take_batch(self):
        """Block until a batch is due; returns [(items, future, t_enqueued)] or None at shutdown."""
        with self._cv:
            while not self._q and not self._stop:
                self._cv.wait()
            if not self._q:
                return None
            deadline = self._q[0][2] + self.timeout
            while self._queued_items < self.max_batch and not self._stop:
                left = deadline - time.monotonic()
                if left <= 0:
                    break
                self._cv.wait(left)
            batch, n = [], 0
            # whole requests only, unless a single request exceeds max_batch (it then runs alone, chunked below)
            while self._q and (n == 0 or n + len(self._q[0][0]) <= self.max_batch):
                entry = self._q.popleft()
                batch.append(entry)
                n += len(entry[0])
            self._queued_items -= n
            return batch

    def _loop(self):
        while True:
            batch = self._take_batch()
            if batch is None:
                return
            flat: List = []
            for items, _, _ in batch:
                flat.extend(items)
            t0 = time.monotonic()
            try:
                parts = [self.fn(flat[i:i + self.max_batch]) for i in range(0, len(flat), self.max_batch)]
                out = {k: np.concatenate([p[k] for p in parts], axis=0) for k in parts[0]}
            except BaseException as e:  # noqa: BLE001 -- every caller of this batch must be released
                if len(batch) == 1:
                    batch[0][1].set_exception(e)
                else:
                    # one bad item (an undecodable image) must not fail the other clients merged into the same
                    # device batch: run the requests one by one, only the offending ones get the exception
                    self.split_batches += 1
                    for items, fut, _ in batch:
                        try:
                            ps = [self.fn(items[i:i + self.max_batch]) for i in range(0, len(items), self.max_batch)]
                            fut.set_result({k: np.concatenate([q[k] for q in ps], axis=0) for k in ps[0]})
                        except BaseException as e1:  # noqa: BLE001
                            fut.set_exception(e1)
                continue
            t1 = time.monotonic()
            self.batches += len(parts)
            self.items += len(flat)
            self.busy_seconds += t1 - t0
            pos = 0
            for items, fut, t_in in batch:
                self.wait_seconds += t0 - t_in
                fut.set_result({k: v[pos:pos + len(items)] for k, v in out.items()})
                pos += len(items)

    def close(self):
        with self._cv:
            self._stop = True
            self._cv.notify_all()
        self._thread.join(timeout=10)
        with self._cv:
            while self._q:
                _, fut, _ = self._q.popleft()
                fut.set_exception(RuntimeError("batcher closed"))
            self._queued_items = 0

    def stats(self) -> Dict[str, float]:
        return {"requests": self.requests, "batches": self.batches, "items": self.items, "rejected": self.rejected,
                "mean_batch": self.items / self.batches if self.batches else 0.0,
                "busy_seconds": self.busy_seconds, "queue_wait_seconds": self.wait_seconds,
                "queued_items": self._queued_items}
