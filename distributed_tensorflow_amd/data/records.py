"""TFRecord training input (``train_input_fn`` of the reference,
recognition_flower/distributed/train_dataset.py:68-94) on the native loader.

Records hold raw decoded pixels: ``image_raw`` (H*W*C uint8, OpenCV BGR
order), ``label``, ``height``, ``width``, ``channel``
(recognition_flower/distributed/prepare.py:45-51).  The C++ loader
(csrc/host/loader.cc) reads / parses / packs batches on worker threads into
pinned buffers; the batch is uploaded with one copy and resized on the GPU to
``[B, 299, 299, 8]`` (bilinear, TensorFlow legacy coordinates, like
``tf.image.resize_images(method=0)``).

On a GPU the upload and the resize of batch *k+1* are enqueued on a dedicated HIP stream while the
training step of batch *k* runs on the compute stream (the device-side half of the reference's
``prefetch(40)``): ``__next__`` hands out the batch prepared during the previous step, makes the compute
stream wait for its event, and immediately starts the next one.  The host never waits for the compute
stream; a pinned slot goes back to the loader threads once the copy stream has passed the batch's event.

Differences from the reference, on purpose: examples (not whole batches) are
shuffled, every rank reads a disjoint shard, pixels are converted to RGB and
normalised to [-1, 1] (``legacy_bgr_unnormalised=True`` restores the reference's
BGR [0, 255] input).
"""
from __future__ import annotations

import concurrent.futures as cf
import ctypes as C
from typing import Iterator, Optional, Tuple

import numpy as np
import torch

from ..io import hostlib
from .preprocess import NORM_INCEPTION, NORM_NONE, _ImageDesc, preprocess_batch, resize_on_device


# numpy mirror of preprocess._ImageDesc (same field order and C layout)
_DESC_DTYPE = np.dtype([("offset", np.uint64), ("h", np.int32), ("w", np.int32), ("c", np.int32),
                        ("crop_y0", np.float32), ("crop_x0", np.float32), ("crop_h", np.float32),
                        ("crop_w", np.float32), ("flip", np.int32), ("pad", np.int32)], align=True)
assert _DESC_DTYPE.itemsize == C.sizeof(_ImageDesc)


class RecordInput:
    def __init__(self, path: str, batch_size: int, device, *, image_size: int = 299, num_epochs: int = 1,
                 rank: int = 0, world: int = 1, threads: int = 4, prefetch: int = 4, shuffle: bool = True,
                 seed: int = 0, drop_remainder: bool = False, legacy_bgr_unnormalised: bool = False,
                 max_image_bytes: int = 1 << 20, channels: int = 8):
        self.device = torch.device(device)
        self.channels = channels   # channel count of the produced images (3 real + zero padding; device path)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.batch_size, self.image_size = batch_size, image_size
        self.legacy = legacy_bgr_unnormalised
        self._L = hostlib.lib()
        cap = batch_size * max_image_bytes
        pin = self.device.type == "cuda"
        self._bufs = [torch.empty(cap, dtype=torch.uint8).pin_memory() if pin else torch.empty(cap, dtype=torch.uint8)
                      for _ in range(prefetch)]
        # per-slot pinned staging for the small side tables (labels, image descriptors): every upload of a batch
        # is an asynchronous copy from page-locked memory
        self._lab_pin = [torch.empty(batch_size, dtype=torch.int32).pin_memory() for _ in range(prefetch)] if pin else []
        self._desc_pin = [torch.empty(batch_size * C.sizeof(_ImageDesc), dtype=torch.uint8).pin_memory()
                          for _ in range(prefetch)] if pin else []
        ptrs = (C.c_void_p * prefetch)(*[b.data_ptr() for b in self._bufs])
        self._h = self._L.dtfa_loader_open(path.encode(), rank, world, batch_size, threads, prefetch, ptrs, cap,
                                           seed, 1 if shuffle else 0, num_epochs, 1 if drop_remainder else 0)
        if not self._h:
            raise IOError("cannot open TFRecord file %s (missing, empty or corrupt)" % path)
        self.num_records = self._L.dtfa_loader_num_records(self._h)
        self.num_batches = self._L.dtfa_loader_num_batches(self._h)
        # device-side prefetch: (x, labels, event, slot) or an exception, prepared on ``_stream``
        self._stream = torch.cuda.Stream(device=self.device) if pin and prefetch >= 2 else None
        self._pending = None
        # the host half of preparing batch k+1 (waiting for the loader threads, side tables, enqueueing the
        # upload) runs on a helper thread, so that the trainer enqueues step k without waiting for it
        self._pool = cf.ThreadPoolExecutor(1, thread_name_prefix="dtfa-prefetch") if self._stream is not None else None

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        return self

    def __next__(self) -> Tuple[torch.Tensor, torch.Tensor]:
        if self._stream is None:
            return self._load()
        if self._pending is None:
            self._pending = self._pool.submit(self._prefetch)
        item, self._pending = self._pending.result(), None
        if isinstance(item, BaseException):
            raise item
        x, labels, event, slot = item
        event.synchronize()  # long done: it was recorded a whole training step ago
        self._L.dtfa_loader_release(self._h, slot)
        main = torch.cuda.current_stream(self.device)
        main.wait_event(event)
        # allocated on the copy stream, consumed (and later freed) on the compute stream
        x.record_stream(main)
        labels.record_stream(main)
        self._pending = self._pool.submit(self._prefetch)
        return x, labels

    def _prefetch(self):
        try:
            torch.cuda.set_device(self.device)
            with torch.cuda.stream(self._stream):
                x, labels, slot = self._load(release=False)
                return x, labels, self._stream.record_event(), slot
        except BaseException as e:  # noqa: BLE001 -- StopIteration included: surfaces when this batch is asked for
            return e

    def _load(self, release: bool = True):
        b = hostlib.Batch()
        rc = self._L.dtfa_loader_next(self._h, C.byref(b))
        if rc == 0:
            raise StopIteration  # the reference surfaces this as OutOfRangeError -> should_stop()
        if rc < 0:
            raise IOError("record loader failed with code %d (bad record, or image larger than max_image_bytes)" % rc)
        held = False
        try:
            n = b.count
            # the batch's side tables as arrays over the loader's memory (no per-image Python work)
            lab_np = np.ctypeslib.as_array(b.labels, (n,))
            hs, ws = np.ctypeslib.as_array(b.heights, (n,)), np.ctypeslib.as_array(b.widths, (n,))
            descs = np.zeros(n, dtype=_DESC_DTYPE)
            descs["offset"] = np.ctypeslib.as_array(b.offsets, (n,))
            descs["h"], descs["w"], descs["c"] = hs, ws, np.ctypeslib.as_array(b.channels, (n,))
            descs["crop_h"], descs["crop_w"] = hs, ws
            labels = torch.from_numpy(lab_np.astype(np.int32))
            norm = NORM_NONE if self.legacy else NORM_INCEPTION
            bgr_to_rgb = not self.legacy
            size = (self.image_size, self.image_size)
            host = self._bufs[b.slot][: int(b.bytes_used)]
            if self.device.type == "cuda":
                dev = host.to(self.device, non_blocking=True)
                nb = n * C.sizeof(_ImageDesc)
                self._desc_pin[b.slot][:nb].copy_(torch.from_numpy(descs.view(np.uint8)))
                dd = self._desc_pin[b.slot][:nb].to(self.device, non_blocking=True)
                x = resize_on_device(dev, dd, n, size, "bilinear_legacy", bgr_to_rgb, norm, channels=self.channels)
                self._lab_pin[b.slot][:n].copy_(labels)
                lab = self._lab_pin[b.slot][:n].to(self.device, non_blocking=True)
                if not release:
                    held = True
                    return x, lab, b.slot  # the caller releases the slot behind the stream's event
                torch.cuda.current_stream().synchronize()  # the slot is reused after release
                return x, lab
            arr = host.numpy()
            images = [arr[b.offsets[i]: b.offsets[i] + b.heights[i] * b.widths[i] * b.channels[i]]
                      .reshape(b.heights[i], b.widths[i], b.channels[i]) for i in range(n)]
            x = preprocess_batch(images, size, "cpu", "bilinear_legacy", None, bgr_to_rgb, norm)
            return x, labels
        finally:
            if not held:
                self._L.dtfa_loader_release(self._h, b.slot)

    def close(self) -> None:
        fut, self._pending = getattr(self, "_pending", None), None
        item = fut.result() if fut is not None else None
        if getattr(self, "_pool", None) is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        if self._h and isinstance(item, tuple):
            item[2].synchronize()
            self._L.dtfa_loader_release(self._h, item[3])
        if self._h:
            self._L.dtfa_loader_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class SyntheticInput:
    """Endless random batches with flower_photos-like shape (for throughput measurement)."""

    def __init__(self, batch_size: int, image_size: int, device, num_classes: int = 5, seed: int = 0,
                 dtype: Optional[torch.dtype] = None, channels: int = 8):
        device = torch.device(device)
        g = torch.Generator().manual_seed(seed)
        dtype = dtype or (torch.bfloat16 if device.type == "cuda" else torch.float32)
        x = torch.zeros(batch_size, image_size, image_size, channels)
        x[..., :3] = torch.rand(batch_size, image_size, image_size, 3, generator=g) * 2 - 1
        self.x = x.to(device=device, dtype=dtype)
        self.y = torch.randint(0, num_classes, (batch_size,), generator=g).to(device=device, dtype=torch.int32)

    def __iter__(self):
        return self

    def __next__(self):
        return self.x, self.y
