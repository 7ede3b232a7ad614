"""JPEG decoding split between host and GPU.

Host (C++, ``csrc/host/jpeg_huffman.cc``): marker parsing + Huffman entropy
decode into quantised DCT coefficients.  GPU (``csrc/hip/jpeg.hip``):
dequantisation, 8x8 inverse DCT, chroma upsampling, YCbCr->RGB.  This is the
on-device replacement for the ``tf.image.decode_jpeg`` op the reference's
serving graph runs (recognition_flower/single/flower_train.py:41-47).

Only baseline / extended-sequential Huffman JPEGs are decoded natively;
anything else (progressive, arithmetic, CMYK) goes through PIL.  Chroma
subsampled 2:1 in either direction (4:2:2, 4:2:0, 4:4:0) is upsampled with
libjpeg's "fancy" triangle filters, other ratios by replication; decoded
images agree with libjpeg to within IDCT / colour-conversion rounding.

Decoded images can stay on the device (:class:`DecodedImageCache`): a flower
dataset is a few GB of RGB bytes against 288 GB of HBM, so from the second
epoch on a batch costs no host decode and no upload at all, only the
augmentation kernels.
"""
from __future__ import annotations

import ctypes as C
import os
import io
from concurrent.futures import ThreadPoolExecutor
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from ..io import hostlib


class _JpegDesc(C.Structure):
    _fields_ = [("coef_offset", C.c_ulonglong * 3), ("plane_offset", C.c_ulonglong * 3), ("rgb_offset", C.c_ulonglong),
                ("width", C.c_int), ("height", C.c_int), ("ncomp", C.c_int),
                ("hs", C.c_int * 3), ("vs", C.c_int * 3), ("blocks_w", C.c_int * 3), ("blocks_h", C.c_int * 3),
                ("hmax", C.c_int), ("vmax", C.c_int), ("qt", (C.c_uint16 * 64) * 3)]


def parse_header(data: bytes) -> Optional[hostlib.JpegInfo]:
    info = hostlib.JpegInfo()
    rc = hostlib.lib().dtfa_jpeg_parse_header(data, len(data), C.byref(info))
    return info if rc == 0 else None


def entropy_decode(data: bytes) -> Optional[Tuple[hostlib.JpegInfo, np.ndarray]]:
    """Quantised DCT coefficients (int16, block-major) or None when the file is not natively supported."""
    L = hostlib.lib()
    info = hostlib.JpegInfo()
    if L.dtfa_jpeg_parse_header(data, len(data), C.byref(info)) != 0 or info.ncomp not in (1, 3):
        return None
    coefs = np.empty(info.total_coefs, dtype=np.int16)
    rc = L.dtfa_jpeg_decode_coefs(data, len(data), C.byref(info), coefs.ctypes.data, coefs.size)
    if rc != 0:
        return None
    return info, coefs


def decode_pil(data: bytes) -> np.ndarray:
    from PIL import Image

    with Image.open(io.BytesIO(data)) as im:
        return np.asarray(im.convert("RGB"))


def decode_reference(data: bytes) -> np.ndarray:
    """Host-only decode (PIL / libjpeg): uint8 RGB [H, W, 3]."""
    return decode_pil(data)


def reconstruct_reference(info, coefs: np.ndarray) -> np.ndarray:
    """numpy version of the GPU half (dequantise, float IDCT, chroma upsampling, YCbCr->RGB);
    the oracle for ``csrc/hip/jpeg.hip`` and a check of the entropy decoder against libjpeg."""
    x = np.arange(8)
    cosm = np.cos((2 * x[:, None] + 1) * x[None, :] * np.pi / 16) * np.where(x[None, :] == 0, np.sqrt(0.125), 0.5)
    planes = []
    for c in range(info.ncomp):
        bw, bh = info.blocks_w[c], info.blocks_h[c]
        blk = coefs[info.coef_offset[c]: info.coef_offset[c] + bw * bh * 64].astype(np.float32).reshape(bh, bw, 8, 8)
        blk = blk * np.asarray(info.qt[c], dtype=np.float32).reshape(8, 8)
        pix = np.einsum("yv,hwvu,xu->hwyx", cosm.astype(np.float32), blk, cosm.astype(np.float32))
        pix = np.clip(np.rint(pix) + 128, 0, 255)
        planes.append(pix.transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8))
    H, W = info.height, info.width
    ys, xs = np.arange(H), np.arange(W)
    comps = []
    for c in range(info.ncomp):
        hs, vs = info.hs[c], info.vs[c]
        hr, vr = info.hmax // hs, info.vmax // vs
        exact = hr * hs == info.hmax and vr * vs == info.vmax
        cw, ch = -(-W * hs // info.hmax), -(-H * vs // info.vmax)
        P = planes[c][:ch, :cw].astype(np.int32)
        if exact and (hr, vr) in ((2, 2), (2, 1), (1, 2)):
            # libjpeg's fancy upsampling (jdsample.c): 3/4 of the nearer sample + 1/4 of the further one per axis,
            # edges replicated, alternating rounding biases
            if vr == 2:
                cy = ys >> 1
                yn = np.where(ys & 1, np.minimum(cy + 1, ch - 1), np.maximum(cy - 1, 0))
                rows = 3 * P[cy] + P[yn]                  # [H, cw], scale 4
                vbias = np.where(ys & 1, 2, 1)[:, None]
            else:
                rows, vbias = P[ys], None
            if hr == 2:
                cx = xs >> 1
                xn = np.where(xs & 1, np.minimum(cx + 1, cw - 1), np.maximum(cx - 1, 0))
                if vr == 2:
                    val = (3 * rows[:, cx] + rows[:, xn] + np.where(xs & 1, 7, 8)[None, :]) >> 4
                else:
                    val = (3 * rows[:, cx] + rows[:, xn] + np.where(xs & 1, 2, 1)[None, :]) >> 2
            else:
                val = (rows[:, xs] + vbias) >> 2
            comps.append(val.astype(np.float64))
        else:
            cy = ys * vs // info.vmax
            cx = xs * hs // info.hmax
            comps.append(planes[c][cy][:, cx])
    if info.ncomp == 1:
        return np.repeat(comps[0][:, :, None], 3, axis=2).astype(np.uint8)
    Y, cb, cr = comps[0], comps[1] - 128.0, comps[2] - 128.0
    rgb = np.stack([Y + 1.402 * cr, Y - 0.344136 * cb - 0.714136 * cr, Y + 1.772 * cb], axis=2)
    return np.clip(np.rint(rgb), 0, 255).astype(np.uint8)


class JpegDecodeError(ValueError):
    """An image of a batch that neither the native decoder nor PIL accepts; ``index`` is its position."""

    def __init__(self, index: int, why: str):
        super().__init__("image %d of the batch cannot be decoded: %s" % (index, why))
        self.index = index


# Upper bound on the DCT coefficients staged for one batch (int16 each): 1 Gi coefficients = 2 GiB of pinned
# memory, ~64 images of 16 Mpixel.  Headers are client-controlled, so the sum is checked before allocating.
MAX_BATCH_COEFS = int(os.environ.get("DTFA_JPEG_MAX_BATCH_COEFS", str(1 << 30)))

_pool: Optional[ThreadPoolExecutor] = None
_pool_size = 0


def default_threads() -> int:
    """Host threads for the entropy decode: DTFA_JPEG_THREADS, else the process's thread budget
    (OMP_NUM_THREADS where the launcher sets one), else the CPUs this process may run on; at most 32."""
    for var in ("DTFA_JPEG_THREADS", "OMP_NUM_THREADS"):
        v = os.environ.get(var, "")
        if v.isdigit() and int(v) > 0:
            return min(int(v), 32)
    try:
        return max(1, min(len(os.sched_getaffinity(0)), 32))
    except AttributeError:
        return 8


def _executor(threads: int) -> ThreadPoolExecutor:
    global _pool, _pool_size
    if _pool is None or _pool_size < threads:
        _pool = ThreadPoolExecutor(max_workers=threads, thread_name_prefix="dtfa-jpeg")
        _pool_size = threads
    return _pool


_staging = {}


def _coef_staging(device: torch.device, n_int16: int) -> torch.Tensor:
    """Pinned int16 host buffers (two per device, used alternately) that the entropy decoders fill in
    place: one asynchronous upload per batch, no host-side concatenation."""
    key = str(device)
    ring = _staging.setdefault(key, {"bufs": [None, None], "events": [None, None], "turn": 0})
    k = ring["turn"]
    ring["turn"] = k ^ 1
    if ring["events"][k] is not None:
        ring["events"][k].synchronize()  # the upload that last used this buffer has finished
    buf = ring["bufs"][k]
    if buf is None or buf.numel() < n_int16:
        buf = torch.empty(max(n_int16, 1 << 20), dtype=torch.int16)
        if device.type == "cuda":
            buf = buf.pin_memory()
        ring["bufs"][k] = buf
    return buf, ring, k


def decode_batch_to_device(jpegs: Sequence[bytes], device, threads: Optional[int] = None):
    """Decode a batch of JPEG byte strings.

    Returns ``(packed uint8 device buffer of RGB HWC images, shapes, offsets)``;
    offsets are 16-byte aligned so the buffer can feed ``resize_on_device``
    directly.  Headers are parsed first so that every image's coefficients have a
    fixed place in one pinned staging buffer; entropy decoding then runs on
    ``threads`` host threads straight into that buffer (the C++ call releases the
    GIL) and the batch goes to the device in one asynchronous copy.  Files the
    native decoder rejects are decoded with PIL and uploaded as pixels.
    """
    from ..ops import hipk

    device = torch.device(device)
    threads = threads or default_threads()
    Lh = hostlib.lib()
    n = len(jpegs)
    infos: List[Optional[hostlib.JpegInfo]] = []
    coef_pos = 0
    coef_off = [0] * n
    for i, data in enumerate(jpegs):
        info = hostlib.JpegInfo()
        if Lh.dtfa_jpeg_parse_header(data, len(data), C.byref(info)) != 0 or info.ncomp not in (1, 3):
            infos.append(None)
            continue
        infos.append(info)
        coef_off[i] = coef_pos
        coef_pos += (info.total_coefs + 7) // 8 * 8
        if coef_pos > MAX_BATCH_COEFS:
            raise JpegDecodeError(i, "the batch exceeds %d DCT coefficients" % MAX_BATCH_COEFS)
    staging, ring, turn = _coef_staging(device, coef_pos)
    base = staging.data_ptr()

    def decode_one(i):
        info = infos[i]
        if info is None:
            return False
        return Lh.dtfa_jpeg_decode_coefs(jpegs[i], len(jpegs[i]), C.byref(info), base + 2 * coef_off[i],
                                         info.total_coefs) == 0

    ok = list(_executor(threads).map(decode_one, range(n)))
    shapes: List[Tuple[int, int, int]] = [None] * n  # type: ignore[list-item]
    rgb_off: List[int] = [0] * n
    pos = 0
    fallback = {}
    for i in range(n):
        if not ok[i]:
            try:
                img = decode_pil(jpegs[i])
            except Exception as e:  # noqa: BLE001 -- PIL raises several unrelated types on malformed files
                raise JpegDecodeError(i, "%s: %s" % (type(e).__name__, e)) from None
            fallback[i] = img
            h, w = img.shape[:2]
        else:
            h, w = infos[i].height, infos[i].width
        shapes[i] = (h, w, 3)
        rgb_off[i] = pos
        pos += (h * w * 3 + 15) // 16 * 16
    out = torch.empty(max(pos, 16), dtype=torch.uint8, device=device)
    native = [i for i in range(n) if ok[i]]
    if native:
        descs = (_JpegDesc * len(native))()
        plane_pos = 0
        max_blocks = max_pixels = 0
        for k, i in enumerate(native):
            info = infos[i]
            d = descs[k]
            d.width, d.height, d.ncomp = info.width, info.height, info.ncomp
            d.hmax, d.vmax = info.hmax, info.vmax
            d.rgb_offset = rgb_off[i]
            nblocks = 0
            for c in range(info.ncomp):
                d.hs[c], d.vs[c] = info.hs[c], info.vs[c]
                d.blocks_w[c], d.blocks_h[c] = info.blocks_w[c], info.blocks_h[c]
                d.coef_offset[c] = coef_off[i] + info.coef_offset[c]
                d.plane_offset[c] = plane_pos
                plane_pos += info.blocks_w[c] * info.blocks_h[c] * 64
                nblocks += info.blocks_w[c] * info.blocks_h[c]
                C.memmove(d.qt[c], info.qt[c], 128)
            max_blocks = max(max_blocks, nblocks)
            max_pixels = max(max_pixels, info.width * info.height)
        coef_dev = staging[:coef_pos].to(device, non_blocking=True)
        if device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record()
            ring["events"][turn] = ev
        desc_dev = torch.from_numpy(np.frombuffer(descs, dtype=np.uint8).copy()).to(device, non_blocking=True)
        planes = torch.empty(plane_pos, dtype=torch.uint8, device=device)
        L = hipk.lib()
        L.dtfa_jpeg_decode_batch.argtypes = [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p]
        L.dtfa_jpeg_decode_batch.restype = C.c_int
        L.dtfa_jpeg_desc_size.restype = C.c_int
        if L.dtfa_jpeg_desc_size() != C.sizeof(_JpegDesc):
            raise RuntimeError("JpegDesc layout mismatch")
        hipk.check(L.dtfa_jpeg_decode_batch(coef_dev.data_ptr(), desc_dev.data_ptr(), planes.data_ptr(),
                                            out.data_ptr(), len(native), max_blocks, max_pixels, hipk.stream()),
                   "jpeg_decode_batch")
    for i, img in fallback.items():
        flat = torch.from_numpy(np.ascontiguousarray(img).reshape(-1)).to(device)
        out[rgb_off[i]:rgb_off[i] + flat.numel()] = flat
    return out, shapes, rgb_off


# --------------------------------------------------------------------------------------------
# decoded images resident on the device
# --------------------------------------------------------------------------------------------
class DecodedImageCache:
    """Decoded RGB images kept in device memory.

    The flower dataset is ~3 700 photos of ~0.5 MB decoded: a couple of GB, where the GPU has 288.  Keeping the
    decoded bytes on the device turns every epoch after the first into pure device work -- no file read, no
    entropy decode on the host (the bound of this input path), no upload: the augmentation kernels read the cached
    image in place.  Entries are keyed by ``(path, mtime_ns, size)``, stored in a few large arenas (one allocation
    per 256 MB, not one per image) and never evicted: once ``budget_bytes`` is reached further images are simply
    decoded every time they are drawn.
    """

    def __init__(self, device, budget_bytes: int, arena_bytes: int = 256 << 20):
        import threading

        self.device = torch.device(device)
        self.budget = int(budget_bytes)
        self.arena_bytes = int(arena_bytes)
        self._arenas: List[torch.Tensor] = []
        self._free = 0           # bytes left in the last arena
        self._used = 0           # bytes handed out
        self._entries = {}       # key -> (address, h, w, event)
        self._lock = threading.Lock()
        self.hits = self.misses = 0

    @staticmethod
    def key_of(path: str):
        st = os.stat(path)
        return (os.path.abspath(path), st.st_mtime_ns, st.st_size)

    def __len__(self):
        return len(self._entries)

    @property
    def bytes_used(self) -> int:
        return self._used

    def lookup(self, key):
        """``(address, h, w)`` of a cached image, after making the current stream wait for the copy that stored it."""
        with self._lock:
            e = self._entries.get(key)
            if e is None:
                self.misses += 1
                return None
            self.hits += 1
        addr, h, w, ev = e
        if ev is not None and not ev.query():
            torch.cuda.current_stream(self.device).wait_event(ev)
        return addr, h, w

    def store(self, key, pixels: torch.Tensor, h: int, w: int, event=None) -> bool:
        """Copy ``pixels`` (uint8 device tensor, h * w * 3 bytes) into the cache on the current stream."""
        n = (h * w * 3 + 15) // 16 * 16
        with self._lock:
            if key in self._entries:
                return True
            if self._used + n > self.budget or n > self.arena_bytes:
                return False
            if self._free < n:
                self._arenas.append(torch.empty(self.arena_bytes, dtype=torch.uint8, device=self.device))
                self._free = self.arena_bytes
            arena = self._arenas[-1]
            off = self.arena_bytes - self._free
            self._free -= n
            self._used += n
            dst = arena[off:off + h * w * 3]
            self._entries[key] = (arena.data_ptr() + off, h, w, event)
        dst.copy_(pixels[:h * w * 3], non_blocking=True)
        return True


_caches = {}


def default_cache(device, budget_gb: float) -> Optional[DecodedImageCache]:
    """The process-wide cache of ``device`` (created on first use; ``budget_gb`` <= 0 disables caching)."""
    device = torch.device(device)
    if device.type != "cuda" or budget_gb <= 0:
        return None
    key = str(device)
    c = _caches.get(key)
    if c is None:
        c = _caches[key] = DecodedImageCache(device, int(budget_gb * (1 << 30)))
    return c


def decode_files_to_device(paths: Sequence[str], device, cache: Optional[DecodedImageCache] = None,
                           threads: Optional[int] = None):
    """Decoded RGB images of ``paths`` on the device: ``(base, shapes, offsets)`` like
    :func:`decode_batch_to_device`, except that ``offsets`` are relative to ``base.data_ptr()`` *modulo 2^64*:
    images served from ``cache`` live in its arenas, not in ``base`` (the consumers add the offset to the base
    pointer, which lands on the right address wherever the image is).  Files not in the cache are read, decoded
    and -- budget permitting -- stored."""
    device = torch.device(device)
    n = len(paths)
    found = [None] * n
    keys = [None] * n
    if cache is not None:
        for i, p in enumerate(paths):
            keys[i] = cache.key_of(p)
            found[i] = cache.lookup(keys[i])
    miss = [i for i in range(n) if found[i] is None]
    shapes: List[Tuple[int, int, int]] = [None] * n  # type: ignore[list-item]
    addr = [0] * n
    if miss:
        blobs = []
        for i in miss:
            with open(paths[i], "rb") as f:
                blobs.append(f.read())
        base, mshapes, moffs = decode_batch_to_device(blobs, device, threads)
        ev = None
        for k, i in enumerate(miss):
            shapes[i] = mshapes[k]
            addr[i] = base.data_ptr() + moffs[k]
        if cache is not None:
            ev = torch.cuda.Event()
            for k, i in enumerate(miss):
                h, w, _ = mshapes[k]
                cache.store(keys[i], base[moffs[k]:moffs[k] + h * w * 3], h, w, ev)
            ev.record()
    else:
        base = torch.empty(16, dtype=torch.uint8, device=device)
    for i in range(n):
        if found[i] is not None:
            a, h, w = found[i]
            shapes[i] = (h, w, 3)
            addr[i] = a
    b0 = base.data_ptr()
    offsets = [(a - b0) % (1 << 64) for a in addr]
    return base, shapes, offsets
