"""Image preprocessing: crop / flip / resize / normalise to the network input.

GPU path: one launch of ``resize_norm`` (csrc/hip/resize.hip) over a packed
batch of variable-size uint8 images.  CPU path (``resize_reference``): the same
arithmetic in numpy, used for CPU training and as the kernel's oracle.

Resize methods reproduce the TensorFlow-1 ops the reference calls:

* ``bilinear_legacy`` -- ``tf.image.resize_images(image, [299, 299], method=0)``
  in the TFRecord reader (recognition_flower/distributed/train_dataset.py:83);
* ``bicubic_legacy``  -- ``tf.image.resize_bicubic(..., align_corners=False)`` in
  the serving preprocess (recognition_flower/single/flower_train.py:56-59);
* ``bilinear``        -- half-pixel-centre bilinear, used for the augmented
  training crops (the reference used ``scipy.misc.imresize`` there,
  recognition_flower/single/flower_input_data.py:50).

Normalisation: ``out = pixel * scale + shift``; ``NORM_INCEPTION`` maps [0,255]
to [-1,1] ("inception input requirement",
recognition_flower/single/flower_input_data.py:79).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

METHODS = {"bilinear_legacy": 0, "bilinear": 1, "bicubic_legacy": 2}
NORM_INCEPTION = ((2.0 / 255.0,) * 3, (-1.0,) * 3)
NORM_UNIT = ((1.0 / 255.0,) * 3, (0.0,) * 3)
NORM_NONE = ((1.0,) * 3, (0.0,) * 3)
CHANNELS_PADDED = 8


@dataclass
class CropSpec:
    y0: float
    x0: float
    h: float
    w: float
    flip: bool = False


def full_crop(h: int, w: int) -> CropSpec:
    return CropSpec(0.0, 0.0, float(h), float(w), False)


def random_crop_spec(h: int, w: int, target_h: int, target_w: int, max_zoom: float, rng: np.random.Generator) -> CropSpec:
    """The reference's augmentation (recognition_flower/single/flower_input_data.py:28-48):
    largest box with the target aspect ratio, shrunk by a zoom factor in
    [1, 1 + max_zoom), placed uniformly at random, flipped with probability 1/2."""
    image_ratio = w / h
    target_ratio = target_w / target_h
    crop_vertically = image_ratio < target_ratio
    crop_w = w if crop_vertically else int(h * target_ratio)
    crop_h = int(w / target_ratio) if crop_vertically else h
    resize_factor = rng.random() * max_zoom + 1.0
    crop_w = max(1, int(crop_w / resize_factor))
    crop_h = max(1, int(crop_h / resize_factor))
    x0 = int(rng.integers(0, w - crop_w + 1))
    y0 = int(rng.integers(0, h - crop_h + 1))
    return CropSpec(float(y0), float(x0), float(crop_h), float(crop_w), bool(rng.random() < 0.5))


def _cubic_w(x, A=-0.75):
    x = np.abs(x)
    return np.where(x <= 1, ((A + 2) * x - (A + 3)) * x * x + 1,
                    np.where(x < 2, ((A * x - 5 * A) * x + 8 * A) * x - 4 * A, 0.0))


def resize_reference(img: np.ndarray, out_hw: Tuple[int, int], method: str = "bilinear_legacy",
                     crop: Optional[CropSpec] = None, bgr: bool = False,
                     norm=NORM_INCEPTION) -> np.ndarray:
    """numpy implementation of the resize kernel.  img: uint8 [H, W, C]; returns float32 [OH, OW, 3]."""
    if img.ndim == 2:
        img = img[:, :, None]
    h, w, c = img.shape
    src = img[:, :, :3].astype(np.float32) if c >= 3 else np.repeat(img[:, :, :1].astype(np.float32), 3, axis=2)
    crop = crop or full_crop(h, w)
    OH, OW = out_hw
    oy = np.arange(OH, dtype=np.float32)
    ox = np.arange(OW, dtype=np.float32)
    if crop.flip:
        ox = (OW - 1) - ox
    ys, xs = np.float32(crop.h / OH), np.float32(crop.w / OW)
    m = METHODS[method]
    if m == 1:
        fy = np.float32(crop.y0) + (oy + np.float32(0.5)) * ys - np.float32(0.5)
        fx = np.float32(crop.x0) + (ox + np.float32(0.5)) * xs - np.float32(0.5)
    else:
        fy = np.float32(crop.y0) + oy * ys
        fx = np.float32(crop.x0) + ox * xs
    if m == 2:
        y0 = np.floor(fy)
        x0 = np.floor(fx)
        out = np.zeros((OH, OW, 3), dtype=np.float32)
        for ky in range(4):
            yy = np.clip(y0 - 1 + ky, 0, h - 1).astype(np.int64)
            wy = _cubic_w(fy - (y0 - 1 + ky)).astype(np.float32)
            row = np.zeros((OH, OW, 3), dtype=np.float32)
            for kx in range(4):
                xx = np.clip(x0 - 1 + kx, 0, w - 1).astype(np.int64)
                wx = _cubic_w(fx - (x0 - 1 + kx)).astype(np.float32)
                row += src[yy][:, xx] * wx[None, :, None]
            out += row * wy[:, None, None]
    else:
        fy = np.maximum(fy, 0)
        fx = np.maximum(fx, 0)
        y0 = np.floor(fy).astype(np.int64)
        x0 = np.floor(fx).astype(np.int64)
        y0c, x0c = np.clip(y0, 0, h - 1), np.clip(x0, 0, w - 1)
        y1, x1 = np.minimum(y0 + 1, h - 1), np.minimum(x0 + 1, w - 1)
        ly = (fy - y0).astype(np.float32)[:, None, None]
        lx = (fx - x0).astype(np.float32)[None, :, None]
        p00, p01 = src[y0c][:, x0c], src[y0c][:, x1]
        p10, p11 = src[y1][:, x0c], src[y1][:, x1]
        top = p00 + (p01 - p00) * lx
        bot = p10 + (p11 - p10) * lx
        out = top + (bot - top) * ly
    if bgr:
        out = out[:, :, ::-1]
    scale = np.asarray(norm[0], dtype=np.float32)
    shift = np.asarray(norm[1], dtype=np.float32)
    return (out * scale + shift).astype(np.float32)


class _ImageDesc(C.Structure):
    _fields_ = [("offset", C.c_ulonglong), ("h", C.c_int), ("w", C.c_int), ("c", C.c_int),
                ("crop_y0", C.c_float), ("crop_x0", C.c_float), ("crop_h", C.c_float), ("crop_w", C.c_float),
                ("flip", C.c_int), ("pad", C.c_int)]


class _ResizeArgs(C.Structure):
    _fields_ = [("src", C.c_void_p), ("desc", C.c_void_p), ("dst", C.c_void_p),
                ("B", C.c_int), ("OH", C.c_int), ("OW", C.c_int), ("CP", C.c_int),
                ("method", C.c_int), ("bgr", C.c_int), ("scale", C.c_float * 3), ("shift", C.c_float * 3)]


def pack_images(images: Sequence[np.ndarray]) -> Tuple[np.ndarray, List[int]]:
    """Concatenate uint8 HWC images into one buffer with 16-byte aligned starts."""
    offsets, pos = [], 0
    for im in images:
        offsets.append(pos)
        pos += (im.size + 15) // 16 * 16
    buf = np.zeros(max(pos, 16), dtype=np.uint8)
    for im, off in zip(images, offsets):
        buf[off:off + im.size] = np.ascontiguousarray(im).reshape(-1)
    return buf, offsets


def make_descs(shapes: Sequence[Tuple[int, int, int]], offsets: Sequence[int],
               crops: Optional[Sequence[Optional[CropSpec]]] = None) -> np.ndarray:
    arr = (_ImageDesc * len(shapes))()
    for i, ((h, w, c), off) in enumerate(zip(shapes, offsets)):
        cs = (crops[i] if crops is not None and crops[i] is not None else full_crop(h, w))
        arr[i] = _ImageDesc(off, h, w, c, cs.y0, cs.x0, cs.h, cs.w, 1 if cs.flip else 0, 0)
    return np.frombuffer(arr, dtype=np.uint8).copy()


def resize_on_device(packed: torch.Tensor, descs: torch.Tensor, batch: int, out_hw: Tuple[int, int],
                     method: str = "bilinear_legacy", bgr: bool = False, norm=NORM_INCEPTION,
                     channels: int = CHANNELS_PADDED) -> torch.Tensor:
    """packed: uint8 device buffer, descs: uint8 device buffer of ``_ImageDesc`` records.
    Returns bf16 [B, OH, OW, channels] (3 real channels, the rest zero; 8 by default, 4 for models whose stem reads
    pixel pairs, ``model.input_channels``)."""
    from ..ops import hipk

    if not packed.is_cuda or packed.dtype != torch.uint8 or not descs.is_cuda:
        raise ValueError("resize_on_device needs uint8 GPU buffers")
    if descs.numel() != batch * C.sizeof(_ImageDesc):
        raise ValueError("descriptor table size mismatch")
    OH, OW = out_hw
    if channels < 3:
        raise ValueError("at least the three colour channels")
    out = torch.empty((batch, OH, OW, channels), dtype=torch.bfloat16, device=packed.device)
    a = _ResizeArgs()
    a.src, a.desc, a.dst = packed.data_ptr(), descs.data_ptr(), out.data_ptr()
    a.B, a.OH, a.OW, a.CP = batch, OH, OW, channels
    a.method, a.bgr = METHODS[method], 1 if bgr else 0
    for k in range(3):
        a.scale[k] = norm[0][k]
        a.shift[k] = norm[1][k]
    L = hipk.lib()
    L.dtfa_resize_norm.argtypes = [C.POINTER(_ResizeArgs), C.c_void_p]
    L.dtfa_resize_norm.restype = C.c_int
    hipk.check(L.dtfa_resize_norm(C.byref(a), hipk.stream()), "resize_norm")
    return out


def preprocess_batch(images: Sequence[np.ndarray], out_hw: Tuple[int, int], device, method: str = "bilinear_legacy",
                     crops: Optional[Sequence[Optional[CropSpec]]] = None, bgr: bool = False,
                     norm=NORM_INCEPTION) -> torch.Tensor:
    """uint8 HWC images -> network input ``[B, OH, OW, 8]`` (bf16 on a GPU, fp32 on the CPU)."""
    device = torch.device(device)
    if device.type == "cuda":
        buf, offs = pack_images(images)
        descs = make_descs([im.shape if im.ndim == 3 else im.shape + (1,) for im in images], offs, crops)
        return resize_on_device(torch.from_numpy(buf).to(device, non_blocking=True),
                                torch.from_numpy(descs).to(device, non_blocking=True), len(images), out_hw, method,
                                bgr, norm)
    out = np.zeros((len(images), out_hw[0], out_hw[1], CHANNELS_PADDED), dtype=np.float32)
    for i, im in enumerate(images):
        out[i, :, :, :3] = resize_reference(im, out_hw, method, crops[i] if crops is not None else None, bgr, norm)
    return torch.from_numpy(out)
