"""TensorFlow Saver-V2 checkpoints ("tensor bundles") and the ``checkpoint`` state file.

Layout on disk (reference: ``CheckpointSaverHook`` writes them,
recognition_flower/distributed/train_dataset.py:176-178; ``save_model.py``
reads the newest one through ``get_checkpoint_state``,
recognition_flower/distributed/save_model.py:110-112)::

    <dir>/model.ckpt-<step>.index                  sorted table: name -> dtype/shape/offset/size/crc
    <dir>/model.ckpt-<step>.data-00000-of-00001    raw little-endian tensor bytes in key order
    <dir>/checkpoint                               text: model_checkpoint_path + all_model_checkpoint_paths

The table / bundle encoding lives in ``csrc/host/{sstable,tensor_bundle}.cc``.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import hostlib

# TensorFlow DataType enum values
DT = {"float32": 1, "float64": 2, "int32": 3, "uint8": 4, "int16": 5, "int8": 6, "int64": 9, "bool": 10,
      "bfloat16": 14, "float16": 19}
_DT_INV = {v: k for k, v in DT.items()}


def write_bundle(prefix: str, tensors: Dict[str, np.ndarray]) -> None:
    """Write ``<prefix>.index`` and ``<prefix>.data-00000-of-00001`` atomically."""
    L = hostlib.lib()
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    h = L.dtfa_bundle_writer_new(prefix.encode())
    if not h:
        raise OSError("cannot create bundle %s" % prefix)
    try:
        for name in sorted(tensors, key=lambda s: s.encode()):
            a = tensors[name]
            if isinstance(a, tuple):  # (raw little-endian bytes as uint16 array, "bfloat16")
                arr, dtname = a
            else:
                arr = np.asarray(a)
                dtname = arr.dtype.name
            if dtname not in DT:
                raise TypeError("unsupported dtype %s for %s" % (dtname, name))
            if not arr.flags.c_contiguous:  # (ascontiguousarray would promote 0-d scalars to 1-d)
                arr = np.ascontiguousarray(arr)
            dims = (C.c_longlong * max(arr.ndim, 1))(*arr.shape)
            rc = L.dtfa_bundle_add(h, name.encode(), DT[dtname], dims, arr.ndim, arr.ctypes.data, arr.nbytes)
            if rc != 0:
                raise OSError("bundle add failed for %s (%d)" % (name, rc))
    except Exception:
        L.dtfa_bundle_writer_abort(h)
        raise
    rc = L.dtfa_bundle_writer_finish(h)
    if rc != 0:
        raise OSError("bundle finish failed (%d)" % rc)


def list_bundle(prefix: str) -> List[Tuple[str, str, Tuple[int, ...], int, int, int]]:
    """[(name, dtype, shape, offset, size, masked_crc)] in key order."""
    L = hostlib.lib()
    h = L.dtfa_bundle_reader_open(prefix.encode())
    try:
        err = L.dtfa_bundle_reader_error(h)
        if err:
            raise IOError("%s: %s" % (prefix, err.decode()))
        if L.dtfa_bundle_num_shards(h) != 1:
            raise IOError("multi-shard bundles are not supported")
        out = []
        for i in range(L.dtfa_bundle_num_entries(h)):
            name = C.c_char_p()
            dtype, shard = C.c_int(), C.c_int()
            dims = (C.c_longlong * 8)()
            off, size = C.c_uint64(), C.c_uint64()
            crc = C.c_uint32()
            nd = L.dtfa_bundle_entry(h, i, C.byref(name), C.byref(dtype), dims, C.byref(shard), C.byref(off),
                                     C.byref(size), C.byref(crc))
            if nd < 0:
                raise IOError("bad entry %d" % i)
            out.append((name.value.decode(), _DT_INV.get(dtype.value, "dt%d" % dtype.value),
                        tuple(int(dims[k]) for k in range(nd)), off.value, size.value, crc.value))
        return out
    finally:
        L.dtfa_bundle_reader_close(h)


def read_bundle(prefix: str, names: Optional[List[str]] = None, verify_crc: bool = True) -> Dict[str, np.ndarray]:
    L = hostlib.lib()
    entries = list_bundle(prefix)
    want = set(names) if names is not None else None
    data = np.memmap(prefix + ".data-00000-of-00001", dtype=np.uint8, mode="r") if entries else None
    out = {}
    for name, dtname, shape, off, size, crc in entries:
        if want is not None and name not in want:
            continue
        raw = np.asarray(data[off:off + size])
        if raw.nbytes != size:
            raise IOError("data file truncated at %s" % name)
        if verify_crc and L.dtfa_crc32c_masked(raw.ctypes.data, raw.nbytes) != crc:
            raise IOError("crc mismatch for %s" % name)
        if dtname == "bfloat16":
            out[name] = (raw.view(np.uint16).reshape(shape).copy(), "bfloat16")
        else:
            out[name] = raw.view(np.dtype(dtname)).reshape(shape).copy()
    return out


# ----------------------------------------------------------------------------
# `checkpoint` state file
# ----------------------------------------------------------------------------
def read_checkpoint_state(directory: str) -> Tuple[Optional[str], List[str]]:
    path = os.path.join(directory, "checkpoint")
    if not os.path.exists(path):
        return None, []
    latest, allp = None, []
    with open(path) as f:
        for line in f:
            m = re.match(r'\s*(model_checkpoint_path|all_model_checkpoint_paths)\s*:\s*"(.*)"\s*$', line)
            if not m:
                continue
            if m.group(1) == "model_checkpoint_path":
                latest = m.group(2)
            else:
                allp.append(m.group(2))
    return latest, allp


def latest_checkpoint(directory: str) -> Optional[str]:
    """Absolute prefix of the newest checkpoint, like ``tf.train.latest_checkpoint``."""
    latest, _ = read_checkpoint_state(directory)
    if latest is None:
        return None
    prefix = latest if os.path.isabs(latest) else os.path.join(directory, latest)
    return prefix if os.path.exists(prefix + ".index") else None


def update_checkpoint_state(directory: str, prefix_basename: str, max_to_keep: int = 5) -> List[str]:
    """Record a new checkpoint, delete the ones beyond ``max_to_keep``; returns deleted basenames."""
    _, allp = read_checkpoint_state(directory)
    allp = [p for p in allp if p != prefix_basename] + [prefix_basename]
    deleted = []
    while max_to_keep and len(allp) > max_to_keep:
        old = allp.pop(0)
        deleted.append(old)
        base = old if os.path.isabs(old) else os.path.join(directory, old)
        for suffix in (".index", ".data-00000-of-00001", ".meta"):
            try:
                os.remove(base + suffix)
            except OSError:
                pass
    tmp = os.path.join(directory, "checkpoint.tmp%d" % os.getpid())
    with open(tmp, "w") as f:
        f.write('model_checkpoint_path: "%s"\n' % prefix_basename)
        for p in allp:
            f.write('all_model_checkpoint_paths: "%s"\n' % p)
    os.replace(tmp, os.path.join(directory, "checkpoint"))
    return deleted
