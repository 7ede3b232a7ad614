"""TensorBoard event files (``tf.summary.FileWriter`` / ``SummarySaverHook`` equivalent).

The reference logs scalars ``learning_rate``, ``loss``, ``accuracy`` and a
histogram ``logits histogram`` (recognition_flower/single/flower_train.py:118,137,146,161;
recognition_flower/distributed/train_dataset.py:153,161).  TensorFlow-1
sanitises summary tags (characters outside ``[-/\\w.]`` become ``_``), so the
histogram shows up as ``logits_histogram``; :func:`clean_tag` reproduces that.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import struct
from typing import Dict, Iterator, Optional, Sequence, Union

import numpy as np

from . import hostlib
from .tfrecord import tfrecord_iterator


def clean_tag(tag: str) -> str:
    tag = re.sub(r"[^-/\w.]", "_", tag)
    return tag.lstrip("/")


class EventWriter:
    def __init__(self, logdir: str, filename_suffix: str = ""):
        os.makedirs(logdir, exist_ok=True)
        self._L = hostlib.lib()
        self._h = self._L.dtfa_events_open(logdir.encode(), filename_suffix.encode())
        if not self._h:
            raise OSError("cannot create event file in %s" % logdir)
        self.path = self._L.dtfa_events_path(self._h).decode()

    def add_summary(self, step: int, scalars: Optional[Dict[str, float]] = None,
                    histograms: Optional[Dict[str, Union[np.ndarray, Sequence[float]]]] = None,
                    wall_time: float = 0.0) -> None:
        """One Event holding a merged Summary (like ``merge_all`` + ``add_summary``)."""
        L = self._L
        L.dtfa_events_begin(self._h, int(step), float(wall_time))
        for tag, v in (scalars or {}).items():
            L.dtfa_events_add_scalar(self._h, clean_tag(tag).encode(), float(v))
        for tag, vals in (histograms or {}).items():
            a = np.ascontiguousarray(np.asarray(vals, dtype=np.float32).reshape(-1))
            L.dtfa_events_add_histogram(self._h, clean_tag(tag).encode(), a.ctypes.data_as(C.POINTER(C.c_float)),
                                        a.size)
        L.dtfa_events_end(self._h)

    def flush(self) -> None:
        self._L.dtfa_events_flush(self._h)

    def close(self) -> None:
        if self._h:
            self._L.dtfa_events_close(self._h)
            self._h = None


# ----------------------------------------------------------------------------
# Minimal reader (tests, tooling): decodes the subset of Event this package writes.
# ----------------------------------------------------------------------------
def _varint(b: bytes, i: int):
    v = shift = 0
    while True:
        c = b[i]
        i += 1
        v |= (c & 0x7F) << shift
        if not c & 0x80:
            return v, i
        shift += 7


def _fields(b: bytes):
    i = 0
    while i < len(b):
        t, i = _varint(b, i)
        f, w = t >> 3, t & 7
        if w == 0:
            v, i = _varint(b, i)
        elif w == 1:
            v = b[i:i + 8]
            i += 8
        elif w == 2:
            n, i = _varint(b, i)
            v = b[i:i + n]
            i += n
        elif w == 5:
            v = b[i:i + 4]
            i += 4
        else:
            raise ValueError("bad wire type")
        yield f, w, v


def read_events(path: str) -> Iterator[dict]:
    for rec in tfrecord_iterator(path):
        ev = {"wall_time": 0.0, "step": 0, "file_version": None, "scalars": {}, "histograms": {}}
        for f, w, v in _fields(rec):
            if f == 1:
                ev["wall_time"] = struct.unpack("<d", v)[0]
            elif f == 2:
                ev["step"] = v
            elif f == 3:
                ev["file_version"] = v.decode()
            elif f == 5:
                for f2, _, val in _fields(v):
                    if f2 != 1:
                        continue
                    tag, simple, histo = None, None, None
                    for f3, _, x in _fields(val):
                        if f3 == 1:
                            tag = x.decode()
                        elif f3 == 2:
                            simple = struct.unpack("<f", x)[0]
                        elif f3 == 5:
                            h = {}
                            for f4, _, y in _fields(x):
                                if f4 in (1, 2, 3, 4, 5):
                                    h[{1: "min", 2: "max", 3: "num", 4: "sum", 5: "sum_squares"}[f4]] = \
                                        struct.unpack("<d", y)[0]
                                elif f4 == 6:
                                    h["bucket_limit"] = list(struct.unpack("<%dd" % (len(y) // 8), y))
                                elif f4 == 7:
                                    h["bucket"] = list(struct.unpack("<%dd" % (len(y) // 8), y))
                            histo = h
                    if histo is not None:
                        ev["histograms"][tag] = histo
                    elif tag is not None:
                        ev["scalars"][tag] = simple if simple is not None else 0.0
        yield ev
