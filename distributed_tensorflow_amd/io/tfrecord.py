"""TFRecord files and tf.train.Example records (native C++ codec behind ctypes).

The reference writes its training set with ``tf.python_io.TFRecordWriter`` +
``tf.train.Example`` (recognition_flower/distributed/prepare.py:36-53) and
reads it back with ``TFRecordDataset`` + ``parse_single_example``
(recognition_flower/distributed/train_dataset.py:68-94).  These classes
implement the same on-disk format without TensorFlow.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterator, List, Sequence, Tuple, Union

import numpy as np

from . import hostlib

FeatureValue = Union[bytes, Sequence[bytes], Sequence[int], Sequence[float], int, float, np.ndarray]


def crc32c(data: bytes) -> int:
    return hostlib.lib().dtfa_crc32c(data, len(data))


def masked_crc32c(data: bytes) -> int:
    return hostlib.lib().dtfa_crc32c_masked(data, len(data))


class TFRecordWriter:
    def __init__(self, path: str, append: bool = False):
        self._L = hostlib.lib()
        self._h = self._L.dtfa_tfrecord_writer_open(path.encode(), 1 if append else 0)
        if not self._h:
            raise OSError("cannot open %s for writing" % path)

    def write(self, record: bytes) -> None:
        if self._L.dtfa_tfrecord_write(self._h, record, len(record)) != 0:
            raise OSError("short write")

    def flush(self) -> None:
        self._L.dtfa_tfrecord_writer_flush(self._h)

    def close(self) -> None:
        if self._h:
            self._L.dtfa_tfrecord_writer_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


_READ_ERRORS = {-1: "truncated record header", -2: "corrupted record length", -3: "truncated record data",
                -4: "corrupted record data (crc mismatch)"}


def tfrecord_iterator(path: str, verify_crc: bool = True) -> Iterator[bytes]:
    L = hostlib.lib()
    h = L.dtfa_tfrecord_reader_open(path.encode(), 1 if verify_crc else 0)
    if not h:
        raise FileNotFoundError(path)
    try:
        data = C.POINTER(C.c_uint8)()
        n = C.c_uint64()
        while True:
            rc = L.dtfa_tfrecord_read_next(h, C.byref(data), C.byref(n))
            if rc == 0:
                return
            if rc < 0:
                raise IOError("%s: %s" % (path, _READ_ERRORS.get(rc, "read error %d" % rc)))
            yield C.string_at(data, n.value)
    finally:
        L.dtfa_tfrecord_reader_close(h)


def tfrecord_index(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """Byte offset and payload length of every record."""
    L = hostlib.lib()
    n = L.dtfa_tfrecord_index(path.encode(), None, None, 0)
    if n < 0:
        raise IOError("cannot index %s (error %d)" % (path, n))
    off = np.zeros(n, dtype=np.uint64)
    ln = np.zeros(n, dtype=np.uint64)
    if n:
        got = L.dtfa_tfrecord_index(path.encode(), off.ctypes.data_as(C.POINTER(C.c_uint64)),
                                    ln.ctypes.data_as(C.POINTER(C.c_uint64)), n)
        if got != n:
            raise IOError("file changed while indexing")
    return off, ln


def make_example(features: Dict[str, FeatureValue]) -> bytes:
    """Serialise a ``tf.train.Example``.  bytes -> BytesList, ints -> Int64List, floats -> FloatList."""
    L = hostlib.lib()
    b = L.dtfa_example_builder_new()
    try:
        for key, v in features.items():
            k = key.encode()
            if isinstance(v, (bytes, bytearray, memoryview)):
                v = [bytes(v)]
            elif isinstance(v, (int, np.integer)) and not isinstance(v, bool):
                v = [int(v)]
            elif isinstance(v, (float, np.floating)):
                v = [float(v)]
            if isinstance(v, np.ndarray):
                v = v.tolist()
            v = list(v)
            if v and isinstance(v[0], (bytes, bytearray)):
                for item in v:
                    L.dtfa_example_add_bytes(b, k, bytes(item), len(item))
            elif v and isinstance(v[0], float):
                arr = (C.c_float * len(v))(*v)
                L.dtfa_example_add_float(b, k, arr, len(v))
            else:
                arr = (C.c_longlong * len(v))(*[int(x) for x in v])
                L.dtfa_example_add_int64(b, k, arr, len(v))
        data = C.POINTER(C.c_uint8)()
        n = C.c_uint64()
        L.dtfa_example_serialize(b, C.byref(data), C.byref(n))
        return C.string_at(data, n.value)
    finally:
        L.dtfa_example_builder_free(b)


def parse_example(record: bytes) -> Dict[str, Union[List[bytes], List[int], List[float]]]:
    L = hostlib.lib()
    h = L.dtfa_example_parse(record, len(record))
    if not h:
        raise ValueError("malformed tf.Example")
    try:
        out = {}
        for i in range(L.dtfa_example_num_features(h)):
            name = L.dtfa_example_feature_name(h, i)
            cnt = C.c_int()
            kind = L.dtfa_example_feature_info(h, name, C.byref(cnt))
            key = name.decode()
            if kind == 1:
                vals = []
                for j in range(cnt.value):
                    p = C.POINTER(C.c_uint8)()
                    n = C.c_uint64()
                    L.dtfa_example_get_bytes(h, name, j, C.byref(p), C.byref(n))
                    vals.append(C.string_at(p, n.value))
                out[key] = vals
            elif kind == 2:
                arr = (C.c_float * max(cnt.value, 1))()
                L.dtfa_example_get_float(h, name, arr, cnt.value)
                out[key] = list(arr)[:cnt.value]
            elif kind == 3:
                arr = (C.c_longlong * max(cnt.value, 1))()
                L.dtfa_example_get_int64(h, name, arr, cnt.value)
                out[key] = list(arr)[:cnt.value]
            else:
                out[key] = []
        return out
    finally:
        L.dtfa_example_free(h)
