"""Minimal ``<prefix>.meta`` companion of a Saver-V2 checkpoint.

TensorFlow's ``Saver.save`` writes ``model.ckpt-N.meta``, a serialized ``MetaGraphDef``, next to the
``.index`` / ``.data`` pair (reference: ``CheckpointSaverHook`` with ``Saver()``,
recognition_flower/distributed/train_dataset.py:169-178).  The graph itself (``graph_def``) is a
TensorFlow program and cannot be produced without TensorFlow, and nothing that reads checkpoints needs
it (``latest_checkpoint``, ``Saver.restore`` and the reference's ``save_model.py`` only use the bundle
and the ``checkpoint`` state file).  What *is* schema-level information is written here, in
TensorFlow's own protobuf layout, so that the file set matches and the variable inventory is
machine-readable:

``MetaGraphDef``
    1  ``meta_info_def``  { 4 tags: "train", 5 tensorflow_version, 6 tensorflow_git_version }
    3  ``saver_def``      { 1 filename_tensor_name, 2 save_tensor_name, 3 restore_op_name,
                            4 max_to_keep, 7 version = V2 }
    4  ``collection_def`` map: "variables", "trainable_variables" -> ``bytes_list`` of ``VariableDef``
                            { 1 variable_name "<name>:0", 2 initializer_name, 3 snapshot_name, 7 trainable },
                            "global_step" -> the step variable

The encoder below is a plain protobuf wire-format writer (varints and length-delimited fields).
"""
from __future__ import annotations

import struct
from typing import Dict, Iterable, List, Sequence, Tuple

PRODUCER = "distributed-tensorflow_amd"


def _varint(v: int) -> bytes:
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field: int, wire: int) -> bytes:
    return _varint((field << 3) | wire)


def _bytes(field: int, payload: bytes) -> bytes:
    return _key(field, 2) + _varint(len(payload)) + payload


def _str(field: int, s: str) -> bytes:
    return _bytes(field, s.encode())


def _int(field: int, v: int) -> bytes:
    return _key(field, 0) + _varint(v)


def variable_def(name: str, trainable: bool) -> bytes:
    return (_str(1, name + ":0") + _str(2, name + "/Assign") + _str(3, name + "/read:0")
            + (_int(7, 1) if trainable else b""))


def _collection(key: str, items: Iterable[bytes]) -> bytes:
    bytes_list = b"".join(_bytes(1, it) for it in items)          # BytesList.value
    coll = _bytes(2, bytes_list)                                    # CollectionDef.bytes_list
    return _bytes(4, _str(1, key) + _bytes(2, coll))               # map entry of MetaGraphDef.collection_def


def encode_meta_graph(variables: Sequence[Tuple[str, bool]], max_to_keep: int = 5, global_step_name: str = "global_step"
                      ) -> bytes:
    """``variables``: (TensorFlow variable name, trainable) in save order."""
    meta_info = _str(4, "train") + _str(5, PRODUCER) + _str(6, "saver-v2-compatible")
    saver = (_str(1, "save/Const:0") + _str(2, "save/control_dependency:0") + _str(3, "save/restore_all")
             + _int(4, max_to_keep) + _key(6, 5) + struct.pack("<f", 10000.0) + _int(7, 2))
    out = _bytes(1, meta_info) + _bytes(3, saver)
    out += _collection("variables", [variable_def(n, t) for n, t in variables])
    out += _collection("trainable_variables", [variable_def(n, True) for n, t in variables if t])
    out += _collection("global_step", [variable_def(global_step_name, False)])
    return out


# ------------------------------------------------------------------ reader (tests, tooling)
def _read_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    v = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, pos
        shift += 7


def _fields(buf: bytes) -> List[Tuple[int, int, object]]:
    pos, out = 0, []
    while pos < len(buf):
        k, pos = _read_varint(buf, pos)
        field, wire = k >> 3, k & 7
        if wire == 0:
            v, pos = _read_varint(buf, pos)
        elif wire == 2:
            n, pos = _read_varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        elif wire == 5:
            v = buf[pos:pos + 4]
            pos += 4
        elif wire == 1:
            v = buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError("unsupported wire type %d" % wire)
        out.append((field, wire, v))
    return out


def decode_meta_graph(buf: bytes) -> Dict[str, object]:
    """{'tags': [...], 'saver': {...}, 'collections': {key: [(variable_name, trainable), ...]}}"""
    res: Dict[str, object] = {"tags": [], "saver": {}, "collections": {}}
    for field, _, v in _fields(buf):
        if field == 1:
            for f, _, x in _fields(v):
                if f == 4:
                    res["tags"].append(bytes(x).decode())
                elif f == 5:
                    res["producer"] = bytes(x).decode()
        elif field == 3:
            names = {1: "filename_tensor_name", 2: "save_tensor_name", 3: "restore_op_name", 4: "max_to_keep", 7: "version"}
            for f, w, x in _fields(v):
                if f in names:
                    res["saver"][names[f]] = bytes(x).decode() if w == 2 else x
        elif field == 4:
            key, items = None, []
            for f, _, x in _fields(v):
                if f == 1:
                    key = bytes(x).decode()
                elif f == 2:
                    for f2, _, bl in _fields(x):
                        if f2 == 2:
                            for _, _, vd in _fields(bl):
                                d = {ff: xx for ff, _, xx in _fields(vd)}
                                items.append((bytes(d[1]).decode(), bool(d.get(7, 0))))
            res["collections"][key] = items
    return res
