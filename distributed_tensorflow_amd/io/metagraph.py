"""``<prefix>.meta`` companion of a Saver-V2 checkpoint and the ``saved_model.pb`` of an export.

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

``saved_model.pb`` (reference: ``SavedModelBuilder.add_meta_graph_and_variables(sess, [SERVING],
signature_def_map=..., legacy_init_op=...)``, recognition_flower/single/flower_train.py:204-250,
recognition_flower/distributed/save_model.py:123-170) is a ``SavedModel`` message whose one
``MetaGraphDef`` is tagged ``serve``.  Everything in it that is schema and not TensorFlow program is
written: tags, the SaverDef of ``variables/variables``, the variable collections, the
``legacy_init_op`` collection and the two ``SignatureDef`` entries with their ``TensorInfo`` (tensor
name, dtype, shape) and method names.  ``graph_def`` stays empty: the network is rebuilt from the
manifest next to it by :class:`..io.export.Servable`, not interpreted from graph nodes.

``SavedModel``     { 1 saved_model_schema_version = 1, 2 meta_graphs: MetaGraphDef }
``MetaGraphDef``   additionally { 5 signature_def map<string, SignatureDef> }
``SignatureDef``   { 1 inputs map<string, TensorInfo>, 2 outputs map<string, TensorInfo>, 3 method_name }
``TensorInfo``     { 1 name, 2 dtype, 3 tensor_shape { 2 dim { 1 size }, 3 unknown_rank } }

The encoder below is a plain protobuf wire-format writer (varints and length-delimited fields).
"""
from __future__ import annotations

import struct
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

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


# ------------------------------------------------------------------ saved_model.pb
DT_FLOAT, DT_STRING = 1, 7
_DTYPES = {"float": DT_FLOAT, "string": DT_STRING}
_DTYPE_NAMES = {v: k for k, v in _DTYPES.items()}


def tensor_info(name: str, dtype: str, shape: Optional[Sequence[int]]) -> bytes:
    """``shape`` None = unknown rank (a placeholder declared without a shape); -1 = unknown dimension."""
    if shape is None:
        shp = _int(3, 1)
    else:
        shp = b"".join(_bytes(2, _int(1, d)) for d in shape)
    return _str(1, name) + _int(2, _DTYPES[dtype]) + _bytes(3, shp)


def signature_def(inputs: Dict[str, bytes], outputs: Dict[str, bytes], method_name: str) -> bytes:
    out = b""
    for k in sorted(inputs):
        out += _bytes(1, _str(1, k) + _bytes(2, inputs[k]))
    for k in sorted(outputs):
        out += _bytes(2, _str(1, k) + _bytes(2, outputs[k]))
    return out + _str(3, method_name)


def encode_saved_model(variables: Sequence[Tuple[str, bool]], signatures: Dict[str, Dict]) -> bytes:
    """``signatures``: {key: {"method_name": str, "inputs" / "outputs": {name: {"tensor": str, "dtype": "string" |
    "float", "shape": [..] | None}}}} -- the manifest layout of :func:`..io.export.signature_manifest`."""
    meta_info = _str(4, "serve") + _str(5, PRODUCER) + _str(6, "saver-v2-compatible")
    saver = (_str(1, "save/Const:0") + _str(2, "save/control_dependency:0") + _str(3, "save/restore_all")
             + _int(5, 1) + _key(6, 5) + struct.pack("<f", 10000.0) + _int(7, 2))      # 5: sharded, as the builder saves
    mg = _bytes(1, meta_info) + _bytes(3, saver)
    mg += _collection("variables", [variable_def(n, t) for n, t in variables])
    mg += _collection("trainable_variables", [variable_def(n, True) for n, t in variables if t])
    node_list = _bytes(1, _str(1, "legacy_init_op"))                                   # CollectionDef.node_list
    mg += _bytes(4, _str(1, "legacy_init_op") + _bytes(2, node_list))
    for key in sorted(signatures):
        sig = signatures[key]
        ti = {side: {n: tensor_info(d["tensor"], d["dtype"], d.get("shape")) for n, d in sig[side].items()}
              for side in ("inputs", "outputs")}
        mg += _bytes(5, _str(1, key) + _bytes(2, signature_def(ti["inputs"], ti["outputs"], sig["method_name"])))
    return _int(1, 1) + _bytes(2, mg)


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


def _decode_tensor_info(buf: bytes) -> Dict[str, object]:
    info: Dict[str, object] = {"tensor": "", "dtype": "", "shape": []}
    for f, _, x in _fields(buf):
        if f == 1:
            info["tensor"] = bytes(x).decode()
        elif f == 2:
            info["dtype"] = _DTYPE_NAMES.get(x, str(x))
        elif f == 3:
            dims: Optional[List[int]] = []
            for f2, _, d in _fields(x):
                if f2 == 3 and d:
                    dims = None
                elif f2 == 2 and dims is not None:
                    size = dict((ff, xx) for ff, _, xx in _fields(d)).get(1, 0)
                    dims.append(size - (1 << 64) if size >> 63 else size)
            info["shape"] = dims
    return info


def decode_saved_model(buf: bytes) -> Dict[str, object]:
    """{'schema_version', 'tags', 'saver', 'collections', 'node_lists', 'signature_def': {key: {'method_name',
    'inputs': {name: {'tensor', 'dtype', 'shape'}}, 'outputs': {...}}}} of the first (only) meta graph."""
    top = _fields(buf)
    res: Dict[str, object] = {"schema_version": next((v for f, _, v in top if f == 1), 0)}
    mg = next(bytes(v) for f, _, v in top if f == 2)
    # the variable collections, tags and SaverDef share the .meta layout; node_list collections are skipped there
    plain = b"".join(_bytes(f, bytes(v)) for f, w, v in _fields(mg) if w == 2 and f in (1, 3))
    node_lists: Dict[str, List[str]] = {}
    sigs: Dict[str, Dict] = {}
    for f, _, v in _fields(mg):
        if f == 4:
            entry = {ff: xx for ff, _, xx in _fields(v)}
            coll = {ff: xx for ff, _, xx in _fields(entry[2])}
            if 1 in coll:
                node_lists[bytes(entry[1]).decode()] = [bytes(x).decode() for ff, _, x in _fields(coll[1]) if ff == 1]
            else:
                plain += _bytes(4, bytes(v))
        elif f == 5:
            entry = {ff: xx for ff, _, xx in _fields(v)}
            sig: Dict[str, object] = {"inputs": {}, "outputs": {}, "method_name": ""}
            for ff, _, x in _fields(entry[2]):
                if ff in (1, 2):
                    kv = {a: b for a, _, b in _fields(x)}
                    sig["inputs" if ff == 1 else "outputs"][bytes(kv[1]).decode()] = _decode_tensor_info(kv[2])
                elif ff == 3:
                    sig["method_name"] = bytes(x).decode()
            sigs[bytes(entry[1]).decode()] = sig
    res.update(decode_meta_graph(plain))
    res["node_lists"] = node_lists
    res["signature_def"] = sigs
    return res
