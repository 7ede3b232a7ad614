"""Protocol-buffer messages of the model-server gRPC API, built at import time.

Clients of a served SavedModel (the reference exports one with a classify and a
predict signature, recognition_flower/distributed/save_model.py:126-171) talk
``tensorflow.serving.PredictionService`` / ``ModelService`` over gRPC.  No
generated ``*_pb2`` modules exist in this environment, so the message types are
declared here as descriptors -- message names, field names, field numbers and
types as in the public ``.proto`` files of TensorFlow (tensor, tensor_shape,
types, example, feature, meta_graph) and TensorFlow Serving (model, predict, input,
classification, get_model_status, get_model_metadata) -- and turned into classes by the protobuf
runtime.  Only the fields a served image classifier uses are declared; unknown
fields sent by a client are preserved and ignored, as protobuf prescribes.

Parity note: the field numbers were written from the published schemas without
the generated stubs at hand to diff against; the tests pin the wire bytes the
declarations produce.
"""
from __future__ import annotations

from google.protobuf import any_pb2, descriptor_pb2, descriptor_pool, message_factory, wrappers_pb2  # noqa: F401

_F = descriptor_pb2.FieldDescriptorProto
_T = {"int32": _F.TYPE_INT32, "int64": _F.TYPE_INT64, "float": _F.TYPE_FLOAT, "double": _F.TYPE_DOUBLE,
      "bool": _F.TYPE_BOOL, "string": _F.TYPE_STRING, "bytes": _F.TYPE_BYTES}


def _field(msg, name, number, typ, repeated=False, packed=None, oneof=None):
    f = msg.field.add()
    f.name, f.number = name, number
    f.label = _F.LABEL_REPEATED if repeated else _F.LABEL_OPTIONAL
    if typ in _T:
        f.type = _T[typ]
    elif typ.startswith("enum:"):
        f.type, f.type_name = _F.TYPE_ENUM, typ[5:]
    else:
        f.type, f.type_name = _F.TYPE_MESSAGE, typ
    if packed is not None:
        f.options.packed = packed
    if oneof is not None:
        f.oneof_index = oneof
    return f


def _map_field(msg, name, number, value_type, parent_full_name):
    """map<string, value_type> name = number;  (a repeated nested *Entry message with map_entry set)"""
    entry = msg.nested_type.add()
    entry.name = "".join(p.capitalize() for p in name.split("_")) + "Entry"
    entry.options.map_entry = True
    _field(entry, "key", 1, "string")
    _field(entry, "value", 2, value_type)
    _field(msg, name, number, "%s.%s" % (parent_full_name, entry.name), repeated=True)


def _build():
    pool = descriptor_pool.Default()
    # ---------------------------------------------------------------- tensorflow core
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name, fd.package, fd.syntax = "dtfa/tensorflow_core.proto", "tensorflow", "proto3"
    dt = fd.enum_type.add()
    dt.name = "DataType"
    for n, v in (("DT_INVALID", 0), ("DT_FLOAT", 1), ("DT_DOUBLE", 2), ("DT_INT32", 3), ("DT_UINT8", 4),
                 ("DT_INT16", 5), ("DT_INT8", 6), ("DT_STRING", 7), ("DT_COMPLEX64", 8), ("DT_INT64", 9),
                 ("DT_BOOL", 10)):
        e = dt.value.add()
        e.name, e.number = n, v
    shape = fd.message_type.add()
    shape.name = "TensorShapeProto"
    dim = shape.nested_type.add()
    dim.name = "Dim"
    _field(dim, "size", 1, "int64")
    _field(dim, "name", 2, "string")
    _field(shape, "dim", 2, ".tensorflow.TensorShapeProto.Dim", repeated=True)
    _field(shape, "unknown_rank", 3, "bool")
    tensor = fd.message_type.add()
    tensor.name = "TensorProto"
    _field(tensor, "dtype", 1, "enum:.tensorflow.DataType")
    _field(tensor, "tensor_shape", 2, ".tensorflow.TensorShapeProto")
    _field(tensor, "version_number", 3, "int32")
    _field(tensor, "tensor_content", 4, "bytes")
    _field(tensor, "float_val", 5, "float", repeated=True, packed=True)
    _field(tensor, "double_val", 6, "double", repeated=True, packed=True)
    _field(tensor, "int_val", 7, "int32", repeated=True, packed=True)
    _field(tensor, "string_val", 8, "bytes", repeated=True)
    _field(tensor, "int64_val", 10, "int64", repeated=True, packed=True)
    _field(tensor, "bool_val", 11, "bool", repeated=True, packed=True)
    for name, typ in (("BytesList", "bytes"), ("FloatList", "float"), ("Int64List", "int64")):
        m = fd.message_type.add()
        m.name = name
        _field(m, "value", 1, typ, repeated=True, packed=None if typ == "bytes" else True)
    feat = fd.message_type.add()
    feat.name = "Feature"
    feat.oneof_decl.add().name = "kind"
    _field(feat, "bytes_list", 1, ".tensorflow.BytesList", oneof=0)
    _field(feat, "float_list", 2, ".tensorflow.FloatList", oneof=0)
    _field(feat, "int64_list", 3, ".tensorflow.Int64List", oneof=0)
    feats = fd.message_type.add()
    feats.name = "Features"
    _map_field(feats, "feature", 1, ".tensorflow.Feature", ".tensorflow.Features")
    ex = fd.message_type.add()
    ex.name = "Example"
    _field(ex, "features", 1, ".tensorflow.Features")
    tinfo = fd.message_type.add()
    tinfo.name = "TensorInfo"
    tinfo.oneof_decl.add().name = "encoding"
    _field(tinfo, "name", 1, "string", oneof=0)
    _field(tinfo, "dtype", 2, "enum:.tensorflow.DataType")
    _field(tinfo, "tensor_shape", 3, ".tensorflow.TensorShapeProto")
    sigdef = fd.message_type.add()
    sigdef.name = "SignatureDef"
    _map_field(sigdef, "inputs", 1, ".tensorflow.TensorInfo", ".tensorflow.SignatureDef")
    _map_field(sigdef, "outputs", 2, ".tensorflow.TensorInfo", ".tensorflow.SignatureDef")
    _field(sigdef, "method_name", 3, "string")
    pool.Add(fd)

    # ---------------------------------------------------------------- tensorflow serving
    sd = descriptor_pb2.FileDescriptorProto()
    sd.name, sd.package, sd.syntax = "dtfa/tensorflow_serving.proto", "tensorflow.serving", "proto3"
    sd.dependency.append("dtfa/tensorflow_core.proto")
    sd.dependency.append("google/protobuf/wrappers.proto")
    sd.dependency.append("google/protobuf/any.proto")
    spec = sd.message_type.add()
    spec.name = "ModelSpec"
    spec.oneof_decl.add().name = "version_choice"
    _field(spec, "name", 1, "string")
    _field(spec, "version", 2, ".google.protobuf.Int64Value", oneof=0)
    _field(spec, "signature_name", 3, "string")
    _field(spec, "version_label", 4, "string", oneof=0)
    preq = sd.message_type.add()
    preq.name = "PredictRequest"
    _field(preq, "model_spec", 1, ".tensorflow.serving.ModelSpec")
    _map_field(preq, "inputs", 2, ".tensorflow.TensorProto", ".tensorflow.serving.PredictRequest")
    _field(preq, "output_filter", 3, "string", repeated=True)
    pres = sd.message_type.add()
    pres.name = "PredictResponse"
    _map_field(pres, "outputs", 1, ".tensorflow.TensorProto", ".tensorflow.serving.PredictResponse")
    _field(pres, "model_spec", 2, ".tensorflow.serving.ModelSpec")
    exl = sd.message_type.add()
    exl.name = "ExampleList"
    _field(exl, "examples", 1, ".tensorflow.Example", repeated=True)
    exc = sd.message_type.add()
    exc.name = "ExampleListWithContext"
    _field(exc, "examples", 1, ".tensorflow.Example", repeated=True)
    _field(exc, "context", 2, ".tensorflow.Example")
    inp = sd.message_type.add()
    inp.name = "Input"
    inp.oneof_decl.add().name = "kind"
    _field(inp, "example_list", 1, ".tensorflow.serving.ExampleList", oneof=0)
    _field(inp, "example_list_with_context", 2, ".tensorflow.serving.ExampleListWithContext", oneof=0)
    cls = sd.message_type.add()
    cls.name = "Class"
    _field(cls, "label", 1, "string")
    _field(cls, "score", 2, "float")
    clss = sd.message_type.add()
    clss.name = "Classifications"
    _field(clss, "classes", 1, ".tensorflow.serving.Class", repeated=True)
    cres = sd.message_type.add()
    cres.name = "ClassificationResult"
    _field(cres, "classifications", 1, ".tensorflow.serving.Classifications", repeated=True)
    creq = sd.message_type.add()
    creq.name = "ClassificationRequest"
    _field(creq, "model_spec", 1, ".tensorflow.serving.ModelSpec")
    _field(creq, "input", 2, ".tensorflow.serving.Input")
    cresp = sd.message_type.add()
    cresp.name = "ClassificationResponse"
    _field(cresp, "result", 1, ".tensorflow.serving.ClassificationResult")
    _field(cresp, "model_spec", 2, ".tensorflow.serving.ModelSpec")
    # model status
    stp = sd.message_type.add()
    stp.name = "StatusProto"
    _field(stp, "error_code", 1, "int32")
    _field(stp, "error_message", 2, "string")
    mvs = sd.message_type.add()
    mvs.name = "ModelVersionStatus"
    st_enum = mvs.enum_type.add()
    st_enum.name = "State"
    for n, v in (("UNKNOWN", 0), ("START", 10), ("LOADING", 20), ("AVAILABLE", 30), ("UNLOADING", 40), ("END", 50)):
        e = st_enum.value.add()
        e.name, e.number = n, v
    _field(mvs, "version", 1, "int64")
    _field(mvs, "state", 2, "enum:.tensorflow.serving.ModelVersionStatus.State")
    _field(mvs, "status", 3, ".tensorflow.serving.StatusProto")
    gsr = sd.message_type.add()
    gsr.name = "GetModelStatusRequest"
    _field(gsr, "model_spec", 1, ".tensorflow.serving.ModelSpec")
    gsp = sd.message_type.add()
    gsp.name = "GetModelStatusResponse"
    _field(gsp, "model_version_status", 1, ".tensorflow.serving.ModelVersionStatus", repeated=True)
    # model metadata
    sdm = sd.message_type.add()
    sdm.name = "SignatureDefMap"
    _map_field(sdm, "signature_def", 1, ".tensorflow.SignatureDef", ".tensorflow.serving.SignatureDefMap")
    gmr = sd.message_type.add()
    gmr.name = "GetModelMetadataRequest"
    _field(gmr, "model_spec", 1, ".tensorflow.serving.ModelSpec")
    _field(gmr, "metadata_field", 2, "string", repeated=True)
    gmp = sd.message_type.add()
    gmp.name = "GetModelMetadataResponse"
    _field(gmp, "model_spec", 1, ".tensorflow.serving.ModelSpec")
    _map_field(gmp, "metadata", 2, ".google.protobuf.Any", ".tensorflow.serving.GetModelMetadataResponse")
    pool.Add(sd)

    def cls_of(full_name):
        return message_factory.GetMessageClass(pool.FindMessageTypeByName(full_name))

    names = {"TensorProto": "tensorflow.TensorProto", "TensorShapeProto": "tensorflow.TensorShapeProto",
             "Example": "tensorflow.Example", "Features": "tensorflow.Features", "Feature": "tensorflow.Feature",
             "TensorInfo": "tensorflow.TensorInfo", "SignatureDef": "tensorflow.SignatureDef"}
    for n in ("ModelSpec", "PredictRequest", "PredictResponse", "Input", "ExampleList", "ClassificationRequest",
              "ClassificationResponse", "ClassificationResult", "Classifications", "Class", "GetModelStatusRequest",
              "GetModelStatusResponse", "ModelVersionStatus", "StatusProto", "SignatureDefMap", "GetModelMetadataRequest",
              "GetModelMetadataResponse"):
        names[n] = "tensorflow.serving." + n
    return {k: cls_of(v) for k, v in names.items()}


_classes = _build()
globals().update(_classes)
__all__ = sorted(_classes) + ["DT_FLOAT", "DT_STRING", "DT_INT64", "string_tensor", "float_tensor", "tensor_strings",
                               "signature_def_map"]

DT_FLOAT, DT_STRING, DT_INT64 = 1, 7, 9


def string_tensor(values, shape=None):
    """TensorProto of dtype DT_STRING holding ``values`` (bytes)."""
    t = _classes["TensorProto"]()
    t.dtype = DT_STRING
    for d in (shape if shape is not None else [len(values)]):
        t.tensor_shape.dim.add().size = int(d)
    t.string_val.extend(bytes(v) for v in values)
    return t


def float_tensor(values, shape):
    t = _classes["TensorProto"]()
    t.dtype = DT_FLOAT
    for d in shape:
        t.tensor_shape.dim.add().size = int(d)
    t.float_val.extend(float(v) for v in values)
    return t


def tensor_strings(t):
    """The byte strings of a DT_STRING TensorProto."""
    if t.dtype != DT_STRING:
        raise ValueError("expected a DT_STRING tensor, got dtype %d" % t.dtype)
    return [bytes(v) for v in t.string_val]


def signature_def_map(sigs):
    """``SignatureDefMap`` from exported signatures ({key: {"method_name", "inputs" / "outputs": {name: {"tensor",
    "dtype": "string" | "float", "shape": [..] | None}}}}, the layout of ``Servable.signature_def``)."""
    out = _classes["SignatureDefMap"]()
    for key, sig in sigs.items():
        sd = out.signature_def[key]
        sd.method_name = sig["method_name"]
        for side, dst in (("inputs", sd.inputs), ("outputs", sd.outputs)):
            for name, d in sig[side].items():
                ti = dst[name]
                ti.name = d["tensor"]
                ti.dtype = {"string": DT_STRING, "float": DT_FLOAT}[d["dtype"]]
                if d.get("shape") is None:
                    ti.tensor_shape.unknown_rank = True
                else:
                    for x in d["shape"]:
                        ti.tensor_shape.dim.add().size = int(x)
    return out
