"""Model server: REST contract of both signatures, dynamic batching, version hot-swap, error paths (CPU)."""
import base64
import glob
import json
import os
import threading
import time
import urllib.error
import urllib.request

import numpy as np
import pytest
import torch

from distributed_tensorflow_amd.io.export import Servable, example_from_jpeg, export_servable
from distributed_tensorflow_amd.models import build_model
from distributed_tensorflow_amd.serving import DynamicBatcher, ModelManager, ModelServer, QueueFull, list_versions
from test_data import _make_dataset

CLASSES = ["daisy", "roses", "tulips"]


def _export(base, version, seed):
    torch.manual_seed(seed)
    model = build_model("flower_cnn", num_classes=len(CLASSES), device=torch.device("cpu"))
    return export_servable(model, CLASSES, str(base), version, model_name="flower_cnn", image_size=32)


def _call(port, path, body=None):
    req = urllib.request.Request("http://127.0.0.1:%d%s" % (port, path),
                                 data=None if body is None else json.dumps(body).encode())
    try:
        with urllib.request.urlopen(req, timeout=60) as r:
            data = r.read()
            return r.status, (json.loads(data) if r.headers["Content-Type"].startswith("application/json") else data.decode())
    except urllib.error.HTTPError as e:
        return e.code, json.loads(e.read())


def _b64(b):
    return {"b64": base64.b64encode(b).decode()}


def _same(a, b):
    """Predictions agree up to the float rounding that depends on the batch an image rode in."""
    return ([p["classes"] for p in a] == [p["classes"] for p in b]
            and np.allclose([p["scores"] for p in a], [p["scores"] for p in b], atol=1e-5))


@pytest.fixture()
def served(tmp_path):
    data = _make_dataset(str(tmp_path / "flowers"), n_per_class=4)
    jpegs = [open(f, "rb").read() for f in sorted(glob.glob(os.path.join(data, "*", "*.jpg")))[:6]]
    base = tmp_path / "models"
    _export(base, 1, seed=1)
    manager = ModelManager(str(base), "flowers", device="cpu", poll_seconds=0).start()
    server = ModelServer(manager, port=0, max_batch=8, batch_timeout_ms=20.0).start()
    yield server, manager, base, jpegs
    server.stop()


def test_batcher_merges_requests_and_splits_results():
    calls = []

    def fn(items):
        calls.append(len(items))
        time.sleep(0.01)
        return {"y": np.array([[i * 2] for i in items]), "z": np.array([i for i in items])}

    b = DynamicBatcher(fn, max_batch=8, timeout_ms=200.0, max_queue=64)
    futs = [b.submit([i, i + 100]) for i in range(4)]  # 8 items: fills one batch without waiting for the timeout
    t0 = time.monotonic()
    outs = [f.result(timeout=5) for f in futs]
    assert time.monotonic() - t0 < 0.19
    assert calls == [8]
    for i, o in enumerate(outs):
        assert o["y"].tolist() == [[2 * i], [2 * (i + 100)]] and o["z"].tolist() == [i, i + 100]
    # a lone request is released by the timeout; an oversized one is chunked
    assert b([7])["z"].tolist() == [7]
    assert b(list(range(20)))["z"].tolist() == list(range(20))
    assert calls[1:] == [1, 8, 8, 4]
    st = b.stats()
    assert st["items"] == 29 and st["batches"] == 5 and st["requests"] == 6
    # overload: requests beyond the queue bound are rejected, not queued
    gate = threading.Event()
    slow = DynamicBatcher(lambda items: (gate.wait(5), {"z": np.array(items)})[1], max_batch=4, timeout_ms=0.0, max_queue=4)
    first = slow.submit([1])
    time.sleep(0.05)          # the worker is now blocked inside fn
    held = slow.submit([2, 3, 4, 5])
    with pytest.raises(QueueFull):
        slow.submit([6])
    gate.set()
    assert first.result(5)["z"].tolist() == [1] and held.result(5)["z"].tolist() == [2, 3, 4, 5]
    # an exception inside fn reaches every caller of the batch
    bad = DynamicBatcher(lambda items: 1 / 0, max_batch=4, timeout_ms=1.0)
    with pytest.raises(ZeroDivisionError):
        bad([1])
    for x in (b, slow, bad):
        x.close()


def test_rest_signatures_match_the_servable(served):
    server, manager, base, jpegs = served
    ref = Servable(str(base / "1"), device="cpu").predict_images(jpegs)
    port = server.port
    code, status = _call(port, "/v1/models/flowers")
    assert code == 200 and status["model_version_status"][0] == {
        "version": "1", "state": "AVAILABLE", "status": {"error_code": "OK", "error_message": ""}}
    code, meta = _call(port, "/v1/models/flowers/metadata")
    sigs = meta["metadata"]["signature_def"]["signature_def"]
    assert sigs["serving_default"]["method_name"] == "tensorflow/serving/classify"
    assert sigs["predict_images"]["method_name"] == "tensorflow/serving/predict" and meta["metadata"]["classes"] == CLASSES
    assert sigs["predict_images"]["inputs"]["images"] == {
        "name": "ParseExample/ParseExample:0", "dtype": "DT_STRING",
        "tensor_shape": {"dim": [{"size": "-1", "name": ""}], "unknown_rank": False}}
    assert sigs["serving_default"]["inputs"]["inputs"]["tensor_shape"] == {"dim": [], "unknown_rank": True}
    assert sigs["serving_default"]["outputs"]["scores"]["dtype"] == "DT_FLOAT"
    # predict, row format (bare values and named values) and columnar format
    code, row = _call(port, "/v1/models/flowers:predict", {"instances": [_b64(j) for j in jpegs]})
    assert code == 200 and len(row["predictions"]) == 6
    assert [p["classes"] for p in row["predictions"]] == ref["classes"].tolist()
    assert np.allclose([p["scores"] for p in row["predictions"]], ref["scores"], atol=1e-5)
    code, named = _call(port, "/v1/models/flowers/versions/1:predict",
                        {"signature_name": "predict_images", "instances": [{"images": _b64(j)} for j in jpegs[:2]]})
    assert code == 200 and _same(named["predictions"], row["predictions"][:2])
    code, col = _call(port, "/v1/models/flowers:predict", {"inputs": {"images": [_b64(j) for j in jpegs[:3]]}})
    assert code == 200 and col["outputs"]["classes"] == ref["classes"].tolist()[:3]
    # predict through the classify signature's input: serialized tf.Example
    code, viaex = _call(port, "/v1/models/flowers:predict",
                        {"signature_name": "serving_default", "instances": [_b64(example_from_jpeg(jpegs[0]))]})
    assert code == 200 and _same(viaex["predictions"], row["predictions"][:1])
    # classify
    code, cls = _call(port, "/v1/models/flowers:classify", {"examples": [{"image/encoded": _b64(j)} for j in jpegs]})
    assert code == 200 and [[r[0][0]] for r in cls["results"]] == ref["classes"].tolist()
    assert np.allclose([[r[0][1]] for r in cls["results"]], ref["scores"], atol=1e-5)
    # errors
    assert _call(port, "/v1/models/other:predict", {"instances": []})[0] == 404
    assert _call(port, "/v1/models/flowers/versions/9:predict", {"instances": [_b64(jpegs[0])]})[0] == 404
    assert _call(port, "/v1/models/flowers:predict", {"instances": ["not-binary"]})[0] == 400
    assert _call(port, "/v1/models/flowers:predict", {"instances": [_b64(b"not a jpeg")]})[0] == 400
    assert _call(port, "/v1/models/flowers:predict", {})[0] == 400
    assert _call(port, "/v1/models/flowers:classify", {"examples": [{"other": 1}]})[0] == 400
    code, text = _call(port, "/metrics")
    assert code == 200 and 'dtfa_http_requests_total{route="predict",code="200"} 4' in text and "dtfa_model_version 1" in text
    assert _call(port, "/healthz")[0] == 200


def test_concurrent_requests_share_batches(served):
    server, manager, base, jpegs = served
    results = [None] * 12

    def client(i):
        results[i] = _call(server.port, "/v1/models/flowers:predict", {"instances": [_b64(jpegs[i % 6])]})

    threads = [threading.Thread(target=client, args=(i,)) for i in range(12)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert all(r[0] == 200 for r in results)
    for i in range(6):  # same image -> same answer, whatever batch it rode in
        assert _same(results[i][1]["predictions"], results[i + 6][1]["predictions"])
    st = server.batcher.stats()
    assert st["items"] == 12 and st["batches"] < 12  # merged


def test_new_version_is_picked_up_and_broken_one_is_skipped(served):
    server, manager, base, jpegs = served
    req = {"instances": [_b64(j) for j in jpegs]}
    before = _call(server.port, "/v1/models/flowers:predict", req)[1]
    # an incomplete version directory (no manifest yet) is not a version
    os.makedirs(base / "2" / "variables")
    assert list_versions(str(base)) == [1] and not manager.refresh()
    # a broken export is reported and skipped
    os.makedirs(base / "3")
    with open(base / "3" / "saved_model.json", "w") as f:
        f.write("{}")
    assert not manager.refresh() and manager.version == 1
    assert any(s["version"] == "3" and s["state"] == "END" for s in manager.status())
    # a good newer version replaces the served one
    _export(base, 4, seed=2)
    assert manager.refresh() and manager.version == 4
    after = _call(server.port, "/v1/models/flowers:predict", req)[1]
    ref = Servable(str(base / "4"), device="cpu").predict_images(jpegs)
    assert np.allclose([p["scores"] for p in after["predictions"]], ref["scores"], atol=1e-5)
    assert not np.allclose([p["scores"] for p in before["predictions"]], [p["scores"] for p in after["predictions"]])
    assert _call(server.port, "/v1/models/flowers/versions/1:predict", req)[0] == 404


def test_serve_model_entrypoint(tmp_path):
    import re
    import subprocess
    import sys

    data = _make_dataset(str(tmp_path / "flowers"), n_per_class=2)
    jpeg = open(sorted(glob.glob(os.path.join(data, "*", "*.jpg")))[0], "rb").read()
    _export(tmp_path / "models", 7, seed=3)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.Popen([sys.executable, "-u", "serve_model.py", "--model_base_path=%s" % (tmp_path / "models"),
                             "--model_name=flowers", "--rest_api_port=0", "--port=0", "--device=cpu"],
                            cwd=os.path.join(root, "recognition_flower", "distributed"), stdout=subprocess.PIPE,
                            stderr=subprocess.STDOUT, text=True)
    try:
        line = proc.stdout.readline()
        m = re.search(r"version 7 on http://127\.0\.0\.1:(\d+) and grpc://127\.0\.0\.1:(\d+)", line)
        assert m, line
        port = int(m.group(1))
        code, out = _call(port, "/v1/models/flowers:classify", {"examples": [{"image/encoded": _b64(jpeg)}]})
        assert code == 200 and out["results"][0][0][0] in CLASSES
    finally:
        proc.terminate()
        proc.wait(timeout=20)


def test_grpc_prediction_service(served):
    """Predict / Classify / GetModelMetadata / GetModelStatus over gRPC agree with the servable and with the REST front end."""
    import grpc

    from distributed_tensorflow_amd.serving import GrpcModelServer
    from distributed_tensorflow_amd.serving import tfs_proto as P

    server, manager, base, jpegs = served
    ref = Servable(str(base / "1"), device="cpu").predict_images(jpegs)
    g = GrpcModelServer(manager, server.batcher, port=0).start()
    try:
        ch = grpc.insecure_channel("127.0.0.1:%d" % g.port)
        predict = ch.unary_unary("/tensorflow.serving.PredictionService/Predict",
                                 request_serializer=P.PredictRequest.SerializeToString,
                                 response_deserializer=P.PredictResponse.FromString)
        classify = ch.unary_unary("/tensorflow.serving.PredictionService/Classify",
                                  request_serializer=P.ClassificationRequest.SerializeToString,
                                  response_deserializer=P.ClassificationResponse.FromString)
        status = ch.unary_unary("/tensorflow.serving.ModelService/GetModelStatus",
                                request_serializer=P.GetModelStatusRequest.SerializeToString,
                                response_deserializer=P.GetModelStatusResponse.FromString)
        # Predict, predict_images signature
        req = P.PredictRequest()
        req.model_spec.name, req.model_spec.signature_name = "flowers", "predict_images"
        req.inputs["images"].CopyFrom(P.string_tensor(jpegs))
        resp = predict(req, timeout=60)
        assert resp.model_spec.name == "flowers" and resp.model_spec.version.value == 1
        assert set(resp.outputs) == {"classes", "scores"}
        assert [d.size for d in resp.outputs["classes"].tensor_shape.dim] == [6, 1] and resp.outputs["classes"].dtype == P.DT_STRING
        assert [[v.decode()] for v in resp.outputs["classes"].string_val] == ref["classes"].tolist()
        assert resp.outputs["scores"].dtype == P.DT_FLOAT
        assert np.allclose(np.array(resp.outputs["scores"].float_val).reshape(6, 1), ref["scores"], atol=1e-5)
        # Predict through the classify signature's input (serialized tf.Example), with an output filter
        req2 = P.PredictRequest()
        req2.model_spec.name, req2.model_spec.signature_name = "flowers", "serving_default"
        req2.inputs["inputs"].CopyFrom(P.string_tensor([example_from_jpeg(j) for j in jpegs[:2]]))
        req2.output_filter.append("scores")
        resp2 = predict(req2, timeout=60)
        assert set(resp2.outputs) == {"scores"}
        assert np.allclose(np.array(resp2.outputs["scores"].float_val).reshape(2, 1), ref["scores"][:2], atol=1e-5)
        # Classify
        creq = P.ClassificationRequest()
        creq.model_spec.name = "flowers"
        for j in jpegs:
            creq.input.example_list.examples.add().features.feature["image/encoded"].bytes_list.value.append(j)
        cresp = classify(creq, timeout=60)
        got = [(c.classes[0].label, c.classes[0].score) for c in cresp.result.classifications]
        assert [[l] for l, _ in got] == ref["classes"].tolist()
        assert np.allclose([[s] for _, s in got], ref["scores"], atol=1e-5)
        # the same answer as the REST front end
        rest = _call(server.port, "/v1/models/flowers:classify", {"examples": [{"image/encoded": _b64(j)} for j in jpegs]})[1]
        assert [r[0][0] for r in rest["results"]] == [l for l, _ in got]
        # status
        sreq = P.GetModelStatusRequest()
        sreq.model_spec.name = "flowers"
        st = status(sreq, timeout=30)
        assert [(m.version, m.state) for m in st.model_version_status] == [(1, 30)]
        # metadata: the SignatureDefMap of saved_model.pb, packed in an Any
        metadata = ch.unary_unary("/tensorflow.serving.PredictionService/GetModelMetadata",
                                  request_serializer=P.GetModelMetadataRequest.SerializeToString,
                                  response_deserializer=P.GetModelMetadataResponse.FromString)
        mreq = P.GetModelMetadataRequest()
        mreq.model_spec.name = "flowers"
        mreq.metadata_field.append("signature_def")
        mresp = metadata(mreq, timeout=30)
        assert mresp.model_spec.version.value == 1
        assert mresp.metadata["signature_def"].type_url == "type.googleapis.com/tensorflow.serving.SignatureDefMap"
        sdm = P.SignatureDefMap()
        assert mresp.metadata["signature_def"].Unpack(sdm) and set(sdm.signature_def) == {"serving_default", "predict_images"}
        sd = sdm.signature_def["serving_default"]
        assert sd.method_name == "tensorflow/serving/classify" and sd.inputs["inputs"].name == "tf_example:0"
        assert sd.inputs["inputs"].dtype == P.DT_STRING and sd.inputs["inputs"].tensor_shape.unknown_rank
        assert [d.size for d in sdm.signature_def["predict_images"].outputs["scores"].tensor_shape.dim] == [-1, 1]
        # the message built by the protobuf runtime and the hand-encoded saved_model.pb agree byte for byte per signature
        from distributed_tensorflow_amd.io.metagraph import signature_def, tensor_info
        from distributed_tensorflow_amd.io.export import signature_manifest
        for key, sig in signature_manifest().items():
            ti = {side: {n: tensor_info(d["tensor"], d["dtype"], d.get("shape")) for n, d in sig[side].items()}
                  for side in ("inputs", "outputs")}
            assert sdm.signature_def[key].SerializeToString(deterministic=True) == \
                signature_def(ti["inputs"], ti["outputs"], sig["method_name"])
        with pytest.raises(grpc.RpcError) as e:
            metadata(P.GetModelMetadataRequest(), timeout=30)
        assert e.value.code() == grpc.StatusCode.INVALID_ARGUMENT
        # errors carry gRPC status codes
        bad = P.PredictRequest()
        bad.model_spec.name = "other"
        bad.inputs["images"].CopyFrom(P.string_tensor(jpegs[:1]))
        with pytest.raises(grpc.RpcError) as e:
            predict(bad, timeout=30)
        assert e.value.code() == grpc.StatusCode.NOT_FOUND
        req.model_spec.version.value = 9
        with pytest.raises(grpc.RpcError) as e:
            predict(req, timeout=30)
        assert e.value.code() == grpc.StatusCode.NOT_FOUND
        junk = P.PredictRequest()
        junk.inputs["images"].CopyFrom(P.string_tensor([b"not a jpeg"]))
        with pytest.raises(grpc.RpcError) as e:
            predict(junk, timeout=30)
        assert e.value.code() == grpc.StatusCode.INVALID_ARGUMENT
        ch.close()
    finally:
        g.stop()


def test_tfs_proto_wire_bytes():
    """The declared messages serialise to the wire bytes the protobuf encoding rules give for the published
    field numbers (hand-assembled below), and tf.Example matches the native encoder used for TFRecords."""
    from distributed_tensorflow_amd.io.tfrecord import make_example
    from distributed_tensorflow_amd.serving import tfs_proto as P

    r = P.PredictRequest()
    r.model_spec.name, r.model_spec.signature_name = "m", "s"
    r.model_spec.version.value = 3
    r.inputs["images"].CopyFrom(P.string_tensor([b"ab"]))
    want = bytes.fromhex(
        "0a0a" "0a016d" "12020803" "1a0173"                    # 1: ModelSpec{1:"m", 2:Int64Value{1:3}, 3:"s"}
        "1216" "0a06696d61676573" "120c" "0807" "12041202" "0801" "42026162")  # 2: {"images": TensorProto{...}}
    assert r.SerializeToString() == want
    t = P.float_tensor([1.0, 2.0], [2, 1])
    assert t.SerializeToString() == bytes.fromhex("0801" "1208" "12020802" "12020801" "2a08" "0000803f" "00000040")
    c = P.ClassificationResponse()
    one = c.result.classifications.add().classes.add()
    one.label, one.score = "x", 1.0
    assert c.SerializeToString() == bytes.fromhex("0a0c" "0a0a" "0a08" "0a0178" "150000803f")
    ex = P.Example()
    ex.features.feature["image/encoded"].bytes_list.value.append(b"JPEG")
    assert ex.SerializeToString() == make_example({"image/encoded": b"JPEG"})
