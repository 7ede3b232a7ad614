"""gRPC front end of the model server: ``tensorflow.serving.PredictionService`` and ``ModelService``.

The SavedModel the reference exports (signatures ``serving_default`` = classify
over ``tf.Example`` and ``predict_images`` = predict over JPEG strings,
recognition_flower/single/flower_train.py:205-251) is consumed by clients through
these RPCs:

``/tensorflow.serving.PredictionService/Predict``
    ``inputs["images"]`` (signature ``predict_images``) or ``inputs["inputs"]`` (``serving_default``, serialized
    ``tf.Example``), DT_STRING tensors -> ``outputs["classes"]`` DT_STRING [N, 1], ``outputs["scores"]`` DT_FLOAT
    [N, 1]; ``output_filter`` restricts the outputs.
``/tensorflow.serving.PredictionService/Classify``
    ``input.example_list`` (or ``example_list_with_context``) -> one ``Classifications`` per example with the
    top-1 ``Class{label, score}``.
``/tensorflow.serving.PredictionService/GetModelMetadata``
    ``metadata_field = ["signature_def"]`` -> ``metadata["signature_def"]`` = ``Any`` wrapping the
    ``SignatureDefMap`` read from the export's ``saved_model.pb``.
``/tensorflow.serving.ModelService/GetModelStatus``
    the versions the manager knows and their state.

Requests go through the same :class:`DynamicBatcher` as the REST front end, so
gRPC and REST clients share GPU batches.  Handlers are registered generically
(no generated service stubs); messages come from :mod:`tfs_proto`.
"""
from __future__ import annotations

from concurrent import futures
from typing import List, Optional

import grpc

from . import tfs_proto as P
from .batcher import QueueFull
from .manager import ModelManager

PREDICTION_SERVICE = "tensorflow.serving.PredictionService"
MODEL_SERVICE = "tensorflow.serving.ModelService"


class GrpcModelServer:
    def __init__(self, manager: ModelManager, batcher, host: str = "127.0.0.1", port: int = 8500,
                 max_workers: int = 64, max_message_mb: int = 64):
        # (handler threads only parse a request and wait for the batcher: their number bounds how many requests
        # can be in flight towards one GPU batch, so it is at least the batcher's maximum batch)
        self.manager = manager
        self.batcher = batcher
        self.calls = {"Predict": 0, "Classify": 0, "GetModelMetadata": 0, "GetModelStatus": 0}
        opts = [("grpc.max_receive_message_length", max_message_mb << 20),
                ("grpc.max_send_message_length", max_message_mb << 20)]
        self._server = grpc.server(futures.ThreadPoolExecutor(max_workers=max_workers), options=opts)
        self._server.add_generic_rpc_handlers((
            grpc.method_handlers_generic_handler(PREDICTION_SERVICE, {
                "Predict": grpc.unary_unary_rpc_method_handler(
                    self._predict, request_deserializer=P.PredictRequest.FromString,
                    response_serializer=P.PredictResponse.SerializeToString),
                "Classify": grpc.unary_unary_rpc_method_handler(
                    self._classify, request_deserializer=P.ClassificationRequest.FromString,
                    response_serializer=P.ClassificationResponse.SerializeToString),
                "GetModelMetadata": grpc.unary_unary_rpc_method_handler(
                    self._metadata, request_deserializer=P.GetModelMetadataRequest.FromString,
                    response_serializer=P.GetModelMetadataResponse.SerializeToString),
            }),
            grpc.method_handlers_generic_handler(MODEL_SERVICE, {
                "GetModelStatus": grpc.unary_unary_rpc_method_handler(
                    self._status, request_deserializer=P.GetModelStatusRequest.FromString,
                    response_serializer=P.GetModelStatusResponse.SerializeToString),
            }),
        ))
        self.port = self._server.add_insecure_port("%s:%d" % (host, port))
        if self.port == 0:
            raise OSError("cannot bind gRPC port %s:%d" % (host, port))

    # ------------------------------------------------------------------ helpers
    def _check_spec(self, spec, context) -> Optional[int]:
        if spec.name and spec.name != self.manager.name:
            context.abort(grpc.StatusCode.NOT_FOUND, "model %r is not served (serving %r)" % (spec.name, self.manager.name))
        version = spec.version.value if spec.HasField("version") else None
        try:
            self.manager.get(version)
        except LookupError as e:
            context.abort(grpc.StatusCode.NOT_FOUND, str(e))
        return version

    def _infer(self, jpegs: List[bytes], context):
        try:
            return self.batcher.submit(jpegs).result()
        except QueueFull as e:
            context.abort(grpc.StatusCode.RESOURCE_EXHAUSTED, "server overloaded: %s" % e)
        except LookupError as e:
            context.abort(grpc.StatusCode.NOT_FOUND, str(e))
        except Exception as e:  # noqa: BLE001 -- e.g. a corrupt JPEG in the batch
            context.abort(grpc.StatusCode.INVALID_ARGUMENT, "%s: %s" % (type(e).__name__, e))

    def _reply_spec(self, msg, signature: str) -> None:
        msg.model_spec.name = self.manager.name
        msg.model_spec.signature_name = signature
        if self.manager.version is not None:
            msg.model_spec.version.value = int(self.manager.version)

    @staticmethod
    def _jpeg_of_example(ex, context_ex, context) -> bytes:
        feat = ex.features.feature.get("image/encoded")
        if feat is None and context_ex is not None:
            feat = context_ex.features.feature.get("image/encoded")
        if feat is None or len(feat.bytes_list.value) != 1:
            context.abort(grpc.StatusCode.INVALID_ARGUMENT, "each example needs one image/encoded bytes value")
        return bytes(feat.bytes_list.value[0])

    # ------------------------------------------------------------------ RPCs
    def _predict(self, req, context):
        self.calls["Predict"] += 1
        self._check_spec(req.model_spec, context)
        sig = req.model_spec.signature_name or "predict_images"
        if sig not in ("predict_images", "serving_default"):
            context.abort(grpc.StatusCode.INVALID_ARGUMENT, "unknown signature %r" % sig)
        name = "images" if sig == "predict_images" else "inputs"
        if set(req.inputs) != {name}:
            context.abort(grpc.StatusCode.INVALID_ARGUMENT, "signature %s takes the single input %r" % (sig, name))
        try:
            blobs = P.tensor_strings(req.inputs[name])
        except ValueError as e:
            context.abort(grpc.StatusCode.INVALID_ARGUMENT, str(e))
        if sig == "serving_default":
            jpegs = []
            for b in blobs:
                try:
                    ex = P.Example.FromString(b)
                except Exception:  # noqa: BLE001
                    context.abort(grpc.StatusCode.INVALID_ARGUMENT, "inputs holds a malformed tf.Example")
                jpegs.append(self._jpeg_of_example(ex, None, context))
        else:
            jpegs = blobs
        out = self._infer(jpegs, context)
        resp = P.PredictResponse()
        self._reply_spec(resp, sig)
        n = len(jpegs)
        want = set(req.output_filter) or {"classes", "scores"}
        unknown = want - {"classes", "scores", "probabilities"}
        if unknown:
            context.abort(grpc.StatusCode.INVALID_ARGUMENT, "unknown outputs %s" % sorted(unknown))
        if "classes" in want:
            resp.outputs["classes"].CopyFrom(P.string_tensor([str(c[0]).encode() for c in out["classes"]], [n, 1]))
        if "scores" in want:
            resp.outputs["scores"].CopyFrom(P.float_tensor(out["scores"].reshape(-1), [n, 1]))
        if "probabilities" in want:
            probs = out["probabilities"]
            resp.outputs["probabilities"].CopyFrom(P.float_tensor(probs.reshape(-1), [n, probs.shape[1] if n else 0]))
        return resp

    def _classify(self, req, context):
        self.calls["Classify"] += 1
        self._check_spec(req.model_spec, context)
        sig = req.model_spec.signature_name or "serving_default"
        if sig != "serving_default":
            context.abort(grpc.StatusCode.INVALID_ARGUMENT, "classify is served by the serving_default signature")
        kind = req.input.WhichOneof("kind")
        if kind == "example_list":
            examples, ctx_ex = req.input.example_list.examples, None
        elif kind == "example_list_with_context":
            examples, ctx_ex = req.input.example_list_with_context.examples, req.input.example_list_with_context.context
        else:
            context.abort(grpc.StatusCode.INVALID_ARGUMENT, "the request carries no examples")
        jpegs = [self._jpeg_of_example(ex, ctx_ex, context) for ex in examples]
        out = self._infer(jpegs, context)
        resp = P.ClassificationResponse()
        self._reply_spec(resp, sig)
        for c, s in zip(out["classes"], out["scores"]):
            one = resp.result.classifications.add().classes.add()
            one.label, one.score = str(c[0]), float(s[0])
        return resp

    def _metadata(self, req, context):
        self.calls["GetModelMetadata"] += 1
        version = self._check_spec(req.model_spec, context)
        fields = list(req.metadata_field)
        if not fields:
            context.abort(grpc.StatusCode.INVALID_ARGUMENT, "GetModelMetadataRequest must specify at least one metadata_field")
        if set(fields) != {"signature_def"}:
            context.abort(grpc.StatusCode.INVALID_ARGUMENT, "metadata field %r is not supported" % sorted(set(fields) - {"signature_def"})[0])
        servable = self.manager.get(version)
        resp = P.GetModelMetadataResponse()
        self._reply_spec(resp, "")
        if version is not None:
            resp.model_spec.version.value = int(version)
        resp.metadata["signature_def"].Pack(P.signature_def_map(servable.signature_def))
        return resp

    def _status(self, req, context):
        self.calls["GetModelStatus"] += 1
        if req.model_spec.name and req.model_spec.name != self.manager.name:
            context.abort(grpc.StatusCode.NOT_FOUND, "model %r is not served" % req.model_spec.name)
        resp = P.GetModelStatusResponse()
        want = req.model_spec.version.value if req.model_spec.HasField("version") else None
        for st in self.manager.status():
            if want is not None and int(st["version"]) != want:
                continue
            m = resp.model_version_status.add()
            m.version = int(st["version"])
            m.state = 30 if st["state"] == "AVAILABLE" else 50
            m.status.error_code = 0 if st["status"]["error_code"] == "OK" else 2
            m.status.error_message = st["status"]["error_message"]
        if want is not None and not resp.model_version_status:
            context.abort(grpc.StatusCode.NOT_FOUND, "version %d is unknown" % want)
        return resp

    # ------------------------------------------------------------------ lifecycle
    def start(self) -> "GrpcModelServer":
        self._server.start()
        return self

    def stop(self, grace: float = 1.0) -> None:
        self._server.stop(grace).wait(grace + 5)
