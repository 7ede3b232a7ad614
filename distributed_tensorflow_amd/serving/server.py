"""HTTP model server for exported servables.

The reference stops at ``builder.save()`` of a SavedModel whose two signatures --
``serving_default`` (classify) and ``predict_images`` (predict) -- are meant to be
served by a model server (recognition_flower/single/flower_train.py:205-251,
recognition_flower/distributed/save_model.py:126-171).  This module is that
server for servable bundles, speaking the REST dialect such servers expose:

    GET  /v1/models/<name>                              version status
    GET  /v1/models/<name>/metadata                     signature definitions
    POST /v1/models/<name>[/versions/<v>]:predict       {"signature_name": "predict_images",
                                                         "instances": [{"b64": "<jpeg>"}, ...]}
                                                        -> {"predictions": [{"classes": [..], "scores": [..]}, ...]}
                                                        (columnar: {"inputs": {"images": [...]}} -> {"outputs": {...}})
    POST /v1/models/<name>[/versions/<v>]:classify      {"examples": [{"image/encoded": {"b64": "<jpeg>"}}, ...]}
                                                        -> {"results": [[["daisy", 7.1]], ...]}
    GET  /metrics                                       Prometheus text: requests, batches, latency, queue depth
    GET  /healthz

``predict`` with ``signature_name`` ``serving_default`` takes base64 serialized
``tf.Example`` protos, as the classify signature's ``inputs`` tensor does.

Threads: one HTTP thread per connection parses and base64-decodes; every image
then goes through one :class:`DynamicBatcher`, whose single worker thread owns
the GPU (JPEG entropy decode on the host, IDCT / resize / network on the device).
"""
from __future__ import annotations

import base64
import json
import re
import threading
import time
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer
from typing import Dict, List, Optional, Tuple

import numpy as np

from ..io.export import signature_def_json
from ..io.tfrecord import parse_example
from .batcher import DynamicBatcher, QueueFull
from .manager import ModelManager

_ROUTE = re.compile(r"^/v1/models/(?P<name>[^/:]+)(?:/versions/(?P<version>\d+))?(?::(?P<verb>predict|classify)|/(?P<meta>metadata))?$")


class BadRequest(ValueError):
    pass


def _bytes_of(value, what: str) -> bytes:
    """A binary JSON value: {"b64": "..."}."""
    if isinstance(value, dict) and set(value) == {"b64"} and isinstance(value["b64"], str):
        try:
            return base64.b64decode(value["b64"], validate=True)
        except Exception:  # noqa: BLE001
            raise BadRequest("%s: invalid base64" % what)
    raise BadRequest('%s: binary values are sent as {"b64": "<base64>"}' % what)


def _jpeg_of_example_bytes(rec: bytes) -> bytes:
    try:
        enc = parse_example(rec).get("image/encoded")
    except Exception as e:  # noqa: BLE001
        raise BadRequest("malformed tf.Example: %s" % e)
    if not enc:
        raise BadRequest("tf.Example lacks the image/encoded feature")
    return enc[0]


class ModelServer:
    def __init__(self, manager: ModelManager, host: str = "127.0.0.1", port: int = 8501, max_batch: int = 64,
                 batch_timeout_ms: float = 2.0, max_queue: int = 4096):
        self.manager = manager
        self.batcher = DynamicBatcher(self._run_batch, max_batch=max_batch, timeout_ms=batch_timeout_ms,
                                      max_queue=max_queue, name="gpu-batcher")
        self._stats_lock = threading.Lock()
        self.http_requests: Dict[Tuple[str, int], int] = {}
        self.latency_sum = 0.0
        self.latency_count = 0
        server = self

        class Handler(BaseHTTPRequestHandler):
            protocol_version = "HTTP/1.1"

            def log_message(self, fmt, *args):  # quiet; counters are exported on /metrics
                pass

            def _send(self, code: int, body, content_type="application/json"):
                data = body if isinstance(body, bytes) else json.dumps(body).encode()
                self.send_response(code)
                self.send_header("Content-Type", content_type)
                self.send_header("Content-Length", str(len(data)))
                self.end_headers()
                self.wfile.write(data)

            def do_GET(self):
                code, body, ctype = server.handle_get(self.path)
                self._send(code, body, ctype)

            def do_POST(self):
                n = int(self.headers.get("Content-Length") or 0)
                raw = self.rfile.read(n) if n else b""
                code, body = server.handle_post(self.path, raw)
                self._send(code, body)

        class Server(ThreadingHTTPServer):
            # socketserver listens with a backlog of 5: a burst of a few dozen connecting clients gets resets
            request_queue_size = 1024
            daemon_threads = True

        self.httpd = Server((host, port), Handler)
        self.port = self.httpd.server_address[1]
        self._thread: Optional[threading.Thread] = None

    # ------------------------------------------------------------------ inference
    def _run_batch(self, jpegs: List[bytes]) -> Dict[str, np.ndarray]:
        return self.manager.get().predict_images(jpegs)

    def infer(self, jpegs: List[bytes], version: Optional[int]) -> Dict[str, np.ndarray]:
        if version is not None:
            self.manager.get(version)  # raises LookupError for a version that is not being served
        return self.batcher.submit(jpegs).result()

    # ------------------------------------------------------------------ request handling
    def _count(self, route: str, code: int, t0: float):
        with self._stats_lock:
            self.http_requests[(route, code)] = self.http_requests.get((route, code), 0) + 1
            if code == 200 and route in ("predict", "classify"):
                self.latency_sum += time.monotonic() - t0
                self.latency_count += 1

    def handle_get(self, path: str):
        t0 = time.monotonic()
        if path == "/healthz":
            ok = self.manager.version is not None
            return (200 if ok else 503), {"status": "ok" if ok else "no model loaded"}, "application/json"
        if path == "/metrics":
            return 200, self.metrics_text().encode(), "text/plain; version=0.0.4"
        m = _ROUTE.match(path)
        if not m or m.group("verb") or m.group("name") != self.manager.name:
            self._count("get", 404, t0)
            return 404, {"error": "not found: %s" % path}, "application/json"
        if m.group("meta"):
            try:
                servable = self.manager.get(int(m.group("version")) if m.group("version") else None)
            except LookupError as e:
                self._count("metadata", 404, t0)
                return 404, {"error": str(e)}, "application/json"
            body = {"model_spec": {"name": self.manager.name, "signature_name": "", "version": str(self.manager.version)},
                    "metadata": {"signature_def": {"signature_def": signature_def_json(servable.signature_def)},
                                 "classes": servable.classes, "preprocess": servable.manifest["preprocess"]}}
            self._count("metadata", 200, t0)
            return 200, body, "application/json"
        self._count("status", 200, t0)
        return 200, {"model_version_status": self.manager.status()}, "application/json"

    def handle_post(self, path: str, raw: bytes):
        t0 = time.monotonic()
        m = _ROUTE.match(path)
        if not m or not m.group("verb") or m.group("name") != self.manager.name:
            self._count("post", 404, t0)
            return 404, {"error": "not found: %s" % path}
        verb = m.group("verb")
        version = int(m.group("version")) if m.group("version") else None
        try:
            try:
                req = json.loads(raw.decode("utf-8"))
            except Exception:  # noqa: BLE001
                raise BadRequest("request body is not JSON")
            if not isinstance(req, dict):
                raise BadRequest("request body must be a JSON object")
            body = self._predict(req, version) if verb == "predict" else self._classify(req, version)
            code = 200
        except BadRequest as e:
            code, body = 400, {"error": str(e)}
        except LookupError as e:
            code, body = 404, {"error": str(e)}
        except QueueFull as e:
            code, body = 503, {"error": "server overloaded: %s" % e}
        except Exception as e:  # noqa: BLE001 -- e.g. a corrupt JPEG in the batch
            code, body = 400, {"error": "%s: %s" % (type(e).__name__, e)}
        self._count(verb, code, t0)
        return code, body

    def _predict(self, req: dict, version: Optional[int]) -> dict:
        sig = req.get("signature_name") or "predict_images"
        if sig not in ("predict_images", "serving_default"):
            raise BadRequest("unknown signature %r" % sig)
        input_name = "images" if sig == "predict_images" else "inputs"
        if ("instances" in req) == ("inputs" in req):
            raise BadRequest('exactly one of "instances" (row format) and "inputs" (columnar format) is required')
        row = "instances" in req
        values = req["instances"] if row else req["inputs"]
        if not row and isinstance(values, dict):
            if set(values) != {input_name}:
                raise BadRequest("signature %s takes the single input %r" % (sig, input_name))
            values = values[input_name]
        if not isinstance(values, list):
            raise BadRequest("a list of inputs is required")
        blobs = []
        for i, v in enumerate(values):
            if row and isinstance(v, dict) and set(v) == {input_name}:
                v = v[input_name]
            blobs.append(_bytes_of(v, "%s[%d]" % (input_name, i)))
        jpegs = blobs if sig == "predict_images" else [_jpeg_of_example_bytes(b) for b in blobs]
        out = self.infer(jpegs, version)
        classes = [[str(c) for c in r] for r in out["classes"].tolist()]
        scores = [[float(s) for s in r] for r in out["scores"].tolist()]
        if row:
            return {"predictions": [{"classes": c, "scores": s} for c, s in zip(classes, scores)]}
        return {"outputs": {"classes": classes, "scores": scores}}

    def _classify(self, req: dict, version: Optional[int]) -> dict:
        sig = req.get("signature_name") or "serving_default"
        if sig != "serving_default":
            raise BadRequest("classify is served by the serving_default signature")
        examples = req.get("examples")
        if not isinstance(examples, list):
            raise BadRequest('"examples" must be a list of feature dictionaries')
        context = req.get("context") or {}
        jpegs = []
        for i, ex in enumerate(examples):
            if not isinstance(ex, dict):
                raise BadRequest("examples[%d] is not an object" % i)
            feat = ex.get("image/encoded", context.get("image/encoded"))
            if feat is None:
                raise BadRequest("examples[%d] lacks the image/encoded feature" % i)
            if isinstance(feat, list):
                if len(feat) != 1:
                    raise BadRequest("examples[%d]: image/encoded holds one value" % i)
                feat = feat[0]
            jpegs.append(_bytes_of(feat, "examples[%d].image/encoded" % i))
        out = self.infer(jpegs, version)
        results = [[[str(c), float(s)] for c, s in zip(cr, sr)]
                   for cr, sr in zip(out["classes"].tolist(), out["scores"].tolist())]
        return {"results": results}

    # ------------------------------------------------------------------ metrics
    def metrics_text(self) -> str:
        b = self.batcher.stats()
        lines = ["# TYPE dtfa_http_requests_total counter"]
        with self._stats_lock:
            for (route, code), n in sorted(self.http_requests.items()):
                lines.append('dtfa_http_requests_total{route="%s",code="%d"} %d' % (route, code, n))
            lines += ["# TYPE dtfa_request_latency_seconds summary",
                      "dtfa_request_latency_seconds_sum %.6f" % self.latency_sum,
                      "dtfa_request_latency_seconds_count %d" % self.latency_count]
        lines += ["# TYPE dtfa_batches_total counter", "dtfa_batches_total %d" % b["batches"],
                  "# TYPE dtfa_batched_items_total counter", "dtfa_batched_items_total %d" % b["items"],
                  "# TYPE dtfa_rejected_requests_total counter", "dtfa_rejected_requests_total %d" % b["rejected"],
                  "# TYPE dtfa_batch_busy_seconds_total counter", "dtfa_batch_busy_seconds_total %.6f" % b["busy_seconds"],
                  "# TYPE dtfa_queue_wait_seconds_total counter", "dtfa_queue_wait_seconds_total %.6f" % b["queue_wait_seconds"],
                  "# TYPE dtfa_queued_items gauge", "dtfa_queued_items %d" % b["queued_items"],
                  "# TYPE dtfa_model_version gauge", "dtfa_model_version %d" % (self.manager.version or 0),
                  "# TYPE dtfa_model_loads_total counter", "dtfa_model_loads_total %d" % self.manager.loads]
        return "\n".join(lines) + "\n"

    # ------------------------------------------------------------------ lifecycle
    def start(self) -> "ModelServer":
        self._thread = threading.Thread(target=self.httpd.serve_forever, name="http", daemon=True)
        self._thread.start()
        return self

    def serve_forever(self):
        self.httpd.serve_forever()

    def stop(self):
        self.httpd.shutdown()
        self.httpd.server_close()
        if self._thread is not None:
            self._thread.join(timeout=10)
        self.batcher.close()
        self.manager.stop()
