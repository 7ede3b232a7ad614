"""Servable export and loading (the SavedModel step of the reference).

The reference exports a TensorFlow SavedModel with two signatures
(recognition_flower/single/flower_train.py:195-251,
recognition_flower/distributed/save_model.py:116-171):

``serving_default``  method ``tensorflow/serving/classify``; input ``inputs`` =
    serialized ``tf.Example`` strings carrying ``image/encoded`` JPEG bytes;
    outputs ``classes`` (string) and ``scores`` (float).
``predict_images``   method ``tensorflow/serving/predict``; input ``images`` =
    JPEG byte strings; outputs ``classes`` and ``scores``.

The export keeps the directory contract and the request / response contract::

    <output_dir>/<model_version>/saved_model.pb              SavedModel message: one MetaGraphDef tagged
                                                             "serve" with the SaverDef, the variable
                                                             collections, legacy_init_op and both
                                                             SignatureDefs (io/metagraph.py)
    <output_dir>/<model_version>/saved_model.json            class vocabulary, preprocessing spec, model recipe
    <output_dir>/<model_version>/variables/variables.index   Saver-V2 bundle (TensorFlow variable names)
    <output_dir>/<model_version>/variables/variables.data-00000-of-00001

``graph_def`` inside ``saved_model.pb`` is empty: a TensorFlow graph is a program for TensorFlow's
executor, which does not exist here.  The network is rebuilt from the model recipe in the manifest and the
variables; the signatures are read back from ``saved_model.pb``.

:class:`Servable` loads such a directory and answers both signatures.  Scores
are the raw top-1 logits, exactly like the reference (``tf.nn.top_k(logits, 1)``,
recognition_flower/single/flower_train.py:154); probabilities are returned as an
extra output.  Unlike the reference's ``save_model.py``, which builds its
serving graph with ``training=True``, the servable always runs in inference
mode (batch-norm running statistics, no dropout).
"""
from __future__ import annotations

import json
import os
import shutil
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import bundle
from .saver import Saver
from .tfrecord import make_example, parse_example

MANIFEST = "saved_model.json"
SAVED_MODEL_PB = "saved_model.pb"


def signature_manifest() -> Dict:
    """The two signatures with the tensors the reference binds them to: the ``tf_example`` placeholder (declared
    without a shape), the parsed ``image/encoded`` column, and ``top_k(logits, 1)`` values / looked-up class strings
    created under ``name_scope('train')`` (recognition_flower/single/flower_train.py:120-127,154-156,207-237).
    Tensor names are nominal: no graph exists for them to resolve in."""
    classes = {"dtype": "string", "tensor": "train/index_to_string_Lookup:0", "shape": [-1, 1]}
    scores = {"dtype": "float", "tensor": "train/TopKV2:0", "shape": [-1, 1]}
    return {
        "serving_default": {
            "method_name": "tensorflow/serving/classify",
            "inputs": {"inputs": {"dtype": "string", "tensor": "tf_example:0", "shape": None,
                                  "description": "serialized tf.Example with image/encoded"}},
            "outputs": {"classes": dict(classes), "scores": dict(scores)},
        },
        "predict_images": {
            "method_name": "tensorflow/serving/predict",
            "inputs": {"images": {"dtype": "string", "tensor": "ParseExample/ParseExample:0", "shape": [-1],
                                  "description": "JPEG bytes"}},
            "outputs": {"classes": dict(classes), "scores": dict(scores)},
        },
    }


def read_signatures(export_path: str) -> Dict:
    """SignatureDefs of an export, from its ``saved_model.pb``."""
    from .metagraph import decode_saved_model

    with open(os.path.join(export_path, SAVED_MODEL_PB), "rb") as f:
        sm = decode_saved_model(f.read())
    if "serve" not in sm["tags"]:
        raise ValueError("%s holds no meta graph tagged 'serve'" % export_path)
    return sm["signature_def"]


def signature_def_json(sigs: Dict) -> Dict:
    """SignatureDefs in the JSON form of a model server's metadata reply (TensorInfo as ``name`` / ``dtype`` =
    ``DT_*`` / ``tensor_shape`` with string dimension sizes, the proto3 JSON mapping of int64)."""
    def info(d):
        shape = d.get("shape")
        return {"name": d["tensor"], "dtype": "DT_" + d["dtype"].upper(),
                "tensor_shape": {"dim": [{"size": str(x), "name": ""} for x in (shape or [])],
                                 "unknown_rank": shape is None}}
    return {k: {"inputs": {n: info(d) for n, d in s["inputs"].items()},
                "outputs": {n: info(d) for n, d in s["outputs"].items()},
                "method_name": s["method_name"]} for k, s in sigs.items()}


def export_servable(model, classes: Sequence[str], output_dir: str, model_version: int, *, model_name: str,
                    image_size: int, model_kwargs: Optional[dict] = None, engine=None,
                    global_step: int = 0, overwrite: bool = True) -> str:
    export_path = os.path.join(output_dir, str(model_version))
    if os.path.exists(export_path):
        if not overwrite:
            raise FileExistsError(export_path)
        # the reference shells out to `rm -r` here (recognition_flower/single/flower_train.py:200-201)
        shutil.rmtree(export_path)
    os.makedirs(os.path.join(export_path, "variables"))
    saver = Saver(model, engine)
    tensors = saver.variables(include_slots=False)
    bundle.write_bundle(os.path.join(export_path, "variables", "variables"), tensors)
    from .metagraph import encode_saved_model

    trainable = {v.name: v.trainable for v in saver.specs}
    names = [(n, trainable.get(n, False)) for n in sorted(tensors, key=lambda k: k.encode())]
    with open(os.path.join(export_path, SAVED_MODEL_PB), "wb") as f:
        f.write(encode_saved_model(names, signature_manifest()))
    manifest = {
        "format": "dtfa-servable-1",
        "tags": ["serve"],
        "signature_def": signature_manifest(),
        "classes": list(classes),
        "model": {"name": model_name, "kwargs": model_kwargs or {}, "num_classes": len(classes)},
        "preprocess": {"decode": "jpeg", "resize": "bicubic_legacy", "image_size": int(image_size),
                       "range": [-1.0, 1.0]},
        "global_step": int(global_step),
    }
    with open(os.path.join(export_path, MANIFEST), "w") as f:
        json.dump(manifest, f, indent=2, sort_keys=True)
    return export_path


class Servable:
    #: batch sizes the forward pass is captured for; a request batch is padded up to the next one
    GRAPH_BUCKETS = (1, 2, 4, 8, 16, 32, 64)

    def __init__(self, export_path: str, device=None, hip_graphs: Optional[bool] = None):
        """``hip_graphs`` (default: on for GPU devices): replay the network's forward pass from a HIP graph
        captured per batch-size bucket.  The eval forward of the Inception trunk is ~300 kernels of a few
        microseconds at small batches; enqueueing them from Python costs several times their run time."""
        from ..models import build_model

        with open(os.path.join(export_path, MANIFEST)) as f:
            self.manifest = json.load(f)
        if self.manifest.get("format") != "dtfa-servable-1":
            raise ValueError("unknown servable format")
        self.classes: List[str] = self.manifest["classes"]
        pb = os.path.join(export_path, SAVED_MODEL_PB)
        #: SignatureDefs as exported (saved_model.pb); exports older than that file carry them in the manifest only
        self.signature_def: Dict = read_signatures(export_path) if os.path.exists(pb) else self.manifest["signature_def"]
        self.device = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        m = self.manifest["model"]
        self.model = build_model(m["name"], num_classes=m["num_classes"], device=self.device, **m["kwargs"])
        Saver(self.model).restore(os.path.join(export_path, "variables", "variables"), strict=True)
        self.model.eval()
        self.image_size = int(self.manifest["preprocess"]["image_size"])
        self.method = self.manifest["preprocess"]["resize"]
        self.hip_graphs = (self.device.type == "cuda") if hip_graphs is None else bool(hip_graphs)
        self._graphs: Dict[int, Tuple] = {}

    # ------------------------------------------------------------------ helpers
    def _preprocess(self, jpegs: Sequence[bytes]) -> torch.Tensor:
        from ..data import jpeg as J
        from ..data.preprocess import NORM_INCEPTION, make_descs, preprocess_batch, resize_on_device

        size = (self.image_size, self.image_size)
        if self.device.type == "cuda":
            packed, shapes, offs = J.decode_batch_to_device(jpegs, self.device)
            descs = torch.from_numpy(make_descs(shapes, offs)).to(self.device)
            return resize_on_device(packed, descs, len(jpegs), size, self.method, False, NORM_INCEPTION)
        images = [J.decode_reference(j) for j in jpegs]
        return preprocess_batch(images, size, "cpu", self.method, None, False, NORM_INCEPTION)

    @torch.no_grad()
    def _infer(self, jpegs: Sequence[bytes]) -> Dict[str, np.ndarray]:
        if len(jpegs) == 0:
            z = np.zeros((0, 1))
            return {"classes": z.astype(object), "scores": z.astype(np.float32), "probabilities": np.zeros((0, len(self.classes)), np.float32)}
        x = self._preprocess(jpegs)
        logits = self._forward(x)[:, :len(self.classes)].float()
        scores, idx = logits.max(dim=1, keepdim=True)
        probs = torch.softmax(logits, dim=1)
        names = np.array([[self.classes[i]] for i in idx.view(-1).tolist()], dtype=object)
        return {"classes": names, "scores": scores.cpu().numpy(), "probabilities": probs.cpu().numpy()}

    def _forward(self, x: torch.Tensor) -> torch.Tensor:
        n = x.shape[0]
        bucket = next((b for b in self.GRAPH_BUCKETS if b >= n), None)
        if not (self.hip_graphs and x.is_cuda and bucket is not None):
            out = self.model(x)
            return out[0] if isinstance(out, tuple) else out
        entry = self._graphs.get(bucket)
        if entry is None:
            static_in = torch.zeros((bucket,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
            static_in[:n].copy_(x)
            for _ in range(2):  # eager passes: convolution plans, workspaces and the caching allocator warm up
                self.model(static_in)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            # (thread-local: other threads of a server may issue HIP calls during the capture)
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                out = self.model(static_in)
                static_out = out[0] if isinstance(out, tuple) else out
            entry = self._graphs[bucket] = (graph, static_in, static_out)
        graph, static_in, static_out = entry
        static_in[:n].copy_(x)  # rows past n keep an earlier request's pixels; their outputs are never read
        graph.replay()
        return static_out[:n].clone()

    # ------------------------------------------------------------------ signatures
    def predict_images(self, images: Sequence[bytes]) -> Dict[str, np.ndarray]:
        """``predict_images`` signature: JPEG bytes in, (class string, raw top-1 logit) out, shape [N, 1]."""
        return self._infer(list(images))

    def serving_default(self, inputs: Sequence[bytes]) -> Dict[str, np.ndarray]:
        """``serving_default`` (classify) signature: serialized tf.Example strings with ``image/encoded``."""
        jpegs = []
        for rec in inputs:
            ex = parse_example(rec)
            enc = ex.get("image/encoded")
            if not enc:
                raise ValueError("tf.Example lacks the image/encoded feature")
            jpegs.append(enc[0])
        return self._infer(jpegs)

    classify = serving_default


def example_from_jpeg(jpeg: bytes) -> bytes:
    """Client-side helper: the tf.Example the classify signature expects."""
    return make_example({"image/encoded": jpeg})
