"""Servable export and loading (the SavedModel step of the reference).

The reference exports a TensorFlow SavedModel with two signatures
(recognition_flower/single/flower_train.py:195-251,
recognition_flower/distributed/save_model.py:116-171):

``serving_default``  method ``tensorflow/serving/classify``; input ``inputs`` =
    serialized ``tf.Example`` strings carrying ``image/encoded`` JPEG bytes;
    outputs ``classes`` (string) and ``scores`` (float).
``predict_images``   method ``tensorflow/serving/predict``; input ``images`` =
    JPEG byte strings; outputs ``classes`` and ``scores``.

``saved_model.pb`` is a serialized TensorFlow graph and cannot be produced
without TensorFlow, so the export here is a *servable bundle* with the same
directory contract and the same request / response contract::

    <output_dir>/<model_version>/saved_model.json            signature manifest, class vocabulary,
                                                             preprocessing spec, model recipe
    <output_dir>/<model_version>/variables/variables.index   Saver-V2 bundle (TensorFlow variable names)
    <output_dir>/<model_version>/variables/variables.data-00000-of-00001

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


def signature_manifest() -> Dict:
    return {
        "serving_default": {
            "method_name": "tensorflow/serving/classify",
            "inputs": {"inputs": {"dtype": "string", "description": "serialized tf.Example with image/encoded"}},
            "outputs": {"classes": {"dtype": "string"}, "scores": {"dtype": "float"}},
        },
        "predict_images": {
            "method_name": "tensorflow/serving/predict",
            "inputs": {"images": {"dtype": "string", "description": "JPEG bytes"}},
            "outputs": {"classes": {"dtype": "string"}, "scores": {"dtype": "float"}},
        },
    }


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
        self.device = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
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
            with torch.cuda.graph(graph):
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
