"""Input pipeline of the single-process trainer.

Same module-level API as the reference file of the same name
(recognition_flower/single/flower_input_data.py:12-81 in the reference tree):
``get_image_path_and_class``, ``prepare_image``, ``prepare_batch``.  The
implementation lives in ``distributed_tensorflow_amd.data.flowers``; on a GPU
the trainer does not call ``prepare_batch`` at all but decodes and augments on
the device (``prepare_batch_on_device``).
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from distributed_tensorflow_amd.data import jpeg as _jpeg  # noqa: E402
from distributed_tensorflow_amd.data.flowers import (get_image_path_and_class, listdir_without_hidden,  # noqa: E402,F401
                                                     prepare_batch, prepare_image, read_image, sample_batch)
from distributed_tensorflow_amd.data.preprocess import (NORM_INCEPTION, make_descs, random_crop_spec,  # noqa: E402
                                                        resize_on_device)


def prepare_batch_on_device(flower_paths_and_classes, batch_size, device, image_size=299, max_zoom=0.2, np_rng=None):
    """GPU version of ``prepare_batch``: JPEG entropy decode on host threads,
    IDCT / colour conversion, random crop + zoom + flip and the bilinear resize
    on the device.  Returns ``X [B, S, S, 8]`` bf16 in [-1, 1] and ``y [B]`` int32."""
    np_rng = np_rng or np.random.default_rng()
    batch = sample_batch(flower_paths_and_classes, batch_size)
    jpegs = []
    for path, _ in batch:
        with open(path, "rb") as f:
            jpegs.append(f.read())
    packed, shapes, offs = _jpeg.decode_batch_to_device(jpegs, device)
    crops = [random_crop_spec(h, w, image_size, image_size, max_zoom, np_rng) for h, w, _ in shapes]
    descs = torch.from_numpy(make_descs(shapes, offs, crops)).to(device)
    X = resize_on_device(packed, descs, len(batch), (image_size, image_size), "bilinear", False, NORM_INCEPTION)
    y = torch.tensor([c for _, c in batch], dtype=torch.int32, device=device)
    return X, y
