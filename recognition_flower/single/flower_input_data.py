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
from distributed_tensorflow_amd.data.preprocess import (NORM_INCEPTION, NORM_UNIT, full_crop, make_descs,  # noqa: E402
                                                        random_crop_spec, resize_on_device, resize_reference)


def prepare_batch_on_device(flower_paths_and_classes, batch_size, device, image_size=299, max_zoom=0.2, np_rng=None,
                            cache=None):
    """GPU version of ``prepare_batch``: JPEG entropy decode on host threads,
    IDCT / colour conversion, random crop + zoom + flip and the bilinear resize
    on the device.  Returns ``X [B, S, S, 8]`` bf16 in [-1, 1] and ``y [B]`` int32.
    ``cache`` (``data.jpeg.DecodedImageCache``) keeps decoded images on the device: a file is read and decoded the
    first time it is drawn only."""
    np_rng = np_rng or np.random.default_rng()
    batch = sample_batch(flower_paths_and_classes, batch_size)
    packed, shapes, offs = _jpeg.decode_files_to_device([p for p, _ in batch], device, cache)
    crops = [random_crop_spec(h, w, image_size, image_size, max_zoom, np_rng) for h, w, _ in shapes]
    descs = torch.from_numpy(make_descs(shapes, offs, crops)).to(device)
    X = resize_on_device(packed, descs, len(batch), (image_size, image_size), "bilinear", False, NORM_INCEPTION)
    y = torch.tensor([c for _, c in batch], dtype=torch.int32, device=device)
    return X, y


def evaluation_batches(flower_paths_and_classes, batch_size, device, image_size=299, cache=None):
    """Every item once, in order, WITHOUT augmentation: the whole image resized (bilinear) to the network size,
    [-1, 1].  Yields ``(X, y)`` like the training batches (device: bf16 NHWC8; CPU: fp32 NHWC8).  This is the
    pass over the held-out 20 % that the reference splits off and never runs
    (recognition_flower/single/flower_train.py:101-106)."""
    device = torch.device(device)
    items = list(flower_paths_and_classes)
    for i in range(0, len(items), batch_size):
        batch = items[i:i + batch_size]
        y = torch.tensor([c for _, c in batch], dtype=torch.int32, device=device)
        if device.type == "cuda":
            packed, shapes, offs = _jpeg.decode_files_to_device([p for p, _ in batch], device, cache)
            descs = torch.from_numpy(make_descs(shapes, offs, [full_crop(h, w) for h, w, _ in shapes])).to(device)
            X = resize_on_device(packed, descs, len(batch), (image_size, image_size), "bilinear", False, NORM_INCEPTION)
        else:
            Xn = np.zeros((len(batch), image_size, image_size, 8), dtype=np.float32)
            for k, (path, _) in enumerate(batch):
                Xn[k, :, :, :3] = 2.0 * resize_reference(read_image(path), (image_size, image_size), "bilinear", None,
                                                         norm=NORM_UNIT) - 1.0
            X = torch.from_numpy(Xn)
        yield X, y
