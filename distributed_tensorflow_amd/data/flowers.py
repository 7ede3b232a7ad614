"""Flower-photo directory dataset (class-per-subdirectory JPEGs).

API of the reference's ``flower_input_data`` module
(recognition_flower/single/flower_input_data.py:12-81) and the listing /
splitting code of its trainer (recognition_flower/single/flower_train.py:88-106):
``get_image_path_and_class``, ``prepare_image``, ``prepare_batch``.  Class names
are sorted so that class ids do not depend on directory enumeration order (the
reference uses raw ``os.listdir`` order).
"""
from __future__ import annotations

import os
import random
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np

from .preprocess import CropSpec, NORM_UNIT, random_crop_spec, resize_reference


def listdir_without_hidden(path: str) -> Iterator[str]:
    for f in os.listdir(path):
        if not f.startswith("."):
            yield f


def list_classes(root_path: str) -> List[str]:
    return sorted(d for d in listdir_without_hidden(root_path) if os.path.isdir(os.path.join(root_path, d)))


def get_image_path_and_class(root_path: str, flower_classes: Sequence[str]) -> List[Tuple[str, int]]:
    """[(path, class id)] for every ``*.jpg`` under ``root/<class>/``; class id = index in ``flower_classes``."""
    out = []
    for cid, cls in enumerate(flower_classes):
        d = os.path.join(root_path, cls)
        for f in sorted(os.listdir(d)):
            if f.endswith(".jpg") and not f.startswith("."):
                out.append((os.path.join(d, f), cid))
    return out


def train_test_split(items: List[Tuple[str, int]], test_ratio: float = 0.2, seed: Optional[int] = None):
    """Shuffle and split (the reference shuffles with numpy and holds out 20 %,
    recognition_flower/single/flower_train.py:101-106)."""
    items = list(items)
    rng = np.random.default_rng(seed)
    rng.shuffle(items)
    n_train = int(len(items) * (1 - test_ratio))
    return items[:n_train], items[n_train:]


def read_image(path: str) -> np.ndarray:
    """Decode to uint8 RGB [H, W, 3]."""
    from PIL import Image

    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))


def prepare_image(image: np.ndarray, target_width: int = 299, target_height: int = 299, max_zoom: float = 0.2,
                  rng: Optional[np.random.Generator] = None) -> np.ndarray:
    """Random crop (target aspect ratio, zoom <= 1 + max_zoom), random horizontal
    flip, bilinear resize; returns float32 in [0, 1]."""
    rng = rng or np.random.default_rng()
    if image.dtype != np.uint8:
        image = np.clip(image * 255.0 if image.max() <= 1.0 else image, 0, 255).astype(np.uint8)
    h, w = image.shape[:2]
    crop = random_crop_spec(h, w, target_height, target_width, max_zoom, rng)
    return resize_reference(image, (target_height, target_width), "bilinear", crop, norm=NORM_UNIT)


def sample_batch(flower_paths_and_classes: Sequence[Tuple[str, int]], batch_size: int,
                 rng: Optional[random.Random] = None) -> List[Tuple[str, int]]:
    """``batch_size`` distinct items (sampling without replacement within a batch)."""
    rng = rng or random
    return rng.sample(list(flower_paths_and_classes), batch_size)


def prepare_batch(flower_paths_and_classes: Sequence[Tuple[str, int]], batch_size: int, image_size: int = 299,
                  np_rng: Optional[np.random.Generator] = None) -> Tuple[np.ndarray, np.ndarray]:
    """numpy batch ``X [B, S, S, 3]`` float32 in [-1, 1] and ``y [B]`` int32."""
    batch = sample_batch(flower_paths_and_classes, batch_size)
    images = [prepare_image(read_image(p), image_size, image_size, rng=np_rng) for p, _ in batch]
    X = 2.0 * np.stack(images) - 1.0
    y = np.array([c for _, c in batch], dtype=np.int32)
    return X.astype(np.float32), y
