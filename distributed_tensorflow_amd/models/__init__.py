"""Model families: Inception-v3 + flower head (the reference model: frozen trunk; ``inception_v3_finetune``
trains the trunk as well), a small flower CNN, ResNet-50."""
from __future__ import annotations

import torch

from .layers import cast_for_device

MODEL_NAMES = ("inception_v3_head", "inception_v3_finetune", "flower_cnn", "resnet50")


def build_model(name: str, num_classes: int = 5, device="cpu", **kw):
    if name == "resnet50":
        from .resnet import ResNet50

        m = ResNet50(num_classes=num_classes, **kw)
    elif name == "flower_cnn":
        from .flower_cnn import FlowerCNN

        m = FlowerCNN(num_classes=num_classes, **kw)
    elif name in ("inception_v3_head", "inception_v3"):
        from .inception_v3 import InceptionV3Flower

        m = InceptionV3Flower(num_classes=num_classes, **kw)
    elif name == "inception_v3_finetune":
        from .inception_v3 import InceptionV3Flower

        m = InceptionV3Flower(num_classes=num_classes, train_trunk=True, **kw)
    else:
        raise ValueError("unknown model %r (choose from %s)" % (name, MODEL_NAMES))
    return cast_for_device(m, device)
