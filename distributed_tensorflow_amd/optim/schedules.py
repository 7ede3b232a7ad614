"""Learning-rate schedules (host scalars).

``exponential_decay`` is TensorFlow's ``tf.train.exponential_decay``; the
reference's single-process trainer uses lr 0.001, decay_steps 200, decay_rate
0.95, non-staircase (recognition_flower/single/flower_train.py:113-117).
"""
from __future__ import annotations

import math
from typing import Callable

Schedule = Callable[[int], float]


def constant(lr: float) -> Schedule:
    return lambda step: lr


def exponential_decay(lr: float, decay_steps: int, decay_rate: float, staircase: bool = False) -> Schedule:
    def f(step: int) -> float:
        p = step / decay_steps
        if staircase:
            p = math.floor(p)
        return lr * decay_rate ** p

    return f


def warmup_cosine(lr: float, warmup_steps: int, total_steps: int, final_ratio: float = 0.0) -> Schedule:
    def f(step: int) -> float:
        if warmup_steps > 0 and step < warmup_steps:
            return lr * (step + 1) / warmup_steps
        t = min(1.0, (step - warmup_steps) / max(1, total_steps - warmup_steps))
        return lr * (final_ratio + (1 - final_ratio) * 0.5 * (1 + math.cos(math.pi * t)))

    return f
