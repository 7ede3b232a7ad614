"""Cost of the reference's checkpoint cadence (CheckpointSaverHook, save_steps=2,
recognition_flower/distributed/train_dataset.py:176-178) on the Inception-v3 + head model, batch 32:
step time without checkpoints, with checkpoints written in line, and with the snapshot taken in the step and the
files written by the helper thread (frozen variables converted once).

    python tools/bench_ckpt.py [--steps 40] [--save_steps 2]
"""
import argparse
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.io.saver import Saver  # noqa: E402
from distributed_tensorflow_amd.models import build_model  # noqa: E402
from distributed_tensorflow_amd.parallel.engine import DataParallelEngine  # noqa: E402
from distributed_tensorflow_amd.train.session import CheckpointSaverHook, TrainingSession  # noqa: E402


def run(steps, save_steps, mode, x, y, dev):
    model = build_model("inception_v3_head", num_classes=5, device=dev)
    eng = DataParallelEngine(model, mode="sync_dp", optimizer="adam", lr=1e-3)
    saver = Saver(model, eng)
    with tempfile.TemporaryDirectory() as d:
        hooks = [] if mode == "none" else [CheckpointSaverHook(d, save_steps, saver, background=(mode == "background"))]
        if mode == "inline":
            saver._frozen_disabled = True
        sess = TrainingSession(model, eng, num_classes=5, hooks=hooks, hip_graph=True)
        with sess:
            for _ in range(6):
                sess.run_step(x, y)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                sess.run_step(x, y)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        return dt / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--save_steps", type=int, default=2)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    x = torch.zeros(32, 299, 299, 8, device=dev, dtype=torch.bfloat16)
    x[..., :3] = (torch.rand(32, 299, 299, 3, device=dev) * 2 - 1).to(torch.bfloat16)
    y = torch.randint(0, 5, (32,), device=dev).to(torch.int32)
    for mode in ("none", "inline", "background"):
        print("checkpoints %-10s (every %d steps): %.2f ms/step" % (mode, args.save_steps,
                                                                    run(args.steps, args.save_steps, mode, x, y, dev)))


if __name__ == "__main__":
    main()
