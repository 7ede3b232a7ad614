"""Throughput of ONE parameter-server worker on a GPU (the reference's distributed configuration: Inception-v3
trunk + flower head, batch 32, Adam on the ps; recognition_flower/distributed/train_dataset.py:147-192), with the
device part of the step run eagerly or replayed from a HIP graph.  The ps runs in this process on the CPU and is
reached over TCP on localhost, as in a one-machine job.

    python tools/bench_ps.py [--model inception_v3_head] [--batch 32] [--steps 40] [--hip_graph 0|1] [--backend tcp|xgmi]

``--backend xgmi`` measures the same worker against the GPU-resident parameter server (``parallel/xps.py``): the
worker GPU owns the variables and Adam slots, push / apply / pull are kernels on peer-mapped memory.  In a world of
one the "peer" is the GPU itself, so the number is the protocol's device cost without xGMI transfer time.
"""
import argparse
import json
import os
import socket
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.models import build_model  # noqa: E402
from distributed_tensorflow_amd.parallel.ps import ParameterServer, PSWorkerEngine  # noqa: E402
from distributed_tensorflow_amd.train.session import TrainingSession  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="inception_v3_head")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--hip_graph", type=int, default=1)
    ap.add_argument("--backend", default="tcp", choices=["tcp", "xgmi"])
    ap.add_argument("--staleness", type=int, default=-1)
    args = ap.parse_args()
    size = 299 if args.model == "inception_v3_head" else 224
    dev = torch.device("cuda", 0)
    model = build_model(args.model, num_classes=5, device=dev)
    ps = None
    if args.backend == "xgmi":
        from distributed_tensorflow_amd.parallel.xps import XgmiPSEngine

        eng = XgmiPSEngine(model, optimizer="adam", lr=1e-3, staleness=args.staleness)
    else:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        ps = ParameterServer("127.0.0.1", port)
        ps.start_background()
        eng = PSWorkerEngine(model, [("127.0.0.1", port)], 0, True, optimizer="adam", lr=1e-3)
    sess = TrainingSession(model, eng, num_classes=5, hip_graph=bool(args.hip_graph), reg_weight=0.01)
    x = torch.zeros(args.batch, size, size, 8, device=dev, dtype=torch.bfloat16)
    x[..., :3] = (torch.rand(args.batch, size, size, 3, device=dev) * 2 - 1).to(torch.bfloat16)
    y = torch.randint(0, 5, (args.batch,), device=dev).to(torch.int32)
    for _ in range(args.warmup):
        sess.run_step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = sess.run_step(x, y)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"model": args.model, "batch": args.batch, "hip_graph": bool(args.hip_graph), "backend": args.backend,
                      "images_per_sec": round(args.batch * args.steps / dt, 1), "ms_per_step": round(dt / args.steps * 1e3, 3),
                      "trainable_values_per_step": sum(g.total for g in eng.flat.groups), "loss": round(r.loss, 4)}))
    eng.close()
    if ps is not None:
        ps.stop()


if __name__ == "__main__":
    main()
