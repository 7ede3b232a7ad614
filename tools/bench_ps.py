"""Throughput of parameter-server workers on a GPU (the reference's distributed configuration: Inception-v3
trunk + flower head, batch 32, Adam on the ps; recognition_flower/distributed/train_dataset.py:147-192), with the
device part of the step run eagerly or replayed from a HIP graph.  The ps runs in this process on the CPU and is
reached over TCP on localhost, as in a one-machine job.

    python tools/bench_ps.py [--model inception_v3_head] [--batch 32] [--steps 40] [--hip_graph 0|1]
                             [--backend tcp|xgmi] [--pipeline 0|1] [--workers N]

``--pipeline 1``: push and pull on a helper thread while the device computes the next step.  ``--workers N``: N - 1
further worker processes (same GPU, their own connections) train against the same ps task; the report then also
gives the ps task's optimizer steps per second over all workers.

``--backend xgmi`` measures the same worker against the GPU-resident parameter server (``parallel/xps.py``): the
worker GPU owns the variables and Adam slots, push / apply / pull are kernels on peer-mapped memory.  In a world of
one the "peer" is the GPU itself, so the number is the protocol's device cost without xGMI transfer time.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.models import build_model  # noqa: E402
from distributed_tensorflow_amd.parallel.ps import ParameterServer, PSWorkerEngine  # noqa: E402
from distributed_tensorflow_amd.train.session import TrainingSession  # noqa: E402


def run_load_only(args, port, index, chief, ps=None):
    """A worker without a device step: pull, push a fixed gradient of the flower head's size (2048 x 1024 + 1024 x 5
    values), as fast as the ps task answers.  Measures the ps task, not the GPU."""
    import torch.nn as nn

    torch.set_num_threads(1)   # (several of these share the machine with the ps task: no spinning thread teams)
    torch.manual_seed(0)
    model = nn.Sequential(nn.Linear(2048, 1024), nn.Linear(1024, 5))
    eng = PSWorkerEngine(model, [("127.0.0.1", port)], index, chief, optimizer="adam", lr=1e-3, pipeline=bool(args.pipeline))
    for g in eng.flat.groups:
        g.grad.normal_(0, 1e-3)

    def step():
        eng.begin_step()
        for p in eng.flat.params:
            p._grad_filled = True
        eng.finish_step()

    for _ in range(args.warmup):
        step()
    g0 = ps.global_step if ps is not None else 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    dt = time.perf_counter() - t0
    g1 = ps.global_step if ps is not None else 0
    out = {"worker": index, "steps_per_sec": round(args.steps / dt, 1), "ms_per_step": round(dt / args.steps * 1e3, 3),
           "images_per_sec": round(args.batch * args.steps / dt, 1), "loss": 0.0,
           "trainable_values_per_step": sum(g.total for g in eng.flat.groups)}
    return eng, out, (g1 - g0) / dt


def run_worker(args, port, index, chief, ps=None):
    if args.load_only:
        return run_load_only(args, port, index, chief, ps)
    size = 299 if args.model == "inception_v3_head" else 224
    dev = torch.device("cuda", 0)
    model = build_model(args.model, num_classes=5, device=dev)
    if args.backend == "xgmi":
        from distributed_tensorflow_amd.parallel.xps import XgmiPSEngine

        eng = XgmiPSEngine(model, optimizer="adam", lr=1e-3, staleness=args.staleness)
    else:
        eng = PSWorkerEngine(model, [("127.0.0.1", port)], index, chief, optimizer="adam", lr=1e-3,
                             pipeline=bool(args.pipeline))
    sess = TrainingSession(model, eng, num_classes=5, hip_graph=bool(args.hip_graph), reg_weight=0.01)
    x = torch.zeros(args.batch, size, size, 8, device=dev, dtype=torch.bfloat16)
    x[..., :3] = (torch.rand(args.batch, size, size, 3, device=dev) * 2 - 1).to(torch.bfloat16)
    y = torch.randint(0, 5, (args.batch,), device=dev).to(torch.int32)
    for _ in range(args.warmup):
        sess.run_step(x, y)
    torch.cuda.synchronize()
    g0 = ps.global_step if ps is not None else 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = sess.run_step(x, y)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    g1 = ps.global_step if ps is not None else 0
    out = {"worker": index, "images_per_sec": round(args.batch * args.steps / dt, 1),
           "ms_per_step": round(dt / args.steps * 1e3, 3), "loss": round(r.loss, 4),
           "trainable_values_per_step": sum(g.total for g in eng.flat.groups)}
    return eng, out, (g1 - g0) / dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="inception_v3_head")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--hip_graph", type=int, default=1)
    ap.add_argument("--backend", default="tcp", choices=["tcp", "xgmi"])
    ap.add_argument("--staleness", type=int, default=-1)
    ap.add_argument("--pipeline", type=int, default=0)
    ap.add_argument("--workers", type=int, default=1)
    ap.add_argument("--load_only", type=int, default=0,
                    help="1: workers without a device step (pull + push of a fixed gradient): the ps task's own capacity")
    ap.add_argument("--role", default="main", choices=["main", "worker"])
    ap.add_argument("--port", type=int, default=0)
    ap.add_argument("--index", type=int, default=0)
    args = ap.parse_args()
    if args.role == "worker":   # a further worker process of a --workers N run
        eng, out, _ = run_worker(args, args.port, args.index, False)
        print("BENCH_PS_WORKER " + json.dumps(out), flush=True)
        eng.close()
        return
    ps, port = None, 0
    if args.backend == "tcp":
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        ps = ParameterServer("127.0.0.1", port)
        ps.start_background()
    others = []
    for i in range(1, args.workers if args.backend == "tcp" else 1):
        cmd = [sys.executable, os.path.abspath(__file__), "--role", "worker", "--port", str(port), "--index", str(i)]
        for k in ("model", "batch", "steps", "warmup", "hip_graph", "pipeline", "load_only"):
            cmd += ["--" + k, str(getattr(args, k))]
        others.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    eng, out, ups = run_worker(args, port, 0, True, ps)
    rows = [out]
    for p in others:
        text = p.communicate(timeout=600)[0]
        line = [ln for ln in text.splitlines() if ln.startswith("BENCH_PS_WORKER ")]
        if p.returncode != 0 or not line:
            raise SystemExit("a worker process failed:\n" + text[-2000:])
        rows.append(json.loads(line[0][len("BENCH_PS_WORKER "):]))
    report = {"model": args.model, "batch": args.batch, "hip_graph": bool(args.hip_graph), "backend": args.backend,
              "pipeline": bool(args.pipeline), "workers": len(rows), "images_per_sec": out["images_per_sec"],
              "ms_per_step": out["ms_per_step"], "trainable_values_per_step": out["trainable_values_per_step"],
              "loss": out["loss"]}
    if len(rows) > 1:
        report["images_per_sec_all_workers"] = round(sum(r["images_per_sec"] for r in rows), 1)
        # (each worker's own timed window; the windows of late-starting processes overlap worker 0's only partly)
        report["steps_per_sec_all_workers"] = round(sum(r["images_per_sec"] for r in rows) / args.batch, 1)
        report["per_worker"] = rows
    if ps is not None:
        # optimizer steps the ps task applied, all workers together, per second of worker 0's timed window
        report["ps_updates_per_sec"] = round(ups, 1)
    print(json.dumps(report))
    eng.close()
    if ps is not None:
        ps.stop()


if __name__ == "__main__":
    main()
