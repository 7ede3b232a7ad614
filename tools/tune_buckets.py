#!/usr/bin/env python
"""Measure the gradient bucket size on the machine at hand and record the winner.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/tune_buckets.py [--model resnet50] [--mode sync_dp] [--sizes 4,8,16,32,64,128] [--steps 8] [--warmup 3]
                              [--batch 0] [--write 1]

Runs whole training steps of ``bench.py``'s configuration (same model, batch, optimizer, synthetic data) once per
bucket size, on all ranks of the job, and prints images/s and the exposed collective wait per step for each.  With
``--write 1`` rank 0 stores the fastest size in ``distributed_tensorflow_amd/parallel/tuned_buckets.json``; the
engine's automatic bucket size (``bucket_mb=0``, the benchmark's default) then uses it for this world size, mode and
models of similar gradient size (``parallel/rccl.py``).  A world of one has no collectives to tune and is refused.
"""
import argparse
import gc
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="resnet50")
    ap.add_argument("--mode", default="sync_dp", choices=["sync_dp", "sharded_ps", "async_ssp"])
    ap.add_argument("--sizes", default="4,8,16,32,64,128")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"])
    ap.add_argument("--image_size", type=int, default=0)
    ap.add_argument("--write", type=int, default=1)
    a = ap.parse_args()

    import torch
    import torch.distributed as dist

    from distributed_tensorflow_amd.parallel.rccl import configure_rccl, record_bucket

    configure_rccl()
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    if world < 2:
        raise SystemExit("tune_buckets.py measures collectives: start it with at least two ranks")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if a.device == "cpu":
        device = torch.device("cpu")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    rows = []
    for mb in [float(x) for x in a.sizes.split(",") if x]:
        args = bench.parse_args(["--gpus", str(world), "--model", a.model, "--mode", a.mode, "--steps", str(a.steps),
                                 "--warmup", str(a.warmup), "--bucket_mb", str(mb), "--device", a.device,
                                 "--image_size", str(a.image_size), "--hip_graph", "0"])
        r = bench.run_config(args, a.model, a.batch or bench.DEFAULT_BATCH[a.model], device, rank, world)
        rows.append({"bucket_mb": mb, "images_per_sec": r["value"], "ms_per_step": r["ms_per_step"],
                     "buckets": r["comm"]["buckets"], "exposed_wait_ms_per_step": r["comm"].get("exposed_wait_ms_per_step"),
                     "gradient_mb": r["comm"]["gradient_bytes"] / float(1 << 20)})
        if rank == 0:
            print(json.dumps(rows[-1]), flush=True)
        gc.collect()
        if device.type == "cuda":
            torch.cuda.empty_cache()
    best = max(rows, key=lambda r: r["images_per_sec"])
    if rank == 0:
        print(json.dumps({"world": world, "mode": a.mode, "model": a.model, "best": best}), flush=True)
        if a.write:
            record_bucket(world, a.mode, a.model, best["gradient_mb"], best["bucket_mb"],
                          [{k: r[k] for k in ("bucket_mb", "images_per_sec", "exposed_wait_ms_per_step")} for r in rows])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
