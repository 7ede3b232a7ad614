#!/usr/bin/env python
"""Collective micro-benchmark: what the gradient buckets of the data-parallel engine cost on the wire.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/bench_collectives.py [--min_kb 64] [--max_mb 256] [--dtype fp32|bf16] [--ops all_reduce,...]
                                   [--device cuda|cpu] [--iters 20] [--json out.json]

For every message size (powers of four between ``--min_kb`` and ``--max_mb``) and every operation the engine uses --
``all_reduce`` (sync_dp), ``reduce_scatter`` + ``all_gather`` (sharded_ps), and the xGMI mailbox all-reduce in its one-shot and two-shot forms
(``mailbox`` / ``mailbox2``, small and medium buckets, GPUs only) -- it prints the time per call, the *algorithmic* bandwidth (message bytes /
time: what a bucket sees) and the *bus* bandwidth (bytes each rank actually moves over its links / time: 2 (N-1)/N
of the message for all-reduce, (N-1)/N for the other two), which is the number to hold against the ~153 GB/s of one
xGMI link or the 7 links of a GPU.  A world of one (``DTFA_FORCE_COLLECTIVES``-style) measures the fixed launch cost
of each call.  On CPU tensors the backend is gloo (plumbing check only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def sizes(min_kb: int, max_mb: int):
    n = min_kb * 1024
    while n <= max_mb * (1 << 20):
        yield n
        n *= 4


def bus_factor(op: str, world: int) -> float:
    if world <= 1:
        return 1.0
    return 2.0 * (world - 1) / world if op in ("all_reduce", "mailbox", "mailbox2") else (world - 1) / world


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--min_kb", type=int, default=64)
    ap.add_argument("--max_mb", type=int, default=256)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--ops", default="all_reduce,reduce_scatter,all_gather,mailbox,mailbox2")
    ap.add_argument("--device", default="auto")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mailbox_max_mb", type=float, default=16.0, help="largest message sent through the mailbox path")
    ap.add_argument("--json", default="")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = a.device == "cuda" or (a.device == "auto" and torch.cuda.is_available())
    if use_cuda:
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    else:
        device = torch.device("cpu")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    dist.init_process_group("nccl" if use_cuda else "gloo", rank=rank, world_size=world,
                            **({"device_id": device} if use_cuda else {}))
    dtype = torch.float32 if a.dtype == "fp32" else torch.bfloat16
    esz = 4 if a.dtype == "fp32" else 2
    ops = [o for o in a.ops.split(",") if o]
    if not use_cuda:
        ops = [o for o in ops if not o.startswith("mailbox")]  # hipIpc-mapped peer mailboxes exist on GPUs only
    mailbox = None
    if "mailbox" in ops or "mailbox2" in ops:
        from distributed_tensorflow_amd.parallel.p2p import P2PAllReduce

        mailbox = P2PAllReduce(int(a.mailbox_max_mb * (1 << 20)))

    def sync():
        if use_cuda:
            torch.cuda.synchronize()
        dist.barrier()

    rows = []
    if rank == 0:
        print("# world %d, %s, %s, backend %s" % (world, a.dtype, device, dist.get_backend()))
        print("%-15s %12s %10s %12s %12s" % ("op", "bytes", "us/call", "alg GB/s", "bus GB/s"))
    for op in ops:
        for nbytes in sizes(a.min_kb, a.max_mb):
            n = nbytes // esz
            n -= n % max(world * 64, 64)
            if n <= 0:
                continue
            if op.startswith("mailbox") and (mailbox is None or n * esz > a.mailbox_max_mb * (1 << 20)):
                continue
            full = torch.ones(n, dtype=dtype, device=device)
            shard = torch.ones(n // world, dtype=dtype, device=device)

            def call():
                if op == "all_reduce":
                    dist.all_reduce(full)
                elif op == "reduce_scatter":
                    dist.reduce_scatter_tensor(shard, full)
                elif op == "all_gather":
                    dist.all_gather_into_tensor(full, shard)
                else:  # mailbox: one-shot form, mailbox2: two-shot form (direct reduce-scatter + all-gather)
                    mailbox.all_reduce_(full, algo="two_shot" if op == "mailbox2" else "one_shot")

            for _ in range(a.warmup):
                call()
                if op in ("all_reduce", "mailbox", "mailbox2"):
                    full.fill_(1.0)  # keep the values finite over many in-place sums
            sync()
            t0 = time.perf_counter()
            for _ in range(a.iters):
                call()
            sync()
            dt = (time.perf_counter() - t0) / a.iters
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            alg = n * esz / dt / 1e9
            row = {"op": op, "bytes": n * esz, "us": round(dt * 1e6, 2), "alg_GBps": round(alg, 2),
                   "bus_GBps": round(alg * bus_factor(op, world), 2)}
            rows.append(row)
            if rank == 0:
                print("%-15s %12d %10.1f %12.2f %12.2f" % (op, row["bytes"], row["us"], row["alg_GBps"], row["bus_GBps"]),
                      flush=True)
            del full, shard
    if rank == 0 and a.json:
        with open(a.json, "w") as f:
            json.dump({"world": world, "dtype": a.dtype, "device": str(device), "rows": rows}, f, indent=1)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
