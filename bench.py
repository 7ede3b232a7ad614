#!/usr/bin/env python
"""Headline benchmark: training throughput (images/sec, whole job) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--model resnet50|flower_cnn|inception_v3_head|inception_v3_finetune]
                    [--mode sync_dp|sharded_ps|async_ssp|async_ps] [--batch PER_GPU_BATCH]

One command starts the whole job: with ``--gpus N`` (N > 1) and no ``RANK`` / ``WORLD_SIZE`` in the environment this
process only launches N rank processes (one per GPU, ``LOCAL_RANK`` = GPU index, rendezvous on a free port of
127.0.0.1), relays rank 0's JSON line and exits with the first non-zero status of any rank.  The launcher never
imports torch and never touches a GPU.  Started under ``python -m torch.distributed.run`` instead, RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* are taken from the environment.  (The reference starts its job from one command
as well: recognition_flower/distributed/run.sh:3-6.)

Every timed step is a complete training step: forward, backward, gradient communication, fused optimizer update.
``--mode`` selects how the ranks exchange gradients: RCCL all-reduce (``sync_dp``), reduce-scatter + owner update +
all-gather (``sharded_ps``), delayed all-reduce (``async_ssp``) or the GPU-resident asynchronous parameter server
with one-sided pushes and pulls over xGMI (``async_ps``: the reference's ps/worker scheme,
recognition_flower/distributed/train_dataset.py:141-165).  Data is synthetic (5-class flower_photos-shaped images,
random-init weights); compute dtype is bf16 with fp32 master weights and fp32 accumulation.  Rank 0 prints one JSON
line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# per-GPU batches sized for 288 GB of HBM: ResNet-50 at 1024 holds ~40 GB of activations and runs ~13 % more
# images/s than at 256 (fixed per-layer costs and tile-quantisation tails amortise); same box, measured launch table
# for each: 2048 (56 GB) 17 312, 4096 (111 GB) 17 776 images/s; 6144 (166 GB) another +0.7 % (17 814 against 17 694 on one box).  At 4096 the stem output and the
# stage-1 activations are 3.3 G elements each (tools/experiments/check_big_batch.py, check_big_ops.py: the kernels'
# indices hold beyond 2^31).
DEFAULT_BATCH = {"resnet50": 4096, "flower_cnn": 1024, "inception_v3_head": 256, "inception_v3_finetune": 512}
IMAGE_SIZE = {"resnet50": 224, "flower_cnn": 224, "inception_v3_head": 299, "inception_v3_finetune": 299}
MODES = ("sync_dp", "sharded_ps", "async_ssp", "async_ps")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="resnet50", choices=sorted(DEFAULT_BATCH))
    ap.add_argument("--mode", default="sync_dp", choices=list(MODES))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default depends on the model)")
    ap.add_argument("--bucket_mb", type=float, default=0.0,
                    help="gradient bucket size in MiB (0: 32, or a quarter of the gradient for models under 64 MiB)")
    ap.add_argument("--overlap_gather", type=int, default=1,
                    help="sharded_ps: 1 = the parameter all-gathers of a step are waited for inside the next forward pass, "
                         "layer by layer; 0 = at the end of the step")
    ap.add_argument("--staleness", type=int, default=1,
                    help="async_ssp: delay in steps; async_ps: how far a worker may run ahead (-1 = unbounded, 0 = lock step)")
    ap.add_argument("--ps_apply", default="auto", choices=["auto", "per_push", "aggregate"],
                    help="async_ps: one optimizer step per arrived gradient (the reference's ps) or one per lock-step round")
    ap.add_argument("--ps_timeout", type=float, default=30.0,
                    help="async_ps: seconds after which a silent peer stops counting")
    ap.add_argument("--grad_dtype", default="fp32", choices=["fp32", "bf16"],
                    help="dtype of the gradient buckets (what the collectives move): bf16 halves the bytes on xGMI")
    ap.add_argument("--p2p_kb", type=int, default=0,
                    help="gradient buckets up to this many KiB use the one-shot xGMI mailbox all-reduce (sync_dp)")
    ap.add_argument("--trunk_bn", default="batch", choices=["batch", "moving"],
                    help="inception_v3_*: trunk batch-norm on batch statistics (reference) or frozen moving statistics")
    ap.add_argument("--also", default="flower_cnn",
                    help="second model measured after the headline one and reported under \"also\" ('' = none)")
    ap.add_argument("--also_batch", type=int, default=0, help="per-GPU batch of the second model (default: its own)")
    ap.add_argument("--also_modes", default="auto",
                    help="distribution modes the headline model is measured in as well, reported under \"also\" ('' = none; "
                         "auto = sharded_ps,async_ssp and, on one GPU, async_ps: across GPUs the GPU-resident parameter "
                         "server maps peer memory between devices, which no box this tree was tested on has, so there it "
                         "runs only when named -- as --mode async_ps or in this list -- and cannot cost the headline record)")
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"],
                    help="cpu: rehearse the multi-rank control flow over gloo (not a measurement)")
    ap.add_argument("--image_size", type=int, default=0, help="square image size (default: the model's)")
    ap.add_argument("--also_timeout", type=int, default=240,
                    help="seconds the secondary (\"also\") runs may take before the headline result is reported "
                         "without them and the process ends with status 3 (0 = no limit)")
    ap.add_argument("--hip_graph", type=int, default=-1,
                    help="1: record the whole step (fwd, bwd, collectives, optimizer) in a HIP graph and replay it; "
                         "default: on for inception_v3_head (hundreds of short kernels per step), off otherwise")
    ap.add_argument("--rccl_channels", type=int, default=0,
                    help="channels RCCL runs side by side (NCCL_MIN/MAX_NCHANNELS; 0: RCCL's own choice for the topology)")
    ap.add_argument("--launch_timeout", type=int, default=0,
                    help="launcher: seconds after which the rank processes are stopped (0 = no limit)")
    args = ap.parse_args(argv)
    if args.hip_graph < 0:
        args.hip_graph = 1 if args.model == "inception_v3_head" else 0
    return args


# ====================================================================================================== launcher
def launch_ranks(args, argv) -> int:
    """Start ``args.gpus`` rank processes of this script and wait for them.  Runs in a process that has not
    imported torch: nothing here initialises the GPU, the children are fresh interpreters."""
    import signal
    import socket
    import subprocess
    import threading

    n = args.gpus
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = os.path.abspath(__file__)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DTFA_BENCH_LAUNCHED="1")
        # rank 0's stdout carries the result line and is relayed; the other ranks' stdout joins stderr
        # (the ranks stay in this process's group, so whoever stops the launcher's group stops them too)
        procs.append(subprocess.Popen([sys.executable, "-u", script] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))

    def relay():
        for line in procs[0].stdout:
            sys.stdout.write(line.decode("utf-8", "replace"))
            sys.stdout.flush()

    t = threading.Thread(target=relay, daemon=True)
    t.start()

    def stop_all():
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 10
        for p in procs:
            try:
                p.wait(max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()

    def on_term(signum, frame):
        raise KeyboardInterrupt

    signal.signal(signal.SIGTERM, on_term)

    t0 = time.time()
    status = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                r, c = bad[0]
                print("bench.py launcher: rank %d exited with status %d; stopping the other ranks" % (r, c), file=sys.stderr)
                status = c if c > 0 else 128 - c
                break
            if all(c == 0 for c in codes):
                break
            if args.launch_timeout and time.time() - t0 > args.launch_timeout:
                print("bench.py launcher: ranks still running after %d s; stopping them" % args.launch_timeout, file=sys.stderr)
                status = 124
                break
            time.sleep(0.05)
    except KeyboardInterrupt:
        status = 130
    finally:
        stop_all()
        t.join(5)
    return status


# ====================================================================================================== one rank
def mode_learning_rate(mode, staleness, world, base):
    """Step size a distribution mode is run with.  Gradients applied late narrow the stable range under heavy-ball
    momentum mu: with one step of delay, on a quadratic of curvature c the iteration
    w' = (1 + mu) w - (mu + lr c) w_prev is stable only for lr c < 1 - mu, against lr c < 2 (1 + mu) without the
    delay -- a 38x smaller range at mu = 0.9.  Measured on the fixed synthetic batch: 0.02 and 0.002 oscillate
    upwards under a delay, 0.0005 descends.  ``async_ssp`` delays every gradient by ``staleness`` steps at any world
    size; ``async_ps`` applies a gradient computed on parameters that are up to world - 1 pushes old (none in a
    world of one, or in lock step)."""
    delayed = (mode == "async_ssp" and staleness > 0) or (mode == "async_ps" and staleness != 0 and world > 1)
    return base / 40 if delayed else base


def run_config(args, model_name, batch, device, rank, world, mode=None):
    """W untimed + K timed complete training steps of one model; returns the whole-job images/s (max time over ranks)."""
    import torch
    import torch.distributed as dist

    from distributed_tensorflow_amd.models import build_model
    from distributed_tensorflow_amd.ops import functional as F

    mode = mode or args.mode
    size = args.image_size or IMAGE_SIZE[model_name]
    torch.manual_seed(1234)
    kw = {"trunk_bn": args.trunk_bn} if model_name.startswith("inception_v3") else {}
    model = build_model(model_name, num_classes=5, device=device, **kw)
    model.train()
    if model_name.startswith("inception_v3"):
        opt, lr, okw = "adam", 1e-3, {}
    else:
        # a fixed synthetic batch with random labels: a small constant rate keeps the loss finite at any world size
        opt, okw = "momentum", {"momentum": 0.9, "weight_decay": 1e-4}
        lr = mode_learning_rate(mode, args.staleness, world, 0.02)
    grad_dtype = {"fp32": torch.float32, "bf16": torch.bfloat16}[args.grad_dtype]
    if mode == "async_ps":
        from distributed_tensorflow_amd.parallel.xps import XgmiPSEngine

        engine = XgmiPSEngine(model, optimizer=opt, lr=lr, opt_kwargs=okw, staleness=args.staleness, apply=args.ps_apply,
                              grad_dtype=grad_dtype, timeout=args.ps_timeout)
    else:
        from distributed_tensorflow_amd.parallel.engine import DataParallelEngine

        engine = DataParallelEngine(model, mode=mode, optimizer=opt, lr=lr, opt_kwargs=okw,
                                    bucket_mb=args.bucket_mb, staleness=args.staleness, p2p_max_bytes=args.p2p_kb * 1024,
                                    grad_dtype=grad_dtype, overlap_gather=bool(args.overlap_gather))
    engine.measure_comm = not args.hip_graph  # (events cannot be timed inside a captured graph)

    # synthetic batch generated on the device (a 2048 x 224 x 224 host tensor per rank would cost seconds and GBs)
    g = torch.Generator(device=device).manual_seed(1000 + rank)
    # image layout as the input pipeline delivers it: NHWC, 3 real channels padded to 8, or to 4 for models with the
    # pixel-pair stem (ResNet-50 on the device, data.preprocess.resize_on_device(channels=4))
    cpad = getattr(model, "input_channels", 8)
    act = torch.bfloat16 if device.type == "cuda" else torch.float32
    images = torch.zeros(batch, size, size, cpad, device=device, dtype=act)
    images[..., :3] = (torch.rand(batch, size, size, 3, device=device, generator=g) * 2 - 1).to(act)
    labels = torch.randint(0, 5, (batch,), device=device, generator=g).to(torch.int32)

    def step():
        engine.begin_step()
        out = model(images)
        logits, reg = out if isinstance(out, tuple) else (out, None)
        loss, _, _, _ = F.softmax_xent(logits, labels, 5)
        total = loss if reg is None else loss + reg
        total.backward()
        engine.finish_step()
        return loss

    def sync():
        if dist.is_initialized():
            dist.barrier()
        if device.type == "cuda":
            torch.cuda.synchronize()

    if args.hip_graph:
        from distributed_tensorflow_amd.train.graph import StepGraph

        def loss_fn(im, lb):
            out = model(im)
            logits, reg = out if isinstance(out, tuple) else (out, None)
            loss, _, _, _ = F.softmax_xent(logits, lb, 5)
            return (loss if reg is None else loss + reg), loss

        for _ in range(2):  # eager steps create plans, workspaces and communicators before the capture
            step()
        sg = StepGraph(engine, loss_fn, images, labels)

        def step():  # noqa: F811
            return sg.run(images, labels)[1]

    first_loss = None
    for i in range(args.warmup):
        loss = step()
        if i == 0:   # (read outside the timed region: shows next to final_loss whether the run descends)
            first_loss = round(float(loss.item()), 4)
    sync()
    engine.reset_comm_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = t.item()
    ips = batch * world * args.steps / dt
    final_loss = round(float(loss.item()), 4)
    comm = engine.comm_stats()
    if args.hip_graph:
        # A captured step holds the communicator's kernels: release the graph here, with the device idle, and not
        # whenever the garbage collector finds the closure cycle -- possibly after the process group is gone, which
        # the communicator's watchdog thread answers with an abort.
        step = None
        sg.graph = None
        sg = None
        import gc
        gc.collect()
        if device.type == "cuda":
            torch.cuda.synchronize()
    if hasattr(engine, "close"):
        if dist.is_initialized():
            dist.barrier()   # (nobody leaves while a peer still pushes to it or pulls from it)
        engine.close()
    return {"value": round(ips, 2), "ms_per_step": round(dt / args.steps * 1e3, 3), "final_loss": final_loss,
            "first_loss": first_loss, "learning_rate": lr,
            "per_gpu_batch": batch, "global_batch": batch * world, "optimizer": opt,
            "peak_mem_gb": round(torch.cuda.max_memory_allocated(device) / 2**30, 2) if device.type == "cuda" else 0.0,
            "comm": comm}


ALSO_KEYS = ("value", "ms_per_step", "per_gpu_batch", "global_batch", "first_loss", "final_loss", "learning_rate")


def run_rank(args) -> int:
    import gc
    import signal
    import threading

    import torch
    import torch.distributed as dist

    from distributed_tensorflow_amd.parallel.rccl import configure_rccl

    rccl_vars = configure_rccl(args.rccl_channels)   # (before the process group exists)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ  # under a launcher, any N
    if args.device == "cpu":
        # plumbing rehearsal only (tests/test_parallel.py): the same control flow over gloo on fp32 CPU tensors
        device = torch.device("cpu")
        if world > 1 or launched:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X GPU")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit("rank %d: LOCAL_RANK %d but only %d GPU(s) are visible (one rank per GPU)"
                             % (rank, local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        if world > 1 or launched:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    if world != args.gpus:
        # a record that says n_gpus = 1 for a run asked to use 8 would be worthless: refuse instead of warning
        raise SystemExit("bench.py: --gpus %d but the job has %d rank(s); start it as `python bench.py --gpus %d` "
                         "(it launches the ranks itself) or under torch.distributed.run with --nproc-per-node %d"
                         % (args.gpus, world, args.gpus, args.gpus))
    devices = [local_rank]
    if dist.is_initialized():
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        gathered = [None] * world
        dist.all_gather_object(gathered, local_rank if device.type == "cuda" else -1)
        devices = gathered

    head = run_config(args, args.model, args.batch or DEFAULT_BATCH[args.model], device, rank, world)

    also = {}
    report_lock = threading.Lock()
    reported = []

    def report():
        with report_lock:   # (the watchdog and the main thread may get here together: one line, once)
            if reported or rank != 0:
                return
            reported.append(True)
            secondary = dict(also)
        print(json.dumps({
            "metric": "train images/sec (whole node)",
            "value": head["value"],
            "unit": "images/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16" if device.type == "cuda" else "fp32 (cpu rehearsal)",
            "data": "synthetic",
            "final_loss": head["final_loss"],
            "first_loss": head["first_loss"],
            "config": {"model": args.model, "global_batch": head["global_batch"], "per_gpu_batch": head["per_gpu_batch"],
                       "image_size": args.image_size or IMAGE_SIZE[args.model], "num_classes": 5, "parallelism": "dp%d" % world,
                       "mode": args.mode, "optimizer": head["optimizer"], "learning_rate": head["learning_rate"],
                       "hip_graph": bool(args.hip_graph)},
            "peak_mem_gb": head["peak_mem_gb"],
            # communication of the headline run: size of the process group (RCCL on GPUs), the GPU index of every
            # rank, gradient buckets (count / bytes each / dtype) and the time per step the compute stream spent
            # waiting for collectives that backward did not hide
            "rccl_world": dist.get_world_size() if dist.is_initialized() else 0,
            "rank_devices": devices,
            "rccl_env": rccl_vars,   # every NCCL_* / RCCL_* variable in force (empty: the library's defaults)
            "comm": head["comm"],
            "also": secondary,
        }), flush=True)

    # The headline number is measured; the secondary runs below must not be able to lose it.  If one of them
    # hangs (a rank that failed inside a collective leaves its peers waiting), the watchdog reports the headline
    # with whatever secondary results exist and ends the process with a non-zero status: the hang is on record.
    done = threading.Event()

    def watchdog():
        if not done.wait(args.also_timeout):
            with report_lock:
                also["error"] = "secondary runs exceeded %d s and were abandoned" % args.also_timeout
            report()
            sys.stdout.flush()
            os._exit(3)

    def put(key, value):
        with report_lock:
            also[key] = value

    if args.also_timeout > 0:
        threading.Thread(target=watchdog, daemon=True).start()
    if rank == 0 and threading.current_thread() is threading.main_thread():
        # a launcher that loses another rank during the secondary runs ends this one with SIGTERM: the headline
        # result is printed first, then the process goes with a failing status
        def on_term(signum, frame):
            with report_lock:
                also.setdefault("error", "terminated during the secondary runs (signal %d)" % signum)
            report()
            sys.stdout.flush()
            os._exit(3)

        signal.signal(signal.SIGTERM, on_term)
    if args.also and args.also != args.model:
        # the second model of the headline metric (flower CNN), same K / W, measured after the headline run
        gc.collect()
        if device.type == "cuda":
            torch.cuda.empty_cache()
        try:
            r = run_config(args, args.also, args.also_batch or DEFAULT_BATCH[args.also], device, rank, world)
            put(args.also, {k: r[k] for k in ALSO_KEYS})
        except Exception as e:  # noqa: BLE001 -- the headline number stands on its own
            put(args.also, {"error": "%s: %s" % (type(e).__name__, e)})
    # the headline model in the other distribution modes (sharded parameter ownership: reduce-scatter + all-gather;
    # staleness-bounded delayed all-reduce; the GPU-resident asynchronous parameter server), same batch, same K / W
    also_modes = args.also_modes
    if also_modes == "auto":
        also_modes = "sharded_ps,async_ssp" + (",async_ps" if world == 1 else "")
    for other in [m for m in also_modes.split(",") if m and m != args.mode and not args.hip_graph]:
        if other == "async_ps" and device.type != "cuda":
            continue   # (its parameters live in GPU memory)
        gc.collect()
        if device.type == "cuda":
            torch.cuda.empty_cache()
        key = "%s/%s" % (args.model, other)
        try:
            r = run_config(args, args.model, head["per_gpu_batch"], device, rank, world, mode=other)
            d = {k: r[k] for k in ALSO_KEYS}
            if other == "async_ps":
                d["comm"] = r["comm"]
            put(key, d)
        except Exception as e:  # noqa: BLE001
            put(key, {"error": "%s: %s" % (type(e).__name__, e)})
    done.set()
    report()
    if dist.is_initialized():
        gc.collect()
        if device.type == "cuda":
            torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
