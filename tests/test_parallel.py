"""Collective data-parallel engine on CPU tensors with gloo (world sizes 2, 4 and 8).

The same engine code runs over RCCL on GPUs; these tests pin down the
arithmetic: gradient averaging, equivalence of the all-reduce and the
reduce-scatter / all-gather (sharded parameter ownership) formulations, and
the staleness semantics of the delayed-apply mode.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from distributed_tensorflow_amd.models import build_model
from distributed_tensorflow_amd.ops import functional as F
from distributed_tensorflow_amd.parallel.engine import DataParallelEngine
from distributed_tensorflow_amd.parallel.flat import FlatParams

WIDTHS = (8, 8, 16, 16, 16)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(rank, step, size=32, n=4):
    g = torch.Generator().manual_seed(1000 * rank + step)
    x = torch.zeros(n, size, size, 8)
    x[..., :3] = torch.randn(n, size, size, 3, generator=g)
    return x, torch.randint(0, 5, (n,), generator=g)


def _model():
    torch.manual_seed(7)
    return build_model("flower_cnn", widths=WIDTHS, dropout_rate=0.0)


def _loss(model, x, y):
    logits, reg = model(x)
    loss, _, _, _ = F.softmax_xent(logits, y, 5)
    return loss + reg


def _worker(rank, world, port, mode, opt, steps, staleness, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _model()
        if rank == 1:  # replicas must be synchronised from rank 0 by the engine
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(1.0)
        okw = {"momentum": 0.9, "weight_decay": 1e-3} if opt == "momentum" else {}
        blocking = mode.endswith("_blocking")   # sharded mode with every all-gather waited for inside finish_step
        eng = DataParallelEngine(model, mode=mode.replace("_blocking", ""), optimizer=opt, lr=0.05 if opt == "momentum" else 1e-2,
                                 opt_kwargs=okw, bucket_mb=0.02, staleness=staleness, overlap_gather=not blocking)
        if mode == "sharded_ps":
            assert eng.overlap_gather and len(eng.flat.buckets()) > 3
        for t in range(steps):
            eng.begin_step()
            _loss(model, *_batch(rank, t)).backward()
            eng.finish_step()
            if mode == "sharded_ps" and world > 1:
                # the parameter all-gathers are still in flight: the next forward's layers wait for them one by one
                assert len(eng._pending_gather) == len(eng.flat.buckets())
        eng.drain()
        eng.full_master()  # (sharded mode: gathers the owners' fp32 shards back into FlatGroup.master)
        # compare parameter by parameter: bucket padding differs between modes, values must not
        flat = torch.cat([g.master[s.offset:s.offset + s.numel].clone() for g in eng.flat.groups for s in g.segments])
        compute = torch.cat([p.detach().float().reshape(-1).clone() for p in eng.flat.params])
        torch.save({"flat": flat, "compute": compute}, os.path.join(out_dir, "%s_%s_r%d.pt" % (mode, opt, rank)))
    finally:
        dist.destroy_process_group()


def _spawn(mode, opt, steps, staleness, out_dir, world=2):
    mp.spawn(_worker, args=(world, _free_port(), mode, opt, steps, staleness, out_dir), nprocs=world, join=True)
    return [torch.load(os.path.join(out_dir, "%s_%s_r%d.pt" % (mode, opt, r))) for r in range(world)]


def _reference(opt, steps, delay=0, world=2):
    """Single process: per-replica gradients computed separately (each replica has its own
    batch-norm statistics), averaged, applied -- optionally ``delay`` steps late."""
    models = [_model() for _ in range(world)]
    okw = {"momentum": 0.9, "weight_decay": 1e-3} if opt == "momentum" else {}
    engs = [DataParallelEngine(m, mode="sync_dp", optimizer=opt, lr=0.05 if opt == "momentum" else 1e-2,
                               opt_kwargs=okw, bucket_mb=0.02) for m in models]
    pending = []
    for t in range(steps):
        grads = []
        for r, (m, e) in enumerate(zip(models, engs)):
            e.begin_step()
            _loss(m, *_batch(r, t)).backward()
            e.flat.collect_autograd_grads()
            grads.append([g.grad.clone() for g in e.flat.groups])
        pending.append([sum(gs) for gs in zip(*grads)])
        if len(pending) > delay:
            avg = pending.pop(0)
            for e in engs:
                for g, s in zip(e.flat.groups, avg):
                    g.grad.copy_(s)
                e.optimizer.step(grad_scale=1.0 / world)
    for avg in pending:
        for e in engs:
            for g, s in zip(e.flat.groups, avg):
                g.grad.copy_(s)
            e.optimizer.step(grad_scale=1.0 / world)
    return torch.cat([g.master[s.offset:s.offset + s.numel] for g in engs[0].flat.groups for s in g.segments])


def test_overlapped_parameter_gather_equals_the_blocking_one(tmp_path):
    """sharded_ps at world 4: all-gathers left in flight after ``finish_step`` and waited for layer by layer in the
    next forward (ops.functional.param_fence) give bit for bit the parameters of gathers waited for on the spot."""
    a = _spawn("sharded_ps", "momentum", 4, 0, str(tmp_path), world=4)
    b = _spawn("sharded_ps_blocking", "momentum", 4, 0, str(tmp_path), world=4)
    for r in range(4):
        assert torch.equal(a[r]["flat"], b[r]["flat"]) and torch.equal(a[r]["compute"], b[r]["compute"])
        assert torch.equal(a[r]["compute"], a[0]["compute"])


@pytest.mark.parametrize("opt", ["momentum", "adam"])
def test_sync_dp_matches_manual_gradient_averaging(tmp_path, opt):
    res = _spawn("sync_dp", opt, 3, 0, str(tmp_path))
    torch.testing.assert_close(res[0]["flat"], res[1]["flat"], rtol=0, atol=0)  # replicas identical
    torch.testing.assert_close(res[0]["flat"], _reference(opt, 3), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("opt", ["momentum", "adam"])
def test_sharded_ps_is_bitwise_equal_to_allreduce_dp(tmp_path, opt):
    a = _spawn("sync_dp", opt, 3, 0, str(tmp_path))
    b = _spawn("sharded_ps", opt, 3, 0, str(tmp_path))
    # reduce-scatter + owner update + all-gather == all-reduce + replicated update, bit for bit in fp32
    torch.testing.assert_close(b[0]["flat"], a[0]["flat"], rtol=0, atol=0)
    torch.testing.assert_close(b[0]["compute"], b[1]["compute"], rtol=0, atol=0)
    torch.testing.assert_close(b[0]["compute"], a[0]["compute"], rtol=0, atol=0)


def test_async_ssp_applies_gradients_exactly_s_steps_late(tmp_path):
    s0 = _spawn("async_ssp", "momentum", 4, 0, str(tmp_path))
    sync = _spawn("sync_dp", "momentum", 4, 0, str(tmp_path))
    torch.testing.assert_close(s0[0]["flat"], sync[0]["flat"], rtol=0, atol=0)  # s = 0 degenerates to sync
    s2 = _spawn("async_ssp", "momentum", 4, 2, str(tmp_path))
    torch.testing.assert_close(s2[0]["flat"], s2[1]["flat"], rtol=0, atol=0)
    torch.testing.assert_close(s2[0]["flat"], _reference("momentum", 4, delay=2), rtol=1e-5, atol=1e-6)
    assert not torch.equal(s2[0]["flat"], sync[0]["flat"])


@pytest.mark.parametrize("world", [4, 8])
def test_all_modes_at_larger_world_sizes(tmp_path, world):
    """The three collective modes at 4 and 8 ranks (the size of an MI355X node): replicas identical, the
    all-reduce result equals manual averaging over ``world`` replicas, sharded ownership is bitwise equal to it
    (1/world of every bucket per rank, stepped, not just laid out), delayed apply equals the delayed reference."""
    sync = _spawn("sync_dp", "momentum", 3, 0, str(tmp_path), world)
    for r in range(1, world):
        torch.testing.assert_close(sync[r]["flat"], sync[0]["flat"], rtol=0, atol=0)
    # gloo reduces in a ring / tree order of its own: equal to the sequential sum up to fp32 rounding
    torch.testing.assert_close(sync[0]["flat"], _reference("momentum", 3, world=world), rtol=2e-5, atol=2e-6)
    shard = _spawn("sharded_ps", "momentum", 3, 0, str(tmp_path), world)
    for r in range(world):
        torch.testing.assert_close(shard[r]["compute"], shard[0]["compute"], rtol=0, atol=0)
    # same arithmetic, reduction order of reduce-scatter vs all-reduce may differ in gloo beyond two ranks
    torch.testing.assert_close(shard[0]["flat"], sync[0]["flat"], rtol=2e-5, atol=2e-6)
    ssp = _spawn("async_ssp", "momentum", 4, 1, str(tmp_path), world)
    for r in range(1, world):
        torch.testing.assert_close(ssp[r]["flat"], ssp[0]["flat"], rtol=0, atol=0)
    torch.testing.assert_close(ssp[0]["flat"], _reference("momentum", 4, delay=1, world=world), rtol=2e-5, atol=2e-6)


def test_async_ssp_drain_then_continue_applies_every_gradient_once():
    """A drain in the middle of a run (what a periodic checkpoint does) applies the in-flight gradients early;
    the following steps must not apply them a second time."""
    steps, s = 6, 2
    model = _model()
    eng = DataParallelEngine(model, mode="async_ssp", optimizer="momentum", lr=0.05, staleness=s, bucket_mb=0.02)
    applied = []
    for t in range(steps):
        eng.begin_step()
        _loss(model, *_batch(0, t)).backward()
        eng.finish_step()
        if t in (2, 3):
            eng.drain()
            assert eng.optimizer.step_count == t + 1  # everything up to this step has landed, once
        applied.append(eng.optimizer.step_count)
    eng.drain()
    assert eng.optimizer.step_count == steps
    # between drains the schedule is the delayed one again: step 4 applies nothing new (its own gradient waits),
    # step 5 still nothing (s = 2), the final drain lands both
    assert applied == [0, 0, 3, 4, 4, 4]


def _ssp_ckpt_worker(rank, world, port, out_dir):
    from distributed_tensorflow_amd.io.saver import Saver
    from distributed_tensorflow_amd.train.session import CheckpointSaverHook, TrainingSession

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _model()
        eng = DataParallelEngine(model, mode="async_ssp", optimizer="momentum", lr=0.05, staleness=1, bucket_mb=0.02)
        saver = Saver(model, eng)
        hooks = [CheckpointSaverHook(os.path.join(out_dir, "ckpt"), 2, saver, background=False)]
        sess = TrainingSession(model, eng, num_classes=5, hooks=hooks, is_chief=rank == 0)
        with sess:
            for t in range(5):
                sess.run_step(*_batch(rank, t))
        eng.drain()
        flat = torch.cat([g.master[s.offset:s.offset + s.numel].clone() for g in eng.flat.groups for s in g.segments])
        torch.save({"flat": flat, "steps": eng.optimizer.step_count}, os.path.join(out_dir, "ssp_ckpt_r%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def test_async_ssp_periodic_checkpoints_keep_replicas_identical(tmp_path):
    """Checkpoints every 2 steps (the reference's cadence) drain the in-flight gradients; that must happen on
    every rank at the same step, not on the chief alone."""
    mp.spawn(_ssp_ckpt_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = [torch.load(os.path.join(str(tmp_path), "ssp_ckpt_r%d.pt" % r)) for r in range(2)]
    assert a["steps"] == b["steps"] == 5
    torch.testing.assert_close(a["flat"], b["flat"], rtol=0, atol=0)
    assert os.path.exists(os.path.join(str(tmp_path), "ckpt", "checkpoint"))


def test_async_ssp_first_s_steps_leave_parameters_untouched(monkeypatch):
    monkeypatch.setenv("DTFA_DEBUG_STALENESS", "1")  # every apply asserts "exactly s steps old" on the slot versions
    model = _model()
    eng = DataParallelEngine(model, mode="async_ssp", optimizer="momentum", lr=0.05, staleness=2, bucket_mb=0.02)
    before = torch.cat([g.master.clone() for g in eng.flat.groups])
    for t in range(2):
        eng.begin_step()
        _loss(model, *_batch(0, t)).backward()
        eng.finish_step()
        torch.testing.assert_close(torch.cat([g.master for g in eng.flat.groups]), before, rtol=0, atol=0)
    eng.begin_step()
    _loss(model, *_batch(0, 2)).backward()
    eng.finish_step()  # step 0's gradient lands now
    assert not torch.equal(torch.cat([g.master for g in eng.flat.groups]), before)


def test_flat_params_bucket_layout():
    model = _model()
    flat = FlatParams(model, bucket_mb=0.02, shard_multiple=8)
    for g in flat.groups:
        assert g.total % (8 * 64) == 0
        prev_end = 0
        for b in g.buckets:
            assert b.start == prev_end and b.numel % (8 * 64) == 0
            prev_end = b.end
            for s in b.segments:
                assert b.start <= s.offset and s.offset + s.numel <= b.end and s.offset % 64 == 0
        assert prev_end == g.total
        assert len(g.buckets) > 1
    # parameters are views into the flat buffers, laid out last-layer-first
    first = flat.groups[0].segments[0]
    assert first.name.startswith("head.")
    p = dict(model.named_parameters())[first.name]
    assert p.data_ptr() == flat.groups[0].compute[first.offset:].data_ptr()
    assert p.main_grad.data_ptr() == flat.groups[0].grad[first.offset:].data_ptr()


def test_bucket_ready_callbacks_fire_once_per_bucket():
    model = _model()
    flat = FlatParams(model, bucket_mb=0.02)
    fired = []
    flat.on_bucket_ready = lambda b: fired.append((b.group.name, b.index))
    flat.begin_step()
    _loss(model, *_batch(0, 0)).backward()
    flat.collect_autograd_grads()
    assert sorted(fired) == sorted((g.name, b.index) for g in flat.groups for b in g.buckets)
    assert len(set(fired)) == len(fired)


def _run_bench_cpu(world, extra, timeout=300, expect_status=0):
    """bench.py launched the way the driver launches it (one rank per device under torch.distributed.run), over gloo
    on CPU tensors: the control flow of a multi-rank run -- rendezvous, barriers around the timed region, the MAX over
    ranks, the secondary runs, one JSON line from rank 0 -- rehearsed without GPUs."""
    import json
    import subprocess
    import sys

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(world), "--device", "cpu",
           "--model", "flower_cnn", "--batch", "4", "--image_size", "32", "--steps", "2", "--warmup", "1"] + extra
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, cwd=root)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    # (torch.distributed.run reports any failed rank as status 1: ``expect_status=None`` asks for "not zero")
    ok = r.returncode != 0 if expect_status is None else r.returncode == expect_status
    assert ok and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    return json.loads(lines[0])


def _run_bench_self_launched(args, timeout=300):
    """``python bench.py --gpus N ...`` with no launcher and no RANK / WORLD_SIZE in the environment: bench.py starts
    its rank processes itself."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=timeout, cwd=root, env=env)


def test_bench_starts_its_own_ranks():
    """One command launches the job (the reference: distributed/run.sh): ``--gpus 2`` without a launcher trains on two
    ranks and says so, instead of silently training on one."""
    import json

    r = _run_bench_self_launched(["--gpus", "2", "--device", "cpu", "--model", "flower_cnn", "--batch", "4", "--image_size",
                                  "32", "--steps", "2", "--warmup", "1", "--also", "", "--also_modes", ""])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_world"] == 2 and d["config"]["parallelism"] == "dp2"
    assert d["config"]["global_batch"] == 8 and len(d["rank_devices"]) == 2
    assert d["comm"]["collective"] and d["comm"]["world"] == 2


def test_bench_launcher_reports_a_failed_rank():
    """A rank that fails ends the job with a non-zero status and no result line (here the ranks ask for GPUs on a
    machine without one)."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs are visible: the ranks would start")
    r = _run_bench_self_launched(["--gpus", "2", "--device", "cuda", "--steps", "1", "--warmup", "0", "--also", "",
                                  "--also_modes", ""], timeout=120)
    assert r.returncode != 0 and "{" not in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-500:])


def test_bench_refuses_a_world_that_is_not_the_one_asked_for():
    """``--gpus 4`` inside a job of two ranks must not produce a record: it ends with an error instead."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--device", "cpu", "--model",
                        "flower_cnn", "--batch", "4", "--image_size", "32", "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120, cwd=root, env=env)
    assert r.returncode != 0 and "--gpus 4" in r.stderr and "{" not in r.stdout, (r.returncode, r.stderr[-500:])


def test_bench_two_ranks_over_gloo():
    d = _run_bench_cpu(2, ["--also", "", "--also_modes", "sharded_ps,async_ssp"])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["rccl_world"] == 2
    assert d["config"]["global_batch"] == 8 and d["config"]["parallelism"] == "dp2" and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] - 8 * 1e3 / d["ms_per_step"]) <= 0.02 * d["value"]
    assert d["comm"]["collective"] and d["comm"]["collectives_waited"] == 2 * d["comm"]["buckets"]
    for mode in ("sharded_ps", "async_ssp"):
        assert d["also"]["flower_cnn/" + mode]["value"] > 0, d["also"]


def test_bench_reports_headline_when_secondary_runs_exceed_their_time():
    """The watchdog: secondary runs that do not finish in ``--also_timeout`` seconds are abandoned and the headline
    result is still printed (exactly once); the ranks end with status 3, so that the abandoned runs are on record."""
    d = _run_bench_cpu(2, ["--also", "resnet50", "--also_batch", "16", "--also_modes", "sharded_ps", "--also_timeout", "1"],
                       expect_status=None)
    assert d["value"] > 0 and "exceeded" in d["also"].get("error", ""), d["also"]


def test_launch_geometry_rules():
    """Pure host logic of the two "whole rounds" rules (ops/conv.py): the tail split of macro-tile launches and the
    weight-gradient split counts."""
    from distributed_tensorflow_amd.ops import conv as cv

    # 7x7 512 -> 512 3x3 at batch 2048: 392 x 2 tiles of 256 x 256 = 3.06 rounds of 256 -> three rounds + a tail
    M = 2048 * 49
    tile = cv.pick_fwd_tile(M, 512, 4608, True)
    assert tile[:2] == (256, 256)
    m_main, tail = cv.split_tail(M, 512, 4608, tile[0], tile[1], True)
    assert m_main == 3 * 256 // 2 * 256 and tail is not None and tail[0] < 256
    assert m_main % tail[0] == 0
    # exactly three rounds: nothing to split; a last round that is mostly full: left alone
    assert cv.split_tail(m_main, 512, 4608, 256, 256, True) == (m_main, None)
    assert cv.split_tail(m_main + 200 * 256 // 2 * 256 // 256 * 256, 512, 4608, 256, 256, True)[1] is None
    # less than one full round, or a tile that is not a macro tile: one launch
    assert cv.split_tail(100 * 256, 512, 4608, 256, 256, True)[1] is None
    assert cv.split_tail(M, 512, 4608, 128, 128, True)[1] is None
    # weight gradient of a layer with taps: tiles x splits is a whole number of rounds of resident workgroups
    for M_, cout, cin in [(2048 * 56 * 56, 64, 64), (2048 * 28 * 28, 128, 128), (2048 * 14 * 14, 256, 256), (2048 * 49, 512, 512)]:
        bmw, bnw, bkp, splits = cv.pick_wgrad_tile(M_, cout, 9 * cin, cin)
        tiles = -(-cout // bmw) * -(-9 * cin // bnw)
        per_cu = cv._WGRAD_RESIDENT[(bmw, bnw, bkp)][0]
        slots = 256.0 * per_cu
        rounds = tiles * splits / slots
        whole = -(-tiles * splits // int(slots))
        # as many splits as fit into a whole number of rounds, in whole splits per XCD (the kernel deals them over the 8
        # XCDs round-robin): eight more splits would start another round
        assert splits < 16 or splits % 8 == 0, splits
        assert 1 <= whole <= 4 and 0 <= whole - rounds < 8 * tiles / slots, (cout, bmw, bnw, splits, rounds)


def test_automatic_bucket_size_cuts_small_models_in_several_buckets():
    """bucket_mb = 0: a model with less than 64 MiB of gradient gets buckets of a quarter of it (at least 1 MiB), so
    most of its reduction overlaps with backward; an explicit size is taken as given."""
    import torch.nn as nn
    from distributed_tensorflow_amd.parallel.engine import DataParallelEngine

    def make():
        torch.manual_seed(0)
        return nn.Sequential(*[nn.Linear(512, 512) for _ in range(8)])   # 8 x 1 MiB

    auto = DataParallelEngine(make(), mode="sync_dp", optimizer="momentum", lr=0.1, bucket_mb=0)
    one = DataParallelEngine(make(), mode="sync_dp", optimizer="momentum", lr=0.1, bucket_mb=32.0)
    a, o = auto.comm_stats(), one.comm_stats()
    assert o["buckets"] == 1 and a["buckets"] >= 4
    assert max(a["bucket_bytes"]) <= 2.1 * (1 << 20) + 4 * 512 * 4
    assert a["gradient_bytes"] == o["gradient_bytes"]


def test_measured_bucket_table_feeds_the_automatic_bucket_size(tmp_path, monkeypatch):
    """parallel/rccl.py: a recorded measurement for (world, mode, gradient size) is what ``bucket_mb=0`` resolves to;
    other worlds / modes / far-away model sizes fall back to the rule; RCCL channel overrides reach the environment."""
    from distributed_tensorflow_amd.parallel import rccl

    monkeypatch.setattr(rccl, "_TABLE", str(tmp_path / "tuned_buckets.json"))
    assert rccl.tuned_bucket_mb(8, "sync_dp", 90.0) is None
    rccl.record_bucket(8, "sync_dp", "resnet50", 89.8, 16.0, [{"bucket_mb": 16.0, "images_per_sec": 1.0}])
    rccl.record_bucket(8, "sync_dp", "flower_cnn", 8.0, 2.0, [])
    assert rccl.tuned_bucket_mb(8, "sync_dp", 95.0) == 16.0
    assert rccl.tuned_bucket_mb(8, "sync_dp", 7.0) == 2.0
    assert rccl.tuned_bucket_mb(8, "sync_dp", 400.0) is None
    assert rccl.tuned_bucket_mb(4, "sync_dp", 90.0) is None and rccl.tuned_bucket_mb(8, "sharded_ps", 90.0) is None
    rccl.record_bucket(8, "sync_dp", "resnet50", 89.8, 64.0, [])     # a new measurement replaces the old one
    assert rccl.tuned_bucket_mb(8, "sync_dp", 90.0) == 64.0
    for k in ("NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS"):
        monkeypatch.delenv(k, raising=False)
    env = rccl.configure_rccl(channels=28)
    assert env["NCCL_MIN_NCHANNELS"] == "28" and env["NCCL_MAX_NCHANNELS"] == "28"
