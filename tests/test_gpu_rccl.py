"""The collective engine on real RCCL.  One rank per process is all a one-GPU box allows, so the job is a
world of one launched through torch.distributed.run with DTFA_FORCE_COLLECTIVES=1: every bucketed
all-reduce / reduce-scatter / all-gather, the staleness ring, the mailbox all-reduce and the captured
step graph issue exactly the calls a multi-GPU job issues."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


def _failure_text(out: str) -> str:
    """The lines of a failed child's output that name the failure (a watchdog abort prints its reason far above the
    launcher's traceback), then its tail."""
    try:   # the whole output next to the other run artefacts, when there is such a directory
        d = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(d):
            with open(os.path.join(d, "rccl_child_failure.log"), "w") as f:
                f.write(out)
    except OSError:
        pass
    keys = ("terminated with exception", "HIP error", "NCCL error", "RuntimeError", "Error:", "what():")
    named = [l for l in out.splitlines() if any(k in l for k in keys)]
    return "\n".join(named[:20]) + "\n...\n" + out[-1500:]


@pytest.mark.parametrize("extra", [
    ["--mode", "sync_dp"],
    ["--mode", "sharded_ps"],
    ["--mode", "async_ssp", "--staleness", "2"],
    ["--mode", "sync_dp", "--p2p_kb", "4096", "--bucket_mb", "1"],
    ["--mode", "sync_dp", "--hip_graph", "1"],
    ["--mode", "sharded_ps", "--hip_graph", "1"],
    ["--mode", "sync_dp", "--grad_dtype", "bf16", "--bucket_mb", "1"],
    ["--mode", "sharded_ps", "--grad_dtype", "bf16"],
])
def test_bench_under_torchrun_one_rank(extra):
    env = dict(os.environ, DTFA_FORCE_COLLECTIVES="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", _port(), "bench.py", "--gpus", "1", "--model", "flower_cnn", "--batch", "32",
           "--steps", "4", "--warmup", "2"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    assert r.returncode == 0, _failure_text(r.stdout)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["final_loss"] == out["final_loss"]  # finite
    assert out["final_loss"] < 5.0
    # communication report: the RCCL world, the gradient buckets and (eager runs) the exposed collective wait
    assert out["rccl_world"] == 1
    comm = out["comm"]
    assert comm["collective"] is True and comm["world"] == 1 and comm["mode"] == extra[1]
    assert comm["buckets"] == len(comm["bucket_bytes"]) >= 1 and sum(comm["bucket_bytes"]) == comm["gradient_bytes"]
    assert comm["grad_dtype"] == ("bfloat16" if "bf16" in extra else "float32")
    if "--bucket_mb" in extra:
        assert comm["buckets"] > 1  # (a bucket holds whole tensors, so one may exceed the nominal size)
    if "--hip_graph" not in extra and "--p2p_kb" not in extra:
        assert comm["collectives_waited"] >= 4 and comm["exposed_wait_ms_per_step"] >= 0.0


@pytest.mark.parametrize("extra", [["--mode", "sync_dp"], ["--mode", "sharded_ps", "--hip_graph", "1"]])
def test_resnet50_bench_under_torchrun_one_rank(extra):
    """The headline model (folded bottleneck tails, pixel-pair stem, gated-gradient handoff between blocks) through
    the same launcher and collectives as a multi-GPU job."""
    env = dict(os.environ, DTFA_FORCE_COLLECTIVES="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", _port(), "bench.py", "--gpus", "1", "--model", "resnet50", "--batch", "16",
           "--steps", "3", "--warmup", "2", "--also", "", "--also_modes", ""] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    assert r.returncode == 0, _failure_text(r.stdout)
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["config"]["model"] == "resnet50" and out["value"] > 0 and out["final_loss"] < 5.0
    assert out["comm"]["collective"] is True and out["comm"]["buckets"] >= 1


@pytest.mark.parametrize("extra,launcher", [
    (["--staleness", "1"], False),
    (["--staleness", "0", "--ps_apply", "aggregate"], True),
    (["--staleness", "-1", "--hip_graph", "1", "--grad_dtype", "bf16"], True),
])
def test_bench_async_ps_world_of_one(extra, launcher):
    """``--mode async_ps``: the GPU-resident parameter server (gate, pull, push into the owner's mailbox, apply on
    the owner, commit) as the headline run, started with and without a launcher; its protocol counters are part of
    the record."""
    cmd = [sys.executable]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    if launcher:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", _port()]
    cmd += ["bench.py", "--gpus", "1", "--model", "flower_cnn", "--batch", "32", "--steps", "4", "--warmup", "2",
            "--mode", "async_ps", "--also", "", "--also_modes", ""] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    assert r.returncode == 0, _failure_text(r.stdout)
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["final_loss"] < 5.0
    assert out["config"]["mode"] == "async_ps" and out["rank_devices"] == [0]
    comm = out["comm"]
    assert comm["mode"] == "async_ps" and comm["backend"] == "xgmi" and comm["world"] == 1
    assert comm["pushes"] == 4 * comm["groups"] or comm["pushes"] == 4   # (counters of the first group's region)
    assert comm["applied"] == 4
    assert comm["dropped_pushes"] == 0 and comm["torn_pulls"] == 0 and comm["timeouts"] == 0 and comm["status"] == 0
    assert comm["pushed_bytes_per_step"] > 0 and comm["pulled_bytes_per_step"] > 0
