"""TCP parameter server: roles, single initialisation, bounded staleness, elasticity, checkpoints."""
import socket
import threading
import time

import numpy as np
import pytest
import torch

from distributed_tensorflow_amd.io import bundle
from distributed_tensorflow_amd.io.saver import Saver
from distributed_tensorflow_amd.models import build_model
from distributed_tensorflow_amd.ops import functional as F
from distributed_tensorflow_amd.parallel.ps import ParameterServer, PSStopped, PSWorkerEngine

WIDTHS = (8, 8, 16, 16, 16)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(seed, n=4, size=32):
    g = torch.Generator().manual_seed(seed)
    x = torch.zeros(n, size, size, 8)
    x[..., :3] = torch.randn(n, size, size, 3, generator=g)
    return x, torch.randint(0, 5, (n,), generator=g)


def _model(seed):
    torch.manual_seed(seed)
    return build_model("flower_cnn", widths=WIDTHS, dropout_rate=0.0)


def _step(model, eng, seed):
    eng.begin_step()
    logits, reg = model(*_batch(seed)[:1])
    loss, _, _, _ = F.softmax_xent(logits, _batch(seed)[1], 5)
    (loss + reg).backward()
    return eng.finish_step(), float(loss)


@pytest.fixture
def servers():
    started = []

    def start(n=1, **kw):
        out = []
        for i in range(n):
            port = _free_port()
            ps = ParameterServer("127.0.0.1", port, ps_index=i, **kw)
            ps.start_background()
            out.append((ps, ("127.0.0.1", port)))
            started.append(ps)
        return out

    yield start
    for ps in started:
        ps.stop()


def test_chief_initialises_once_and_everyone_sees_the_same_variables(servers):
    (ps, addr), = servers(1)
    chief_model = _model(1)
    chief = PSWorkerEngine(chief_model, [addr], 0, True, lr=1e-2)
    other_model = _model(2)  # different local init: must be overwritten by the chief's values
    other = PSWorkerEngine(other_model, [addr], 1, False, lr=1e-2)
    for (n, a), (_, b) in zip(chief_model.named_parameters(), other_model.named_parameters()):
        torch.testing.assert_close(a, b, rtol=0, atol=0, msg=n)
    # a late "chief" must not reset the variables (the reference re-runs init on every worker)
    gs, _ = _step(chief_model, chief, 0)
    assert gs == 1
    late = PSWorkerEngine(_model(3), [addr], 2, True, lr=1e-2)
    assert late.global_step == 1
    other.pull()
    for (n, a), (_, b) in zip(other_model.named_parameters(), late.model.named_parameters()):
        torch.testing.assert_close(a, b, rtol=0, atol=0, msg=n)
    for e in (chief, other, late):
        e.close()


def test_ps_applies_tf1_adam_and_counts_global_steps(servers):
    (ps, addr), = servers(1)
    model = _model(1)
    eng = PSWorkerEngine(model, [addr], 0, True, optimizer="adam", lr=1e-2)
    # local replica of what the ps should compute
    ref = _model(1)
    from distributed_tensorflow_amd.parallel.engine import DataParallelEngine

    ref_eng = DataParallelEngine(ref, mode="sync_dp", optimizer="adam", lr=1e-2, bucket_mb=1 << 20)
    for t in range(3):
        gs, _ = _step(model, eng, t)
        assert gs == t + 1
        ref_eng.begin_step()
        logits, reg = ref(_batch(t)[0])
        loss, _, _, _ = F.softmax_xent(logits, _batch(t)[1], 5)
        (loss + reg).backward()
        ref_eng.finish_step()
    eng.pull()
    for (n, a), (_, b) in zip(model.named_parameters(), ref.named_parameters()):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6, msg=n)
    eng.close()


def test_variables_are_placed_round_robin_over_ps_tasks(servers):
    pair = servers(2)
    addrs = [a for _, a in pair]
    model = _model(1)
    eng = PSWorkerEngine(model, addrs, 0, True, lr=1e-2)
    n_params = len(list(model.parameters()))
    assert [len(p) for p in eng.placement] == [(n_params + 1) // 2, n_params // 2]
    names = [n for n, _ in model.named_parameters()]
    assert eng.placement[0][0][1].name == names[0] and eng.placement[1][0][1].name == names[1]
    sizes = [sum(s.numel for _, s in pl) for pl in eng.placement]
    assert pair[0][0].params.numel() == sizes[0] and pair[1][0].params.numel() == sizes[1]
    before = [p.detach().clone() for p in model.parameters()]
    gs, _ = _step(model, eng, 0)
    eng.pull()
    assert gs == 1 and pair[1][0].apply_count == 1
    assert all(not torch.equal(a, b) for a, b in zip(before, model.parameters()))
    eng.close()


def test_staleness_bound_is_never_exceeded(servers):
    (ps, addr), = servers(1, staleness=1, debug=True)
    fast_model, slow_model = _model(1), _model(1)
    fast = PSWorkerEngine(fast_model, [addr], 0, True, lr=1e-3)
    slow = PSWorkerEngine(slow_model, [addr], 1, False, lr=1e-3)
    log = []

    def run(model, eng, n, delay, tag):
        for t in range(n):
            _step(model, eng, t)
            log.append((tag, eng.clock, time.time()))
            time.sleep(delay)

    tf = threading.Thread(target=run, args=(fast_model, fast, 6, 0.0, "fast"))
    ts = threading.Thread(target=run, args=(slow_model, slow, 6, 0.15, "slow"))
    tf.start(); ts.start(); tf.join(60); ts.join(60)
    assert not tf.is_alive() and not ts.is_alive()
    assert ps.max_lag_seen <= 1
    assert ps.global_step == 12
    assert ps.violations == [] and fast.status()["debug"] is True  # version-counter assertions held throughout
    # reconstruct clocks over time: the fast worker never started a step more than s ahead
    clocks = {"fast": 0, "slow": 0}
    for tag, c, _ in sorted(log, key=lambda r: r[2]):
        clocks[tag] = c
        assert clocks["fast"] - clocks["slow"] <= 1 + 1  # finished-step clocks may differ by s plus the step in flight
    fast.close(); slow.close()


def test_dead_worker_does_not_block_the_others(servers):
    (ps, addr), = servers(1, staleness=0, worker_timeout=5.0)
    m0, m1 = _model(1), _model(1)
    w0 = PSWorkerEngine(m0, [addr], 0, True, lr=1e-3)
    w1 = PSWorkerEngine(m1, [addr], 1, False, lr=1e-3)
    _step(m0, w0, 0)
    _step(m1, w1, 0)
    # worker 1 dies without saying goodbye
    for c in w1.conns:
        c.sock.close()
    t0 = time.time()
    for t in range(1, 4):  # with s = 0 this would wait forever for worker 1 if it still counted
        _step(m0, w0, t)
    assert time.time() - t0 < 10
    assert ps.global_step == 5
    w0.close()


def test_workers_stop_cleanly_when_the_chief_shuts_the_ps_down(servers):
    (ps, addr), = servers(1)
    m0, m1 = _model(1), _model(1)
    w0 = PSWorkerEngine(m0, [addr], 0, True, lr=1e-3)
    w1 = PSWorkerEngine(m1, [addr], 1, False, lr=1e-3)
    _step(m1, w1, 0)
    w0.close(shutdown_ps=True)
    with pytest.raises(PSStopped):
        _step(m1, w1, 1)
    w1.close()


def test_checkpoint_through_ps_contains_owner_side_adam_slots(servers, tmp_path):
    pair = servers(2)
    addrs = [a for _, a in pair]
    model = _model(1)
    eng = PSWorkerEngine(model, addrs, 0, True, optimizer="adam", lr=1e-2)
    for t in range(2):
        _step(model, eng, t)
    prefix = Saver(model, eng).save(str(tmp_path), eng.global_step)
    data = bundle.read_bundle(prefix)
    assert int(data["global_step"]) == 2
    assert "fc_trainable/flower_logits/kernel/Adam" in data and "FlowerCNN/conv1/weights/Adam_1" in data
    assert float(np.abs(data["FlowerCNN/conv1/weights/Adam"]).sum()) > 0
    np.testing.assert_allclose(data["train/beta1_power"], 0.9 ** 3, rtol=1e-6)
    # values are the ps's fp32 values
    eng.pull()
    w = dict(model.named_parameters())["conv1.weight"].detach()
    np.testing.assert_allclose(data["FlowerCNN/conv1/weights"], w[..., :3].permute(1, 2, 3, 0).numpy(), rtol=0, atol=0)
    eng.close()


def test_debug_mode_rejects_a_push_with_a_wrong_clock(servers):
    """Debug mode asserts on version counters: a push whose clock does not match the ps's record of that
    worker (a lost or duplicated update) is refused and recorded."""
    (ps, addr), = servers(1, staleness=2, debug=True)
    model = _model(3)
    eng = PSWorkerEngine(model, [addr], 0, True, lr=1e-3)
    _step(model, eng, 0)
    eng.clock -= 1  # replay the same clock: a duplicated update
    with pytest.raises(Exception):
        _step(model, eng, 1)
    assert any("pushed clock" in v for v in ps.violations)
    eng.close()
