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


def test_learning_rate_decay_runs_on_the_ps(servers):
    """``lr_decay`` travels to the ps tasks with the initialisation: their optimizer steps follow
    ``lr * rate ** (global_step / decay_steps)`` exactly like a local optimizer with the same schedule."""
    from distributed_tensorflow_amd.optim.schedules import exponential_decay
    from distributed_tensorflow_amd.parallel.engine import DataParallelEngine

    (ps, addr), = servers(1)
    model = _model(1)
    eng = PSWorkerEngine(model, [addr], 0, True, optimizer="adam", lr=1e-2, lr_decay=(2, 0.5))
    ref = _model(1)
    ref_eng = DataParallelEngine(ref, mode="sync_dp", optimizer="adam", lr=exponential_decay(1e-2, 2, 0.5), bucket_mb=1 << 20)
    for t in range(4):
        _step(model, eng, t)
        ref_eng.begin_step()
        logits, reg = ref(_batch(t)[0])
        loss, _, _, _ = F.softmax_xent(logits, _batch(t)[1], 5)
        (loss + reg).backward()
        ref_eng.finish_step()
    eng.pull()
    for (n, a), (_, b) in zip(model.named_parameters(), ref.named_parameters()):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6, msg=n)
    assert abs(eng.current_lr() - 1e-2 * 0.5 ** 2) < 1e-9
    eng.close()


def test_pipelined_worker_overlaps_network_and_compute_without_losing_a_push(servers):
    """``pipeline=True``: push and pull run on a helper thread; every gradient is applied exactly once, a step never
    starts from variables older than the push before last, and training still descends."""
    (ps, addr), (ps2, addr2) = servers(2, debug=True)
    model = _model(1)
    eng = PSWorkerEngine(model, [addr, addr2], 0, True, optimizer="adam", lr=1e-2, pipeline=True)
    losses = []
    for t in range(12):
        eng.begin_step()
        logits, reg = model(_batch(0)[0])          # one fixed batch: the loss must fall
        loss, _, _, _ = F.softmax_xent(logits, _batch(0)[1], 5)
        (loss + reg).backward()
        eng.finish_step()
        losses.append(float(loss))
        # at most one push / pull pair in flight: everything but the newest push has been acknowledged
        assert eng._job_seq == t + 1 and eng._pulled_seq >= t
        assert eng._uploaded_seq >= t - 1
    st = eng.status()   # (waits for the pair in flight)
    assert st["global_step"] == 12 and ps.global_step == 12 and ps2.global_step == 12
    assert st["violations"] == [] and ps2.violations == []
    assert losses[-1] < 0.7 * losses[0], losses
    eng.pull()
    a = torch.cat([p.detach().flatten() for p in model.parameters()])
    eng2 = PSWorkerEngine(_model(5), [addr, addr2], 1, False, optimizer="adam", lr=1e-2)
    b = torch.cat([p.detach().flatten() for p in eng2.model.parameters()])
    torch.testing.assert_close(a, b, rtol=0, atol=0)
    eng.close()
    eng2.close()


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


def _params(model):
    return torch.cat([p.detach().float().reshape(-1).clone() for p in model.parameters()])


def test_resume_restores_variables_slots_and_global_step(servers, tmp_path):
    """A run interrupted after 3 steps and resumed from its checkpoint (new ps task, new workers) ends where the
    uninterrupted run ends: Adam moments, beta powers (step count) and the global step travel to the ps."""
    (ps, addr), = servers(1)
    model = _model(1)
    eng = PSWorkerEngine(model, [addr], 0, True, optimizer="adam", lr=1e-2)
    for t in range(3):
        _step(model, eng, t)
    saver = Saver(model, eng)
    prefix = saver.save(str(tmp_path), 3)
    assert prefix.endswith("model.ckpt-3")
    for t in range(3, 5):
        gs, _ = _step(model, eng, t)
    assert gs == 5
    eng.pull()
    want = _params(model)
    eng.close()

    (ps2, addr2), = servers(1)
    chief_model = _model(9)  # a fresh process would start from some other initial values
    chief = PSWorkerEngine(chief_model, [addr2], 0, True, optimizer="adam", lr=1e-2, defer_init=True)
    assert not ps2.initialized
    # a second worker arriving early waits for the chief's (restored) initialisation
    joined = {}

    def join():
        m = _model(5)
        joined["eng"] = PSWorkerEngine(m, [addr2], 1, False, optimizer="adam", lr=1e-2)
        joined["model"] = m

    th = threading.Thread(target=join)
    th.start()
    time.sleep(0.3)
    assert "eng" not in joined
    assert Saver(chief_model, chief).restore_latest(str(tmp_path)) == 3
    chief.initialize()
    th.join(timeout=20)
    assert ps2.global_step == 3 and ps2.optimizer.step_count == 3
    assert joined["eng"].global_step == 3
    torch.testing.assert_close(_params(joined["model"]), _params(chief_model), rtol=0, atol=0)
    slots = ps2.optimizer.slots()
    assert all(float(v.abs().sum()) > 0 for v in slots.values())  # the moments came along
    for t in range(3, 5):
        gs, _ = _step(chief_model, chief, t)
    assert gs == 5
    chief.pull()
    torch.testing.assert_close(_params(chief_model), want, rtol=1e-6, atol=1e-7)
    chief.close()
    joined["eng"].close()


def test_set_state_after_init_restores_slots_too(servers, tmp_path):
    """Restore into an already initialised cluster (``load_master`` / ``load_full_slots`` over OP_SET_STATE)."""
    (ps, addr), = servers(1)
    model = _model(1)
    eng = PSWorkerEngine(model, [addr], 0, True, optimizer="adam", lr=1e-2)
    for t in range(2):
        _step(model, eng, t)
    saver = Saver(model, eng)
    saver.save(str(tmp_path), 2)
    m_before = {k: v.clone() for k, v in ps.optimizer.slots().items()}
    for t in range(2, 4):
        _step(model, eng, t)
    assert saver.restore_latest(str(tmp_path)) == 2
    assert ps.global_step == 2 and ps.optimizer.step_count == 2
    for k, v in ps.optimizer.slots().items():
        torch.testing.assert_close(v, m_before[k], rtol=0, atol=0)
    eng.close()


def test_periodic_hooks_fire_when_the_chief_only_sees_odd_global_steps(servers, tmp_path):
    """Two asynchronous workers alternate; the chief's updates produce global steps 1, 3, 5, ...  Hooks that fire
    'every 2 global steps' must still fire (when a multiple of 2 has been passed), and the checkpoint carries
    the global step of the state fetched from the ps."""
    from distributed_tensorflow_amd.io.events import read_events
    from distributed_tensorflow_amd.train.session import CheckpointSaverHook, SummarySaverHook, TrainingSession

    (ps, addr), = servers(1)
    chief_model, other_model = _model(1), _model(1)
    chief = PSWorkerEngine(chief_model, [addr], 0, True, lr=1e-2)
    other = PSWorkerEngine(other_model, [addr], 1, False, lr=1e-2)
    saver = Saver(chief_model, chief)
    ckpt_dir, ev_dir = str(tmp_path / "ckpt"), str(tmp_path / "events")
    hooks = [CheckpointSaverHook(ckpt_dir, 2, saver, save_at_end=False, background=False),
             SummarySaverHook(2, ev_dir, histogram=False)]
    sess = TrainingSession(chief_model, chief, num_classes=5, hooks=hooks, is_chief=True)
    seen = []
    with sess:
        for t in range(3):
            r = sess.run_step(*_batch(t))
            seen.append(r.global_step)
            _step(other_model, other, 100 + t)
    assert seen == [1, 3, 5]
    import glob
    import os

    names = sorted(f for f in os.listdir(ckpt_dir) if f.endswith(".index"))
    assert names == ["model.ckpt-3.index", "model.ckpt-5.index"], names
    assert bundle.latest_checkpoint(ckpt_dir).endswith("model.ckpt-5")
    steps = sorted({e["step"] for f in glob.glob(os.path.join(ev_dir, "events.out.tfevents.*"))
                    for e in read_events(f) if e["scalars"]})
    assert steps == [3, 5], steps
    chief.close()
    other.close()
