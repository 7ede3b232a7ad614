"""Parameter-server worker on the GPU: the device part of its step replayed from a HIP graph."""
import socket

import pytest
import torch

from distributed_tensorflow_amd.models import build_model
from distributed_tensorflow_amd.parallel.ps import ParameterServer, PSWorkerEngine
from distributed_tensorflow_amd.train.session import TrainingSession

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(hip_graph):
    port = _free_port()
    ps = ParameterServer("127.0.0.1", port)
    ps.start_background()
    try:
        torch.manual_seed(0)
        model = build_model("flower_cnn", num_classes=5, device="cuda", dropout_rate=0.0)
        eng = PSWorkerEngine(model, [("127.0.0.1", port)], 0, True, optimizer="adam", lr=1e-3)
        assert eng.graph_capturable
        sess = TrainingSession(model, eng, num_classes=5, hip_graph=hip_graph, graph_warmup_steps=2)
        g = torch.Generator().manual_seed(3)
        losses = []
        for _ in range(6):
            x = torch.zeros(8, 64, 64, 8)
            x[..., :3] = torch.rand(8, 64, 64, 3, generator=g) * 2 - 1
            y = torch.randint(0, 5, (8,), generator=g, dtype=torch.int32)
            losses.append(sess.run_step(x.cuda().bfloat16(), y.cuda()).loss)
        assert (sess._graph is not None) == hip_graph
        eng.pull()
        masters = [gr.master.clone() for gr in eng.flat.groups]
        gs, clock = eng.global_step, eng.clock
        eng.close()
        return losses, masters, gs, clock
    finally:
        ps.stop()


def test_ps_worker_step_graph_matches_eager():
    """Pull (host) -> replayed forward/backward -> push (host) gives the losses, the variables on the ps and the
    step counters of the eager worker, bit for bit."""
    l0, m0, gs0, c0 = _run(False)
    l1, m1, gs1, c1 = _run(True)
    assert gs0 == gs1 == 6 and c0 == c1 == 6
    assert l0 == l1
    for a, b in zip(m0, m1):
        assert torch.equal(a, b)
