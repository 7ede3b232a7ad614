"""Mailbox all-reduce, one-shot and two-shot forms: two or three rank processes on one GPU (IPC-mapped
mailboxes, flag protocol, double buffering, rank-ordered summation, timeout when a peer never arrives)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "p2p_worker.py")


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


def _launch(mode, world=2, device_per_rank=False):
    port = _port()
    env = dict(os.environ, DTFA_TEST_DEVICE_PER_RANK="1" if device_per_rank else "0")
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), port, mode], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=150)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return procs, outs


def test_p2p_allreduce_two_ranks_one_gpu():
    procs, outs = _launch("run")
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "P2P_OK" in o, o[-3000:]


def test_p2p_allreduce_three_ranks_one_gpu():
    """Three ranks: slices of unequal length in the two-shot form, ``auto`` picks it from 256 KiB."""
    procs, outs = _launch("run", world=3)
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "P2P_OK" in o, o[-3000:]


def test_p2p_allreduce_times_out_without_peer():
    procs, outs = _launch("timeout")
    assert procs[0].returncode == 0 and "TIMEOUT_DETECTED" in outs[0], outs[0][-3000:]


@pytest.mark.multigpu
def test_p2p_allreduce_across_devices():
    """Every rank on a GPU of its own: mailboxes mapped between different devices, flags and payload over xGMI.
    Only collected on boxes with two or more GPUs (the pool's test boxes have one: there this path is unpinned)."""
    import torch

    procs, outs = _launch("run", world=min(4, torch.cuda.device_count()), device_per_rank=True)
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "P2P_OK" in o, o[-3000:]
