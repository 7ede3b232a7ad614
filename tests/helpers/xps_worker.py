"""Rank process of tests/test_gpu_xps.py: N ranks share one GPU; IPC handles travel over gloo, the data path is
peer memory only."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_tensorflow_amd.models import build_model  # noqa: E402
from distributed_tensorflow_amd.ops import functional as F  # noqa: E402
from distributed_tensorflow_amd.parallel.engine import DataParallelEngine  # noqa: E402
from distributed_tensorflow_amd.parallel.xps import XgmiPSEngine, XgmiPSServer, XpsCluster  # noqa: E402

WIDTHS = (8, 8, 16, 16, 16)


def batch(rank, step, n=8, size=32):
    g = torch.Generator().manual_seed(1000 * rank + step)
    x = torch.zeros(n, size, size, 8)
    x[..., :3] = torch.randn(n, size, size, 3, generator=g)
    return x.cuda().bfloat16(), torch.randint(0, 5, (n,), generator=g, dtype=torch.int32).cuda()


def model():
    torch.manual_seed(7)
    return build_model("flower_cnn", num_classes=5, device="cuda", widths=WIDTHS, dropout_rate=0.0)


def step(m, eng, rank, t):
    eng.begin_step()
    logits, reg = m(batch(rank, t)[0])
    loss, _, _, _ = F.softmax_xent(logits, batch(rank, t)[1], 5)
    (loss + reg).backward()
    eng.finish_step()
    return loss


def _device_of(rank):
    """Device of a rank process: GPU 0 for everybody (the pool's boxes have one), or one GPU per rank when the test
    asks for it (``DTFA_TEST_DEVICE_PER_RANK=1`` on a node with several)."""
    if os.environ.get("DTFA_TEST_DEVICE_PER_RANK") == "1":
        return rank % torch.cuda.device_count()
    return 0


def masters(eng):
    eng.full_master()
    return torch.cat([g.master[s.offset:s.offset + s.numel].clone().cpu() for g in eng.flat.groups for s in g.segments])


def ps_cluster_role(rank, world, port, mode, out_dir):
    """ps tasks own the variables: process 0 and 1 are ps tasks, the others workers 0 .. world - 3.  Worker 1 dies
    after two steps without a word; the last worker starts when the others are well under way."""
    torch.cuda.set_device(_device_of(rank))
    n_ps, n_workers = 2, world - 2
    cluster = XpsCluster("127.0.0.1", int(port), n_ps, n_workers, connect_timeout=60.0)
    result = {"rank": rank}
    if rank < n_ps:
        server = XgmiPSServer(cluster, rank, optimizer="adam", lr=1e-2, timeout=2.0)
        applied = server.serve_forever()
        result.update(role="ps", applied=applied)
        assert applied > 0
    else:
        w = rank - n_ps
        late = w == n_workers - 1
        if late:
            time.sleep(4.0)
        m = model()
        if w != 0:
            with torch.no_grad():   # only the chief's initial values may survive
                for p in m.parameters():
                    p.add_(1.0)
        eng = XgmiPSEngine(m, optimizer="adam", lr=1e-2, staleness=2, timeout=2.0, cluster=cluster, worker_index=w)
        joined_at = eng.global_step
        if late:
            # a worker that joins at step k starts from the values of step k, not from the initial ones
            assert joined_at > 0, joined_at
        losses = []
        steps = 6 if late else 14
        for t in range(steps):
            if w == 1 and t == 2:
                print("XPS_DYING worker", w, flush=True)
                os._exit(0)
            if not late:
                time.sleep(0.5)
            losses.append(float(step(m, eng, 0, 0).item()))   # one fixed batch: the loss must fall
            torch.cuda.synchronize()
        st = eng.state()
        result.update(role="worker", worker=w, joined_at=int(joined_at), losses=losses, state=st)
        assert st["dropped_pushes"] == 0 and st["torn_pulls"] == 0, st
        assert losses[-1] < losses[0], losses
        if w == 0:
            time.sleep(1.0)   # (the late worker's last pushes reach the ps tasks before they are asked to stop)
            st = eng.state()
            # every push of every worker was applied to every shard: 14 (chief) + 2 (the dead one) + 6 (the late one)
            assert min(st["applied_per_shard"]) >= 14 + 2, st
        eng.close(shutdown_ps=(w == 0))
    print("XPS_OK " + json.dumps(result), flush=True)
    os._exit(0)


def main():
    rank, world, port, mode, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    if mode == "ps_owned":
        return ps_cluster_role(rank, world, port, mode, out_dir)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + port, rank=rank, world_size=world)
    torch.cuda.set_device(_device_of(rank))
    opt = "momentum" if "momentum" in mode else "adam"
    okw = {"momentum": 0.9, "weight_decay": 1e-3} if opt == "momentum" else {}
    lr = 0.05 if opt == "momentum" else 1e-2
    result = {"rank": rank}
    if mode.startswith("lockstep"):
        # staleness 0 + aggregated rounds: the arithmetic of synchronous data parallelism
        m = model()
        eng = XgmiPSEngine(m, optimizer=opt, lr=lr, opt_kwargs=okw, staleness=0, timeout=30.0)
        for t in range(4):
            step(m, eng, rank, t)
        torch.cuda.synchronize()
        dist.barrier()  # every owner has published the last round before anybody reads the final values
        flat = masters(eng)
        st = eng.state()
        assert st["clocks"] == [4] * world and st["global_step"] == 4 and st["status"] == 0, st
        torch.save(flat, os.path.join(out_dir, "xps_%s_r%d.pt" % (mode, rank)))
        dist.barrier()
        eng.close()
    elif mode.startswith("syncdp"):
        # the same run through the collective engine with the mailbox all-reduce (rank-ordered sums, no RCCL: ranks
        # sharing one GPU cannot form an RCCL communicator)
        m = model()
        eng = DataParallelEngine(m, mode="sync_dp", optimizer=opt, lr=lr, opt_kwargs=okw, bucket_mb=float(1 << 20),
                                 p2p_max_bytes=64 << 20)
        assert eng._p2p is not None
        for t in range(4):
            step(m, eng, rank, t)
        eng._p2p.check()
        flat = torch.cat([g.master[s.offset:s.offset + s.numel].clone().cpu() for g in eng.flat.groups for s in g.segments])
        torch.save(flat, os.path.join(out_dir, "xps_%s_r%d.pt" % (mode, rank)))
        dist.barrier()
    elif mode == "stale":
        # bounded staleness: rank 0 is fast, the others sleep; the lag a worker observes at the start of a step
        # never exceeds the bound
        s = 1
        m = model()
        eng = XgmiPSEngine(m, optimizer="adam", lr=1e-3, staleness=s, timeout=30.0)
        lags = []
        for t in range(6):
            if rank != 0:
                time.sleep(0.15)
            eng.begin_step()
            torch.cuda.synchronize()
            clocks = eng.state()["clocks"]  # read AFTER the gate let this step start
            lags.append(clocks[rank] - min(clocks))
            logits, reg = m(batch(rank, t)[0])
            loss, _, _, _ = F.softmax_xent(logits, batch(rank, t)[1], 5)
            (loss + reg).backward()
            eng.finish_step()
        torch.cuda.synchronize()
        dist.barrier()
        eng.drain()
        torch.cuda.synchronize()
        dist.barrier()
        st = eng.state()
        result.update(lags=lags, state=st, applied_total=st["applied"])
        assert max(lags) <= s, lags
        assert st["dropped_pushes"] == 0 and st["timeouts"] == 0, st
        # every pushed gradient was applied exactly once on this owner: world workers x 6 steps
        assert st["applied"] == world * 6, st
        eng.close()
    elif mode == "dead":
        # one worker dies after two steps without saying goodbye; the others finish their steps, paying the
        # timeout once, and the lowest live rank takes the dead worker's shard over: it keeps being applied
        m = model()
        eng = XgmiPSEngine(m, optimizer="adam", lr=1e-2, staleness=1, timeout=1.5)
        t_steps, losses, applied_mid = [], [], None
        for t in range(12):
            if rank == world - 1 and t == 2:
                print("XPS_DYING rank", rank, flush=True)
                os._exit(0)
            t0 = time.time()
            losses.append(float(step(m, eng, 0, 0).item()))   # one fixed batch for everybody: the loss must keep falling
            torch.cuda.synchronize()
            t_steps.append(time.time() - t0)
            if t == 6:
                applied_mid = eng.state()["applied_per_shard"]
        torch.cuda.synchronize()
        # the survivors meet through files (the dead rank cannot take part in a barrier): nobody reads the final
        # state or leaves before both have finished their steps
        open(os.path.join(out_dir, "dead_done_r%d" % rank), "w").close()
        t_end = time.time() + 60
        while not all(os.path.exists(os.path.join(out_dir, "dead_done_r%d" % r)) for r in range(world - 1)):
            assert time.time() < t_end
            time.sleep(0.05)
        time.sleep(0.5)
        eng.drain()
        st = eng.state()
        result.update(step_seconds=t_steps, state=st, losses=losses, applied_mid=applied_mid)
        assert st["clock"] == 12, st
        assert st["dead_mask"] == 1 << (world - 1), st
        slow = [x for x in t_steps if x > 1.0]
        assert len(slow) <= 3, t_steps  # one timeout of its own, and the wait for the other survivor's
        # (whichever wait notices the dead process first writes it off for the other one; the survivors see each
        # other's beat and do not write each other off)
        assert st["timeouts"] <= 2 and st["dropped_pushes"] <= 2, st
        # the dead worker's shard was taken over by rank 0 and kept training like the others
        assert st["adopters"][world - 1] == 1, st
        assert (st["adopted_mask"] != 0) == (rank == 0), st
        for o in range(world):
            assert st["applied_per_shard"][o] > applied_mid[o] > 2, (o, applied_mid, st)
        assert min(st["applied_per_shard"]) >= 2 * 9, st     # two live workers x the steps after the take-over
        assert losses[-1] < 0.5 * losses[0], losses
        open(os.path.join(out_dir, "dead_checked_r%d" % rank), "w").close()
        while not all(os.path.exists(os.path.join(out_dir, "dead_checked_r%d" % r)) for r in range(world - 1)):
            assert time.time() < t_end + 60
            time.sleep(0.05)
        eng.close()
    elif mode == "ckpt":
        # checkpoint through the Saver: masters and slots gathered by one-sided reads, restored into a new engine
        from distributed_tensorflow_amd.io.saver import Saver

        m = model()
        eng = XgmiPSEngine(m, optimizer="adam", lr=1e-2, staleness=0, timeout=30.0)
        for t in range(3):
            step(m, eng, rank, t)
        torch.cuda.synchronize()
        dist.barrier()
        saver = Saver(m, eng)
        if rank == 0:
            saver.save(out_dir, 3)
        dist.barrier()
        for t in range(3, 5):
            step(m, eng, rank, t)
        torch.cuda.synchronize()
        dist.barrier()
        want = masters(eng)
        dist.barrier()
        m2 = model()
        with torch.no_grad():
            for p in m2.parameters():
                p.add_(0.5)
        eng2 = XgmiPSEngine(m2, optimizer="adam", lr=1e-2, staleness=0, timeout=30.0)
        assert Saver(m2, eng2).restore_latest(out_dir) == 3
        assert eng2.global_step == 3
        for t in range(3, 5):
            step(m2, eng2, rank, t)
        torch.cuda.synchronize()
        dist.barrier()
        got = masters(eng2)
        assert eng2.global_step == 5
        err = (got - want).abs().max().item()
        assert err <= 1e-6, err
        dist.barrier()
        eng.close()
        eng2.close()
    print("XPS_OK " + json.dumps(result), flush=True)
    dist.destroy_process_group()
    os._exit(0)


if __name__ == "__main__":
    main()
