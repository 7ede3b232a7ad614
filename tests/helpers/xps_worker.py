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
from distributed_tensorflow_amd.parallel.xps import XgmiPSEngine  # noqa: E402

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


def masters(eng):
    eng.full_master()
    return torch.cat([g.master[s.offset:s.offset + s.numel].clone().cpu() for g in eng.flat.groups for s in g.segments])


def main():
    rank, world, port, mode, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + port, rank=rank, world_size=world)
    torch.cuda.set_device(0)
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
        # timeout once
        m = model()
        eng = XgmiPSEngine(m, optimizer="adam", lr=1e-3, staleness=1, timeout=1.5)
        t_steps = []
        for t in range(7):
            if rank == world - 1 and t == 2:
                print("XPS_DYING rank", rank, flush=True)
                os._exit(0)
            t0 = time.time()
            step(m, eng, rank, t)
            torch.cuda.synchronize()
            t_steps.append(time.time() - t0)
        st = eng.state()
        result.update(step_seconds=t_steps, state=st)
        assert st["clock"] == 7, st
        assert st["dead_mask"] == 1 << (world - 1), st
        slow = [x for x in t_steps if x > 1.0]
        assert len(slow) <= 2, t_steps  # the gate (and at most one slot wait) ran into the timeout once
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
