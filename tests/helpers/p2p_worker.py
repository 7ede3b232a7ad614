"""Rank process of tests/test_gpu_p2p.py: two ranks share one GPU; handles travel over gloo."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_tensorflow_amd.parallel.p2p import P2PAllReduce  # noqa: E402


def data(rank, n, dtype, it):
    g = torch.Generator().manual_seed(1000 * it + rank)
    return torch.randn(n, generator=g).to(dtype)


def main():
    rank, world, port, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + port, rank=rank, world_size=world)
    # GPU 0 for every rank, or one GPU per rank where the test asks for it on a box with several
    torch.cuda.set_device(rank % torch.cuda.device_count() if os.environ.get("DTFA_TEST_DEVICE_PER_RANK") == "1" else 0)
    p2p = P2PAllReduce(max_bytes=1 << 20, timeout_s=2.0 if mode == "timeout" else 20.0)
    if mode == "timeout":
        if rank == 0:
            t = torch.ones(1024, device="cuda")
            p2p.all_reduce_(t)
            try:
                p2p.check()
            except RuntimeError as e:
                print("TIMEOUT_DETECTED", e, flush=True)
        dist.barrier()
        os._exit(0)
    it = 0
    # one-shot and two-shot calls share the mailbox, its sequence numbers and its two parities: interleave them.
    # The two-shot form must also cope with sizes that do not divide into ``world`` slices (4, 4096 + 8 elements).
    for dtype in (torch.float32, torch.bfloat16):
        for n in (8, 1024, 4096 + 8, 65536, (1 << 20) // 4 if dtype == torch.float32 else (1 << 20) // 2):
            for rep in range(6):
                it += 1
                algo = ("one_shot", "two_shot", "auto")[rep % 3]
                mine = data(rank, n, dtype, it).cuda()
                ref = mine.clone()
                want = sum(data(r, n, dtype, it).float() for r in range(world)) * 0.5
                p2p.all_reduce_(mine, scale=0.5, algo=algo)
                got = mine.float().cpu()
                tol = 0 if dtype == torch.float32 else 2e-2
                err = (got - want).abs().max().item()
                assert err <= tol * max(1.0, want.abs().max().item()), (dtype, n, rep, algo, err)
                if algo == "two_shot":  # same bits as the one-shot form
                    p2p.all_reduce_(ref, scale=0.5, algo="one_shot")
                    assert torch.equal(ref, mine), (dtype, n, rep)
    p2p.check()
    # results are bit-identical across ranks
    t = data(rank, 4096, torch.float32, 999).cuda()
    p2p.all_reduce_(t)
    both = [None] * world
    dist.all_gather_object(both, t.cpu())
    assert all(torch.equal(both[0], b) for b in both[1:])
    t = data(rank, 40000, torch.bfloat16, 998).cuda()
    p2p.all_reduce_(t, algo="two_shot")
    dist.all_gather_object(both, t.cpu())
    assert all(torch.equal(both[0], b) for b in both[1:])
    p2p.close()
    print("P2P_OK rank", rank, "calls", it, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
