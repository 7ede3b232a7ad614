import os, sys, time, pickle
sys.path.insert(0, "/root/repo")
import distributed_tensorflow_amd  # noqa
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29571")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
from torch._C._distributed_c10d import _dump_nccl_trace
x = torch.ones(1<<20, device="cuda")
for i in range(3):
    dist.all_reduce(x)
torch.cuda.synchronize()
t0=time.time()
for k in range(30):
    d = pickle.loads(_dump_nccl_trace(includeCollectives=True, includeStackTraces=False, onlyActive=False))
    es = d.get("entries")
    if k == 0:
        print("keys", list(d.keys()))
        if es: print("entry keys", list(es[0].keys()))
    print(round(time.time()-t0,3), [(e.get("record_id"), e.get("state"), e.get("retired")) for e in (es or [])])
    if es and all(e.get("retired") for e in es): break
    time.sleep(0.02)
from distributed_tensorflow_amd.train import graph as G
dist.all_reduce(x)
torch.cuda.synchronize()
t0=time.time(); G.retire_eager_collectives(); print("retire wait", round(time.time()-t0,3))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    dist.all_reduce(x)
G.note_captured_collectives()
print("captured set", G._captured_records)
g.replay(); torch.cuda.synchronize()
dist.all_reduce(x); torch.cuda.synchronize()
t0=time.time(); G.retire_eager_collectives(); print("retire wait 2", round(time.time()-t0,3), x[0].item())
g = None
dist.destroy_process_group()
