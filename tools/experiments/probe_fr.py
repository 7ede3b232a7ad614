"""How long after an eager RCCL collective does the process group's flight recorder mark it retired, and does
``train.graph.retire_eager_collectives`` return only then?  (The step-graph capture waits on exactly that: the
watchdog polls the completion events of eager collectives, and polling one while its stream is capturing aborts the
process.)  One GPU, world of one:  python tools/experiments/probe_fr.py"""
import os
import pickle
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import distributed_tensorflow_amd  # noqa: F401,E402  (sets the flight-recorder buffer size before torch initialises)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from torch._C._distributed_c10d import _dump_nccl_trace  # noqa: E402

from distributed_tensorflow_amd.train import graph as G  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29571")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.ones(1 << 20, device="cuda")
for _ in range(3):
    dist.all_reduce(x)
torch.cuda.synchronize()
t0 = time.time()
for k in range(30):
    d = pickle.loads(_dump_nccl_trace(includeCollectives=True, includeStackTraces=False, onlyActive=False))
    es = d.get("entries") or []
    print(round(time.time() - t0, 3), [(e.get("record_id"), e.get("state"), e.get("retired")) for e in es])
    if es and all(e.get("retired") for e in es):
        break
    time.sleep(0.02)
dist.all_reduce(x)
torch.cuda.synchronize()
t0 = time.time()
G.retire_eager_collectives()
print("retire wait", round(time.time() - t0, 3))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    dist.all_reduce(x)
G.note_captured_collectives()
g.replay()
torch.cuda.synchronize()
print("captured + replayed")
dist.destroy_process_group()
