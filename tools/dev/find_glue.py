"""List the ATen ops (with shapes) that launch elementwise / copy / fill kernels inside one ResNet-50 training step."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import argparse, bench
from torch.profiler import profile, ProfilerActivity
model = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
from distributed_tensorflow_amd.models import build_model
from distributed_tensorflow_amd.ops import functional as F
from distributed_tensorflow_amd.parallel.engine import DataParallelEngine
dev = torch.device("cuda", 0)
m = build_model(model, num_classes=5, device=dev); m.train()
eng = DataParallelEngine(m, mode="sync_dp", optimizer="momentum", lr=0.02, opt_kwargs={"momentum": 0.9, "weight_decay": 1e-4})
size = bench.IMAGE_SIZE[model]
cpad = getattr(m, "input_channels", 8)
x = torch.zeros(batch, size, size, cpad, device=dev, dtype=torch.bfloat16); x[..., :3].normal_()
y = torch.randint(0, 5, (batch,), device=dev).int()
def step():
    eng.begin_step(); out = m(x); logits, reg = out if isinstance(out, tuple) else (out, None)
    loss = F.softmax_xent(logits, y, 5)[0]; (loss if reg is None else loss + reg).backward(); eng.finish_step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key.startswith("aten::") and e.device_time_total > 0:
        rows.append((e.device_time_total, e.key, e.count, str(e.input_shapes)[:110]))
for t, k, c, s in sorted(rows, reverse=True)[:25]:
    print("%9.1f us  %-28s x%-3d %s" % (t, k, c, s))
