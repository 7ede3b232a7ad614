import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distributed_tensorflow_amd.models import build_model
from distributed_tensorflow_amd.ops import functional as F
from distributed_tensorflow_amd.parallel.engine import DataParallelEngine
dev = torch.device("cuda", 0)
B = int(os.environ.get("B", "256"))
def run(fold, steps=14):
    F.EXPAND_FOLD = fold
    torch.manual_seed(0)
    model = build_model("resnet50", num_classes=5, device=dev).train()
    eng = DataParallelEngine(model, mode="sync_dp", optimizer="momentum", lr=0.05, opt_kwargs={"momentum": 0.9, "weight_decay": 1e-4})
    g = torch.Generator().manual_seed(1)
    x = torch.zeros(B, 224, 224, 8); x[..., :3] = torch.randn(B, 224, 224, 3, generator=g)
    x = x.to(dev, torch.bfloat16); y = torch.randint(0, 5, (B,), generator=g, dtype=torch.int32).to(dev)
    # make the labels learnable: class-dependent mean shift
    x[..., 0] += (y.view(-1, 1, 1).float() * 0.5).to(torch.bfloat16)
    out = []
    for _ in range(steps):
        eng.begin_step()
        loss, _, _, correct = F.softmax_xent(model(x), y, 5)
        loss.backward()
        eng.finish_step()
        out.append((round(float(loss.item()), 4), int(correct.sum().item())))
    return out
a = run(True); b = run(False)
print("fold  ", a)
print("nofold", b)
