"""Host enqueue time vs device time of one training step: tells whether a model is launch-bound.

    python tools/cpu_overhead.py resnet50 256
"""
import sys
import time

import torch

sys.path.insert(0, ".")
from distributed_tensorflow_amd.models import build_model  # noqa: E402
from distributed_tensorflow_amd.ops import functional as F  # noqa: E402
from distributed_tensorflow_amd.parallel.engine import DataParallelEngine  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    size = 299 if name.startswith("inception") else 224
    dev = torch.device("cuda", 0)
    model = build_model(name, num_classes=5, device=dev)
    model.train()
    engine = DataParallelEngine(model, mode="sync_dp", optimizer="momentum", lr=0.01, opt_kwargs={"momentum": 0.9})
    x = torch.zeros(batch, size, size, getattr(model, "input_channels", 8), device=dev, dtype=torch.bfloat16)
    x[..., :3].uniform_(-1, 1)
    y = torch.randint(0, 5, (batch,), device=dev, dtype=torch.int32)

    def step():
        engine.begin_step()
        out = model(x)
        logits, reg = out if isinstance(out, tuple) else (out, None)
        loss, _, _, _ = F.softmax_xent(logits, y, 5)
        (loss if reg is None else loss + reg).backward()
        engine.finish_step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s b%d: host enqueue %.2f ms/step, wall %.2f ms/step" % (name, batch, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))


if __name__ == "__main__":
    main()
