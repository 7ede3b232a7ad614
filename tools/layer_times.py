"""Per-call device time of the framework's kernel entry points during a training step, grouped by entry point and
operand shape (HIP events around every call; no library changes: the entry points are wrapped from here).

    python tools/layer_times.py [--model resnet50] [--batch 2048] [--steps 2]
"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.models import build_model  # noqa: E402
from distributed_tensorflow_amd.ops import conv as cv  # noqa: E402
from distributed_tensorflow_amd.ops import functional as F  # noqa: E402
from distributed_tensorflow_amd.ops import kernels as K  # noqa: E402
from distributed_tensorflow_amd.parallel.engine import DataParallelEngine  # noqa: E402

records = []
enabled = [False]


def shape_of(a):
    if isinstance(a, torch.Tensor):
        return "x".join(str(d) for d in a.shape)
    if isinstance(a, cv.ConvPlan):
        return "plan(%dx%d %d->%d k%dx%d s%d)" % (a.ih, a.iw, a.cin, a.cout, a.kh, a.kw, a.sy)
    return None


def wrap(mod, name, nargs=3):
    fn = getattr(mod, name)

    def inner(*args, **kw):
        if not enabled[0]:
            return fn(*args, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*args, **kw)
        e1.record()
        extra = [k for k in ("addend", "bn_tap", "x2", "residual", "mask_out", "want_stats") if kw.get(k) is not None and kw.get(k) is not False]
        label = name + " " + " ".join(s for s in (shape_of(a) for a in args[:nargs]) if s) + (" +" + ",".join(extra) if extra else "")
        records.append((label, e0, e1))
        return out

    setattr(mod, name, inner)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="resnet50")
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--top", type=int, default=70)
    args = ap.parse_args()
    for nm in ("conv_fwd", "conv_dgrad", "conv_wgrad", "conv_pro"):
        wrap(cv, nm)
    for nm in ("bn_apply", "bn_finalize", "bn_backward_coefs", "bn_backward_apply", "bn_stats_partials", "mask_mul",
               "bn_relu_maxpool_fwd", "bn_pool_backward", "maxpool_fwd", "maxpool_bwd", "gap_fwd", "gap_bwd", "reduce_partials"):
        wrap(K, nm, nargs=2)
    dev = torch.device("cuda", 0)
    model = build_model(args.model, num_classes=5, device=dev).train()
    eng = DataParallelEngine(model, mode="sync_dp", optimizer="momentum", lr=0.05, opt_kwargs={"momentum": 0.9, "weight_decay": 1e-4})
    x = torch.zeros(args.batch, args.size, args.size, getattr(model, "input_channels", 8), device=dev, dtype=torch.bfloat16)
    x[..., :3] = torch.randn(args.batch, args.size, args.size, 3, device=dev).to(torch.bfloat16)
    y = torch.randint(0, 5, (args.batch,), dtype=torch.int32, device=dev)

    def step():
        eng.begin_step()
        out = model(x)
        logits, reg = out if isinstance(out, tuple) else (out, None)
        loss, _, _, _ = F.softmax_xent(logits, y, 5)
        (loss if reg is None else loss + reg).backward()
        eng.finish_step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    enabled[0] = True
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(args.steps):
        step()
    t1.record()
    torch.cuda.synchronize()
    tot = collections.OrderedDict()
    for label, e0, e1 in records:
        d = tot.setdefault(label, [0, 0.0])
        d[0] += 1
        d[1] += e0.elapsed_time(e1)
    step_ms = t0.elapsed_time(t1) / args.steps
    covered = sum(v[1] for v in tot.values()) / args.steps
    print("step %.2f ms, wrapped entry points %.2f ms" % (step_ms, covered))
    print("%-100s %6s %9s %9s" % ("entry point / operands", "calls", "ms/step", "us/call"))
    for label, (n, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:args.top]:
        print("%-100s %6d %9.3f %9.1f" % (label[:100], n // args.steps, ms / args.steps, ms / n * 1e3))
    by_fn = collections.Counter()
    for label, (n, ms) in tot.items():
        by_fn[label.split(" ")[0]] += ms / args.steps
    print({k: round(v, 2) for k, v in by_fn.most_common()})


if __name__ == "__main__":
    main()
