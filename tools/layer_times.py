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


def _tensors(obj, seen):
    """Every distinct tensor reachable from a call's arguments / results (batch-norm taps carry theirs as attributes)."""
    if isinstance(obj, torch.Tensor):
        if obj.data_ptr() not in seen and obj.numel() > 4096:   # (per-channel vectors do not count)
            seen[obj.data_ptr()] = obj.numel() * obj.element_size()
    elif isinstance(obj, (tuple, list)):
        for o in obj:
            _tensors(o, seen)
    elif isinstance(obj, cv.BnBackwardTap):
        for nm in ("y", "mask"):
            _tensors(getattr(obj, nm, None), seen)


def min_bytes(args, kw, out) -> int:
    """Least memory traffic of a call: every operand and every result moved once."""
    seen: dict = {}
    _tensors(list(args), seen)
    _tensors(list(kw.values()), seen)
    _tensors(out, seen)
    return sum(seen.values())


def flops_of(name, args) -> float:
    if name in ("conv_fwd", "conv_dgrad", "conv_wgrad"):
        plan = next((a for a in args if isinstance(a, cv.ConvPlan)), None)
        t = next((a for a in args if isinstance(a, torch.Tensor)), None)
        if plan is not None and t is not None:
            return 2.0 * t.shape[0] * plan.oh * plan.ow * plan.cout * plan.K
    return 0.0


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
        records.append((label, e0, e1, min_bytes(args, kw, out), flops_of(name, args)))
        return out

    setattr(mod, name, inner)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="resnet50")
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--top", type=int, default=70)
    ap.add_argument("--hbm_tbs", type=float, default=5.5, help="achievable HBM rate for the bound column (TB/s)")
    ap.add_argument("--mfma_tfs", type=float, default=2500.0, help="dense bf16 MFMA peak for the bound column (TFLOP/s)")
    args = ap.parse_args()
    for nm in ("conv_fwd", "conv_dgrad", "conv_wgrad"):
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
    for label, e0, e1, nbytes, fl in records:
        d = tot.setdefault(label, [0, 0.0, 0, 0.0])
        d[0] += 1
        d[1] += e0.elapsed_time(e1)
        d[2] += nbytes
        d[3] += fl
    step_ms = t0.elapsed_time(t1) / args.steps
    covered = sum(v[1] for v in tot.values()) / args.steps
    print("step %.2f ms, wrapped entry points %.2f ms" % (step_ms, covered))
    # the bound of a call: its operands and results moved once at HBM_TBS, or its FLOPs at MFMA_TFS, whichever is longer
    print("%-100s %6s %9s %9s %7s %7s %6s" % ("entry point / operands", "calls", "ms/step", "us/call", "TB/s", "TF/s", "bound"))
    floor_ms = 0.0
    for label, (n, ms, nbytes, fl) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        bound_ms = max(nbytes / (args.hbm_tbs * 1e9), fl / (args.mfma_tfs * 1e9))
        floor_ms += bound_ms / args.steps
    for label, (n, ms, nbytes, fl) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:args.top]:
        bound_ms = max(nbytes / (args.hbm_tbs * 1e9), fl / (args.mfma_tfs * 1e9))
        print("%-100s %6d %9.3f %9.1f %7.2f %7.0f %5.0f%%" % (label[:100], n // args.steps, ms / args.steps, ms / n * 1e3,
                                                           nbytes / ms / 1e9, fl / ms / 1e9, 100.0 * bound_ms / ms))
    print("sum of the calls' bounds (%.1f TB/s, %.0f TF/s): %.2f ms/step = %.0f %% of the measured %.2f ms" % (
        args.hbm_tbs, args.mfma_tfs, floor_ms, 100.0 * floor_ms / covered, covered))
    by_fn = collections.Counter()
    for label, (n, ms, _, _) in tot.items():
        by_fn[label.split(" ")[0]] += ms / args.steps
    print({k: round(v, 2) for k, v in by_fn.most_common()})


if __name__ == "__main__":
    main()
