"""Micro-benchmark of the implicit-GEMM conv kernels over ResNet-50 / Inception layer shapes.

Prints achieved TFLOP/s for fwd / dgrad / wgrad and, with --reference, the time of
torch's own convolution (MIOpen) on the same shapes as an on-hardware yardstick.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import conv as cv  # noqa: E402

RESNET = [
    # H, W, Cin, Cout, k, s, pad
    (224, 224, 8, 64, 7, 2, 3),
    (56, 56, 64, 64, 1, 1, 0),
    (56, 56, 64, 64, 3, 1, 1),
    (56, 56, 64, 256, 1, 1, 0),
    (56, 56, 256, 64, 1, 1, 0),
    (56, 56, 256, 128, 1, 1, 0),
    (56, 56, 128, 128, 3, 2, 1),
    (28, 28, 128, 512, 1, 1, 0),
    (28, 28, 512, 128, 1, 1, 0),
    (28, 28, 128, 128, 3, 1, 1),
    (28, 28, 512, 256, 1, 1, 0),
    (28, 28, 256, 256, 3, 2, 1),
    (14, 14, 256, 1024, 1, 1, 0),
    (14, 14, 1024, 256, 1, 1, 0),
    (14, 14, 256, 256, 3, 1, 1),
    (14, 14, 1024, 512, 1, 1, 0),
    (14, 14, 512, 512, 3, 2, 1),
    (7, 7, 512, 2048, 1, 1, 0),
    (7, 7, 2048, 512, 1, 1, 0),
    (7, 7, 512, 512, 3, 1, 1),
]


# Inception-v3 at 299x299: (H, W, Cin, Cout, (kh, kw), stride, (top, bottom, left, right)); each shape's weight in
# the trunk (how many layers have it) is the last entry.
INCEPTION = [
    (149, 149, 32, 32, (3, 3), 1, (0, 0, 0, 0), 1),
    (147, 147, 32, 64, (3, 3), 1, (1, 1, 1, 1), 1),
    (73, 73, 64, 80, (1, 1), 1, (0, 0, 0, 0), 1),
    (73, 73, 80, 192, (3, 3), 1, (0, 0, 0, 0), 1),
    (35, 35, 288, 64, (1, 1), 1, (0, 0, 0, 0), 7),
    (35, 35, 288, 48, (1, 1), 1, (0, 0, 0, 0), 3),
    (35, 35, 48, 64, (5, 5), 1, (2, 2, 2, 2), 3),
    (35, 35, 64, 96, (3, 3), 1, (1, 1, 1, 1), 4),
    (35, 35, 96, 96, (3, 3), 1, (1, 1, 1, 1), 3),
    (35, 35, 288, 384, (3, 3), 2, (0, 0, 0, 0), 1),
    (17, 17, 768, 192, (1, 1), 1, (0, 0, 0, 0), 12),
    (17, 17, 768, 160, (1, 1), 1, (0, 0, 0, 0), 4),
    (17, 17, 768, 128, (1, 1), 1, (0, 0, 0, 0), 2),
    (17, 17, 160, 160, (1, 7), 1, (0, 0, 3, 3), 6),
    (17, 17, 160, 192, (7, 1), 1, (3, 3, 0, 0), 4),
    (17, 17, 192, 192, (1, 7), 1, (0, 0, 3, 3), 8),
    (17, 17, 128, 128, (7, 1), 1, (3, 3, 0, 0), 3),
    (17, 17, 192, 320, (3, 3), 2, (0, 0, 0, 0), 1),
    (8, 8, 1280, 320, (1, 1), 1, (0, 0, 0, 0), 1),
    (8, 8, 2048, 384, (1, 1), 1, (0, 0, 0, 0), 2),
    (8, 8, 2048, 448, (1, 1), 1, (0, 0, 0, 0), 1),
    (8, 8, 448, 384, (3, 3), 1, (1, 1, 1, 1), 2),
    (8, 8, 384, 384, (1, 3), 1, (0, 0, 1, 1), 8),
    (8, 8, 2048, 192, (1, 1), 1, (0, 0, 0, 0), 1),
]


# Flower CNN at 224x224 (models/flower_cnn.py): TF "SAME" padding (stride 2 on an even size pads the end only)
FLOWER = [
    (224, 224, 8, 32, (3, 3), 2, (0, 1, 0, 1)),
    (112, 112, 32, 64, (3, 3), 1, (1, 1, 1, 1)),
    (56, 56, 64, 128, (3, 3), 1, (1, 1, 1, 1)),
    (28, 28, 128, 256, (3, 3), 1, (1, 1, 1, 1)),
    (14, 14, 256, 512, (3, 3), 1, (1, 1, 1, 1)),
]


def normalise(shape):
    """(H, W, Cin, Cout, (kh, kw), stride, pads4, weight) from either table's row format."""
    H, W, Cin, Cout, k, s, pad = shape[:7]
    k = (k, k) if isinstance(k, int) else tuple(k)
    pad = (pad,) * 4 if isinstance(pad, int) else tuple(pad)
    return H, W, Cin, Cout, k, s, pad, (shape[7] if len(shape) > 7 else 1)


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--reference", action="store_true")
    ap.add_argument("--only", default="", help="comma-separated row indices of the shape table (default: all rows)")
    ap.add_argument("--shapes", default="resnet", choices=["resnet", "inception", "flower"])
    ap.add_argument("--fwd_only", action="store_true")
    args = ap.parse_args()
    B = args.batch
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0, "ref_fwd": 0.0, "ref_bwd": 0.0}
    print("%-28s %8s | %8s %7s | %8s %7s | %8s %7s" % ("shape", "GFLOP", "fwd us", "TF/s", "dgrad us", "TF/s", "wgrad us", "TF/s"))
    table = {"resnet": RESNET, "inception": INCEPTION, "flower": FLOWER}[args.shapes]
    only = {int(v) for v in args.only.split(",") if v.strip()}
    for i, shape in enumerate(table):
        if only and i not in only:
            continue
        H, W, Cin, Cout, (kh, kw), s, pads, weight = normalise(shape)
        pad = pads[0]
        plan = cv.get_plan(H, W, Cin, Cout, kh, kw, s, s, pads, "cuda")
        x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
        w = (torch.randn(Cout, kh, kw, Cin, device="cuda") * 0.05).bfloat16()
        dy = torch.randn(B, plan.oh, plan.ow, Cout, device="cuda").bfloat16()
        flop = 2.0 * B * plan.oh * plan.ow * Cout * kh * kw * Cin
        tf = timeit(lambda: cv.conv_fwd(x, w, plan, want_stats=True), args.iters)
        td = timeit(lambda: cv.conv_dgrad(dy, w, plan), args.iters) if (Cin > 8 and not args.fwd_only) else 0.0
        tw = timeit(lambda: cv.conv_wgrad(x, dy, plan), args.iters) if not args.fwd_only else 0.0
        tot["fwd"] += tf * weight; tot["dgrad"] += td * weight; tot["wgrad"] += tw * weight
        k = kh
        line = "%-28s %8.2f | %8.1f %7.1f | %8.1f %7.1f | %8.1f %7.1f" % (
            "%dx%d %d->%d k%dx%d s%d" % (H, W, Cin, Cout, kh, kw, s), flop / 1e9,
            tf * 1e6, flop / tf / 1e12, td * 1e6, (flop / td / 1e12) if td else 0.0, tw * 1e6, (flop / tw / 1e12) if tw else 0.0)
        if args.reference and kh == kw:
            xn = x.permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last).requires_grad_(i > 0)
            wn = w.permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            conv = lambda: torch.nn.functional.conv2d(xn, wn, stride=s, padding=pad)
            trf = timeit(lambda: conv(), args.iters)
            yn = conv()
            g = torch.randn_like(yn)
            def bwd():
                xn.grad = None; wn.grad = None
                conv().backward(g)
            trb = timeit(bwd, args.iters) - trf
            tot["ref_fwd"] += trf; tot["ref_bwd"] += trb
            line += " | ref fwd %8.1f us bwd %8.1f us" % (trf * 1e6, trb * 1e6)
        print(line, flush=True)
    print("totals (us):", {k: round(v * 1e6, 1) for k, v in tot.items()})


if __name__ == "__main__":
    main()
