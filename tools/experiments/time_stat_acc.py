"""What the fp64 accumulator statistics cost a small convolution launch: the same launch with accumulators, with
partial rows and with no statistics at all, each captured 20x into a HIP graph (kernel time, not enqueue time)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_tensorflow_amd.ops import conv as cv  # noqa: E402

BF16 = torch.bfloat16
SHAPES = [(32, 17, 17, 768, 192, 1, 1), (32, 17, 17, 160, 160, 1, 7), (32, 35, 35, 288, 64, 1, 1), (32, 8, 8, 2048, 384, 1, 1),
          (32, 35, 35, 96, 96, 3, 3), (32, 17, 17, 192, 192, 7, 1)]


def graph_time(fn, reps=20, rounds=30):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(rounds):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (rounds * reps) * 1e3


for B, H, W, Cin, Cout, kh, kw in SHAPES:
    x = torch.randn(B, H, W, Cin, device="cuda").to(BF16)
    w = (torch.randn(Cout, kh, kw, Cin, device="cuda") * 0.05).to(BF16)
    pads = cv.resolve_pads("SAME", H, W, kh, kw, 1, 1)
    plan = cv.get_plan(H, W, Cin, Cout, kh, kw, 1, 1, pads, "cuda")
    acc = torch.zeros(2 * Cout, dtype=torch.float64, device="cuda")
    bias = torch.zeros(Cout, device="cuda")
    t_acc = graph_time(lambda: cv.conv_fwd(x, w, plan, stat_acc=acc))
    t_rows = graph_time(lambda: cv.conv_fwd(x, w, plan, want_stats=True))
    t_none = graph_time(lambda: cv.conv_fwd(x, w, plan, bias=bias, relu=True))
    print("%dx%d %d->%d k%dx%d: accumulators %.1f us, partial rows %.1f us, bias+relu epilogue %.1f us"
          % (H, W, Cin, Cout, kh, kw, t_acc, t_rows, t_none), flush=True)
