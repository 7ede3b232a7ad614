"""Vendor GEMM on the weight-gradient shapes of the deep 1x1 layers (dy^T x with the pixels as the reduction)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_tensorflow_amd.ops import conv as cv
dev = torch.device("cuda", 0)
B = 2048
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for hw, cin, cout in [(14, 256, 1024), (14, 1024, 256), (7, 512, 2048), (7, 2048, 512), (28, 128, 512), (14, 512, 1024), (7, 1024, 2048), (14, 256, 256), (7, 512, 512)]:
    plan = cv.get_plan(hw, hw, cin, cout, 1, 1, 1, 1, (0, 0, 0, 0), dev)
    x = torch.randn(B, hw, hw, cin, device=dev).bfloat16()
    dy = torch.randn(B, hw, hw, cout, device=dev).bfloat16()
    x2, d2 = x.view(-1, cin), dy.view(-1, cout)
    mine = t(lambda: cv.conv_wgrad(x, dy, plan))
    lib = t(lambda: torch.matmul(d2.t(), x2))
    fl = 2.0 * x2.shape[0] * cin * cout
    print("%dx%d %d->%d: conv_wgrad %.0f us (%.0f TF/s) | torch.matmul(dy^T, x) %.0f us (%.0f TF/s)" % (
        hw, hw, cin, cout, mine, fl / mine / 1e6, lib, fl / lib / 1e6), flush=True)
