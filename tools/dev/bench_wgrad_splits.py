"""Sweep the split count of the weight-gradient kernel over the ResNet-50 shapes (default tile of each shape)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_tensorflow_amd.ops import conv as cv
BF16 = torch.bfloat16
dev = torch.device("cuda", 0)
def timeit(fn, iters=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
B = int(os.environ.get("B", "2048"))
shapes = [(56, 64, 64, 3, 1), (28, 128, 128, 3, 1), (14, 256, 256, 3, 1), (7, 512, 512, 3, 1),
          (56, 64, 256, 1, 1), (28, 128, 512, 1, 1), (14, 256, 1024, 1, 1), (7, 512, 2048, 1, 1),
          (56, 256, 64, 1, 1), (28, 512, 128, 1, 1), (14, 1024, 256, 1, 1), (7, 2048, 512, 1, 1),
          (56, 64, 64, 1, 1), (28, 128, 128, 1, 1), (14, 256, 256, 1, 1), (7, 512, 512, 1, 1),
          (28, 256, 512, 1, 1), (14, 512, 1024, 1, 1), (7, 1024, 2048, 1, 1)]
targets = [256, 512, 768, 1024, 1536, 2048]
for hw, cin, cout, k, s in shapes:
    pad = (k // 2,) * 4
    plan = cv.get_plan(hw, hw, cin, cout, k, k, s, s, pad, dev)
    x = torch.randn(B, hw, hw, cin, device=dev).to(BF16)
    dy = torch.randn(B, plan.oh, plan.ow, cout, device=dev).to(BF16)
    os.environ.pop("DTFA_WGRAD_TILE", None)
    M = B * plan.oh * plan.ow
    bm, bn, bk, sp = cv.pick_wgrad_tile(M, cout, plan.K, cin)
    tiles = -(-cout // bm) * -(-plan.K // bn)
    row = "%dx%d %d->%d k%d: tile %dx%dx%d tiles %d default splits %d %.0f us |" % (
        hw, hw, cin, cout, k, bm, bn, bk, tiles, sp, timeit(lambda: cv.conv_wgrad(x, dy, plan)))
    for t in targets:
        spl = max(1, min(-(-t // tiles), max(1, -(-M // bk) // 8)))
        os.environ["DTFA_WGRAD_TILE"] = "%d,%d,%d,%d" % (bm, bn, bk, spl)
        row += "  %d(%d): %.0f" % (t, spl, timeit(lambda: cv.conv_wgrad(x, dy, plan)))
    os.environ.pop("DTFA_WGRAD_TILE", None)
    print(row, flush=True)
