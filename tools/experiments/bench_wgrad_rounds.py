"""Weight-gradient split counts against the number of resident workgroups: for the layers with taps, time splits
chosen so that tiles x splits fills r = 1 .. 6 rounds of resident workgroups exactly, next to the default choice."""
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
# workgroups resident per CU (registers / LDS of each instantiation)
PER_CU = {(256, 128, 32): 2, (128, 256, 32): 2, (64, 256, 32): 2, (128, 128, 32): 3, (64, 128, 32): 4, (128, 64, 32): 4,
          (64, 64, 32): 6, (128, 128, 64): 2, (64, 128, 64): 3, (128, 64, 64): 3, (64, 64, 64): 5}
shapes = [(56, 64, 64, 3, 1), (28, 128, 128, 3, 1), (14, 256, 256, 3, 1), (7, 512, 512, 3, 1),
          (56, 128, 128, 3, 2), (28, 256, 256, 3, 2), (14, 512, 512, 3, 2)]
if os.environ.get("SHAPES") == "flower":
    B = int(os.environ.get("B", "1024"))
    shapes = [(112, 32, 64, 3, 1), (56, 64, 128, 3, 1), (28, 128, 256, 3, 1), (14, 256, 512, 3, 1)]
for hw, cin, cout, k, s in shapes:
    pad = (k // 2,) * 4
    plan = cv.get_plan(hw, hw, cin, cout, k, k, s, s, pad, dev)
    x = torch.randn(B, hw, hw, cin, device=dev).to(BF16)
    dy = torch.randn(B, plan.oh, plan.ow, cout, device=dev).to(BF16)
    os.environ.pop("DTFA_WGRAD_TILE", None)
    M = B * plan.oh * plan.ow
    bm, bn, bk, sp = cv.pick_wgrad_tile(M, cout, plan.K, cin)
    tiles = -(-cout // bm) * -(-plan.K // bn)
    slots = PER_CU[(bm, bn, bk)] * 256
    row = "%dx%d %d->%d k%d s%d: tile %dx%dx%d tiles %d slots %d default splits %d (%.2f rounds) %.0f us |" % (
        hw, hw, cin, cout, k, s, bm, bn, bk, tiles, slots, sp, tiles * sp / slots, timeit(lambda: cv.conv_wgrad(x, dy, plan)))
    for r in (1, 2, 3, 4, 5, 6):
        spl = max(1, (slots * r) // tiles)
        os.environ["DTFA_WGRAD_TILE"] = "%d,%d,%d,%d" % (bm, bn, bk, spl)
        row += "  r%d(%d): %.0f" % (r, spl, timeit(lambda: cv.conv_wgrad(x, dy, plan)))
    os.environ.pop("DTFA_WGRAD_TILE", None)
    print(row, flush=True)
