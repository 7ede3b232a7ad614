import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_tensorflow_amd.ops import conv as cv
BF16 = torch.bfloat16
dev = torch.device("cuda", 0)
def timeit(fn, iters=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
B = int(os.environ.get("B", "2048"))
shapes = [(56, 64, 64, 3, 1), (28, 128, 128, 3, 1), (14, 256, 256, 3, 1), (7, 512, 512, 3, 1), (56, 128, 128, 3, 2),
          (56, 64, 256, 1, 1), (28, 128, 512, 1, 1), (14, 256, 1024, 1, 1), (56, 256, 64, 1, 1), (14, 1024, 256, 1, 1)]
tiles = [None, (256, 128, 32), (128, 256, 32), (64, 256, 32)]
for hw, cin, cout, k, s in shapes:
    pad = (k // 2,) * 4
    plan = cv.get_plan(hw, hw, cin, cout, k, k, s, s, pad, dev)
    x = torch.randn(B, hw, hw, cin, device=dev).to(BF16)
    dy = torch.randn(B, plan.oh, plan.ow, cout, device=dev).to(BF16)
    os.environ.pop("DTFA_WGRAD_TILE", None)
    ref = cv.conv_wgrad(x, dy, plan).clone()
    row = "%dx%d %d->%d k%d s%d:" % (hw, hw, cin, cout, k, s)
    for t in tiles:
        if t is None:
            os.environ.pop("DTFA_WGRAD_TILE", None)
            name = "default%s" % (cv.pick_wgrad_tile(B * plan.oh * plan.ow, cout, plan.K, cin),)
        else:
            if t[0] > max(cout, 64) * 2 or t[1] > max(plan.K, 64) * 2:
                continue
            os.environ["DTFA_WGRAD_TILE"] = ",".join(str(v) for v in t)
            name = "%dx%d" % t[:2]
        try:
            out = cv.conv_wgrad(x, dy, plan)
            err = ((out - ref).abs().max() / ref.abs().max()).item()
            us = timeit(lambda: cv.conv_wgrad(x, dy, plan))
            row += "  %s %.0f us (err %.1e)" % (name, us, err)
        except Exception as e:
            row += "  %s FAILED %s" % (name, str(e)[:40])
    print(row, flush=True)
