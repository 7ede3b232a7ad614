"""A forced weight-gradient tile against the default on the deep ResNet-50 layers: time and agreement.
usage: try_wgrad_tile.py <bmw,bnw,bkp[,splits]> [batch]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_tensorflow_amd.ops import conv as cv
dev = torch.device("cuda", 0)
tile = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
def t(fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for hw, cin, cout, k in [(14, 256, 256, 3), (14, 256, 256, 3), (7, 512, 512, 3), (14, 1024, 256, 1), (14, 256, 1024, 1), (7, 2048, 512, 1),
                         (7, 512, 2048, 1), (28, 512, 256, 1), (14, 1024, 512, 1), (28, 128, 512, 1)]:
    plan = cv.get_plan(hw, hw, cin, cout, k, k, 1, 1, (k // 2,) * 4, dev)
    x = torch.randn(B, hw, hw, cin, device=dev).bfloat16()
    dy = torch.randn(B, hw, hw, cout, device=dev).bfloat16()
    os.environ.pop("DTFA_WGRAD_TILE", None)
    t0 = t(lambda: cv.conv_wgrad(x, dy, plan))
    ref = cv.conv_wgrad(x, dy, plan)
    os.environ["DTFA_WGRAD_TILE"] = tile
    t1 = t(lambda: cv.conv_wgrad(x, dy, plan))
    got = cv.conv_wgrad(x, dy, plan)
    cfg = cv.pick_wgrad_tile(B * hw * hw, cout, plan.K, cin)
    os.environ.pop("DTFA_WGRAD_TILE", None)
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    print("%dx%dx%d %d->%d k%d: wgrad %.0f -> %.0f us (%s), max err / max %.2e" % (B, hw, hw, cin, cout, k, t0, t1, cfg, err), flush=True)
