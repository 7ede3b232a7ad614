"""Two forced tiles against each other on the deep layers of ResNet-50 (fwd with statistics, dgrad).
usage: try_tile2.py <tile a> <tile b> [batch]   e.g. 256,256,64,2 256,256,32,4"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_tensorflow_amd.ops import conv as cv
dev = torch.device("cuda", 0)
ta, tb = sys.argv[1], sys.argv[2]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
def t(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for hw, cin, cout, k in [(14, 256, 256, 3), (7, 512, 512, 3), (14, 1024, 256, 1), (14, 256, 1024, 1), (7, 2048, 512, 1),
                         (7, 512, 2048, 1), (28, 512, 256, 1), (14, 1024, 512, 1)]:
    plan = cv.get_plan(hw, hw, cin, cout, k, k, 1, 1, (k // 2,) * 4, dev)
    x = torch.randn(B, hw, hw, cin, device=dev).bfloat16()
    dy = torch.randn(B, hw, hw, cout, device=dev).bfloat16()
    w = (torch.randn(cout, k, k, cin, device=dev) * 0.05).bfloat16()
    res = []
    for tile in (ta, tb):
        os.environ["DTFA_CONV_TILE"] = tile
        f, d = t(lambda: cv.conv_fwd(x, w, plan, want_stats=True)), t(lambda: cv.conv_dgrad(dy, w, plan))
        res.append((f, d, cv.conv_fwd(x, w, plan, want_stats=True)[0], cv.conv_dgrad(dy, w, plan)))
    os.environ.pop("DTFA_CONV_TILE", None)
    ok = torch.equal(res[0][2], res[1][2]) and torch.equal(res[0][3], res[1][3])
    print("%dx%dx%d %d->%d k%d: fwd %.0f -> %.0f us, dgrad %.0f -> %.0f us, same bits: %s" % (
        B, hw, hw, cin, cout, k, res[0][0], res[1][0], res[0][1], res[1][1], ok), flush=True)
