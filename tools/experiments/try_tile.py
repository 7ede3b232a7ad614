"""Time one forced tile against the default on a few layer shapes (fwd with statistics, dgrad)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_tensorflow_amd.ops import conv as cv
dev = torch.device("cuda", 0)
tile = sys.argv[1]
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B, hw, cin, cout, k in [(1024, 112, 32, 64, 3), (2048, 56, 64, 64, 3), (2048, 56, 256, 64, 1), (256, 149, 32, 32, 3), (256, 147, 32, 64, 3)]:
    plan = cv.get_plan(hw, hw, cin, cout, k, k, 1, 1, (k // 2,) * 4, dev)
    x = torch.randn(B, hw, hw, cin, device=dev).bfloat16()
    dy = torch.randn(B, hw, hw, cout, device=dev).bfloat16()
    w = (torch.randn(cout, k, k, cin, device=dev) * 0.05).bfloat16()
    os.environ.pop("DTFA_CONV_TILE", None)
    f0, d0 = t(lambda: cv.conv_fwd(x, w, plan, want_stats=True)), t(lambda: cv.conv_dgrad(dy, w, plan))
    ref_f, ref_d = cv.conv_fwd(x, w, plan, want_stats=True)[0], cv.conv_dgrad(dy, w, plan)
    os.environ["DTFA_CONV_TILE"] = tile
    try:
        f1, d1 = t(lambda: cv.conv_fwd(x, w, plan, want_stats=True)), t(lambda: cv.conv_dgrad(dy, w, plan))
        ok = torch.equal(cv.conv_fwd(x, w, plan, want_stats=True)[0], ref_f) and torch.equal(cv.conv_dgrad(dy, w, plan), ref_d)
    except Exception as e:
        f1 = d1 = float("nan"); ok = str(e)[:60]
    os.environ.pop("DTFA_CONV_TILE", None)
    print("%dx%dx%d %d->%d k%d: fwd %.0f -> %.0f us, dgrad %.0f -> %.0f us, same bits: %s" % (B, hw, hw, cin, cout, k, f0, f1, d0, d1, ok), flush=True)
