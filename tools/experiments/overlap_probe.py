"""Do a data-gradient launch and a weight-gradient launch of different layers run faster side by side (two streams)
than one after the other?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_tensorflow_amd.ops import conv as cv
dev = torch.device("cuda", 0)
B = 2048
def mk(hw, cin, cout, k):
    plan = cv.get_plan(hw, hw, cin, cout, k, k, 1, 1, (k // 2,) * 4, dev)
    x = torch.randn(B, hw, hw, cin, device=dev).bfloat16()
    dy = torch.randn(B, hw, hw, cout, device=dev).bfloat16()
    w = (torch.randn(cout, k, k, cin, device=dev) * 0.05).bfloat16()
    return plan, x, dy, w
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
side = torch.cuda.Stream()
pairs = [((56, 256, 64, 1), (56, 64, 64, 3)), ((56, 64, 64, 3), (56, 64, 256, 1)), ((28, 512, 128, 1), (28, 128, 128, 3)),
         ((14, 256, 256, 3), (14, 256, 1024, 1)), ((14, 1024, 256, 1), (14, 256, 256, 3)), ((56, 64, 256, 1), (56, 256, 64, 1))]
for da, wa in pairs:
    pd, xd, dyd, wd = mk(*da)
    pw, xw, dyw, ww = mk(*wa)
    dgrad = lambda: cv.conv_dgrad(dyd, wd, pd)
    wgrad = lambda: cv.conv_wgrad(xw, dyw, pw)
    def both():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            wgrad()
        dgrad()
        main.wait_stream(side)
    a, b, c = t(dgrad), t(wgrad), t(both)
    print("dgrad %s %.0f us + wgrad %s %.0f us = %.0f | side by side %.0f us (%.1f%%)" % (da, a, wa, b, a + b, c, 100 * (c / (a + b) - 1)), flush=True)
