"""Graph-timed comparison of the in-workgroup reduction split (1 / 2 / 4 wave groups) on the small-launch tile."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_tensorflow_amd.ops import conv as cv
from tools.bench_conv import INCEPTION, normalise
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
REP = 10
def graph_time(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * REP) * 1e3
for shape in INCEPTION:
    H, W, Cin, Cout, (kh, kw), s, pads, _ = normalise(shape)
    plan = cv.get_plan(H, W, Cin, Cout, kh, kw, s, s, pads, "cuda")
    M = B * plan.oh * plan.ow
    if cv.pick_fwd_tile(M, Cout, plan.K, True) != (64, 64, 64, 2):
        continue
    x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
    w = (torch.randn(Cout, kh, kw, Cin, device="cuda") * 0.05).bfloat16()
    fn = lambda: cv.conv_fwd(x, w, plan, want_stats=True)
    res = {}
    ref = None
    for ks in ("0", "2", "4"):
        cv._KSPLIT = ks
        y = fn()[0]
        if ref is None: ref = y
        err = (y.float() - ref.float()).abs().max().item() / (ref.float().abs().max().item() + 1e-6)
        res[ks] = (graph_time(fn), err)
    print("%-26s M %6d K %5d tiles %4d | 1: %5.1f  2: %5.1f (%.4f)  4: %5.1f (%.4f)" % (
        "%dx%d %d->%d k%dx%d s%d" % (H, W, Cin, Cout, kh, kw, s), M, plan.K, -(-M // 64) * -(-Cout // 64),
        res["0"][0], res["2"][0], res["2"][1], res["4"][0], res["4"][1]), flush=True)
