"""Tile sweep for small launches: each candidate is captured 10x into a HIP graph, so the time is the kernel's, not
the host's enqueue.  usage: sweep_small.py BATCH [inception|resnet|flower]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_tensorflow_amd.ops import conv as cv
from tools.bench_conv import FLOWER, INCEPTION, RESNET, normalise
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
TABLE = {"inception": INCEPTION, "flower": FLOWER}.get(sys.argv[2] if len(sys.argv) > 2 else "inception", RESNET)
FWD = [(bm, bn, bk, nst) for (bm, bn) in ((128, 192), (128, 128), (128, 64), (64, 64)) for bk in (32, 64) for nst in (2, 3)]
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
for shape in TABLE:
    H, W, Cin, Cout, (kh, kw), s, pads, _ = normalise(shape)
    plan = cv.get_plan(H, W, Cin, Cout, kh, kw, s, s, pads, "cuda")
    x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
    w = (torch.randn(Cout, kh, kw, Cin, device="cuda") * 0.05).bfloat16()
    fn = lambda: cv.conv_fwd(x, w, plan, want_stats=True)
    res = []
    for t in FWD:
        os.environ["DTFA_CONV_TILE"] = ",".join(map(str, t))
        try:
            res.append((graph_time(fn), t))
        except Exception as e:
            res.append((float("inf"), t))
    os.environ.pop("DTFA_CONV_TILE")
    auto = graph_time(fn)
    M = B * plan.oh * plan.ow
    res.sort()
    print("%-26s M %6d K %5d auto %s %6.1f | best %6.1f %s | %s" % (
        "%dx%d %d->%d k%dx%d s%d" % (H, W, Cin, Cout, kh, kw, s), M, plan.K, cv.pick_fwd_tile(M, Cout, plan.K, True), auto,
        res[0][0], res[0][1], " ".join("%s:%.1f" % ("x".join(map(str, t)), v) for v, t in res[1:5])), flush=True)
