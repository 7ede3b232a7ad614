"""Sweep tile configurations of the implicit-GEMM kernels per layer shape (fwd / dgrad / wgrad)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import conv as cv  # noqa: E402
from tools.bench_conv import FLOWER, INCEPTION, RESNET, normalise, timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
which = sys.argv[2] if len(sys.argv) > 2 else "fwd"
FWD = [(bm, bn, bk, nst) for (bm, bn) in ((128, 192), (128, 128), (128, 64), (64, 64)) for bk in (32, 64) for nst in (2, 3)]
FWD += [(256, 128, 32, 3), (256, 256, 64, 2)]
TABLE = {"inception": INCEPTION, "flower": FLOWER}.get(sys.argv[3] if len(sys.argv) > 3 else "", RESNET)
WG = [(128, 128, 32), (64, 128, 32), (128, 64, 32), (64, 64, 32), (128, 128, 64), (64, 128, 64)]
for shape in TABLE:
    H, W, Cin, Cout, (kh, kw), s, pads, _ = normalise(shape)
    k = kh
    plan = cv.get_plan(H, W, Cin, Cout, kh, kw, s, s, pads, "cuda")
    x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
    w = (torch.randn(Cout, kh, kw, Cin, device="cuda") * 0.05).bfloat16()
    dy = torch.randn(B, plan.oh, plan.ow, Cout, device="cuda").bfloat16()
    res = []
    if which in ("fwd", "dgrad"):
        if which == "dgrad" and Cin == 8:
            continue
        for t in FWD:
            os.environ["DTFA_CONV_TILE"] = ",".join(map(str, t))
            fn = (lambda: cv.conv_fwd(x, w, plan, want_stats=True)) if which == "fwd" else (lambda: cv.conv_dgrad(dy, w, plan))
            try:
                res.append((timeit(fn, 5) * 1e6, t))
            except Exception as e:  # noqa: BLE001
                res.append((float("inf"), t))
        os.environ.pop("DTFA_CONV_TILE")
        auto = timeit((lambda: cv.conv_fwd(x, w, plan, want_stats=True)) if which == "fwd" else (lambda: cv.conv_dgrad(dy, w, plan)), 5) * 1e6
    else:
        for t in WG:
            for sp in (0,):
                os.environ["DTFA_WGRAD_TILE"] = "%d,%d,%d" % t
                try:
                    res.append((timeit(lambda: cv.conv_wgrad(x, dy, plan), 5) * 1e6, t))
                except Exception as e:  # noqa: BLE001
                    res.append((float("inf"), t))
        os.environ.pop("DTFA_WGRAD_TILE")
        auto = timeit(lambda: cv.conv_wgrad(x, dy, plan), 5) * 1e6
    if os.environ.get("SWEEP_JSON"):
        import json
        with open(os.environ["SWEEP_JSON"], "a") as f:
            f.write(json.dumps({"which": which, "B": B, "shape": [H, W, Cin, Cout, kh, kw, s], "oh": plan.oh, "ow": plan.ow,
                                "auto": auto, "times": {",".join(map(str, t)): v for v, t in res}}) + "\n")
    res.sort()
    print("%-26s auto %7.1f | best %7.1f %s | %s" % ("%dx%d %d->%d k%dx%d s%d" % (H, W, Cin, Cout, kh, kw, s), auto, res[0][0], res[0][1],
                                                   " ".join("%s:%.0f" % ("x".join(map(str, t)), v) for v, t in res[1:4])), flush=True)
