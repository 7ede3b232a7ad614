"""Fused batch-norm prologue vs the separate elementwise pass + GEMM it replaces, on the bottleneck shapes of
ResNet-50 (csrc/hip/conv_pro.hip).

    python tools/bench_pro.py [--batch 1024] [--iters 10]

forward : z = relu(y*s + t + r) ; y1 = conv1x1(z) with statistics      (block output -> next block's conv1)
backward: dy = ca*g + cb*y + cc ; dx = dgrad1x1(dy) (+ batch-norm sums) (conv3 of a block)
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import conv as cv  # noqa: E402
from distributed_tensorflow_amd.ops import kernels as K  # noqa: E402

BF16 = torch.bfloat16


def timeit(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--variants", type=int, default=0, help="also time the forward kernel without side outputs")
    ap.add_argument("--shapes", default="56:256:64,56:256:128,28:512:128,28:512:256,14:1024:256,14:1024:512,7:2048:512")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    rows_out = []
    for spec in args.shapes.split(","):
        hw, k, n = (int(v) for v in spec.split(":"))
        B = args.batch
        y = torch.randn(B, hw, hw, k, device=dev).to(BF16)
        r = torch.randn(B, hw, hw, k, device=dev).to(BF16)
        w = (torch.randn(n, 1, 1, k, device=dev) * k ** -0.5).to(BF16)
        sc, sh = torch.rand(k, device=dev) + 0.5, torch.randn(k, device=dev) * 0.1
        plan = cv.get_plan(hw, hw, k, n, 1, 1, 1, 1, (0, 0, 0, 0), dev)
        z = torch.empty_like(y)
        mask = torch.empty(y.numel() // 8, dtype=torch.uint8, device=dev)

        def fwd_sep():
            zz, _ = K.bn_apply(y, sc, sh, True, r, want_mask=True)
            cv.conv_fwd(zz, w, plan, want_stats=True)

        def fwd_conv_only():
            cv.conv_fwd(z, w, plan, want_stats=True)

        def fwd_fused():
            cv.conv_pro(y, w, n, cv.PRO_FWD_RES, sc, sh, src2=r, side=z, side_mask=mask, want_stats=True)

        def fwd_fused_nomask():
            cv.conv_pro(y, w, n, cv.PRO_FWD_RES, sc, sh, src2=r, side=z, want_stats=True)

        def fwd_fused_noside():
            cv.conv_pro(y, w, n, cv.PRO_FWD_RES, sc, sh, src2=r, want_stats=True)

        def fwd_fused_1src():
            cv.conv_pro(y, w, n, cv.PRO_FWD, sc, sh, relu=True, want_stats=True)

        # backward: conv with cout = k (the wide side), cin = n; dgrad maps [.., k] -> [.., n]
        planb = cv.get_plan(hw, hw, n, k, 1, 1, 1, 1, (0, 0, 0, 0), dev)
        wb = (torch.randn(k, 1, 1, n, device=dev) * n ** -0.5).to(BF16)
        wd = wb.view(k, n).t().contiguous()
        dz = torch.randn(B, hw, hw, k, device=dev).to(BF16)
        coef = torch.randn(3, k, device=dev) * 0.3
        ylow = torch.randn(B, hw, hw, n, device=dev).to(BF16)
        tap = cv.BnBackwardTap()
        tap.y, tap.mean, tap.invstd = ylow, torch.zeros(n, device=dev), torch.ones(n, device=dev)
        tap.scale, tap.shift, tap.relu_mode = torch.ones(n, device=dev), torch.zeros(n, device=dev), 2
        side = torch.empty_like(dz)

        def bwd_sep():
            dy, _ = K.bn_backward_apply(dz, y, coef, mask=mask)
            cv.conv_dgrad(dy, wb, planb, bn_tap=tap)

        def bwd_fused():
            cv.conv_pro(dz, wd, n, cv.PRO_BWD_MASK, coef[0], coef[1], coef[2], src2=y, gate=mask, side=side, bn_tap=tap)

        t = {"shape": "%dx%d %d->%d" % (hw, hw, k, n), "tile": cv.pick_pro_tile(B * hw * hw, n, k, cv.PRO_FWD_RES),
             "fwd_sep_us": timeit(fwd_sep, args.iters), "fwd_conv_only_us": timeit(fwd_conv_only, args.iters),
             "fwd_fused_us": timeit(fwd_fused, args.iters),
             "fwd_fused_nomask_us": timeit(fwd_fused_nomask, args.iters) if args.variants else 0.0,
             "fwd_fused_noside_us": timeit(fwd_fused_noside, args.iters) if args.variants else 0.0,
             "fwd_fused_1src_us": timeit(fwd_fused_1src, args.iters) if args.variants else 0.0,
             "bwd_sep_us": timeit(bwd_sep, args.iters), "bwd_fused_us": timeit(bwd_fused, args.iters)}
        gb = y.numel() * 2 / 1e9
        t["fwd_fused_TBps"] = round((3 * gb + gb / 16 + gb * n / k) / t["fwd_fused_us"] * 1e3, 2)
        t["bwd_fused_TBps"] = round((3 * gb + gb / 16 + 2 * gb * n / k) / t["bwd_fused_us"] * 1e3, 2)
        for kk in list(t):
            if kk.endswith("_us"):
                t[kk] = round(t[kk], 1)
        print(json.dumps(t), flush=True)
        rows_out.append(t)
        del y, r, z, dz, side, ylow
        torch.cuda.empty_cache()
    tot = {k: round(sum(r[k] for r in rows_out), 1) for k in rows_out[0] if k.endswith("_us")}
    print(json.dumps({"totals": tot}))


if __name__ == "__main__":
    main()
