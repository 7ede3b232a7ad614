"""Time conv_p8_kernel against the ordinary implicit-GEMM kernel on the image layers (A/B inside one process)."""
import sys
import torch
from distributed_tensorflow_amd.ops import conv as cv

CASES = {"stem": (2048, 224, 112, 8, 64, 7, 4, 2, 1, (3, 3, 2, 1)),
         "flower1": (1024, 224, 224, 8, 32, 3, 3, 2, 2, (0, 1, 0, 1))}


def main():
    for name in (sys.argv[1:] or list(CASES)):
        B, H, W, Cin, Cout, kh, kw, sy, sx, pads = CASES[name]
        x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
        w = (torch.randn(Cout, kh, kw, Cin, device="cuda") * 0.05).bfloat16()
        plan = cv.get_plan(H, W, Cin, Cout, kh, kw, sy, sx, pads, "cuda")
        bias = torch.zeros(Cout, device="cuda")
        for mode, kwargs in (("stats", dict(want_stats=True)), ("bias+relu", dict(bias=bias, relu=True))):
            for p8 in (False, True, False, True):
                cv._P8 = p8
                for _ in range(3):
                    cv.conv_fwd(x, w, plan, **kwargs)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    cv.conv_fwd(x, w, plan, **kwargs)
                e1.record()
                torch.cuda.synchronize()
                print("%s %s p8=%d: %.1f us" % (name, mode, p8, e0.elapsed_time(e1) * 100), flush=True)


if __name__ == "__main__":
    main()
