"""Run one convolution shape a few times (fwd, dgrad, wgrad) for counter collection under rocprofv3 --pmc."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import conv as cv  # noqa: E402

H, W, Cin, Cout, k, s, pad, B = [int(v) for v in sys.argv[1:9]]
which = sys.argv[9] if len(sys.argv) > 9 else "fwd"
plan = cv.get_plan(H, W, Cin, Cout, k, k, s, s, (pad,) * 4, "cuda")
x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
w = (torch.randn(Cout, k, k, Cin, device="cuda") * 0.05).bfloat16()
dy = torch.randn(B, plan.oh, plan.ow, Cout, device="cuda").bfloat16()
for _ in range(3):
    if which == "fwd":
        cv.conv_fwd(x, w, plan, want_stats=True)
    elif which == "dgrad":
        cv.conv_dgrad(dy, w, plan)
    else:
        cv.conv_wgrad(x, dy, plan)
torch.cuda.synchronize()
