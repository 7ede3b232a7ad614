"""Strided data gradient (3x3 / stride 2, 56x56 128 -> 128 at batch 4096): time of each output-stride phase, in the
plan's order and reversed."""
import torch
from distributed_tensorflow_amd.ops import conv as cv


def main():
    B = 4096
    plan = cv.get_plan(56, 56, 128, 128, 3, 3, 2, 2, (1, 1, 1, 1), "cuda")
    dy = torch.randn(B, 28, 28, 128, device="cuda").bfloat16()
    w = (torch.randn(128, 3, 3, 128, device="cuda") * 0.03).bfloat16()
    phases = list(plan.phases)
    for order in ("plan", "reversed", "plan"):
        plan.phases = phases if order == "plan" else phases[::-1]
        for single in [None] + list(range(4)):
            saved = plan.phases
            if single is not None:
                plan.phases = [saved[single]]
            for _ in range(2):
                cv.conv_dgrad(dy, w, plan)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                cv.conv_dgrad(dy, w, plan)
            e1.record()
            torch.cuda.synchronize()
            ph = plan.phases[0]
            print("%-8s %-22s %8.1f us" % (order, "all four" if single is None else "phase (%d,%d) T=%d" % (ph.oy0, ph.ox0, ph.T),
                                           e0.elapsed_time(e1) * 200), flush=True)
            plan.phases = saved
    plan.phases = phases


if __name__ == "__main__":
    main()
