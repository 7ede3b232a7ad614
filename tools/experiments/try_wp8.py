"""Time conv_wp8_kernel against the ordinary weight-gradient kernel on the ResNet-50 pixel-pair stem (A/B in one process)."""
import torch
from distributed_tensorflow_amd.ops import conv as cv


def main():
    B, H, W = 2048, 224, 112
    plan = cv.get_plan(H, W, 8, 64, 7, 4, 2, 1, (3, 3, 2, 1), "cuda")
    x = torch.randn(B, H, W, 8, device="cuda").bfloat16()
    dy = torch.randn(B, plan.oh, plan.ow, 64, device="cuda").bfloat16()
    for p8 in (False, True, False, True):
        cv._WGRAD_P8 = p8
        for _ in range(3):
            cv.conv_wgrad(x, dy, plan)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            cv.conv_wgrad(x, dy, plan)
        e1.record()
        torch.cuda.synchronize()
        print("stem wgrad p8=%d: %.1f us" % (p8, e0.elapsed_time(e1) * 100), flush=True)


if __name__ == "__main__":
    main()
