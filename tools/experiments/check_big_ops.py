"""Convolution kernels on tensors beyond 2^31 elements (batch 4096 at 112x112x64 and 56x56x256): the batch is the same
half twice over, so data gradients and outputs of the two halves must be the same bits, and a weight gradient twice that
of one half."""
import torch
from distributed_tensorflow_amd.ops import conv as cv

B = 4096
h = B // 2


def dup(*shape):
    t = torch.randn(h, *shape, device="cuda").bfloat16()
    return torch.cat([t, t]), t


def halves(name, t):
    ok = torch.equal(t[:h], t[h:])
    print("%-44s halves %s" % (name, "same bits" if ok else "DIFFER: %.3g" % (t[:h].float() - t[h:].float()).abs().max().item()), flush=True)


def wg(name, plan, x2, x1, dy2, dy1):
    full = cv.conv_wgrad(x2, dy2, plan)
    one = cv.conv_wgrad(x1, dy1, plan)
    d = (full - 2 * one).abs().max().item() / (full.abs().max().item() + 1e-12)
    print("%-44s wgrad of the whole against twice a half: %.3g %s" % (name, d, "ok" if d < 1e-3 else "DIFFERS"), flush=True)


def main():
    # the pixel-pair stem
    plan = cv.get_plan(224, 112, 8, 64, 7, 4, 2, 1, (3, 3, 2, 1), "cuda")
    x2, x1 = dup(224, 112, 8)
    w = (torch.randn(64, 7, 4, 8, device="cuda") * 0.05).bfloat16()
    y, part = cv.conv_fwd(x2, w, plan, want_stats=True)
    halves("stem forward [4096,112,112,64]", y)
    dy2, dy1 = dup(112, 112, 64)
    wg("stem", plan, x2, x1, dy2, dy1)
    del x2, x1, y, dy2, dy1
    # 56x56 64 -> 256 (expanding 1x1): output and its gradient beyond 2^31 elements
    plan = cv.get_plan(56, 56, 64, 256, 1, 1, 1, 1, (0, 0, 0, 0), "cuda")
    x2, x1 = dup(56, 56, 64)
    w = (torch.randn(256, 1, 1, 64, device="cuda") * 0.1).bfloat16()
    y, _ = cv.conv_fwd(x2, w, plan, want_stats=True)
    halves("1x1 64->256 forward [4096,56,56,256]", y)
    dy2, dy1 = dup(56, 56, 256)
    halves("1x1 64->256 data gradient (reads 256 ch)", cv.conv_dgrad(dy2, w, plan))
    wg("1x1 64->256", plan, x2, x1, dy2, dy1)
    # 56x56 256 -> 64 (reducing 1x1): input and its gradient beyond 2^31 elements
    plan2 = cv.get_plan(56, 56, 256, 64, 1, 1, 1, 1, (0, 0, 0, 0), "cuda")
    w2 = (torch.randn(64, 1, 1, 256, device="cuda") * 0.05).bfloat16()
    y2, _ = cv.conv_fwd(dy2, w2, plan2, want_stats=True)
    halves("1x1 256->64 forward (reads 256 ch)", y2)
    halves("1x1 256->64 data gradient [4096,56,56,256]", cv.conv_dgrad(x2, w2, plan2, addend=dy2))
    wg("1x1 256->64", plan2, dy2, dy1, x2, x1)
    del y, y2
    # 56x56 256 -> 128 stride-2 3x3's predecessor: 1x1 256 -> 128 at 56x56
    plan3 = cv.get_plan(56, 56, 256, 128, 1, 1, 1, 1, (0, 0, 0, 0), "cuda")
    w3 = (torch.randn(128, 1, 1, 256, device="cuda") * 0.05).bfloat16()
    y3, _ = cv.conv_fwd(dy2, w3, plan3, want_stats=True)
    halves("1x1 256->128 forward (reads 256 ch)", y3)
    g3, g31 = y3, y3[:h].contiguous()
    halves("1x1 256->128 data gradient [4096,56,56,256]", cv.conv_dgrad(g3, w3, plan3))
    wg("1x1 256->128", plan3, dy2, dy1, g3, g31)
    print("peak memory %.1f GB" % (torch.cuda.max_memory_allocated() / 2**30))


if __name__ == "__main__":
    main()
