import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distributed_tensorflow_amd.models.resnet import Bottleneck
from distributed_tensorflow_amd.models.layers import cast_for_device
from distributed_tensorflow_amd.ops import conv as cv, functional as F
BF16 = torch.bfloat16
torch.manual_seed(0)
blocks = [Bottleneck(256, 64, 1, "a", zero_init_residual=False), Bottleneck(256, 64, 1, "b", zero_init_residual=False)]
cpu_blocks = copy.deepcopy(blocks)
for b in blocks:
    cast_for_device(b, "cuda").train()
for cb, b in zip(cpu_blocks, blocks):
    cb.train()
    for pc, pg in zip(cb.parameters(), b.parameters()):
        pc.data = pg.data.float().cpu()
x0 = torch.randn(8, 14, 14, 256).to(BF16)
dz = torch.randn(8, 14, 14, 256).to(BF16)
def run(blks, dev, fold, dt):
    old = F.EXPAND_FOLD; F.EXPAND_FOLD = fold
    for b in blks: b.zero_grad(set_to_none=True)
    x = x0.to(dev, dt).clone().requires_grad_(True)
    tap = cv.BnBackwardTap() if dev == "cuda" else None
    h = blks[0](x, in_tap=None, out_tap=tap)
    out = blks[1](h, in_tap=tap, out_tap=None)
    out.backward(dz.to(dev, dt))
    F.EXPAND_FOLD = old
    names = [n for b in blks for n, _ in b.named_parameters()]
    return names + ["x"], [p.grad.detach().float().cpu() for b in blks for p in b.parameters()] + [x.grad.float().cpu()], out.detach().float().cpu()
names, ref, oref = run(cpu_blocks, "cpu", False, torch.float32)
_, gf, of = run(blocks, "cuda", True, BF16)
_, gb, ob = run(blocks, "cuda", False, BF16)
print("out err fold %.4f base %.4f" % ((of-oref).abs().max()/oref.abs().max(), (ob-oref).abs().max()/oref.abs().max()))
for i,(n, r, a, b) in enumerate(zip(names, ref, gf, gb)):
    s = r.abs().max().item() + 1e-9
    rs = r.pow(2).mean().sqrt().item() + 1e-9
    print("%2d %-14s max|ref| %9.3f  fold: max %.4f rms %.4f   base: max %.4f rms %.4f" % (i, n, s, (a-r).abs().max()/s, (a-r).pow(2).mean().sqrt()/rs, (b-r).abs().max()/s, (b-r).pow(2).mean().sqrt()/rs))
