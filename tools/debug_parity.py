import copy, torch, sys
sys.path.insert(0, '.')
from distributed_tensorflow_amd.models import build_model
from distributed_tensorflow_amd.models.layers import cast_for_device
from distributed_tensorflow_amd.ops import functional as F
torch.manual_seed(0)
name = sys.argv[1] if len(sys.argv) > 1 else "flower_cnn"
kw = {"dropout_rate": 0.0} if name == "flower_cnn" else {"zero_init_residual": False}
cpu = build_model(name, device="cpu", **kw).train()
for p in cpu.parameters():
    if p.dim() > 1: p.data = p.data.bfloat16().float()
gpu = cast_for_device(copy.deepcopy(cpu), "cuda").train()
g = torch.Generator().manual_seed(0)
b, size = 16, 64
x = torch.zeros(b, size, size, 8); x[..., :3] = torch.randn(b, size, size, 3, generator=g)
y = torch.randint(0, 5, (b,), generator=g)
xb = x.bfloat16()
def run(model, xin, yin):
    out = model(xin)
    logits, reg = out if isinstance(out, tuple) else (out, None)
    loss, _, _, correct = F.softmax_xent(logits, yin, 5)
    total = loss if reg is None else loss + reg
    total.backward()
    return loss.detach().float().cpu(), logits.detach().float().cpu()
lc, lgc = run(cpu, xb.float(), y)
lg, lgg = run(gpu, xb.cuda(), y.cuda())
gpu2 = cast_for_device(copy.deepcopy(cpu), "cuda").train()
with F.reference_mode():
    l2, lg2 = run(gpu2, xb.cuda(), y.cuda())
g2 = {n: p.grad.float().cpu() for n, p in gpu2.named_parameters() if p.grad is not None}
print("loss", lc.item(), lg.item(), "logit rel", ((lgc[:, :5]-lgg[:, :5]).norm()/lgc[:, :5].norm()).item())
gc = {n: p.grad.float() for n, p in cpu.named_parameters() if p.grad is not None}
gg = {n: p.grad.float().cpu() for n, p in gpu.named_parameters() if p.grad is not None}
for n in gc:
    a, b_ = gc[n], gg[n]
    print("%-40s rel %.4f  cos %.4f  norm %.3e | torch-bf16 rel %.4f" % (n, ((a-b_).norm()/(a.norm()+1e-9)).item(), torch.nn.functional.cosine_similarity(a.flatten(), b_.flatten(), dim=0).item(), a.norm().item(), ((a-g2[n]).norm()/(a.norm()+1e-9)).item()))
