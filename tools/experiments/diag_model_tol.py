"""What the train-mode parity test of tests/test_gpu_models.py actually needs: loss and gradient deviations of the HIP
path and of PyTorch's own bf16 GPU ops from the fp32 CPU oracle."""
import copy, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from distributed_tensorflow_amd.models import build_model
from distributed_tensorflow_amd.models.layers import cast_for_device
from distributed_tensorflow_amd.ops import functional as F
from test_gpu_models import _images, _grads_by_name

CFG = [tuple(c.split(":")) for c in os.environ.get("CFG", "flower_cnn:64:16,resnet50:96:16").split(",")]
for name, size, batch in [(n, int(s_), int(b)) for n, s_, b in CFG]:
    for seed in (0, 1):
        torch.manual_seed(seed)
        kw = {"dropout_rate": 0.0} if name == "flower_cnn" else {"zero_init_residual": False}
        cpu = build_model(name, device="cpu", **kw).train()
        damp = float(os.environ.get("DAMP", "1"))
        if damp != 1:
            for n, p in cpu.named_parameters():
                if n.endswith("conv3.gamma"):
                    p.data.mul_(damp)
        for p in cpu.parameters():
            if p.dim() > 1:
                p.data = p.data.bfloat16().float()
        gpu = cast_for_device(copy.deepcopy(cpu), "cuda").train()
        x, y = _images(batch, size, seed=seed) if seed else _images(batch, size)
        xb = x.bfloat16()

        def run(model, xin, yin):
            out = model(xin)
            logits, reg = out if isinstance(out, tuple) else (out, None)
            loss, _, _, correct = F.softmax_xent(logits, yin, 5)
            total = loss if reg is None else loss + reg
            total.backward()
            return loss.detach().float().cpu()

        lc = run(cpu, xb.float(), y)
        lg = run(gpu, xb.cuda(), y.cuda())
        gref = cast_for_device(copy.deepcopy(cpu), "cuda").train()
        with F.reference_mode():
            lr = run(gref, xb.cuda(), y.cuda())
        gc, gg, gr = _grads_by_name(cpu), _grads_by_name(gpu), _grads_by_name(gref)
        worst = (0, 0, 0, "")
        worst_excess = (-1e9, "")
        for n in gc:
            d = gc[n].norm().item() + 1e-6
            rel = (gc[n] - gg[n]).norm().item() / d
            rl = (gc[n] - gr[n]).norm().item() / d
            if rel > worst[0]:
                worst = (rel, rl, d, n)
            if rel - rl > worst_excess[0]:
                worst_excess = (rel - rl, n, rel, rl)
        cat = lambda g: torch.cat([g[n].flatten() for n in gc])
        vc, vg, vr = cat(gc), cat(gg), cat(gr)
        glob_h = ((vc - vg).norm() / vc.norm()).item()
        glob_l = ((vc - vr).norm() / vc.norm()).item()
        small = max((((gc[n] - gg[n]).norm() - 1.5 * (gc[n] - gr[n]).norm()) / (gc[n].norm() + 1e-6)).item() for n in gc)
        gl = vc.norm().item()
        pert = max(((gc[n] - gg[n]).norm().item() / gl, n) for n in gc)
        pert_l = max(((gc[n] - gr[n]).norm().item() / gl, n) for n in gc)
        print("   global rel hip %.4f lib %.4f | max(rel - 1.5 rel_lib) %.4f | worst tensor error / global norm: hip %.4f %s lib %.4f %s" % (glob_h, glob_l, small, pert[0], pert[1], pert_l[0], pert_l[1]))
        print(name, size, batch, "seed", seed, "loss cpu %.4f hip dev %.4f lib dev %.4f | worst rel %.4f (lib %.4f) %s | worst excess over lib %.4f %s (%.4f vs %.4f)" % (
            lc.item(), abs(lc.item() - lg.item()), abs(lc.item() - lr.item()), worst[0], worst[1], worst[3],
            worst_excess[0], worst_excess[1], worst_excess[2], worst_excess[3]), flush=True)
