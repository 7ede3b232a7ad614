"""End-to-end model parity: HIP kernel path (GPU, bf16) vs reference path (CPU, fp32)."""
import copy

import pytest
import torch

from distributed_tensorflow_amd.models import build_model
from distributed_tensorflow_amd.models.layers import cast_for_device
from distributed_tensorflow_amd.ops import functional as F
from distributed_tensorflow_amd.parallel.engine import DataParallelEngine

pytestmark = pytest.mark.gpu


def _images(b, size, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.zeros(b, size, size, 8)
    x[..., :3] = torch.randn(b, size, size, 3, generator=g)
    return x, torch.randint(0, 5, (b,), generator=g)


def _grads_by_name(model):
    return {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("name,size,batch,damp", [("flower_cnn", 64, 16, 1.0), ("resnet50", 96, 16, 0.1),
                                                   ("resnet50", 128, 32, 0.1)])
def test_forward_backward_matches_cpu_reference(name, size, batch, damp):
    """Loss and every parameter gradient of a training step against fp32 CPU autograd.

    ``damp`` scales the last batch-norm scale of every residual branch (the usual near-zero initialisation of that
    scale).  At 1.0 a random 50-layer network at these sizes is chaotic in bf16: *any* bf16 evaluation -- PyTorch's own
    GPU ops included -- returns gradients uncorrelated with the fp32 ones (relative error 1.3 over the whole gradient),
    and a comparison proves nothing.  Damped, the same comparison is sharp: whole-gradient error 0.09 .. 0.12, loss
    within 0.002 (tools/experiments/diag_model_tol.py prints the calibration)."""
    torch.manual_seed(0)
    kw = {"dropout_rate": 0.0} if name == "flower_cnn" else {"zero_init_residual": False}
    cpu = build_model(name, device="cpu", **kw).train()
    if damp != 1.0:
        for n, p in cpu.named_parameters():
            if n.endswith("conv3.gamma"):
                p.data.mul_(damp)
    # round weights to bf16 so that both paths start from identical values
    for p in cpu.parameters():
        if p.dim() > 1:
            p.data = p.data.bfloat16().float()
    gpu = cast_for_device(copy.deepcopy(cpu), "cuda").train()
    x, y = _images(batch, size)
    xb = x.bfloat16()

    def run(model, xin, yin):
        out = model(xin)
        logits, reg = out if isinstance(out, tuple) else (out, None)
        loss, _, _, correct = F.softmax_xent(logits, yin, 5)
        total = loss if reg is None else loss + reg
        total.backward()
        return loss.detach().float().cpu(), logits.detach().float().cpu()

    loss_c, logit_c = run(cpu, xb.float(), y)
    loss_g, logit_g = run(gpu, xb.cuda(), y.cuda())
    # calibration: the same network through PyTorch's own bf16 GPU ops.  bf16 activations (ReLU-mask flips,
    # batch-norm cancellation) put both bf16 paths the same distance from the fp32 oracle; the HIP path must not be
    # further away than the library path (measured: loss deviations <= 0.0015 on both, per-tensor errors within
    # 1.25 x the library's, whole-gradient error 0.95 .. 1.06 x the library's).
    gpu_ref = cast_for_device(copy.deepcopy(cpu), "cuda").train()
    with F.reference_mode():
        loss_r, _ = run(gpu_ref, xb.cuda(), y.cuda())
    lib_dev = abs(loss_c.item() - loss_r.item())
    assert abs(loss_c.item() - loss_g.item()) <= 1.5 * lib_dev + 0.005
    gc, gg, gr = _grads_by_name(cpu), _grads_by_name(gpu), _grads_by_name(gpu_ref)
    assert gc.keys() == gg.keys()
    bad = []
    for n in gc:
        a = gc[n]
        denom = a.norm().item() + 1e-6
        rel = (a - gg[n]).norm().item() / denom
        rel_lib = (a - gr[n]).norm().item() / denom
        if rel > 1.3 * rel_lib + 0.02:
            bad.append((n, rel, rel_lib))
    assert not bad, bad[:10]
    whole = lambda g: torch.cat([g[n].flatten() for n in gc])
    vc, vg, vr = whole(gc), whole(gg), whole(gr)
    err_hip, err_lib = ((vc - vg).norm() / vc.norm()).item(), ((vc - vr).norm() / vc.norm()).item()
    assert err_lib < 0.2, err_lib   # the comparison is in its meaningful regime
    assert err_hip <= 1.1 * err_lib + 0.01, (err_hip, err_lib)


@pytest.mark.parametrize("name,size,batch", [("flower_cnn", 64, 8), ("resnet50", 96, 8), ("inception_v3_head", 299, 2)])
def test_inference_forward_matches_cpu_reference(name, size, batch):
    """Eval mode: running statistics folded into the convolution epilogues (one kernel per conv+bn+relu layer)
    against the fp32 CPU definition of the same network."""
    torch.manual_seed(1)
    cpu = build_model(name, device="cpu").eval()
    g = torch.Generator().manual_seed(2)
    for n, b in cpu.named_buffers():
        if n.endswith("running_mean"):
            b.copy_(torch.randn(b.shape, generator=g) * 0.1)
        elif n.endswith("running_var"):
            b.copy_(torch.rand(b.shape, generator=g) + 0.5)
    for p in cpu.parameters():
        if p.dim() > 1:
            p.data = p.data.bfloat16().float()
    gpu = cast_for_device(copy.deepcopy(cpu), "cuda").eval()
    x, _ = _images(batch, size, seed=3)
    xb = x.bfloat16()
    with torch.no_grad():
        want = cpu(xb.float())
        got = gpu(xb.cuda())
        again = gpu(xb.cuda())  # second call: cached (scale, shift)
    want = (want[0] if isinstance(want, tuple) else want)[:, :5].float()
    got_l = (got[0] if isinstance(got, tuple) else got)[:, :5].float().cpu()
    assert torch.equal(got_l, (again[0] if isinstance(again, tuple) else again)[:, :5].float().cpu())
    scale = want.abs().max().item() + 1e-3
    assert (got_l - want).abs().max().item() <= 0.05 * scale, ((got_l - want).abs().max().item(), scale)
    # restoring other statistics invalidates the cached coefficients
    for n, b in gpu.named_buffers():
        if n.endswith("running_var"):
            b.mul_(4.0)
    with torch.no_grad():
        changed = gpu(xb.cuda())
    changed = (changed[0] if isinstance(changed, tuple) else changed)[:, :5].float().cpu()
    assert not torch.equal(changed, got_l)


def test_resnet50_loss_decreases_on_fixed_batch():
    torch.manual_seed(0)
    model = build_model("resnet50", device="cuda").train()
    eng = DataParallelEngine(model, mode="sync_dp", optimizer="momentum", lr=0.02,
                             opt_kwargs={"momentum": 0.9, "weight_decay": 1e-4})
    x, y = _images(16, 96, seed=1)
    x, y = x.cuda().bfloat16(), y.cuda().to(torch.int32)
    losses = []
    for _ in range(25):
        eng.begin_step()
        loss, _, _, _ = F.softmax_xent(model(x), y, 5)
        loss.backward()
        eng.finish_step()
        losses.append(loss.item())
    assert losses[-1] < 0.2 * losses[0], losses


def test_inception_head_trains_only_head():
    torch.manual_seed(0)
    model = build_model("inception_v3_head", device="cuda").train()
    eng = DataParallelEngine(model, mode="sync_dp", optimizer="adam", lr=1e-3)
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    assert all(n.startswith("head.") for n in names) and len(names) == 4
    x, y = _images(8, 299, seed=2)
    x, y = x.cuda().bfloat16(), y.cuda().to(torch.int32)
    trunk_before = model.trunk.Conv2d_1a_3x3.weight.detach().clone()
    losses = []
    for _ in range(8):
        eng.begin_step()
        logits, reg = model(x)
        loss, _, _, _ = F.softmax_xent(logits, y, 5, act_l2=model.head.act_l2)
        (loss + reg).backward()
        eng.finish_step()
        losses.append(loss.item())
    assert torch.equal(trunk_before, model.trunk.Conv2d_1a_3x3.weight)
    assert all(l == l for l in losses)


def test_inception_head_gradient_matches_cpu_autograd():
    """The reference trains ``fc_trainable/*`` only (recognition_flower/single/flower_train.py:141-152): on the
    features the frozen trunk produces in training mode, the head's loss (cross-entropy + both activity-L2 terms)
    and the gradients of its four variables equal fp32 CPU autograd of the same head."""
    torch.manual_seed(0)
    model = build_model("inception_v3_head", device="cuda", head_dropout_rate=0.0).train()
    x, y = _images(8, 299, seed=4)
    with torch.no_grad():
        feats = model.trunk(x.cuda().bfloat16())
    assert feats.shape == (8, 2048) and torch.isfinite(feats.float()).all()
    cpu_head = copy.deepcopy(model.head).float().cpu().train()
    logits, reg = model.head(feats)
    loss, _, _, _ = F.softmax_xent(logits, y.cuda().to(torch.int32), 5, act_l2=model.head.act_l2)
    (loss + reg).backward()
    logits_c, reg_c = cpu_head(feats.float().cpu())
    loss_c, _, _, _ = F.softmax_xent(logits_c, y, 5, act_l2=cpu_head.act_l2)
    (loss_c + reg_c).backward()
    assert abs((loss + reg).item() - (loss_c + reg_c).item()) <= 2e-2 * abs((loss_c + reg_c).item())
    for (n, pg), (_, pc) in zip(model.head.named_parameters(), cpu_head.named_parameters()):
        a, b = pg.grad.float().cpu(), pc.grad
        assert (a - b).norm().item() <= 2e-2 * b.norm().item() + 1e-6, n
    assert all(p.grad is None for p in model.trunk.parameters())


def test_inception_finetune_matches_cpu_reference_and_learns():
    """``inception_v3_finetune`` trains the trunk too: loss and the gradients of all trunk and head variables of one
    training step against fp32 CPU autograd, judged against PyTorch's own bf16 GPU ops on the same network (the
    calibration of ``test_forward_backward_matches_cpu_reference``); the variables no loss touches (AuxLogits,
    Logits) receive no gradient; a few Adam steps on a fixed batch reduce the loss."""
    torch.manual_seed(0)
    cpu = build_model("inception_v3_finetune", device="cpu", head_dropout_rate=0.0).train()
    cpu.trunk.prelogits_dropout_rate = 0.0
    for p in cpu.parameters():
        if p.dim() > 1:
            p.data = p.data.bfloat16().float()
    gpu = cast_for_device(copy.deepcopy(cpu), "cuda").train()
    gpu_ref = cast_for_device(copy.deepcopy(cpu), "cuda").train()
    x, y = _images(8, 139, seed=5)
    xb = x.bfloat16()

    def run(model, xin, yin):
        logits, reg = model(xin)
        loss, _, _, _ = F.softmax_xent(logits, yin, 5, act_l2=model.head.act_l2)
        (loss + reg).backward()
        return (loss + reg).item()

    loss_c = run(cpu, xb.float(), y)
    loss_g = run(gpu, xb.cuda(), y.cuda().to(torch.int32))
    with F.reference_mode():
        loss_r = run(gpu_ref, xb.cuda(), y.cuda().to(torch.int32))
    assert abs(loss_c - loss_g) <= 1.5 * abs(loss_c - loss_r) + 0.01 * abs(loss_c), (loss_c, loss_g, loss_r)
    gc, gg, gr = _grads_by_name(cpu), _grads_by_name(gpu), _grads_by_name(gpu_ref)
    assert gc.keys() == gg.keys()
    assert any(n.startswith("trunk.Conv2d_1a_3x3") for n in gc) and any(n.startswith("trunk.Mixed_7c") for n in gc)
    assert not any("AuxLogits" in n or "Logits_Conv2d_1c_1x1" in n for n in gg)
    whole = lambda g: torch.cat([g[n].flatten() for n in gc])
    vc, vg, vr = whole(gc), whole(gg), whole(gr)
    err_hip, err_lib = ((vc - vg).norm() / vc.norm()).item(), ((vc - vr).norm() / vc.norm()).item()
    assert err_hip <= 1.2 * err_lib + 0.02, (err_hip, err_lib)

    eng = DataParallelEngine(gpu, mode="sync_dp", optimizer="adam", lr=1e-3)
    xg, yg = xb.cuda(), y.cuda().to(torch.int32)
    before = gpu.trunk.Mixed_5b.b0.weight.detach().clone()
    losses = []
    for _ in range(12):
        eng.begin_step()
        logits, reg = gpu(xg)
        loss, _, _, _ = F.softmax_xent(logits, yg, 5, act_l2=gpu.head.act_l2)
        (loss + reg).backward()
        eng.finish_step()
        losses.append(loss.item())
    assert not torch.equal(before, gpu.trunk.Mixed_5b.b0.weight)
    assert losses[-1] < 0.5 * losses[0], losses


def test_step_graph_matches_eager():
    """A step replayed from a HIP graph is the same arithmetic as the eager step (deterministic kernels,
    no dropout in this model): parameters agree bit for bit after several steps."""
    from distributed_tensorflow_amd.parallel.engine import DataParallelEngine
    from distributed_tensorflow_amd.train.session import TrainingSession

    def run(hip_graph):
        torch.manual_seed(7)
        model = build_model("flower_cnn", num_classes=5, device="cuda", dropout_rate=0.0)
        eng = DataParallelEngine(model, mode="sync_dp", optimizer="momentum", lr=0.05,
                                 opt_kwargs={"momentum": 0.9, "weight_decay": 1e-4})
        sess = TrainingSession(model, eng, num_classes=5, hip_graph=hip_graph, graph_warmup_steps=2)
        g = torch.Generator().manual_seed(3)
        losses = []
        for i in range(6):
            x = torch.zeros(8, 64, 64, 8)
            x[..., :3] = torch.rand(8, 64, 64, 3, generator=g) * 2 - 1
            y = torch.randint(0, 5, (8,), generator=g, dtype=torch.int32)
            r = sess.run_step(x.cuda().bfloat16(), y.cuda())
            losses.append(r.loss)
        assert (sess._graph is not None) == hip_graph
        return losses, [gr.master.clone() for gr in eng.flat.groups], eng.optimizer.step_count

    l0, w0, n0 = run(False)
    l1, w1, n1 = run(True)
    assert n0 == n1 == 6
    assert l0 == l1
    for a, b in zip(w0, w1):
        assert torch.equal(a, b)


def test_inception_branch_slices_equal_concat(monkeypatch):
    """Branch outputs written straight into channel slices of the block output equal torch.cat of
    separately stored branches, bit for bit."""
    torch.manual_seed(0)
    model = build_model("inception_v3_head", device="cuda").eval()
    x, _ = _images(4, 299, seed=5)
    x = x.cuda().bfloat16()
    with torch.no_grad():
        direct = model.trunk(x)
        monkeypatch.setattr(F, "concat_plan", lambda *a, **k: False)
        joined = model.trunk(x)
    assert direct.shape == joined.shape == (4, 2048)
    assert torch.equal(direct, joined)


def test_inception_branches_on_streams_equal_sequential(monkeypatch):
    """Branches of the Inception blocks enqueued on separate HIP streams (forced here outside graph capture, and
    captured + replayed) give the bits of the one-stream forward, in batch-statistics and in moving-statistics mode."""
    torch.manual_seed(0)
    x, _ = _images(4, 299, seed=6)
    x = x.cuda().bfloat16()
    for mode in ("train", "eval"):
        model = getattr(build_model("inception_v3_head", device="cuda"), mode)()

        def last_block():  # (PreLogits sits behind a dropout that draws a new mask per call in train mode)
            return dict(model.trunk.endpoints(x, fuse_pools=True))["Mixed_7c"]

        with torch.no_grad():
            monkeypatch.setattr(F, "BRANCH_PARALLEL", "never")
            seq = last_block()
            monkeypatch.setattr(F, "BRANCH_PARALLEL", "always")
            par = last_block()
            torch.cuda.synchronize()
            assert seq.shape == (4, 8, 8, 2048) and torch.equal(seq, par)
            monkeypatch.setattr(F, "BRANCH_PARALLEL", "capture")
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    out = last_block()
                graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(seq, out)


def test_resnet_pixel_pair_stem_equals_the_8_channel_stem():
    """[B, H, W, 4] images (pixel-pair stem: a 7x4 / stride-(2,1) convolution over [B, H, W/2, 8], K = 224) give the
    logits and the parameter gradients of the same images stored with 8 channels (7x7 / stride 2, K = 392)."""
    torch.manual_seed(0)
    model = build_model("resnet50", device="cuda", zero_init_residual=False).train()
    assert model.input_channels == 4
    x8, y = _images(8, 64, seed=5)
    x8 = x8.cuda().bfloat16()
    x4 = x8[..., :4].contiguous()
    y = y.cuda().to(torch.int32)

    def run(x):
        model.zero_grad(set_to_none=True)
        for m in model.modules():
            if hasattr(m, "running_mean"):
                m.running_mean.zero_()
                m.running_var.fill_(1.0)
        logits = model(x)
        loss, _, _, _ = F.softmax_xent(logits, y, 5)
        loss.backward()
        return logits.detach().float().cpu(), {n: p.grad.detach().float().cpu() for n, p in model.named_parameters()}

    l8, g8 = run(x8)
    l4, g4 = run(x4)
    # same products in a different summation order (k-tiles of the stem differ): bf16-level agreement
    assert (l8 - l4).abs().max().item() <= 3e-2 * l8.abs().max().item() + 1e-3
    gs8, gs4 = g8["stem.weight"], g4["stem.weight"]
    assert gs4[..., 3:].abs().max().item() == 0.0           # pad channels receive no gradient
    assert (gs8[..., :3] - gs4[..., :3]).norm().item() <= 5e-2 * gs8[..., :3].norm().item()
    for n in ("blocks.0.conv1.weight", "blocks.15.conv3.weight", "fc.weight"):
        assert (g8[n] - g4[n]).norm().item() <= 8e-2 * g8[n].norm().item() + 1e-6, n


def test_inception_trunk_statistics_in_accumulators_match_partial_rows(monkeypatch):
    """Inside ``F.stat_scope`` the forward-only batch-statistics layers of few pixels keep their statistics in fp64
    accumulators (no reduction launch per layer); features and running statistics match the partial-rows path."""
    torch.manual_seed(0)
    x, _ = _images(4, 299, seed=7)
    x = x.cuda().bfloat16()
    outs, stats = [], []
    for rows in (1 << 16, 0):
        torch.manual_seed(1)
        model = build_model("inception_v3_head", device="cuda").train()
        monkeypatch.setattr(F, "STAT_ACC_MAX_ROWS", rows)
        with torch.no_grad(), F.stat_scope(x.device):
            outs.append(dict(model.trunk.endpoints(x, fuse_pools=True))["Mixed_7c"].float())
        stats.append(torch.cat([m.running_var.flatten() for m in model.trunk.modules() if hasattr(m, "running_var")]))
    a, b = outs
    # 94 layers deep in bf16: a last-bit difference in a coefficient moves a few activations by one rounding step
    assert (a - b).abs().mean() <= 2e-3 * b.abs().mean() + 1e-4
    torch.testing.assert_close(a, b, rtol=5e-2, atol=5e-2)
    torch.testing.assert_close(stats[0], stats[1], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("captured", [False, True])
def test_statistics_accumulators_with_forked_branches(monkeypatch, captured):
    """The accumulator buffer is zeroed before the branches of a block fork onto side streams, eagerly
    (DTFA_BRANCH_PARALLEL=always) and inside a HIP graph replayed several times: the features equal those of the
    partial-rows path on one stream."""
    torch.manual_seed(0)
    x, _ = _images(4, 299, seed=7)
    x = x.cuda().bfloat16()
    torch.manual_seed(1)
    model = build_model("inception_v3_head", device="cuda").train()

    def features():
        with torch.no_grad(), F.stat_scope(x.device):
            return dict(model.trunk.endpoints(x, fuse_pools=True))["Mixed_7c"].float()

    monkeypatch.setattr(F, "BRANCH_PARALLEL", "never")
    monkeypatch.setattr(F, "STAT_ACC_MAX_ROWS", 0)
    ref = features()
    monkeypatch.setattr(F, "STAT_ACC_MAX_ROWS", 1 << 16)
    monkeypatch.setattr(F, "BRANCH_PARALLEL", "always")
    if not captured:
        outs = [features() for _ in range(3)]
    else:
        features()   # plans and workspaces exist before the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                out = features()
        outs = []
        for _ in range(4):
            g.replay()
            torch.cuda.synchronize()
            outs.append(out.clone())
    for a in outs:
        assert torch.isfinite(a).all()
        assert (a - ref).abs().mean() <= 2e-3 * ref.abs().mean() + 1e-4
        torch.testing.assert_close(a, ref, rtol=5e-2, atol=5e-2)


@pytest.mark.parametrize("relu", [True, False])
def test_frozen_statistics_layer_trains_its_scale_and_offset(relu):
    """A conv + batch-norm layer in eval mode (moving statistics) inside a training graph -- fine-tuning with
    ``trunk_bn="moving"`` -- returns gradients for weight, gamma and beta that match fp32 autograd of
    ``act(gamma * (conv(x) - moving_mean) / sqrt(moving_var + eps) + beta)``."""
    from distributed_tensorflow_amd.models.layers import ConvBnAct
    from distributed_tensorflow_amd.ops import ref

    torch.manual_seed(3)
    layer = ConvBnAct(16, 32, 3, 1, "SAME", relu=relu)
    with torch.no_grad():
        layer.gamma.uniform_(0.5, 1.5)
        layer.beta.uniform_(-0.3, 0.3)
        layer.running_mean.uniform_(-0.2, 0.2)
        layer.running_var.uniform_(0.5, 2.0)
    x = torch.randn(4, 12, 12, 16)
    # fp32 reference on the bf16-rounded operands
    xr = x.bfloat16().float().requires_grad_(True)
    wr = layer.weight.detach().bfloat16().float().requires_grad_(True)
    gr = layer.gamma.detach().clone().requires_grad_(True)
    br = layer.beta.detach().clone().requires_grad_(True)
    y = ref.conv2d(xr, wr, (1, 1), (1, 1, 1, 1))
    z = ref.batch_norm_eval(y, gr, br, layer.running_mean, layer.running_var, layer.eps)
    if relu:
        z = torch.relu(z)
    up = torch.randn_like(z)
    (z * up).sum().backward()

    dev = cast_for_device(copy.deepcopy(layer), "cuda").eval()
    xd = x.cuda().bfloat16().requires_grad_(True)
    zd = dev(xd)
    (zd.float() * up.cuda()).sum().backward()
    assert (zd.float().cpu() - z).abs().max().item() <= 3e-2 * z.abs().max().item()
    for name, got, want in (("gamma", dev.gamma.grad, gr.grad), ("beta", dev.beta.grad, br.grad),
                            ("weight", dev.weight.grad, wr.grad), ("x", xd.grad, xr.grad)):
        assert got is not None, name
        err = (got.float().cpu() - want).norm().item()
        assert err <= 3e-2 * want.norm().item() + 1e-4, (name, err, want.norm().item())


def test_inception_finetune_on_moving_statistics_takes_a_step():
    """``inception_v3_finetune`` with ``trunk_bn="moving"``: the trunk's convolutions and batch-norm offsets train while
    every batch-norm normalises with its (frozen) moving statistics; one step gives finite, non-zero gradients all
    the way down to the stem and leaves the moving statistics untouched."""
    torch.manual_seed(0)
    model = build_model("inception_v3_finetune", device="cuda", trunk_bn="moving").train()
    eng = DataParallelEngine(model, mode="sync_dp", optimizer="adam", lr=1e-4)
    before = model.trunk.Conv2d_1a_3x3.running_var.clone()
    x, y = _images(2, 299, seed=3)
    eng.begin_step()
    logits, reg = model(x.cuda().bfloat16())
    loss, _, _, _ = F.softmax_xent(logits, y.cuda().to(torch.int32), 5)
    (loss + reg).backward()
    stem = model.trunk.Conv2d_1a_3x3
    gw, gb = stem.weight.main_grad.float(), stem.beta.main_grad.float()
    eng.finish_step()
    torch.cuda.synchronize()
    assert torch.isfinite(loss).item() and torch.isfinite(gw).all() and torch.isfinite(gb).all()
    assert gw.abs().max().item() > 0 and gb.abs().max().item() > 0
    assert torch.equal(model.trunk.Conv2d_1a_3x3.running_var, before)
