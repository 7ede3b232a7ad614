"""Expanding 1x1 convolution + batch-norm + residual + ReLU with the normalisation folded through the convolution
(ops.functional._ExpandBnResFn): forward value, every gradient and the running statistics against the fp32 PyTorch
definition, and against the unfolded kernels (conv -> statistics -> apply)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from distributed_tensorflow_amd.ops import conv as cv  # noqa: E402
from distributed_tensorflow_amd.ops import functional as F  # noqa: E402
from distributed_tensorflow_amd.ops import ref  # noqa: E402

BF16 = torch.bfloat16


def _inputs(B, H, C, N, seed, gamma_zero=False):
    g = torch.Generator().manual_seed(seed)
    x = torch.relu(torch.randn(B, H, H, C, generator=g) + 0.3)         # a post-ReLU activation, like conv2's output
    r = torch.randn(B, H, H, N, generator=g)
    w = torch.randn(N, 1, 1, C, generator=g) * C ** -0.5
    gamma = torch.zeros(N) if gamma_zero else torch.rand(N, generator=g) + 0.5
    beta = torch.randn(N, generator=g) * 0.2
    dz = torch.randn(B, H, H, N, generator=g)
    return x, r, w, gamma, beta, dz


def _run(x, r, w, gamma, beta, dz, device, fold, dtype):
    leaf = lambda t, dt: t.to(device, dt).detach().clone().requires_grad_(True)  # noqa: E731
    xs, rs, ws = leaf(x, dtype), leaf(r, dtype), leaf(w, dtype)
    gs, bs = leaf(gamma, torch.float32), leaf(beta, torch.float32)
    rm, rv = torch.zeros(w.shape[0], device=device), torch.ones(w.shape[0], device=device)
    old = F.EXPAND_FOLD
    F.EXPAND_FOLD = fold
    try:
        z = F.conv_bn_act(xs, ws, gs, bs, rm, rv, stride=(1, 1), padding="VALID", eps=1e-5, momentum=0.1, relu=True,
                          training=True, residual=rs)
        z.backward(dz.to(device, dtype))
    finally:
        F.EXPAND_FOLD = old
    return [t.detach().float().cpu() for t in (z, xs.grad, rs.grad, ws.grad, gs.grad, bs.grad, rm, rv)]


@pytest.mark.parametrize("B,H,C,N", [(8, 14, 64, 256), (4, 7, 128, 512), (3, 9, 72, 160)])
@pytest.mark.parametrize("gamma_zero", [False, True])
def test_folded_expand_matches_fp32_and_unfolded(B, H, C, N, gamma_zero):
    inp = _inputs(B, H, C, N, seed=C + N, gamma_zero=gamma_zero)
    # the fp32 definition, evaluated on the bf16-rounded inputs the device sees
    rounded = [t.to(BF16).float() if i in (0, 1, 2, 5) else t for i, t in enumerate(inp)]
    want = _run(*rounded, device="cpu", fold=False, dtype=torch.float32)
    got = _run(*inp, device="cuda", fold=True, dtype=BF16)
    base = _run(*inp, device="cuda", fold=False, dtype=BF16)
    names = ["z", "dx", "dres", "dw", "dgamma", "dbeta", "running_mean", "running_var"]
    # the folded kernel rounds the normalised value to bf16 before it adds the residual: where the pre-activation is
    # ~ 0 the ReLU may fall on the other side, which changes that element's residual gradient by its full size;
    # those (rare) elements are left out of the elementwise comparison of the residual gradient
    same_sign = (got[0] > 0) == (want[0] > 0)
    assert same_sign.float().mean().item() > 0.995
    for nm, w_, g_, b_ in zip(names, want, got, base):
        scale = w_.abs().max().item() + 1e-6
        keep = same_sign if nm == "dres" else torch.ones_like(w_, dtype=torch.bool)
        err_fold = ((g_ - w_).abs() * keep).max().item() / scale
        err_base = ((b_ - w_).abs() * keep).max().item() / scale
        # as close to fp32 as the unfolded kernels are (plus a little), and within bf16 resolution anyway.  With
        # gamma == 0 every pre-activation is beta + r, so sign flips are as frequent as they get (~1e-4 of the
        # elements); one flipped element moves a per-channel sum over these few rows by a few per cent
        slack = 8e-2 if (gamma_zero and nm in ("dgamma", "dbeta", "dw", "dx")) else 2e-2
        assert err_fold <= 2.0 * err_base + slack, (nm, err_fold, err_base)
    # ReLU sign disagreements between folded and unfolded are confined to pre-activations ~ 0
    z_f, z_b = got[0], base[0]
    assert ((z_f > 0) != (z_b > 0)).float().mean().item() < 5e-3


def test_folded_expand_in_a_block_uses_gated_gradient_handoff():
    """Two stacked bottlenecks: the second block's conv1 data-gradient stores the gated gradient for the first
    block's folded tail (BnBackwardTap.gate_store).  Against the fp32 CPU autograd of the same blocks the folded
    path is as accurate as the unfolded kernels (random gradients through six batch-norms at 1568 rows are noisy in
    bf16 either way: ~10 % of the largest entry, so the two device paths are compared through the reference)."""
    import copy

    from distributed_tensorflow_amd.models.layers import cast_for_device
    from distributed_tensorflow_amd.models.resnet import Bottleneck

    torch.manual_seed(0)
    blocks = [Bottleneck(256, 64, 1, "a", zero_init_residual=False), Bottleneck(256, 64, 1, "b", zero_init_residual=False)]
    cpu_blocks = copy.deepcopy(blocks)
    for b in blocks:
        cast_for_device(b, "cuda").train()
    for cb, b in zip(cpu_blocks, blocks):
        cb.train()
        for pc, pg in zip(cb.parameters(), b.parameters()):
            pc.data = pg.data.float().cpu()   # the bf16-rounded weights the device uses
    x0 = torch.randn(8, 14, 14, 256).to(BF16)
    dz = torch.randn(8, 14, 14, 256).to(BF16)

    def run(blks, device, dtype, fold):
        old = F.EXPAND_FOLD
        F.EXPAND_FOLD = fold
        try:
            for b in blks:
                b.zero_grad(set_to_none=True)
            x = x0.to(device, dtype).clone().requires_grad_(True)
            tap = cv.BnBackwardTap() if device == "cuda" else None
            h = blks[0](x, in_tap=None, out_tap=tap)
            armed = bool(tap is not None and tap.armed and tap.gate_store)
            out = blks[1](h, in_tap=tap, out_tap=None)
            out.backward(dz.to(device, dtype))
            grads = [p.grad.detach().float().cpu() for b in blks for p in b.parameters()] + [x.grad.float().cpu()]
            return grads, armed, out.detach().float().cpu()
        finally:
            F.EXPAND_FOLD = old

    g_ref, _, o_ref = run(cpu_blocks, "cpu", torch.float32, False)
    g_fold, armed, o_fold = run(blocks, "cuda", BF16, True)
    g_base, not_armed, o_base = run(blocks, "cuda", BF16, False)
    assert armed and not not_armed
    assert (o_fold - o_ref).abs().max().item() <= 1.25 * (o_base - o_ref).abs().max().item() + 1e-2 * o_ref.abs().max().item()
    for r, a, b in zip(g_ref, g_fold, g_base):
        rms = r.pow(2).mean().sqrt().item() + 1e-9
        e_fold = (a - r).pow(2).mean().sqrt().item() / rms
        e_base = (b - r).pow(2).mean().sqrt().item() / rms
        assert e_fold <= 1.25 * e_base + 1e-2, (e_fold, e_base)


@pytest.mark.parametrize("cin,width,stride,hw", [(64, 64, 1, 14), (256, 128, 2, 14), (128, 64, 2, 9)])
def test_folded_projection_block_matches_fp32_like_the_unfolded_kernels(cin, width, stride, hw):
    """A projection bottleneck (strided 1x1 shortcut with its own batch-norm): conv3 and the shortcut run as one
    folded GEMM over [h | x at the stride].  Outputs, every parameter gradient and the input gradient against fp32
    CPU autograd, with the unfolded device kernels as the yardstick for bf16 noise."""
    import copy

    from distributed_tensorflow_amd.models.layers import cast_for_device
    from distributed_tensorflow_amd.models.resnet import Bottleneck

    torch.manual_seed(cin + stride)
    blk = Bottleneck(cin, width, stride, "p", zero_init_residual=False)
    with torch.no_grad():
        for m in (blk.conv3, blk.shortcut):
            m.gamma.uniform_(0.5, 1.5)
            m.beta.normal_(0.0, 0.2)
    ref_blk = copy.deepcopy(blk)
    cast_for_device(blk, "cuda").train()
    ref_blk.train()
    for pc, pg in zip(ref_blk.parameters(), blk.parameters()):
        pc.data = pg.data.float().cpu()
    x0 = torch.relu(torch.randn(6, hw, hw, cin)).to(BF16)
    oh = (hw - 1) // stride + 1
    dz = torch.randn(6, oh, oh, width * 4).to(BF16)

    def run(b, device, dtype, fold):
        old = F.EXPAND_FOLD
        F.EXPAND_FOLD = fold
        try:
            b.zero_grad(set_to_none=True)
            x = x0.to(device, dtype).clone().requires_grad_(True)
            # the block input must itself come out of an op for the slots to engage as in the network
            out = b(x * 1.0, in_tap=None, out_tap=None)
            out.backward(dz.to(device, dtype))
            stats = [b.conv3.running_mean, b.conv3.running_var, b.shortcut.running_mean, b.shortcut.running_var]
            return ([out.detach().float().cpu()] + [p.grad.detach().float().cpu() for p in b.parameters()]
                    + [x.grad.float().cpu()] + [t.detach().float().cpu().clone() for t in stats])
        finally:
            F.EXPAND_FOLD = old

    want = run(ref_blk, "cpu", torch.float32, False)
    got = run(blk, "cuda", BF16, True)
    for m in (blk.conv3, blk.shortcut):   # running statistics: start both device runs from the same state
        m.running_mean.zero_()
        m.running_var.fill_(1.0)
    base = run(blk, "cuda", BF16, False)
    for i, (r, a, b_) in enumerate(zip(want, got, base)):
        rms = r.pow(2).mean().sqrt().item() + 1e-9
        e_fold = (a - r).pow(2).mean().sqrt().item() / rms
        e_base = (b_ - r).pow(2).mean().sqrt().item() / rms
        assert e_fold <= 1.25 * e_base + 1e-2, (i, e_fold, e_base)
