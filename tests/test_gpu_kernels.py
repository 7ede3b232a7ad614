"""Non-convolution HIP kernels vs plain PyTorch fp32 references."""
import math

import pytest
import torch

from distributed_tensorflow_amd.ops import kernels as K
from distributed_tensorflow_amd.ops import ref

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0, shift=0.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale + shift).bfloat16()


@pytest.mark.parametrize("shape", [(4, 7, 7, 64), (2, 5, 9, 48), (37, 3, 3, 2048), (64, 28, 28, 64)])
@pytest.mark.parametrize("relu,res", [(True, False), (True, True), (False, False)])
def test_bn_forward_backward(shape, relu, res):
    C = shape[-1]
    y = _rand(*shape, seed=1, scale=1.5, shift=0.3)
    r = _rand(*shape, seed=2) if res else None
    gen = torch.Generator().manual_seed(C)
    gamma = torch.rand(C, generator=gen) + 0.5
    beta = torch.randn(C, generator=gen) * 0.1
    eps, mom = 1e-3, 0.1
    rm, rv = torch.zeros(C), torch.ones(C)
    # reference
    yf = y.float().requires_grad_(True)
    gf, bf = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rf = r.float().requires_grad_(True) if res else None
    rm_ref, rv_ref = rm.clone(), rv.clone()
    out, mean, invstd = ref.batch_norm_train(yf, gf, bf, rm_ref, rv_ref, eps, mom)
    if res:
        out = out + rf
    pre = out.detach().clone()
    if relu:
        out = torch.relu(out)
    dz = _rand(*shape, seed=3)
    out.backward(dz.float())
    # kernels
    yc = y.cuda()
    part = K.bn_stats_partials(yc)
    rmc, rvc = rm.cuda(), rv.cuda()
    m, i, sc, sh = K.bn_finalize(part, yc.numel() // C, eps, gamma.cuda(), beta.cuda(), rmc, rvc, mom)
    z = K.bn_apply(yc, sc, sh, relu, r.cuda() if res else None)
    torch.testing.assert_close(m.cpu(), mean.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(i.cpu(), invstd.detach(), rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(rmc.cpu(), rm_ref, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rvc.cpu(), rv_ref, rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(z.float().cpu(), out.detach(), rtol=2e-2, atol=2e-2)
    # without a residual the ReLU mask is recomputed from y (zout not read)
    dy, dres, dgamma, dbeta = K.bn_backward(dz.cuda(), yc, z if res else None, m, i, gamma.cuda(), relu, res,
                                            scale=sc, shift=sh)
    # the reference masks with its own fp32 output; the mask from the bf16 data can differ only where the
    # pre-activation is ~ 0 (there a flipped sign changes the element's gradient by its full size): those
    # elements are left out of the comparison
    sure = (pre.abs() > 3e-2) if relu else torch.ones_like(pre, dtype=torch.bool)
    assert sure.float().mean().item() > 0.95
    tol = 3e-2 * (yf.grad.abs().max().item() + 1e-6)
    assert ((dy.float().cpu() - yf.grad).abs() * sure).max().item() <= tol
    torch.testing.assert_close(dgamma.cpu(), gf.grad, rtol=2e-2, atol=2e-2 * gf.grad.abs().max().item())
    torch.testing.assert_close(dbeta.cpu(), bf.grad, rtol=2e-2, atol=2e-2 * bf.grad.abs().max().item())
    if res:
        assert ((dres.float().cpu() - rf.grad).abs() * sure).max().item() <= 2e-2 * rf.grad.abs().max().item()
    if relu:
        # the stored sign bitmask is an exact substitute for reading the activation output
        z2, mask = K.bn_apply(yc, sc, sh, True, r.cuda() if res else None, want_mask=True)
        assert torch.equal(z2, z) and mask.dtype == torch.uint8 and mask.numel() * 8 == z.numel()
        bits = ((mask.view(-1, 1) >> torch.arange(8, device="cuda", dtype=torch.uint8)) & 1).bool().view(z.shape)
        assert torch.equal(bits, z > 0)
        dy_z, dres_z, dg_z, db_z = K.bn_backward(dz.cuda(), yc, z, m, i, gamma.cuda(), True, True)
        dy_m, _, dg_m, db_m = K.bn_backward(dz.cuda(), yc, None, m, i, gamma.cuda(), True, False, mask=mask)
        assert torch.equal(dy_z, dy_m) and torch.equal(dg_z, dg_m) and torch.equal(db_z, db_m)
        assert torch.equal(K.mask_mul(dz.cuda(), mask), dres_z)


def test_bn_eval_coeffs():
    C = 96
    gamma, beta = torch.rand(C) + 0.5, torch.randn(C)
    rm, rv = torch.randn(C), torch.rand(C) + 0.2
    sc, sh = K.bn_eval_coeffs(gamma.cuda(), beta.cuda(), rm.cuda(), rv.cuda(), 1e-3)
    y = _rand(3, 4, 4, C, seed=4)
    want = ref.batch_norm_eval(y.float(), gamma, beta, rm, rv, 1e-3)
    got = K.bn_apply(y.cuda(), sc, sh, False)
    torch.testing.assert_close(got.float().cpu(), want, rtol=2e-2, atol=2e-2)


POOLS = [
    ((2, 15, 15, 16), (3, 3), (2, 2), (0, 0, 0, 0)),
    ((2, 14, 14, 64), (3, 3), (2, 2), (1, 1, 1, 1)),
    ((3, 9, 7, 24), (3, 3), (1, 1), (1, 1, 1, 1)),
    ((1, 8, 8, 32), (2, 2), (2, 2), (0, 0, 0, 0)),
]


@pytest.mark.parametrize("shape,k,s,pads", POOLS)
def test_maxpool(shape, k, s, pads):
    x = _rand(*shape, seed=5)
    xf = x.float().requires_grad_(True)
    want = ref.max_pool(xf, k, s, pads)
    dy = _rand(*want.shape, seed=6)
    want.backward(dy.float())
    y, idx = K.maxpool_fwd(x.cuda(), k, s, pads)
    torch.testing.assert_close(y.float().cpu(), want.detach(), rtol=0, atol=0)
    dx = K.maxpool_bwd(dy.cuda(), idx, shape, k, s, pads)
    torch.testing.assert_close(dx.float().cpu(), xf.grad, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("shape,k,s,pads", POOLS)
@pytest.mark.parametrize("include_pad", [False, True])
def test_avgpool(shape, k, s, pads, include_pad):
    x = _rand(*shape, seed=7)
    xf = x.float().requires_grad_(True)
    want = ref.avg_pool(xf, k, s, pads, include_pad)
    dy = _rand(*want.shape, seed=8)
    want.backward(dy.float())
    y = K.avgpool_fwd(x.cuda(), k, s, pads, include_pad)
    torch.testing.assert_close(y.float().cpu(), want.detach(), rtol=1e-2, atol=1e-2)
    dx = K.avgpool_bwd(dy.cuda(), shape, k, s, pads, include_pad)
    torch.testing.assert_close(dx.float().cpu(), xf.grad, rtol=1e-2, atol=1e-2)


def test_global_avg_pool():
    x = _rand(5, 8, 8, 2048, seed=9)
    y = K.gap_fwd(x.cuda())
    torch.testing.assert_close(y.float().cpu(), x.float().mean((1, 2)), rtol=1e-2, atol=1e-2)
    dy = _rand(5, 2048, seed=10)
    dx = K.gap_bwd(dy.cuda(), x.shape)
    want = (dy.float() / 64)[:, None, None, :].expand(5, 8, 8, 2048)
    torch.testing.assert_close(dx.float().cpu(), want, rtol=1e-2, atol=1e-3)
    # channel-group counts that are not powers of two, pixel counts that leave a tail behind the 4-deep unroll
    for shape in ((3, 14, 14, 512), (2, 7, 7, 192), (4, 5, 3, 72), (1, 1, 1, 8), (2, 9, 9, 4096)):
        x = _rand(*shape, seed=11)
        torch.testing.assert_close(K.gap_fwd(x.cuda()).float().cpu(), x.float().mean((1, 2)), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("B,n,ld", [(32, 5, 8), (7, 1001, 1008), (64, 64, 64)])
def test_softmax_xent(B, n, ld):
    logits = _rand(B, ld, seed=11, scale=3.0)
    labels = torch.randint(0, n, (B,), dtype=torch.int32)
    lf = logits.float().requires_grad_(True)
    loss, top1, score, correct = ref.softmax_xent(lf, labels, n, act_l2=5e-4)
    loss.mean().backward()
    l, dl, t1, sc, co = K.softmax_xent(logits.cuda(), labels.cuda(), n, 1.0 / B, act_l2=5e-4)
    torch.testing.assert_close(l.cpu(), loss.detach(), rtol=1e-3, atol=1e-3)
    assert torch.equal(t1.cpu(), top1)
    assert torch.equal(co.cpu(), correct)
    torch.testing.assert_close(sc.cpu(), score.detach(), rtol=0, atol=0)
    torch.testing.assert_close(dl.float().cpu(), lf.grad, rtol=2e-2, atol=1e-4)
    m = K.mean_f32(l)
    torch.testing.assert_close(m.cpu()[0], loss.mean().detach(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("shape", [(32, 1024), (5, 8), (1024, 1024), (3, 7, 64)])
def test_activity_l2_forward_backward(shape):
    """Fused regulariser of the head's hidden layer: coef/2 * sum(h^2) and dh = dh_in + coef * g * h."""
    from distributed_tensorflow_amd.ops import functional as F

    h = _rand(*shape, seed=21)
    dh_in = _rand(*shape, seed=22)
    coef = 5e-4
    hf = h.float().requires_grad_(True)
    reg_ref = 0.5 * coef * (hf ** 2).sum()
    (0.7 * reg_ref + (hf * dh_in.float()).sum()).backward()
    hg = h.cuda().requires_grad_(True)
    out, reg = F.activity_l2(hg, coef)
    assert out.data_ptr() == hg.data_ptr()
    torch.testing.assert_close(reg.cpu(), reg_ref.detach(), rtol=1e-4, atol=1e-6)
    (0.7 * reg + (out.float() * dh_in.cuda().float()).sum()).backward()
    torch.testing.assert_close(hg.grad.float().cpu(), hf.grad, rtol=1e-2, atol=1e-3)
    # regulariser only (no gradient from above)
    hg2 = h.cuda().requires_grad_(True)
    F.activity_l2(hg2, coef)[1].backward()
    torch.testing.assert_close(hg2.grad.float().cpu(), coef * h.float(), rtol=1e-2, atol=1e-6)


def test_softmax_xent_activity_term_is_a_batch_sum():
    from distributed_tensorflow_amd.ops import functional as F

    B, n = 32, 5
    logits = _rand(B, 8, seed=5, scale=2.0)
    labels = torch.randint(0, n, (B,), dtype=torch.int32)
    lf = logits.float()[:, :n]
    want = torch.nn.functional.cross_entropy(lf, labels.long()) + 5e-4 * 0.5 * (lf ** 2).sum()
    lg = logits.cuda().requires_grad_(True)
    loss, _, _, _ = F.softmax_xent(lg, labels.cuda(), n, act_l2=5e-4)
    torch.testing.assert_close(loss.cpu(), want, rtol=1e-3, atol=1e-4)
    loss.backward()
    lfr = logits.float().requires_grad_(True)
    (torch.nn.functional.cross_entropy(lfr[:, :n], labels.long()) + 5e-4 * 0.5 * (lfr[:, :n] ** 2).sum()).backward()
    torch.testing.assert_close(lg.grad.float().cpu(), lfr.grad, rtol=2e-2, atol=2e-4)


@pytest.mark.parametrize("gdtype", [torch.float32, torch.bfloat16])
def test_adam_matches_tf1_formula(gdtype):
    n = 100003
    g = torch.Generator().manual_seed(12)
    w = torch.randn(n, generator=g)
    m = torch.zeros(n)
    v = torch.zeros(n)
    wc, mc, vc = w.cuda(), m.cuda(), v.cuda()
    w16 = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-8
    for t in range(1, 4):
        grad = torch.randn(n, generator=g).to(gdtype)
        gf = grad.float()
        lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        m = b1 * m + (1 - b1) * gf
        v = b2 * v + (1 - b2) * gf * gf
        w = w - lr_t * m / (v.sqrt() + eps)
        hyper = torch.tensor([lr_t, 1.0], dtype=torch.float32, device="cuda")
        K.adam_step(wc, mc, vc, grad.cuda(), w16, hyper, b1, b2, eps)
    torch.testing.assert_close(wc.cpu(), w, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(w16.float().cpu(), wc.cpu().bfloat16().float(), rtol=0, atol=0)


def test_momentum_sgd():
    n = 4099
    g = torch.Generator().manual_seed(13)
    w = torch.randn(n, generator=g)
    p = torch.nn.Parameter(w.clone())
    opt = torch.optim.SGD([p], lr=0.1, momentum=0.9, weight_decay=1e-4)
    wc, buf = w.cuda(), torch.zeros(n, device="cuda")
    hyper = torch.tensor([0.1, 1.0], dtype=torch.float32, device="cuda")
    for _ in range(3):
        grad = torch.randn(n, generator=g)
        p.grad = grad.clone()
        opt.step()
        K.momentum_step(wc, buf, grad.cuda(), None, hyper, 0.9, 1e-4, False)
    torch.testing.assert_close(wc.cpu(), p.detach(), rtol=1e-5, atol=1e-6)


def test_dropout_deterministic_and_unbiased():
    x = torch.ones(1 << 16, dtype=torch.bfloat16, device="cuda")
    a = K.dropout(x, 0.5, seed=1234)
    b = K.dropout(x, 0.5, seed=1234)
    c = K.dropout(x, 0.5, seed=99)
    assert torch.equal(a, b) and not torch.equal(a, c)
    kept = (a != 0).float().mean().item()
    assert abs(kept - 0.5) < 0.02
    assert abs(a.float().mean().item() - 1.0) < 0.04
    d = K.dropout(x, 0.8, seed=5)
    assert abs((d != 0).float().mean().item() - 0.2) < 0.02


@pytest.mark.parametrize("shape,k,stride,pads,relu", [
    ((2, 17, 19, 24), (3, 3), (2, 2), (1, 1, 1, 1), True),
    ((3, 16, 16, 64), (3, 3), (2, 2), (0, 1, 0, 1), True),
    ((2, 9, 11, 8), (2, 2), (2, 2), (0, 0, 0, 0), False),
    ((2, 12, 10, 16), (3, 3), (2, 2), (0, 1, 0, 1), True),    # TF SAME on even sizes: padding at the end only
    ((2, 13, 15, 16), (3, 3), (2, 2), (1, 1, 1, 1), False),   # odd sizes: the last 2 x 2 cell is partial
    ((1, 8, 8, 8), (2, 2), (2, 2), (1, 1, 1, 1), True),       # 2 x 2 windows that start in the padding
    ((2, 10, 9, 8), (3, 2), (2, 2), (1, 1, 0, 1), True),      # rectangular window
    ((2, 9, 9, 8), (3, 3), (1, 1), (1, 1, 1, 1), True),       # stride 1: per-pixel search path
])
def test_bn_relu_maxpool_fused_matches_unfused(shape, k, stride, pads, relu):
    """The fused batch-norm + ReLU + max-pool pair (forward and backward) is bit-identical to the separate
    batch-norm and pooling kernels."""
    C = shape[-1]
    y = _rand(*shape, seed=21).cuda()
    gamma, beta = (torch.rand(C) + 0.5).cuda(), (torch.randn(C) * 0.3).cuda()
    part = K.bn_stats_partials(y)
    mean, invstd, sc, sh = K.bn_finalize(part, y.numel() // C, 1e-5, gamma, beta, None, None, 0.1)
    z = K.bn_apply(y, sc, sh, relu)
    p_ref, idx_ref = K.maxpool_fwd(z, k, stride, pads)
    p_fused, idx = K.bn_relu_maxpool_fwd(y, sc, sh, relu, k, stride, pads)
    assert torch.equal(p_ref, p_fused) and torch.equal(idx_ref, idx)
    dp = _rand(*p_ref.shape, seed=22).cuda()
    dz = K.maxpool_bwd(dp, idx_ref, tuple(z.shape), k, stride, pads)
    dy_ref, _, dg_ref, db_ref = K.bn_backward(dz, y, None, mean, invstd, gamma, relu, False, scale=sc, shift=sh)
    dy, dg, db = K.bn_pool_backward(dp, idx, y, mean, invstd, gamma, sc, sh, relu, k, stride, pads)
    # dz is rounded to bf16 in the unfused path before it enters the sums; the fused path keeps fp32
    torch.testing.assert_close(dg, dg_ref, rtol=2e-2, atol=2e-2 * dg_ref.abs().max().item())
    torch.testing.assert_close(db, db_ref, rtol=2e-2, atol=2e-2 * db_ref.abs().max().item())
    assert (dy.float() - dy_ref.float()).abs().max().item() <= 3e-2 * dy_ref.float().abs().max().item()
    # the winners' values kept by the forward pass: the pooled output is their image under the affine map + ReLU,
    # and the backward sums read them instead of gathering from y -- same results, bit for bit
    p_w, idx_w, ywin = K.bn_relu_maxpool_fwd(y, sc, sh, relu, k, stride, pads, want_winners=True)
    assert torch.equal(p_w, p_fused) and torch.equal(idx_w, idx)
    zw = ywin.float() * sc + sh
    zw = (torch.relu(zw) if relu else zw).bfloat16()
    assert torch.equal(zw, p_fused)
    dy_w, dg_w, db_w = K.bn_pool_backward(dp, idx, y, mean, invstd, gamma, sc, sh, relu, k, stride, pads, ywin=ywin)
    assert torch.equal(dy_w, dy)
    torch.testing.assert_close(dg_w, dg, rtol=1e-5, atol=1e-5 * dg.abs().max().item())
    torch.testing.assert_close(db_w, db, rtol=1e-5, atol=1e-5 * db.abs().max().item())


@pytest.mark.parametrize("shape,relu,sliced", [((32, 17, 17, 192), True, True), ((32, 8, 8, 448), True, False),
                                               ((5, 7, 9, 48), False, True), ((2, 3, 3, 2048), True, False)])
def test_bn_apply_from_column_sums_matches_finalize_then_apply(shape, relu, sliced):
    """One launch (coefficients from fp64 column sums, running statistics, elementwise pass) against the
    partial rows -> finalize -> apply sequence."""
    g = torch.Generator().manual_seed(21)
    y = (torch.randn(*shape, generator=g) * 1.5 + 0.3).bfloat16().cuda()
    Cc = shape[-1]
    M = y.numel() // Cc
    gamma = (torch.rand(Cc, generator=g) + 0.5).cuda()
    beta = (torch.randn(Cc, generator=g) * 0.2).cuda()
    rm0, rv0 = torch.randn(Cc, generator=g).cuda() * 0.1, (torch.rand(Cc, generator=g) + 0.5).cuda()
    rm1, rv1 = rm0.clone(), rv0.clone()
    part = K.bn_stats_partials(y)
    _, _, scale, shift = K.bn_finalize(part, M, 1e-3, gamma, beta, rm0, rv0, 0.1)
    yf = y.double().reshape(-1, Cc)
    sums = torch.cat((yf.sum(0), (yf * yf).sum(0))).contiguous()

    def target():
        if not sliced:
            return None, None
        buf = torch.full(shape[:-1] + (Cc + 24,), 7.0, dtype=torch.bfloat16, device="cuda")
        return buf, buf[..., 8:8 + Cc]

    b0, o0 = target()
    b1, o1 = target()
    want = K.bn_apply(y, scale, shift, relu, None, out=o0)
    got = K.bn_apply_sums(y, sums, M, 1e-3, gamma, beta, rm1, rv1, 0.1, relu, out=o1)
    torch.testing.assert_close(got.float(), want.float(), rtol=2e-2, atol=2e-2)
    assert (got.float() - want.float()).abs().mean() < 1e-3
    torch.testing.assert_close(rm1, rm0, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rv1, rv0, rtol=1e-4, atol=1e-6)
    if sliced:
        assert torch.equal(b1[..., :8], b0[..., :8]) and torch.equal(b1[..., 8 + Cc:], b0[..., 8 + Cc:])


@pytest.mark.parametrize("shape,stride", [((3, 8, 10, 64), (2, 2)), ((2, 7, 9, 256), (2, 2)), ((2, 9, 6, 40), (1, 2)),
                                          ((1, 5, 5, 8), (3, 2))])
def test_subsample_is_the_strided_slice(shape, stride):
    """``K.subsample`` (the block input on the grid of a strided projection shortcut) against the slice it replaces."""
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(3)).bfloat16().cuda()
    got = K.subsample(x, stride)
    want = x[:, ::stride[0], ::stride[1], :].contiguous()
    assert got.shape == want.shape and torch.equal(got, want)

