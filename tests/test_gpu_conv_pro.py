"""Pointwise GEMM with a fused batch-norm prologue (csrc/hip/conv_pro.hip) against fp32 PyTorch and against the
separate elementwise pass + convolution it replaces (the side outputs must be bit-identical to that pass)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from distributed_tensorflow_amd.ops import conv as cv  # noqa: E402
from distributed_tensorflow_amd.ops import kernels as K  # noqa: E402

BF16 = torch.bfloat16


def _rand(*shape, seed, scale=1.0, shift=0.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale + shift).to(BF16).cuda()


def _coefs(k, seed, n):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(k, generator=g) * 0.5 + (1.0 if i == 0 else 0.0)).cuda() for i in range(n)]


def _bits(t):
    """bool [..., K] -> packed uint8, bit j of byte i = element 8*i + j"""
    b = t.reshape(-1, 8).to(torch.uint8)
    w = (1 << torch.arange(8, device=t.device, dtype=torch.int32)).to(torch.uint8)
    return (b * w).sum(1).to(torch.uint8)


def _gemm_ref(a_bf16, w):
    return a_bf16.float().reshape(-1, a_bf16.shape[-1]) @ w.float().t()


CASES = [  # (rows, K, N)
    (4 * 9 * 9, 256, 64),      # partial row tile
    (2 * 14 * 14, 512, 128),
    (1000, 1024, 256),
    (3 * 7 * 7, 2048, 512),    # two column tiles of 256
    (777, 64, 256),            # one k-tile of 64, K < BK multiple handled by tail
    (513, 72, 40),             # K and N tails
]


@pytest.mark.parametrize("rows,k,n", CASES)
@pytest.mark.parametrize("mode", ["fwd", "fwd_relu", "fwd_res"])
def test_forward_prologue(rows, k, n, mode):
    y = _rand(rows, k, seed=1, scale=1.5, shift=0.2)
    r = _rand(rows, k, seed=2)
    w = _rand(n, k, seed=3, scale=k ** -0.5)
    sc, sh = _coefs(k, 4, 2)
    relu = mode != "fwd"
    res = r if mode == "fwd_res" else None
    # the pass being fused
    if relu:
        z_ref, mask_ref = K.bn_apply(y, sc, sh, True, res, want_mask=True)
    else:
        z_ref, mask_ref = K.bn_apply(y, sc, sh, False, res), None
    side = torch.full_like(y, float("nan"))
    smask = torch.zeros(rows * k // 8, dtype=torch.uint8, device="cuda") if relu else None
    pro = cv.PRO_FWD_RES if mode == "fwd_res" else cv.PRO_FWD
    out, part = cv.conv_pro(y, w, n, pro, sc, sh, src2=res, side=side, side_mask=smask, relu=relu, want_stats=True)
    torch.cuda.synchronize()
    assert torch.equal(side.view(torch.int16), z_ref.view(torch.int16))
    if relu:
        assert torch.equal(smask, mask_ref)
    ref = _gemm_ref(z_ref, w)
    err = (out.float() - ref).abs().max().item()
    assert err <= 2e-2 * ref.abs().max().item() + 1e-3, err
    # statistics epilogue: column sums / sums of squares of the fp32 accumulators
    s = part.sum(0)
    torch.testing.assert_close(s[0], ref.sum(0), rtol=2e-2, atol=2e-2 * ref.abs().sum(0).max().item())
    torch.testing.assert_close(s[1], (ref * ref).sum(0), rtol=2e-2, atol=1e-3)
    # without a side output (second consumer absent) the result is the same
    out2, _ = cv.conv_pro(y, w, n, pro, sc, sh, src2=res, relu=relu)
    assert torch.equal(out2.view(torch.int16), out.view(torch.int16))


@pytest.mark.parametrize("rows,k,n", CASES)
@pytest.mark.parametrize("mode", ["relu", "mask", "plain"])
def test_backward_prologue(rows, k, n, mode):
    dz = _rand(rows, k, seed=5)
    y = _rand(rows, k, seed=6, scale=1.5, shift=0.2)
    w = _rand(n, k, seed=7, scale=k ** -0.5)
    ca, cb, cc, gs, gt = _coefs(k, 8, 5)
    cc = cc * 0.1
    addend = _rand(rows, n, seed=9)
    gate = None
    if mode == "relu":
        keep = (y.float() * gs + gt) > 0
        pro = cv.PRO_BWD_RELU
    elif mode == "mask":
        keep = torch.rand(rows, k, device="cuda") > 0.4
        gate = _bits(keep)
        pro = cv.PRO_BWD_MASK
    else:
        keep = torch.ones(rows, k, dtype=torch.bool, device="cuda")
        pro = cv.PRO_BWD_PLAIN
    g = torch.where(keep, dz.float(), torch.zeros((), device="cuda"))
    dy_ref = (ca * g + (cb * y.float() + cc)).to(BF16)
    side = torch.full_like(dz, float("nan"))
    amask = _bits(torch.rand(rows, n, device="cuda") > 0.5)
    out, _ = cv.conv_pro(dz, w, n, pro, ca, cb, cc, src2=y, gate=gate, gscale=gs, gshift=gt, side=side,
                         addend=addend, addend_mask=amask)
    torch.cuda.synchronize()
    # fma contraction order differs from the torch expression by at most one bf16 ulp on a few elements
    d = (side.float() - dy_ref.float()).abs()
    assert (d <= 2 ** -7 * dy_ref.float().abs() + 1e-6).all()
    ref = _gemm_ref(side, w)
    bits = amask.view(-1, 1).bitwise_right_shift(torch.arange(8, device="cuda", dtype=torch.uint8)).bitwise_and(1).bool()
    ref = ref + torch.where(bits.reshape(rows, n), addend.float(), torch.zeros((), device="cuda"))
    err = (out.float() - ref).abs().max().item()
    assert err <= 2e-2 * ref.abs().max().item() + 1e-3, err


def test_backward_prologue_matches_separate_pass_and_emits_tap_sums():
    """dy of the fused prologue == bn_bwd_apply bit for bit; the data-gradient epilogue's batch-norm sums of the
    layer below equal those of the separate reduction."""
    rows, k, n = 2 * 28 * 28, 512, 128
    dz = _rand(rows, k, seed=11)
    y = _rand(rows, k, seed=12, scale=1.3)
    w = _rand(n, k, seed=13, scale=k ** -0.5)
    mean, invstd = y.float().mean(0), 1.0 / (y.float().var(0, unbiased=False) + 1e-5).sqrt()
    gamma = torch.rand(k, device="cuda") + 0.5
    keep = torch.rand(rows, k, device="cuda") > 0.3
    gate = _bits(keep)
    dy_sep, _, dgamma, dbeta = K.bn_backward(dz, y, None, mean, invstd, gamma, True, False, mask=gate)
    # the same coefficients, then the fused kernel
    coef = K.bn_backward_coefs(dz, y, mean, invstd, gamma, mask=gate)[0]
    tap = cv.BnBackwardTap()
    ylow = _rand(rows, n, seed=14)
    tap.y, tap.mean, tap.invstd = ylow, ylow.float().mean(0), torch.ones(n, device="cuda")
    tap.scale, tap.shift = torch.ones(n, device="cuda"), torch.zeros(n, device="cuda")
    tap.relu_mode = 2
    side = torch.empty_like(dz)
    out, part = cv.conv_pro(dz, w, n, cv.PRO_BWD_MASK, coef[0], coef[1], coef[2], src2=y, gate=gate, side=side, bn_tap=tap)
    torch.cuda.synchronize()
    assert torch.equal(side.view(torch.int16), dy_sep.view(torch.int16))
    g = torch.where(ylow.float() > 0, out.float(), torch.zeros((), device="cuda"))
    s = part.sum(0)
    torch.testing.assert_close(s[0], g.sum(0), rtol=1e-2, atol=1e-2 * g.abs().sum(0).max().item())
    xhat = ylow.float() - tap.mean
    torch.testing.assert_close(s[1], (g * xhat).sum(0), rtol=1e-2, atol=1e-2 * (g * xhat).abs().sum(0).max().item())
