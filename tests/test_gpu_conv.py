"""HIP implicit-GEMM convolution kernels vs a plain PyTorch fp32 reference."""
import pytest
import torch

from distributed_tensorflow_amd.ops import conv as cv
from distributed_tensorflow_amd.ops import ref

pytestmark = pytest.mark.gpu

# (B, H, W, Cin, Cout, kh, kw, sy, sx, padding)
CASES = [
    (2, 8, 8, 8, 8, 1, 1, 1, 1, "VALID"),
    (2, 9, 11, 16, 24, 3, 3, 1, 1, "SAME"),
    (3, 15, 15, 8, 32, 3, 3, 2, 2, "VALID"),       # inception stem style
    (2, 17, 17, 64, 64, 3, 3, 1, 1, (1, 1, 1, 1)),  # resnet 3x3
    (2, 16, 16, 64, 128, 3, 3, 2, 2, (1, 1, 1, 1)),  # resnet stride-2 3x3
    (2, 14, 14, 128, 256, 1, 1, 2, 2, "VALID"),     # resnet downsample 1x1 s2
    (2, 23, 23, 8, 64, 7, 7, 2, 2, (3, 3, 3, 3)),   # resnet stem (Cin padded to 8)
    (2, 17, 17, 128, 160, 1, 7, 1, 1, "SAME"),      # inception 1x7
    (2, 17, 17, 160, 192, 7, 1, 1, 1, "SAME"),      # inception 7x1
    (1, 35, 35, 48, 64, 5, 5, 1, 1, "SAME"),        # inception 5x5
    (4, 8, 8, 448, 384, 3, 3, 1, 1, "SAME"),
    (5, 12, 12, 32, 40, 3, 3, 2, 2, "SAME"),        # odd Cout tile tail, SAME stride 2
    (3, 10, 12, 32, 128, 3, 3, 1, 1, "SAME"),       # one 32-channel chunk per tap, row tiles span images
    (2, 28, 28, 128, 128, 3, 3, 1, 1, "SAME"),      # resnet 3x3, four chunks per tap
    (1, 57, 57, 64, 72, 3, 3, 1, 1, "SAME"),        # wide rows, Cout tail
    (2, 20, 20, 96, 64, 3, 1, 1, 1, "SAME"),        # 3 taps, narrow output tile
    (2, 14, 14, 256, 256, 3, 3, 1, 1, "SAME"),      # weight gradient on the 256 x 128 tile
    (3, 9, 9, 64, 512, 1, 1, 1, 1, "VALID"),        # 256 x 128 tile with a half-filled K tile
    (2, 12, 12, 256, 64, 1, 1, 1, 1, "VALID"),      # weight gradient on the 64 x 256 tile
]


def _mk(case, seed=0):
    B, H, W, Cin, Cout, kh, kw, sy, sx, padding = case
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(Cout, kh, kw, Cin, generator=g) / (kh * kw * Cin) ** 0.5
    pads = cv.resolve_pads(padding, H, W, kh, kw, sy, sx)
    return x, w, pads


@pytest.mark.parametrize("case", CASES)
def test_conv_fwd(case):
    B, H, W, Cin, Cout, kh, kw, sy, sx, _ = case
    x, w, pads = _mk(case)
    xb, wb = x.bfloat16(), w.bfloat16()
    want = ref.conv2d(xb.float(), wb.float(), (sy, sx), pads)
    plan = cv.get_plan(H, W, Cin, Cout, kh, kw, sy, sx, pads, "cuda")
    y, part = cv.conv_fwd(xb.cuda(), wb.cuda(), plan, want_stats=True)
    torch.cuda.synchronize()
    got = y.float().cpu()
    assert got.shape == want.shape
    torch.testing.assert_close(got, want, rtol=2e-2, atol=2e-2)
    # fused batch-norm partial statistics are those of the fp32 accumulators (before rounding to bf16)
    s = part.sum(0).cpu()
    flat = want.reshape(-1, Cout)
    torch.testing.assert_close(s[0], flat.sum(0), rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(s[1], (flat * flat).sum(0), rtol=1e-3, atol=1e-2)


@pytest.mark.parametrize("case", CASES)
def test_conv_bias_relu(case):
    B, H, W, Cin, Cout, kh, kw, sy, sx, _ = case
    x, w, pads = _mk(case, 1)
    xb, wb = x.bfloat16(), w.bfloat16()
    bias = torch.randn(Cout)
    want = ref.conv2d(xb.float(), wb.float(), (sy, sx), pads, bias=bias, relu=True)
    plan = cv.get_plan(H, W, Cin, Cout, kh, kw, sy, sx, pads, "cuda")
    y, _ = cv.conv_fwd(xb.cuda(), wb.cuda(), plan, bias=bias.cuda(), relu=True)
    torch.testing.assert_close(y.float().cpu(), want, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("case", CASES)
def test_conv_folded_batchnorm_into_channel_slice(case):
    """Inference epilogue: relu(conv * scale + shift) on the fp32 accumulators, written into a channel slice of a
    wider NHWC buffer whose other channels stay untouched."""
    B, H, W, Cin, Cout, kh, kw, sy, sx, _ = case
    x, w, pads = _mk(case, 4)
    xb, wb = x.bfloat16(), w.bfloat16()
    g = torch.Generator().manual_seed(5)
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    want = torch.relu(ref.conv2d(xb.float(), wb.float(), (sy, sx), pads) * scale + shift)
    plan = cv.get_plan(H, W, Cin, Cout, kh, kw, sy, sx, pads, "cuda")
    y, part = cv.conv_fwd(xb.cuda(), wb.cuda(), plan, bias=shift.cuda(), relu=True, scale=scale.cuda())
    assert part is None
    torch.testing.assert_close(y.float().cpu(), want, rtol=2e-2, atol=2e-2)
    buf = torch.full((B, plan.oh, plan.ow, 16 + Cout + 24), 7.0, dtype=torch.bfloat16, device="cuda")
    out, _ = cv.conv_fwd(xb.cuda(), wb.cuda(), plan, bias=shift.cuda(), relu=True, scale=scale.cuda(),
                         out=buf[..., 16:16 + Cout])
    torch.cuda.synchronize()
    assert out.data_ptr() == buf[..., 16:].data_ptr()
    assert torch.equal(buf[..., 16:16 + Cout], y)
    assert (buf[..., :16] == 7).all() and (buf[..., 16 + Cout:] == 7).all()
    with pytest.raises(ValueError):
        cv.conv_fwd(xb.cuda(), wb.cuda(), plan, bias=shift.cuda(), out=buf[..., 4:4 + Cout])  # not 16-byte aligned
    # tail of a residual block: relu(conv * scale + shift + residual), with and without the activation
    res = torch.randn(want.shape, generator=g).bfloat16()
    pre = ref.conv2d(xb.float(), wb.float(), (sy, sx), pads) * scale + shift + res.float()
    for relu in (True, False):
        z, _ = cv.conv_fwd(xb.cuda(), wb.cuda(), plan, bias=shift.cuda(), relu=relu, scale=scale.cuda(),
                           residual=res.cuda())
        torch.testing.assert_close(z.float().cpu(), torch.relu(pre) if relu else pre, rtol=2e-2, atol=3e-2)


@pytest.mark.parametrize("case", CASES)
def test_conv_backward(case):
    B, H, W, Cin, Cout, kh, kw, sy, sx, _ = case
    x, w, pads = _mk(case, 2)
    xb = x.bfloat16().float().requires_grad_(True)
    wb = w.bfloat16().float().requires_grad_(True)
    y = ref.conv2d(xb, wb, (sy, sx), pads)
    g = torch.Generator().manual_seed(3)
    dy = torch.randn(y.shape, generator=g).bfloat16()
    y.backward(dy.float())
    plan = cv.get_plan(H, W, Cin, Cout, kh, kw, sy, sx, pads, "cuda")
    dx = cv.conv_dgrad(dy.cuda(), wb.detach().bfloat16().cuda(), plan)
    dw = cv.conv_wgrad(xb.detach().bfloat16().cuda(), dy.cuda(), plan)
    torch.cuda.synchronize()
    scale_x = xb.grad.abs().max().item() + 1e-6
    scale_w = wb.grad.abs().max().item() + 1e-6
    assert (dx.float().cpu() - xb.grad).abs().max().item() <= 2e-2 * scale_x
    assert (dw.cpu() - wb.grad).abs().max().item() <= 2e-2 * scale_w


def test_conv_large_tiles():
    # exercises the 128x128 / 256x128 tiles, BK=64 and multi-split wgrad
    case = (16, 28, 28, 128, 128, 3, 3, 1, 1, "SAME")
    B, H, W, Cin, Cout, kh, kw, sy, sx, _ = case
    x, w, pads = _mk(case, 5)
    xb, wb = x.bfloat16(), w.bfloat16()
    want = ref.conv2d(xb.float(), wb.float(), (sy, sx), pads)
    plan = cv.get_plan(H, W, Cin, Cout, kh, kw, sy, sx, pads, "cuda")
    y, _ = cv.conv_fwd(xb.cuda(), wb.cuda(), plan)
    torch.testing.assert_close(y.float().cpu(), want, rtol=2e-2, atol=2e-2)
    g = torch.Generator().manual_seed(6)
    dy = torch.randn(want.shape, generator=g).bfloat16()
    xr = xb.float().requires_grad_(True)
    wr = wb.float().requires_grad_(True)
    ref.conv2d(xr, wr, (sy, sx), pads).backward(dy.float())
    dx = cv.conv_dgrad(dy.cuda(), wb.cuda(), plan)
    dw = cv.conv_wgrad(xb.cuda(), dy.cuda(), plan)
    assert (dx.float().cpu() - xr.grad).abs().max().item() <= 2e-2 * xr.grad.abs().max().item()
    assert (dw.cpu() - wr.grad).abs().max().item() <= 2e-2 * wr.grad.abs().max().item()


@pytest.mark.parametrize("case", [(2, 56, 56), (3, 20, 17), (1, 16, 180), (5, 7, 16), (30, 56, 56)])
def test_wgrad_of_same_size_3x3_by_the_persistent_kernel(case, monkeypatch):
    """conv_wp3_kernel (3x3, stride 1, 64 -> 64: every resident workgroup holds the whole 64 x 576 gradient in registers,
    walks a range of 128-pixel tiles and stages each tile's dY rows and X patch once; taps read X at row offsets through
    transposing LDS reads, out-of-image taps from a row of zeros) against fp32 autograd and against the ordinary kernel;
    written into a new tensor, into a bf16 buffer and accumulated into an fp32 one.  The last case has more tiles than
    resident workgroups."""
    B, H, W = case
    full = (B, H, W, 64, 64, 3, 3, 1, 1, "SAME")
    x, w, pads = _mk(full, 41)
    xr = x.bfloat16().float().requires_grad_(True)
    wr = w.bfloat16().float().requires_grad_(True)
    y = ref.conv2d(xr, wr, (1, 1), pads)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(42)).bfloat16()
    y.backward(dy.float())
    plan = cv.get_plan(H, W, 64, 64, 3, 3, 1, 1, pads, "cuda")
    xb, dyb = xr.detach().bfloat16().cuda(), dy.cuda()
    monkeypatch.setattr(cv.tune, "MODE", "off")
    monkeypatch.setattr(cv, "_WGRAD_P3", False)
    want = cv.conv_wgrad(xb, dyb, plan)
    monkeypatch.setattr(cv, "_WGRAD_P3", True)
    monkeypatch.setattr(cv, "_WGRAD_P3_MIN_ROWS", 0)
    assert cv.wgrad_p3_takes(plan, B * H * W)
    got = cv.conv_wgrad(xb, dyb, plan)
    scale = wr.grad.abs().max().item()
    assert (got.cpu() - wr.grad).abs().max().item() <= 2e-2 * scale
    assert (got - want).abs().max().item() <= 1e-4 * scale          # both sum bf16 products in fp32, in different orders
    assert torch.equal(got, cv.conv_wgrad(xb, dyb, plan))            # reproducible
    acc = torch.ones_like(got)
    cv.conv_wgrad(xb, dyb, plan, out=acc, accumulate=True)
    torch.testing.assert_close(acc, got + 1.0, rtol=0, atol=1e-5 * scale + 1e-6)
    b16 = torch.empty(got.shape, dtype=torch.bfloat16, device="cuda")
    cv.conv_wgrad(xb, dyb, plan, out=b16)
    assert torch.equal(b16, got.bfloat16())


@pytest.mark.parametrize("case", [(2, 64, 32), (3, 224, 112), (1, 40, 150), (2, 12, 33)])
def test_wgrad_of_the_pixel_pair_stem_by_the_persistent_kernel(case, monkeypatch):
    """conv_wp8_kernel (8-channel pixels, 7 x 4 taps, stride (2, 1), 64 output channels: every resident workgroup holds
    the whole 64 x 224 gradient in registers and stages each tile's dY rows and contiguous input range once; both
    operands through transposing LDS reads, out-of-image taps from a pixel of zeros) against fp32 autograd and against
    the ordinary kernel.  The second case is the stem's own size with many tiles per workgroup, the third has rows wider
    than a tile, the last a tile that ends past the tensor."""
    B, H, W = case
    full = (B, H, W, 8, 64, 7, 4, 2, 1, (3, 3, 2, 1))
    x, w, pads = _mk(full, 51)
    xr = x.bfloat16().float().requires_grad_(True)
    wr = w.bfloat16().float().requires_grad_(True)
    y = ref.conv2d(xr, wr, (2, 1), pads)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(52)).bfloat16()
    y.backward(dy.float())
    plan = cv.get_plan(H, W, 8, 64, 7, 4, 2, 1, pads, "cuda")
    xb, dyb = xr.detach().bfloat16().cuda(), dy.cuda()
    monkeypatch.setattr(cv.tune, "MODE", "off")
    monkeypatch.setattr(cv, "_WGRAD_P8", False)
    want = cv.conv_wgrad(xb, dyb, plan)
    monkeypatch.setattr(cv, "_WGRAD_P8", True)
    monkeypatch.setattr(cv, "_WGRAD_P3_MIN_ROWS", 0)
    assert cv.wgrad_p8_takes(plan, B * plan.oh * plan.ow)
    got = cv.conv_wgrad(xb, dyb, plan)
    scale = wr.grad.abs().max().item()
    assert (got.cpu() - wr.grad).abs().max().item() <= 2e-2 * scale
    assert (got - want).abs().max().item() <= 1e-4 * scale          # both sum bf16 products in fp32, in different orders
    assert torch.equal(got, cv.conv_wgrad(xb, dyb, plan))            # reproducible
    acc = torch.ones_like(got)
    cv.conv_wgrad(xb, dyb, plan, out=acc, accumulate=True)
    torch.testing.assert_close(acc, got + 1.0, rtol=0, atol=1e-5 * scale + 1e-6)


@pytest.mark.parametrize("tile", ["256,256,32,5", "256,256,64,3", "256,256,32,1"])
@pytest.mark.parametrize("case", [(6, 14, 14, 256, 256, 3, 3, 1, 1, "SAME"), (5, 7, 9, 264, 520, 1, 1, 1, 1, "VALID"),
                                  (4, 15, 15, 256, 256, 3, 3, 2, 2, "SAME")])
def test_wgrad_on_the_eight_wave_tile(case, tile, monkeypatch):
    """The 256 x 256 weight-gradient tile (512 threads, 2 x 4 waves) against fp32 autograd: whole tiles, tiles with
    padding rows and columns (264 -> 520 channels), a strided layer; several splits and one."""
    B, H, W, Cin, Cout, kh, kw, sy, sx, _ = case
    x, w, pads = _mk(case, 11)
    xr = x.bfloat16().float().requires_grad_(True)
    wr = w.bfloat16().float().requires_grad_(True)
    y = ref.conv2d(xr, wr, (sy, sx), pads)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(12)).bfloat16()
    y.backward(dy.float())
    plan = cv.get_plan(H, W, Cin, Cout, kh, kw, sy, sx, pads, "cuda")
    monkeypatch.setenv("DTFA_WGRAD_TILE", tile)
    assert cv.pick_wgrad_tile(B * plan.oh * plan.ow, Cout, plan.K, Cin) == tuple(int(t) for t in tile.split(","))
    dw = cv.conv_wgrad(xr.detach().bfloat16().cuda(), dy.cuda(), plan)
    torch.cuda.synchronize()
    assert (dw.cpu() - wr.grad).abs().max().item() <= 2e-2 * wr.grad.abs().max().item()


@pytest.mark.parametrize("case", [(10, 16, 128, 1024, 1024, 1, 1, 1, 1, "VALID"), (12, 102, 128, 32, 128, 3, 3, 1, 1, "SAME")])
def test_tail_of_last_round_recut_into_small_tiles(case, monkeypatch):
    """A layer whose macro tiles fill whole rounds of the chip plus a fraction of one is covered by two launches
    (the rows of the fraction re-cut into small tiles).  Outputs are the bits of the single launch; the partial
    statistics rows of both launches add up to the same column sums; the same holds for a data gradient that
    emits batch-norm backward sums and adds a residual gradient."""
    B, H, W, Cin, Cout, kh, kw, sy, sx, _ = case
    x, w, pads = _mk(case, 11)
    xb, wb = x.bfloat16().cuda(), w.bfloat16().cuda()
    plan = cv.get_plan(H, W, Cin, Cout, kh, kw, sy, sx, pads, "cuda")
    M = B * plan.oh * plan.ow
    monkeypatch.setattr(cv, "_P3", False)           # (the second case is a shape the persistent kernel would take)
    monkeypatch.setattr(cv, "_WIDE", True)          # the rules may hand out the 256 x 128 macro tile (second case)
    monkeypatch.setattr(cv.tune, "MODE", "off")     # ... and the rules, not a measured table, choose here
    tile = cv.pick_fwd_tile(M, Cout, plan.K, True)
    m_main, tail = cv.split_tail(M, Cout, plan.K, tile[0], tile[1], True)
    assert tail is not None and 0 < m_main < M, (tile, m_main, tail)
    y2, part2 = cv.conv_fwd(xb, wb, plan, want_stats=True)
    g = torch.Generator().manual_seed(12)
    dy = torch.randn(B, plan.oh, plan.ow, Cout, generator=g).bfloat16().cuda()
    addend = torch.randn(B, H, W, Cin, generator=g).bfloat16().cuda()

    def dgrad():
        tap = cv.BnBackwardTap()
        gg = torch.Generator().manual_seed(13)
        tap.y = torch.randn(B, H, W, Cin, generator=gg).bfloat16().cuda()
        tap.mean = torch.randn(Cin, generator=gg).cuda() * 0.1
        tap.invstd = (torch.rand(Cin, generator=gg) + 0.5).cuda()
        tap.scale = (torch.rand(Cin, generator=gg) + 0.5).cuda()
        tap.shift = torch.randn(Cin, generator=gg).cuda() * 0.2
        tap.mask = None
        tap.relu_mode = 2
        dx = cv.conv_dgrad(dy, wb, plan, addend=addend, bn_tap=tap)
        return dx, tap.partials

    dx2, bp2 = dgrad()
    monkeypatch.setattr(cv, "_TAIL", False)
    y1, part1 = cv.conv_fwd(xb, wb, plan, want_stats=True)
    dx1, bp1 = dgrad()
    assert torch.equal(y1, y2)
    assert torch.equal(dx1, dx2)
    assert part2.shape[0] != part1.shape[0] or tail[0] >= 128
    torch.testing.assert_close(part2.double().sum(0), part1.double().sum(0), rtol=1e-5, atol=1e-3)
    torch.testing.assert_close(bp2.double().sum(0), bp1.double().sum(0), rtol=1e-5, atol=1e-3)
    # and against the fp32 definition on a slice of rows that spans the boundary between the two launches
    b0 = m_main // (plan.oh * plan.ow)
    want = ref.conv2d(xb[b0:b0 + 1].float().cpu(), wb.float().cpu(), (sy, sx), pads)
    torch.testing.assert_close(y2[b0:b0 + 1].float().cpu(), want, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("ksp", ["2", "4"])
@pytest.mark.parametrize("case", [(4, 8, 8, 448, 384, 3, 3, 1, 1, "SAME"), (2, 17, 17, 192, 160, 1, 7, 1, 1, "SAME"),
                                  (1, 8, 8, 2048, 200, 1, 1, 1, 1, "VALID")])
def test_reduction_split_inside_the_workgroup(case, ksp, monkeypatch):
    """Launches of few tiles with a deep reduction split every k-tile over 2 or 4 wave groups of one workgroup and sum
    the groups' accumulators through LDS: every epilogue (statistics rows, statistics by atomics, scale / bias / ReLU
    into a channel slice, data gradient with addend and batch-norm sums) agrees with the unsplit launch up to the
    rounding of a different summation order, and with the fp32 definition."""
    B, H, W, Cin, Cout, kh, kw, sy, sx, _ = case
    x, w, pads = _mk(case, 21)
    xb, wb = x.bfloat16().cuda(), w.bfloat16().cuda()
    plan = cv.get_plan(H, W, Cin, Cout, kh, kw, sy, sx, pads, "cuda")
    M = B * plan.oh * plan.ow
    assert cv.pick_fwd_tile(M, Cout, plan.K, True) == (64, 64, 64, 2)
    g = torch.Generator().manual_seed(22)
    dy = torch.randn(B, plan.oh, plan.ow, Cout, generator=g).bfloat16().cuda()
    addend = torch.randn(B, H, W, Cin, generator=g).bfloat16().cuda()
    scale = (torch.rand(Cout, generator=g) + 0.5).cuda()
    shift = torch.randn(Cout, generator=g).cuda()

    def run():
        y, part = cv.conv_fwd(xb, wb, plan, want_stats=True)
        acc = torch.zeros(2 * Cout, dtype=torch.float64, device="cuda")
        y_acc, _ = cv.conv_fwd(xb, wb, plan, stat_acc=acc)
        buf = torch.zeros(B, plan.oh, plan.ow, Cout + 16, dtype=torch.bfloat16, device="cuda")
        cv.conv_fwd(xb, wb, plan, bias=shift, relu=True, scale=scale, out=buf[..., 8:8 + Cout])
        tap = cv.BnBackwardTap()
        gg = torch.Generator().manual_seed(23)
        tap.y = torch.randn(B, H, W, Cin, generator=gg).bfloat16().cuda()
        tap.mean = torch.randn(Cin, generator=gg).cuda() * 0.1
        tap.invstd = (torch.rand(Cin, generator=gg) + 0.5).cuda()
        tap.scale = (torch.rand(Cin, generator=gg) + 0.5).cuda()
        tap.shift = torch.randn(Cin, generator=gg).cuda() * 0.2
        tap.mask = None
        tap.relu_mode = 2
        dx = cv.conv_dgrad(dy, wb, plan, addend=addend, bn_tap=tap)
        return y, part.double().sum(0), y_acc, acc.clone(), buf, dx, tap.partials.double().sum(0)

    monkeypatch.setattr(cv.tune, "MODE", "off")
    monkeypatch.setattr(cv, "_KSPLIT", "0")
    ref_out = run()
    monkeypatch.setattr(cv, "_KSPLIT", ksp)
    got = run()
    for a, b in zip(got, ref_out):
        scale_ab = b.abs().max().item() + 1e-6
        assert (a.double() - b.double()).abs().max().item() <= 1e-2 * scale_ab
    assert torch.equal(got[0], got[2])                    # both statistics paths write the same convolution output
    assert torch.equal(got[4][..., :8], torch.zeros_like(got[4][..., :8]))   # the slice's neighbours stay untouched
    want = ref.conv2d(xb.float().cpu(), wb.float().cpu(), (sy, sx), pads)
    torch.testing.assert_close(got[0].float().cpu(), want, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("case", [(2, 56, 56, 64, 64), (3, 20, 12, 64, 32), (1, 9, 87, 64, 48), (2, 7, 5, 64, 64),
                                  (70, 56, 56, 64, 64), (2, 9, 112, 64, 32), (1, 6, 240, 64, 64), (40, 30, 112, 64, 32),
                                  (2, 28, 28, 64, 128), (3, 20, 12, 64, 96),
                                  (2, 30, 112, 32, 64), (3, 17, 9, 32, 32), (1, 12, 147, 32, 96), (30, 40, 56, 32, 64)])
def test_same_size_3x3_by_the_persistent_kernel(case, monkeypatch):
    """conv_p3_kernel (3x3, stride 1, 64 input channels: one workgroup per CU walks a range of row tiles with the
    weights resident in LDS; the pixels a tile touches over all nine taps are brought into LDS once, one tile ahead,
    taps read at row offsets, out-of-image taps from a zero row) gives the bits of the ordinary kernel -- the
    reduction order is the same -- for the forward with statistics, the folded scale / bias / ReLU epilogue and the
    data gradient with an addend.  The last case has more tiles than the chip has CUs, so
    workgroups walk several tiles and both patch buffers are reused.  Images wider than 87 pixels run with one patch
    buffer per workgroup (several tiles per workgroup in that mode: 40 x 30 x 112); more than 64 output channels are
    slices of 64 columns, one workgroup each; the last four cases have 32 input channels (pixel rows of 64 bytes)."""
    B, H, W, Cin, Cout = case
    full = (B, H, W, Cin, Cout, 3, 3, 1, 1, "SAME")
    x, w, pads = _mk(full, 31)
    xb, wb = x.bfloat16().cuda(), w.bfloat16().cuda()
    plan = cv.get_plan(H, W, Cin, Cout, 3, 3, 1, 1, pads, "cuda")
    assert plan.near_taps
    g = torch.Generator().manual_seed(32)
    dy = torch.randn(B, H, W, Cin, generator=g).bfloat16().cuda()
    wd = (torch.randn(Cin, 3, 3, Cout, generator=g) * 0.05).bfloat16().cuda()   # a layer Cout -> Cin: its dgrad reads Cin
    dplan = cv.get_plan(H, W, Cout, Cin, 3, 3, 1, 1, pads, "cuda")
    addend = torch.randn(B, H, W, Cout, generator=g).bfloat16().cuda()
    scale = (torch.rand(Cout, generator=g) + 0.5).cuda()
    shift = torch.randn(Cout, generator=g).cuda()

    def run():
        y, part = cv.conv_fwd(xb, wb, plan, want_stats=True)
        z, _ = cv.conv_fwd(xb, wb, plan, bias=shift, relu=True, scale=scale)
        tap = cv.BnBackwardTap()
        gg = torch.Generator().manual_seed(33)
        tap.y = torch.randn(B, H, W, Cout, generator=gg).bfloat16().cuda()
        tap.mean = torch.randn(Cout, generator=gg).cuda() * 0.1
        tap.invstd = (torch.rand(Cout, generator=gg) + 0.5).cuda()
        tap.scale = (torch.rand(Cout, generator=gg) + 0.5).cuda()
        tap.shift = torch.randn(Cout, generator=gg).cuda() * 0.2
        tap.mask = None
        tap.relu_mode = 2
        dx = cv.conv_dgrad(dy, wd, dplan, addend=addend, bn_tap=tap)   # (stays with the ordinary kernel)
        dx0 = cv.conv_dgrad(dy, wd, dplan, addend=addend)
        return y, part, z, dx, tap.partials, dx0

    monkeypatch.setattr(cv.tune, "MODE", "off")
    monkeypatch.setattr(cv, "_P3_MIN_ROWS", 0)
    monkeypatch.setattr(cv, "_KSPLIT", "0")     # (a reduction split over wave groups sums in another order)
    monkeypatch.setattr(cv, "_P3", False)
    want = run()
    monkeypatch.setattr(cv, "_P3", True)
    assert cv.p3_takes(plan, B, Cin, Cout) and cv.p3_takes(dplan, B, Cin, Cout)   # ... or the comparison would be idle
    got = run()
    assert cv.pick_fwd_tile(B * H * W, Cout, 9 * Cin, True)[0] in (64, 128)
    for a, b in zip(got, want):
        if a.dtype == torch.bfloat16:
            assert torch.equal(a, b)
        else:   # partial rows: one per 128-row tile here, per tile of the rule's choice there
            torch.testing.assert_close(a.double().sum(0), b.double().sum(0), rtol=1e-5, atol=1e-3)
    if B <= 3:
        ref_y = ref.conv2d(xb.float().cpu(), wb.float().cpu(), (1, 1), pads)
        torch.testing.assert_close(got[0].float().cpu(), ref_y, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("case", [(2, 16, 24, 8, 64, 7, 4, 2, 1, (3, 3, 2, 1)),     # the pixel-pair stem, small
                                  (3, 224, 112, 8, 64, 7, 4, 2, 1, (3, 3, 2, 1)),   # ... at its size: tiles cross rows, many tiles per workgroup
                                  (2, 20, 18, 8, 40, 3, 3, 2, 2, "SAME"),           # odd tap count, idle columns
                                  (5, 224, 224, 8, 64, 3, 3, 2, 2, "SAME"),         # stride 2 both ways at the flower CNN's image size
                                  (1, 6, 10, 8, 48, 3, 3, 1, 1, "SAME"),            # unit stride, output narrower than a tile
                                  (2, 10, 150, 8, 64, 7, 4, 2, 1, (3, 3, 2, 1))])   # rows wider than a tile
def test_strided_layers_over_8_channel_pixels_by_the_persistent_kernel(case, monkeypatch):
    """conv_p8_kernel (8 input channels, input = stride x output: the contiguous range of input pixels a 128-row tile
    touches over all taps is staged in LDS once, a tap is a ds_read at the pixel's offset plus the tap's) gives the bits
    of the ordinary kernel -- same reduction order -- with statistics and with the scale / bias / ReLU epilogue and a
    residual."""
    B, H, W, Cin, Cout, kh, kw, sy, sx, _ = case
    x, w, pads = _mk(case, 41)
    xb, wb = x.bfloat16().cuda(), w.bfloat16().cuda()
    plan = cv.get_plan(H, W, Cin, Cout, kh, kw, sy, sx, pads, "cuda")
    g = torch.Generator().manual_seed(42)
    scale = (torch.rand(Cout, generator=g) + 0.5).cuda()
    shift = torch.randn(Cout, generator=g).cuda()
    res = torch.randn(B, plan.oh, plan.ow, Cout, generator=g).bfloat16().cuda()

    def run():
        y, part = cv.conv_fwd(xb, wb, plan, want_stats=True)
        z, _ = cv.conv_fwd(xb, wb, plan, bias=shift, relu=True, scale=scale, residual=res)
        return y, part, z

    monkeypatch.setattr(cv.tune, "MODE", "off")
    monkeypatch.setattr(cv, "_P3_MIN_ROWS", 0)
    monkeypatch.setattr(cv, "_KSPLIT", "0")
    monkeypatch.setattr(cv, "_P8", False)
    want = run()
    monkeypatch.setattr(cv, "_P8", True)
    assert cv.p8_takes(Cin, Cout, (kh, kw), H, W, plan.oh, plan.ow, sy, sx, B * plan.oh * plan.ow)
    got = run()
    for a, b in zip(got, want):
        if a.dtype == torch.bfloat16:
            assert torch.equal(a, b)
        else:
            torch.testing.assert_close(a.double().sum(0), b.double().sum(0), rtol=1e-5, atol=1e-3)
    if B * H * W <= 1 << 16:
        ref_y = ref.conv2d(xb.float().cpu(), wb.float().cpu(), (sy, sx), pads)
        torch.testing.assert_close(got[0].float().cpu(), ref_y, rtol=2e-2, atol=2e-2)


def test_convolutions_on_tensors_beyond_2_31_elements():
    """The benchmark's per-GPU batch of 4096 makes the stage-1 activations of ResNet-50 3.3 G elements (4096 x 56 x 56 x
    256).  The batch is the same 2048 images twice over: outputs and data gradients of the two halves must be the same
    bits, and the weight gradient twice that of one half -- a kernel whose indices wrapped would break one of them."""
    B, h = 4096, 2048
    plan = cv.get_plan(56, 56, 64, 256, 1, 1, 1, 1, (0, 0, 0, 0), "cuda")
    g = torch.Generator(device="cuda").manual_seed(61)
    x1 = torch.randn(h, 56, 56, 64, device="cuda", generator=g).bfloat16()
    x2 = torch.cat([x1, x1])
    w = (torch.randn(256, 1, 1, 64, device="cuda", generator=g) * 0.1).bfloat16()
    y, part = cv.conv_fwd(x2, w, plan, want_stats=True)
    assert y.numel() > 1 << 31
    assert torch.equal(y[:h], y[h:])
    ref_rows = x1[:2].float().reshape(-1, 64) @ w.float().reshape(256, 64).t()      # the upper half against fp32
    torch.testing.assert_close(y[h:h + 2].reshape(-1, 256).float(), ref_rows, rtol=2e-2, atol=2e-2)
    dy1 = y[:h]
    dx = cv.conv_dgrad(y, w, plan)                     # reads the 3.3 G element tensor
    assert torch.equal(dx[:h], dx[h:])
    dw2 = cv.conv_wgrad(x2, y, plan)
    dw1 = cv.conv_wgrad(x1, dy1.contiguous(), plan)
    assert (dw2 - 2 * dw1).abs().max().item() <= 1e-4 * dw2.abs().max().item()
    plan2 = cv.get_plan(56, 56, 256, 64, 1, 1, 1, 1, (0, 0, 0, 0), "cuda")
    w2 = (torch.randn(64, 1, 1, 256, device="cuda", generator=g) * 0.05).bfloat16()
    dbig = cv.conv_dgrad(x2, w2, plan2, addend=y)      # writes one, with an addend of the same size
    assert dbig.numel() > 1 << 31 and torch.equal(dbig[:h], dbig[h:])
    del y, dx, dbig
    torch.cuda.empty_cache()


def test_dgrad_masked_addend():
    """The dgrad epilogue adds only the elements of ``addend`` whose bit is set."""
    case = (2, 14, 14, 64, 32, 1, 1, 1, 1, "VALID")
    B, H, W, Cin, Cout, kh, kw, sy, sx, _ = case
    x, w, pads = _mk(case)
    plan = cv.get_plan(H, W, Cin, Cout, kh, kw, sy, sx, pads, "cuda")
    g = torch.Generator().manual_seed(11)
    dy = torch.randn(B, plan.oh, plan.ow, Cout, generator=g).bfloat16().cuda()
    add = torch.randn(B, H, W, Cin, generator=g).bfloat16().cuda()
    keep = torch.rand(B, H, W, Cin, generator=g).cuda() > 0.5
    mask = (keep.view(-1, 8).to(torch.uint8) << torch.arange(8, device="cuda", dtype=torch.uint8)).sum(1).to(torch.uint8)
    base = cv.conv_dgrad(dy, w.bfloat16().cuda(), plan)
    got = cv.conv_dgrad(dy, w.bfloat16().cuda(), plan, addend=add, addend_mask=mask)
    want = (base.float() + torch.where(keep, add.float(), torch.zeros_like(add, dtype=torch.float32))).bfloat16()
    # the kernel adds before rounding the data-gradient to bf16; allow one rounding step of difference
    torch.testing.assert_close(got.float(), want.float(), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("gated", [False, True])
@pytest.mark.parametrize("case,stride", [((2, 14, 14, 64, 32, 1, 1, 1, 1, "VALID"), (2, 2)),
                                         ((3, 7, 9, 128, 64, 1, 1, 1, 1, "VALID"), (2, 2)),
                                         ((2, 12, 10, 256, 128, 1, 1, 1, 1, "VALID"), (2, 1)),
                                         ((2, 9, 11, 40, 72, 3, 3, 1, 1, "SAME"), (4, 2))])
def test_dgrad_coarse_grid_addend(case, stride, gated):
    """An addend on the pixel grid [::sy, ::sx] of the data gradient (the gradient of a strided shortcut) is added in
    place: same result as scattering it into a zero tensor of the full shape first -- also when the launch stores the
    gated gradient and emits its per-channel sums."""
    B, H, W, Cin, Cout, kh, kw, sy, sx, _ = case
    x, w, pads = _mk(case)
    plan = cv.get_plan(H, W, Cin, Cout, kh, kw, sy, sx, pads, "cuda")
    assert cv.dgrad_coarse_addend_ok(plan, stride)
    g = torch.Generator().manual_seed(13)
    dy = torch.randn(B, plan.oh, plan.ow, Cout, generator=g).bfloat16().cuda()
    add = torch.randn(B, -(-H // stride[0]), -(-W // stride[1]), Cin, generator=g).bfloat16().cuda()
    full = torch.zeros(B, H, W, Cin, dtype=torch.bfloat16, device="cuda")
    full[:, ::stride[0], ::stride[1], :] = add

    def tap():
        if not gated:
            return None
        t = cv.BnBackwardTap()
        keep = torch.rand(B, H, W, Cin, generator=torch.Generator().manual_seed(3)).cuda() > 0.4
        t.mask = (keep.view(-1, 8).to(torch.uint8) << torch.arange(8, device="cuda", dtype=torch.uint8)).sum(1).to(torch.uint8)
        t.relu_mode, t.gate_store = 3, True
        return t

    t0, t1 = tap(), tap()
    want = cv.conv_dgrad(dy, w.bfloat16().cuda(), plan, addend=full, bn_tap=t0)
    got = cv.conv_dgrad(dy, w.bfloat16().cuda(), plan, addend=add, addend_stride=stride, bn_tap=t1)
    assert torch.equal(got, want)
    if gated:
        assert torch.equal(t0.partials, t1.partials)
    with pytest.raises(ValueError):
        cv.conv_dgrad(dy, w.bfloat16().cuda(), plan, addend=full, addend_stride=stride)


@pytest.mark.parametrize("relu_mode", [0, 2, 3])
@pytest.mark.parametrize("case", [(2, 14, 14, 64, 32, 1, 1, 1, 1, "VALID"), (3, 9, 11, 40, 72, 3, 3, 1, 1, "SAME"),
                                  (2, 12, 12, 64, 64, 3, 3, 1, 1, "SAME"), (2, 12, 12, 64, 64, 3, 3, 2, 2, "SAME"),
                                  (3, 9, 11, 40, 72, 3, 3, 2, 2, "SAME"), (2, 14, 14, 128, 128, 3, 3, 2, 2, (1, 1, 1, 1))])
def test_dgrad_emits_bn_backward_partials(case, relu_mode):
    """A data-gradient launch can emit (sum g, sum g*xhat) of the batch-norm layer below it; the sums equal
    those of the gradient tensor it wrote."""
    B, H, W, Cin, Cout, kh, kw, sy, sx, _ = case
    x, w, pads = _mk(case)
    plan = cv.get_plan(H, W, Cin, Cout, kh, kw, sy, sx, pads, "cuda")
    assert cv.dgrad_can_tap(plan)
    g = torch.Generator().manual_seed(5)
    dy = torch.randn(B, plan.oh, plan.ow, Cout, generator=g).bfloat16().cuda()
    tap = cv.BnBackwardTap()
    tap.y = torch.randn(B, H, W, Cin, generator=g).bfloat16().cuda()
    tap.mean = torch.randn(Cin, generator=g).cuda() * 0.1
    tap.invstd = (torch.rand(Cin, generator=g) + 0.5).cuda()
    tap.scale = (torch.rand(Cin, generator=g) + 0.5).cuda()
    tap.shift = torch.randn(Cin, generator=g).cuda() * 0.2
    keep = torch.rand(B, H, W, Cin, generator=g).cuda() > 0.4
    tap.mask = (keep.view(-1, 8).to(torch.uint8) << torch.arange(8, device="cuda", dtype=torch.uint8)).sum(1).to(torch.uint8)
    tap.relu_mode = relu_mode
    dx = cv.conv_dgrad(dy, w.bfloat16().cuda(), plan, bn_tap=tap)
    assert torch.equal(dx, cv.conv_dgrad(dy, w.bfloat16().cuda(), plan))
    yf, gf = tap.y.float(), dx.float()
    if relu_mode == 2:
        gf = torch.where(yf * tap.scale + tap.shift > 0, gf, torch.zeros_like(gf))
    if relu_mode == 3:
        gf = torch.where(keep, gf, torch.zeros_like(gf))
    xhat = (yf - tap.mean) * tap.invstd
    want1 = gf.reshape(-1, Cin).double().sum(0)
    want2 = (gf * xhat).reshape(-1, Cin).double().sum(0)
    got = tap.partials.double().sum(0)
    torch.testing.assert_close(got[0], want1, rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(got[1], want2, rtol=1e-3, atol=1e-2)


@pytest.mark.parametrize("case", [(32, 17, 17, 192, 192, 1, 7, 1, 1, "SAME"), (32, 8, 8, 448, 384, 3, 3, 1, 1, "SAME"),
                                  (4, 35, 35, 48, 64, 5, 5, 1, 1, "SAME"), (3, 9, 11, 40, 72, 3, 3, 2, 2, "VALID")])
def test_statistics_accumulated_by_atomics_match_partial_rows(case):
    """``stat_acc``: the column sums land in one fp64 accumulator (added by every tile, from every XCD) and equal the
    sum of the partial rows of the ordinary statistics epilogue; the output tensor is the same."""
    B, H, W, Cin, Cout, kh, kw, sy, sx, _ = case
    x, w, pads = _mk(case)
    plan = cv.get_plan(H, W, Cin, Cout, kh, kw, sy, sx, pads, "cuda")
    xb, wb = x.bfloat16().cuda(), w.bfloat16().cuda()
    y0, part = cv.conv_fwd(xb, wb, plan, want_stats=True)
    acc = torch.zeros(2 * Cout, dtype=torch.float64, device="cuda")
    y1, none = cv.conv_fwd(xb, wb, plan, stat_acc=acc)
    assert none is None and torch.equal(y0, y1)
    want = part.double().sum(0).reshape(-1)
    torch.testing.assert_close(acc, want, rtol=1e-6, atol=1e-4)
    # accumulators add up over launches
    cv.conv_fwd(xb, wb, plan, stat_acc=acc)
    torch.testing.assert_close(acc, 2 * want, rtol=1e-6, atol=2e-4)
    with pytest.raises(ValueError):
        cv.conv_fwd(xb, wb, plan, stat_acc=acc, want_stats=True)
    # replicas: row tile t adds to replica t mod R; the replicas add up to the same sums, and the normalisation
    # kernel that consumes them gives the bits it gives from a single accumulator
    from distributed_tensorflow_amd.ops import kernels as K

    rep = 8
    accr = torch.zeros(rep * 2 * Cout, dtype=torch.float64, device="cuda")
    y2, _ = cv.conv_fwd(xb, wb, plan, stat_acc=accr)
    assert torch.equal(y2, y0)
    torch.testing.assert_close(accr.view(rep, -1).sum(0), want, rtol=1e-6, atol=1e-4)
    if (B * plan.oh * plan.ow + 63) // 64 >= rep:
        assert (accr.view(rep, -1).abs().sum(1) > 0).all()
    acc1 = torch.zeros(2 * Cout, dtype=torch.float64, device="cuda")
    cv.conv_fwd(xb, wb, plan, stat_acc=acc1)
    rows = B * plan.oh * plan.ow
    beta = torch.linspace(-1, 1, Cout, device="cuda")
    z1 = K.bn_apply_sums(y0, acc1, rows, 1e-3, None, beta, None, None, 0.1, True)
    zr = K.bn_apply_sums(y0, accr, rows, 1e-3, None, beta, None, None, 0.1, True)
    torch.testing.assert_close(zr.float(), z1.float(), rtol=0, atol=2e-2)   # (sums of fp64 atomics in another order)
    assert (zr.float() - z1.float()).abs().gt(0).float().mean().item() < 1e-3
    assert cv.stat_replicas(32 * 35 * 35) == 8 and cv.stat_replicas(32 * 17 * 17) == 2 and cv.stat_replicas(64) == 1


@pytest.mark.parametrize("M,N,K,relu,bias", [(32, 1024, 2048, True, True), (1, 8, 1024, False, True), (5, 40, 48, True, False),
                                             (128, 1008, 2048, False, True), (33, 72, 80, True, True), (7, 2048, 16, False, False)])
def test_dense_skinny_matches_fp32_reference(M, N, K, relu, bias):
    """Few-row dense layer (reduction spread over the waves of a workgroup) against an fp32 matmul of the same bf16
    operands."""
    g = torch.Generator().manual_seed(M * 31 + N)
    x = torch.randn(M, K, generator=g).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().cuda()
    b = torch.randn(N, generator=g).cuda() if bias else None
    assert cv.dense_skinny_ok(M, K, N)
    got = cv.dense_skinny(x, w, b, relu)
    want = x.float() @ w.float().t()
    if bias:
        want = want + b
    if relu:
        want = torch.relu(want)
    assert got.shape == (M, N) and got.dtype == torch.bfloat16
    torch.testing.assert_close(got.float(), want, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("B,hw,C,N", [(64, 28, 64, 64), (32, 14, 128, 128), (16, 14, 256, 256), (8, 7, 512, 512), (3, 5, 64, 64),
                                      (16, 14, 128, 512)])
def test_wgrad_column_sums_ride_along(B, hw, C, N):
    """``colsum``: x^T 1 from the launch that computes dy^T x, equal to the separately computed gradient and the
    fp32 column sums of x."""
    g = torch.Generator().manual_seed(C + N)
    x = torch.randn(B, hw, hw, C, generator=g).bfloat16().cuda()
    dy = x if N == C else torch.randn(B, hw, hw, N, generator=g).bfloat16().cuda()
    plan = cv.get_plan(hw, hw, C, N, 1, 1, 1, 1, (0, 0, 0, 0), "cuda")
    M = B * hw * hw
    if not cv.wgrad_colsum_ok(plan, M):
        pytest.skip("tile of this shape has no column-sum build")
    want = cv.conv_wgrad(x, dy, plan)
    got, s = cv.conv_wgrad(x, dy, plan, colsum=True)
    assert torch.equal(got, want)
    ref = x.double().reshape(-1, C).sum(0)
    torch.testing.assert_close(s.double(), ref, rtol=1e-4, atol=1e-2)
