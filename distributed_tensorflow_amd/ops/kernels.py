"""Host-side launch helpers for the non-convolution HIP kernels.

Each helper validates operands, allocates outputs through the torch caching
allocator (so the launches are stream-ordered and hipGraph-capturable) and
launches on the current stream.
"""
from __future__ import annotations

import os

import ctypes as C
from typing import Optional, Tuple

import torch

from . import hipk

BF16 = torch.bfloat16
F32 = torch.float32


def _as2d(t: torch.Tensor) -> Tuple[int, int]:
    c = t.shape[-1]
    return t.numel() // c, c


# ------------------------------------------------------------------ batch norm
_REDUCE_ROWS_PER_SPLIT = int(os.environ.get("DTFA_REDUCE_ROWS_PER_SPLIT", "16"))
_DIRECT_ROWS = int(os.environ.get("DTFA_FINALIZE_DIRECT_ROWS", "2048"))


def reduce_partials(part: torch.Tensor, max_rows: int = 1024, direct_rows: int = 0) -> torch.Tensor:
    """[R][NV][C] -> [RS<=max_rows][NV][C] (fixed summation order).  Up to ``direct_rows`` rows are left
    alone: the finalize kernels sum that many themselves (32 row lanes), which saves a launch per layer.
    Above that each output row sums ~16 input rows, so the pass is wide (R/16 x NV*C threads) and short instead
    of a few long serial chains of dependent loads."""
    R, NV, Cc = part.shape
    if R <= (direct_rows or _DIRECT_ROWS):
        return part
    rs = min(max_rows, max(64, -(-R // _REDUCE_ROWS_PER_SPLIT)))
    out = torch.empty((rs, NV, Cc), dtype=F32, device=part.device)
    hipk.check(hipk.lib().dtfa_partial_reduce(part.data_ptr(), out.data_ptr(), R, NV, Cc, rs, hipk.stream()),
               "partial_reduce")
    return out


def sum_partial_rows(part: torch.Tensor) -> torch.Tensor:
    """[R][NV][C] -> [NV][C]: all rows summed in a fixed order (two wide passes instead of one long chain)."""
    R, NV, Cc = part.shape
    L = hipk.lib()
    if R > 64:
        mid = torch.empty((64, NV, Cc), dtype=F32, device=part.device)
        hipk.check(L.dtfa_partial_reduce(part.data_ptr(), mid.data_ptr(), R, NV, Cc, 64, hipk.stream()), "partial_reduce")
        part, R = mid, 64
    out = torch.empty((1, NV, Cc), dtype=F32, device=part.device)
    hipk.check(L.dtfa_partial_reduce(part.data_ptr(), out.data_ptr(), R, NV, Cc, 1, hipk.stream()), "partial_reduce")
    return out[0]


def fold_forward(weight: torch.Tensor, T: torch.Tensor, s: torch.Tensor, gamma, beta, run_mean, run_var, count: int,
                 eps: float, momentum: float):
    """Statistics of ``y = x W^T`` from ``T = W (x^T x)`` and ``s = x^T 1``: (mean, invstd, scale, shift), running
    statistics updated in place (csrc/hip/fold.hip)."""
    hipk.require(weight, BF16, "weight")
    hipk.require(T, F32, "T")
    hipk.require(s, F32, "s")
    N = weight.shape[0]
    Cc = weight.numel() // N
    if tuple(T.shape) != (N, Cc) or s.numel() != Cc:
        raise ValueError("fold_forward shapes")
    for t, nm in ((gamma, "gamma"), (beta, "beta"), (run_mean, "running_mean"), (run_var, "running_var")):
        if t is not None:
            hipk.require(t, F32, nm)
            if t.numel() != N:
                raise ValueError("%s has %d elements, expected %d" % (nm, t.numel(), N))
    out = torch.empty((4, N), dtype=F32, device=weight.device)
    hipk.check(hipk.lib().dtfa_fold_fwd(weight.data_ptr(), T.data_ptr(), s.data_ptr(), hipk.ptr(gamma), hipk.ptr(beta),
                                        out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(),
                                        hipk.ptr(run_mean), hipk.ptr(run_var), N, Cc, float(count), float(eps),
                                        float(momentum), hipk.stream()), "fold_fwd")
    return out[0], out[1], out[2], out[3]


def fold_backward(weight, P, T, s, sum_g, mean, invstd, scale, count: int, dgamma_out=None, dbeta_out=None, dw_out=None,
                  acc_gamma=False, acc_beta=False, acc_w=False, want_dx=True):
    """Backward algebra of the folded batch-norm (csrc/hip/fold.hip).  Writes ``dgamma`` / ``dbeta`` / ``dW`` into the
    given fp32 buffers (added when the ``acc_*`` flag is set) or fresh ones, and returns
    ``(dgamma, dbeta, dW, wd, W1, c2)`` with ``wd`` [C, N + C] bf16 whose first N columns hold ``(c0 W)^T`` (the
    rest is for ``(W^T W1)``), ``W1 = c1 W`` fp32 and ``c2`` -- the last three None unless ``want_dx``."""
    hipk.require(weight, BF16, "weight")
    N = weight.shape[0]
    Cc = weight.numel() // N
    for t, nm, shp in ((P, "P", (N, Cc)), (T, "T", (N, Cc))):
        hipk.require(t, F32, nm)
        if tuple(t.shape) != shp:
            raise ValueError("fold_backward: %s shape" % nm)
    for t, nm, n in ((s, "s", Cc), (sum_g, "sum_g", N), (mean, "mean", N), (invstd, "invstd", N), (scale, "scale", N)):
        hipk.require(t, F32, nm)
        if t.numel() != n:
            raise ValueError("fold_backward: %s size" % nm)
    dev = weight.device
    for o, n in ((dgamma_out, N), (dbeta_out, N), (dw_out, N * Cc)):
        if o is not None and (o.dtype != F32 or o.numel() != n or not o.is_contiguous()):
            raise ValueError("fold_backward: bad gradient output buffer")
    dgamma = dgamma_out if dgamma_out is not None else torch.empty(N, dtype=F32, device=dev)
    dbeta = dbeta_out if dbeta_out is not None else torch.empty(N, dtype=F32, device=dev)
    dW = dw_out if dw_out is not None else torch.empty((N, Cc), dtype=F32, device=dev)
    wd = W1 = c2 = None
    if want_dx:
        wd = torch.empty((Cc, N + Cc), dtype=BF16, device=dev)
        W1 = torch.empty((N, Cc), dtype=F32, device=dev)
        c2 = torch.empty(N, dtype=F32, device=dev)
    hipk.check(hipk.lib().dtfa_fold_bwd(weight.data_ptr(), P.data_ptr(), T.data_ptr(), s.data_ptr(), sum_g.data_ptr(),
                                        mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), float(count),
                                        dgamma.data_ptr(), dbeta.data_ptr(), dW.data_ptr(), hipk.ptr(wd), hipk.ptr(W1),
                                        hipk.ptr(c2), N, Cc, N + Cc, 1 if (acc_gamma and dgamma_out is not None) else 0,
                                        1 if (acc_beta and dbeta_out is not None) else 0,
                                        1 if (acc_w and dw_out is not None) else 0, hipk.stream()), "fold_bwd")
    return dgamma, dbeta, dW, wd, W1, c2


def bn_stats_partials(x: torch.Tensor) -> torch.Tensor:
    hipk.require(x, BF16, "x")
    M, Cc = _as2d(x)
    L = hipk.lib()
    rows = L.dtfa_col_partials_rows(M, Cc)
    part = torch.empty((rows, 2, Cc), dtype=F32, device=x.device)
    hipk.check(L.dtfa_bn_stats_partials(x.data_ptr(), part.data_ptr(), M, Cc, hipk.stream()), "bn_stats_partials")
    return part


def bn_finalize(part: torch.Tensor, count: int, eps: float, gamma, beta, run_mean, run_var, momentum: float):
    part = reduce_partials(part)
    R, _, Cc = part.shape
    dev = part.device
    for t, nm in ((gamma, "gamma"), (beta, "beta"), (run_mean, "running_mean"), (run_var, "running_var")):
        if t is not None:
            hipk.require(t, F32, nm)
            if t.numel() != Cc:
                raise ValueError("%s has %d elements, expected %d" % (nm, t.numel(), Cc))
    out = torch.empty((4, Cc), dtype=F32, device=dev)
    mean, invstd, scale, shift = out[0], out[1], out[2], out[3]
    hipk.check(hipk.lib().dtfa_bn_finalize(part.data_ptr(), R, Cc, float(count), float(eps), hipk.ptr(gamma),
                                           hipk.ptr(beta), mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(),
                                           shift.data_ptr(), hipk.ptr(run_mean), hipk.ptr(run_var), float(momentum),
                                           hipk.stream()), "bn_finalize")
    return mean, invstd, scale, shift


def bn_eval_coeffs(gamma, beta, run_mean, run_var, eps: float):
    hipk.require(run_mean, F32, "running_mean")
    hipk.require(run_var, F32, "running_var")
    Cc = run_mean.numel()
    out = torch.empty((2, Cc), dtype=F32, device=run_mean.device)
    hipk.check(hipk.lib().dtfa_bn_eval_coeffs(Cc, float(eps), hipk.ptr(gamma), hipk.ptr(beta), run_mean.data_ptr(),
                                              run_var.data_ptr(), out[0].data_ptr(), out[1].data_ptr(),
                                              hipk.stream()), "bn_eval_coeffs")
    return out[0], out[1]


def channel_slice_pitch(out: torch.Tensor, like: torch.Tensor) -> int:
    """Validate that ``out`` is a channel slice ``buf[..., a:b]`` of a contiguous NHWC buffer with the
    shape of ``like``; returns its row pitch in elements."""
    if out.dtype != like.dtype or out.shape != like.shape or out.device != like.device:
        raise ValueError("output slice does not match the result (shape %s vs %s)" % (tuple(out.shape), tuple(like.shape)))
    pitch = out.stride(-2)
    want = [pitch]
    for d in range(out.dim() - 2, 0, -1):
        want.insert(0, want[0] * out.shape[d])
    if out.stride(-1) != 1 or list(out.stride()[:-1]) != want or pitch % 8 or out.data_ptr() % 16:
        raise ValueError("output must be a 16-byte aligned channel slice of a contiguous NHWC tensor")
    return pitch


def bn_apply(y: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, relu: bool,
             residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
             want_mask: bool = False):
    """``act(y*scale + shift (+ residual))``; ``out`` may be a channel slice of a wider NHWC tensor.
    With ``want_mask`` (ReLU only) also returns the activation's sign as one bit per element
    (uint8 [numel / 8]), which is all the backward pass needs of the output: ``(z, mask)``."""
    hipk.require(y, BF16, "y")
    M, Cc = _as2d(y)
    if scale.numel() != Cc or shift.numel() != Cc:
        raise ValueError("scale/shift size mismatch")
    if residual is not None:
        hipk.require(residual, BF16, "residual")
        if residual.shape != y.shape:
            raise ValueError("residual shape mismatch")
    if out is None:
        z, ldz = torch.empty_like(y), Cc
    else:
        z, ldz = out, channel_slice_pitch(out, y)
    mask = None
    if want_mask:
        if not relu or out is not None:
            raise ValueError("the sign bitmask needs a ReLU and a dense output")
        mask = torch.empty(y.numel() // 8, dtype=torch.uint8, device=y.device)
    hipk.check(hipk.lib().dtfa_bn_apply(y.data_ptr(), hipk.ptr(residual), z.data_ptr(), scale.data_ptr(),
                                        shift.data_ptr(), hipk.ptr(mask), M, Cc, ldz, 1 if relu else 0,
                                        hipk.stream()), "bn_apply")
    return (z, mask) if want_mask else z


BN_SUMS_MAX_C = 4096


def bn_apply_sums(y: torch.Tensor, sums: torch.Tensor, count: int, eps: float, gamma, beta, run_mean, run_var,
                  momentum: float, relu: bool, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``act(bn(y))`` with batch statistics given as fp64 column sums ``sums`` = [sum y | sum y^2] (2C values, as the
    convolution's statistics epilogue accumulates them): coefficients, running-statistics update and the elementwise
    pass in one launch.  ``out`` may be a channel slice of a wider NHWC tensor.  ``sums`` may hold several replicas of the
    2C values (``conv.stat_replicas``), which the kernel adds up."""
    hipk.require(y, BF16, "y")
    M, Cc = _as2d(y)
    rep = sums.numel() // (2 * Cc) if Cc else 0
    if (sums.dtype != torch.float64 or rep < 1 or sums.numel() != rep * 2 * Cc or rep & (rep - 1) or rep > 64
            or not sums.is_contiguous() or Cc > BN_SUMS_MAX_C):
        raise ValueError("column sums must be 2^k x %d contiguous fp64 values" % (2 * Cc))
    for t, nm in ((gamma, "gamma"), (beta, "beta"), (run_mean, "running_mean"), (run_var, "running_var")):
        if t is not None:
            hipk.require(t, F32, nm)
            if t.numel() != Cc:
                raise ValueError("%s has %d elements, expected %d" % (nm, t.numel(), Cc))
    if out is None:
        z, ldz = torch.empty_like(y), Cc
    else:
        z, ldz = out, channel_slice_pitch(out, y)
    hipk.check(hipk.lib().dtfa_bn_apply_sums(y.data_ptr(), z.data_ptr(), sums.data_ptr(), float(count), float(eps),
                                             hipk.ptr(gamma), hipk.ptr(beta), hipk.ptr(run_mean), hipk.ptr(run_var),
                                             float(momentum), M, Cc, ldz, 1 if relu else 0, rep, hipk.stream()),
               "bn_apply_sums")
    return z


def _bn_bwd_mode(dz, y, zout, relu, scale, shift, mask):
    hipk.require(dz, BF16, "dz")
    hipk.require(y, BF16, "y")
    if dz.shape != y.shape:
        raise ValueError("dz/y shape mismatch")
    mode = 0
    if mask is not None:
        if mask.dtype != torch.uint8 or mask.numel() * 8 != y.numel() or not mask.is_contiguous():
            raise ValueError("bitmask size mismatch")
        mode = 3
    elif relu:
        if zout is not None:
            hipk.require(zout, BF16, "zout")
            if zout.shape != y.shape:
                raise ValueError("zout shape mismatch")
            mode = 1
        else:
            if scale is None or shift is None:
                raise ValueError("ReLU mask needs either zout or scale/shift")
            mode = 2
    zp = hipk.ptr(zout) if mode == 1 else (mask.data_ptr() if mode == 3 else None)
    sp = scale.data_ptr() if mode == 2 else None
    hp = shift.data_ptr() if mode == 2 else None
    return mode, zp, sp, hp


def bn_backward_coefs(dz, y, mean, invstd, gamma, relu: bool = False, zout=None, dgamma_out=None, dbeta_out=None,
                      accumulate=False, scale=None, shift=None, mask=None, partials=None):
    """First half of the batch-norm backward: the per-channel reductions (``partials`` if a producer kernel
    already emitted them, else a pass over ``dz`` / ``y``), then ``dgamma``, ``dbeta`` and the three coefficient
    rows of the elementwise half ``dy = coef[0]*g + coef[1]*y + coef[2]`` (``g`` = gated ``dz``).
    Returns ``(coef [3, C], dgamma, dbeta)``; gating arguments as in :func:`bn_backward`."""
    mode, zp, sp, hp = _bn_bwd_mode(dz, y, zout, relu, scale, shift, mask)
    M, Cc = _as2d(y)
    L = hipk.lib()
    if partials is not None:
        if partials.dim() != 3 or partials.shape[1] != 2 or partials.shape[2] != Cc or partials.dtype != F32:
            raise ValueError("bad batch-norm backward partials")
        part = partials
    else:
        rows = L.dtfa_col_partials_rows(M, Cc)
        part = torch.empty((rows, 2, Cc), dtype=F32, device=y.device)
        hipk.check(L.dtfa_bn_bwd_partials(dz.data_ptr(), y.data_ptr(), zp, mean.data_ptr(), invstd.data_ptr(), sp,
                                          hp, part.data_ptr(), M, Cc, mode, hipk.stream()), "bn_bwd_partials")
    part = reduce_partials(part)
    coef = torch.empty((3, Cc), dtype=F32, device=y.device)
    for o in (dgamma_out, dbeta_out):
        if o is not None and (o.dtype != F32 or o.numel() != Cc or not o.is_contiguous()):
            raise ValueError("bad per-channel gradient output buffer")
    acc = 1 if (accumulate and dbeta_out is not None) else 0
    # gamma-less layers still produce a dgamma; it lands in scratch so accumulation never touches a real buffer
    dgamma = dgamma_out if dgamma_out is not None else torch.zeros(Cc, dtype=F32, device=y.device) if acc \
        else torch.empty(Cc, dtype=F32, device=y.device)
    dbeta = dbeta_out if dbeta_out is not None else torch.empty(Cc, dtype=F32, device=y.device)
    hipk.check(L.dtfa_bn_bwd_finalize(part.data_ptr(), part.shape[0], Cc, float(M), hipk.ptr(gamma),
                                      mean.data_ptr(), invstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                      coef[0].data_ptr(), coef[1].data_ptr(), coef[2].data_ptr(), acc,
                                      hipk.stream()), "bn_bwd_finalize")
    return coef, dgamma, dbeta


def bn_frozen_param_grads(g: torch.Tensor, y: torch.Tensor, mean: torch.Tensor, invstd: torch.Tensor):
    """Gradients of the scale and offset of a batch-norm that normalises with FIXED statistics (moving averages):
    ``z = gamma * (y - mean) * invstd + beta`` gives ``dbeta = sum g`` and ``dgamma = sum g * (y - mean) * invstd``
    over the pixels, with ``g`` the (already gated) gradient of ``z``.  Returns ``(dgamma, dbeta)``, fp32 ``[C]``."""
    hipk.require(g, BF16, "g")
    hipk.require(y, BF16, "y")
    if g.shape != y.shape:
        raise ValueError("g/y shape mismatch")
    M, Cc = _as2d(y)
    L = hipk.lib()
    rows = L.dtfa_col_partials_rows(M, Cc)
    part = torch.empty((rows, 2, Cc), dtype=F32, device=y.device)
    hipk.check(L.dtfa_bn_bwd_partials(g.data_ptr(), y.data_ptr(), None, mean.data_ptr(), invstd.data_ptr(), None, None,
                                      part.data_ptr(), M, Cc, 0, hipk.stream()), "bn_bwd_partials")
    tot = sum_partial_rows(part)
    return tot[1], tot[0]


def bn_backward_apply(dz, y, coef, relu: bool = False, zout=None, want_dres: bool = False, scale=None, shift=None,
                      mask=None):
    """Second half: ``dy = coef[0]*g + coef[1]*y + coef[2]`` (and ``dres = g`` on request)."""
    mode, zp, sp, hp = _bn_bwd_mode(dz, y, zout, relu, scale, shift, mask)
    M, Cc = _as2d(y)
    if coef.shape != (3, Cc) or coef.dtype != F32 or not coef.is_contiguous():
        raise ValueError("bad batch-norm backward coefficients")
    dy = torch.empty_like(y)
    dres = torch.empty_like(y) if want_dres else None
    hipk.check(hipk.lib().dtfa_bn_bwd_apply(dz.data_ptr(), y.data_ptr(), zp, dy.data_ptr(), hipk.ptr(dres),
                                            coef[0].data_ptr(), coef[1].data_ptr(), coef[2].data_ptr(), sp, hp, M, Cc,
                                            mode, hipk.stream()), "bn_bwd_apply")
    return dy, dres


def bn_backward(dz, y, zout, mean, invstd, gamma, relu: bool, want_dres: bool, dgamma_out=None, dbeta_out=None,
                accumulate=False, scale=None, shift=None, mask=None, partials=None):
    """Returns (dy, dres or None, dgamma, dbeta).

    ``partials`` ([R,2,C] sums of g and g*xhat, e.g. emitted by the data-gradient kernel that produced
    ``dz``) replaces the reduction pass over ``dz`` / ``y``.

    ``mask`` (uint8, one bit per element) gates ``dz`` -- the stored sign of a ReLU output, this layer's
    own or that of the activation this layer's output was added into -- and takes precedence over
    ``relu`` / ``zout``.

    With ``relu`` the activation mask is taken from ``zout`` (the stored
    activation output) when it is given, otherwise it is recomputed from
    ``y*scale + shift > 0`` (valid when no residual was added), which saves
    reading one full tensor in each of the two passes."""
    coef, dgamma, dbeta = bn_backward_coefs(dz, y, mean, invstd, gamma, relu, zout, dgamma_out, dbeta_out, accumulate,
                                            scale, shift, mask, partials)
    dy, dres = bn_backward_apply(dz, y, coef, relu, zout, want_dres, scale, shift, mask)
    return dy, dres, dgamma, dbeta


def mask_mul(dz: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """``dz`` where the bit of ``mask`` is set, zero elsewhere."""
    hipk.require(dz, BF16, "dz")
    if mask.dtype != torch.uint8 or mask.numel() * 8 != dz.numel():
        raise ValueError("bitmask size mismatch")
    out = torch.empty_like(dz)
    hipk.check(hipk.lib().dtfa_mask_mul(dz.data_ptr(), mask.data_ptr(), out.data_ptr(), dz.numel(), hipk.stream()),
               "mask_mul")
    return out


def relu_bwd(dz: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
    hipk.require(dz, BF16, "dz")
    hipk.require(z, BF16, "z")
    if dz.numel() != z.numel() or dz.numel() % 8:
        raise ValueError("relu_bwd size")
    dx = torch.empty_like(dz)
    hipk.check(hipk.lib().dtfa_relu_bwd(dz.data_ptr(), z.data_ptr(), dx.data_ptr(), dz.numel(), hipk.stream()),
               "relu_bwd")
    return dx


def add_bf16(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    hipk.require(a, BF16, "a")
    hipk.require(b, BF16, "b")
    if a.shape != b.shape or a.numel() % 8:
        raise ValueError("add_bf16 size")
    o = torch.empty_like(a)
    hipk.check(hipk.lib().dtfa_add_bf16(a.data_ptr(), b.data_ptr(), o.data_ptr(), a.numel(), hipk.stream()), "add")
    return o


# ------------------------------------------------------------------ pooling
def _pool_args(x_shape, k, stride, pads, oh, ow, include_pad=False) -> hipk.PoolArgs:
    B, IH, IW, Cc = x_shape
    p = hipk.PoolArgs()
    p.B, p.IH, p.IW, p.C, p.OH, p.OW = B, IH, IW, Cc, oh, ow
    p.kh, p.kw, p.sy, p.sx = k[0], k[1], stride[0], stride[1]
    p.pt, p.pl = pads[0], pads[2]
    p.count_include_pad = 1 if include_pad else 0
    return p


def pool_out_shape(x_shape, k, stride, pads):
    B, IH, IW, Cc = x_shape
    oh = (IH + pads[0] + pads[1] - k[0]) // stride[0] + 1
    ow = (IW + pads[2] + pads[3] - k[1]) // stride[1] + 1
    if oh <= 0 or ow <= 0:
        raise ValueError("empty pooling output")
    return oh, ow


def maxpool_fwd(x, k, stride, pads, want_index=True):
    hipk.require(x, BF16, "x")
    oh, ow = pool_out_shape(x.shape, k, stride, pads)
    p = _pool_args(x.shape, k, stride, pads, oh, ow)
    y = torch.empty((x.shape[0], oh, ow, x.shape[3]), dtype=BF16, device=x.device)
    idx = torch.empty(y.shape, dtype=torch.uint8, device=x.device) if want_index else None
    hipk.check(hipk.lib().dtfa_maxpool_fwd(x.data_ptr(), y.data_ptr(), hipk.ptr(idx), C.byref(p), hipk.stream()),
               "maxpool_fwd")
    return y, idx


def maxpool_bwd(dy, idx, x_shape, k, stride, pads):
    hipk.require(dy, BF16, "dy")
    hipk.require(idx, torch.uint8, "idx")
    oh, ow = pool_out_shape(x_shape, k, stride, pads)
    if tuple(dy.shape) != (x_shape[0], oh, ow, x_shape[3]) or idx.shape != dy.shape:
        raise ValueError("maxpool_bwd shapes")
    p = _pool_args(x_shape, k, stride, pads, oh, ow)
    dx = torch.empty(tuple(x_shape), dtype=BF16, device=dy.device)
    hipk.check(hipk.lib().dtfa_maxpool_bwd(dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), C.byref(p), hipk.stream()),
               "maxpool_bwd")
    return dx


def bn_relu_maxpool_fwd(y, scale, shift, relu: bool, k, stride, pads, want_index=True, want_winners=False):
    """``maxpool(act(y*scale + shift))`` without storing the full-resolution activation.
    Returns (pooled, winning-tap bytes or None); with ``want_winners`` also the un-normalised value ``y`` of every
    window's winner (pooled shape, bf16) -- what the batch-norm backward sums need of ``y``."""
    hipk.require(y, BF16, "y")
    Cc = y.shape[3]
    if scale.numel() != Cc or shift.numel() != Cc or scale.dtype != F32 or shift.dtype != F32:
        raise ValueError("scale/shift size mismatch")
    oh, ow = pool_out_shape(y.shape, k, stride, pads)
    p = _pool_args(y.shape, k, stride, pads, oh, ow)
    out = torch.empty((y.shape[0], oh, ow, Cc), dtype=BF16, device=y.device)
    idx = torch.empty(out.shape, dtype=torch.uint8, device=y.device) if want_index else None
    ywin = torch.empty_like(out) if want_winners else None
    hipk.check(hipk.lib().dtfa_bn_relu_maxpool_fwd(y.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.data_ptr(),
                                                   hipk.ptr(idx), hipk.ptr(ywin), C.byref(p), 1 if relu else 0,
                                                   hipk.stream()),
               "bn_relu_maxpool_fwd")
    return (out, idx, ywin) if want_winners else (out, idx)


def bn_pool_backward(dp, idx, y, mean, invstd, gamma, scale, shift, relu: bool, k, stride, pads,
                     dgamma_out=None, dbeta_out=None, accumulate=False, ywin=None):
    """Backward of ``maxpool(act(bn(y)))`` given the pooled gradient ``dp`` and the winning-tap bytes:
    returns (dy, dgamma, dbeta).  The full-resolution gradient of the pool is gathered on the fly in both
    passes and never stored.  ``ywin`` (the winners' values kept by the forward pass) spares the first pass its
    gather into ``y``."""
    hipk.require(dp, BF16, "dp")
    hipk.require(y, BF16, "y")
    hipk.require(idx, torch.uint8, "idx")
    oh, ow = pool_out_shape(y.shape, k, stride, pads)
    Cc = y.shape[3]
    if tuple(dp.shape) != (y.shape[0], oh, ow, Cc) or idx.shape != dp.shape:
        raise ValueError("bn_pool_backward shapes")
    for t in (mean, invstd, scale, shift):
        if t.dtype != F32 or t.numel() != Cc:
            raise ValueError("bn_pool_backward statistics size")
    p = _pool_args(y.shape, k, stride, pads, oh, ow)
    L = hipk.lib()
    M = y.numel() // Cc
    rows = L.dtfa_bn_pool_partials_rows(C.byref(p))
    part = torch.empty((rows, 2, Cc), dtype=F32, device=y.device)
    if ywin is not None:
        hipk.require(ywin, BF16, "ywin")
        if ywin.shape != dp.shape:
            raise ValueError("bn_pool_backward: winners' values must have the pooled shape")
    hipk.check(L.dtfa_bn_pool_bwd_partials(dp.data_ptr(), idx.data_ptr(), y.data_ptr(), hipk.ptr(ywin), mean.data_ptr(),
                                           invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), part.data_ptr(),
                                           C.byref(p), 1 if relu else 0, hipk.stream()), "bn_pool_bwd_partials")
    part = reduce_partials(part)
    coef = torch.empty((3, Cc), dtype=F32, device=y.device)
    for o in (dgamma_out, dbeta_out):
        if o is not None and (o.dtype != F32 or o.numel() != Cc or not o.is_contiguous()):
            raise ValueError("bad per-channel gradient output buffer")
    acc = 1 if (accumulate and dbeta_out is not None) else 0
    dgamma = dgamma_out if dgamma_out is not None else torch.zeros(Cc, dtype=F32, device=y.device) if acc \
        else torch.empty(Cc, dtype=F32, device=y.device)
    dbeta = dbeta_out if dbeta_out is not None else torch.empty(Cc, dtype=F32, device=y.device)
    hipk.check(L.dtfa_bn_bwd_finalize(part.data_ptr(), part.shape[0], Cc, float(M), hipk.ptr(gamma),
                                      mean.data_ptr(), invstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                      coef[0].data_ptr(), coef[1].data_ptr(), coef[2].data_ptr(), acc,
                                      hipk.stream()), "bn_bwd_finalize")
    dy = torch.empty_like(y)
    hipk.check(L.dtfa_bn_pool_bwd_apply(dp.data_ptr(), idx.data_ptr(), y.data_ptr(), dy.data_ptr(),
                                        coef[0].data_ptr(), coef[1].data_ptr(), coef[2].data_ptr(), scale.data_ptr(),
                                        shift.data_ptr(), C.byref(p), 1 if relu else 0, hipk.stream()),
               "bn_pool_bwd_apply")
    return dy, dgamma, dbeta


def avgpool_fwd(x, k, stride, pads, include_pad=False):
    hipk.require(x, BF16, "x")
    oh, ow = pool_out_shape(x.shape, k, stride, pads)
    p = _pool_args(x.shape, k, stride, pads, oh, ow, include_pad)
    y = torch.empty((x.shape[0], oh, ow, x.shape[3]), dtype=BF16, device=x.device)
    hipk.check(hipk.lib().dtfa_avgpool_fwd(x.data_ptr(), y.data_ptr(), C.byref(p), hipk.stream()), "avgpool_fwd")
    return y


def avgpool_bwd(dy, x_shape, k, stride, pads, include_pad=False):
    hipk.require(dy, BF16, "dy")
    oh, ow = pool_out_shape(x_shape, k, stride, pads)
    if tuple(dy.shape) != (x_shape[0], oh, ow, x_shape[3]):
        raise ValueError("avgpool_bwd shapes")
    p = _pool_args(x_shape, k, stride, pads, oh, ow, include_pad)
    dx = torch.empty(tuple(x_shape), dtype=BF16, device=dy.device)
    hipk.check(hipk.lib().dtfa_avgpool_bwd(dy.data_ptr(), dx.data_ptr(), C.byref(p), hipk.stream()), "avgpool_bwd")
    return dx


def gap_fwd(x):
    hipk.require(x, BF16, "x")
    B, H, W, Cc = x.shape
    y = torch.empty((B, Cc), dtype=BF16, device=x.device)
    hipk.check(hipk.lib().dtfa_gap_fwd(x.data_ptr(), y.data_ptr(), B, H * W, Cc, hipk.stream()), "gap_fwd")
    return y


def gap_bwd(dy, x_shape):
    hipk.require(dy, BF16, "dy")
    B, H, W, Cc = x_shape
    if tuple(dy.shape) != (B, Cc):
        raise ValueError("gap_bwd shapes")
    dx = torch.empty(tuple(x_shape), dtype=BF16, device=dy.device)
    hipk.check(hipk.lib().dtfa_gap_bwd(dy.data_ptr(), dx.data_ptr(), B, H * W, Cc, hipk.stream()), "gap_bwd")
    return dx


# ------------------------------------------------------------------ loss
def softmax_xent(logits: torch.Tensor, labels: torch.Tensor, n_classes: int, grad_scale: float, act_l2: float = 0.0,
                 want_grad: bool = True):
    """logits [B, ld] bf16 (ld >= n_classes), labels [B] int32.
    Returns (loss[B] f32, dlogits[B, ld] bf16 or None, top1[B] i32, score[B] f32, correct[B] i32)."""
    hipk.require(logits, BF16, "logits")
    hipk.require(labels, torch.int32, "labels")
    B, ld = logits.shape
    if labels.numel() != B or not (1 <= n_classes <= ld):
        raise ValueError("softmax_xent shapes")
    dev = logits.device
    loss = torch.empty(B, dtype=F32, device=dev)
    score = torch.empty(B, dtype=F32, device=dev)
    top1 = torch.empty(B, dtype=torch.int32, device=dev)
    correct = torch.empty(B, dtype=torch.int32, device=dev)
    dl = torch.empty_like(logits) if want_grad else None
    hipk.check(hipk.lib().dtfa_softmax_xent(logits.data_ptr(), labels.data_ptr(), loss.data_ptr(), hipk.ptr(dl),
                                            top1.data_ptr(), score.data_ptr(), correct.data_ptr(), B, n_classes, ld,
                                            float(grad_scale), float(act_l2), hipk.stream()), "softmax_xent")
    return loss, dl, top1, score, correct


def act_l2_fwd(h: torch.Tensor, coef: float) -> torch.Tensor:
    """``coef / 2 * sum(h**2)`` of a bf16 activation as an fp32 scalar tensor [1] (two launches, fixed grid)."""
    hipk.require(h, BF16, "h")
    n = h.numel()
    if n == 0 or n % 8:
        raise ValueError("activity L2: element count must be a positive multiple of 8")
    L = hipk.lib()
    part = torch.empty(L.dtfa_act_l2_blocks(n), dtype=F32, device=h.device)
    out = torch.empty(1, dtype=F32, device=h.device)
    hipk.check(L.dtfa_act_l2_fwd(h.data_ptr(), n, float(coef), part.data_ptr(), out.data_ptr(), hipk.stream()), "act_l2_fwd")
    return out


def act_l2_bwd(h: torch.Tensor, dh_in: Optional[torch.Tensor], g: torch.Tensor, coef: float) -> torch.Tensor:
    """``dh_in + coef * g * h`` in one pass (``dh_in`` may be None); ``g`` is an fp32 device scalar."""
    hipk.require(h, BF16, "h")
    hipk.require(g, F32, "g")
    if dh_in is not None:
        hipk.require(dh_in, BF16, "dh_in")
        if dh_in.numel() != h.numel():
            raise ValueError("activity L2 backward: gradient / activation size mismatch")
    if g.numel() != 1 or h.numel() % 8:
        raise ValueError("activity L2 backward shapes")
    out = torch.empty_like(h)
    hipk.check(hipk.lib().dtfa_act_l2_bwd(h.data_ptr(), hipk.ptr(dh_in), g.data_ptr(), float(coef), out.data_ptr(),
                                          h.numel(), hipk.stream()), "act_l2_bwd")
    return out


def mean_f32(v: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    hipk.require(v, F32, "v")
    out = torch.empty(1, dtype=F32, device=v.device)
    hipk.check(hipk.lib().dtfa_mean_f32(v.data_ptr(), out.data_ptr(), v.numel(), float(scale), hipk.stream()), "mean")
    return out


# ------------------------------------------------------------------ dropout
_rng_steps = {}


def rng_step(device) -> torch.Tensor:
    """Device-resident step counter folded into every dropout seed (int64[1]).  Advancing it between
    replays of a captured step graph gives each replay fresh masks."""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    t = _rng_steps.get(key)
    if t is None:
        t = torch.zeros(1, dtype=torch.int64, device=device)
        _rng_steps[key] = t
    return t


def subsample(x: torch.Tensor, stride) -> torch.Tensor:
    """``x[:, ::sy, ::sx, :]`` of a contiguous NHWC bf16 tensor as a new contiguous tensor (the block input on the pixel
    grid of a strided projection shortcut)."""
    hipk.require(x, BF16, "x")
    sy, sx = stride
    B, H, W, Cc = x.shape
    if Cc % 8:
        raise ValueError("subsample: channels must be a multiple of 8")
    y = torch.empty((B, -(-H // sy), -(-W // sx), Cc), dtype=BF16, device=x.device)
    hipk.check(hipk.lib().dtfa_subsample(x.data_ptr(), y.data_ptr(), B, H, W, Cc, sy, sx, hipk.stream()), "subsample")
    return y


def dropout(x: torch.Tensor, rate: float, seed: int) -> torch.Tensor:
    hipk.require(x, BF16, "x")
    y = torch.empty_like(x)
    hipk.check(hipk.lib().dtfa_dropout(x.data_ptr(), y.data_ptr(), x.numel(), seed & ((1 << 64) - 1),
                                       rng_step(x.device).data_ptr(), float(rate), hipk.stream()), "dropout")
    return y


# ------------------------------------------------------------------ optimizers
def adam_step(w, m, v, g, w16, hyper, b1, b2, eps):
    for t, nm in ((w, "w"), (m, "m"), (v, "v")):
        hipk.require(t, F32, nm)
    n = w.numel()
    if m.numel() != n or v.numel() != n or g.numel() != n:
        raise ValueError("adam_step sizes")
    if g.dtype not in (F32, BF16) or not g.is_contiguous():
        raise ValueError("adam_step gradient dtype")
    if w16 is not None:
        hipk.require(w16, BF16, "w16")
        if w16.numel() != n:
            raise ValueError("adam_step shadow size")
    hipk.require(hyper, F32, "hyper")
    hipk.check(hipk.lib().dtfa_adam(w.data_ptr(), m.data_ptr(), v.data_ptr(), g.data_ptr(),
                                    1 if g.dtype == BF16 else 0, hipk.ptr(w16), n, hyper.data_ptr(), float(b1),
                                    float(b2), float(eps), hipk.stream()), "adam")


def momentum_step(w, buf, g, w16, hyper, mu, wd, nesterov):
    hipk.require(w, F32, "w")
    hipk.require(buf, F32, "buf")
    n = w.numel()
    if buf.numel() != n or g.numel() != n:
        raise ValueError("momentum_step sizes")
    if g.dtype not in (F32, BF16) or not g.is_contiguous():
        raise ValueError("momentum_step gradient dtype")
    if w16 is not None:
        hipk.require(w16, BF16, "w16")
    hipk.require(hyper, F32, "hyper")
    hipk.check(hipk.lib().dtfa_momentum(w.data_ptr(), buf.data_ptr(), g.data_ptr(), 1 if g.dtype == BF16 else 0,
                                        hipk.ptr(w16), n, hyper.data_ptr(), float(mu), float(wd),
                                        1 if nesterov else 0, hipk.stream()), "momentum")


def cast_f32_bf16(src: torch.Tensor, dst: torch.Tensor) -> None:
    hipk.require(src, F32, "src")
    hipk.require(dst, BF16, "dst")
    if src.numel() != dst.numel():
        raise ValueError("cast sizes")
    hipk.check(hipk.lib().dtfa_cast_f32_bf16(src.data_ptr(), dst.data_ptr(), src.numel(), hipk.stream()), "cast")
