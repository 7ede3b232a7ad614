"""Plain PyTorch reference implementations of every op.

These are (a) the numerics oracle the HIP kernels are tested against and
(b) the execution path for CPU tensors (the CPU plumbing configurations:
single-process small CNN, ps/worker over TCP/gloo).  All tensors are NHWC;
weights are ``[Cout, KH, KW, Cin]``.  Semantics follow TensorFlow where the two
frameworks differ: explicit (possibly asymmetric) padding, average pooling
that can exclude padded cells, biased batch variance in the normalisation.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn.functional as F


def conv2d(x, w, stride: Tuple[int, int], pads: Tuple[int, int, int, int], bias=None, relu=False):
    """x [B,H,W,Cin], w [Cout,KH,KW,Cin], pads = (top, bottom, left, right)."""
    pt, pb, pl, pr = pads
    xn = x.permute(0, 3, 1, 2)
    if pt or pb or pl or pr:
        xn = F.pad(xn, (pl, pr, pt, pb))
    y = F.conv2d(xn, w.permute(0, 3, 1, 2), None if bias is None else bias.to(x.dtype), stride=stride)
    if relu:
        y = F.relu(y)
    return y.permute(0, 2, 3, 1).contiguous()


def batch_norm_train(y, gamma, beta, run_mean, run_var, eps: float, momentum: float, update_running=True):
    """Per-channel batch statistics over all but the last axis (biased variance)."""
    yf = y.float()
    red = tuple(range(yf.dim() - 1))
    mean = yf.mean(red)
    var = yf.var(red, unbiased=False)
    n = yf.numel() // yf.shape[-1]
    if update_running and run_mean is not None:
        with torch.no_grad():
            unb = var * (n / max(n - 1, 1))
            run_mean.mul_(1 - momentum).add_(mean.detach() * momentum)
            run_var.mul_(1 - momentum).add_(unb.detach() * momentum)
    invstd = torch.rsqrt(var + eps)
    out = (yf - mean) * invstd
    if gamma is not None:
        out = out * gamma.float()
    if beta is not None:
        out = out + beta.float()
    return out.to(y.dtype), mean, invstd


def batch_norm_eval(y, gamma, beta, run_mean, run_var, eps: float):
    scale = torch.rsqrt(run_var.float() + eps)
    if gamma is not None:
        scale = scale * gamma.float()
    shift = -run_mean.float() * scale
    if beta is not None:
        shift = shift + beta.float()
    return (y.float() * scale + shift).to(y.dtype)


def max_pool(x, k: Tuple[int, int], stride: Tuple[int, int], pads: Tuple[int, int, int, int]):
    pt, pb, pl, pr = pads
    xn = x.permute(0, 3, 1, 2)
    if pt or pb or pl or pr:
        xn = F.pad(xn, (pl, pr, pt, pb), value=float("-inf"))
    y = F.max_pool2d(xn, k, stride)
    return y.permute(0, 2, 3, 1).contiguous()


def avg_pool(x, k: Tuple[int, int], stride: Tuple[int, int], pads: Tuple[int, int, int, int],
             count_include_pad: bool = False):
    pt, pb, pl, pr = pads
    xn = x.permute(0, 3, 1, 2).float()
    xp = F.pad(xn, (pl, pr, pt, pb))
    s = F.avg_pool2d(xp, k, stride) * (k[0] * k[1])
    if count_include_pad:
        y = s / (k[0] * k[1])
    else:
        ones = torch.ones(1, 1, xn.shape[2], xn.shape[3], dtype=xn.dtype, device=xn.device)
        cnt = F.avg_pool2d(F.pad(ones, (pl, pr, pt, pb)), k, stride) * (k[0] * k[1])
        y = s / cnt.clamp_min(1.0)
    return y.permute(0, 2, 3, 1).contiguous().to(x.dtype)


def global_avg_pool(x):
    return x.float().mean(dim=(1, 2)).to(x.dtype)


def dense(x, w, bias=None, relu=False):
    """x [B, In], w [Out, In]."""
    y = F.linear(x, w, None if bias is None else bias.to(x.dtype))
    return F.relu(y) if relu else y


def dropout(x, rate: float, training: bool, generator: Optional[torch.Generator] = None):
    if not training or rate <= 0.0:
        return x
    keep = 1.0 - rate
    mask = (torch.rand(x.shape, device=x.device, generator=generator) >= rate).to(x.dtype)
    return x * mask / keep


def softmax_xent(logits, labels, n_classes: Optional[int] = None, act_l2: float = 0.0):
    """Returns (per-row loss, top1 index, top1 raw logit, correct flag).

    ``logits`` may carry padded columns beyond ``n_classes``; they are ignored.
    ``act_l2`` adds ``0.5 * act_l2 * sum(logits**2)`` per row (TensorFlow's
    ``l2_regularizer(scale)(t) = scale * l2_loss(t)`` applied to the logits).
    """
    n = n_classes if n_classes is not None else logits.shape[-1]
    lg = logits[..., :n].float()
    loss = F.cross_entropy(lg, labels.long(), reduction="none")
    if act_l2:
        loss = loss + 0.5 * act_l2 * (lg * lg).sum(-1)
    score, top1 = lg.max(-1)
    correct = (top1 == labels.long()).to(torch.int32)
    return loss, top1.to(torch.int32), score, correct
