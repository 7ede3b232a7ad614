"""Building blocks shared by the model families.

All modules take and return NHWC tensors.  Convolution weights are stored
``[Cout, KH, KW, Cin]`` (the layout the implicit-GEMM kernels read); TensorFlow
checkpoints use ``[KH, KW, Cin, Cout]`` and are converted at the checkpoint
boundary (see ``io.name_map``).  Every parametrised module records the
TensorFlow variable scope it corresponds to in ``tf_scope`` so that checkpoints
use the reference's variable names.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from ..ops import functional as F

Pad = Union[str, int, Tuple[int, int], Tuple[int, int, int, int]]


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class ConvBnAct(nn.Module):
    """conv (no bias) -> batch-norm -> (+ residual) -> optional ReLU, fused into
    one autograd node on the GPU.

    ``bn_scale=False`` reproduces TF-slim's ``batch_norm(scale=False)`` (no
    gamma).  ``cin_pad`` pads the input-channel axis of the *weight* with zeros
    (used by the 3-channel stems, whose images are stored with 8 channels).
    """

    def __init__(self, cin, cout, kernel, stride=1, padding: Pad = "SAME", relu=True, bn_scale=True,
                 bn_eps=1e-5, bn_momentum=0.1, tf_scope: Optional[str] = None, cin_pad: Optional[int] = None,
                 zero_init_gamma=False):
        super().__init__()
        kh, kw = _pair(kernel)
        self.cin_logical = cin
        cin_phys = cin_pad if cin_pad is not None else cin
        self.stride = _pair(stride)
        self.padding = padding
        self.relu = relu
        self.eps, self.momentum = bn_eps, bn_momentum
        self.tf_scope = tf_scope
        w = torch.zeros(cout, kh, kw, cin_phys)
        std = math.sqrt(2.0 / (kh * kw * cin))
        nn.init.trunc_normal_(w[..., :cin], std=std, a=-2 * std, b=2 * std)
        self.weight = nn.Parameter(w)
        if bn_scale:
            self.gamma = nn.Parameter(torch.zeros(cout) if zero_init_gamma else torch.ones(cout))
        else:
            self.register_parameter("gamma", None)
        self.beta = nn.Parameter(torch.zeros(cout))
        self.register_buffer("running_mean", torch.zeros(cout))
        self.register_buffer("running_var", torch.ones(cout))
        # batch-norm parameters and statistics stay fp32 when the module is cast to bf16
        for t in (self.gamma, self.beta):
            if t is not None:
                t._keep_fp32 = True

    def forward_pool(self, x, kernel=(3, 3), stride=(2, 2), padding="VALID", weight=None, conv_stride=None,
                     conv_padding=None):
        """This layer followed by a max-pool, fused on the device (the un-pooled activation is not stored).
        ``weight`` / ``conv_stride`` / ``conv_padding`` run the layer as an equivalent convolution over a
        re-interpreted input (a differentiable function of ``self.weight``, see ``models.resnet``)."""
        F.param_fence(self)
        if self.training:
            self._coeff_cache = None  # the kernels update the running statistics in place, behind torch's versions
        return F.conv_bn_act_pool(x, self.weight if weight is None else weight, self.gamma, self.beta,
                                  self.running_mean, self.running_var,
                                  stride=self.stride if conv_stride is None else conv_stride,
                                  padding=self.padding if conv_padding is None else conv_padding,
                                  eps=self.eps, momentum=self.momentum,
                                  relu=self.relu, training=self.training, pool_kernel=_pair(kernel),
                                  pool_stride=_pair(stride), pool_padding=padding)

    def _eval_coeffs(self):
        """(scale, shift) of the running statistics on the device, recomputed only when one of their source
        tensors has been written or replaced (inference calls this layer thousands of times per restore)."""
        src = (self.gamma, self.beta, self.running_mean, self.running_var)
        key = tuple((t.data_ptr(), t._version) for t in src if t is not None)
        cached = getattr(self, "_coeff_cache", None)
        if cached is None or cached[0] != key:
            from ..ops import kernels as K
            cached = (key, K.bn_eval_coeffs(self.gamma, self.beta, self.running_mean, self.running_var, self.eps))
            self._coeff_cache = cached
        return cached[1]

    def forward(self, x, residual=None, links=None, out=None):
        F.param_fence(self)
        coeffs = None
        if self.training:
            self._coeff_cache = None  # the kernels update the running statistics in place, behind torch's versions
        elif x.is_cuda and not torch.is_grad_enabled():
            coeffs = self._eval_coeffs()
        return F.conv_bn_act(x, self.weight, self.gamma, self.beta, self.running_mean, self.running_var,
                             stride=self.stride, padding=self.padding, eps=self.eps, momentum=self.momentum,
                             relu=self.relu, training=self.training, residual=residual, links=links, out=out,
                             eval_coeffs=coeffs)


class Conv2d(nn.Module):
    """Plain convolution with optional bias / ReLU (no normalisation)."""

    def __init__(self, cin, cout, kernel, stride=1, padding: Pad = "SAME", bias=True, relu=False,
                 tf_scope: Optional[str] = None):
        super().__init__()
        kh, kw = _pair(kernel)
        self.stride, self.padding, self.relu, self.tf_scope = _pair(stride), padding, relu, tf_scope
        w = torch.empty(cout, kh, kw, cin)
        std = math.sqrt(2.0 / (kh * kw * cin))
        nn.init.trunc_normal_(w, std=std, a=-2 * std, b=2 * std)
        self.weight = nn.Parameter(w)
        if bias:
            self.bias = nn.Parameter(torch.zeros(cout))
            self.bias._keep_fp32 = True
        else:
            self.register_parameter("bias", None)

    def forward(self, x):
        F.param_fence(self)
        return F.conv2d(x, self.weight, self.bias, stride=self.stride, padding=self.padding, relu=self.relu)


class Dense(nn.Module):
    """Fully connected layer, ``tf.layers.dense`` initialisation (Glorot
    uniform kernel, zero bias).  The output width is padded to a multiple of 8
    rows for the GPU kernels; ``forward`` returns the padded activations and
    ``out_features`` tells consumers how many columns are real."""

    def __init__(self, n_in, n_out, relu=False, tf_scope: Optional[str] = None):
        super().__init__()
        self.in_features, self.out_features = n_in, n_out
        self.out_padded = (n_out + 7) // 8 * 8
        self.relu, self.tf_scope = relu, tf_scope
        w = torch.zeros(self.out_padded, n_in)
        limit = math.sqrt(6.0 / (n_in + n_out))
        nn.init.uniform_(w[:n_out], -limit, limit)
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.zeros(self.out_padded))
        self.bias._keep_fp32 = True

    def forward(self, x):
        F.param_fence(self)
        return F.dense(x, self.weight, self.bias, relu=self.relu)


def cast_for_device(module: nn.Module, device, compute_dtype=None) -> nn.Module:
    """Move a model to ``device``; on a GPU the conv/dense weights become bf16
    while batch-norm parameters, biases and running statistics stay fp32."""
    device = torch.device(device)
    if compute_dtype is None:
        compute_dtype = torch.bfloat16 if device.type == "cuda" else torch.float32
    for p in module.parameters():
        # 1-D parameters (batch-norm scale/offset, biases) are consumed in fp32 by the kernels
        keep = p.dim() <= 1
        p.data = p.data.to(device=device, dtype=torch.float32 if keep else compute_dtype)
        p._keep_fp32 = keep
    for name, b in list(module.named_buffers()):
        b.data = b.data.to(device=device)
    module.compute_dtype = compute_dtype
    if hasattr(module, "input_channels") and hasattr(module, "_pair_idx"):
        # preferred image layout of this replica (both layouts are accepted wherever they are valid)
        module.input_channels = 4 if device.type == "cuda" else 8
    return module
