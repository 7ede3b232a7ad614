"""ResNet-50 (v1.5: the stride sits on the 3x3 convolution) for NHWC bf16.

The reference repository contains no ResNet; it is one of the target
configurations of this framework (synthetic 224x224 flower-shaped data, large
per-GPU batches).  Variable scopes follow TF-slim's ``resnet_v1_50`` so
checkpoints use the familiar names.  The 3-channel input is stored with 8
channels (the stem weight is zero-padded along Cin) so that every convolution
meets the kernels' ``Cin % 8 == 0`` requirement.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..ops import conv as cv
from ..ops import functional as F
from .layers import ConvBnAct, Dense

IMAGE_CHANNELS = 8  # 3 real + 5 zero channels
PAIRED_IMAGE_CHANNELS = 4  # 3 real + 1 zero channel: the layout of the pixel-pair stem (GPU, even image width)


def _pair_tables():
    """Gather table of the pixel-pair stem.  A [B, H, W, 4] image with W even is, without moving a byte, a
    [B, H, W/2, 8] image of horizontal pixel pairs; the 7x7 / stride-2 stem over the former is a 7x4 / stride-(2,1)
    convolution over the latter (pair taps -2 .. 1; input column 2*ox + kx - 3 = 2*(ox + dp) + parity).  Entry
    (j, parity, c) of a pair-tap row takes the weight of kx = 2*j + parity - 1, channel c; kx = -1 and the pad
    channel are zero.  K = 7*4*8 = 224 instead of 7*7*8 = 392."""
    idx = torch.zeros(32, dtype=torch.long)
    mask = torch.zeros(32)
    for j in range(4):
        for par in range(2):
            for c in range(4):
                kx = 2 * j + par - 1
                if 0 <= kx <= 6 and c < 3:
                    idx[j * 8 + par * 4 + c] = kx * IMAGE_CHANNELS + c
                    mask[j * 8 + par * 4 + c] = 1.0
    return idx, mask


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, width, stride, scope, zero_init_residual=True):
        super().__init__()
        cout = width * self.expansion
        bn = dict(bn_eps=1e-5, bn_momentum=0.1)
        self.conv1 = ConvBnAct(cin, width, 1, 1, "VALID", tf_scope=scope + "/conv1", **bn)
        self.conv2 = ConvBnAct(width, width, 3, stride, (1, 1, 1, 1), tf_scope=scope + "/conv2", **bn)
        self.conv3 = ConvBnAct(width, cout, 1, 1, "VALID", relu=True, tf_scope=scope + "/conv3",
                               zero_init_gamma=zero_init_residual, **bn)
        if stride != 1 or cin != cout:
            self.shortcut = ConvBnAct(cin, cout, 1, stride, "VALID", relu=False, tf_scope=scope + "/shortcut", **bn)
        else:
            self.shortcut = None

    def forward(self, x, in_tap=None, out_tap=None):
        """``in_tap`` / ``out_tap`` (GPU training): batch-norm backward taps shared with the previous /
        next block -- ``in_tap`` belongs to the layer that produced ``x``, ``out_tap`` is filled by conv3."""
        if x.is_cuda and x.requires_grad and torch.is_grad_enabled() and self.training:
            # Training on the GPU: the layers of the block pass gradients to each other directly instead
            # of through autograd sums (see ops.functional.GradSlot), and every data-gradient kernel emits
            # the batch-norm backward reduction of the layer below it (conv.BnBackwardTap).
            to_conv1 = F.GradSlot()
            tap1, tap2 = cv.BnBackwardTap(), cv.BnBackwardTap()
            if self.shortcut is None:
                # identity block: conv1's dgrad epilogue adds the masked gradient of the block output
                h = self.conv1(x, links=F.Links(input_grad=to_conv1, input_tap=in_tap, output_tap=tap1))
                h = self.conv2(h, links=F.Links(input_tap=tap1, output_tap=tap2))
                return self.conv3(h, residual=x, links=F.Links(residual_grad=to_conv1, input_tap=tap2,
                                                               output_tap=out_tap))
            # projection block: the shortcut gates its incoming gradient with conv3's sign bits and hands
            # its input gradient to conv1's dgrad epilogue.  It is called after conv2 so that its backward
            # runs before conv1's (autograd runs later-created nodes first).
            h = self.conv1(x, links=F.Links(input_grad=to_conv1, input_tap=in_tap, output_tap=tap1))
            h = self.conv2(h, links=F.Links(input_tap=tap1, output_tap=tap2))
            if F.expand_dual_ok(self.conv3, self.shortcut, h, x, self.training):
                # both batch-norms folded through their 1x1 convolutions, one GEMM over [h | x at the stride]
                for m in (self.conv3, self.shortcut):
                    m._coeff_cache = None
                return F.expand_dual(h, self.conv3, x, self.shortcut,
                                     links=F.Links(input_tap=tap2, output_tap=out_tap, input_grad_out=to_conv1))
            sign = F.GradSlot()
            identity = self.shortcut(x, links=F.Links(output_mask=sign, input_grad_out=to_conv1))
            return self.conv3(h, residual=identity, links=F.Links(residual_mask=sign, input_tap=tap2,
                                                                  output_tap=out_tap))
        identity = x if self.shortcut is None else self.shortcut(x)
        out = self.conv2(self.conv1(x))
        return self.conv3(out, residual=identity)


class ResNet50(nn.Module):
    def __init__(self, num_classes=5, zero_init_residual=True, root="resnet_v1_50"):
        super().__init__()
        self.num_classes = num_classes
        self.stem = ConvBnAct(3, 64, 7, 2, (3, 3, 3, 3), tf_scope=root + "/conv1", cin_pad=IMAGE_CHANNELS)
        blocks = []
        cin = 64
        for bi, (width, n, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)], start=1):
            for ui in range(n):
                scope = "%s/block%d/unit_%d/bottleneck_v1" % (root, bi, ui + 1)
                blocks.append(Bottleneck(cin, width, stride if ui == 0 else 1, scope, zero_init_residual))
                cin = width * 4
        self.blocks = nn.ModuleList(blocks)
        self.fc = Dense(2048, num_classes, tf_scope=root + "/logits")
        idx, mask = _pair_tables()
        self.input_channels = IMAGE_CHANNELS   # cast_for_device switches GPU models to the pixel-pair layout
        self.register_buffer("_pair_idx", idx, persistent=False)
        self.register_buffer("_pair_mask", mask, persistent=False)

    def _stem(self, x):
        if x.shape[3] == PAIRED_IMAGE_CHANNELS:
            B, H, W, _ = x.shape
            if not x.is_cuda or W % 2:
                raise ValueError("4-channel images are the pixel-pair layout of the device stem (even width)")
            F.param_fence(self.stem)   # (the weight is read here, ahead of the layer's own fence)
            w7 = self.stem.weight
            wp = (w7.reshape(w7.shape[0], 7, 7 * IMAGE_CHANNELS)[:, :, self._pair_idx] * self._pair_mask.to(w7.dtype))
            return self.stem.forward_pool(x.view(B, H, W // 2, 8), (3, 3), (2, 2), (1, 1, 1, 1),
                                          weight=wp.view(w7.shape[0], 7, 4, 8), conv_stride=(2, 1),
                                          conv_padding=(3, 3, 2, 1))
        return self.stem.forward_pool(x, (3, 3), (2, 2), (1, 1, 1, 1))

    def forward(self, x):
        """x: [B, H, W, 8] (or, on the device, [B, H, W, 4] with W even: half the input bytes and 43 % fewer stem
        multiply-adds) -> logits [B, padded classes] (first ``num_classes`` columns are real)."""
        x = self._stem(x)
        tap = None
        for i, block in enumerate(self.blocks):
            nxt = cv.BnBackwardTap() if (x.is_cuda and i + 1 < len(self.blocks)) else None
            x = block(x, in_tap=tap, out_tap=nxt)
            tap = nxt
        x = F.global_avg_pool(x)
        return self.fc(x)
