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

    def forward(self, x):
        """x: [B, H, W, 8] -> logits [B, padded classes] (first ``num_classes`` columns are real)."""
        x = self.stem.forward_pool(x, (3, 3), (2, 2), (1, 1, 1, 1))
        tap = None
        for i, block in enumerate(self.blocks):
            nxt = cv.BnBackwardTap() if (x.is_cuda and i + 1 < len(self.blocks)) else None
            x = block(x, in_tap=tap, out_tap=nxt)
            tap = nxt
        x = F.global_avg_pool(x)
        return self.fc(x)
