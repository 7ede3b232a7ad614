"""Inception-v3 trunk with TF-slim variable scopes + the reference's flower head.

The reference fine-tunes a new 2048->1024->5 head on top of
``slim.nets.inception.inception_v3(X, num_classes=1001)`` and reads the
``PreLogits`` endpoint (recognition_flower/single/flower_train.py:66-85,
recognition_flower/distributed/train_dataset.py:36-56).  The network body is
TF-slim's; this file is a native NHWC definition with the same topology,
endpoint shapes and variable names so that checkpoints carry the keys
``InceptionV3/<scope>/weights`` and ``.../BatchNorm/{beta,moving_mean,
moving_variance}`` (slim's arg scope: no conv bias, batch-norm without gamma,
decay 0.9997, epsilon 0.001, ReLU).

``AuxLogits`` and ``Logits/Conv2d_1c_1x1`` exist as variables only (the
reference's loss never touches them, so TensorFlow prunes their ops); they are
created here for checkpoint-key parity and are never executed.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..ops import functional as F
from .flower_cnn import FlowerHead
from .layers import Conv2d, ConvBnAct

IMAGE_CHANNELS = 8
ROOT = "InceptionV3"
_BN = dict(bn_scale=False, bn_eps=1e-3, bn_momentum=1.0 - 0.9997)


def _c(cin, cout, k, scope, stride=1, padding="SAME", **kw):
    return ConvBnAct(cin, cout, k, stride, padding, tf_scope=scope, **_BN, **kw)


def _concat(block: nn.Module, x, hw, parts):
    """Channel concatenation of parallel branches.

    ``parts`` is a list of ``(channels, producer)``; ``producer(out)`` computes one branch and, when
    ``out`` (a channel slice of the result) is given, has its last layer write into it.  With nothing
    requiring grad (the frozen trunk of the flower model) the branches write straight into one
    buffer -- side by side on separate HIP streams when the batch is small; otherwise the outputs are joined with ``torch.cat`` so autograd sees ordinary tensors."""
    if not F.concat_plan(x, *block.parameters()):
        return torch.cat([prod(None) for _, prod in parts], dim=3)
    buf = torch.empty((x.shape[0], hw[0], hw[1], sum(c for c, _ in parts)), dtype=x.dtype, device=x.device)

    def job(sl, prod):
        def run():
            r = prod(sl)
            if r.data_ptr() != sl.data_ptr():  # producer without a fused writer (pooling branch)
                sl.copy_(r)
        return run

    jobs, off = [], 0
    for c, prod in parts:
        jobs.append(job(buf[..., off:off + c], prod))
        off += c
    # the branches only share their input; at small batches they run on separate streams (F.run_branches)
    F.run_branches(x, jobs)
    return buf


def _reduced(x):
    """Spatial size after a 3x3 / stride 2 / VALID layer."""
    return (x.shape[1] - 3) // 2 + 1, (x.shape[2] - 3) // 2 + 1


class _Mixed5(nn.Module):
    """35x35 blocks Mixed_5b/5c/5d."""

    def __init__(self, cin, pool_ch, scope, b1_names=("Conv2d_0a_1x1", "Conv2d_0b_5x5")):
        super().__init__()
        s = scope
        self.b0 = _c(cin, 64, 1, s + "/Branch_0/Conv2d_0a_1x1")
        self.b1a = _c(cin, 48, 1, s + "/Branch_1/" + b1_names[0])
        self.b1b = _c(48, 64, 5, s + "/Branch_1/" + b1_names[1])
        self.b2a = _c(cin, 64, 1, s + "/Branch_2/Conv2d_0a_1x1")
        self.b2b = _c(64, 96, 3, s + "/Branch_2/Conv2d_0b_3x3")
        self.b2c = _c(96, 96, 3, s + "/Branch_2/Conv2d_0c_3x3")
        self.b3 = _c(cin, pool_ch, 1, s + "/Branch_3/Conv2d_0b_1x1")

    def forward(self, x):
        return _concat(self, x, x.shape[1:3], [
            (64, lambda o: self.b0(x, out=o)),
            (64, lambda o: self.b1b(self.b1a(x), out=o)),
            (96, lambda o: self.b2c(self.b2b(self.b2a(x)), out=o)),
            (self.b3.weight.shape[0], lambda o: self.b3(F.avg_pool(x, (3, 3), (1, 1), "SAME"), out=o)),
        ])


class _Mixed6a(nn.Module):
    def __init__(self, cin, scope):
        super().__init__()
        s = scope
        self.b0 = _c(cin, 384, 3, s + "/Branch_0/Conv2d_1a_1x1", stride=2, padding="VALID")
        self.b1a = _c(cin, 64, 1, s + "/Branch_1/Conv2d_0a_1x1")
        self.b1b = _c(64, 96, 3, s + "/Branch_1/Conv2d_0b_3x3")
        self.b1c = _c(96, 96, 3, s + "/Branch_1/Conv2d_1a_1x1", stride=2, padding="VALID")

    def forward(self, x):
        return _concat(self, x, _reduced(x), [
            (384, lambda o: self.b0(x, out=o)),
            (96, lambda o: self.b1c(self.b1b(self.b1a(x)), out=o)),
            (x.shape[3], lambda o: F.max_pool(x, (3, 3), (2, 2), "VALID")),
        ])


class _Mixed6(nn.Module):
    """17x17 blocks Mixed_6b..6e with factorised 7x7 convolutions."""

    def __init__(self, cin, c, scope):
        super().__init__()
        s = scope
        self.b0 = _c(cin, 192, 1, s + "/Branch_0/Conv2d_0a_1x1")
        self.b1a = _c(cin, c, 1, s + "/Branch_1/Conv2d_0a_1x1")
        self.b1b = _c(c, c, (1, 7), s + "/Branch_1/Conv2d_0b_1x7")
        self.b1c = _c(c, 192, (7, 1), s + "/Branch_1/Conv2d_0c_7x1")
        self.b2a = _c(cin, c, 1, s + "/Branch_2/Conv2d_0a_1x1")
        self.b2b = _c(c, c, (7, 1), s + "/Branch_2/Conv2d_0b_7x1")
        self.b2c = _c(c, c, (1, 7), s + "/Branch_2/Conv2d_0c_1x7")
        self.b2d = _c(c, c, (7, 1), s + "/Branch_2/Conv2d_0d_7x1")
        self.b2e = _c(c, 192, (1, 7), s + "/Branch_2/Conv2d_0e_1x7")
        self.b3 = _c(cin, 192, 1, s + "/Branch_3/Conv2d_0b_1x1")

    def forward(self, x):
        return _concat(self, x, x.shape[1:3], [
            (192, lambda o: self.b0(x, out=o)),
            (192, lambda o: self.b1c(self.b1b(self.b1a(x)), out=o)),
            (192, lambda o: self.b2e(self.b2d(self.b2c(self.b2b(self.b2a(x)))), out=o)),
            (192, lambda o: self.b3(F.avg_pool(x, (3, 3), (1, 1), "SAME"), out=o)),
        ])


class _Mixed7a(nn.Module):
    def __init__(self, cin, scope):
        super().__init__()
        s = scope
        self.b0a = _c(cin, 192, 1, s + "/Branch_0/Conv2d_0a_1x1")
        self.b0b = _c(192, 320, 3, s + "/Branch_0/Conv2d_1a_3x3", stride=2, padding="VALID")
        self.b1a = _c(cin, 192, 1, s + "/Branch_1/Conv2d_0a_1x1")
        self.b1b = _c(192, 192, (1, 7), s + "/Branch_1/Conv2d_0b_1x7")
        self.b1c = _c(192, 192, (7, 1), s + "/Branch_1/Conv2d_0c_7x1")
        self.b1d = _c(192, 192, 3, s + "/Branch_1/Conv2d_1a_3x3", stride=2, padding="VALID")

    def forward(self, x):
        return _concat(self, x, _reduced(x), [
            (320, lambda o: self.b0b(self.b0a(x), out=o)),
            (192, lambda o: self.b1d(self.b1c(self.b1b(self.b1a(x))), out=o)),
            (x.shape[3], lambda o: F.max_pool(x, (3, 3), (2, 2), "VALID")),
        ])


class _Mixed7(nn.Module):
    """8x8 blocks Mixed_7b / Mixed_7c (the second 3x1 of Branch_1 is named
    ``Conv2d_0b_3x1`` in 7b and ``Conv2d_0c_3x1`` in 7c upstream)."""

    def __init__(self, cin, scope, b1_3x1_name):
        super().__init__()
        s = scope
        self.b0 = _c(cin, 320, 1, s + "/Branch_0/Conv2d_0a_1x1")
        self.b1a = _c(cin, 384, 1, s + "/Branch_1/Conv2d_0a_1x1")
        self.b1b = _c(384, 384, (1, 3), s + "/Branch_1/Conv2d_0b_1x3")
        self.b1c = _c(384, 384, (3, 1), s + "/Branch_1/" + b1_3x1_name)
        self.b2a = _c(cin, 448, 1, s + "/Branch_2/Conv2d_0a_1x1")
        self.b2b = _c(448, 384, 3, s + "/Branch_2/Conv2d_0b_3x3")
        self.b2c = _c(384, 384, (1, 3), s + "/Branch_2/Conv2d_0c_1x3")
        self.b2d = _c(384, 384, (3, 1), s + "/Branch_2/Conv2d_0d_3x1")
        self.b3 = _c(cin, 192, 1, s + "/Branch_3/Conv2d_0b_1x1")

    def forward(self, x):
        b1 = self.b1a(x)
        b2 = self.b2b(self.b2a(x))
        return _concat(self, x, x.shape[1:3], [
            (320, lambda o: self.b0(x, out=o)),
            (384, lambda o: self.b1b(b1, out=o)),
            (384, lambda o: self.b1c(b1, out=o)),
            (384, lambda o: self.b2c(b2, out=o)),
            (384, lambda o: self.b2d(b2, out=o)),
            (192, lambda o: self.b3(F.avg_pool(x, (3, 3), (1, 1), "SAME"), out=o)),
        ])


class _AuxLogitsVars(nn.Module):
    """Variables of the auxiliary classifier (never executed; checkpoint parity)."""

    def __init__(self, num_classes=1001):
        super().__init__()
        s = ROOT + "/AuxLogits"
        self.conv_1b = _c(768, 128, 1, s + "/Conv2d_1b_1x1")
        self.conv_2a = _c(128, 768, 5, s + "/Conv2d_2a_5x5", padding="VALID")
        self.conv_2b = Conv2d(768, num_classes, 1, tf_scope=s + "/Conv2d_2b_1x1")


class InceptionV3(nn.Module):
    """Trunk up to the ``PreLogits`` endpoint ([B, 2048])."""

    def __init__(self, num_classes=1001, prelogits_dropout_rate=0.2, create_unused_vars=True):
        super().__init__()
        r = ROOT
        self.Conv2d_1a_3x3 = _c(3, 32, 3, r + "/Conv2d_1a_3x3", stride=2, padding="VALID", cin_pad=IMAGE_CHANNELS)
        self.Conv2d_2a_3x3 = _c(32, 32, 3, r + "/Conv2d_2a_3x3", padding="VALID")
        self.Conv2d_2b_3x3 = _c(32, 64, 3, r + "/Conv2d_2b_3x3")
        self.Conv2d_3b_1x1 = _c(64, 80, 1, r + "/Conv2d_3b_1x1", padding="VALID")
        self.Conv2d_4a_3x3 = _c(80, 192, 3, r + "/Conv2d_4a_3x3", padding="VALID")
        self.Mixed_5b = _Mixed5(192, 32, r + "/Mixed_5b")
        self.Mixed_5c = _Mixed5(256, 64, r + "/Mixed_5c", b1_names=("Conv2d_0b_1x1", "Conv_1_0c_5x5"))
        self.Mixed_5d = _Mixed5(288, 64, r + "/Mixed_5d")
        self.Mixed_6a = _Mixed6a(288, r + "/Mixed_6a")
        self.Mixed_6b = _Mixed6(768, 128, r + "/Mixed_6b")
        self.Mixed_6c = _Mixed6(768, 160, r + "/Mixed_6c")
        self.Mixed_6d = _Mixed6(768, 160, r + "/Mixed_6d")
        self.Mixed_6e = _Mixed6(768, 192, r + "/Mixed_6e")
        self.Mixed_7a = _Mixed7a(768, r + "/Mixed_7a")
        self.Mixed_7b = _Mixed7(1280, r + "/Mixed_7b", "Conv2d_0b_3x1")
        self.Mixed_7c = _Mixed7(2048, r + "/Mixed_7c", "Conv2d_0c_3x1")
        self.prelogits_dropout_rate = prelogits_dropout_rate
        if create_unused_vars:
            self.AuxLogits = _AuxLogitsVars(num_classes)
            self.Logits_Conv2d_1c_1x1 = Conv2d(2048, num_classes, 1, tf_scope=r + "/Logits/Conv2d_1c_1x1")

    def endpoints(self, x, fuse_pools=False):
        """Yields (name, tensor) for the main-path endpoints (shape walk / tests).  With ``fuse_pools`` the
        two stem max-pools are computed together with the layer before them, whose own (un-pooled)
        endpoint is then not produced."""
        for name in ("Conv2d_1a_3x3", "Conv2d_2a_3x3"):
            x = getattr(self, name)(x)
            yield name, x
        if fuse_pools:
            x = self.Conv2d_2b_3x3.forward_pool(x, (3, 3), (2, 2), "VALID")
        else:
            x = self.Conv2d_2b_3x3(x)
            yield "Conv2d_2b_3x3", x
            x = F.max_pool(x, (3, 3), (2, 2), "VALID")
        yield "MaxPool_3a_3x3", x
        x = self.Conv2d_3b_1x1(x)
        yield "Conv2d_3b_1x1", x
        if fuse_pools:
            x = self.Conv2d_4a_3x3.forward_pool(x, (3, 3), (2, 2), "VALID")
        else:
            x = self.Conv2d_4a_3x3(x)
            yield "Conv2d_4a_3x3", x
            x = F.max_pool(x, (3, 3), (2, 2), "VALID")
        yield "MaxPool_5a_3x3", x
        for name in ("Mixed_5b", "Mixed_5c", "Mixed_5d", "Mixed_6a", "Mixed_6b", "Mixed_6c", "Mixed_6d", "Mixed_6e",
                     "Mixed_7a", "Mixed_7b", "Mixed_7c"):
            x = getattr(self, name)(x)
            yield name, x
        x = F.global_avg_pool(x)
        x = F.dropout(x, self.prelogits_dropout_rate, self.training)
        yield "PreLogits", x

    def forward(self, x):
        out = None
        # (forward-only batch-statistics layers of few pixels keep their statistics in accumulators: F.stat_scope)
        with F.stat_scope(x.device):
            for _, out in self.endpoints(x, fuse_pools=True):
                pass
        return out


class InceptionV3Flower(nn.Module):
    """Frozen Inception-v3 trunk + trainable flower head (the reference model).

    Only ``fc_trainable/*`` is trained (reference: ``var_list=fc_vars``,
    recognition_flower/single/flower_train.py:141-152), so the trunk runs
    without autograd bookkeeping; its batch-norm layers still follow the
    module's train/eval mode like the reference's ``is_training`` flag.

    ``trunk_bn="moving"`` keeps the trunk's batch-norm layers on their moving statistics while training
    (the usual frozen-feature-extractor setting: the reference never updates those statistics anyway, so
    its own evaluation pass, ``training=False`` at recognition_flower/single/flower_train.py:188-189, sees
    features normalised differently from the ones the head was trained on).  Every conv + batch-norm +
    ReLU of the trunk is then one kernel, with the statistics folded into the convolution epilogue;
    ``PreLogits`` dropout still follows the train/eval mode.

    ``train_trunk=True`` (model name ``inception_v3_finetune``) is the step the reference stops short of: every
    convolution and batch-norm offset of the trunk is trained together with the head (the reference's
    ``var_list=fc_vars`` dropped), so the step runs the data- and weight-gradient kernels of all 94 trunk layers,
    the pooling backward passes and autograd's channel concatenation of the block branches.  The variables of
    ``AuxLogits`` / ``Logits``, which no loss touches, stay frozen.
    """

    def __init__(self, num_classes=5, head_dropout_rate=0.5, act_l2=0.0005, trunk_bn="batch", train_trunk=False):
        super().__init__()
        if trunk_bn not in ("batch", "moving"):
            raise ValueError("trunk_bn must be 'batch' or 'moving'")
        self.trunk_bn = trunk_bn
        self.train_trunk = bool(train_trunk)
        self.num_classes = num_classes
        self.trunk = InceptionV3()
        unused = {id(p) for name in ("AuxLogits", "Logits_Conv2d_1c_1x1") if hasattr(self.trunk, name)
                  for p in getattr(self.trunk, name).parameters()}
        for p in self.trunk.parameters():
            p.requires_grad_(self.train_trunk and id(p) not in unused)
        self.head = FlowerHead(2048, num_classes, dropout_rate=head_dropout_rate, act_l2=act_l2)

    def train(self, mode: bool = True):
        super().train(mode)
        if self.trunk_bn == "moving":
            for m in self.trunk.modules():
                if isinstance(m, ConvBnAct):
                    m.training = False
        return self

    def forward(self, x):
        if self.train_trunk:
            return self.head(self.trunk(x))
        with torch.no_grad():
            feats = self.trunk(x)
        return self.head(feats)
