"""Small flower CNN: the trainable-end-to-end classifier used for the
single-process CPU configuration and the 1->8 GPU sync data-parallel scaling
configuration (5-class flower_photos-shaped input).

Five conv-bn-relu stages with max-pooling, global average pooling, then the
same two-layer head as the reference's Inception fine-tuning script
(1024-unit ReLU dense, dropout, 5 logits; reference head:
recognition_flower/single/flower_train.py:72-84) under the same variable scope
``fc_trainable`` so head checkpoints are interchangeable.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..ops import functional as F
from .layers import ConvBnAct, Dense

IMAGE_CHANNELS = 8


class FlowerHead(nn.Module):
    """dense(1024, relu) -> dropout(rate) -> dense(n_outputs) with the optional
    L2 *activity* regulariser of the reference (scale 0.0005 on both dense
    outputs; recognition_flower/single/flower_train.py:71,77,84)."""

    def __init__(self, n_in, n_outputs=5, hidden=1024, dropout_rate=0.5, act_l2=0.0005):
        super().__init__()
        self.n_outputs = n_outputs
        self.dropout_rate = dropout_rate
        self.act_l2 = act_l2
        self.dense_prelogit1 = Dense(n_in, hidden, relu=True, tf_scope="fc_trainable/dense_prelogit1")
        self.flower_logits = Dense(hidden, n_outputs, tf_scope="fc_trainable/flower_logits")

    def forward(self, feats):
        h = self.dense_prelogit1(feats)
        reg = None
        if self.act_l2:
            h, reg = F.activity_l2(h, self.act_l2)
        h = F.dropout(h, self.dropout_rate, self.training)
        logits = self.flower_logits(h)
        return logits, reg


class FlowerCNN(nn.Module):
    def __init__(self, num_classes=5, widths=(32, 64, 128, 256, 512), dropout_rate=0.5, act_l2=0.0005):
        super().__init__()
        self.num_classes = num_classes
        w = widths
        self.conv1 = ConvBnAct(3, w[0], 3, 2, "SAME", tf_scope="FlowerCNN/conv1", cin_pad=IMAGE_CHANNELS)
        self.conv2 = ConvBnAct(w[0], w[1], 3, 1, "SAME", tf_scope="FlowerCNN/conv2")
        self.conv3 = ConvBnAct(w[1], w[2], 3, 1, "SAME", tf_scope="FlowerCNN/conv3")
        self.conv4 = ConvBnAct(w[2], w[3], 3, 1, "SAME", tf_scope="FlowerCNN/conv4")
        self.conv5 = ConvBnAct(w[3], w[4], 3, 1, "SAME", tf_scope="FlowerCNN/conv5")
        self.head = FlowerHead(w[4], num_classes, dropout_rate=dropout_rate, act_l2=act_l2)

    def features(self, x):
        x = self.conv1(x)
        x = self.conv2.forward_pool(x, (3, 3), (2, 2), "SAME")
        x = self.conv3.forward_pool(x, (3, 3), (2, 2), "SAME")
        x = self.conv4.forward_pool(x, (3, 3), (2, 2), "SAME")
        x = self.conv5(x)
        return F.global_avg_pool(x)

    def forward(self, x):
        """Returns ``(logits [B, padded], activity_regulariser or None)``."""
        return self.head(self.features(x))
