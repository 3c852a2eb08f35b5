"""Loss modules of the experiment scripts, on HIP kernels.

``BCELoss``  -- ``torch.nn.BCELoss(reduction='mean')`` on probabilities (exp126a_musicnet_cnn_basic.py:87,323).
``PolyphonyLoss`` -- ``BCELoss(y_pred, y) + CrossEntropyLoss(n_pred, sum_pitch y) / 25`` (exp195f...py:95-96,331-334).
"""
import torch
import torch.nn as nn

from . import ops


class BCELoss(nn.Module):
    def __init__(self, reduction="mean"):
        super().__init__()
        if reduction != "mean":
            raise NotImplementedError("the reference only uses reduction='mean'")

    def forward(self, y_pred, y):
        return ops.BCELossFn.apply(y_pred, y)


class CrossEntropyLoss(nn.Module):
    def forward(self, logits, target, scale=1.0):
        return ops.CrossEntropyFn.apply(logits, target, scale)


class PolyphonyLoss(nn.Module):
    """loss1 + loss2 with loss2 = CE/25; n_target = sum over pitch bins of the labels (classes >= K raise upstream)."""

    def forward(self, y_pred, n_pred, y):
        n_target = torch.sum(y, dim=-1, keepdims=True).long().squeeze(3)      # label bookkeeping, as exp195f...py:331
        loss1 = ops.BCELossFn.apply(y_pred, y)
        loss2 = ops.CrossEntropyFn.apply(n_pred, n_target, 1.0 / 25.0)
        return ops.AddFn.apply(loss1.reshape(1), loss2.reshape(1)).reshape(())
