"""Host-side mirror of reference vision_mtl/losses.py (+ the CrossEntropyLoss the reference takes
from torch.nn at lit_module.py:31) on the HIP loss kernels."""
from __future__ import annotations

import typing as t

import torch
from torch import nn

from . import ops


class SILogLoss(nn.Module):
    """reference vision_mtl/losses.py:7-36.  pred / target: (B,H,W,1) (or any equal-numel pair).

    The reference first resizes ``pred`` to ``target.shape[-2:]`` with a bilinear interpolate;
    for the (B,H,W,1) tensors of this pipeline that is an identity, so the
    fused kernel requires equal shapes and raises otherwise rather than silently resampling."""

    def __init__(self, min_depth: float = 1e-3):
        super().__init__()
        self.min_depth = min_depth

    def forward(self, pred: torch.Tensor, target: torch.Tensor, mask: t.Optional[torch.Tensor] = None,
                interpolate: bool = True, min_depth: t.Optional[float] = None) -> torch.Tensor:
        if mask is not None:
            raise NotImplementedError("explicit masks are not used on the reference's step path")
        if pred.dim() < 2 or target.dim() < 2:
            raise IndexError("SILogLoss expects at least 2-D pred/target (the reference indexes shape[-2:])")
        if interpolate and pred.shape[-2:] != target.shape[-2:]:
            raise NotImplementedError("SILogLoss: pred/target spatial sizes differ; resample before the loss")
        if pred.shape != target.shape:
            raise ValueError(f"SILogLoss: shape mismatch {tuple(pred.shape)} vs {tuple(target.shape)}")
        return ops.silog(pred, target, self.min_depth if min_depth is None else min_depth)


class CrossEntropyLoss(nn.Module):
    """torch.nn.CrossEntropyLoss() with default arguments (mean, no weights / ignore_index / smoothing):
    logits (B,C,H,W), target int64 (B,H,W)."""

    def forward(self, logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        return ops.cross_entropy(logits, target)

    def forward_with_predictions(self, logits: torch.Tensor, target: torch.Tensor):
        """(loss, argmax_c logits): the loss pass finds each pixel's maximum anyway (one launch instead of two)."""
        return ops.cross_entropy_with_argmax(logits, target)


class L1Loss(nn.Module):
    """mean |pred - target| — the depth MAE metric of reference lit_module.py:68,112, usable as a loss."""

    def forward(self, pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        return ops.l1_loss(pred, target)
