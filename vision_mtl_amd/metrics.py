"""Per-step metrics of reference vision_mtl/lit_module.py:48-69,106-118 on the HIP kernels:
one confusion-matrix pass yields accuracy (micro), Jaccard (class mean) and F-beta (support
weighted); MAE is the L1 kernel.  Values stay on the device (no host sync on the step path)."""
from __future__ import annotations

import torch

from . import ops
from ._lib import lib


def _k(name, **kw):
    lib().callk(name, stream=torch.cuda.current_stream().cuda_stream, **kw)


def confusion_matrix(pred: torch.Tensor, target: torch.Tensor, num_classes: int) -> torch.Tensor:
    """cm[t, p] = #pixels with target t predicted as p; int32 (C, C) on the device."""
    if not pred.is_cuda:
        raise RuntimeError("confusion_matrix: predictions are not on the GPU (no CPU fallback)")
    pred, target = pred.contiguous(), target.contiguous()
    if pred.dtype != torch.int64 or target.dtype != torch.int64 or pred.numel() != target.numel():
        raise TypeError("confusion_matrix expects int64 predictions and targets of equal size")
    cm = torch.empty((num_classes, num_classes), dtype=torch.int32, device=pred.device)
    _k("vmtl_confusion_matrix", pred=pred, target=target, cm=cm, P=pred.numel(), C=num_classes)
    return cm


def _derived(cm: torch.Tensor, beta: float = 1.0) -> torch.Tensor:
    out = torch.empty((3,), dtype=torch.float32, device=cm.device)
    _k("vmtl_segm_metrics", cm=cm, C=cm.shape[0], beta=beta, out=out)
    return out


class _CMMetric:
    index = 0

    def __init__(self, num_classes: int, beta: float = 1.0):
        self.num_classes, self.beta = num_classes, beta

    def to(self, *a, **k):
        return self

    def from_confusion(self, cm: torch.Tensor) -> torch.Tensor:
        return _derived(cm, self.beta)[self.index]

    def __call__(self, preds: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        return self.from_confusion(confusion_matrix(preds, target, self.num_classes))


class Accuracy(_CMMetric):
    index = 0


class JaccardIndex(_CMMetric):
    index = 1


class FBetaScore(_CMMetric):
    index = 2


class MeanAbsoluteError:
    def to(self, *a, **k):
        return self

    def __call__(self, preds: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        return ops.l1_loss(preds.detach(), target)
