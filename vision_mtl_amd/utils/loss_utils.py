"""Host-side mirror of reference vision_mtl/utils/loss_utils.py."""
from __future__ import annotations

import numbers
import typing as t

import torch

from .. import ops


def calc_loss(out: dict, gt_mask, gt_depth, segm_criterion, depth_criterion) -> torch.Tensor:
    """reference utils/loss_utils.py:8-24: CE(segm) + SILog(sigmoid(depth) as (B,H,W,1)), unweighted."""
    loss_segm = segm_criterion(out["segm"], gt_mask)
    depth_predictions = ops.sigmoid(out["depth"]).permute(0, 2, 3, 1)
    return loss_segm + depth_criterion(depth_predictions, gt_depth)


def summarize_epoch_metrics(step_results: dict, metric_name_prefix: t.Optional[str] = None) -> dict:
    """reference utils/loss_utils.py:27-44: mean of every per-step list, then clear the lists."""
    prefix = "" if metric_name_prefix is None else metric_name_prefix + "/"
    metrics = {}
    for k, vals in step_results.items():
        metrics[f"{prefix}{k}"] = torch.mean(torch.tensor([float(v) for v in vals])).item()
    for k in step_results:
        step_results[k].clear()
    return metrics


def print_metrics(prefix: str, epoch_metrics: dict) -> str:
    """reference utils/loss_utils.py:47-64."""
    s = ""
    for k, v in epoch_metrics.items():
        if isinstance(v, torch.Tensor):
            value = v[-1] if v.numel() > 1 else v.item()
        else:
            value = v if isinstance(v, numbers.Number) else v[-1]
        print(f"{prefix}/{k}: {float(value):.3f} ")
        s += f"{k}: {float(value):.3f} "
    return s
