"""Host-side mirror of reference vision_mtl/utils/model_utils.py: same public names and
constructor arguments, HIP arithmetic.  (Backbone and get_model_with_dense_preds live in
vision_mtl_amd/models/unet_mobilenetv3.py because they restate third-party smp/timm code.)"""
from __future__ import annotations

import typing as t

import torch
from torch import nn

from .. import layers as L
from ..ops import ACT_RELU


def concat_slightly_diff_sized_tensors(x1, x2):
    """reference utils/model_utils.py:46-58.  Accepts internal activation handles or NCHW tensors."""
    if isinstance(x1, L.Act):
        return L.pad_cat(x1, x2)
    return L.to_nchw(L.pad_cat(L.from_nchw(x1), L.from_nchw(x2)))


class DoubleConv(nn.Module):
    """(conv3x3 -> BN -> ReLU) x 2, reference utils/model_utils.py:61-80.  ``double_conv`` keeps the
    reference's Sequential indices (0,1,3,4) so checkpoints interchange."""

    def __init__(self, in_channels: int, out_channels: int, mid_channels: t.Optional[int] = None):
        super().__init__()
        mid = mid_channels or out_channels
        self.double_conv = nn.Sequential(
            nn.Conv2d(in_channels, mid, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(mid),
            nn.ReLU(inplace=True),
            nn.Conv2d(mid, out_channels, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True),
        )

    def run(self, x: L.Act) -> L.Act:
        s = self.double_conv
        ops = L.ops
        # conv -> [BN + ReLU + conv as ONE pre-activation node: the second conv's data gradient ends with the first
        # BatchNorm's backward reduction] -> BN + ReLU
        out = ops.conv2d(x.t, s[0].weight, None, 1, 1, want_stats=s[1].training)
        y1, st1 = out if s[1].training else (out, None)
        rpb1 = getattr(st1, "_vmtl_rpb", 0) if st1 is not None else 0
        y2, st2, rpb2 = ops.bn_act_conv(y1, st1, rpb1, s[1], s[0].out_channels, ACT_RELU, s[3].weight,
                                        want_stats=s[4].training)
        return L.bn_act(L.Act(y2, s[3].out_channels), s[4], ACT_RELU, stats=st2, stats_rpb=rpb2)

    def forward(self, x):
        if isinstance(x, L.Act):
            return self.run(x)
        return L.to_nchw(self.run(L.from_nchw(x)))


def get_module_by_name(module: nn.Module, access_string: str) -> nn.Module:
    """reference utils/utils.py:52-58."""
    out = module
    for name in access_string.split("."):
        out = getattr(out, name)
    return out
