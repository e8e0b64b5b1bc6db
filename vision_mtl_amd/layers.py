"""Execution helpers shared by the three model mirrors.

The models keep their parameters in ordinary ``nn.Conv2d`` / ``nn.BatchNorm2d`` /
``nn.ConvTranspose2d`` modules — used purely as parameter containers, so ``state_dict()`` keys,
shapes, default initialisation and optimizer behaviour are exactly the reference's — while
the arithmetic goes through the HIP kernels in :mod:`vision_mtl_amd.ops`.  ``Act`` is the
internal activation handle: an NHWC tensor with zero-padded channels plus its logical
channel count.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import nn

from . import ops


@dataclass
class Act:
    t: torch.Tensor  # [B, H, W, Cs] fp32, channels [C, Cs) are zero
    C: int

    @property
    def hw(self):
        return self.t.shape[1], self.t.shape[2]


def from_nchw(x: torch.Tensor) -> Act:
    if x.dim() != 4:
        raise ValueError(f"expected a (B, C, H, W) tensor, got shape {tuple(x.shape)}")
    return Act(ops.to_nhwc(x.float()), x.shape[1])


def to_nchw(a: Act) -> torch.Tensor:
    return ops.to_nchw(a.t, a.C)


def fork(a: Act, n: int = 2):
    """n handles on an activation with n consumers; their gradients are summed by this library's kernel (one launch)."""
    if n == 1:
        return (a,)
    if not a.t.requires_grad:
        return tuple(a for _ in range(n))
    return tuple(Act(t, a.C) for t in ops.fork(a.t, n))


def _momentum(bn: nn.BatchNorm2d) -> float:
    if bn.momentum is None:
        # torch switches to a cumulative moving average (factor 1/num_batches_tracked), which needs the counter on
        # the host; the reference never uses it (every BatchNorm2d is built with the default 0.1)
        raise NotImplementedError("BatchNorm2d(momentum=None) (cumulative moving average) is not implemented")
    return float(bn.momentum)


def _pair(v):
    return v[0] if isinstance(v, (tuple, list)) else v


def conv(x: Act, m: nn.Conv2d, stitch=None) -> Act:
    """Dense conv (+bias), no normalisation.  stitch = (CrossStitchLayer.weights, task): the stitch scale of x, folded
    into the conv's operands (ops.conv2d)."""
    if m.groups != 1:
        raise ValueError("conv(): use dwconv() for depthwise modules")
    y = ops.conv2d(x.t, m.weight, m.bias, _pair(m.stride), _pair(m.padding), stitch=stitch)
    return Act(y, m.out_channels)


def conv_bn_act(x, c: nn.Conv2d, bn: nn.BatchNorm2d, act: int, mul: Act | None = None,
                res: Act | None = None, stitch=None) -> Act:
    """act(BN(conv(x))) [* mul] [+ res] with the BatchNorm column sums taken from the conv epilogue.
    x: an Act, or a pair (xa, xb) standing for torch.cat((xa, xb), dim=1): a 1x1 conv then reads both maps directly
    (ops.conv1x1_cat), anything else gets the materialised concat."""
    train = bn.training
    if isinstance(x, tuple):
        xa, xb = x
        if (c.groups == 1 and c.kernel_size == (1, 1) and _pair(c.stride) == 1 and _pair(c.padding) == 0
                and xa.hw == xb.hw and ops.conv1x1_cat_supported(xa.t, xa.C, xb.t)):
            y, stats = ops.conv1x1_cat(xa.t, xb.t, xb.C, c.weight, c.bias, want_stats=train,
                                       zero_bias_grad=train and c.bias is not None)
            return bn_act(Act(y, c.out_channels), bn, act, mul, res, stats,
                          stats_rpb=getattr(stats, "_vmtl_rpb", 0) if stats is not None else 0)
        x = cat(xa, xb)
    if c.groups == 1:
        out = ops.conv2d(x.t, c.weight, c.bias, _pair(c.stride), _pair(c.padding), want_stats=train,
                         zero_bias_grad=train and c.bias is not None, stitch=stitch)
        y, stats = out if train else (out, None)
    else:
        if stitch is not None:
            raise ValueError("a stitch scale can only be folded into a dense conv")
        if c.groups != c.in_channels or c.in_channels != c.out_channels or c.bias is not None:
            raise ValueError("only depthwise grouped convs are supported")
        y, stats = ops.dwconv(x.t, c.weight, _pair(c.stride), _pair(c.padding)), None
    return bn_act(Act(y, c.out_channels), bn, act, mul, res, stats)


def conv_raw(x, c: nn.Conv2d, train: bool):
    """(raw output, BatchNorm partial rows or None, pixels per row) of a dense conv whose BatchNorm + activation the
    NEXT layer applies (pre-activation chains); x: an Act or a pair standing for their concat (1x1 convs only)."""
    zb = train and c.bias is not None
    if isinstance(x, tuple):
        xa, xb = x
        if (c.kernel_size == (1, 1) and _pair(c.stride) == 1 and _pair(c.padding) == 0 and xa.hw == xb.hw
                and ops.conv1x1_cat_supported(xa.t, xa.C, xb.t)):
            y, stats = ops.conv1x1_cat(xa.t, xb.t, xb.C, c.weight, c.bias, want_stats=train, zero_bias_grad=zb)
            return y, stats, getattr(stats, "_vmtl_rpb", 0) if stats is not None else 0
        x = cat(xa, xb)
    out = ops.conv2d(x.t, c.weight, c.bias, _pair(c.stride), _pair(c.padding), want_stats=train, zero_bias_grad=zb)
    y, stats = out if train else (out, None)
    return y, stats, getattr(stats, "_vmtl_rpb", 0) if stats is not None else 0


def bn_act(x: Act, bn: nn.BatchNorm2d, act: int, mul: Act | None = None, res: Act | None = None,
           stats=None, stats_rpb: int = 0) -> Act:
    momentum = _momentum(bn)
    y = ops.bn_act(x.t, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, x.C,
                   bn.training, momentum, bn.eps, act, mul=None if mul is None else mul.t,
                   res=None if res is None else res.t, stats=stats, stats_rpb=stats_rpb)
    return Act(y, x.C)


def conv_bn_act_maxpool2(x: Act, c: nn.Conv2d, bn: nn.BatchNorm2d, act: int) -> Act:
    """maxpool2(act(BN(conv(x)))) with the BatchNorm + activation + pool as ONE node (ops.bn_act_pool2): for a dense conv
    whose full-resolution activation nobody else reads."""
    if c.groups != 1:
        raise ValueError("conv_bn_act_maxpool2 expects a dense conv")
    train = bn.training
    out = ops.conv2d(x.t, c.weight, c.bias, _pair(c.stride), _pair(c.padding), want_stats=train,
                     zero_bias_grad=train and c.bias is not None)
    y, stats = out if train else (out, None)
    p = ops.bn_act_pool2(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, c.out_channels,
                         bn.training, _momentum(bn), bn.eps, act, stats=stats)
    return Act(p, c.out_channels)


def activation(x: Act, act: int) -> Act:
    return Act(ops.activation(x.t, act, x.C), x.C)


def dwconv(x: Act, m: nn.Conv2d) -> Act:
    return Act(ops.dwconv(x.t, m.weight, _pair(m.stride), _pair(m.padding)), m.out_channels)


def conv_transpose(x: Act, m: nn.ConvTranspose2d) -> Act:
    return Act(ops.conv_transpose2x2(x.t, m.weight, m.bias), m.out_channels)


def cat(a: Act, b: Act) -> Act:
    """torch.cat((a, b), dim=1) for equal spatial sizes."""
    if a.hw != b.hw:
        raise AssertionError(f"cat: spatial sizes differ: {a.hw} vs {b.hw}")
    return Act(ops.concat2(a.t, a.C, b.t, b.C), a.C + b.C)


def pad_cat(x1: Act, x2: Act) -> Act:
    """reference utils/model_utils.py:46-58: zero-pad x1 into x2's canvas, cat [x2, x1]."""
    (h1, w1), (h2, w2) = x1.hw, x2.hw
    dy, dx = h2 - h1, w2 - w1
    if dy < 0 or dx < 0:
        raise ValueError("pad_cat: x1 must not be larger than x2")
    return Act(ops.concat2(x2.t, x2.C, x1.t, x1.C, out_hw=(h2, w2), off_b=(dy // 2, dx // 2)), x1.C + x2.C)


def up2_cat(x: Act, skip: Act | None) -> Act:
    """smp DecoderBlock entry: nearest x2 upsample of x, then cat [x, skip]."""
    if skip is None:
        return Act(ops.concat2(x.t, x.C, up_a=2), x.C)
    return Act(ops.concat2(x.t, x.C, skip.t, skip.C, up_a=2), x.C + skip.C)


def up2_conv_bn_act(x: Act, skip: Act | None, c: nn.Conv2d, bn: nn.BatchNorm2d, act: int) -> Act:
    """act(BN(conv3x3(cat[nearest_x2(x), skip]))) - smp DecoderBlock entry - through the phase-decomposed
    kernel (no upsampled / concatenated tensor, 4 instead of 9 taps on the upsampled channels)."""
    if _pair(c.kernel_size) != 3 or _pair(c.padding) != 1 or _pair(c.stride) != 1 or c.bias is not None:
        raise ValueError("up2_conv_bn_act expects a 3x3 / pad 1 / stride 1 conv without bias")
    y, stats = ops.up2_conv(x.t, x.C, None if skip is None else skip.t, c.weight, want_stats=bn.training)
    return bn_act(Act(y, c.out_channels), bn, act, None, None, stats,
                  stats_rpb=getattr(stats, "_vmtl_rpb", 0) if stats is not None else 0)


def maxpool2(x: Act) -> Act:
    return Act(ops.maxpool2(x.t), x.C)


def bilinear_up2(x: Act) -> Act:
    return Act(ops.bilinear_up2(x.t), x.C)
