"""MobileNetV3-Large feature encoder + U-Net decoder + segmentation head on the HIP kernels.

The reference gets these from third-party packages that are not under /root/reference and not
installable offline:  segmentation_models_pytorch==0.3.3 (requirements.txt:14) -> timm==0.9.2,
instantiated at reference vision_mtl/utils/model_utils.py:25-34 (smp.Unet(encoder_name=
"timm-mobilenetv3_large_100", encoder_depth=5, decoder_channels=...)) and
models/basic_model.py:30-41 / utils/model_utils.py:125-130 (SegmentationHead(kernel_size=3)).
This file restates their published architecture with the same module
names, so state_dict keys / shapes match what smp+timm produce; results at this boundary are
"reference-unpinned" (no reference test or importable code pins them) and are checked against
the CPU oracle in oracle/unet_mobilenetv3.py instead.

Param-less children (act / drop / Identity modules) are kept on purpose: the reference's CSNet
(models/cross_stitch_model.py:102-157) walks named_modules() of exactly this tree.
"""
from __future__ import annotations

import math
import os
import typing as t

import torch
from torch import nn

from .. import layers as L
from ..ops import ACT_HSIGMOID, ACT_HSWISH, ACT_NONE, ACT_RELU

_ACT_CODE = {"relu": ACT_RELU, "hard_swish": ACT_HSWISH, None: ACT_NONE}


def _act_module(name):
    return {"relu": nn.ReLU, "hard_swish": nn.Hardswish, None: nn.Identity}[name]()


def make_divisible(v, divisor=8, min_value=None, round_limit=0.9):
    min_value = min_value or divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < round_limit * v:
        new_v += divisor
    return new_v


# timm `mobilenetv3_large_100` arch_def: (type, kernel, stride, expansion, out_channels, se, act)
ARCH = [
    [("ds", 3, 1, 1.0, 16, False, "relu")],
    [("ir", 3, 2, 4.0, 24, False, "relu"), ("ir", 3, 1, 3.0, 24, False, "relu")],
    [("ir", 5, 2, 3.0, 40, True, "relu"), ("ir", 5, 1, 3.0, 40, True, "relu"), ("ir", 5, 1, 3.0, 40, True, "relu")],
    [("ir", 3, 2, 6.0, 80, False, "hard_swish"), ("ir", 3, 1, 2.5, 80, False, "hard_swish"),
     ("ir", 3, 1, 2.3, 80, False, "hard_swish"), ("ir", 3, 1, 2.3, 80, False, "hard_swish")],
    [("ir", 3, 1, 6.0, 112, True, "hard_swish"), ("ir", 3, 1, 6.0, 112, True, "hard_swish")],
    [("ir", 5, 2, 6.0, 160, True, "hard_swish"), ("ir", 5, 1, 6.0, 160, True, "hard_swish"),
     ("ir", 5, 1, 6.0, 160, True, "hard_swish")],
    [("cn", 1, 1, 1.0, 960, False, "hard_swish")],
]
STEM = 16


class BatchNormAct2d(nn.BatchNorm2d):
    """timm BatchNormAct2d: BatchNorm2d parameters + child modules `drop` (Identity) and `act`."""

    def __init__(self, num_features, act=None):
        super().__init__(num_features)
        self.drop = nn.Identity()
        self.act = _act_module(act)
        self.act_code = _ACT_CODE[act]


ENC_CHAIN = os.environ.get("VMTL_ENC_CHAIN", "1") != "0"  # blocks hand their bn3 (+ skip) to the next expand conv


class Pending:
    """A block output whose closing BatchNorm (bn3 / bn2 of the depthwise-separable block: no activation) and skip
    connection have NOT been applied yet: the next layer's 1x1 conv applies them on its operand fragments
    (ops.bn_act_conv1x1 with a residual operand) and hands the materialised map back for its other consumers (that
    block's own skip branch, a decoder tap).  raw: the conv output, stats / rpb its BatchNorm partial rows."""

    def __init__(self, raw, stats, rpb, bn, C, res):
        self.raw, self.stats, self.rpb, self.bn, self.C, self.res = raw, stats, rpb, bn, C, res
        self.is_tap = False  # an encoder feature: whoever materialises it leaves a handle in .act
        self.act = None

    def materialize(self) -> L.Act:
        """Fallback: an ordinary BatchNorm (+ skip) pass."""
        y = L.bn_act(L.Act(self.raw, self.C), self.bn, ACT_NONE, res=self.res, stats=self.stats, stats_rpb=self.rpb)
        if self.is_tap:
            y, self.act = L.fork(y)
        return y

    def feed_conv1x1(self, conv: nn.Conv2d, want_stats: bool):
        """(raw output of conv, its stats, rpb, the materialised input map) with bn (+ skip) applied by conv itself."""
        out = L.ops.bn_act_conv1x1(self.raw, self.stats, self.rpb, self.bn, self.C, ACT_NONE, conv.weight, conv.bias,
                                   want_stats=want_stats, res=None if self.res is None else self.res.t, return_act=True)
        raw1, st1, rpb1, a = out
        xin = L.Act(a, self.C)
        if self.is_tap:
            xin, self.act = L.fork(xin)
        return raw1, st1, rpb1, xin

    def fusable_into(self, conv: nn.Conv2d) -> bool:
        return (ENC_CHAIN and conv.groups == 1 and conv.kernel_size == (1, 1) and L._pair(conv.stride) == 1
                and L._pair(conv.padding) == 0 and self.bn.momentum is not None
                and L.ops.bn_act_conv1x1_supported(self.raw, ACT_NONE))


def _close_block(y: L.Act, conv: nn.Conv2d, bn: nn.BatchNorm2d, res):
    """A block's closing 1x1 conv + BatchNorm (+ skip): left pending for the next layer when the chain is on."""
    if ENC_CHAIN and bn.momentum is not None:
        raw, st, rpb = L.conv_raw(y, conv, bn.training)
        return Pending(raw, st, rpb, bn, conv.out_channels, res)
    return L.conv_bn_act(y, conv, bn, ACT_NONE, res=res)


class SqueezeExcite(nn.Module):
    """timm SqueezeExcite(gate=hard_sigmoid, act=ReLU): x * hsigmoid(W_e relu(W_r mean_hw(x) + b_r) + b_e)."""

    def __init__(self, chs, rd_chs):
        super().__init__()
        self.conv_reduce = nn.Conv2d(chs, rd_chs, 1, bias=True)
        self.act1 = nn.ReLU(inplace=True)
        self.conv_expand = nn.Conv2d(rd_chs, chs, 1, bias=True)
        self.gate = nn.Hardsigmoid()

    def run(self, x: L.Act) -> L.Act:
        if x.t.shape[0] <= L.ops.squeeze_excite_max_batch():
            # fused path: per-image partial sums -> two batch-sized GEMMs -> scale
            y = L.ops.squeeze_excite(x.t, self.conv_reduce.weight, self.conv_reduce.bias, self.conv_expand.weight,
                                     self.conv_expand.bias, ACT_RELU, ACT_HSIGMOID)
            return L.Act(y, x.C)
        s = L.Act(L.ops.spatial_mean(x.t), x.C)
        s = L.activation(L.conv(s, self.conv_reduce), ACT_RELU)
        s = L.activation(L.conv(s, self.conv_expand), ACT_HSIGMOID)
        return L.Act(L.ops.channel_scale(x.t, s.t), x.C)


class DepthwiseSeparableConv(nn.Module):
    def __init__(self, in_chs, out_chs, k, stride, act):
        super().__init__()
        self.has_skip = stride == 1 and in_chs == out_chs
        self.conv_dw = nn.Conv2d(in_chs, in_chs, k, stride, (k - 1) // 2, groups=in_chs, bias=False)
        self.bn1 = BatchNormAct2d(in_chs, act)
        self.se = nn.Identity()
        self.conv_pw = nn.Conv2d(in_chs, out_chs, 1, bias=False)
        self.bn2 = BatchNormAct2d(out_chs, None)
        self.drop_path = nn.Identity()

    def run(self, x):
        if isinstance(x, Pending):
            x = x.materialize()
        x, res = L.fork(x) if self.has_skip else (x, None)
        y = L.conv_bn_act(x, self.conv_dw, self.bn1, self.bn1.act_code)
        return _close_block(y, self.conv_pw, self.bn2, res)

    def run_pre(self, raw, stats, rpb, bn_in, act_in: int) -> L.Act:
        """The block on a pre-activation input: raw = the producing conv's output (with its BatchNorm partial rows),
        bn_in / act_in that conv's BatchNorm + activation.  They run while the depthwise taps load (ops.bn_act_dwconv),
        whose epilogue also yields bn1's statistics; the activated input comes back for the residual branch."""
        dw = self.conv_dw
        raw1, st1, rpb1, a = L.ops.bn_act_dwconv(raw, stats, rpb, bn_in, dw.in_channels, act_in, dw.weight,
                                                  L._pair(dw.stride), L._pair(dw.padding),
                                                  want_stats=self.bn1.training, return_act=True)
        y = L.bn_act(L.Act(raw1, dw.out_channels), self.bn1, self.bn1.act_code, stats=st1, stats_rpb=rpb1)
        return _close_block(y, self.conv_pw, self.bn2, L.Act(a, dw.in_channels) if self.has_skip else None)


class InvertedResidual(nn.Module):
    def __init__(self, in_chs, out_chs, k, stride, exp, se, act):
        super().__init__()
        mid = make_divisible(in_chs * exp)
        self.has_skip = stride == 1 and in_chs == out_chs
        self.conv_pw = nn.Conv2d(in_chs, mid, 1, bias=False)
        self.bn1 = BatchNormAct2d(mid, act)
        self.conv_dw = nn.Conv2d(mid, mid, k, stride, (k - 1) // 2, groups=mid, bias=False)
        self.bn2 = BatchNormAct2d(mid, act)
        self.se = SqueezeExcite(mid, make_divisible(mid * 0.25)) if se else nn.Identity()
        self.conv_pwl = nn.Conv2d(mid, out_chs, 1, bias=False)
        self.bn3 = BatchNormAct2d(out_chs, None)
        self.drop_path = nn.Identity()

    def run(self, x):
        ops = L.ops
        fused = ops.FUSE_DW and self.bn1.act_code in (ACT_NONE, ACT_RELU, ACT_HSWISH)
        train = self.bn1.training
        if isinstance(x, Pending) and fused and x.fusable_into(self.conv_pw):
            # the previous block's bn3 (+ its skip) rides on this block's expand conv; the materialised map comes back
            # for this block's own skip branch
            raw1, st1, rpb1, xin = x.feed_conv1x1(self.conv_pw, train)
            res = xin if self.has_skip else None
        else:
            if isinstance(x, Pending):
                x = x.materialize()
            x, res = L.fork(x) if self.has_skip else (x, None)
            if fused:
                out = ops.conv2d(x.t, self.conv_pw.weight, None, 1, 0, want_stats=train)
                raw1, st1 = out if train else (out, None)
                rpb1 = getattr(st1, "_vmtl_rpb", 0) if st1 is not None else 0
        if fused:
            # [bn1 + act + conv_dw as one pre-activation node, bn2's statistics from its epilogue] -> bn2 + act
            dw = self.conv_dw
            raw2, st2, rpb2 = ops.bn_act_dwconv(raw1, st1, rpb1, self.bn1, dw.out_channels, self.bn1.act_code, dw.weight,
                                                L._pair(dw.stride), L._pair(dw.padding), want_stats=self.bn2.training)
            if (not isinstance(self.se, SqueezeExcite) and self.bn2.momentum is not None
                    and ops.bn_act_conv1x1_supported(raw2, self.bn2.act_code)):
                # no squeeze-excite between bn2 and conv_pwl: conv_pwl applies bn2 + act on its operand fragments and
                # its data gradient ends with their backward reduction
                raw3, st3, rpb3 = ops.bn_act_conv1x1(raw2, st2, rpb2, self.bn2, dw.out_channels, self.bn2.act_code,
                                                     self.conv_pwl.weight, None, want_stats=self.bn3.training)
                if ENC_CHAIN and self.bn3.momentum is not None:
                    return Pending(raw3, st3, rpb3, self.bn3, self.conv_pwl.out_channels, res)
                return L.bn_act(L.Act(raw3, self.conv_pwl.out_channels), self.bn3, ACT_NONE, res=res, stats=st3,
                                stats_rpb=rpb3)
            y = L.bn_act(L.Act(raw2, dw.out_channels), self.bn2, self.bn2.act_code, stats=st2, stats_rpb=rpb2)
        else:
            y = L.conv_bn_act(x, self.conv_pw, self.bn1, self.bn1.act_code)
            y = L.conv_bn_act(y, self.conv_dw, self.bn2, self.bn2.act_code)
        if isinstance(self.se, SqueezeExcite):
            y = self.se.run(y)
        return _close_block(y, self.conv_pwl, self.bn3, res)


class ConvBnAct(nn.Module):
    def __init__(self, in_chs, out_chs, k, act):
        super().__init__()
        self.conv = nn.Conv2d(in_chs, out_chs, k, 1, (k - 1) // 2, bias=False)
        self.bn1 = BatchNormAct2d(out_chs, act)
        self.drop_path = nn.Identity()

    def run(self, x):
        if isinstance(x, Pending):
            if x.fusable_into(self.conv):
                raw, st, rpb, _ = x.feed_conv1x1(self.conv, self.bn1.training)
                return L.bn_act(L.Act(raw, self.conv.out_channels), self.bn1, self.bn1.act_code, stats=st, stats_rpb=rpb)
            x = x.materialize()
        return L.conv_bn_act(x, self.conv, self.bn1, self.bn1.act_code)


class MobileNetV3Features(nn.Module):
    """timm.create_model("mobilenetv3_large_100", features_only=True): conv_stem, bn1, act1, blocks."""

    def __init__(self, in_chans=3):
        super().__init__()
        self.conv_stem = nn.Conv2d(in_chans, STEM, 3, 2, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(STEM)
        self.act1 = nn.Hardswish()
        stages, c = [], STEM
        for stage in ARCH:
            blocks = []
            for kind, k, s, e, out, se, act in stage:
                if kind == "ds":
                    blocks.append(DepthwiseSeparableConv(c, out, k, s, act))
                elif kind == "ir":
                    blocks.append(InvertedResidual(c, out, k, s, e, se, act))
                else:
                    blocks.append(ConvBnAct(c, out, k, act))
                c = out
            stages.append(nn.Sequential(*blocks))
        self.blocks = nn.Sequential(*stages)
        self._init_weights()

    def _init_weights(self):  # timm efficientnet_init_weights
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
                nn.init.normal_(m.weight, 0.0, math.sqrt(2.0 / fan_out))
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)


class MobileNetV3Encoder(nn.Module):
    """smp encoders/timm_mobilenetv3.py: stages = [Identity, stem+blocks[0], blocks[1], blocks[2],
    blocks[3:5], blocks[5:]] -> features with 3,16,24,40,112,960 channels at strides 1..32."""

    out_channels = (3, 16, 24, 40, 112, 960)

    def __init__(self, in_channels=3, depth=5):
        super().__init__()
        self._depth = depth
        self.model = MobileNetV3Features(in_channels)

    def run(self, x: L.Act) -> t.List[L.Act]:
        m = self.model
        feats = [x]
        first = m.blocks[0][0]
        fuse_stem = (L.ops.FUSE_DW and isinstance(first, DepthwiseSeparableConv) and m.bn1.momentum is not None
                     and first.conv_dw.kernel_size[0] in (3, 5) and os.environ.get("VMTL_FUSE_STEM", "1") != "0")
        if fuse_stem:
            # stem conv -> [stem BatchNorm + hardswish + the first block's depthwise conv as one pre-activation node]
            train = m.bn1.training
            out = L.ops.conv2d(x.t, m.conv_stem.weight, None, L._pair(m.conv_stem.stride), L._pair(m.conv_stem.padding),
                               want_stats=train)
            raw0, st0 = out if train else (out, None)
            rpb0 = getattr(st0, "_vmtl_rpb", 0) if st0 is not None else 0
            y = first.run_pre(raw0, st0, rpb0, m.bn1, ACT_HSWISH)
        else:
            y = L.conv_bn_act(x, m.conv_stem, m.bn1, ACT_HSWISH)
        groups = [[0], [1], [2], [3, 4], [5, 6]]
        for g in groups[: self._depth]:
            for si in g:
                for blk in m.blocks[si]:
                    if fuse_stem and blk is first:
                        continue
                    y = blk.run(y)
            if g is groups[self._depth - 1]:  # the deepest feature: nothing downstream in the encoder
                feats.append(y.materialize() if isinstance(y, Pending) else y)
            elif isinstance(y, Pending):  # goes to the decoder AND on: the next block materialises it and leaves a handle
                y.is_tap = True
                feats.append(y)
            else:
                y, tap = L.fork(y)
                feats.append(tap)
        return [f.act if isinstance(f, Pending) else f for f in feats]


class Conv2dReLU(nn.Sequential):
    def __init__(self, in_channels, out_channels):
        super().__init__(nn.Conv2d(in_channels, out_channels, 3, padding=1, bias=False), nn.BatchNorm2d(out_channels),
                         nn.ReLU(inplace=True))


class Attention(nn.Module):
    def __init__(self):
        super().__init__()
        self.attention = nn.Identity()


class DecoderBlock(nn.Module):
    """smp DecoderBlock: nearest x2, cat[x, skip], Conv2dReLU x 2."""

    def __init__(self, in_channels, skip_channels, out_channels):
        super().__init__()
        self.conv1 = Conv2dReLU(in_channels + skip_channels, out_channels)
        self.attention1 = Attention()
        self.conv2 = Conv2dReLU(out_channels, out_channels)
        self.attention2 = Attention()

    def run(self, x: L.Act, skip: t.Optional[L.Act]) -> L.Act:
        y = L.up2_conv_bn_act(x, skip, self.conv1[0], self.conv1[1], ACT_RELU)  # nearest x2 + cat + conv + BN + ReLU
        return L.conv_bn_act(y, self.conv2[0], self.conv2[1], ACT_RELU)

    def run_conv1_raw(self, x: L.Act, skip: t.Optional[L.Act]):
        """Only conv1 of the block, un-normalised: (raw output, its BatchNorm partial rows or None, pixels per row).
        The caller fuses conv1's BatchNorm + ReLU into what follows (ops.decoder_tail)."""
        c, bn = self.conv1[0], self.conv1[1]
        y, stats = L.ops.up2_conv(x.t, x.C, None if skip is None else skip.t, c.weight, want_stats=bn.training)
        rpb = 0
        if stats is not None:
            B, H2, W2, _ = x.t.shape
            rpb = L.ops.lib().raw("vmtl_conv2d_up2_stats_block")(B, H2, W2, y.shape[3])
        return L.Act(y, c.out_channels), stats, rpb


class UnetDecoder(nn.Module):
    def __init__(self, encoder_channels, decoder_channels):
        super().__init__()
        enc = list(encoder_channels[1:])[::-1]
        in_ch = [enc[0]] + list(decoder_channels[:-1])
        skip_ch = enc[1:] + [0]
        self.center = nn.Identity()
        self.blocks = nn.ModuleList([DecoderBlock(i, s, o) for i, s, o in zip(in_ch, skip_ch, decoder_channels)])
        for m in self.modules():  # smp initialize_decoder
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def run(self, feats: t.List[L.Act], raw_tail: bool = False):
        """The decoder as a chain of PRE-activation nodes: every conv hands its raw output (+ BatchNorm partial rows
        from its epilogue) to the next node, which owns that BatchNorm + ReLU together with its own conv
        (ops.bn_act_conv) so that the backward pass can fuse them.
        raw_tail: stop after conv1 of the LAST block and return (raw output, its statistics rows, pixels per row) -
        the caller continues with ops.decoder_tail."""
        feats = feats[1:][::-1]
        x, skips = feats[0], feats[1:]
        ops = L.ops
        raw = st = None
        rpb, bn, C = 0, None, 0
        for i, blk in enumerate(self.blocks):
            skip = skips[i] if i < len(skips) else None
            c1, bn1, c2, bn2 = blk.conv1[0], blk.conv1[1], blk.conv2[0], blk.conv2[1]
            skt = None if skip is None else skip.t
            if raw is None:  # first block: its input is an (already activated) encoder feature
                raw, st, rpb = blk.run_conv1_raw(x, skip)
                raw = raw.t
            else:
                raw, st, rpb = ops.bn_act_conv(raw, st, rpb, bn, C, ACT_RELU, c1.weight, skip=skt, up2=True,
                                               want_stats=bn1.training)
            bn, C = bn1, c1.out_channels
            if raw_tail and i == len(self.blocks) - 1:
                return L.Act(raw, C), st, rpb
            raw, st, rpb = ops.bn_act_conv(raw, st, rpb, bn, C, ACT_RELU, c2.weight, want_stats=bn2.training)
            bn, C = bn2, c2.out_channels
        return L.bn_act(L.Act(raw, C), bn, ACT_RELU, stats=st, stats_rpb=rpb)


class Activation(nn.Module):
    def __init__(self, name=None):
        super().__init__()
        if name not in (None, "identity"):
            raise NotImplementedError("only activation=None is used by the reference (basic_model.py:13)")
        self.activation = nn.Identity()


class SegmentationHead(nn.Sequential):
    """smp SegmentationHead(in, out, kernel_size=3, activation=None, upsampling=1)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, activation=None, upsampling=1):
        if upsampling != 1:
            raise NotImplementedError("upsampling > 1 is not used by the reference")
        conv = nn.Conv2d(in_channels, out_channels, kernel_size, padding=kernel_size // 2)
        nn.init.xavier_uniform_(conv.weight)  # smp initialize_head
        nn.init.constant_(conv.bias, 0)
        super().__init__(conv, nn.Identity(), Activation(activation))

    def run(self, x: L.Act) -> L.Act:
        return L.conv(x, self[0])


class Backbone(nn.Module):
    """reference vision_mtl/utils/model_utils.py:10-43."""

    def __init__(self, encoder_name: str = "timm-mobilenetv3_large_100", encoder_weights: t.Optional[str] = "imagenet",
                 decoder_first_channel: int = 256, num_decoder_layers: int = 5, in_channels: int = 3):
        super().__init__()
        if encoder_name != "timm-mobilenetv3_large_100":
            raise NotImplementedError(f"encoder {encoder_name!r}: only timm-mobilenetv3_large_100 is restated")
        weights_file = None
        if encoder_weights is not None:
            # the reference hands "imagenet" to smp, which downloads the timm checkpoint (a network fetch).  Offline the
            # same state_dict can be supplied as a local file: a path given directly, or $VMTL_ENCODER_WEIGHTS
            weights_file = encoder_weights if os.path.isfile(str(encoder_weights)) else os.environ.get("VMTL_ENCODER_WEIGHTS")
            if not weights_file or not os.path.isfile(weights_file):
                raise RuntimeError(
                    f"encoder_weights={encoder_weights!r} is a network download in the reference (smp -> timm).  Offline: "
                    "pass encoder_weights=None (random init; the reference CLI's default --backbone_weights), a path to a "
                    "local mobilenetv3_large_100 state_dict, or set VMTL_ENCODER_WEIGHTS to such a file")
        self.decoder_channels = [decoder_first_channel // (2 ** i) for i in range(num_decoder_layers)]
        self.encoder = MobileNetV3Encoder(in_channels, depth=num_decoder_layers)
        self.decoder = UnetDecoder(self.encoder.out_channels[: num_decoder_layers + 1], self.decoder_channels)
        if weights_file is not None:
            sd = torch.load(weights_file, map_location="cpu", weights_only=True)  # a state_dict: tensors only, no pickled code
            sd = sd.get("state_dict", sd) if isinstance(sd, dict) else sd
            own = self.encoder.model.state_dict()
            # timm's classifier checkpoint also carries conv_head / classifier: features_only drops them
            missing = [k for k in own if k not in sd]
            if missing:
                raise RuntimeError(f"{weights_file}: not a mobilenetv3_large_100 state_dict (missing {missing[:3]} ...)")
            self.encoder.model.load_state_dict({k: sd[k] for k in own})

    def run(self, x: L.Act, raw_tail: bool = False):
        return self.decoder.run(self.encoder.run(x), raw_tail=raw_tail)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return L.to_nchw(self.run(L.from_nchw(x)))


def get_model_with_dense_preds(segm_classes: int = 10, activation: t.Any = None,
                               backbone_params: t.Optional[dict] = None) -> nn.Module:
    """reference vision_mtl/utils/model_utils.py:118-132: Sequential(Backbone, SegmentationHead)."""
    backbone = Backbone(in_channels=3, **(backbone_params or {}))
    head = SegmentationHead(backbone.decoder_channels[-1], segm_classes, activation=activation, kernel_size=3)
    return nn.Sequential(backbone, head)
