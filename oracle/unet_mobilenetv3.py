"""ORACLE (test infrastructure): CPU restatement, as pure functions of a state_dict, of the
third-party architecture the reference's `basic` and `csnet` models execute:
smp.Unet("timm-mobilenetv3_large_100", encoder_depth=5) + SegmentationHead(kernel_size=3)
(reference vision_mtl/utils/model_utils.py:25-34,118-132; models/basic_model.py:30-51), plus the
CSNet leaf walk (reference models/cross_stitch_model.py:102-157).

PARITY UNPINNED at this boundary: segmentation_models_pytorch==0.3.3 / timm==0.9.2 are absent
from /root/reference and cannot be installed offline, and the reference holds no test or fixture
for them.  The architecture below follows SURVEY.md Appendix A (cross-checked there by parameter
count: 5,483,032 = published MobileNetV3-Large).  What IS pinned: every in-repo rule that
constrains it (decoder channel rule, head kernel size, dict keys/order, stitch sites and channel
counts from get_stitch_channels, pad+concat order, nearest x2 only in the last decoder block).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

# (type, kernel, stride, mid_channels, out_channels, se_reduce (0 = none), act)
ARCH = [
    [("ds", 3, 1, 16, 16, 0, "relu")],
    [("ir", 3, 2, 64, 24, 0, "relu"), ("ir", 3, 1, 72, 24, 0, "relu")],
    [("ir", 5, 2, 72, 40, 24, "relu"), ("ir", 5, 1, 120, 40, 32, "relu"), ("ir", 5, 1, 120, 40, 32, "relu")],
    [("ir", 3, 2, 240, 80, 0, "hs"), ("ir", 3, 1, 200, 80, 0, "hs"), ("ir", 3, 1, 184, 80, 0, "hs"),
     ("ir", 3, 1, 184, 80, 0, "hs")],
    [("ir", 3, 1, 480, 112, 120, "hs"), ("ir", 3, 1, 672, 112, 168, "hs")],
    [("ir", 5, 2, 672, 160, 168, "hs"), ("ir", 5, 1, 960, 160, 240, "hs"), ("ir", 5, 1, 960, 160, 240, "hs")],
    [("cn", 1, 1, 0, 960, 0, "hs")],
]
ENC_CHANNELS = (3, 16, 24, 40, 112, 960)
_ACT = {"relu": F.relu, "hs": F.hardswish, None: lambda v: v}


class _Net:
    def __init__(self, sd, prefix, training):
        self.sd, self.p, self.training = sd, prefix, training

    def w(self, name):
        return self.sd[self.p + name]

    def conv(self, x, name, stride=1, pad=0, groups=1):
        return F.conv2d(x, self.w(name + ".weight"), self.sd.get(self.p + name + ".bias"), stride, pad, 1, groups)

    def bn(self, x, name):
        k = self.p + name
        if self.training and (k + ".num_batches_tracked") in self.sd:
            self.sd[k + ".num_batches_tracked"] += 1
        return F.batch_norm(x, self.sd[k + ".running_mean"], self.sd[k + ".running_var"], self.sd[k + ".weight"],
                            self.sd[k + ".bias"], self.training, 0.1, 1e-5)


def _se(n, x, name):
    s = x.mean((2, 3), keepdim=True)
    s = F.relu(n.conv(s, name + ".conv_reduce"))
    return x * F.hardsigmoid(n.conv(s, name + ".conv_expand"))


def _block(n, x, name, spec, c_in):
    kind, k, s, mid, out, se, act = spec
    a = _ACT[act]
    if kind == "ds":
        y = a(n.bn(n.conv(x, name + ".conv_dw", s, (k - 1) // 2, c_in), name + ".bn1"))
        y = n.bn(n.conv(y, name + ".conv_pw"), name + ".bn2")
    elif kind == "ir":
        y = a(n.bn(n.conv(x, name + ".conv_pw"), name + ".bn1"))
        y = a(n.bn(n.conv(y, name + ".conv_dw", s, (k - 1) // 2, mid), name + ".bn2"))
        if se:
            y = _se(n, y, name + ".se")
        y = n.bn(n.conv(y, name + ".conv_pwl"), name + ".bn3")
    else:
        return a(n.bn(n.conv(x, name + ".conv"), name + ".bn1"))
    return y + x if (s == 1 and c_in == out) else y


def encoder_features(n: _Net, x):
    """smp MobileNetV3Encoder.forward: [x, stem+stage0, stage1, stage2, stages3-4, stages5-6]."""
    feats = [x]
    y = F.hardswish(n.bn(n.conv(x, "conv_stem", 2, 1), "bn1"))
    c = 16
    for group in ([0], [1], [2], [3, 4], [5, 6]):
        for si in group:
            for bi, spec in enumerate(ARCH[si]):
                y = _block(n, y, f"blocks.{si}.{bi}", spec, c)
                c = spec[4]
        feats.append(y)
    return feats


def unet_decoder(n: _Net, feats, n_blocks=5):
    feats = feats[1:][::-1]
    x, skips = feats[0], feats[1:]
    for i in range(n_blocks):
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        if i < len(skips):
            x = torch.cat([x, skips[i]], 1)
        x = F.relu(n.bn(n.conv(x, f"blocks.{i}.conv1.0", 1, 1), f"blocks.{i}.conv1.1"))
        x = F.relu(n.bn(n.conv(x, f"blocks.{i}.conv2.0", 1, 1), f"blocks.{i}.conv2.1"))
    return x


def basic_forward(sd: dict, x: torch.Tensor, training: bool = True) -> dict:
    """reference models/basic_model.py:43-51 over state_dict keys backbone.{encoder.model,decoder}.*,
    {segm,depth}_head.0.*"""
    feats = encoder_features(_Net(sd, "backbone.encoder.model.", training), x)
    dec = unet_decoder(_Net(sd, "backbone.decoder.", training), feats)
    depth = F.conv2d(dec, sd["depth_head.0.weight"], sd["depth_head.0.bias"], padding=1)
    segm = F.conv2d(dec, sd["segm_head.0.weight"], sd["segm_head.0.bias"], padding=1)
    return dict(depth=depth, segm=segm)
