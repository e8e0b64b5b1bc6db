"""ORACLE (test infrastructure): CPU restatement of the reference's CSNet forward
(vision_mtl/models/cross_stitch_model.py:102-157) applied to two smp-Unet(MobileNetV3)+head
networks, written as straight-line code over a state_dict (no module tree, no name walk), so it
cross-checks the mirror's walk-based implementation rather than sharing its logic.

Semantics restated:
  * only LEAF modules run: inside timm blocks the BatchNormAct2d modules have children (drop, act),
    so their normalisation is skipped and only the activation runs; no residual adds; SqueezeExcite
    degenerates to conv_reduce -> ReLU -> conv_expand -> hard_sigmoid applied to the full map;
  * the stem BatchNorm and the decoder BatchNorms are plain leaves and do run;
  * skips = stage inputs 1,2,3,5 (clones); decoder blocks 0-3 zero-pad x into the skip canvas and
    concat [skip, x] (utils/model_utils.py:46-58); block 4 does nearest x2;
  * stitch (diagonal scale w[a,a,(c)]) at the entry of encoder stages 1-6 and decoder blocks 0-4,
    after the skip save / merge.
PARITY UNPINNED for the network part (smp/timm absent); the stitch arithmetic itself is pinned by
tests/golden/components.pt (reference CrossStitchLayer).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .mtan import pad_concat
from .unet_mobilenetv3 import ARCH, _ACT, _Net


def _stitch(sd, name, feats, tasks):
    w = sd[f"cross_stitch_layers.{name}.weights"]
    out = {}
    for a, t in enumerate(tasks):
        d = w[a, a]
        out[t] = feats[t] * (d[None, :, None, None] if w.dim() == 3 else d)
    return out


def _leaf_block(n: _Net, x, name, spec, c_in):
    kind, k, s, mid, out, se, act = spec
    a = _ACT[act]
    if kind == "ds":
        return n.conv(a(n.conv(x, name + ".conv_dw", s, (k - 1) // 2, c_in)), name + ".conv_pw")
    if kind == "cn":
        return a(n.conv(x, name + ".conv"))
    y = a(n.conv(x, name + ".conv_pw"))
    y = a(n.conv(y, name + ".conv_dw", s, (k - 1) // 2, mid))
    if se:
        y = F.hardsigmoid(n.conv(F.relu(n.conv(y, name + ".se.conv_reduce")), name + ".se.conv_expand"))
    return n.conv(y, name + ".conv_pwl")


def csnet_forward(sd: dict, x: torch.Tensor, tasks: list, training: bool = True, debug=None) -> dict:
    enc = {t: _Net(sd, f"models.{t}.0.encoder.model.", training) for t in tasks}
    dec = {t: _Net(sd, f"models.{t}.0.decoder.", training) for t in tasks}
    f = {t: F.hardswish(enc[t].bn(enc[t].conv(x.clone(), "conv_stem", 2, 1), "bn1")) for t in tasks}
    skips = {t: [] for t in tasks}
    c = 16
    for si, stage in enumerate(ARCH):
        if si != 0:
            if si not in (4, 6):  # idx != num_decoder_layers-1 and != num_encoder_layers-1
                for t in tasks:
                    skips[t].append(f[t].clone())
            f = _stitch(sd, f"0_encoder_model_blocks_{si}", f, tasks)
        for bi, spec in enumerate(stage):
            for t in tasks:
                f[t] = _leaf_block(enc[t], f[t], f"blocks.{si}.{bi}", spec, c)
            c = spec[4]
    for i in range(5):
        for t in tasks:
            f[t] = pad_concat(f[t], skips[t][-i - 1]) if i != 4 else F.interpolate(f[t], scale_factor=2, mode="nearest")
            if debug is not None:
                if f[t].requires_grad:
                    f[t].retain_grad()
                debug.append((f"merge{i}", t, f[t]))
        f = _stitch(sd, f"0_decoder_blocks_{i}", f, tasks)
        for t in tasks:
            n = dec[t]
            y = F.relu(n.bn(n.conv(f[t], f"blocks.{i}.conv1.0", 1, 1), f"blocks.{i}.conv1.1"))
            f[t] = F.relu(n.bn(n.conv(y, f"blocks.{i}.conv2.0", 1, 1), f"blocks.{i}.conv2.1"))
            if debug is not None:
                if y.requires_grad:
                    y.retain_grad()
                    f[t].retain_grad()
                debug.append((f"block{i}.conv1", t, y))
                debug.append((f"block{i}.conv2", t, f[t]))
    return {t: F.conv2d(f[t], sd[f"models.{t}.1.0.weight"], sd[f"models.{t}.1.0.bias"], padding=1) for t in tasks}
