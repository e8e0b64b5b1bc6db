"""ORACLE (test infrastructure, never shipped or measured as product):
CPU restatement of the reference's MTAN forward as a pure function of a state_dict.

Follows reference vision_mtl/models/mtan_model.py (whole file) and
vision_mtl/utils/model_utils.py:46-80 (concat helper, DoubleConv), written
state-dict-first so it can be checked two ways:
  * against the REAL reference module (oracle/gen_golden.py, run in the build container:
    same state_dict in, bit-identical tensors out), and
  * against the HIP path (tests/, on the GPU box, where /root/reference does not exist).
Parity status: PINNED by tests/golden/mtan_*.pt (generated from the reference itself).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


class _SD:
    def __init__(self, sd, training):
        self.sd, self.training = sd, training

    def conv(self, x, name, pad):
        return F.conv2d(x, self.sd[name + ".weight"], self.sd.get(name + ".bias"), padding=pad)

    def bn(self, x, name):
        sd = self.sd
        if self.training and (name + ".num_batches_tracked") in sd:
            sd[name + ".num_batches_tracked"] += 1
        return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"],
                            sd[name + ".bias"], self.training, 0.1, 1e-5)

    def double_conv(self, x, name):  # model_utils.py:70-77
        x = F.relu(self.bn(self.conv(x, name + ".double_conv.0", 1), name + ".double_conv.1"))
        return F.relu(self.bn(self.conv(x, name + ".double_conv.3", 1), name + ".double_conv.4"))


def pad_concat(x1, x2):
    """model_utils.py:46-58: zero-pad x1 to x2's size (smaller half top/left), cat [x2, x1]."""
    dy, dx = x2.shape[2] - x1.shape[2], x2.shape[3] - x1.shape[3]
    x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    return torch.cat([x2, x1], 1)


def _enc_attention(m, name, shared1, shared2, prev):  # mtan_model.py:51-83
    a = shared1 if prev is None else torch.cat((shared1, prev), 1)
    a = F.relu(m.bn(m.conv(a, name + ".conv1", 0), name + ".bn1"))
    a = torch.sigmoid(m.bn(m.conv(a, name + ".conv2", 0), name + ".bn2"))
    g = shared2 * a
    g = F.relu(m.bn(m.conv(g, name + ".conv3", 1), name + ".bn3"))
    return F.max_pool2d(g, 2)


def _dec_attention(m, name, shared1, prev, shared2):  # mtan_model.py:133-169
    p = F.relu(m.bn(m.conv(prev, name + ".conv3", 1), name + ".bn3"))
    if shared1.shape[2:] != p.shape[2:]:
        p = F.interpolate(p, scale_factor=2, mode="bilinear", align_corners=True)
    a = torch.cat((shared1, p), 1)
    a = F.relu(m.bn(m.conv(a, name + ".conv1", 0), name + ".bn1"))
    a = torch.sigmoid(m.bn(m.conv(a, name + ".conv2", 0), name + ".bn2"))
    g = shared2 * a
    return F.relu(m.bn(m.conv(g, name + ".conv_out", 1), name + ".bn_out"))


def mtan_forward(sd: dict, x: torch.Tensor, tasks: list, n_levels: int = 4, training: bool = True) -> dict:
    """sd: state_dict with the reference's key names (mutated in place for BN running stats when
    training, as nn.BatchNorm2d does).  tasks: task names in map_tasks_to_num_channels order."""
    m = _SD(sd, training)
    T = len(tasks)
    feats, prev, enc = [], None, x
    for i in range(n_levels):  # mtan_model.py:382-388 + MTANDown.forward :187-201
        d = m.double_conv(enc, f"enc_layers.{i}.dconv")
        prev = [_enc_attention(m, f"enc_layers.{i}.task_attn_modules.{t}", enc, d, None if prev is None else prev[t])
                for t in range(T)]
        feats.append(d)
        enc = F.max_pool2d(d, 2)
    dec = m.double_conv(enc, "bottleneck")
    for i in range(n_levels):  # mtan_model.py:394-399 + MTANUp.forward :222-243
        up = F.conv_transpose2d(dec, sd[f"dec_layers.{i}.up.weight"], sd[f"dec_layers.{i}.up.bias"], stride=2)
        merged = pad_concat(up, feats[-(i + 1)])
        dec = m.double_conv(merged, f"dec_layers.{i}.conv")
        prev = [_dec_attention(m, f"dec_layers.{i}.task_attn_modules.{t}", merged, prev[t], dec) for t in range(T)]
    return {task: m.conv(prev[t], f"map_tasks_to_heads.{task}", 0) for t, task in enumerate(tasks)}
