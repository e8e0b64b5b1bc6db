"""Host-side mirror of reference vision_mtl/models/mtan_model.py.

Same class names, constructor arguments, parameter names/shapes (state_dicts interchange with
the reference) and — because sub-modules are created in the reference's order with the same
torch initialisers — the same random initialisation under the same seed.  forward() runs
entirely on the HIP kernels: NHWC activations, BatchNorm statistics taken from the conv
epilogue, the sigmoid gate fused into the BatchNorm-apply pass.
"""
from __future__ import annotations

import typing as t

import torch
from torch import nn

from .. import layers as L
from ..ops import ACT_RELU, ACT_SIGMOID
from ..utils.model_utils import DoubleConv


class _Attention(nn.Module):
    """Shared tail of both attention modules: 1x1 -> BN -> ReLU -> 1x1 -> BN -> sigmoid gate."""

    def _mask_and_gate(self, merged, shared2: L.Act) -> L.Act:  # merged: an Act or the pair a torch.cat would join
        c1, c2 = self.conv1, self.conv2
        if (c1.groups == 1 and c2.kernel_size == (1, 1) and c2.groups == 1 and L._pair(c2.stride) == 1
                and L._pair(c2.padding) == 0 and self.bn1.momentum is not None
                and L.ops.bn_act_conv1x1_supported((merged[0] if isinstance(merged, tuple) else merged).t, ACT_RELU)):
            # conv1 hands its raw output to conv2, which applies bn1 + ReLU on its operand fragments and whose data
            # gradient ends with their backward reduction: no apply pass forward, no reduce pass backward
            raw1, st1, rpb1 = L.conv_raw(merged, c1, self.bn1.training)
            train2 = self.bn2.training
            raw2, st2, rpb2 = L.ops.bn_act_conv1x1(raw1, st1, rpb1, self.bn1, c1.out_channels, ACT_RELU, c2.weight, c2.bias,
                                                   want_stats=train2, zero_bias_grad=train2 and c2.bias is not None)
            return L.bn_act(L.Act(raw2, c2.out_channels), self.bn2, ACT_SIGMOID, mul=shared2, stats=st2, stats_rpb=rpb2)
        a = L.conv_bn_act(merged, self.conv1, self.bn1, ACT_RELU)
        return L.conv_bn_act(a, self.conv2, self.bn2, ACT_SIGMOID, mul=shared2)  # shared2 * sigmoid(bn2(..))


class AttentionModuleEncoder(_Attention):
    """reference models/mtan_model.py:12-83."""

    def __init__(self, shared_1_channels: int, out_channels: int, shared_2_channels: int,
                 prev_layer_out_channels: t.Optional[int] = None, hidden_channels: int = 64):
        super().__init__()
        self.is_first = prev_layer_out_channels is None
        prev = prev_layer_out_channels or 0
        self.conv1 = nn.Conv2d(shared_1_channels + prev, hidden_channels, kernel_size=1)
        self.bn1 = nn.BatchNorm2d(hidden_channels)
        self.conv2 = nn.Conv2d(hidden_channels, shared_2_channels, kernel_size=1)
        self.bn2 = nn.BatchNorm2d(shared_2_channels)
        self.conv3 = nn.Conv2d(shared_2_channels, out_channels, kernel_size=3, padding=1)
        self.bn3 = nn.BatchNorm2d(out_channels)

    def forward(self, conv1_shared: L.Act, conv2_shared: L.Act, prev_layer_outs: t.Optional[L.Act] = None) -> L.Act:
        if self.is_first:
            merged = conv1_shared
        else:
            assert prev_layer_outs is not None, "prev_layer_outs must be provided for non-first AttentionModuleEncoder"
            merged = (conv1_shared, prev_layer_outs)  # read by conv1 as two sources: no concat pass
        g = self._mask_and_gate(merged, conv2_shared)
        return L.conv_bn_act_maxpool2(g, self.conv3, self.bn3, ACT_RELU)  # bn3 + ReLU + MaxPool2d: one node


class AttentionModuleDecoder(_Attention):
    """reference models/mtan_model.py:86-169."""

    def __init__(self, shared_1_channels: int, shared_2_channels: int, prev_layer_out_channels: int,
                 out_channels: int, hidden_channels: int = 64):
        super().__init__()
        self.conv1 = nn.Conv2d(shared_1_channels + hidden_channels, hidden_channels, kernel_size=1)
        self.bn1 = nn.BatchNorm2d(hidden_channels)
        self.conv2 = nn.Conv2d(hidden_channels, shared_2_channels, kernel_size=1)
        self.bn2 = nn.BatchNorm2d(shared_2_channels)
        self.conv3 = nn.Conv2d(prev_layer_out_channels, hidden_channels, kernel_size=3, padding=1)
        self.bn3 = nn.BatchNorm2d(hidden_channels)
        self.conv_out = nn.Conv2d(shared_2_channels, out_channels, kernel_size=3, padding=1)
        self.bn_out = nn.BatchNorm2d(out_channels)

    def forward(self, conv1_shared: L.Act, prev_layer_outs: L.Act, conv2_shared: L.Act) -> L.Act:
        p = L.conv_bn_act(prev_layer_outs, self.conv3, self.bn3, ACT_RELU)
        if conv1_shared.hw != p.hw:
            p = L.bilinear_up2(p)
        assert conv1_shared.hw == conv2_shared.hw
        g = self._mask_and_gate((conv1_shared, p), conv2_shared)
        return L.conv_bn_act(g, self.conv_out, self.bn_out, ACT_RELU)


class MTANDown(nn.Module):
    """reference models/mtan_model.py:172-201."""

    def __init__(self, in_channels: int, out_channels: int, task_attn_modules, apply_pool: bool = True):
        super().__init__()
        self.dconv = DoubleConv(in_channels, out_channels)
        self.pool = nn.MaxPool2d(2) if apply_pool else nn.Identity()
        self.task_attn_modules = task_attn_modules

    def forward(self, x: L.Act, prev_layer_outs=None, n_out: int = 1):
        """n_out > 1: the shared feature comes back as that many handles (one per downstream consumer), so that ALL its
        gradients - both attention gates, the next stage, the decoder skip - are summed by one launch (L.fork)."""
        T = len(self.task_attn_modules)
        xs = L.fork(x, T + 1)  # x feeds the shared double conv and every task's attention module
        ds = L.fork(self.dconv.run(xs[0]), T + n_out)
        outs = [m(xs[1 + i], ds[i], prev_layer_outs[i] if prev_layer_outs else None)
                for i, m in enumerate(self.task_attn_modules)]
        if isinstance(self.pool, nn.MaxPool2d):
            return L.maxpool2(ds[T]), outs
        return (ds[T] if n_out == 1 else ds[T:]), outs


class MTANUp(nn.Module):
    """reference models/mtan_model.py:204-243."""

    def __init__(self, in_channels: int, out_channels: int, task_attn_modules):
        super().__init__()
        self.up = nn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size=2, stride=2)
        self.conv = DoubleConv(in_channels, out_channels)
        self.task_attn_modules = task_attn_modules
        self.out_channels, self.in_channels = out_channels, in_channels

    def forward(self, x1: L.Act, x2: L.Act, task_attn_prev_outs, last: bool = False):
        """last: nobody downstream reads the shared feature (the final decoder stage: only the attention gates do)."""
        T = len(self.task_attn_modules)
        ms = L.fork(L.pad_cat(L.conv_transpose(x1, self.up), x2), T + 1)
        cs = L.fork(self.conv.run(ms[0]), T if last else T + 1)
        outs = [m(ms[1 + i], task_attn_prev_outs[i], cs[i]) for i, m in enumerate(self.task_attn_modules)]
        return (None if last else cs[T]), outs


class MTANMiniUnet(nn.Module):
    """reference models/mtan_model.py:246-404.  forward(x: (B,C,H,W)) -> {task: (B,C_task,H,W)}."""

    def __init__(self, in_channels: int, map_tasks_to_num_channels: t.Dict[str, int],
                 task_subnets_hidden_channels: int = 128, encoder_first_channel: int = 64,
                 encoder_num_channels: int = 4):
        super().__init__()
        self.num_tasks = len(map_tasks_to_num_channels)
        self.in_channels = in_channels
        enc_out = [encoder_first_channel * 2 ** i for i in range(encoder_num_channels)]
        enc_in = [in_channels] + enc_out[:-1]
        dec_out = enc_out[::-1]
        dec_in = [enc_out[-1] * 2] + dec_out[:-1]
        hid, T = task_subnets_hidden_channels, self.num_tasks
        # creation order (bottleneck, encoder attention, decoder attention, encoder, decoder, heads)
        # follows the reference so a shared seed yields identical parameters
        self.bottleneck = DoubleConv(enc_out[-1], enc_out[-1] * 2)
        attn_enc = [nn.ModuleList([AttentionModuleEncoder(
            shared_1_channels=enc_in[i], shared_2_channels=enc_out[i], out_channels=enc_out[i],
            prev_layer_out_channels=None if i == 0 else enc_out[i - 1], hidden_channels=hid) for _ in range(T)])
            for i in range(encoder_num_channels)]
        attn_dec = [nn.ModuleList([AttentionModuleDecoder(
            shared_1_channels=dec_in[i], shared_2_channels=dec_out[i],
            prev_layer_out_channels=enc_out[-1] if i == 0 else dec_out[i - 1], out_channels=dec_out[i],
            hidden_channels=hid) for _ in range(T)]) for i in range(encoder_num_channels)]
        self.enc_layers = nn.ModuleList([MTANDown(enc_in[i], enc_out[i], attn_enc[i], apply_pool=False)
                                         for i in range(encoder_num_channels)])
        self.dec_layers = nn.ModuleList([MTANUp(dec_in[i], dec_out[i], attn_dec[i])
                                         for i in range(encoder_num_channels)])
        self.pool = nn.MaxPool2d(2)
        self.map_tasks_to_heads = nn.ModuleDict({
            task: nn.Conv2d(dec_out[-1], n_out, kernel_size=1) for task, n_out in map_tasks_to_num_channels.items()})

    def forward(self, x: torch.Tensor) -> t.Dict[str, torch.Tensor]:
        L.ops.packs.refresh()  # one batched weight-packing launch for the whole step
        enc = L.from_nchw(x)
        feats, attn = [], None
        for layer in self.enc_layers:
            (d_skip, d_next), attn = layer(enc, attn, n_out=2)  # the shared feature goes to the decoder AND the next stage
            feats.append(d_skip)
            enc = L.maxpool2(d_next)
        dec = self.bottleneck.run(enc)
        n = len(self.dec_layers)
        for i, layer in enumerate(self.dec_layers):
            dec, attn = layer(dec, feats[-(i + 1)], attn, last=i == n - 1)
        return {task: L.to_nchw(L.conv(attn[i], head)) for i, (task, head) in enumerate(self.map_tasks_to_heads.items())}
