"""Host-side mirror of reference vision_mtl/models/basic_model.py (hard parameter sharing:
one backbone, two 3x3 heads) on the HIP kernels."""
from __future__ import annotations

import typing as t

import torch
from torch import nn

from .. import layers as L
from .. import ops
from .unet_mobilenetv3 import Backbone, SegmentationHead


class BasicMTLModel(nn.Module):
    """reference models/basic_model.py:10-60.  forward(x: (B,3,H,W)) -> {"depth": (B,1,H,W), "segm": (B,C,H,W)}."""

    def __init__(self, segm_classes: int, activation: t.Any = None, encoder_name: str = "timm-mobilenetv3_large_100",
                 encoder_weights: t.Optional[str] = "imagenet", decoder_first_channel: int = 256,
                 num_decoder_layers: int = 5, in_channels: int = 3):
        super().__init__()
        self.backbone = Backbone(encoder_name=encoder_name, encoder_weights=encoder_weights,
                                 decoder_first_channel=decoder_first_channel, num_decoder_layers=num_decoder_layers,
                                 in_channels=in_channels)
        last = self.backbone.decoder_channels[-1]
        self.segm_head = SegmentationHead(last, segm_classes, activation=activation, kernel_size=3)
        self.depth_head = SegmentationHead(last, 1, activation=activation, kernel_size=3)

    def forward(self, x: torch.Tensor) -> t.Dict[str, torch.Tensor]:
        ops.packs.refresh()  # one batched weight-packing launch for the whole step
        sh, dh = self.segm_head[0], self.depth_head[0]
        last = self.backbone.decoder.blocks[-1]
        x1, stats1, rpb1 = self.backbone.run(L.from_nchw(x), raw_tail=True)  # raw conv1 output of the last block
        c2, bn1, bn2 = last.conv2[0], last.conv1[1], last.conv2[1]
        heads_3x3 = tuple(sh.kernel_size) == (3, 3) and tuple(dh.kernel_size) == (3, 3) and sh.padding[0] == 1
        if heads_3x3 and ops.decoder_tail_supported(x1.t.shape, x1.C, c2.out_channels, sh.out_channels + dh.out_channels):
            # the narrow full-resolution tail (BN+ReLU -> conv2 -> BN+ReLU -> both heads) on the halo-tile kernel
            segm, depth = ops.decoder_tail(x1.t, stats1, rpb1, bn1, c2.weight, bn2, sh.weight, sh.bias, dh.weight, dh.bias)
        else:
            dec = L.conv_bn_act(L.bn_act(x1, bn1, ops.ACT_RELU, stats=stats1, stats_rpb=rpb1), c2, bn2, ops.ACT_RELU)
            # both 3x3 heads read the same decoder map: one implicit GEMM with N = C + 1 output channels
            segm, depth = ops.dual_head(dec.t, sh.weight, sh.bias, dh.weight, dh.bias, pad=sh.padding[0])
        return dict(depth=depth, segm=segm)

    @torch.no_grad()
    def predict(self, x: torch.Tensor) -> t.Dict[str, torch.Tensor]:
        if self.training:
            self.eval()
        return self.forward(x)
