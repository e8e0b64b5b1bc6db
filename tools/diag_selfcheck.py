"""Diagnostic (GPU box): for every decoder conv-BN-ReLU of the HIP CSNet, take the in-network input
activation and output gradient, recompute the layer backward (a) isolated on HIP, (b) on CPU fp64,
and compare with the in-network input gradient."""
import argparse
import sys

import torch
import torch.nn.functional as F

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.losses import synthetic_batch
from tests.util import from_dev_nhwc, to_dev_nhwc
from vision_mtl_amd import ops
from vision_mtl_amd.lit_module import MTLModule
from vision_mtl_amd.utils.model_utils import get_module_by_name
from vision_mtl_amd.utils.pipeline_utils import build_model

dev = torch.device("cuda:0")
torch.manual_seed(11)
model = build_model(argparse.Namespace(model_name="csnet", backbone_weights=None, channel_wise_stitching=True),
                    argparse.Namespace(num_classes=19)).to(dev).train()
batch = {k: v.to(dev) for k, v in synthetic_batch(2, 128, 128, 19, seed=11, masked=0.1).items()}
model.debug_acts = []
module = MTLModule(model, num_classes=19, device="cuda:0")
module.training_step(batch, 0).backward()


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))


acts = model.debug_acts
for task in ("depth", "segm"):
    seq = [r for r in acts if r[2] == task]
    for idx, (op, arg, _, v, g) in enumerate(seq):
        if op != "conv_bn_relu" or idx == 0:
            continue
        pop, parg, _, pv, pg = seq[idx - 1]
        if pop != "conv_bn_relu":
            continue  # only conv2 of each block: its input is the previous record (conv1 output)
        net = model.models[task]
        conv, bn = get_module_by_name(net, arg[0]), get_module_by_name(net, arg[1])
        w, gam, bet = conv.weight.detach().cpu(), bn.weight.detach().cpu(), bn.bias.detach().cpu()
        C = w.shape[0]
        x64 = pv.double().requires_grad_(True)
        y64 = F.relu(F.batch_norm(F.conv2d(x64, w.double(), None, padding=1), None, None, gam.double(), bet.double(),
                                  training=True, eps=1e-5))
        y64.backward(g.double())
        xd = to_dev_nhwc(pv, dev).requires_grad_(True)
        zd, stats = ops.conv2d(xd, conv.weight.detach(), None, 1, 1, want_stats=True)
        nbt = torch.zeros((), dtype=torch.int64, device=dev)
        yd = ops.bn_act(zd, bn.weight.detach(), bn.bias.detach(), torch.zeros(C, device=dev), torch.ones(C, device=dev),
                        nbt, C, True, 0.1, 1e-5, ops.ACT_RELU, stats=stats)
        yd.backward(to_dev_nhwc(g, dev))
        iso = from_dev_nhwc(xd.grad, pv.shape[1])
        print(f"{task} {arg[0]}: out recomputed vs in-net {rel(from_dev_nhwc(yd.detach(), C), v):.1e} | "
              f"dx: in-net vs cpu64 {rel(pg, x64.grad):.1e}  isolated-hip vs cpu64 {rel(iso, x64.grad):.1e}  "
              f"in-net vs isolated-hip {rel(pg, iso):.1e}")
