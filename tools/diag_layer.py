"""Diagnostic (GPU box): backward of ONE real CSNet layer (segm decoder block4.conv2: conv3x3 16->16 +
BN + ReLU) with the actual tensors of a training step; each sub-step HIP vs CPU fp64."""
import argparse
import sys

import torch
import torch.nn.functional as F

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.cross_stitch import csnet_forward
from oracle.losses import step_losses, synthetic_batch
from tests.util import from_dev_nhwc, to_dev_nhwc
from vision_mtl_amd import ops
from vision_mtl_amd.utils.pipeline_utils import build_model

dev = torch.device("cuda:0")
torch.manual_seed(11)
model = build_model(argparse.Namespace(model_name="csnet", backbone_weights=None, channel_wise_stitching=True),
                    argparse.Namespace(num_classes=19))
sd = {k: v.clone() for k, v in model.state_dict().items()}
for k, v in sd.items():
    if v.is_floating_point() and "running" not in k:
        v.requires_grad_(True)
batch = synthetic_batch(2, 128, 128, 19, seed=11, masked=0.1)
dbg = []
out = csnet_forward(sd, batch["img"], ["depth", "segm"], True, debug=dbg)
step_losses(out, batch["mask"], batch["depth"])["loss"].backward()
ref = {(n, t): v for n, t, v in dbg}


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))


for task in ("depth", "segm"):
    for blk in (4, 2):
        xin = ref[(f"block{blk}.conv1", task)].detach()         # input of conv2 (post-ReLU)
        gout = ref[(f"block{blk}.conv2", task)].grad.detach()   # gradient wrt conv2-BN-ReLU output
        gin_ref32 = ref[(f"block{blk}.conv1", task)].grad.detach()
        pre = f"models.{task}.0.decoder.blocks.{blk}."
        w, gam, bet = sd[pre + "conv2.0.weight"].detach(), sd[pre + "conv2.1.weight"].detach(), sd[pre + "conv2.1.bias"].detach()
        C = w.shape[0]
        # CPU fp64, keeping the intermediate (gradient wrt raw conv output)
        x64 = xin.double().requires_grad_(True)
        z64 = F.conv2d(x64, w.double(), None, padding=1)
        z64.retain_grad()
        y64 = F.relu(F.batch_norm(z64, None, None, gam.double(), bet.double(), training=True, eps=1e-5))
        y64.backward(gout.double())
        # statistics of the conv output: how large is mean vs std per channel?
        zz = z64.detach()
        ratio = (zz.mean((0, 2, 3)).abs() / zz.std((0, 2, 3)).clamp_min(1e-30))
        # HIP, step by step
        xd = to_dev_nhwc(xin, dev).requires_grad_(True)
        zd, stats = ops.conv2d(xd, w.to(dev), None, 1, 1, want_stats=True)
        zd.retain_grad()
        nbt = torch.zeros((), dtype=torch.int64, device=dev)
        yd = ops.bn_act(zd, gam.to(dev), bet.to(dev), torch.zeros(C, device=dev), torch.ones(C, device=dev), nbt, C, True,
                        0.1, 1e-5, ops.ACT_RELU, stats=stats)
        yd.backward(to_dev_nhwc(gout, dev))
        dz_hip, dx_hip = from_dev_nhwc(zd.grad, C), from_dev_nhwc(xd.grad, xin.shape[1])
        # dgrad alone from the EXACT (fp64->fp32) dz
        dz_exact = to_dev_nhwc(z64.grad.float(), dev)
        zd2, _ = ops.conv2d(to_dev_nhwc(xin, dev).requires_grad_(True), w.to(dev), None, 1, 1, want_stats=True)
        xd2 = to_dev_nhwc(xin, dev).requires_grad_(True)
        z2 = ops.conv2d(xd2, w.to(dev), None, 1, 1)
        z2.backward(dz_exact)
        # CPU fp32 of the same isolated layer
        x32 = xin.clone().requires_grad_(True)
        z32 = F.conv2d(x32, w, None, padding=1)
        z32.retain_grad()
        F.relu(F.batch_norm(z32, None, None, gam, bet, training=True, eps=1e-5)).backward(gout)
        print(f"{task} block{blk}.conv2: |mean|/std of conv out: max {float(ratio.max()):.1f} med {float(ratio.median()):.2f};"
              f" invstd max {float(1 / (zz.var((0, 2, 3), unbiased=False) + 1e-5).sqrt().max()):.1f}\n"
              f"   z fwd  hip {rel(from_dev_nhwc(zd.detach(), C), zz):.1e}\n"
              f"   dz(BN bwd out)  hip {rel(dz_hip, z64.grad):.1e}   cpu32 {rel(z32.grad, z64.grad):.1e}   max|dz| {float(z64.grad.abs().max()):.2e}\n"
              f"   dx(full)        hip {rel(dx_hip, x64.grad):.1e}   cpu32 {rel(x32.grad, x64.grad):.1e}   max|dx| {float(x64.grad.abs().max()):.2e}"
              f"   in-net cpu32 {rel(gin_ref32, x64.grad):.1e}\n"
              f"   dx(dgrad only, exact dz) hip {rel(from_dev_nhwc(xd2.grad, xin.shape[1]), x64.grad):.1e}")
