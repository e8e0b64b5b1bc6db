"""Diagnostic (GPU box): per-parameter gradient error of the HIP path vs the fp64 CPU oracle, next
to the fp32 CPU oracle's own error, for csnet / basic.  Not a test; prints a table."""
import argparse
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.cross_stitch import csnet_forward
from oracle.losses import step_losses, synthetic_batch
from oracle.unet_mobilenetv3 import basic_forward
from vision_mtl_amd.lit_module import MTLModule
from vision_mtl_amd.utils.pipeline_utils import build_model

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="csnet")
ap.add_argument("--shape", default="2,128,128")
args = ap.parse_args()
shape = tuple(int(v) for v in args.shape.split(","))
dev = torch.device("cuda:0")
torch.manual_seed(11)
model = build_model(argparse.Namespace(model_name=args.model, backbone_weights=None, channel_wise_stitching=True),
                    argparse.Namespace(num_classes=19))
sd0 = {k: v.clone() for k, v in model.state_dict().items()}
batch = synthetic_batch(*shape, 19, seed=11, masked=0.1)
fwd = (lambda sd, x: csnet_forward(sd, x, ["depth", "segm"], True)) if args.model == "csnet" else (
    lambda sd, x: basic_forward(sd, x, True))


def cpu(dtype):
    sd = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    lv = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    o = fwd(sd, batch["img"].to(dtype))
    ls = step_losses(o, batch["mask"], batch["depth"].to(dtype))
    ls["loss"].backward()
    return o, ls, lv


o64, l64, g64 = cpu(torch.float64)
o32, l32, g32 = cpu(torch.float32)
model = model.to(dev).train()
module = MTLModule(model, num_classes=19, device="cuda:0")
db = {k: v.to(dev) for k, v in batch.items()}
loss = module.training_step(db, 0)
loss.backward()
print("loss hip/cpu32/cpu64", loss.item(), l32["loss"].item(), l64["loss"].item())
rows = []
for k, p in model.named_parameters():
    if g64[k].grad is None:
        continue
    ref = g64[k].grad
    m = float(ref.abs().max())
    eh = float((p.grad.cpu().double() - ref).abs().max()) / (m + 1e-300)
    ec = float((g32[k].grad.double() - ref).abs().max()) / (m + 1e-300)
    l2h = float((p.grad.cpu().double() - ref).norm() / (ref.norm() + 1e-300))
    l2c = float((g32[k].grad.double() - ref).norm() / (ref.norm() + 1e-300))
    rows.append((eh, ec, m, k, l2h, l2c))
print(f"{'hip_max':>10} {'cpu32_max':>10} {'hip_L2':>10} {'cpu32_L2':>10} {'max|g|':>10}  name   (in parameter order)")
for eh, ec, m, k, l2h, l2c in rows:
    flag = " <<<" if eh > 1e-3 and eh > 3 * ec else ""
    print(f"{eh:10.2e} {ec:10.2e} {l2h:10.2e} {l2c:10.2e} {m:10.2e}  {k}{flag}")
import statistics
print("SUMMARY worst max-norm hip", max(r[0] for r in rows), "cpu32", max(r[1] for r in rows))
print("SUMMARY worst L2 hip", max(r[4] for r in rows), "cpu32", max(r[5] for r in rows))
print("SUMMARY median L2 hip", statistics.median(r[4] for r in rows), "cpu32", statistics.median(r[5] for r in rows))
# whole-model gradient vector
num = sum(float((p.grad.cpu().double() - g64[k].grad).pow(2).sum()) for k, p in model.named_parameters() if g64[k].grad is not None)
den = sum(float(g64[k].grad.pow(2).sum()) for k, p in model.named_parameters() if g64[k].grad is not None)
num32 = sum(float((g32[k].grad.double() - g64[k].grad).pow(2).sum()) for k in g64 if g64[k].grad is not None)
print("SUMMARY whole-gradient relative L2: hip", (num / den) ** 0.5, "cpu32", (num32 / den) ** 0.5)
