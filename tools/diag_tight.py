"""Diagnostic (GPU box): per-tensor gradient errors of one model against the fp64 oracle under identity activations
(the quantities tests/test_tight_grads_gpu.py asserts on, listed for EVERY parameter instead of stopping at the first)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tests.test_tight_grads_gpu import _oracle_grads
from tests.util import identity_activations

ap = argparse.ArgumentParser()
ap.add_argument("--kind", default="mtan")
ap.add_argument("--shape", type=int, nargs=3, default=[2, 32, 32])
ap.add_argument("--classes", type=int, default=14)
a = ap.parse_args()
from oracle.losses import synthetic_batch
from vision_mtl_amd.lit_module import MTLModule
from vision_mtl_amd.utils.pipeline_utils import build_model

dev = torch.device("cuda:0")
name = "csnet" if a.kind.startswith("csnet") else a.kind
torch.manual_seed(11)
model = build_model(argparse.Namespace(model_name=name, backbone_weights=None, channel_wise_stitching=a.kind == "csnet"),
                    argparse.Namespace(num_classes=a.classes))
g = torch.Generator().manual_seed(5)
with torch.no_grad():
    for n, p in model.named_parameters():
        if p.dim() == 1 and p.numel() > 1 and float(p.detach().abs().max()) in (0.0, 1.0):
            p.add_(torch.randn(p.shape, generator=g) * 0.1)
sd0 = {k: v.clone() for k, v in model.state_dict().items()}
B, H, W = a.shape
batch = synthetic_batch(B, H, W, a.classes, seed=11, masked=0.1)
with identity_activations():
    loss64, g64 = _oracle_grads(name, sd0, batch, torch.float64, {"levels": 4})
    model = model.to(dev).train()
    module = MTLModule(model, num_classes=a.classes, device=str(dev))
    loss = module.training_step({k: v.to(dev) for k, v in batch.items()}, 0)
    loss.backward()
    torch.cuda.synchronize()
print("loss", float(loss), float(loss64))
for k, p in model.named_parameters():
    if p.grad is None or g64.get(k) is None:
        continue
    ref = g64[k].double()
    mag = float(ref.abs().max())
    err = float((p.grad.cpu().double() - ref).abs().max()) / max(mag, 1e-30)
    flag = "  <<<<" if err > 1e-4 else ""
    print(f"{err:9.2e}  mag {mag:9.2e}  {k} {tuple(p.shape)}{flag}")
