"""Diagnostic (GPU box): run the same CSNet training step twice; all kernels are deterministic, so
every gradient must be bit-identical.  Any difference = race / stale read."""
import argparse
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.losses import synthetic_batch
from vision_mtl_amd.lit_module import MTLModule
from vision_mtl_amd.utils.pipeline_utils import build_model

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="csnet")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(11)
model = build_model(argparse.Namespace(model_name=args.model, backbone_weights=None, channel_wise_stitching=True),
                    argparse.Namespace(num_classes=19)).to(dev).train()
sd0 = {k: v.clone() for k, v in model.state_dict().items()}
batch = {k: v.to(dev) for k, v in synthetic_batch(2, 128, 128, 19, seed=11, masked=0.1).items()}
module = MTLModule(model, num_classes=19, device="cuda:0")
runs = []
for it in range(3):
    model.load_state_dict(sd0)
    model.zero_grad(set_to_none=True)
    loss = module.training_step(batch, 0)
    loss.backward()
    torch.cuda.synchronize()
    runs.append((loss.item(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
for it in (1, 2):
    nd, worst = 0, (0.0, "")
    for k, g in runs[0][1].items():
        d = float((runs[it][1][k] - g).abs().max())
        if d != 0.0:
            nd += 1
            r = d / (float(g.abs().max()) + 1e-30)
            if r > worst[0]:
                worst = (r, k)
    print(f"run {it} vs run 0: loss {runs[it][0]!r} vs {runs[0][0]!r}; {nd} / {len(runs[0][1])} tensors differ; worst {worst}")
