"""Diagnostic (GPU box): CSNet decoder activations AND their gradients, HIP vs CPU fp32 oracle."""
import argparse
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.cross_stitch import csnet_forward
from oracle.losses import step_losses, synthetic_batch
from vision_mtl_amd.lit_module import MTLModule
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
ref = {(name, t): v for name, t, v in dbg}
model = model.to(dev).train()
model.debug_acts = []
module = MTLModule(model, num_classes=19, device="cuda:0")
loss = module.training_step({k: v.to(dev) for k, v in batch.items()}, 0)
loss.backward()
cnt = {"depth": {"merge": 0, "cbr": 0}, "segm": {"merge": 0, "cbr": 0}}
for op, arg, task, v, g in model.debug_acts:
    c = cnt[task]
    if op in ("merge", "up"):
        name = f"merge{c['merge']}"
        c["merge"] += 1
    else:
        i, which = divmod(c["cbr"], 2)
        name = f"block{i}.conv{which + 1}"
        c["cbr"] += 1
    r = ref[(name, task)]
    err = float((v - r.detach()).abs().max() / r.detach().abs().max().clamp_min(1e-30))
    line = f"{task:5s} {name:14s} act relerr {err:.1e}"
    if g is not None and r.grad is not None:
        rg = r.grad
        gerr = float((g - rg).abs().max() / rg.abs().max().clamp_min(1e-30))
        # where is the error?  per-channel max error, and error restricted to pixels where the activation is 0
        d = (g - rg).abs()
        ch = d.amax((0, 2, 3)) / rg.abs().amax().clamp_min(1e-30)
        zmask = (r.detach() == 0)
        ez = float(d[zmask].max()) if zmask.any() else 0.0
        enz = float(d[~zmask].max()) if (~zmask).any() else 0.0
        line += (f" | grad relerr {gerr:.1e}  max|g| {float(rg.abs().max()):.2e}  worst ch {int(ch.argmax())} ({float(ch.max()):.1e})"
                 f"  err@act==0 {ez:.1e} err@act!=0 {enz:.1e}  sum(g) hip {float(g.sum()):.4e} cpu {float(rg.sum()):.4e}")
    print(line)
