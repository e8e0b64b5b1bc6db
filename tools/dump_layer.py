"""Diagnostic (GPU box): dump activations/gradients around segm decoder block4 from the HIP run and the
fp32 CPU oracle into gpurun_out/layer_dump.pt for offline analysis."""
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
out["segm"].retain_grad()
step_losses(out, batch["mask"], batch["depth"])["loss"].backward()
ref = {(n, t): v for n, t, v in dbg}
model = model.to(dev).train()
model.debug_acts = []
module = MTLModule(model, num_classes=19, device="cuda:0")
db = {k: v.to(dev) for k, v in batch.items()}
raw = model(db["img"])
raw["segm"].retain_grad()
o = module.postprocess_raw_out(raw)
module.calc_losses(db["mask"], db["depth"], o)["loss"].backward()
cnt = {"merge": 0, "cbr": 0}
dump = {"cpu": {}, "hip": {}}
for op, arg, task, v, g in model.debug_acts:
    if task != "segm":
        continue
    if op in ("merge", "up"):
        name = f"merge{cnt['merge']}"
        cnt["merge"] += 1
    else:
        i, which = divmod(cnt["cbr"], 2)
        name = f"block{i}.conv{which + 1}"
        cnt["cbr"] += 1
    if name in ("block4.conv1", "block4.conv2", "merge4", "block3.conv2"):
        dump["hip"][name] = (v, g)
        dump["cpu"][name] = (ref[(name, "segm")].detach(), ref[(name, "segm")].grad)
dump["hip"]["logits"] = (raw["segm"].detach().cpu(), raw["segm"].grad.cpu())
dump["cpu"]["logits"] = (out["segm"].detach(), out["segm"].grad)
pre = "models.segm.0.decoder.blocks.4."
dump["params"] = {k: sd[k].detach() for k in sd if k.startswith(pre) or k.startswith("models.segm.1.")}
dump["mask"] = batch["mask"]
torch.save(dump, "gpurun_out/layer_dump.pt")
print("saved", {k: [tuple(t.shape) for t in v] for k, v in dump["hip"].items()})
