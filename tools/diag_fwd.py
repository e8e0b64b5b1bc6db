"""Diagnostic (GPU box): forward accuracy of the HIP CSNet vs an fp64 oracle, next to the fp32 CPU
oracle's own error, stage by stage (decoder) and at the logits."""
import argparse
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.cross_stitch import csnet_forward
from oracle.losses import synthetic_batch
from vision_mtl_amd.utils.pipeline_utils import build_model

dev = torch.device("cuda:0")
torch.manual_seed(11)
model = build_model(argparse.Namespace(model_name="csnet", backbone_weights=None, channel_wise_stitching=True),
                    argparse.Namespace(num_classes=19))
sd0 = {k: v.clone() for k, v in model.state_dict().items()}
batch = synthetic_batch(2, 128, 128, 19, seed=11, masked=0.1)


def cpu(dtype):
    sd = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    dbg = []
    with torch.no_grad():
        out = csnet_forward(sd, batch["img"].to(dtype), ["depth", "segm"], True, debug=dbg)
    return out, {(n, t): v for n, t, v in dbg}


o64, r64 = cpu(torch.float64)
o32, r32 = cpu(torch.float32)
model = model.to(dev).train()
model.debug_acts = []
with torch.no_grad():
    oh = model(batch["img"].to(dev))
cnt = {"depth": {"merge": 0, "cbr": 0}, "segm": {"merge": 0, "cbr": 0}}


def errs(a, ref):
    a, ref = a.double(), ref.double()
    d = a - ref
    return float(d.abs().max() / ref.abs().max()), float(d.norm() / ref.norm())


for op, arg, task, v, g in model.debug_acts:
    c = cnt[task]
    if op in ("merge", "up"):
        name = f"merge{c['merge']}"
        c["merge"] += 1
        continue
    i, which = divmod(c["cbr"], 2)
    name = f"block{i}.conv{which + 1}"
    c["cbr"] += 1
    hm, hl = errs(v, r64[(name, task)])
    cm, cl = errs(r32[(name, task)], r64[(name, task)])
    print(f"{task:5s} {name:13s} HIP max {hm:.1e} L2 {hl:.1e} | CPU32 max {cm:.1e} L2 {cl:.1e} | max|act| {float(r64[(name, task)].abs().max()):.2e}")
for t in ("depth", "segm"):
    hm, hl = errs(oh[t].cpu(), o64[t])
    cm, cl = errs(o32[t], o64[t])
    ab = float((oh[t].cpu().double() - o64[t]).abs().max())
    abc = float((o32[t].double() - o64[t]).abs().max())
    print(f"{t} logits: HIP max {hm:.1e} L2 {hl:.1e} abs {ab:.2e} | CPU32 max {cm:.1e} L2 {cl:.1e} abs {abc:.2e} | max|z| {float(o64[t].abs().max()):.2e}")
