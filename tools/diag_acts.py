"""Diagnostic (GPU box): CSNet decoder activations, HIP vs CPU fp32 oracle: max error and ReLU-mask
mismatches per stage (looks for whole-region mask flips over constant zero-padded areas)."""
import argparse
import sys

import torch

sys.path.insert(0, ".")
from oracle.cross_stitch import csnet_forward
from oracle.losses import synthetic_batch
from vision_mtl_amd.utils.pipeline_utils import build_model

dev = torch.device("cuda:0")
torch.manual_seed(11)
model = build_model(argparse.Namespace(model_name="csnet", backbone_weights=None, channel_wise_stitching=True),
                    argparse.Namespace(num_classes=19))
sd = {k: v.clone() for k, v in model.state_dict().items()}
batch = synthetic_batch(2, 128, 128, 19, seed=11, masked=0.1)
dbg = []
csnet_forward(sd, batch["img"], ["depth", "segm"], True, debug=dbg)
ref = {}
for name, t, v in dbg:
    ref[(name, t)] = v
model = model.to(dev).train()
model.debug_acts = []
model(batch["img"].to(dev))
# map HIP ops to oracle names in order per task
cnt = {"depth": {"merge": 0, "cbr": 0}, "segm": {"merge": 0, "cbr": 0}}
for op, arg, task, v in model.debug_acts:
    c = cnt[task]
    if op in ("merge", "up"):
        name = f"merge{c['merge']}"
        c["merge"] += 1
    else:
        i, which = divmod(c["cbr"], 2)
        name = f"block{i}.conv{which + 1}"
        c["cbr"] += 1
    r = ref[(name, task)]
    err = float((v - r).abs().max() / r.abs().max().clamp_min(1e-30))
    mism = int(((v > 0) != (r > 0)).sum())
    # per-channel constant fraction: share of pixels equal to the channel's most common value
    print(f"{task:5s} {name:14s} shape {tuple(v.shape)} relerr {err:.2e} mask mismatches {mism} / {v.numel()}"
          f"  zeros hip {int((v == 0).sum())} cpu {int((r == 0).sum())}")
