"""Checkpoint wire format of reference vision_mtl/utils/pipeline_utils.py:139-167,217-238:
model_{epoch}.pt = {"model": module.state_dict()} (keys prefixed "model."),
session_{epoch}.pt = {"optimizer", "scheduler", "epoch"}."""
from __future__ import annotations

import glob
import os
import re

import torch


def save_ckpt(module, optimizer, scheduler, epoch: int, save_path_model: str, save_path_session: str) -> None:
    torch.save({"model": module.state_dict()}, save_path_model)
    torch.save({"optimizer": optimizer.state_dict(), "scheduler": scheduler.state_dict() if scheduler else None,
                "epoch": epoch}, save_path_session)


def load_ckpt_model(ckpt_dir: str, epoch=None) -> dict:
    """Picks model_{max epoch}.pt (or model.pt) like the reference."""
    cands = glob.glob(os.path.join(ckpt_dir, "model_*.pt"))
    if epoch is None and cands:
        epoch = max(int(re.search(r"model_(\d+)\.pt$", c).group(1)) for c in cands)
    path = os.path.join(ckpt_dir, "model.pt" if epoch is None else f"model_{epoch}.pt")
    return torch.load(path, map_location="cpu")
