"""Checkpoint wire format of reference vision_mtl/utils/pipeline_utils.py:139-167,207-244:
model_{epoch}.pt = {"model": module.state_dict()} (keys prefixed "model."),
session_{epoch}.pt = {"optimizer", "scheduler", "epoch"}.  Same file names, same selection rule (highest epoch
matched by the artifact regex over os.listdir), same errors.  Weights stay in the torch layout on disk (the
NHWC / packed GEMM layouts are device-side only), so files move between the reference and this build in both
directions for every module whose key names the build pins (all of `mtan`; `basic` / `csnet` follow the
published smp / timm names)."""
from __future__ import annotations

import os
import re
import typing as t

import torch


def save_ckpt(module: torch.nn.Module, optimizer: torch.optim.Optimizer, scheduler: t.Any, epoch: int,
              save_path_model: str, save_path_session: str, exp: t.Any = None) -> None:
    """reference utils/pipeline_utils.py:139-167 (`exp`, the Comet experiment, is accepted and ignored: remote
    logging is outside the hot path)."""
    torch.save({"model": module.state_dict()}, save_path_model)
    torch.save({"optimizer": optimizer.state_dict(), "scheduler": scheduler.state_dict(), "epoch": epoch},
               save_path_session)
    print(f"Saved model to {save_path_model}")


def load_ckpt_model(ckpt_dir: str, epoch: t.Optional[int] = None, artifact_name_regex: str = r"model_(\d+).pt") -> t.Any:
    """reference utils/pipeline_utils.py:217-238: model_{epoch}.pt, or the highest epoch found in ckpt_dir;
    ValueError("No model ckpt found") when there is none."""
    if epoch is not None:
        artifact_name = f"model_{epoch}.pt"
    else:
        available = [f for f in os.listdir(ckpt_dir) if re.match(artifact_name_regex, f)]
        if len(available) == 0:
            raise ValueError("No model ckpt found")
        artifact_name = sorted(available, key=lambda x: int(re.match(artifact_name_regex, x).group(1)))[-1]
    path = os.path.join(ckpt_dir, artifact_name)
    print(f"Loading model from {path}")
    return torch.load(path, map_location="cpu")


def load_ckpt_session(ckpt_dir: str, filename: str = "session.pt") -> t.Any:
    """reference utils/pipeline_utils.py:241-244."""
    return torch.load(os.path.join(ckpt_dir, filename), map_location="cpu")


def load_ckpt(ckpt_dir: str, epoch: t.Optional[int] = None, session_filename: t.Optional[str] = None) -> t.Tuple:
    """(session_ckpt, model_ckpt).  The reference's version (utils/pipeline_utils.py:207-214) tuple-unpacks the dict
    load_ckpt_model returns and therefore cannot work; this is what its docstring promises.  The session file is
    `session_filename`, else session_{epoch}.pt for the epoch the model file carries, else the reference's default
    "session.pt"."""
    model_ckpt = load_ckpt_model(ckpt_dir, epoch=epoch)
    if session_filename is None:
        if epoch is None:
            found = [int(re.match(r"model_(\d+).pt", f).group(1)) for f in os.listdir(ckpt_dir) if re.match(r"model_(\d+).pt", f)]
            epoch = max(found) if found else None
        cand = f"session_{epoch}.pt" if epoch is not None else "session.pt"
        session_filename = cand if os.path.exists(os.path.join(ckpt_dir, cand)) else "session.pt"
    return load_ckpt_session(ckpt_dir, session_filename), model_ckpt
