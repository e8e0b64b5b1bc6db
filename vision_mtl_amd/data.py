"""Input side of the hot path: the dataset SAMPLE CONTRACT of the reference and
the host->device hand-over, MI355X-first.

The reference's datasets store an image as HWC float in [0,1] (`.npy`, data_modules/cityscapes.py:69-83) or an
8-bit PNG (nyuv2.py:100-141); a host transform (albumentations ToTensorV2 / torchvision ToTensor, cfg.py:103-114,
144-155) transposes every sample to CHW before batching, and `transfer_batch_to_device` (lit_module.py:211-219)
uploads it.  The HIP path stores activations as NHWC, so the host transpose would only be undone on the device.
Here the sample keeps its storage layout: `prepare_sample` applies the reference's value rules (mask -1 ->
C-1, dtypes, depth normalisation, depth as (H, W, 1)), `collate` stacks into PINNED host tensors, and
`upload_batch` copies asynchronously and re-lays the image with ONE kernel (vmtl_hwc_to_nhwc_pad) straight into
the model's input storage.  What the caller sees stays the reference's contract: batch["img"] is a (B, 3, H, W)
float tensor (a channels-last view over that storage; the models pick the storage up without another copy).
Dataset file I/O, augmentation and DataLoader workers stay out of scope (host code).
"""
from __future__ import annotations

import typing as t

import numpy as np
import torch


def synthetic_batch(B: int, H: int, W: int, C: int, seed: int = 11, masked: float = 0.0) -> dict:
    """SURVEY.md section 8(d) synthetic inputs in the reference's batch contract (seed 11 = reference cfg.py:194):
    img (B,3,H,W) in [0,1), mask (B,H,W) int64 in [0,C), depth (B,H,W,1) in [0.002, 0.5) (`masked`: that share of
    depth pixels set to 0 = invalid)."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, 3, H, W, generator=g)
    mask = torch.randint(0, C, (B, H, W), generator=g)
    depth = 0.002 + 0.498 * torch.rand(B, H, W, 1, generator=g)
    if masked > 0:
        depth[torch.rand(B, H, W, 1, generator=g) < masked] = 0.0
    return {"img": img, "mask": mask, "depth": depth}


def prepare_sample(raw: dict, num_classes: int, max_depth: float = 1.0, dataset: str = "cityscapes") -> dict:
    """Value rules of the reference's two datasets on one raw sample {"img": (H,W,3), "mask": (H,W), "depth": (H,W) or
    (H,W,1)} of numpy arrays / tensors, WITHOUT the CHW transpose.

    dataset="cityscapes" (reference data_modules/cityscapes.py:39-67): mask == -1 -> num_classes-1, img float32, mask
    int64, depth float32 divided by max_depth when its maximum exceeds 1 (common_ds.py:47-50).
    dataset="nyuv2" (reference data_modules/nyuv2.py:100-141): img divided by 255 when its maximum exceeds 1 (8-bit
    PNG), a mask that a ToTensor transform scaled into [0,1] is multiplied back by 255 (`mask.max() <= 1.0`), mask
    squeezed to (H,W) int64 with NO -1 remap, depth = uint16 PNG value / 1e4 and THEN the max_depth rule
    (NYUv2Config.max_depth = 10: cfg.py:117-155).  Leaving the /1e4 step out trains SILog on targets in the 1e3-1e4
    range with no error, so the dataset has to be named - there is no way to tell a raw uint16 map from metres.

    Both: depth comes back as (H, W, 1) (SILog needs the trailing 1)."""
    if dataset not in ("cityscapes", "nyuv2"):
        raise ValueError(f"prepare_sample: dataset must be 'cityscapes' or 'nyuv2', got {dataset!r}")
    as_t = lambda a: a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
    img, mask, depth = as_t(raw["img"]).float(), as_t(raw["mask"]).clone(), as_t(raw["depth"]).float().clone()
    if img.dim() != 3 or img.shape[-1] != 3:
        raise ValueError(f"prepare_sample: img must be (H, W, 3), got {tuple(img.shape)}")
    if dataset == "nyuv2":
        if img.max() > 1.0:  # nyuv2.py:118-119
            img = img / 255
        if mask.is_floating_point() and mask.max() <= 1.0:  # nyuv2.py:121-123: ToTensor scaled the class ids by 1/255
            mask = (mask * 255).round()
        mask = mask.squeeze().long()  # nyuv2.py:124
        depth = depth / 1e4  # nyuv2.py:126-127: the depth PNG is uint16, 1e-4 m per count
    else:
        mask = mask.long()
        mask[mask == -1] = num_classes - 1  # cityscapes.py:42
    if depth.max() > 1.0:  # common_ds.py:47-50
        depth /= max_depth
    if depth.dim() == 3 and depth.shape[0] == 1 and depth.shape[-1] != 1:  # nyuv2.py:130-131: (1,H,W) -> (H,W,1)
        depth = depth.permute(1, 2, 0)
    if depth.dim() == 2:
        depth = depth.unsqueeze(-1)
    if tuple(mask.shape) != tuple(img.shape[:2]) or tuple(depth.shape) != (*img.shape[:2], 1):
        raise ValueError("prepare_sample: img / mask / depth sizes differ")
    return {"img": img.contiguous(), "mask": mask, "depth": depth.contiguous()}


def collate(samples: t.Sequence[dict], pin: bool = True) -> dict:
    """Stack prepared samples into {"img": (B,H,W,3), "mask": (B,H,W), "depth": (B,H,W,1)}; pinned host memory
    when a GPU is present, so that upload_batch's copies are asynchronous."""
    out = {k: torch.stack([s[k] for s in samples]) for k in ("img", "mask", "depth")}
    if pin and torch.cuda.is_available():
        out = {k: v.pin_memory() for k, v in out.items()}
    return out


def upload_batch(batch: dict, device) -> dict:
    """Host batch -> device batch in the reference's contract.  An image stacked in sample layout (B,H,W,3) is
    re-laid by one HIP kernel into the model's NHWC storage and returned as a (B,3,H,W) view of it; everything
    else (an NCHW image included) is an asynchronous copy when pinned (reference lit_module.py:211-219)."""
    from . import ops

    def up(v):
        return v.to(device, non_blocking=v.device.type == "cpu" and v.is_pinned())

    out = {}
    for k, v in batch.items():
        if k == "img" and v.dim() == 4 and v.shape[-1] == 3 and v.shape[1] != 3:
            out[k] = ops.hwc_to_model_input(up(v.float()))
        else:
            out[k] = up(v)
    return out
