"""ORACLE (test infrastructure): CPU restatement of the reference's loss composition.

  silog       : reference vision_mtl/losses.py:14-36 (incl. the bilinear interpolate, which is an
                identity for (B,H,W,1) inputs)
  step_losses : reference vision_mtl/lit_module.py:120-144 (postprocess_raw_out + calc_losses),
                also stated at vision_mtl/utils/loss_utils.py:8-24
Parity status: PINNED by tests/golden/losses.pt (values and gradients produced by the reference's
own SILogLoss / torch.nn.CrossEntropyLoss in the build container).
"""
import torch
import torch.nn.functional as F


def silog(pred, target, min_depth=1e-3, interpolate=True):
    if interpolate:  # losses.py:23-27
        pred = F.interpolate(pred, target.shape[-2:], mode="bilinear", align_corners=True)
    mask = target > min_depth
    g = torch.log(pred[mask]) - torch.log(target[mask])
    return 10 * torch.sqrt(torch.var(g) + 0.15 * torch.mean(g) ** 2)


def postprocess(raw):  # lit_module.py:133-144
    return {
        "segm_logits": raw["segm"],
        "segm_predictions": torch.argmax(F.softmax(raw["segm"], dim=1), dim=1),
        "depth_predictions": torch.sigmoid(raw["depth"]).permute(0, 2, 3, 1),
    }


def step_losses(raw, gt_mask, gt_depth, w_segm=1.0, w_depth=1.0):  # lit_module.py:120-131
    out = postprocess(raw)
    ls = F.cross_entropy(out["segm_logits"], gt_mask)
    ld = silog(out["depth_predictions"], gt_depth)
    return {"loss": w_segm * ls + w_depth * ld, "loss_segm": ls, "loss_depth": ld}


def synthetic_batch(B, H, W, C, seed=11, masked=0.0):
    """SURVEY.md §8(d) synthetic inputs (seed 11 = reference cfg.py:194)."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, 3, H, W, generator=g)
    mask = torch.randint(0, C, (B, H, W), generator=g)
    depth = 0.002 + 0.498 * torch.rand(B, H, W, 1, generator=g)
    if masked > 0:
        depth[torch.rand(B, H, W, 1, generator=g) < masked] = 0.0
    return {"img": img, "mask": mask, "depth": depth}
