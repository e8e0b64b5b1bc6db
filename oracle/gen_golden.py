"""Generates tests/golden/*.pt from the REAL reference (build container only).

Run:  python oracle/gen_golden.py            (needs /root/reference; never runs on the GPU box)

What it does
  1. imports the reference's pure-torch pieces unchanged (MTANMiniUnet, DoubleConv,
     concat_slightly_diff_sized_tensors, CrossStitchLayer, SILogLoss, calc_loss).  The reference
     imports segmentation_models_pytorch at module top level; that package is absent offline, so
     two names are pre-seeded with classes that raise if instantiated.
  2. PINS the oracle: asserts oracle/*.py reproduce the reference bit-for-bit (same state_dict in,
     torch.equal outputs / losses / gradients out) and that the host-side mirror
     (vision_mtl_amd.models.mtan_model) initialises to the same parameters under the same seed.
  3. writes fixtures = data only (inputs, state_dicts, expected outputs / gradients).

The fixtures are what tests/ compare the oracle (CPU, -m "not gpu") and the HIP path (-m gpu)
against; reference source is never copied.
"""
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference():
    sys.dont_write_bytecode = True  # /root/reference is read-only
    sys.path.insert(0, REF)
    smp = types.ModuleType("segmentation_models_pytorch")
    base = types.ModuleType("segmentation_models_pytorch.base")

    class _Missing(nn.Module):
        def __init__(self, *a, **k):
            raise RuntimeError("segmentation_models_pytorch is unavailable offline")

    smp.Unet = _Missing
    base.SegmentationHead = _Missing
    smp.base = base
    sys.modules["segmentation_models_pytorch"] = smp
    sys.modules["segmentation_models_pytorch.base"] = base
    from vision_mtl.losses import SILogLoss
    from vision_mtl.models.cross_stitch_model import CrossStitchLayer
    from vision_mtl.models.mtan_model import MTANMiniUnet
    from vision_mtl.utils.loss_utils import calc_loss
    from vision_mtl.utils.model_utils import DoubleConv, concat_slightly_diff_sized_tensors

    return dict(SILogLoss=SILogLoss, CrossStitchLayer=CrossStitchLayer, MTANMiniUnet=MTANMiniUnet,
                calc_loss=calc_loss, DoubleConv=DoubleConv, pad_concat=concat_slightly_diff_sized_tensors)


def clone_sd(sd):
    return {k: v.detach().clone() for k, v in sd.items()}


def gen_mtan(ref, name, cfg, shape, seed=11, masked=0.1):
    from oracle.losses import step_losses, synthetic_batch
    from oracle.mtan import mtan_forward
    from vision_mtl_amd.models.mtan_model import MTANMiniUnet as Mirror

    tasks = {"depth": 1, "segm": cfg["C"]}
    kw = dict(in_channels=3, map_tasks_to_num_channels=tasks, task_subnets_hidden_channels=cfg["hidden"],
              encoder_first_channel=cfg["first"], encoder_num_channels=cfg["levels"])
    torch.manual_seed(seed)
    model = ref["MTANMiniUnet"](**kw)
    torch.manual_seed(seed)
    mirror = Mirror(**kw)
    sd0 = clone_sd(model.state_dict())
    msd = mirror.state_dict()
    assert set(sd0) == set(msd), "mirror state_dict keys differ from the reference"
    for k in sd0:
        assert torch.equal(sd0[k], msd[k]), f"mirror init differs from the reference at {k}"

    B, H, W = shape
    batch = synthetic_batch(B, H, W, cfg["C"], seed=seed, masked=masked)
    crit_d = ref["SILogLoss"]()
    crit_s = nn.CrossEntropyLoss()

    # --- reference: train-mode forward, reference loss composition, backward
    model.train()
    out = model(batch["img"])
    loss_ref = ref["calc_loss"](out, batch["mask"], batch["depth"], crit_s, crit_d)
    loss_ref.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    sd_after = clone_sd(model.state_dict())
    model.eval()
    with torch.no_grad():
        out_eval = model(batch["img"])

    # --- oracle restatement on the same state_dict must be bit-identical
    sd = clone_sd(sd0)
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    o = mtan_forward(sd, batch["img"], list(tasks), cfg["levels"], training=True)
    lo = step_losses(o, batch["mask"], batch["depth"])
    lo["loss"].backward()
    for t in tasks:
        assert torch.equal(o[t], out[t]), f"oracle train forward differs ({t})"
    assert torch.equal(lo["loss"], loss_ref), "oracle loss differs"
    for k, g in grads.items():
        assert torch.equal(leaves[k].grad, g), f"oracle gradient differs at {k}"
    for k in sd_after:
        assert torch.equal(sd[k].detach(), sd_after[k]), f"oracle BN buffer differs at {k}"
    with torch.no_grad():
        oe = mtan_forward(clone_sd(sd_after), batch["img"], list(tasks), cfg["levels"], training=False)
    for t in tasks:
        assert torch.equal(oe[t], out_eval[t]), f"oracle eval forward differs ({t})"

    fx = dict(cfg=cfg, tasks=tasks, seed=seed, batch=batch, state_dict=sd0, state_dict_after=sd_after,
              out_train={k: v.detach() for k, v in out.items()}, out_eval=out_eval,
              loss=loss_ref.detach(), loss_segm=lo["loss_segm"].detach(), loss_depth=lo["loss_depth"].detach(),
              grads=grads)
    torch.save(fx, os.path.join(OUT, name))
    n = sum(v.numel() for v in sd0.values())
    print(f"{name}: {n} state elements, loss {loss_ref.item():.6f}  [reference == oracle == mirror-init: OK]")


def gen_components(ref):
    from oracle.losses import silog
    from oracle.mtan import pad_concat

    g = torch.Generator().manual_seed(5)
    fx = {}
    # CrossStitchLayer, both modes (reference cross_stitch_model.py:15-37)
    for cw in (True, False):
        torch.manual_seed(3)
        layer = ref["CrossStitchLayer"](2, 6 if cw else None)
        x = torch.randn(2, 3, 6, 4, 5, generator=g, requires_grad=True)
        y = layer(x)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        fx[f"stitch_cw{int(cw)}"] = dict(w=layer.weights.detach().clone(), x=x.detach().clone(), y=y.detach(), gy=gy,
                                        dx=x.grad.clone(), dw=layer.weights.grad.clone())
    # SILogLoss value + gradient (all-valid and ~10% masked) and CE at C=19/14
    for tag, masked in (("valid", 0.0), ("masked", 0.1)):
        z = torch.randn(2, 1, 12, 20, generator=g, requires_grad=True)
        t = 0.002 + 0.498 * torch.rand(2, 12, 20, 1, generator=g)
        if masked:
            t[torch.rand(2, 12, 20, 1, generator=g) < masked] = 0.0
        p = torch.sigmoid(z).permute(0, 2, 3, 1)
        l = ref["SILogLoss"]()(p, t)
        l.backward()
        lo = silog(torch.sigmoid(z.detach()).permute(0, 2, 3, 1), t)
        assert torch.equal(lo, l.detach()), "oracle silog differs from the reference"
        fx[f"silog_{tag}"] = dict(z=z.detach().clone(), t=t, loss=l.detach(), dz=z.grad.clone())
    for C in (19, 14):
        z = torch.randn(2, C, 9, 11, generator=g, requires_grad=True)
        t = torch.randint(0, C, (2, 9, 11), generator=g)
        l = nn.CrossEntropyLoss()(z, t)
        l.backward()
        fx[f"ce_{C}"] = dict(z=z.detach().clone(), t=t, loss=l.detach(), dz=z.grad.clone())
    # DoubleConv (reference model_utils.py:61-80), train mode
    torch.manual_seed(7)
    dc = ref["DoubleConv"](5, 7)
    x = torch.randn(2, 5, 9, 6, generator=g, requires_grad=True)
    sd0 = clone_sd(dc.state_dict())
    y = dc(x)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    fx["double_conv"] = dict(state_dict=sd0, x=x.detach().clone(), y=y.detach(), gy=gy, dx=x.grad.clone(),
                             grads={k: p.grad.clone() for k, p in dc.named_parameters()},
                             state_dict_after=clone_sd(dc.state_dict()))
    # concat helper with odd size differences (reference model_utils.py:46-58)
    x1, x2 = torch.randn(2, 3, 5, 4, generator=g), torch.randn(2, 2, 8, 9, generator=g)
    y = ref["pad_concat"](x1, x2)
    assert torch.equal(pad_concat(x1, x2), y)
    fx["pad_concat"] = dict(x1=x1, x2=x2, y=y)
    torch.save(fx, os.path.join(OUT, "components.pt"))
    print("components.pt:", sorted(fx))


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    torch.use_deterministic_algorithms(True)
    ref = import_reference()
    gen_mtan(ref, "mtan_tiny.pt", dict(first=4, hidden=8, levels=4, C=5), (2, 32, 48))
    gen_mtan(ref, "mtan_small3.pt", dict(first=8, hidden=16, levels=3, C=14), (2, 24, 40), seed=12, masked=0.0)
    gen_components(ref)


if __name__ == "__main__":
    main()
