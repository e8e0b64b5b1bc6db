"""-m gpu: end-to-end parameter gradients of all three models held to a TIGHT bar in a variant of the network in which
ReLU-mask flips cannot happen (tests/util.py::identity_activations: every kinked activation -> identity on both sides,
train-mode BatchNorm kept).  The ordinary end-to-end tests have to allow a 5e-2 per-tensor floor because one flipped
mask in 5e5 pre-activations moves a gradient by O(|g|); that floor would also hide an indexing bug in a small tensor.
Here every gradient must agree with the fp64 oracle to 1e-4 of its magnitude (or 4x the fp32 CPU oracle's own error)."""
import argparse

import pytest
import torch

from tests.util import assert_close, assert_grads_tight, identity_activations

pytestmark = pytest.mark.gpu


def _leaves(sd, dtype):
    sd = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    return sd, leaves


def _oracle_grads(kind, sd0, batch, dtype, extra):
    from oracle.losses import step_losses

    sd, leaves = _leaves(sd0, dtype)
    img = batch["img"].to(dtype)
    if kind == "basic":
        from oracle.unet_mobilenetv3 import basic_forward

        raw = basic_forward(sd, img, True)
    elif kind == "csnet":
        from oracle.cross_stitch import csnet_forward

        raw = csnet_forward(sd, img, ["depth", "segm"], True)
    else:
        from oracle.mtan import mtan_forward

        raw = mtan_forward(sd, img, ["depth", "segm"], extra["levels"], True)
    losses = step_losses(raw, batch["mask"], batch["depth"].to(dtype))
    losses["loss"].backward()
    return losses["loss"].detach(), {k: v.grad for k, v in leaves.items()}


@pytest.mark.parametrize("kind,shape,C", [("basic", (2, 64, 64), 19), ("csnet", (2, 64, 64), 19), ("csnet_layer", (2, 64, 64), 19),
                                          ("mtan", (2, 32, 32), 14)])
def test_every_gradient_is_tight_without_mask_flips(dev, kind, shape, C):
    from oracle.losses import synthetic_batch
    from vision_mtl_amd.lit_module import MTLModule
    from vision_mtl_amd.utils.pipeline_utils import build_model

    name = "csnet" if kind.startswith("csnet") else kind
    torch.manual_seed(11)
    model = build_model(argparse.Namespace(model_name=name, backbone_weights=None, channel_wise_stitching=kind == "csnet"),
                        argparse.Namespace(num_classes=C))  # production widths (mtan: 13.28 M parameters)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():  # non-trivial BatchNorm affine parameters
        for n, p in model.named_parameters():
            if p.dim() == 1 and p.numel() > 1 and float(p.detach().abs().max()) in (0.0, 1.0):
                p.add_(torch.randn(p.shape, generator=g) * 0.1)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    B, H, W = shape
    batch = synthetic_batch(B, H, W, C, seed=11, masked=0.1)
    extra = {"levels": 4}
    with identity_activations():
        loss32, g32 = _oracle_grads(name, sd0, batch, torch.float32, extra)
        loss64, g64 = _oracle_grads(name, sd0, batch, torch.float64, extra)
        model = model.to(dev).train()
        module = MTLModule(model, num_classes=C, device=str(dev))
        loss = module.training_step({k: v.to(dev) for k, v in batch.items()}, 0)
        loss.backward()
        torch.cuda.synchronize()
    assert_close(loss.detach().cpu(), loss64.float(), tol=1e-4, what=f"{kind} loss (identity activations)")
    hip = {k: p.grad.cpu() for k, p in model.named_parameters() if p.grad is not None}
    missing = [k for k, p in model.named_parameters() if p.grad is None and k in g64 and g64[k] is not None
               and float(g64[k].abs().max()) > 0]
    assert not missing, f"no gradient for {missing[:5]}"
    eh, ec, k = assert_grads_tight(hip, g64, g32)
    print(f"{kind}: worst gradient error {eh:.2e} of its magnitude at {k} (fp32 CPU oracle there: {ec:.2e})")
