"""-m gpu: the HIP `basic` model (MobileNetV3-Large encoder + U-Net decoder 540..33 + two 3x3 heads)
against the CPU oracle (oracle/unet_mobilenetv3.py) on identical weights and inputs: outputs and
step loss within 1e-4 (max-norm); parameter gradients as accurate (rel-L2 vs an fp64 oracle run) as
the fp32 CPU oracle itself - the ~50 stacked train-mode BatchNorms make this model ill-conditioned.
(`basic` is reference-unpinned: smp/timm are not available offline, see the oracle header.)"""
import argparse

import pytest
import torch

from tests.util import assert_close, assert_grads_as_good_as_fp32_cpu

pytestmark = pytest.mark.gpu


def _cpu_step(sd, batch, training=True, dtype=torch.float32):
    from oracle.losses import step_losses
    from oracle.unet_mobilenetv3 import basic_forward

    sd = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    b = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in batch.items()}
    out = basic_forward(sd, b["img"], training)
    losses = step_losses(out, b["mask"], b["depth"])
    losses["loss"].backward()
    return out, losses, leaves, sd


@pytest.mark.parametrize("shape", [(2, 128, 128), (3, 64, 96)])
def test_basic_step_matches_oracle(dev, shape):
    from oracle.losses import synthetic_batch
    from vision_mtl_amd.lit_module import MTLModule
    from vision_mtl_amd.utils.pipeline_utils import build_model

    torch.manual_seed(11)
    args = argparse.Namespace(model_name="basic", backbone_weights=None)
    model = build_model(args, argparse.Namespace(num_classes=19))
    # perturb BN affine params / running stats so they are not the trivial 1/0
    g = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if k.endswith("running_mean"):
                v.copy_(torch.randn(v.shape, generator=g) * 0.1)
            elif k.endswith("running_var"):
                v.copy_(torch.rand(v.shape, generator=g) + 0.5)
        for n, p in model.named_parameters():
            if p.dim() == 1 and ("bn" in n or ".1." in n):
                p.copy_(p + torch.randn(p.shape, generator=g) * 0.1)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    B, H, W = shape
    batch = synthetic_batch(B, H, W, 19, seed=11, masked=0.1)
    out_ref, losses_ref, leaves, sd_after = _cpu_step(sd0, batch, training=True)
    _, _, leaves64, _ = _cpu_step(sd0, batch, training=True, dtype=torch.float64)  # "true" gradients

    model = model.to(dev).train()
    module = MTLModule(model, num_classes=19, device=str(dev))
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    out = model(dbatch["img"])
    for t in ("depth", "segm"):
        assert out[t].shape == out_ref[t].shape and out[t].is_contiguous()
        assert_close(out[t].detach().cpu(), out_ref[t].detach(), tol=1e-4, what=f"train out {t}")
    model.load_state_dict(sd0)  # undo the BN buffer update of the probe forward
    loss = module.training_step(dbatch, 0)
    loss.backward()
    assert_close(loss.detach().cpu(), losses_ref["loss"].detach(), tol=1e-4, what="step loss")
    # gradient bar: see tests/util.py::assert_grads_as_good_as_fp32_cpu
    hip = {k: p.grad.cpu() for k, p in model.named_parameters()}
    assert all(g is not None for g in hip.values())
    assert_grads_as_good_as_fp32_cpu(hip, {k: v.grad for k, v in leaves64.items()},
                                     {k: v.grad for k, v in leaves.items()})
    sd = model.state_dict()
    for k, v in sd_after.items():
        if "running" in k:
            assert_close(sd[k].cpu(), v.detach(), tol=1e-4, what=k)
    # eval mode
    from oracle.unet_mobilenetv3 import basic_forward

    with torch.no_grad():
        ref_eval = basic_forward({k: v.detach().clone() for k, v in sd_after.items()}, batch["img"], False)
        model.eval()
        oe = model.predict(dbatch["img"])
    for t in ("depth", "segm"):
        assert_close(oe[t].cpu(), ref_eval[t], tol=1e-4, what=f"eval out {t}")


def test_build_model_errors(dev):
    from vision_mtl_amd.utils.pipeline_utils import build_model

    with pytest.raises(NotImplementedError):
        build_model(argparse.Namespace(model_name="nope", backbone_weights=None), argparse.Namespace(num_classes=3))
    with pytest.raises(RuntimeError, match="download"):
        build_model(argparse.Namespace(model_name="basic", backbone_weights="imagenet"),
                    argparse.Namespace(num_classes=3))


def test_basic_fallback_tail_uses_the_up2_statistics_block(dev, monkeypatch):
    """VMTL_SMALL_TAIL=0: the last decoder block's conv2 + heads through the implicit-GEMM kernels instead of the fused
    halo-tile node.  The BatchNorm after the block's phase-decomposed conv1 must merge that launch's statistics rows
    with ITS row block (the up2 tile picker's, carried as `_vmtl_rpb` / stats_rpb), not conv_pick_tile's: loss and
    every gradient equal the default path's."""
    from oracle.losses import synthetic_batch
    from vision_mtl_amd.lit_module import MTLModule
    from vision_mtl_amd.utils.pipeline_utils import build_model

    torch.manual_seed(11)
    model = build_model(argparse.Namespace(model_name="basic", backbone_weights=None), argparse.Namespace(num_classes=19))
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev).train()
    module = MTLModule(model, num_classes=19, device=str(dev))
    batch = {k: v.to(dev) for k, v in synthetic_batch(2, 64, 128, 19, seed=11, masked=0.1).items()}

    def run():
        model.load_state_dict(sd0)
        for p in model.parameters():
            p.grad = None
        loss = module.training_step(batch, 0)
        loss.backward()
        return loss.detach().cpu(), {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()}

    l0, g0 = run()
    monkeypatch.setenv("VMTL_SMALL_TAIL", "0")
    l1, g1 = run()
    assert_close(l1, l0, tol=1e-5, what="fallback tail loss")
    from tests.util import rel_l2

    gmax = max(float(v.abs().max()) for v in g0.values())
    for k in g0:  # two fp32 summation orders of a ReLU network: the rel-L2 bar of the other end-to-end tests
        if float(g0[k].abs().max()) > 1e-6 * gmax:
            assert rel_l2(g1[k], g0[k]) <= 5e-2, f"fallback tail grad {k}: {rel_l2(g1[k], g0[k]):.2e}"
