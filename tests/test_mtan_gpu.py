"""-m gpu: the HIP MTAN path against golden vectors produced by the real reference
(tests/golden/mtan_*.pt) and against the CPU oracle at another size.  Bars: outputs and loss
within 1e-4 (north_star), parameter gradients within 1e-3 of each tensor's max magnitude
(40+ layers of train-mode BatchNorm amplify fp32 summation-order differences)."""
import os

import pytest
import torch

from tests.util import assert_close

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _build(fx, dev):
    from vision_mtl_amd.models.mtan_model import MTANMiniUnet

    c = fx["cfg"]
    m = MTANMiniUnet(3, dict(fx["tasks"]), c["hidden"], c["first"], c["levels"])
    m.load_state_dict(fx["state_dict"])
    return m.to(dev)


@pytest.mark.parametrize("name", ["mtan_tiny.pt", "mtan_small3.pt"])
def test_mtan_matches_reference_golden(dev, name):
    from vision_mtl_amd.lit_module import MTLModule

    fx = torch.load(os.path.join(G, name), weights_only=False)
    model = _build(fx, dev)
    module = MTLModule(model, num_classes=fx["cfg"]["C"], device=str(dev))
    batch = {k: v.to(dev) for k, v in fx["batch"].items()}
    model.train()
    loss = module.training_step(batch, 0)
    loss.backward()
    assert_close(loss.detach().cpu(), fx["loss"], tol=1e-4, what="step loss")
    grads = dict(model.named_parameters())
    gscale = max(float(g.abs().max()) for g in fx["grads"].values())
    for k, g in fx["grads"].items():
        assert grads[k].grad is not None, f"no gradient for {k}"
        # atol: biases in front of a train-mode BatchNorm have an analytically zero gradient
        assert_close(grads[k].grad.cpu(), g, tol=1e-3, atol=1e-6 * gscale, what=f"grad {k}")
    # BN running statistics and counters were updated like the reference's
    sd = model.state_dict()
    for k, v in fx["state_dict_after"].items():
        if "running" in k:
            assert_close(sd[k].cpu(), v, tol=1e-4, what=k)
        elif "num_batches" in k:
            assert int(sd[k]) == int(v), k
    # eval-mode forward from the post-step buffers (parameters are unchanged: no optimizer ran)
    model.eval()
    with torch.no_grad():
        oe = model(batch["img"])
    for t, ref in fx["out_eval"].items():
        assert tuple(oe[t].shape) == tuple(ref.shape) and oe[t].is_contiguous()
        assert_close(oe[t].cpu(), ref, tol=1e-4, what=f"eval out {t}")
    assert list(oe.keys()) == list(fx["tasks"])  # depth, segm — the reference's dict order


def test_mtan_train_forward_outputs(dev):
    fx = torch.load(os.path.join(G, "mtan_tiny.pt"), weights_only=False)
    model = _build(fx, dev).train()
    out = model(fx["batch"]["img"].to(dev))
    for t, ref in fx["out_train"].items():
        assert_close(out[t].detach().cpu(), ref, tol=1e-4, what=f"train out {t}")


def test_mtan_needs_multiple_of_16(dev):
    """reference fact: 24x40 fails inside torch.cat (mtan_model.py:152); here the cat helper asserts."""
    from vision_mtl_amd.models.mtan_model import MTANMiniUnet

    m = MTANMiniUnet(3, {"depth": 1, "segm": 3}, 8, 4, 4).to(dev)
    with pytest.raises((AssertionError, RuntimeError, ValueError)):
        m(torch.rand(1, 3, 24, 40, device=dev))


def test_predict_step_surface(dev):
    from vision_mtl_amd.lit_module import MTLModule

    fx = torch.load(os.path.join(G, "mtan_tiny.pt"), weights_only=False)
    model = _build(fx, dev)
    module = MTLModule(model, num_classes=fx["cfg"]["C"], device=str(dev)).eval()
    batch = {k: v.to(dev) for k, v in fx["batch"].items()}
    with torch.no_grad():
        preds = module.predict_step(batch, 0, 0)
    B, _, H, W = fx["batch"]["img"].shape
    assert preds["segm"].shape == (B, H, W) and preds["segm"].dtype == torch.int64
    assert preds["depth"].shape == (B, H, W, 1)
    m = module.on_predict_epoch_end()
    assert set(m) == {"predict/loss", "predict/accuracy", "predict/jaccard_index", "predict/fbeta_score", "predict/mae"}
    # metrics against their definitions on the CPU
    sp, st = preds["segm"].cpu().flatten(), fx["batch"]["mask"].flatten()
    assert abs(m["predict/accuracy"] - (sp == st).float().mean().item()) < 1e-6
    assert abs(m["predict/mae"] - (preds["depth"].cpu() - fx["batch"]["depth"]).abs().mean().item()) < 1e-6
