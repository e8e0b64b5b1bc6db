"""-m "not gpu": the CPU oracle (oracle/*.py) against the golden vectors produced by the REAL
reference (oracle/gen_golden.py, committed under tests/golden/).  This is what pins the oracle."""
import os

import pytest
import torch
import torch.nn.functional as F

from tests.util import assert_close

G = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return torch.load(os.path.join(G, name), weights_only=False)


@pytest.mark.parametrize("name", ["mtan_tiny.pt", "mtan_small3.pt"])
def test_mtan_oracle_matches_reference(name):
    from oracle.losses import step_losses
    from oracle.mtan import mtan_forward

    fx = _load(name)
    sd = {k: v.clone() for k, v in fx["state_dict"].items()}
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    tasks = list(fx["tasks"])
    out = mtan_forward(sd, fx["batch"]["img"], tasks, fx["cfg"]["levels"], training=True)
    losses = step_losses(out, fx["batch"]["mask"], fx["batch"]["depth"])
    losses["loss"].backward()
    for t in tasks:
        assert_close(out[t].detach(), fx["out_train"][t], tol=1e-5, what=f"train out {t}")
    assert_close(losses["loss"].detach(), fx["loss"], tol=1e-5, what="loss")
    gscale = max(float(g.abs().max()) for g in fx["grads"].values())
    for k, g in fx["grads"].items():
        assert_close(leaves[k].grad, g, tol=2e-4, atol=1e-6 * gscale, what=f"grad {k}")
    for k, v in fx["state_dict_after"].items():
        if "running" in k:
            assert_close(sd[k].detach(), v, tol=1e-5, what=k)
        elif "num_batches" in k:
            assert int(sd[k]) == int(v)
    with torch.no_grad():
        oe = mtan_forward({k: v.clone() for k, v in fx["state_dict_after"].items()}, fx["batch"]["img"], tasks,
                          fx["cfg"]["levels"], training=False)
    for t in tasks:
        assert_close(oe[t], fx["out_eval"][t], tol=1e-5, what=f"eval out {t}")


def test_mirror_init_equals_reference_init():
    """Same seed -> the host-side mirror creates exactly the reference's parameters."""
    from vision_mtl_amd.models.mtan_model import MTANMiniUnet

    fx = _load("mtan_tiny.pt")
    c = fx["cfg"]
    torch.manual_seed(fx["seed"])
    m = MTANMiniUnet(3, dict(fx["tasks"]), c["hidden"], c["first"], c["levels"])
    sd = m.state_dict()
    assert list(fx["tasks"]) == list(m.map_tasks_to_heads.keys())
    assert set(sd) == set(fx["state_dict"])
    for k, v in fx["state_dict"].items():
        assert torch.equal(sd[k], v), k


def test_component_oracles():
    from oracle.losses import silog
    from oracle.mtan import pad_concat

    fx = _load("components.pt")
    for tag in ("valid", "masked"):
        f = fx[f"silog_{tag}"]
        z = f["z"].clone().requires_grad_(True)
        l = silog(torch.sigmoid(z).permute(0, 2, 3, 1), f["t"])
        l.backward()
        assert_close(l.detach(), f["loss"], tol=1e-6, what="silog")
        assert_close(z.grad, f["dz"], tol=1e-5, what="silog grad")
    for C in (19, 14):
        f = fx[f"ce_{C}"]
        z = f["z"].clone().requires_grad_(True)
        l = F.cross_entropy(z, f["t"])
        l.backward()
        assert_close(l.detach(), f["loss"], tol=1e-6, what="ce")
        assert_close(z.grad, f["dz"], tol=1e-5, what="ce grad")
    f = fx["pad_concat"]
    assert torch.equal(pad_concat(f["x1"], f["x2"]), f["y"])
    for cw in (0, 1):
        f = fx[f"stitch_cw{cw}"]
        w = f["w"]
        diag = torch.stack([w[a, a] for a in range(w.shape[0])])  # (T[,C]) — only the diagonal acts
        y = f["x"] * (diag[:, None, :, None, None] if cw else diag[:, None, None, None, None])
        assert_close(y, f["y"], tol=1e-6, what="stitch = diagonal scale")
        assert float(f["dw"][0, 1].abs().max()) == 0.0 and float(f["dw"][1, 0].abs().max()) == 0.0
