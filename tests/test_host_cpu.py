"""-m "not gpu": host logic around the hot path that needs no GPU -
checkpoint wire format (reference utils/pipeline_utils.py:139-167,207-244, init_model :22-30), construction
manifests (parameter totals the README / SURVEY quote), the Adam state wire format of dp.ArenaAdam, the dataset
sample contract of vision_mtl_amd/data.py (reference data_modules/cityscapes.py:39-67), fetch_data_cfg."""
import argparse
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


def _args(name, **kw):
    return argparse.Namespace(model_name=name, backbone_weights=None, **kw)


# ------------------------------------------------------------------------------------------ construction
@pytest.mark.parametrize("name,total,stitch", [("basic", 13_564_301, 0), ("csnet", 13_374_260, 8_256), ("mtan", 13_277_743, 0)])
def test_parameter_totals(name, total, stitch):
    """reference README.md:134 ("approximately 13.3M parameters" for all three); exact counts: mtan measured on the
    real reference, basic / csnet from the published smp / timm architecture (Appendix A.3)."""
    from vision_mtl_amd.utils.pipeline_utils import build_model

    torch.manual_seed(0)
    C = 14 if name == "mtan" else 19
    m = build_model(_args(name, channel_wise_stitching=True), argparse.Namespace(num_classes=C))
    n = sum(p.numel() for p in m.parameters())
    n_stitch = sum(p.numel() for k, p in m.named_parameters() if "stitch" in k)
    assert n - n_stitch == total, (name, n, n_stitch)
    if name == "csnet":
        # 11 stitch sites x (T=2, T=2, C) channel-wise matrices (only their diagonals are used): 4 * 2064 channels
        assert n_stitch == stitch, n_stitch


def test_basic_state_dict_manifest():
    """Key names / shapes smp 0.3.3 + timm 0.9.2 produce for Backbone + the two heads (spot checks on every family)."""
    from vision_mtl_amd.utils.pipeline_utils import build_model

    sd = build_model(_args("basic"), argparse.Namespace(num_classes=19)).state_dict()
    expect = {
        "backbone.encoder.model.conv_stem.weight": (16, 3, 3, 3),
        "backbone.encoder.model.bn1.running_var": (16,),
        "backbone.encoder.model.blocks.0.0.conv_dw.weight": (16, 1, 3, 3),
        "backbone.encoder.model.blocks.2.0.se.conv_reduce.weight": (24, 72, 1, 1),
        "backbone.encoder.model.blocks.2.0.conv_dw.weight": (72, 1, 5, 5),
        "backbone.encoder.model.blocks.5.2.conv_pwl.weight": (160, 960, 1, 1),
        "backbone.encoder.model.blocks.6.0.conv.weight": (960, 160, 1, 1),
        "backbone.decoder.blocks.0.conv1.0.weight": (540, 1072, 3, 3),
        "backbone.decoder.blocks.1.conv1.0.weight": (270, 580, 3, 3),
        "backbone.decoder.blocks.4.conv2.0.weight": (33, 33, 3, 3),
        "backbone.decoder.blocks.4.conv2.1.num_batches_tracked": (),
        "segm_head.0.weight": (19, 33, 3, 3),
        "segm_head.0.bias": (19,),
        "depth_head.0.weight": (1, 33, 3, 3),
    }
    for k, shape in expect.items():
        assert k in sd, k
        assert tuple(sd[k].shape) == shape, (k, tuple(sd[k].shape))
    enc = sum(v.numel() for k, v in sd.items() if k.startswith("backbone.encoder") and "running" not in k and "num_batches" not in k)
    assert enc == 2_971_952  # MobileNetV3-Large features (5,483,032 - conv_head - classifier)


def test_mtan_keys_match_reference_golden():
    from vision_mtl_amd.utils.pipeline_utils import build_model

    fx = torch.load(os.path.join(G, "mtan_tiny.pt"), weights_only=False)
    sd = build_model(_args("mtan"), argparse.Namespace(num_classes=14)).state_dict()
    assert set(sd) == set(fx["state_dict"])  # same module tree as the real reference (only widths differ)
    assert len(sd) == 512


def test_fetch_data_cfg():
    from vision_mtl_amd.utils.pipeline_utils import fetch_data_cfg

    c, n = fetch_data_cfg("cityscapes"), fetch_data_cfg("nyuv2")
    assert (c.num_classes, c.height, c.width) == (19, 128, 256)  # reference cfg.py:68-71
    assert (n.num_classes, n.height, n.width) == (14, 256, 256)  # reference cfg.py:124,147
    with pytest.raises(ValueError):
        fetch_data_cfg("kitti")


# ------------------------------------------------------------------------------------------ checkpoints
def test_ckpt_round_trip_and_init_model(tmp_path):
    from vision_mtl_amd.utils import ckpt
    from vision_mtl_amd.utils.pipeline_utils import init_model

    torch.manual_seed(3)
    module = init_model(_args("mtan", lr=5e-4), argparse.Namespace(num_classes=5))
    assert all(k.startswith("model.") for k in module.state_dict())  # reference: keys prefixed "model."
    opt = torch.optim.Adam(module.parameters(), lr=5e-4)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, patience=2, factor=0.9)
    for epoch in (3, 12):
        with torch.no_grad():
            for p in module.parameters():
                p.add_(0.01 * epoch)
        ckpt.save_ckpt(module, opt, sched, epoch, str(tmp_path / f"model_{epoch}.pt"), str(tmp_path / f"session_{epoch}.pt"))
    want = {k: v.clone() for k, v in module.state_dict().items()}
    (tmp_path / "notes.txt").write_text("not a checkpoint")
    got = ckpt.load_ckpt_model(str(tmp_path))["model"]  # highest epoch: 12, not the lexicographic "3"
    assert set(got) == set(want) and all(torch.equal(got[k], want[k]) for k in want)
    old = ckpt.load_ckpt_model(str(tmp_path), epoch=3)["model"]
    assert not torch.equal(old["model.bottleneck.double_conv.0.weight"], want["model.bottleneck.double_conv.0.weight"])
    session, model = ckpt.load_ckpt(str(tmp_path))
    assert session["epoch"] == 12 and set(session) == {"optimizer", "scheduler", "epoch"}
    assert torch.equal(model["model"]["model.bottleneck.double_conv.0.weight"], want["model.bottleneck.double_conv.0.weight"])
    # init_model(ckpt_dir=...) restores the weights (reference pipeline_utils.py:28-29)
    torch.manual_seed(99)
    restored = init_model(_args("mtan", lr=5e-4, ckpt_dir=str(tmp_path)), argparse.Namespace(num_classes=5))
    rs = restored.state_dict()
    assert all(torch.equal(rs[k], want[k]) for k in want)
    (tmp_path / "empty").mkdir()
    with pytest.raises(ValueError, match="No model ckpt found"):
        ckpt.load_ckpt_model(str(tmp_path / "empty"))


def test_reference_wire_format_loads(tmp_path):
    """A state_dict produced by the REAL reference (tests/golden/mtan_tiny.pt), written the way the reference's
    save_ckpt writes it, loads through init_model's path (key names, shapes, buffers all line up)."""
    from vision_mtl_amd.lit_module import MTLModule
    from vision_mtl_amd.models.mtan_model import MTANMiniUnet
    from vision_mtl_amd.utils import ckpt

    fx = torch.load(os.path.join(G, "mtan_tiny.pt"), weights_only=False)
    torch.save({"model": {f"model.{k}": v for k, v in fx["state_dict_after"].items()}}, tmp_path / "model_7.pt")
    c = fx["cfg"]
    module = MTLModule(MTANMiniUnet(3, dict(fx["tasks"]), c["hidden"], c["first"], c["levels"]), num_classes=c["C"])
    missing = module.load_state_dict(ckpt.load_ckpt_model(str(tmp_path))["model"])
    assert not missing.missing_keys and not missing.unexpected_keys
    sd = module.model.state_dict()
    assert all(torch.equal(sd[k], v) for k, v in fx["state_dict_after"].items())


# ------------------------------------------------------------------------------------------ optimizer state
def test_arena_adam_state_dict_is_torch_adams(tmp_path):
    """dp.ArenaAdam <-> torch.optim.Adam through the reference's session file (pipeline_utils.py:139-167)."""
    from vision_mtl_amd import dp
    from vision_mtl_amd.models.mtan_model import MTANMiniUnet

    torch.manual_seed(1)
    model = MTANMiniUnet(3, {"depth": 1, "segm": 3}, 8, 4, 2)
    ref_opt = torch.optim.Adam(model.parameters(), lr=5e-4)
    g = torch.Generator().manual_seed(2)
    for _ in range(3):
        for p in model.parameters():
            p.grad = torch.randn(p.shape, generator=g)
        ref_opt.step()
    sd = ref_opt.state_dict()
    arena = dp.FlatArena(model)
    opt = dp.ArenaAdam(arena, lr=123.0)
    opt.load_state_dict(sd)
    assert opt.param_groups[0]["lr"] == 5e-4 and opt.param_groups[0]["params"][0] is arena.flat_param
    assert float(arena._adam["step"]) == 3.0
    flat_m = torch.cat([sd["state"][i]["exp_avg"].reshape(-1) for i in range(len(arena.params))])
    flat_v = torch.cat([sd["state"][i]["exp_avg_sq"].reshape(-1) for i in range(len(arena.params))])
    assert torch.equal(arena._adam["m"], flat_m) and torch.equal(arena._adam["v"], flat_v)
    out = opt.state_dict()
    assert set(out) == {"state", "param_groups"} and out["param_groups"][0]["params"] == list(range(len(arena.params)))
    assert out["param_groups"][0]["lr"] == 5e-4
    for i in range(len(arena.params)):
        for k in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(out["state"][i][k], sd["state"][i][k])
        assert float(out["state"][i]["step"]) == 3.0
    # and back into a fresh torch.optim.Adam (what the reference would do with our session file)
    torch.save({"optimizer": out, "epoch": 1}, tmp_path / "session_1.pt")
    again = torch.optim.Adam(model.parameters(), lr=1.0)
    again.load_state_dict(torch.load(tmp_path / "session_1.pt", weights_only=False)["optimizer"])
    assert again.param_groups[0]["lr"] == 5e-4
    assert torch.equal(again.state[arena.params[5]]["exp_avg"], sd["state"][5]["exp_avg"])
    # foreign / partial formats fail loudly instead of dropping the moments
    with pytest.raises(ValueError):
        opt.load_state_dict({"param_groups": sd["param_groups"], "arena": {}})
    bad = {"state": {0: sd["state"][0]}, "param_groups": sd["param_groups"]}
    with pytest.raises(ValueError):
        opt.load_state_dict(bad)
    fewer = {"state": sd["state"], "param_groups": [dict(sd["param_groups"][0], params=[0, 1])]}
    with pytest.raises(ValueError):
        opt.load_state_dict(fewer)


def test_arena_mixed_gradient_paths_raise():
    """a parameter must not get a kernel-written slot AND an autograd-accumulated gradient in one pass."""
    from vision_mtl_amd import dp, ops

    lin = torch.nn.Linear(4, 3)
    arena = dp.FlatArena(lin)
    x = torch.randn(2, 4)
    (lin(x).sum() + lin.weight.pow(2).sum()).backward()  # ordinary autograd only: accumulates into the arena views
    assert float(arena.flat_grad.abs().sum()) > 0
    arena.zero_grad()
    assert float(arena.flat_grad.abs().sum()) == 0.0
    assert ops._slot(lin.weight) is lin.weight.grad  # what a HIP backward Function does in its forward
    with pytest.raises(RuntimeError, match="ordinary autograd AND"):
        lin(x).sum().backward()


# ------------------------------------------------------------------------------------------ input contract
def test_prepare_sample_and_collate_follow_the_reference_contract():
    from vision_mtl_amd import data

    rng = np.random.default_rng(0)
    H, W, C = 6, 10, 19
    raw = {"img": rng.random((H, W, 3), dtype=np.float32), "mask": rng.integers(-1, C - 1, (H, W)).astype(np.float32),
           "depth": rng.random((H, W, 1), dtype=np.float32) * 0.49}
    raw["mask"][0, 0] = -1
    s = data.prepare_sample(raw, num_classes=C)
    assert s["img"].dtype == torch.float32 and tuple(s["img"].shape) == (H, W, 3)      # stays HWC on the host
    assert s["mask"].dtype == torch.int64 and int(s["mask"][0, 0]) == C - 1            # cityscapes.py:42
    assert int(s["mask"].min()) >= 0 and tuple(s["depth"].shape) == (H, W, 1)          # notebook: depth (128,256,1)
    assert torch.equal(s["depth"], torch.from_numpy(raw["depth"]))                     # max <= 1: not rescaled
    # NYUv2 sample (nyuv2.py:100-141; the module itself needs h5py / torchvision, absent offline, so the expectations are
    # written out from its rules): 8-bit image -> /255; a class-id mask that ToTensor scaled by 1/255 -> back to ids;
    # depth = uint16 PNG counts / 1e4 (metres), then / max_depth (cfg.py NYUv2Config.max_depth = 10) because max > 1
    ids = rng.integers(0, 14, (H, W))
    counts = rng.integers(5000, 60000, (H, W)).astype(np.uint16)
    raw2 = {"img": rng.integers(0, 256, (H, W, 3)).astype(np.float32), "mask": (ids / 255.0).astype(np.float32)[None],
            "depth": counts.astype(np.int32)[None]}
    s2 = data.prepare_sample(raw2, num_classes=14, max_depth=10.0, dataset="nyuv2")
    assert float(s2["img"].max()) <= 1.0 and torch.allclose(s2["img"], torch.from_numpy(raw2["img"]) / 255)
    assert s2["mask"].dtype == torch.int64 and torch.equal(s2["mask"], torch.from_numpy(ids))
    assert tuple(s2["depth"].shape) == (H, W, 1)
    assert torch.allclose(s2["depth"][..., 0], torch.from_numpy(counts.astype(np.float32)) / 1e4 / 10.0)
    assert 0.05 <= float(s2["depth"].min()) and float(s2["depth"].max()) <= 0.6   # SILog-range targets, not 1e3..1e4
    # an integer-id mask is left alone, and without the dataset name the uint16 depth would NOT be rescaled by 1e4:
    s3 = data.prepare_sample({"img": raw2["img"], "mask": ids, "depth": counts.astype(np.float32)}, 14, 10.0, dataset="nyuv2")
    assert torch.equal(s3["mask"], torch.from_numpy(ids)) and torch.allclose(s3["depth"], s2["depth"])
    s4 = data.prepare_sample({"img": raw["img"], "mask": ids, "depth": counts.astype(np.float32)}, 14, 10.0)
    assert float(s4["depth"].min()) >= 500.0  # cityscapes rule on the same numbers: only / max_depth
    with pytest.raises(ValueError):
        data.prepare_sample(raw2, 14, 10.0, dataset="kitti")
    b = data.collate([s, s], pin=False)
    assert tuple(b["img"].shape) == (2, H, W, 3) and tuple(b["mask"].shape) == (2, H, W) and tuple(b["depth"].shape) == (2, H, W, 1)
    with pytest.raises(ValueError):
        data.prepare_sample({"img": raw["img"].transpose(2, 0, 1), "mask": raw["mask"], "depth": raw["depth"]}, C)
    sb, ob = data.synthetic_batch(2, 8, 8, 5, seed=4, masked=0.2), __import__("oracle.losses", fromlist=["x"]).synthetic_batch(2, 8, 8, 5, seed=4, masked=0.2)
    assert all(torch.equal(sb[k], ob[k]) for k in sb)  # product-side generator == the oracle's (same seeds, same draws)


def test_flat_arena_does_not_keep_the_model_alive():
    """The arena's gradient hooks sit in C++ autograd metadata (invisible to the cycle collector): they must not hold
    the arena or the parameters strongly, or every model ever given an arena stays alive with its packed operands."""
    import gc
    import weakref

    import torch

    from vision_mtl_amd import dp

    model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4))
    arena = dp.FlatArena(model, broadcast=False)
    wm, wa, wp = weakref.ref(model), weakref.ref(arena), weakref.ref(next(model.parameters()))
    del model, arena
    gc.collect()
    assert wm() is None and wa() is None and wp() is None

    model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3))
    arena = dp.FlatArena(model, broadcast=False)
    p = next(model.parameters())
    assert getattr(p, "_vmtl_gslot", None) is not None
    arena.close()
    assert not hasattr(p, "_vmtl_gslot") and not hasattr(p, "_vmtl_arena")
    (model(torch.randn(1, 3, 5, 5)).sum()).backward()  # ordinary autograd accumulation into the (still valid) .grad view
    assert p.grad is not None and float(p.grad.abs().sum()) > 0
