"""-m gpu: the reference's train iteration (training_lit.py:82-87: training_step -> backward -> Adam.step)
run for several steps.  Step k+1 only matches if the optimizer's in-place parameter update reached the
packed GEMM operands (pack-cache invalidation through the parameter version / arena epoch), the BatchNorm
running buffers and Adam's own state - none of which a single-step parity test exercises.
Checker: the CPU oracle (oracle/mtan.py, pinned to the reference) driven by torch.optim.Adam."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
STEPS, LR = 4, 2e-3


def _scheduler(opt):
    # the reference's scheduler class with a trigger-happy setting so that 4 steps exercise it
    return torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="max", patience=0, factor=0.5)


def _oracle_losses(fx, batch, scheduled=False):
    from oracle.losses import step_losses
    from oracle.mtan import mtan_forward

    sd = {k: v.clone() for k, v in fx["state_dict"].items()}
    leaves = [v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k]
    opt = torch.optim.Adam(leaves, lr=LR)
    sched = _scheduler(opt) if scheduled else None
    out = []
    for _ in range(STEPS):
        raw = mtan_forward(sd, batch["img"], list(dict(fx["tasks"])), fx["cfg"]["levels"], training=True)
        loss = step_losses(raw, batch["mask"], batch["depth"])["loss"]
        opt.zero_grad()
        loss.backward()
        opt.step()
        if sched is not None:
            sched.step(float(loss.detach()))  # mode="max" on a falling loss: the rate halves every step
        out.append(float(loss.detach()))
    return out


def _model(fx, dev):
    from vision_mtl_amd.models.mtan_model import MTANMiniUnet

    c = fx["cfg"]
    m = MTANMiniUnet(3, dict(fx["tasks"]), c["hidden"], c["first"], c["levels"])
    m.load_state_dict(fx["state_dict"])
    return m.to(dev).train()


@pytest.mark.parametrize("mode", ["torch_adam", "arena_adam", "arena_adam_graph", "arena_optimizer_scheduler"])
def test_training_loop_matches_oracle(dev, mode):
    from vision_mtl_amd import dp
    from vision_mtl_amd.lit_module import MTLModule

    fx = torch.load(os.path.join(G, "mtan_tiny.pt"), weights_only=False)
    ref = _oracle_losses(fx, fx["batch"], scheduled=mode == "arena_optimizer_scheduler")
    assert ref[-1] < ref[0]  # the loop actually trains

    model = _model(fx, dev)
    module = MTLModule(model, num_classes=fx["cfg"]["C"], device=str(dev))
    batch = {k: v.to(dev) for k, v in fx["batch"].items()}
    got = []
    if mode == "torch_adam":  # exactly run_pipe(): parameters are ordinary nn.Parameters
        opt = torch.optim.Adam(module.parameters(), lr=LR)
        for _ in range(STEPS):
            loss = module.training_step(batch, 0)
            opt.zero_grad()
            loss.backward()
            opt.step()
            got.append(float(loss.detach()))
    else:  # data-parallel form: flat arena, gradients written into slots, one fused Adam launch
        arena = dp.FlatArena(model)
        module.dp_arena = arena

        def step():
            loss = module.training_step(batch, 0)
            loss.backward()
            return loss.detach()

        if mode == "arena_adam":
            for _ in range(STEPS):
                got.append(float(step()))
                arena.adam_step(lr=LR)
        elif mode == "arena_optimizer_scheduler":  # dp.ArenaAdam behind torch's scheduler + state_dict round trip
            opt = dp.ArenaAdam(arena, lr=LR)
            sched = _scheduler(opt)
            for k in range(STEPS):
                opt.zero_grad()  # the reference's order (training_lit.py:82-87): zero_grad, step, backward, optimizer.step
                loss = step()
                opt.step()
                sched.step(float(loss))
                got.append(float(loss))
                if k == 1:  # checkpoint / restore in the middle of the run
                    sd = opt.state_dict()
                    opt = dp.ArenaAdam(arena, lr=123.0)
                    opt.load_state_dict(sd)
                    sched2 = _scheduler(opt)
                    sched2.load_state_dict(sched.state_dict())
                    sched = sched2
            assert opt.param_groups[0]["lr"] < LR
        else:  # fwd + bwd replayed from a hipGraph, optimizer between replays
            got.append(float(step()))  # eager warm-up step (builds the packed-operand table)
            arena.adam_step(lr=LR)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_loss = step()
            # the capture itself does not execute: replay is step 2
            for _ in range(STEPS - 1):
                graph.replay()
                got.append(float(static_loss))
                arena.adam_step(lr=LR)
    for k, (a, b) in enumerate(zip(got, ref)):
        assert abs(a - b) <= 2e-4 * abs(b), f"{mode}: step {k} loss {a} vs oracle {b}"


def test_graphed_step_matches_eager_with_changing_batches(dev):
    """vision_mtl_amd.graphed.GraphedStep (the product-side captured step): 4 replays on 4 DIFFERENT batches, Adam
    between them, against the same loop run eagerly from the same start - losses, parameters after the run, the
    per-step metrics appended to step_outputs; optimizer.zero_grad() with torch's default set_to_none=True and a
    loss.backward() on the returned handle are both accepted (the reference loop, training_lit.py:81-98, keeps both)."""
    from vision_mtl_amd import dp
    from vision_mtl_amd.data import synthetic_batch
    from vision_mtl_amd.graphed import GraphedStep
    from vision_mtl_amd.lit_module import MTLModule

    fx = torch.load(os.path.join(G, "mtan_tiny.pt"), weights_only=False)
    C = fx["cfg"]["C"]
    B, _, H, W = fx["batch"]["img"].shape
    batches = [synthetic_batch(B, H, W, C, seed=100 + i, masked=0.1) for i in range(STEPS)]
    example = synthetic_batch(B, H, W, C, seed=99)

    def run(graphed):
        model = _model(fx, dev)
        module = MTLModule(model, num_classes=C, device=str(dev))
        arena = dp.FlatArena(model)
        opt = torch.optim.Adam(module.parameters(), lr=LR)  # torch's own optimizer over the arena's parameter views
        if graphed:
            gstep = GraphedStep(module, example, arena=arena)
            model.load_state_dict(fx["state_dict"])  # undo the BatchNorm-buffer drift of the warm-up / rehearsal steps
        else:  # the same warm-up history, eagerly (BatchNorm buffers are restored right after anyway)
            model.load_state_dict(fx["state_dict"])
        losses = []
        for b in batches:
            opt.zero_grad()  # set_to_none=True by default
            if graphed:
                loss = gstep(b)
            else:
                arena.rebind_grads()
                loss = module.training_step({k: v.to(dev) for k, v in b.items()}, 0)
            loss.backward()
            opt.step()
            dp.ops.packs.invalidate()
            losses.append(float(loss.detach()))
        so = {k: [float(v) for v in vals] for k, vals in module.step_outputs["train"].items()}
        return losses, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, so

    le, sde, soe = run(False)
    lg, sdg, sog = run(True)
    assert le[-1] != le[0]
    for k, (a, b) in enumerate(zip(lg, le)):
        assert abs(a - b) <= 1e-5 * abs(b), f"step {k}: replayed loss {a} vs eager {b}"
    for k in sde:
        if sde[k].is_floating_point():
            assert_close_(sdg[k], sde[k], k)
    assert len(sog["loss"]) == STEPS and all(len(v) == STEPS for v in sog.values())
    for k in soe:
        for a, b in zip(sog[k], soe[k]):
            assert (a != a and b != b) or abs(a - b) <= 1e-5 * max(abs(b), 1e-6), f"step_outputs[{k}]: {a} vs {b}"


def assert_close_(a, b, what):
    err, ref = float((a.double() - b.double()).abs().max()), float(b.double().abs().max())
    assert err <= 1e-5 * ref + 1e-8, f"{what}: {err:.3e} vs magnitude {ref:.3e}"
