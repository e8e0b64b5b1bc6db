"""-m gpu: gradients written straight into dp.FlatArena slots (the data-parallel path: one flat buffer,
one all-reduce) must equal the gradients autograd accumulates without an arena - bit for bit, because
the same kernels run in the same order per tensor - whether the parameter-gradient launches stay on the
main stream, run on the side stream (ops.side), or the whole step is replayed from a hipGraph."""
import argparse

import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(dev, name):
    from vision_mtl_amd.utils.pipeline_utils import build_model

    torch.manual_seed(5)
    args = argparse.Namespace(model_name=name, backbone_weights=None)
    return build_model(args, argparse.Namespace(num_classes=19)).to(dev).train()


@pytest.mark.parametrize("name", ["basic", "mtan", "csnet"])
def test_arena_slots_side_stream_and_graph(dev, name):
    from oracle.losses import synthetic_batch
    from vision_mtl_amd import dp, ops
    from vision_mtl_amd.lit_module import MTLModule

    model = _build(dev, name)
    module = MTLModule(model, num_classes=19, device=str(dev))
    batch = {k: v.to(dev) for k, v in synthetic_batch(2, 64, 96, 19, seed=3, masked=0.1).items()}
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}

    def step():
        model.load_state_dict(sd0)  # same BatchNorm running buffers every time
        ops.packs.invalidate()
        loss = module.training_step(batch, 0)
        loss.backward()
        return loss.detach().clone()

    loss_ref = step()
    params = [p for p in model.parameters() if p.requires_grad]
    used = [p.grad is not None for p in params]  # CSNet holds stitch layers for non-stitch sites: never used
    assert name == "csnet" or all(used)
    ref = [p.grad.clone() if u else torch.zeros_like(p) for p, u in zip(params, used)]
    for p in model.parameters():
        p.grad = None

    arena = dp.FlatArena(model)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    flat_ref = torch.cat([g.reshape(-1) for g in ref])
    live = torch.cat([torch.full((p.numel(),), u, dtype=torch.bool) for p, u in zip(params, used)]).to(dev)

    def same(flat):  # slots of parameters the step never touches keep whatever they held
        return torch.equal(flat[live], flat_ref[live])
    was = ops.side.enabled
    try:
        for enabled in (False, True):
            ops.side.enabled = enabled
            # every slot must be overwritten, not accumulated into (CSNet: the off-diagonal stitch entries are
            # never written and keep the arena's initial zero, as their reference gradient is zero)
            arena.flat_grad.fill_(0.0 if name == "csnet" else float("nan"))
            arena.slots_clobbered()  # written behind the arena's back: structurally-zero slots get re-zeroed
            loss = step()
            torch.cuda.synchronize()
            assert ops.side.pending is None
            assert torch.equal(loss, loss_ref)
            assert same(arena.flat_grad), f"side stream {enabled}: slot gradients differ"
        # whole step as a hipGraph with the side-stream fork/join captured as parallel branches
        ops.side.enabled = True
        step()  # warm: packed-operand table, side stream exist before capture
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_loss = step()
        for _ in range(2):
            arena.flat_grad.fill_(0.0 if name == "csnet" else float("nan"))
            # slots the arena KNOWS to be zero (biases in front of a train-mode BatchNorm, ops._bias_grad) are not
            # rewritten by the captured step: keep their zero, poison everything else
            for p in arena.params:
                if id(p) in arena._zero_bias:
                    p._vmtl_gslot.zero_()
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(static_loss, loss_ref)
            assert same(arena.flat_grad), "hipGraph replay: slot gradients differ"
    finally:
        ops.side.enabled = was
