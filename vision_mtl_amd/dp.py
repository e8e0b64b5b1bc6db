"""Data-parallel layer: one process per GPU, parameters replicated, gradients in ONE flat fp32
buffer, ONE all-reduce per step over RCCL/xGMI (torch.distributed backend "nccl" is RCCL on ROCm).

The reference has no distributed code at all — this is new, MI355X-first work:
images shard across ranks, every rank runs the same fwd/bwd on its shard with per-replica
BatchNorm statistics (what DDP does by default), gradients are averaged.  ~13.5 M parameters =
54 MB: a single collective per step (no bucketing: on 7 x 153 GB/s point-to-point xGMI links one
54 MB ring all-reduce is ~0.6 ms against a ~15 ms step, and fewer, larger collectives are the right
shape for this fabric).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from . import ops


class FlatArena:
    """Re-homes every parameter of `model` into one contiguous fp32 buffer and gives each a slot in
    a second contiguous gradient buffer.  The backward kernels write parameter gradients directly
    into the slots (see ops._slot), so after loss.backward() `flat_grad` holds the whole gradient.

    Use from a driver loop that owns the optimizer step (bench.py, Trainer below).  Do not combine
    with optimizer.zero_grad(set_to_none=True): .grad must stay bound to the arena."""

    def __init__(self, model: torch.nn.Module):
        params = [p for p in model.parameters() if p.requires_grad]
        if not params:
            raise ValueError("model has no trainable parameters")
        dev = params[0].device  # build the arena AFTER model.to(device): .to() would drop the views
        total = sum(p.numel() for p in params)
        self.flat_param = torch.empty(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.params, off = params, 0
        for p in params:
            n = p.numel()
            slot = self.flat_param[off:off + n].view(p.shape)
            slot.copy_(p.data)
            p.data = slot
            p.grad = self.flat_grad[off:off + n].view(p.shape)
            p._vmtl_gslot = p.grad
            off += n
        self.numel = total
        self._adam = None

    def all_reduce_mean(self):
        """Average the flat gradient over all ranks: one RCCL all-reduce."""
        ops.side.join()  # no-op unless a backward pass ended abnormally with side-stream work un-joined
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM)
            return 1.0 / dist.get_world_size()
        return 1.0

    def sync_loss(self, loss: torch.Tensor) -> torch.Tensor:
        """Return `loss` with the gradient averaging hooked onto its backward pass: when the autograd
        engine finishes `loss.backward()` (engine callback, queued from the root of the graph) the side
        stream is joined, the flat gradient is all-reduced once over RCCL and scaled by 1/world - so the
        reference's `loss.backward(); optimizer.step()` (training_lit.py:85-87) needs no change
        (last row).  MTLModule does this for the train stage when `dp_arena` is set."""
        return _SyncGrads.apply(loss, self)

    def _end_of_backward(self):
        scale = self.all_reduce_mean()
        if scale != 1.0:
            self.flat_grad.mul_(scale)

    def adam_step(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
        """torch.optim.Adam semantics (reference training_lit.py:51,87) as ONE fused launch over the arena."""
        if self._adam is None:
            z = torch.zeros_like(self.flat_param)
            self._adam = {"m": z, "v": z.clone(), "step": torch.zeros(1, dtype=torch.float32, device=z.device)}
        ops.side.join()
        st = self._adam
        st["step"] += 1
        ops.adam_step(self.flat_param, self.flat_grad, st["m"], st["v"], st["step"], lr, betas, eps, weight_decay,
                      grad_scale)
        ops.packs.invalidate()  # parameters changed through raw pointers: packed operands are stale


class ArenaAdam(torch.optim.Optimizer):
    """torch.optim.Adam's interface over FlatArena.adam_step (one fused launch over the flat buffers):
    `param_groups`, `state_dict` / `load_state_dict` and `zero_grad` behave like an optimizer's, so the
    reference's scheduler (`ReduceLROnPlateau`, training_lit.py:51-55) and checkpoint writer
    (`save_ckpt`, pipeline_utils.py:139-167) drive the arena path unchanged."""

    def __init__(self, arena: "FlatArena", lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.arena = arena
        super().__init__([arena.flat_param], dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        self.arena.adam_step(lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"], weight_decay=g["weight_decay"])
        return loss

    def zero_grad(self, set_to_none: bool = False):
        """Nothing to clear: every backward pass overwrites the gradient slots (it never accumulates)."""

    def state_dict(self):
        st = self.arena._adam
        return {"param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups],
                "arena": None if st is None else {k: v.clone() for k, v in st.items()}}

    def load_state_dict(self, sd):
        for g, saved in zip(self.param_groups, sd["param_groups"]):
            g.update(saved)
        if sd.get("arena") is not None:
            dev = self.arena.flat_param.device
            self.arena._adam = {k: v.to(dev).clone() for k, v in sd["arena"].items()}


class _SyncGrads(torch.autograd.Function):
    """Identity on the loss; its backward (the first node the engine runs) queues FlatArena._end_of_backward."""

    @staticmethod
    def forward(ctx, loss, arena):
        ctx.arena = arena
        return loss.view_as(loss)

    @staticmethod
    def backward(ctx, g):
        torch.autograd.Variable._execution_engine.queue_callback(ctx.arena._end_of_backward)
        return g, None


def init_distributed():
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run contract).  Returns
    (rank, world, local_rank); initialises RCCL only when world > 1."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    return rank, world, local_rank


def shard_batch(batch: dict, rank: int, world: int) -> dict:
    """Equal contiguous shards of the leading (image) axis; the global batch must divide evenly."""
    out = {}
    for k, v in batch.items():
        if v.shape[0] % world:
            raise ValueError(f"global batch {v.shape[0]} does not divide over {world} ranks")
        n = v.shape[0] // world
        out[k] = v[rank * n:(rank + 1) * n]
    return out
