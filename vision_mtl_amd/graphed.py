"""Product-side captured training step: the reference's train iteration (training_lit.py:81-98) at hipGraph
replay speed.

The reference drives the step surface from an eager Python loop::

    for batch in train_dataloader:
        optimizer.zero_grad()
        batch = module.transfer_batch_to_device(batch, device)
        loss = module.training_step(batch, 0)
        loss.backward()
        optimizer.step()

One step of `basic` is ~560 kernel launches of 4-25 us each: issued one by one from Python the step is host-bound
(bench.py reports both numbers: config.ms_per_step_eager vs ms_per_step).  `GraphedStep` captures forward + losses +
backward ONCE into a hipGraph over static input buffers and a FlatArena (gradients land in the arena's slots, which
stay bound to `p.grad`), then every call copies the new batch into the static buffers and replays::

    gstep = GraphedStep(module, example_batch)            # once, after module.to(device)
    for batch in train_dataloader:
        optimizer.zero_grad()
        loss = gstep(batch)                               # replaces transfer_batch_to_device + training_step
        loss.backward()                                   # no-op handle kept for source compatibility
        optimizer.step()

What stays outside the graph on purpose: the data-parallel gradient all-reduce (one RCCL collective, issued right
after the replay by the same end-of-backward routine the eager path uses) and the optimizer (torch.optim.Adam over
the arena's parameter views, or dp.ArenaAdam: one fused launch).  BatchNorm running statistics, the per-step packing
of the GEMM operands and the per-step metrics are kernels inside the graph.
"""
from __future__ import annotations

import torch

from . import dp, ops


class _Replayed(torch.autograd.Function):
    """The loss of a replayed step as a tensor with a grad_fn: `loss.backward()` in the caller's loop is accepted and
    does nothing (the captured backward pass already wrote every parameter gradient into the arena)."""

    @staticmethod
    def forward(ctx, value, anchor):
        return value.clone()

    @staticmethod
    def backward(ctx, g):
        return None, None


class GraphedStep:
    """fwd + losses + bwd of `module.training_step` captured once, replayed per batch.

    module: an MTLModule already on its device; example_batch: a batch of the shapes / dtypes / layout every later
    batch will have (host or device; an image in dataset sample layout (B,H,W,3) keeps the one-kernel re-layout of
    data.upload_batch inside the graph).  arena: an existing dp.FlatArena of module.model (default: module.dp_arena,
    else a new one - built here, so construct the GraphedStep on every rank).  warmup: eager steps before capture
    (allocator pools, code objects, packed-operand table).  NOTE: the warm-up steps and the capture rehearsal run real
    training steps on `example_batch` (BatchNorm running statistics move, no optimizer step is taken)."""

    def __init__(self, module, example_batch: dict, arena: "dp.FlatArena | None" = None, warmup: int = 2,
                 stage: str = "train"):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedStep needs an MI355X: the hot path has no CPU fallback")
        self.module, self.stage = module, stage
        self.device = next(module.model.parameters()).device
        if arena is None:
            arena = module.dp_arena if module.dp_arena is not None else dp.FlatArena(module.model)
        self.arena = arena
        self.static = {k: self._static_like(v) for k, v in example_batch.items()}
        self._fill(example_batch)
        self._sample_layout = (self.static["img"].dim() == 4 and self.static["img"].shape[-1] == 3
                               and self.static["img"].shape[1] != 3)
        attached, module.dp_arena = module.dp_arena, None  # the collective stays outside the graph (see __call__)
        try:
            so = module.step_outputs[stage]
            mark = {k: len(v) for k, v in so.items()}
            for _ in range(max(1, warmup)):
                self._step().backward()
            torch.cuda.synchronize(self.device)
            # rehearsal on a side stream (what torch.cuda.graph does internally needs the allocations of one step to
            # have happened on a non-default stream), then the capture itself
            s = _rehearsal_stream(self.device)
            s.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(s):
                self._step().backward()
            torch.cuda.current_stream(self.device).wait_stream(s)
            for k, v in so.items():
                del v[mark[k]:]
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                loss = self._step()
                loss.backward()
                # everything the step appended to step_outputs (loss + the four metrics), as ONE static vector
                self._keys = [k for k, v in so.items() if len(v) > mark[k]]
                self._stats = torch.stack([so[k][-1].detach().reshape(()).float() for k in self._keys])
                self._loss = loss.detach()
            for k, v in so.items():
                del v[mark[k]:]
        finally:
            module.dp_arena = attached
        self._anchor = torch.zeros((), device=self.device, requires_grad=True)
        self.replays = 0

    # ---- helpers
    def _static_like(self, v: torch.Tensor) -> torch.Tensor:
        return torch.empty(v.shape, dtype=torch.float32 if v.is_floating_point() else v.dtype, device=self.device)

    def _fill(self, batch: dict) -> None:
        for k, dst in self.static.items():
            src = batch[k]
            if tuple(src.shape) != tuple(dst.shape):
                raise ValueError(f"GraphedStep: batch[{k!r}] has shape {tuple(src.shape)}, the captured step expects "
                                 f"{tuple(dst.shape)} (capture one GraphedStep per batch shape; drop_last=True)")
            dst.copy_(src, non_blocking=src.device.type == "cpu" and src.is_pinned())

    def _step(self) -> torch.Tensor:
        batch = dict(self.static)
        if self._sample_layout:
            batch["img"] = ops.hwc_to_model_input(self.static["img"])
        return self.module.shared_step(batch, self.stage)

    # ---- the step
    def __call__(self, batch: dict) -> torch.Tensor:
        self._fill(batch)
        self.graph.replay()
        self.replays += 1
        if self.module.dp_arena is not None or dp.world_size() > 1:
            self.arena._end_of_backward()  # ONE all-reduce of the flat gradient (no-op on one rank)
        # optimizer.zero_grad() defaults to set_to_none=True in torch 2.x: re-bind .grad to the arena slots
        if self.arena.params[0].grad is None or self.arena.params[-1].grad is None:
            self.arena.rebind_grads()
        stats = self._stats.clone()  # one tiny copy: the static vector is overwritten by the next replay
        so = self.module.step_outputs[self.stage]
        for i, k in enumerate(self._keys):
            so[k].append(stats[i])
        return _Replayed.apply(stats[self._keys.index("loss")] if "loss" in self._keys else self._loss, self._anchor)


_STREAMS = {}


def _rehearsal_stream(device) -> torch.cuda.Stream:
    """One per device and process: HIP maps streams onto few hardware queues round-robin."""
    s = _STREAMS.get(device.index)
    if s is None:
        s = _STREAMS[device.index] = torch.cuda.Stream(device=device)
    return s
