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

import weakref

import torch
import torch.distributed as dist

from . import ops


class FlatArena:
    """Re-homes every parameter of `model` into one contiguous fp32 buffer and gives each a slot in
    a second contiguous gradient buffer.  The backward kernels write parameter gradients directly
    into the slots (see ops._slot), so after loss.backward() `flat_grad` holds the whole gradient.

    Replica consistency: when torch.distributed is initialised the flat parameter buffer and every
    floating-point buffer (BatchNorm running statistics) are broadcast from rank 0 here, so ranks
    whose models were initialised under different seeds start identical (what DDP does in its
    constructor).  BatchNorm statistics are per replica afterwards (PyTorch-DDP default; the reference is
    single-device, there is no SyncBN to mirror): running buffers drift apart by the shard statistics and
    `broadcast_buffers()` re-aligns them to rank 0's (call it before evaluating / checkpointing).

    Gradient semantics: a backward pass OVERWRITES the slot of every parameter its kernels compute a
    gradient for (it does not accumulate: one fwd+bwd per optimizer step, as the reference's loop,
    training_lit.py:82-87).  Gradients that reach an arena parameter through ordinary autograd (a
    torch-native loss term on a parameter) are added in place by AccumulateGrad instead; a parameter
    that gets BOTH kinds in one backward pass would depend on their order, so that raises.  zero_grad()
    is one memset of the flat buffer (needed only for the autograd-accumulated kind).

    Use from a driver loop that owns the optimizer step (bench.py, ArenaAdam below).  Do not combine
    with optimizer.zero_grad(set_to_none=True): .grad must stay bound to the arena (rebind_grads() re-attaches it).

    COLLECTIVE CONSTRUCTOR: with an initialised process group, FlatArena(model) broadcasts rank 0's parameters and
    buffers (broadcast=True, the default) - EVERY rank must construct its arena, in the same order, or the ranks that do
    wait forever.  Pass broadcast=False for a rank-local arena (and call broadcast_parameters() yourself later)."""

    def __init__(self, model: torch.nn.Module, broadcast: bool = True):
        params = [p for p in model.parameters() if p.requires_grad]
        if not params:
            raise ValueError("model has no trainable parameters")
        dev = params[0].device  # build the arena AFTER model.to(device): .to() would drop the views
        total = sum(p.numel() for p in params)
        self.flat_param = torch.empty(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.params, self.offsets, off = params, [], 0
        self._kernel_written = set()  # ids of parameters whose slot a HIP backward kernel writes (ops._slot); per step
        self._zero_bias = set()       # ids of bias parameters whose slot is KNOWN to hold zeros (see ops._bias_grad)
        self._scale_consumer = None   # weak reference to the ArenaAdam that folds pending_scale into its update
        self._hooks = []
        for p in params:
            n = p.numel()
            slot = self.flat_param[off:off + n].view(p.shape)
            slot.copy_(p.data)
            p.data = slot
            p.grad = self.flat_grad[off:off + n].view(p.shape)
            p._vmtl_gslot = p.grad
            p._vmtl_arena = self
            self._hooks.append(p.register_hook(self._make_mixed_use_guard(p)))
            self.offsets.append(off)
            off += n
        self.numel = total
        self.model = model
        self._adam = None
        self.pending_scale = 1.0  # 1/world still to be applied to flat_grad (see all_reduce_mean / ArenaAdam)
        if broadcast:
            self.broadcast_parameters()

    # ---- replica consistency
    def broadcast_parameters(self, src: int = 0) -> None:
        """Rank `src`'s parameters and floating-point buffers to every rank (no-op without a process group)."""
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        dist.broadcast(self.flat_param, src=src)
        self.broadcast_buffers(src)
        ops.packs.invalidate()

    def broadcast_buffers(self, src: int = 0) -> None:
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        for b in self.model.buffers():
            dist.broadcast(b, src=src)

    def checksum(self) -> float:
        """Cheap cross-rank consistency probe: max over ranks of |sum(params) - rank 0's sum| (0.0 when identical)."""
        s = self.flat_param.double().sum().reshape(1)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            ref = s.clone()
            dist.broadcast(ref, src=0)
            d = (s - ref).abs()
            dist.all_reduce(d, op=dist.ReduceOp.MAX)
            return float(d.item())
        return 0.0

    # ---- gradient bookkeeping
    def _make_mixed_use_guard(self, p):
        # The hook lives in the tensor's C++ autograd metadata, where Python's cycle collector cannot see it: a closure
        # holding the arena (or p) strongly would keep arena, parameters and model alive for the life of the process
        # (and with them their packed operands, re-packed every step: bench.py's later configurations measured up to
        # 1 ms/step slower).  It holds a weak reference and the parameter's id instead.
        wself, pid = weakref.ref(self), id(p)

        def guard(grad):
            arena = wself()
            if arena is not None and grad is not None and pid in arena._kernel_written:  # kernels hand autograd None
                raise RuntimeError(
                    "a parameter of the FlatArena receives a gradient through ordinary autograd AND from a HIP "
                    "backward kernel that overwrites its slot in the same backward pass: the result would depend on "
                    "their order.  Keep torch-native loss terms off parameters the vmtl kernels differentiate.")
            return grad
        return guard

    def close(self) -> None:
        """Detach from the model: hooks removed, kernels stop writing into the slots (the parameters keep their storage
        inside the flat buffer and their .grad views, which stay valid tensors)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self.params:
            for attr in ("_vmtl_gslot", "_vmtl_arena"):
                if hasattr(p, attr):
                    delattr(p, attr)
        self._scale_consumer = None
        self._kernel_written.clear()
        if getattr(self.model, "dp_arena", None) is self:
            self.model.dp_arena = None

    def zero_grad(self) -> None:
        """One memset of the whole gradient buffer (slots written by kernels do not need it: they are overwritten)."""
        self.flat_grad.zero_()
        self.pending_scale = 1.0

    def slots_clobbered(self) -> None:
        """Tell the arena that flat_grad was written from outside (a test poisoning it, a checkpoint restore): slots
        whose gradient is structurally zero are then zeroed again by the next backward pass instead of being trusted."""
        self._zero_bias.clear()

    def rebind_grads(self) -> None:
        """Re-attach every parameter's .grad to its arena slot (after an optimizer.zero_grad(set_to_none=True), torch
        2.x's default, which only drops the Python binding: the kernels keep writing into the slots)."""
        for p, off in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                p.grad = self.flat_grad[off:off + p.numel()].view(p.shape)

    def all_reduce_mean(self):
        """Average the flat gradient over all ranks: ONE collective.  On RCCL (backend nccl) the averaging is the
        collective's own reduction op (no extra pass over the 54 MB buffer) and 1.0 is returned; on backends without
        AVG (gloo: the CPU tests) the sum is reduced and the factor 1/world is returned for the caller to fold into
        its next pass over the gradient (ArenaAdam: grad_scale of the fused Adam launch)."""
        ops.side.join()  # no-op unless a backward pass ended abnormally with side-stream work un-joined
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            if dist.get_backend() == "nccl":
                dist.all_reduce(self.flat_grad, op=dist.ReduceOp.AVG)
                return 1.0
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM)
            return 1.0 / dist.get_world_size()
        return 1.0

    def sync_loss(self, loss: torch.Tensor) -> torch.Tensor:
        """Return `loss` with the gradient averaging hooked onto its backward pass: when the autograd
        engine finishes `loss.backward()` (engine callback, queued from the root of the graph) the side
        stream is joined and the flat gradient is all-reduced once over RCCL - so the reference's
        `loss.backward(); optimizer.step()` (training_lit.py:85-87) needs no change (SURVEY.md section 8b,
        last row).  MTLModule does this for the train stage when `dp_arena` is set.  A remaining 1/world factor
        (gloo only) is applied in place unless an ArenaAdam is attached, which folds it into its update."""
        return _SyncGrads.apply(loss, self)

    def _end_of_backward(self):
        scale = self.all_reduce_mean()
        self._kernel_written.clear()  # the guard covers one backward pass; the next forward re-registers its slots
        if scale != 1.0:
            consumer = self._scale_consumer() if self._scale_consumer is not None else None
            if consumer is not None:
                # the attached ArenaAdam folds the factor into its fused update (no extra pass over 54 MB).  Until its
                # step() - or adam_step() - runs, flat_grad holds the SUM over ranks: anything else that reads gradients
                # in between (clipping, logging) must multiply by arena.pending_scale
                self.pending_scale = scale
            else:
                self.flat_grad.mul_(scale)

    def adam_step(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
        """torch.optim.Adam semantics (reference training_lit.py:51,87) as ONE fused launch over the arena."""
        if self._adam is None:
            z = torch.zeros_like(self.flat_param)
            self._adam = {"m": z, "v": z.clone(), "step": torch.zeros(1, dtype=torch.float32, device=z.device)}
        ops.side.join()
        st = self._adam
        st["step"] += 1
        # a 1/world factor still owed to the gradient (gloo: all_reduce_mean returned it to _end_of_backward) is consumed
        # HERE, whoever calls: a direct adam_step() after a discarded ArenaAdam cannot silently drop it
        grad_scale, self.pending_scale = grad_scale * self.pending_scale, 1.0
        ops.adam_step(self.flat_param, self.flat_grad, st["m"], st["v"], st["step"], lr, betas, eps, weight_decay,
                      grad_scale)
        self._kernel_written.clear()
        ops.packs.invalidate()  # parameters changed through raw pointers: packed operands are stale


class ArenaAdam(torch.optim.Optimizer):
    """torch.optim.Adam's interface over FlatArena.adam_step (one fused launch over the flat buffers).
    `param_groups`, `zero_grad` and the `state_dict` WIRE FORMAT are torch.optim.Adam's: state_dict() emits
    {"state": {i: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [{..., "params": [0..n-1]}]} with the
    moments sliced per parameter out of the flat buffers (parameter order = model.parameters(), the order
    torch.optim.Adam(module.parameters()) uses), and load_state_dict() accepts exactly that - so the reference's
    scheduler (`ReduceLROnPlateau`, training_lit.py:51-55) drives it unchanged and session_{epoch}.pt files
    written by `save_ckpt` (pipeline_utils.py:139-167) move between this optimizer and torch.optim.Adam in both
    directions.  Anything else raises (never a silent partial load)."""

    def __init__(self, arena: "FlatArena", lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.arena = arena
        super().__init__([arena.flat_param], dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        arena._scale_consumer = weakref.ref(self)  # weak: a discarded optimizer stops deferring the 1/world factor

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        self.arena.adam_step(lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"], weight_decay=g["weight_decay"])
        return loss

    def zero_grad(self, set_to_none: bool = False):
        """One memset of the flat gradient buffer (`set_to_none` is ignored: .grad must stay bound to the arena)."""
        self.arena.zero_grad()

    def state_dict(self):
        a, st = self.arena, self.arena._adam
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        group["params"] = list(range(len(a.params)))
        state = {}
        if st is not None:
            for i, (p, off) in enumerate(zip(a.params, a.offsets)):
                n = p.numel()
                state[i] = {"step": st["step"].detach().clone().reshape(()).cpu(),
                            "exp_avg": st["m"][off:off + n].view(p.shape).clone(),
                            "exp_avg_sq": st["v"][off:off + n].view(p.shape).clone()}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        a = self.arena
        if not isinstance(sd, dict) or "param_groups" not in sd or "state" not in sd:
            raise ValueError("ArenaAdam.load_state_dict: expected a torch.optim.Adam state_dict ('state' + 'param_groups')")
        groups = sd["param_groups"]
        ids = [i for g in groups for i in g["params"]]
        if ids != list(range(len(a.params))):
            raise ValueError(f"ArenaAdam.load_state_dict: the checkpoint covers {len(ids)} parameters, the arena holds "
                             f"{len(a.params)} (parameter order must be model.parameters())")
        hyper = [{k: v for k, v in g.items() if k != "params"} for g in groups]
        if any(h != hyper[0] for h in hyper[1:]):
            raise ValueError("ArenaAdam.load_state_dict: parameter groups with different hyper-parameters are not supported")
        for k in ("amsgrad", "maximize"):
            if hyper[0].get(k):
                raise ValueError(f"ArenaAdam.load_state_dict: {k}=True is not implemented by the fused update")
        self.param_groups[0].update({k: v for k, v in hyper[0].items() if k in self.param_groups[0]})
        state = sd["state"]
        if not state:
            a._adam = None
            return
        if sorted(state.keys()) != list(range(len(a.params))):
            raise ValueError("ArenaAdam.load_state_dict: 'state' must hold every parameter (a partially stepped optimizer "
                             "cannot be represented by one fused step counter)")
        dev = a.flat_param.device
        m, v = torch.zeros_like(a.flat_param), torch.zeros_like(a.flat_param)
        steps = set()
        for i, (p, off) in enumerate(zip(a.params, a.offsets)):
            e, n = state[i], p.numel()
            if tuple(e["exp_avg"].shape) != tuple(p.shape) or tuple(e["exp_avg_sq"].shape) != tuple(p.shape):
                raise ValueError(f"ArenaAdam.load_state_dict: moment shapes of parameter {i} do not match {tuple(p.shape)}")
            m[off:off + n].copy_(e["exp_avg"].reshape(-1))
            v[off:off + n].copy_(e["exp_avg_sq"].reshape(-1))
            steps.add(float(e["step"]))
        if len(steps) != 1:
            raise ValueError("ArenaAdam.load_state_dict: parameters with different step counts cannot share the fused update")
        a._adam = {"m": m, "v": v, "step": torch.full((1,), steps.pop(), dtype=torch.float32, device=dev)}


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


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def init_distributed():
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run contract).  Returns
    (rank, world, local_rank); initialises RCCL only when world > 1."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # VMTL_DIST_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (ranks then share
        # devices round-robin; RCCL refuses two ranks on one device)
        backend = os.environ.get("VMTL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            if backend != "nccl":
                local_rank %= max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local_rank)
        if backend == "nccl":
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
