"""Autograd bindings of the vmtl C ABI (include/vmtl.h).

Everything here is host plumbing: allocate output / workspace tensors with the torch
allocator, hand raw device pointers + the current HIP stream to libvmtl.so, register
the matching backward call with autograd.  No arithmetic happens in Python and there
is no fallback: tensors must live on a GPU and the extension must load.

Internal activation format: contiguous fp32 [B, H, W, Cs] (NHWC) with
Cs = ceil4(C) and zero padding channels; the logical channel count C travels with
the calling module.
"""
from __future__ import annotations

import contextlib
import os
import weakref

import torch

from ._lib import lib

ACT_NONE, ACT_RELU, ACT_HSWISH, ACT_HSIGMOID, ACT_SIGMOID = 0, 1, 2, 3, 4
ACT_CODES = {None: 0, "none": 0, "relu": 1, "hardswish": 2, "hardsigmoid": 3, "sigmoid": 4}


def ceil4(c: int) -> int:
    return (c + 3) // 4 * 4


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _req(t: torch.Tensor, name: str = "tensor") -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} is on {t.device}: the vmtl hot path only runs as HIP kernels on an MI355X "
            "(there is deliberately no CPU fallback)"
        )
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


_RECORD = None  # bench.py sets this to a list to capture the GEMM-kernel launches of one step


def _k(name, _flop=None, _xflop=None, **kw):
    if _RECORD is not None and _flop is not None:
        # _flop: algorithmic FLOPs of the reference formulation (logical channels); _xflop: FLOPs the launch
        # really executes when an algebraic rewrite makes them differ (up2_conv)
        _RECORD.append((name, dict(kw), float(_flop), float(_flop if _xflop is None else _xflop)))
    lib().callk(name, stream=_stream(), **kw)


# (the launch-dropping ablation switches of round 2 - VMTL_DBG_SKIP_SIDE / VMTL_DBG_SKIP / VMTL_DBG_EXTRA - produce WRONG
# results by design and are no longer part of the product: tools/dbg_hooks.py installs them around _k on request)
_STAMPS = None  # bench.py (VMTL_STAMPS=1) sets this to a list to collect a two-stream timeline


def stamp(tag):
    """Tuning aid: device wall-clock probe on the current stream (no-op unless _STAMPS is a list)."""
    if _STAMPS is None:
        return
    t = torch.zeros(1, dtype=torch.int64, device="cuda")
    lib().callk("vmtl_timestamp", out=t, stream=_stream())
    _STAMPS.append((tag, t))


def _slot(p):
    """Gradient slot of a parameter inside a dp.FlatArena (None when no arena is attached).  When a
    slot exists the backward kernels write the parameter gradient straight into it and autograd is
    told there is nothing to accumulate: gradients of the whole model then form ONE contiguous buffer
    (one RCCL all-reduce, one fused Adam launch) without per-parameter add / copy kernels."""
    if p is None:
        return None
    slot = getattr(p, "_vmtl_gslot", None)
    if slot is not None:
        p._vmtl_arena._kernel_written.add(id(p))  # dp.FlatArena's guard against mixed gradient paths
    return slot


def _empty(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def _reduce_rows(M: int) -> int:
    return lib().raw("vmtl_reduce_rows")(M)


# ----------------------------------------------------------------------------- weight-gradient branch
class _SideBranch:
    """Second HIP stream for the parameter-gradient launches of the backward pass.

    The data-gradient chain (dgrad -> BN backward -> dgrad ...) is the critical path of the backward pass;
    the weight gradients hang off it as leaves whose results nobody needs before the optimizer /
    all-reduce.  On the MobileNetV3 encoder those launches are small (tens of workgroups), so running
    them on a second stream lets them fill CUs the critical path leaves idle.  Inside a hipGraph capture
    the fork/join events become parallel graph branches.  Only used when the gradient goes to a
    FlatArena slot (nothing is handed back to autograd, which would expect it on the main stream).
    The side stream is joined by an end-of-backward engine callback (and by FlatArena before it reads
    the gradient buffer).  VMTL_SIDE_STREAM=0 turns the branch off; VMTL_SIDE_MAX_ROWS bounds the GEMM
    rows (pixels) of a launch that may leave the main stream."""

    def __init__(self):
        self.enabled = os.environ.get("VMTL_SIDE_STREAM", "1") != "0"
        self.max_rows = int(os.environ.get("VMTL_SIDE_MAX_ROWS", str(1 << 30)))
        self.nstreams = max(1, int(os.environ.get("VMTL_SIDE_STREAMS", "1")))
        self.streams = {}
        self.pending = None  # side streams with un-joined work (dict: stream -> True)
        self.in_branch = False
        self.rr = 0
        self.task_streams = {}  # device index -> stream that runs the second task network of CSNet
        # CSNet's two task networks as two parallel graph branches.  Measured on MI355X (csnet 128x256 bs 32): 17.7 ms/step
        # on one stream with the weight gradients on the side stream, 17.7 with task streams AND the side branch (three
        # hardware queues), 15.4 with the task streams alone - so while a task-parallel model is stepping (task_mode,
        # set by its forward through packs.refresh) the weight gradients stay on their task's stream
        self.task_parallel = os.environ.get("VMTL_TASK_STREAMS", "1") != "0"
        self.task_mode = False

    def task_stream(self, device):
        """Stream for the second of two independent task networks (CSNet: the cross-stitch layers of the
        reference only scale each task's own features, so the two U-Nets never exchange data).  join() also
        waits for it: gradients written into arena slots from that stream have no AccumulateGrad node the
        autograd engine could synchronise on."""
        s = self.task_streams.get(device.index)
        if s is None:
            s = self.task_streams[device.index] = torch.cuda.Stream(device=device)
        return s

    def stream(self, device):
        """One side stream by default.  VMTL_SIDE_STREAMS > 1 spreads branches round-robin over several;
        measured on MI355X that LOSES (16.4 -> 18.3 ms/step with 2, 19.2 with 4): every extra hardware
        queue adds cross-queue dependency latency to the graph's critical path."""
        pool = self.streams.get(device.index)
        if pool is None:
            pool = self.streams[device.index] = [torch.cuda.Stream(device=device) for _ in range(self.nstreams)]
        self.rr = (self.rr + 1) % len(pool)
        return pool[self.rr]

    def join(self):
        pend, self.pending = self.pending, None
        if pend is not None:
            for s in pend:
                torch.cuda.current_stream(s.device).wait_stream(s)
            for s in self.task_streams.values():
                torch.cuda.current_stream(s.device).wait_stream(s)
        packs.join()

    def join_at_end_of_backward(self):
        """Called from a backward function: the stream that called backward() waits for the side / task streams
        when the engine is done (gradients written into arena slots have no AccumulateGrad node it could sync on)."""
        if self.pending is None:
            self.pending = {}
            torch.autograd.Variable._execution_engine.queue_callback(self.join)

    def mark(self):
        """Record the fork point on the current (main) stream.  Taken on entry of a backward function so the
        branch depends on what was queued BEFORE the function's own main-stream launches, while those
        launches are still issued first (inside a hipGraph capture the main-stream node then precedes the
        branch node in creation order, which keeps the critical path on one hardware queue)."""
        if not self.enabled:
            return None
        ev = torch.cuda.Event()
        ev.record()
        return ev

    @contextlib.contextmanager
    def branch(self, use, rows, mark, *tensors):
        """Run the enclosed launches on the side stream, ordered after `mark`.  `tensors` are main-stream
        tensors the launches read: the allocator must not recycle them for main-stream work until the
        side stream is past these launches."""
        if not (self.enabled and use and mark is not None and rows <= self.max_rows) or self.task_mode:
            yield
            return
        main = torch.cuda.current_stream()
        s = self.stream(main.device)
        s.wait_event(mark)
        for t in tensors:
            if t is not None:
                t.record_stream(s)
        self.join_at_end_of_backward()
        self.pending[s] = True
        with torch.cuda.stream(s):

            self.in_branch = True
            try:
                yield
            finally:
                self.in_branch = False


side = _SideBranch()


def _remember_mode(ctx):
    """Output nodes of a model (the first nodes its backward pass runs) keep the stream mode their FORWARD was built in:
    side.task_mode is process-global and another model's forward may have flipped it before this graph's backward."""
    ctx.task_mode = side.task_mode


def _restore_mode(ctx):
    side.task_mode = ctx.task_mode
    if ctx.task_mode:  # task streams were used by this graph: the calling stream joins them when the engine is done
        side.join_at_end_of_backward()


# ----------------------------------------------------------------------------- packing
class _PackCache:
    """Packed GEMM operands of the model's weights, refreshed by ONE batched launch per step.

    get() hands out a persistent packed buffer for (weight, layout).  An entry is valid while the weight's
    storage, autograd version and the global epoch are unchanged (torch optimizers bump the version;
    FlatArena.adam_step() and refresh() bump the epoch).  refresh() re-packs every known entry with a
    single vmtl_pack_weights_batch launch; a miss falls back to an individual pack and marks the
    descriptor table for a rebuild (done outside graph capture, during warm-up).

    Weights are held by WEAK reference: the entries (and their packed device buffers) of a model that has
    been dropped are purged by the next refresh() instead of being re-packed forever - a process that
    builds several models (the reference's hyperparam_tuning.py loop, a train-then-eval pair, the test
    session) neither leaks device memory nor slows down with the number of dead models."""

    def __init__(self):
        self.entries = {}  # key -> dict(dst, params, wref, ptr, version, epoch)
        self.custom = {}   # key -> dict(dst, fn, wref, ptr, version, epoch): operands with their own pack kernel
        self.shared_bufs = {}  # key -> (weakref of the anchoring weight, buffer)
        self.epoch = 0
        self.table = None  # (device uint8 tensor, n, total): entries packed on the main stream
        self.table_side = None  # the large entries, packed on the side stream
        self.live = []
        self.dirty = True
        self.custom_ready = None  # event: this step's custom operands are packed (side stream)

    def invalidate(self):
        self.epoch += 1

    @staticmethod
    def _alive(e):
        w = e["wref"]()
        if "sref" in e and e["sref"]() is None:
            return False
        return w is not None and w.data_ptr() == e["ptr"]

    def purge(self):
        """Drop the entries of weights that no longer exist (or moved, e.g. into a FlatArena)."""
        n = len(self.entries) + len(self.custom)
        self.entries = {k: e for k, e in self.entries.items() if self._alive(e)}
        self.custom = {k: e for k, e in self.custom.items() if self._alive(e)}
        self.shared_bufs = {k: e for k, e in self.shared_bufs.items() if e[0]() is not None}
        if len(self.entries) + len(self.custom) != n:
            self.dirty = True

    def get(self, weight, kind, params, offset=0, out=None, scale=None):
        """params = (R1, R0, T, C, Cs, sr1, sr0, st, sc, flip); offset = first element of the weight to read;
        out: destination for a NEW entry (a slice of a caller-owned persistent buffer, see shared());
        scale = (tensor, first element, stride, smode): a cross-stitch factor folded into the operand
        (vmtl_pack_weights_scaled) - the entry is then also invalidated by a change of that tensor."""
        key = (id(weight), kind, params, offset)
        e = self.entries.get(key)
        if e is not None and e["wref"]() is not weight:  # id() of a dead tensor reused by a new one
            e = None
        sver = None if scale is None else (scale[0].data_ptr(), scale[0]._version, scale[1], scale[2], scale[3])
        if (e is not None and e["ptr"] == weight.data_ptr() and e["version"] == weight._version and e["epoch"] == self.epoch
                and e.get("sver") == sver):
            if e.get("side"):
                self.join()  # packed on the side stream this step: wait for its event (no-op after the first time)
            return e["dst"]
        R1, R0, T, C, Cs = params[:5]
        if (e is None or e["ptr"] != weight.data_ptr() or (out is not None and e["dst"].data_ptr() != out.data_ptr())
                or (e.get("sver") or (None,))[0] != (sver or (None,))[0]):
            e = {"dst": _empty((R1 * R0, T * Cs), weight) if out is None else out, "params": params,
                 "wref": weakref.ref(weight), "ptr": weight.data_ptr(), "offset": offset}
            if scale is not None:  # the table needs the factor's address; weak: the cache must not keep parameters alive
                e["sref"], e["scale"] = weakref.ref(scale[0]), (scale[0].data_ptr() + 4 * scale[1], scale[2], scale[3])
            self.entries[key] = e
            self.dirty = True
        src = weight.view(-1)[offset:] if offset else weight
        if scale is None:
            pack(src, *params, out=e["dst"])
        else:
            R1, R0, T, C, Cs, sr1, sr0, st, sc, flip = params
            _k("vmtl_pack_weights_scaled", src=src, dst=e["dst"], R1=R1, R0=R0, T=T, C=C, Cs=Cs, sr1=sr1, sr0=sr0, st=st,
               sc=sc, flip=flip, scale=scale[0].view(-1)[scale[1]:], sstride=scale[2], smode=scale[3])
        e["version"], e["epoch"], e["sver"] = weight._version, self.epoch, sver
        return e["dst"]

    def shared(self, anchor, kind, shape):
        """A persistent buffer several entries pack into (e.g. the two heads' weights as ONE GEMM operand); it lives as
        long as `anchor` (a weight) does."""
        key = (id(anchor), kind, tuple(shape))
        e = self.shared_bufs.get(key)
        if e is None or e[0]() is not anchor:
            e = (weakref.ref(anchor), _empty(shape, anchor))
            self.shared_bufs[key] = e
        return e[1]

    def get_custom(self, weight, kind, shape, fn, deps=()):
        """Operand built by its own kernel (fn(weight, dst) launches it): cached like get(); refresh() rebuilds
        all of them on the side stream at the start of a step, so inside the step this is a lookup (+ one
        event wait on first use).  deps: further tensors fn reads (through weak references of its own: the cache must
        not keep parameters alive) - a change of any of them invalidates the entry like a change of `weight`."""
        key = (id(weight), kind)
        e = self.custom.get(key)
        if e is not None and e["wref"]() is not weight:
            e = None
        ver = (weight._version,) + tuple((d.data_ptr(), d._version) for d in deps)
        if e is not None and e["ptr"] == weight.data_ptr() and e["version"] == ver and e["epoch"] == self.epoch:
            self.join()
            return e["dst"]
        if e is None or e["ptr"] != weight.data_ptr():
            e = {"dst": _empty(shape, weight), "wref": weakref.ref(weight), "ptr": weight.data_ptr()}
            self.custom[key] = e
        e["fn"], e["deps"] = fn, tuple(weakref.ref(d) for d in deps)
        fn(weight, e["dst"])
        e["version"], e["epoch"] = ver, self.epoch
        return e["dst"]

    def join(self):
        """Make the current stream wait for the side-stream packing of this step (no-op when already waited)."""
        ev, self.custom_ready = self.custom_ready, None
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    # entries at least this large are packed on the SIDE stream (second table): the decoder's operands are 50 of the
    # 54 MB and are not needed before the encoder has run (~1.5 ms into the step); packing them on the main stream put
    # 0.12 ms in front of the first conv
    SIDE_TABLE_MIN_ELEMS = 1 << 18

    def _build_table(self):
        import struct

        def table(live):
            if not live:
                return None
            recs, start = [], 0
            for e in live:
                R1, R0, T, C, Cs, sr1, sr0, st, sc, flip = e["params"]
                if R1 * R0 * T * Cs >= 1 << 31:
                    raise ValueError("packed operand too large for the batched pack kernel (32-bit offsets inside one operand)")
                sptr, sstride, smode = e.get("scale", (0, 0, 0))
                recs.append(struct.pack("<QQqqqqqQiiiiiiii", e["ptr"] + 4 * e.get("offset", 0), e["dst"].data_ptr(), sr1, sr0,
                                        st, sc, start, sptr, R1, R0, T, C, Cs, flip, sstride, smode))
                start += R1 * R0 * T * Cs
            size = lib().raw("vmtl_pack_desc_bytes")()
            blob = b"".join(r.ljust(size, b"\0") for r in recs)
            dev = live[0]["dst"].device
            return torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev), len(recs), start

        live = list(self.entries.values())
        big = lambda e: e["params"][0] * e["params"][1] * e["params"][2] * e["params"][4] >= self.SIDE_TABLE_MIN_ELEMS
        for e in live:
            e["side"] = bool(side.enabled and big(e))
        self.live = live
        self.table = table([e for e in live if not e["side"]])
        self.table_side = table([e for e in live if e["side"]])
        self.dirty = False

    def refresh(self, task_mode=False):
        """Re-pack every known weight (call once at the start of a step, before the forward).  task_mode: the model
        stepping runs its task networks on parallel streams (see _SideBranch.task_parallel)."""
        side.task_mode = bool(task_mode)
        self.join()
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing:
            self.purge()  # never while capturing: the captured table must keep describing the same launches
        if not self.entries and not self.custom:
            return
        self.epoch += 1
        if self.dirty:
            if capturing:
                return  # keep per-call packing inside this capture; the table is rebuilt on the next eager step
            self._build_table()
        if self.table is not None:
            table, n, total = self.table
            _k("vmtl_pack_weights_batch", descs=table, n=n, total=total)
        for e in self.live:
            w = e["wref"]()
            if w is not None:
                e["version"], e["epoch"] = w._version, self.epoch
                if "sref" in e:  # the batched launch read the stitch factor as it is now
                    sw = e["sref"]()
                    if sw is not None and e.get("sver") is not None:
                        e["sver"] = (sw.data_ptr(), sw._version) + tuple(e["sver"][2:])
        # the large operands and the ones with their own pack kernels (up2 phase / gradient matrices): none is needed
        # before the decoder, so they are built on the side stream while the encoder runs; get() / get_custom() make
        # the consuming stream wait for the event on first use
        if self.custom or self.table_side is not None:
            # task_mode: both task streams consume packed operands, and only the first get() waits for the side
            # stream's event - pack on the calling stream instead
            use_side = side.enabled and not side.task_mode
            if use_side:
                main = torch.cuda.current_stream()
                s = side.stream(main.device)
                s.wait_stream(main)
                ctx = torch.cuda.stream(s)
            else:
                ctx = contextlib.nullcontext()
            with ctx:
                if self.table_side is not None:
                    table, n, total = self.table_side
                    _k("vmtl_pack_weights_batch", descs=table, n=n, total=total)
                for e in self.custom.values():
                    w = e["wref"]()
                    if w is None:
                        continue
                    deps = [d() for d in e.get("deps", ())]
                    if any(d is None for d in deps):
                        continue
                    e["fn"](w, e["dst"])
                    e["version"] = (w._version,) + tuple((d.data_ptr(), d._version) for d in deps)
                    e["epoch"] = self.epoch
                if use_side:
                    ev = torch.cuda.Event()
                    ev.record()
                    self.custom_ready = ev


packs = _PackCache()


def pack(src, R1, R0, T, C, Cs, sr1, sr0, st, sc, flip=0, out=None):
    dst = _empty((R1 * R0, T * Cs), src) if out is None else out
    _k("vmtl_pack_weights", src=src, dst=dst, R1=R1, R0=R0, T=T, C=C, Cs=Cs, sr1=sr1, sr0=sr0, st=st, sc=sc, flip=flip)
    return dst


def unpack(packed, shape, R1, R0, T, C, Cs, sr1, sr0, st, sc, flip=0, out=None, nslabs=1, slab_stride=0):
    grad = _empty(shape, packed) if out is None else out
    _k("vmtl_unpack_weights", packed=packed, grad=grad, R1=R1, R0=R0, T=T, C=C, Cs=Cs, sr1=sr1, sr0=sr0, st=st,
       sc=sc, flip=flip, nslabs=nslabs, slab_stride=slab_stride)
    return grad


def _wgrad(x, dy, B, H, W, Cs, Ho, Wo, ldy, Nw, KH, KW, stride, pad, flop, xflop=None):
    """Weight-gradient slabs [splits][Nw][KH*KW*Cs] (summed later by unpack)."""
    if (KH == 3 and KW == 3 and stride == 1 and pad == 1 and B * H * W * max(Cs, ldy) * 4 < 1 << 32
            and lib().raw("vmtl_conv3x3_wgrad_small_supported")(Cs, ldy, W)):
        # narrow full-resolution layers: the strip-walking halo kernel reads x once instead of once per tap
        ns = lib().raw("vmtl_conv3x3_wgrad_small_slabs")(B, H, W)
        slabs = _empty((ns, Nw, 9 * Cs), x)
        _k("vmtl_conv3x3_wgrad_small", _flop=flop, _xflop=xflop, x=x, dy=dy, slabs=slabs, nslabs=ns, B=B, H=H, W=W, Cs=Cs, ldy=ldy,
           Nw=Nw)
        return slabs, ns
    splits = lib().raw("vmtl_conv2d_wgrad_splits")(B * Ho * Wo, Nw, KH * KW * Cs)
    slabs = _empty((splits, Nw, KH * KW * Cs), x)
    _k("vmtl_conv2d_wgrad", _flop=flop, _xflop=xflop, x=x, dy=dy, slabs=slabs, splits=splits, B=B, H=H, W=W, Cs=Cs, Ho=Ho, Wo=Wo,
       ldy=ldy, Nw=Nw, KH=KH, KW=KW, stride=stride, pad=pad)
    return slabs, splits


def _colsum(a, b, M, C, Cs, mode=0, reduce_all=0, out=None):
    partial = _empty((_reduce_rows(M) + 1, Cs), a)  # + 1: scratch row of the reduce_all form
    if out is None:
        out = _empty((1 if reduce_all else C,), a)
    _k("vmtl_colsum", a=a, b=b, M=M, C=C, Cs=Cs, mode=mode, reduce_all=reduce_all, partial=partial, out=out)
    return out


def _bias_grad(bias, slot, dy, M, Cout, ldy, zero, fork):
    """dL/dbias of a conv (column sums of dy), into its arena slot when there is one (returns None then).
    zero: the conv feeds a TRAIN-mode BatchNorm - the batch mean absorbs the bias, its gradient is exactly 0.  This
    function is the slot's only writer, so the zero is written ONCE (first backward pass) and remembered
    (FlatArena._zero_bias): eager steps launch nothing afterwards (MTAN: 56 memset launches per step before).  Column
    sums written later (an eval-mode BatchNorm step) or FlatArena.slots_clobbered() forget it.
    Inside a hipGraph CAPTURE the (4..2048-byte) memset is still recorded, on the side branch: measured on MI355X /
    ROCm 7.0 runtime (MTAN 256x256 bs 16, round 3), the replayed graph runs its side branch CONCURRENTLY with the main
    chain only when these memset nodes are part of it - 52.3 ms/step with them, 55.7-57.8 without (side branch started
    after the last main-chain kernel: two-stream timeline of VMTL_STAMPS=1); one memset per step, one per branch entry
    and DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 did not reproduce it.  They are off the critical path (side branch, ~2 us)."""
    arena = getattr(bias, "_vmtl_arena", None) if slot is not None else None
    if zero:
        if slot is None:
            db = _empty((Cout,), dy)
            _k("vmtl_fill_zero", p=db, n=Cout)
            return db
        capturing = torch.cuda.is_current_stream_capturing()
        if arena is None or id(bias) not in arena._zero_bias or (capturing and side.enabled and not side.task_mode):
            with side.branch(True, M, fork, dy):
                _k("vmtl_fill_zero", p=slot, n=Cout)
            if arena is not None and not capturing:
                arena._zero_bias.add(id(bias))
        return None
    with side.branch(slot is not None, M, fork, dy):
        db = _colsum(dy, None, M, Cout, ldy, out=slot)
    if slot is not None:
        if arena is not None:
            arena._zero_bias.discard(id(bias))
        return None
    return db


FUSE_DW = os.environ.get("VMTL_FUSE_DW", "1") != "0"  # encoder: bn1 + act + depthwise conv as one node (vmtl_dwconv_bn_fwd)
_PW = os.environ.get("VMTL_PW", "1") != "0"  # pointwise GEMM kernel for 1x1 convs (csrc/conv_pw.hip)
_PW_MAX_ROWS = int(os.environ.get("VMTL_PW_MAX_ROWS", str(1 << 21)))  # measured on MTAN (M = 2^20): 57.1 -> 55.5 ms/step


def _is_pw(B, Ho, Wo, KH, KW, stride, pad, shuffle=0):
    return _PW and KH == 1 and KW == 1 and stride == 1 and pad == 0 and not shuffle and B * Ho * Wo <= _PW_MAX_ROWS


_SMALL_MIN_ROWS = int(os.environ.get("VMTL_SMALL_MIN_ROWS", str(1 << 16)))  # plain narrow 3x3 launches on the halo-tile kernel


def conv_ksplit(B, Ho, Wo, Cs, ldy, KH, KW, stride, pad, shuffle=0) -> int:
    """K slices a dense conv launch of this shape runs as (1 = none).  A split launch has no statistics epilogue: the
    BatchNorm that follows takes its statistics from its own sweep of the (small) output instead."""
    if shuffle or _is_pw(B, Ho, Wo, KH, KW, stride, pad, shuffle):
        return 1
    return lib().raw("vmtl_conv2d_ksplit")(B, Ho, Wo, ldy, KH * KW * Cs)


_SMALL_STATS = os.environ.get("VMTL_SMALL_STATS", "1") != "0"  # statistics-epilogue launches of narrow layers there too


def _small_route(B, H, W, Cs, ldy, KH, KW, stride, pad, shuffle=0, with_stats=False) -> bool:
    """A narrow full-resolution 3x3 / stride 1 / pad 1 launch that runs on the halo-tile kernel (vmtl_conv3x3_small) instead
    of the implicit GEMM: 11-20 % faster on 16/32-channel layers at 1 M pixels (tools/bench_small.py), with or without the
    statistics epilogue (whose tiles must be whole: H % 4 == 0, W % 32 == 0)."""
    if shuffle or KH != 3 or KW != 3 or stride != 1 or pad != 1 or ldy > 36 or B * H * W < _SMALL_MIN_ROWS:
        return False
    if B * H * W * 36 > 0x7FFFFFFF:  # that kernel's 32-bit element offsets
        return False
    if not conv3x3_small_supported(Cs, ldy):  # weight rows <= ldy
        return False
    return not with_stats or (_SMALL_STATS and H % 4 == 0 and W % 32 == 0)


def conv_stats_geometry(B, Ho, Wo, Cs, ldy, KH, KW, stride, pad):
    """(rows, pixels per row) of the BatchNorm partial rows a conv launch of this shape emits from its epilogue."""
    if _small_route(B, Ho, Wo, Cs, ldy, KH, KW, stride, pad, with_stats=True):
        return lib().raw("vmtl_conv3x3_small_stat_rows")(B, Ho, Wo), lib().raw("vmtl_conv3x3_small_stat_block")(B, Ho, Wo)
    if _is_pw(B, Ho, Wo, KH, KW, stride, pad):
        M = B * Ho * Wo
        return lib().raw("vmtl_conv1x1_stats_rows")(M, ldy, Cs, 0), lib().raw("vmtl_conv1x1_stats_block")(M, ldy, Cs, 0)
    return lib().raw("vmtl_conv2d_stats_rows")(B, Ho, Wo, ldy), lib().raw("vmtl_conv2d_stats_block")(B, Ho, Wo, ldy)


def _conv_launch(x, wp, bias, y, stats, B, H, W, Cs, Ho, Wo, ldy, Nw, Cout, KH, KW, stride, pad, shuffle=0, cin=None,
                 algo_flop=None):
    flop = 2.0 * B * Ho * Wo * Nw * KH * KW * (Cs if cin is None else cin)
    if _is_pw(B, Ho, Wo, KH, KW, stride, pad, shuffle):
        _k("vmtl_conv1x1_fwd", _flop=flop if algo_flop is None else algo_flop, _xflop=flop, x=x, wp=wp, bias=bias, y=y,
           stats=stats, M=B * Ho * Wo, Ks=Cs, ldy=ldy, Nw=Nw, Cout=Cout)
        return
    if _small_route(B, H, W, Cs, ldy, KH, KW, stride, pad, shuffle, with_stats=stats is not None):
        # narrow full-resolution layer: the halo-tile kernel reads the input once (statistics rows: conv_stats_geometry)
        _small(x, wp, y, B, H, W, Cs, ldy, Nw, Cout, flop if algo_flop is None else algo_flop, bias=bias, stats=stats,
               ep_mode=1 if stats is not None else 0)
        return
    if stats is None and not shuffle:
        # contraction without a statistics epilogue (data gradients; forward convs of tile-starved layers, see
        # conv_ksplit): split K when the tile grid alone cannot fill the chip
        ks = lib().raw("vmtl_conv2d_ksplit")(B, Ho, Wo, ldy, KH * KW * Cs)
        if ks > 1:
            ws = _empty((ks, B * Ho * Wo, ldy), x)
            _k("vmtl_conv2d_fwd_ws", _flop=flop if algo_flop is None else algo_flop, _xflop=flop, x=x, wp=wp, bias=bias, y=y,
               ws=ws, B=B, H=H, W=W, Cs=Cs, Ho=Ho, Wo=Wo, ldy=ldy, Nw=Nw, Cout=Cout, KH=KH, KW=KW, stride=stride, pad=pad)
            return
    _k("vmtl_conv2d_fwd", _flop=flop if algo_flop is None else algo_flop, _xflop=flop, x=x, wp=wp, bias=bias, y=y, stats=stats, B=B, H=H, W=W, Cs=Cs, Ho=Ho, Wo=Wo, ldy=ldy,
       Nw=Nw, Cout=Cout, KH=KH, KW=KW, stride=stride, pad=pad, act=0, shuffle=shuffle)


# ----------------------------------------------------------------------------- conv2d
def _stitch_view(stitch_w, task, C):
    """(first element, stride) of the diagonal w[task, task, (c)] inside the flattened CrossStitchLayer parameter
    (reference models/cross_stitch_model.py:21-37: (T, T, C) channel-wise or (T, T) layer-wise)."""
    T = stitch_w.shape[0]
    if stitch_w.dim() == 3:
        if stitch_w.shape[2] != C:
            raise ValueError(f"stitched conv: the stitch layer has {stitch_w.shape[2]} channels, the conv reads {C}")
        return (task * T + task) * C, 1
    return task * T + task, 0


class _Conv2d(torch.autograd.Function):
    """y = conv2d(x, weight) (+ bias); weight stays in torch (Cout, Cin, KH, KW) layout.
    stitch_w (optional): the CrossStitchLayer parameter whose diagonal entry of `stitch_task` scales x first
    (y = conv(w[a,a,(c)] * x): reference models/cross_stitch_model.py:32-37 + the conv that follows every stitch site in
    the CSNet walk :108-142).  The scale is FOLDED into the packed operands (forward and data gradient) and its weight
    gradient is taken from the conv's own weight-gradient slabs: no pass over the activations for the stitch at all."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, want_stats, zero_bias_grad=False, stitch_w=None, stitch_task=0):
        x = _req(x, "x")
        weight = _req(weight, "weight")
        B, H, W, Cs = x.shape
        Cout, Cin, KH, KW = weight.shape
        if ceil4(Cin) != Cs:
            raise ValueError(f"conv2d: input has {Cs} storage channels, weight expects Cin={Cin}")
        KK = KH * KW
        Ho = (H + 2 * pad - KH) // stride + 1
        Wo = (W + 2 * pad - KW) // stride + 1
        ldy = ceil4(Cout)
        if stitch_w is None:
            wp = packs.get(weight, "fwd", (1, Cout, KK, Cin, Cs, 0, Cin * KK, 1, KK, 0))
        else:
            stitch_w = _req(stitch_w, "stitch weights")
            soff, sstride = _stitch_view(stitch_w, stitch_task, Cin)
            wp = packs.get(weight, f"fwd_st{stitch_task}", (1, Cout, KK, Cin, Cs, 0, Cin * KK, 1, KK, 0),
                           scale=(stitch_w, soff, sstride, 1))
        y = _empty((B, Ho, Wo, ldy), x)
        stats = None
        if want_stats and conv_ksplit(B, Ho, Wo, Cs, ldy, KH, KW, stride, pad) > 1:
            want_stats = False  # tile-starved layer: split K, the BatchNorm sweeps the (small) output itself
        if want_stats:
            rows, _ = conv_stats_geometry(B, Ho, Wo, Cs, ldy, KH, KW, stride, pad)
            stats = _empty((rows, 2, ldy), x)
        _conv_launch(x, wp, bias, y, stats, B, H, W, Cs, Ho, Wo, ldy, Cout, Cout, KH, KW, stride, pad, cin=Cin)
        ctx.save_for_backward(x, weight, stitch_w)
        ctx.cfg = (stride, pad, bias is not None)
        ctx.zero_bias_grad = bool(zero_bias_grad)
        ctx.slots = (_slot(weight), _slot(bias))
        ctx.stitch = (stitch_task, _slot(stitch_w))
        ctx.bias = bias
        ctx.set_materialize_grads(False)  # no zero-filled "gradient" tensor for the stats output
        if want_stats:
            ctx.mark_non_differentiable(stats)
            return y, stats
        return y, None

    @staticmethod
    def backward(ctx, dy, _dstats):
        x, weight, stitch_w = ctx.saved_tensors
        stride, pad, has_bias = ctx.cfg
        stitch_task, stitch_slot = ctx.stitch
        if dy is None:
            return (None,) * 9
        dy = _req(dy, "dy")
        B, H, W, Cs = x.shape
        Cout, Cin, KH, KW = weight.shape
        KK = KH * KW
        _, Ho, Wo, ldy = dy.shape
        dx = dw = db = dst_w = None
        if stitch_w is not None:
            soff, sstride = _stitch_view(stitch_w, stitch_task, Cin)
        stamp(f"main conv M={B * Ho * Wo} N={Cout} K={KK * Cin}")
        fork = side.mark()  # parameter gradients branch off here, before the data gradient
        if ctx.needs_input_grad[0]:
            if stride != 1:
                raise NotImplementedError("data gradient of a strided dense conv is not on the hot path")
            if stitch_w is None:
                wd = packs.get(weight, "dgrad", (1, Cin, KK, Cout, ldy, 0, KK, 1, Cin * KK, 1))
            else:  # rows of the data-gradient operand are the input channels: d(x) = s * d(s * x)
                wd = packs.get(weight, f"dgrad_st{stitch_task}", (1, Cin, KK, Cout, ldy, 0, KK, 1, Cin * KK, 1),
                               scale=(stitch_w, soff, sstride, 2))
            dx = _empty((B, H, W, Cs), x)
            _conv_launch(dy, wd, None, dx, None, B, Ho, Wo, ldy, H, W, Cs, Cin, Cin, KH, KW, 1, KH - 1 - pad, cin=Cout)
        if ctx.needs_input_grad[1] or (stitch_w is not None and ctx.needs_input_grad[7]):
            use_side = ctx.slots[0] is not None and (stitch_w is None or stitch_slot is not None)
            with side.branch(use_side, B * Ho * Wo, fork, x, dy):
                slabs, ns = _wgrad(x, dy, B, H, W, Cs, Ho, Wo, ldy, Cout, KH, KW, stride, pad,
                                   2.0 * B * Ho * Wo * Cout * KK * Cin)
                if stitch_w is None:
                    dw = unpack(slabs, weight.shape, 1, Cout, KK, Cin, Cs, 0, Cin * KK, 1, KK, out=ctx.slots[0], nslabs=ns)
                else:
                    # the slabs hold dL/d(W*s): dW = slabs * s, and the stitch weight's gradient is their contraction
                    # with W (one small launch pair on the weight-sized tensor; off-diagonal entries stay zero)
                    dw = _empty(weight.shape, x) if ctx.slots[0] is None else ctx.slots[0]
                    n = Cin if sstride else 1
                    if stitch_slot is not None:
                        ds = stitch_slot.view(-1)[soff:soff + n]
                    else:
                        dst_w = torch.zeros_like(stitch_w)
                        ds = dst_w.view(-1)[soff:soff + n]
                    _k("vmtl_unpack_weights_stitch", packed=slabs, grad=dw, w=weight, scale=stitch_w.view(-1)[soff:],
                       sstride=sstride, ds=ds, work=_empty((Cout * Cin * KK + Cin,), x), R0=Cout, T=KK, C=Cin, Cs=Cs,
                       nslabs=ns, slab_stride=0, reduce_all=0 if sstride else 1)
                stamp(f"side conv M={B * Ho * Wo} N={Cout} K={KK * Cin}")
            if ctx.slots[0] is not None:
                dw = None
        if has_bias and ctx.needs_input_grad[2]:
            db = _bias_grad(ctx.bias, ctx.slots[1], dy, B * Ho * Wo, Cout, ldy, ctx.zero_bias_grad, fork)
        return dx, dw, db, None, None, None, None, dst_w, None


class _BNActPw(torch.autograd.Function):
    """(y, stats) = conv1x1(act(BN(x))) (+ bias): a 1x1 conv that owns the BatchNorm + activation of its producer
    (timm InvertedResidual bn2 + act -> conv_pwl [3P]; MTAN attention bn1 + ReLU -> conv2, reference
    models/mtan_model.py:60-66,142-148) - the pointwise counterpart of _BNActConv.  x is the RAW output of the
    producing conv with its BatchNorm partial rows; normalise + activation run on the GEMM's operand fragments
    (vmtl_conv1x1_bn_fwd; the activated matrix is written back once for the weight gradient), and the data gradient
    ends with the activation's and BatchNorm's backward reduction (vmtl_conv1x1_bnbwd): no apply pass forward, no
    reduce pass backward."""

    @staticmethod
    def forward(ctx, x, stats, rpb, gamma, beta, rm, rv, nbt, weight, bias, res, cfg):
        C, training, momentum, eps, act, want_stats, zero_bias_grad, return_act = cfg
        x, weight = _req(x, "x"), _req(weight, "weight")
        B, H, W, Cs = x.shape
        M = B * H * W
        Cout, Cin = weight.shape[0], weight.shape[1]
        if ceil4(C) != Cs or Cin != C or tuple(weight.shape[2:]) != (1, 1):
            raise ValueError("bn_act_conv1x1: (Cout, C, 1, 1) weight over x's channels expected")
        if res is not None:
            res = _req(res, "res")
            if tuple(res.shape) != tuple(x.shape) or act != ACT_NONE:
                raise ValueError("bn_act_conv1x1: the residual operand needs x's shape and no activation (bn3 + skip)")
        mean, invstd, ca, cc = _bn_fwd_coef(x, stats, rpb, gamma, beta, rm, rv, nbt, C, training, momentum, eps)
        wp = packs.get(weight, "fwd", (1, Cout, 1, Cin, Cs, 0, Cin, 1, 1, 0))
        ldy = ceil4(Cout)
        a, y = _empty(x.shape, x), _empty((B, H, W, ldy), x)
        ostats = None
        if want_stats:
            ostats = _empty((lib().raw("vmtl_conv1x1_stats_rows")(M, ldy, Cs, 0 if res is None else 1), 2, ldy), x)
        if res is None:
            _k("vmtl_conv1x1_bn_fwd", _flop=2.0 * M * Cout * Cin, x=x, coef_a=ca, coef_c=cc, act_in=act, a_out=a, wp=wp,
               bias=bias, y=y, stats=ostats, M=M, Ks=Cs, ldy=ldy, Nw=Cout, Cout=Cout)
        else:
            _k("vmtl_conv1x1_bn_res_fwd", _flop=2.0 * M * Cout * Cin, x=x, coef_a=ca, coef_c=cc, act_in=act, res=res, a_out=a,
               wp=wp, bias=bias, y=y, stats=ostats, M=M, Ks=Cs, ldy=ldy, Nw=Cout, Cout=Cout)
        ctx.save_for_backward(x, a, weight, mean, invstd, gamma, beta)
        ctx.cfg = (C, training, act, bias is not None, bool(zero_bias_grad), res is not None)
        ctx.slots = (_slot(gamma), _slot(beta), _slot(weight), _slot(bias))
        ctx.bias = bias
        ctx.set_materialize_grads(False)
        if ostats is not None:
            ctx.mark_non_differentiable(ostats)
        if return_act:  # a = act(BN(x)) [+ res] as a differentiable output: the block's own skip branch / a decoder tap
            return y, ostats, a
        return y, ostats

    @staticmethod
    def backward(ctx, dy, _dstats, d_a=None):
        x, a, weight, mean, invstd, gamma, beta = ctx.saved_tensors
        C, training, act, has_bias, zero_bias, has_res = ctx.cfg
        sg, sb, sw, sbias = ctx.slots
        if dy is None:
            if d_a is not None:
                raise NotImplementedError("bn_act_conv1x1: gradient through the activation output only")
            return (None,) * 12
        dy = _req(dy, "dy")
        B, H, W, Cs = x.shape
        M = B * H * W
        Cout, Cin = weight.shape[0], weight.shape[1]
        ldy = dy.shape[3]
        fork = side.mark()
        # ---- data gradient w.r.t. a = act(BN(x)), with act' and the BatchNorm-backward column sums in the epilogue
        wd = packs.get(weight, "dgrad", (1, Cin, 1, Cout, ldy, 0, 1, 1, Cin, 1))
        dgamma = _empty((C,), x) if sg is None else sg
        dbeta = _empty((C,), x) if sb is None else sb
        dz = _empty(x.shape, x)
        rows = lib().raw("vmtl_conv1x1_stats_rows")(M, Cs, ldy, 0 if d_a is None else 1)
        part = _empty((rows, 2, Cs), x)
        if d_a is None:
            _k("vmtl_conv1x1_bnbwd", _flop=2.0 * M * Cin * Cout, dy=dy, wp=wd, dz=dz, stats=part, ez_x=x, ez_mean=mean,
               ez_invstd=invstd, ez_gamma=gamma, ez_beta=beta, ez_act=act, M=M, Ks=ldy, ldy=Cs, Nw=Cin, Cout=Cin)
        else:  # + the gradient that reached a through its other consumers, added before act' and the reduction
            if act != ACT_NONE:
                raise NotImplementedError("bn_act_conv1x1: a second consumer of the activation needs act = none")
            _k("vmtl_conv1x1_bnbwd_add", _flop=2.0 * M * Cin * Cout, dy=dy, wp=wd, addend=_req(d_a, "d_a"), dz=dz, stats=part,
               ez_x=x, ez_mean=mean, ez_invstd=invstd, ez_gamma=gamma, ez_beta=beta, ez_act=act, M=M, Ks=ldy, ldy=Cs, Nw=Cin,
               Cout=Cin)
        _k("vmtl_bn_bwd_finalize", partial=part, nblk=rows, M=M, C=C, Cs=Cs, sum_dz=dbeta, sum_dzx=dgamma, mean=None,
           invstd=None, gamma=None, training=1 if training else 0, coef_a=None, coef_b=None, coef_c=None)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _empty(x.shape, x)
            _k("vmtl_bn_bwd_apply", x=x, dz=dz, mean=mean, invstd=invstd, gamma=gamma, sum_dz=dbeta, sum_dzx=dgamma, dx=dx,
               M=M, C=C, Cs=Cs, training=1 if training else 0)
        # ---- parameter gradients (side stream when they go to arena slots)
        dw = _empty(weight.shape, x) if sw is None else sw
        with side.branch(sw is not None, M, fork, dy, a):
            slabs, ns = _wgrad(a, dy, B, H, W, Cs, H, W, ldy, Cout, 1, 1, 1, 0, 2.0 * M * Cout * Cin)
            unpack(slabs, weight.shape, 1, Cout, 1, Cin, Cs, 0, Cin, 1, 1, out=dw, nslabs=ns)
        db = None
        if has_bias and ctx.needs_input_grad[9]:
            db = _bias_grad(ctx.bias, sbias, dy, M, Cout, ldy, zero_bias, fork)
        nif = lambda g, slot: None if slot is not None else g
        # act = none with a residual: d(res) is the gradient w.r.t. a itself = dz (no activation derivative in it)
        dres = dz if has_res and ctx.needs_input_grad[10] else None
        return dx, None, None, nif(dgamma, sg), nif(dbeta, sb), None, None, None, nif(dw, sw), db, dres, None


_BN_PW_MAX_ROWS = int(os.environ.get("VMTL_BN_PW_MAX_ROWS", str(1 << 30)))


def bn_act_conv1x1_supported(x, act):
    """Pointwise pre-activation node: pointwise-GEMM sized problems, activations with act(0) == 0."""
    return (_PW and x.shape[0] * x.shape[1] * x.shape[2] <= min(_PW_MAX_ROWS, _BN_PW_MAX_ROWS)
            and act in (ACT_NONE, ACT_RELU, ACT_HSWISH) and os.environ.get("VMTL_BN_PW", "1") != "0")


def bn_act_conv1x1(x, stats, rpb, bn, C, act, weight, bias=None, want_stats=True, zero_bias_grad=False, res=None,
                   return_act=False):
    """(y_raw, stats, rows_per_block[, a]) = conv1x1(a) (+ bias) with a = act(bn(x_raw)) [+ res]; bn = the nn.BatchNorm2d
    container of x's layer; res (act = none only): the skip connection added to the normalised map; return_act: also
    hand back a (differentiable) for its other consumers (the block's own skip branch, a decoder tap)."""
    if bn.momentum is None:
        raise NotImplementedError("BatchNorm2d(momentum=None) (cumulative moving average) is not implemented")
    cfg = (C, bn.training, float(bn.momentum), bn.eps, act, bool(want_stats), bool(zero_bias_grad), bool(return_act))
    out = _BNActPw.apply(x, stats, rpb, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                         weight, bias, res, cfg)
    y, ostats = out[0], out[1]
    orpb = 0
    if ostats is not None:
        orpb = lib().raw("vmtl_conv1x1_stats_block")(y.shape[0] * y.shape[1] * y.shape[2], y.shape[3], x.shape[3],
                                                     0 if res is None else 1)
    return (y, ostats, orpb, out[2]) if return_act else (y, ostats, orpb)


class _Conv1x1Cat(torch.autograd.Function):
    """(y, stats) = conv1x1(cat[xa, xb], weight) (+ bias) WITHOUT the concat (MTAN attention modules, reference
    models/mtan_model.py:57-59,139-141): the pointwise GEMM reads its K axis from two tensors, its data gradient
    writes the two input gradients directly, the weight gradient is taken per source into the two column ranges of dW.
    xa must fill its storage (Ca % 4 == 0) so that the ordinary packing of the (Cout, Ca + Cb) weight is the operand."""

    @staticmethod
    def forward(ctx, xa, xb, weight, bias, Cb, want_stats, zero_bias_grad):
        xa, xb, weight = _req(xa, "xa"), _req(xb, "xb"), _req(weight, "weight")
        B, H, W, Ca = xa.shape
        Cbs = xb.shape[3]
        Cout, Cin = weight.shape[0], weight.shape[1]
        if tuple(xb.shape[:3]) != (B, H, W) or tuple(weight.shape[2:]) != (1, 1) or Cin != Ca + Cb or ceil4(Cb) != Cbs:
            raise ValueError("conv1x1_cat: weight must be (Cout, Ca + Cb, 1, 1) over two maps of equal extent")
        M, Ks, ldy = B * H * W, Ca + Cbs, ceil4(Cout)
        wp = packs.get(weight, "fwd", (1, Cout, 1, Cin, Ks, 0, Cin, 1, 1, 0))
        y = _empty((B, H, W, ldy), xa)
        stats = None
        if want_stats:
            stats = _empty((lib().raw("vmtl_conv1x1_stats_rows")(M, ldy, Ks, 0), 2, ldy), xa)
        _k("vmtl_conv1x1_cat_fwd", _flop=2.0 * M * Cout * Cin, x=xa, K1=Ca, x2=xb, K2s=Cbs, wp=wp, bias=bias, y=y, stats=stats,
           M=M, ldy=ldy, Nw=Cout, Cout=Cout)
        ctx.save_for_backward(xa, xb, weight)
        ctx.cfg = (Cb, bias is not None, bool(zero_bias_grad))
        ctx.slots = (_slot(weight), _slot(bias))
        ctx.bias = bias
        ctx.set_materialize_grads(False)
        if want_stats:
            ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        xa, xb, weight = ctx.saved_tensors
        Cb, has_bias, zero_bias = ctx.cfg
        if dy is None:
            return (None,) * 7
        dy = _req(dy, "dy")
        B, H, W, Ca = xa.shape
        Cbs = xb.shape[3]
        Cout, Cin = weight.shape[0], weight.shape[1]
        M, ldy = B * H * W, dy.shape[3]
        sw, sbias = ctx.slots
        dxa = dxb = dw = db = None
        fork = side.mark()
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            wd = packs.get(weight, "dgrad", (1, Cin, 1, Cout, ldy, 0, 1, 1, Cin, 1))  # [Cin][ldy]
            dxa, dxb = _empty(xa.shape, xa), _empty(xb.shape, xa)
            _k("vmtl_conv1x1_cat_dgrad", _flop=2.0 * M * Cin * Cout, dy=dy, wp=wd, dx=dxa, N1=Ca, dx2=dxb, N2s=Cbs, N2=Cb,
               M=M, Ks=ldy)
        if ctx.needs_input_grad[2]:
            with side.branch(sw is not None, M, fork, xa, xb, dy):
                dwt = _empty(weight.shape, xa) if sw is None else sw
                flat = dwt.view(-1)
                Ks = Ca + Cbs
                ns = lib().raw("vmtl_conv2d_wgrad_splits")(M, Cout, Ks)
                slabs = _empty((ns, Cout, Ks), xa)
                _k("vmtl_conv1x1_cat_wgrad", _flop=2.0 * M * Cout * Cin, x=xa, K1=Ca, x2=xb, K2s=Cbs, dy=dy, slabs=slabs,
                   splits=ns, M=M, ldy=ldy, Nw=Cout)
                unpack(slabs, None, 1, Cout, 1, Cin, Ks, 0, Cin, 1, 1, out=flat, nslabs=ns)
            dw = None if sw is not None else dwt
        if has_bias and ctx.needs_input_grad[3]:
            db = _bias_grad(ctx.bias, sbias, dy, M, Cout, ldy, zero_bias, fork)
        return dxa, dxb, dw, db, None, None, None


def conv1x1_cat_supported(xa, Ca, xb):
    return _PW and Ca % 4 == 0 and xa.shape[3] == Ca and xa.shape[0] * xa.shape[1] * xa.shape[2] <= _PW_MAX_ROWS \
        and tuple(xa.shape[:3]) == tuple(xb.shape[:3]) and os.environ.get("VMTL_CAT_CONV", "1") != "0"


def conv1x1_cat(xa, xb, Cb, weight, bias=None, want_stats=False, zero_bias_grad=False):
    """conv1x1(cat[xa, xb]) without materialising the concat; returns (y, stats-or-None)."""
    y, stats = _Conv1x1Cat.apply(xa, xb, weight, bias, Cb, want_stats, zero_bias_grad)
    if stats is not None:
        stats._vmtl_rpb = lib().raw("vmtl_conv1x1_stats_block")(y.shape[0] * y.shape[1] * y.shape[2], y.shape[3],
                                                                xa.shape[3] + xb.shape[3], 0)
    return y, stats


class _Up2Conv(torch.autograd.Function):
    """conv3x3(pad 1, no bias)(cat[nearest_x2(xl), skip]) - the entry of every smp U-Net decoder block
    (reference utils/model_utils.py:25-34) - without materialising the upsample or the concat and with 4
    instead of 9 taps on the upsampled channels (four 2x2 phase convolutions on the low-res map).
    weight keeps the torch layout (Cout, C0 + C1, 3, 3), input channels ordered [xl | skip]."""

    @staticmethod
    def forward(ctx, xl, skip, weight, C0, want_stats):
        xl, weight = _req(xl, "xl"), _req(weight, "weight")
        B, H2, W2, C0s = xl.shape
        Cout, Cin = weight.shape[0], weight.shape[1]
        if tuple(weight.shape[2:]) != (3, 3):
            raise ValueError("up2_conv: 3x3 kernels only")
        if skip is not None:
            skip = _req(skip, "skip")
            C1s = skip.shape[3]
            if tuple(skip.shape[:3]) != (B, 2 * H2, 2 * W2):
                raise ValueError("up2_conv: skip must be at twice the resolution of xl")
        else:
            C1s = 0
        C1 = Cin - C0
        if ceil4(C0) != C0s or ceil4(C1) != C1s or (C1 > 0) != (skip is not None):
            raise ValueError(f"up2_conv: weight has {Cin} input channels, sources hold {C0s}+{C1s} storage channels")
        ldy = ceil4(Cout)
        Ktot = 4 * C0s + 9 * C1s
        wp = packs.get_custom(weight, "up2_fwd", (4, Cout, Ktot), lambda w, dst: _k(
            "vmtl_pack_up2_fwd", w=w, dst=dst, Cout=Cout, C0=C0, C0s=C0s, C1=C1, C1s=C1s))
        y = _empty((B, 2 * H2, 2 * W2, ldy), xl)
        stats = None
        M = B * 4 * H2 * W2
        # algorithmic FLOPs = the reference formulation (9 taps on every channel); executed: 4 taps on xl's
        ks = lib().raw("vmtl_conv2d_up2_ksplit")(B, H2, W2, ldy, Ktot)
        if ks > 1:  # tile-starved (deep decoder blocks at small batch): split K, no statistics epilogue
            _k("vmtl_conv2d_up2_fwd_ws", _flop=2.0 * M * Cout * 9 * Cin, _xflop=2.0 * M * Cout * (4 * C0 + 9 * C1), xl=xl,
               skip=skip, wp_eff=wp, y=y, ws=_empty((ks, B, 2 * H2, 2 * W2, ldy), xl), B=B, H2=H2, W2=W2, C0s=C0s, C1s=C1s,
               ldy=ldy, Cout=Cout)
        else:
            if want_stats:
                bm = lib().raw("vmtl_conv2d_up2_stats_block")(B, H2, W2, ldy)
                Mq = B * H2 * W2
                if Mq % bm == 0:
                    stats = _empty((4 * (Mq // bm), 2, ldy), xl)
            _k("vmtl_conv2d_up2_fwd", _flop=2.0 * M * Cout * 9 * Cin, _xflop=2.0 * M * Cout * (4 * C0 + 9 * C1), xl=xl,
               skip=skip, wp_eff=wp, y=y, stats=stats, B=B, H2=H2, W2=W2, C0s=C0s, C1s=C1s, ldy=ldy, Cout=Cout)
        ctx.save_for_backward(xl, skip, weight)
        ctx.cfg = (C0, C1, stats.shape[0] if stats is not None else 0)
        ctx.slot = _slot(weight)
        ctx.set_materialize_grads(False)
        if stats is not None:
            ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        xl, skip, weight = ctx.saved_tensors
        C0, C1, _ = ctx.cfg
        if dy is None:
            return None, None, None, None
        dy = _req(dy, "dy")
        B, H2, W2, C0s = xl.shape
        Cout, Cin = weight.shape[0], weight.shape[1]
        H, W, ldy = 2 * H2, 2 * W2, dy.shape[3]
        dxl = dskip = dw = None
        stamp(f"main up2 M={B * H * W} N={Cout} Cin={Cin}")
        fork = side.mark()
        if ctx.needs_input_grad[0]:  # 4x4 / stride 2 / pad 1 convolution over dY with pre-summed taps
            wd = packs.get_custom(weight, "up2_dgrad", (C0, 16 * ldy), lambda w, dst: _k(
                "vmtl_pack_up2_dgrad", w=w, dst=dst, Cout=Cout, Cos=ldy, C0=C0, Cin=Cin))
            dxl = _empty((B, H2, W2, C0s), xl)
            _conv_launch(dy, wd, None, dxl, None, B, H, W, ldy, H2, W2, C0s, C0, C0, 4, 4, 2, 1, cin=Cout,
                         algo_flop=2.0 * B * H * W * C0 * 9 * Cout)
        if skip is not None and ctx.needs_input_grad[1]:  # plain 3x3 data gradient restricted to the skip channels
            C1s = skip.shape[3]
            wds = packs.get(weight, "up2_dskip", (1, C1, 9, Cout, ldy, 0, 9, 1, Cin * 9, 1), offset=C0 * 9)
            dskip = _empty((B, H, W, C1s), xl)
            _conv_launch(dy, wds, None, dskip, None, B, H, W, ldy, H, W, C1s, C1, C1, 3, 3, 1, 1, cin=Cout)
        if ctx.needs_input_grad[2]:
            dw = _empty(weight.shape, xl) if ctx.slot is None else ctx.slot
            with side.branch(ctx.slot is not None, B * H * W, fork, dy, xl, skip):
                # low-res part: weight gradient of that 4x4/s2/p1 convolution (dY in the role of its input)
                slabs, ns = _wgrad(dy, xl, B, H, W, ldy, H2, W2, C0s, C0, 4, 4, 2, 1, 2.0 * B * H * W * Cout * 9 * C0,
                                   xflop=2.0 * B * H2 * W2 * C0 * 16 * Cout)
                _k("vmtl_unpack_up2", slabs=slabs, grad=dw, Cout=Cout, Cos=ldy, C0=C0, Cin=Cin, nslabs=ns)
                if skip is not None:
                    C1s = skip.shape[3]
                    slabs, ns = _wgrad(skip, dy, B, H, W, C1s, H, W, ldy, Cout, 3, 3, 1, 1,
                                       2.0 * B * H * W * Cout * 9 * C1)
                    unpack(slabs, None, 1, Cout, 9, C1, C1s, 0, Cin * 9, 1, 9, out=dw.view(-1)[C0 * 9:], nslabs=ns)
                stamp(f"side up2 M={B * H * W} N={Cout} Cin={Cin}")
            if ctx.slot is not None:
                dw = None
        return dxl, dskip, dw, None, None


class _BNActConv(torch.autograd.Function):
    """(y, stats) = conv3x3(act(BN(x)))   or, with up2,   conv3x3(cat[nearest_x2(act(BN(x))), skip])
    - one smp `Conv2dReLU` boundary (reference utils/model_utils.py:25-34, 72-76) taken in PRE-activation form:
    x is the RAW output of the previous conv (stats = its BatchNorm partial rows from the conv epilogue, rpb pixels
    each), y the raw output of this one.  Owning BatchNorm+activation AND the consuming conv in one autograd node
    lets the backward pass fuse them: the data gradient of the conv ends with the activation + BatchNorm backward of
    its own input (dz and its per-row-block column sums come out of the implicit GEMM's epilogue,
    vmtl_conv2d_bnbwd), so the BatchNorm backward keeps only its finalize and apply launches - no reduce pass."""

    @staticmethod
    def forward(ctx, x, stats, rpb, gamma, beta, rm, rv, nbt, weight, skip, cfg):
        C, training, momentum, eps, act, up2, want_stats = cfg
        x, weight = _req(x, "x"), _req(weight, "weight")
        B, H, W, Cs = x.shape
        M = B * H * W
        Cout, Cin = weight.shape[0], weight.shape[1]
        if ceil4(C) != Cs or tuple(weight.shape[2:]) != (3, 3):
            raise ValueError("bn_act_conv: 3x3 weight over x's channels expected")
        # ---- BatchNorm + activation (materialised: the weight gradient reads it)
        mean, invstd = _empty((Cs,), x), _empty((Cs,), x)
        a = _empty(x.shape, x)
        if training:
            if stats is not None:
                partial, nblk = stats, stats.shape[0]
            else:
                partial, nblk, rpb = _empty((_reduce_rows(M), 2, Cs), x), 0, 0
            if 0 < nblk <= _BN_FUSE_ROWS:
                _k("vmtl_bn_apply_fused", x=x, partial=partial, nblk=nblk, rows_per_blk=rpb, eps=eps, momentum=momentum,
                   running_mean=rm, running_var=rv, num_batches_tracked=nbt, save_mean=mean, save_invstd=invstd,
                   gamma=gamma, beta=beta, mul=None, res=None, y=a, M=M, C=C, Cs=Cs, act=act)
            else:
                _k("vmtl_bn_stats", x=x, M=M, C=C, Cs=Cs, partial=partial, nblk_from_conv=nblk, rows_per_blk_from_conv=rpb,
                   eps=eps, momentum=momentum, running_mean=rm, running_var=rv, num_batches_tracked=nbt, save_mean=mean,
                   save_invstd=invstd)
                _k("vmtl_bn_apply", x=x, mean=mean, invstd=invstd, gamma=gamma, beta=beta, mul=None, res=None, y=a, M=M,
                   C=C, Cs=Cs, act=act)
        else:
            _k("vmtl_bn_eval_stats", running_mean=rm, running_var=rv, C=C, Cs=Cs, eps=eps, save_mean=mean, save_invstd=invstd)
            _k("vmtl_bn_apply", x=x, mean=mean, invstd=invstd, gamma=gamma, beta=beta, mul=None, res=None, y=a, M=M, C=C,
               Cs=Cs, act=act)
        # ---- the conv on a
        ldy = ceil4(Cout)
        ostats, orpb = None, 0
        if up2:
            if skip is not None:
                skip = _req(skip, "skip")
                C1s = skip.shape[3]
                if tuple(skip.shape[:3]) != (B, 2 * H, 2 * W):
                    raise ValueError("bn_act_conv(up2): skip must be at twice the resolution of x")
            else:
                C1s = 0
            C1 = Cin - C
            if ceil4(C1) != C1s or (C1 > 0) != (skip is not None):
                raise ValueError("bn_act_conv(up2): weight channels do not match x + skip")
            Ktot = 4 * Cs + 9 * C1s
            wp = packs.get_custom(weight, "up2_fwd", (4, Cout, Ktot), lambda w, dst: _k(
                "vmtl_pack_up2_fwd", w=w, dst=dst, Cout=Cout, C0=C, C0s=Cs, C1=C1, C1s=C1s))
            y = _empty((B, 2 * H, 2 * W, ldy), x)
            Mo = 4 * M
            ks = lib().raw("vmtl_conv2d_up2_ksplit")(B, H, W, ldy, Ktot)
            if ks > 1:  # tile-starved (deep decoder blocks at small batch): split K, no statistics epilogue
                _k("vmtl_conv2d_up2_fwd_ws", _flop=2.0 * Mo * Cout * 9 * Cin, _xflop=2.0 * Mo * Cout * (4 * C + 9 * C1),
                   xl=a, skip=skip, wp_eff=wp, y=y, ws=_empty((ks, B, 2 * H, 2 * W, ldy), x), B=B, H2=H, W2=W, C0s=Cs,
                   C1s=C1s, ldy=ldy, Cout=Cout)
            else:
                if want_stats:
                    bm = lib().raw("vmtl_conv2d_up2_stats_block")(B, H, W, ldy)
                    if M % bm == 0:
                        ostats, orpb = _empty((4 * (M // bm), 2, ldy), x), bm
                _k("vmtl_conv2d_up2_fwd", _flop=2.0 * Mo * Cout * 9 * Cin, _xflop=2.0 * Mo * Cout * (4 * C + 9 * C1), xl=a,
                   skip=skip, wp_eff=wp, y=y, stats=ostats, B=B, H2=H, W2=W, C0s=Cs, C1s=C1s, ldy=ldy, Cout=Cout)
        else:
            if Cin != C or skip is not None:
                raise ValueError("bn_act_conv: weight expects x's channels (skip only with up2)")
            wp = packs.get(weight, "fwd", (1, Cout, 9, Cin, Cs, 0, Cin * 9, 1, 9, 0))
            y = _empty((B, H, W, ldy), x)
            if want_stats and conv_ksplit(B, H, W, Cs, ldy, 3, 3, 1, 1) == 1:
                rows, orpb = conv_stats_geometry(B, H, W, Cs, ldy, 3, 3, 1, 1)
                ostats = _empty((rows, 2, ldy), x)
            _conv_launch(a, wp, None, y, ostats, B, H, W, Cs, H, W, ldy, Cout, Cout, 3, 3, 1, 1, cin=Cin)
        ctx.save_for_backward(x, a, skip, weight, mean, invstd, gamma, beta)
        ctx.cfg = (C, training, act, up2)
        ctx.slots = (_slot(gamma), _slot(beta), _slot(weight))
        ctx.orpb = orpb
        ctx.set_materialize_grads(False)
        if ostats is not None:
            ctx.mark_non_differentiable(ostats)
        return y, ostats

    @staticmethod
    def backward(ctx, dy, _dstats):
        x, a, skip, weight, mean, invstd, gamma, beta = ctx.saved_tensors
        C, training, act, up2 = ctx.cfg
        sg, sb, sw = ctx.slots
        if dy is None:
            return (None,) * 11
        dy = _req(dy, "dy")
        B, H, W, Cs = x.shape
        M = B * H * W
        Cout, Cin = weight.shape[0], weight.shape[1]
        ldy = dy.shape[3]
        stamp(f"main bnconv M={dy.shape[0] * dy.shape[1] * dy.shape[2]} N={Cout} Cin={Cin}")
        fork = side.mark()
        # ---- data gradient w.r.t. a = act(BN(x)), with act' and the BatchNorm-backward column sums in the epilogue
        if up2:
            Hd, Wd = 2 * H, 2 * W  # dy's extent
            wd = packs.get_custom(weight, "up2_dgrad", (C, 16 * ldy), lambda w, dst: _k(
                "vmtl_pack_up2_dgrad", w=w, dst=dst, Cout=Cout, Cos=ldy, C0=C, Cin=Cin))
            geo = dict(B=B, H=Hd, W=Wd, Cs=ldy, Ho=H, Wo=W, ldy=Cs, Nw=C, Cout=C, KH=4, KW=4, stride=2, pad=1)
            flop, xflop = 2.0 * B * Hd * Wd * C * 9 * Cout, 2.0 * M * C * 16 * Cout
        else:
            Hd, Wd = H, W
            wd = packs.get(weight, "dgrad", (1, Cin, 9, Cout, ldy, 0, 9, 1, Cin * 9, 1))
            geo = dict(B=B, H=H, W=W, Cs=ldy, Ho=H, Wo=W, ldy=Cs, Nw=Cin, Cout=Cin, KH=3, KW=3, stride=1, pad=1)
            flop = xflop = 2.0 * M * Cin * 9 * Cout
        dgamma = _empty((C,), x) if sg is None else sg
        dbeta = _empty((C,), x) if sb is None else sb
        dx = _empty(x.shape, x) if ctx.needs_input_grad[0] else None
        fuse = lib().raw("vmtl_conv2d_ksplit")(B, H, W, Cs, geo["KH"] * geo["KW"] * ldy) <= 1 \
            and os.environ.get("VMTL_BNBWD_FUSE", "1") != "0"
        if fuse:
            dz = _empty(x.shape, x)
            if not up2 and _small_route(B, H, W, ldy, Cs, 3, 3, 1, 1, with_stats=True):
                # narrow layer: the same fused data gradient on the halo-tile kernel (ep_mode 2)
                rows = lib().raw("vmtl_conv3x3_small_stat_rows")(B, H, W)
                part = _empty((rows, 2, Cs), x)
                _small(dy, wd, dz, B, H, W, ldy, Cs, Cin, Cin, flop, stats=part, ep_mode=2,
                       ez=(x, mean, invstd, gamma, beta, act))
            else:
                rows = lib().raw("vmtl_conv2d_stats_rows")(B, H, W, Cs)
                part = _empty((rows, 2, Cs), x)
                _k("vmtl_conv2d_bnbwd", _flop=flop, _xflop=xflop, x=dy, wp=wd, y=dz, stats=part, ez_x=x, ez_mean=mean,
                   ez_invstd=invstd, ez_gamma=gamma, ez_beta=beta, ez_act=act, **geo)
            _k("vmtl_bn_bwd_finalize", partial=part, nblk=rows, M=M, C=C, Cs=Cs, sum_dz=dbeta, sum_dzx=dgamma, mean=None,
               invstd=None, gamma=None, training=1 if training else 0, coef_a=None, coef_b=None, coef_c=None)
            if dx is not None:
                _k("vmtl_bn_bwd_apply", x=x, dz=dz, mean=mean, invstd=invstd, gamma=gamma, sum_dz=dbeta, sum_dzx=dgamma,
                   dx=dx, M=M, C=C, Cs=Cs, training=1 if training else 0)
        else:  # split-K data gradient (tile-starved layers): unfused BatchNorm backward
            da = _empty(x.shape, x)
            _conv_launch(dy, wd, None, da, None, geo["B"], geo["H"], geo["W"], geo["Cs"], geo["Ho"], geo["Wo"], geo["ldy"],
                         geo["Nw"], geo["Cout"], geo["KH"], geo["KW"], geo["stride"], geo["pad"], cin=Cout, algo_flop=flop)
            part = _empty((_reduce_rows(M), 2, Cs), x)
            _k("vmtl_bn_bwd", x=x, dy=da, mean=mean, invstd=invstd, gamma=gamma, beta=beta, mul=None, dmul=None,
               partial=part, sum_dz=dbeta, sum_dzx=dgamma, dx=dx if dx is not None else _empty(x.shape, x), M=M, C=C, Cs=Cs,
               act=act, training=1 if training else 0)
        dskip = None
        if up2 and skip is not None and ctx.needs_input_grad[9]:
            C1, C1s = Cin - C, skip.shape[3]
            wds = packs.get(weight, "up2_dskip", (1, C1, 9, Cout, ldy, 0, 9, 1, Cin * 9, 1), offset=C * 9)
            dskip = _empty(skip.shape, x)
            _conv_launch(dy, wds, None, dskip, None, B, Hd, Wd, ldy, Hd, Wd, C1s, C1, C1, 3, 3, 1, 1, cin=Cout)
        # ---- weight gradient (side stream when it goes to an arena slot)
        dw = _empty(weight.shape, x) if sw is None else sw
        with side.branch(sw is not None, B * Hd * Wd, fork, dy, a, skip):
            if up2:
                slabs, ns = _wgrad(dy, a, B, Hd, Wd, ldy, H, W, Cs, C, 4, 4, 2, 1, 2.0 * B * Hd * Wd * Cout * 9 * C,
                                   xflop=2.0 * M * C * 16 * Cout)
                _k("vmtl_unpack_up2", slabs=slabs, grad=dw, Cout=Cout, Cos=ldy, C0=C, Cin=Cin, nslabs=ns)
                if skip is not None:
                    C1, C1s = Cin - C, skip.shape[3]
                    slabs, ns = _wgrad(skip, dy, B, Hd, Wd, C1s, Hd, Wd, ldy, Cout, 3, 3, 1, 1,
                                       2.0 * B * Hd * Wd * Cout * 9 * C1)
                    unpack(slabs, None, 1, Cout, 9, C1, C1s, 0, Cin * 9, 1, 9, out=dw.view(-1)[C * 9:], nslabs=ns)
            else:
                slabs, ns = _wgrad(a, dy, B, H, W, Cs, H, W, ldy, Cout, 3, 3, 1, 1, 2.0 * M * Cout * 9 * Cin)
                unpack(slabs, weight.shape, 1, Cout, 9, Cin, Cs, 0, Cin * 9, 1, 9, out=dw, nslabs=ns)
            stamp(f"side bnconv N={Cout} Cin={Cin}")
        nif = lambda g, slot: None if slot is not None else g
        return dx, None, None, nif(dgamma, sg), nif(dbeta, sb), None, None, None, nif(dw, sw), dskip, None


def bn_act_conv(x, stats, rpb, bn, C, act, weight, skip=None, up2=False, want_stats=True):
    """(y_raw, stats, rows_per_block) = conv3x3(act(bn(x_raw)))  (up2: nearest-x2 of it, concatenated with skip, first);
    bn is the nn.BatchNorm2d parameter container of x's layer, C its logical channel count."""
    if bn.momentum is None:
        raise NotImplementedError("BatchNorm2d(momentum=None) (cumulative moving average) is not implemented")
    cfg = (C, bn.training, float(bn.momentum), bn.eps, act, bool(up2), bool(want_stats))
    y, ostats = _BNActConv.apply(x, stats, rpb, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                 bn.num_batches_tracked, weight, skip, cfg)
    orpb = 0
    if ostats is not None:
        B, H, W, _ = x.shape
        ldy = y.shape[3]
        orpb = (lib().raw("vmtl_conv2d_up2_stats_block")(B, H, W, ldy) if up2
                else conv_stats_geometry(B, H, W, x.shape[3], ldy, 3, 3, 1, 1)[1])
    return y, ostats, orpb


def up2_conv(xl, C0, skip, weight, want_stats=False):
    """(y, stats) = conv3x3(cat[nearest_x2(xl), skip]); C0 = logical channels of xl; stats may be None.  Like conv2d's, the
    statistics rows carry the pixels each covers (`_vmtl_rpb`: the up2 tile picker's row block, not conv_pick_tile's)."""
    y, stats = _Up2Conv.apply(xl, skip, weight, C0, want_stats)
    if stats is not None:
        stats._vmtl_rpb = lib().raw("vmtl_conv2d_up2_stats_block")(xl.shape[0], xl.shape[1], xl.shape[2], y.shape[3])
    return y, stats


def conv2d(x, weight, bias=None, stride=1, pad=0, want_stats=False, zero_bias_grad=False, stitch=None):
    """zero_bias_grad: the caller normalises y with a TRAIN-mode BatchNorm next, which makes dL/dbias exactly zero.
    stitch = (CrossStitchLayer weights, task index): y = conv(w[task, task, (c)] * x), the scale folded into the operand."""
    y, stats = _Conv2d.apply(x, weight, bias, stride, pad, want_stats, zero_bias_grad,
                             None if stitch is None else stitch[0], 0 if stitch is None else int(stitch[1]))
    if stats is not None:  # pixels per statistics row, for whoever finalizes them (bn_act)
        stats._vmtl_rpb = conv_stats_geometry(y.shape[0], y.shape[1], y.shape[2], x.shape[3], y.shape[3], weight.shape[2],
                                              weight.shape[3], stride, pad)[1]
    return (y, stats) if want_stats else y


# ----------------------------------------------------------------------------- ConvTranspose2d(k=2, s=2)
class _ConvT2x2(torch.autograd.Function):
    """weight in torch (Cin, Cout, 2, 2) layout; reference models/mtan_model.py:214-216."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = _req(x, "x")
        weight = _req(weight, "weight")
        B, H, W, Cs = x.shape
        Cin, Cout = weight.shape[0], weight.shape[1]
        if tuple(weight.shape[2:]) != (2, 2) or ceil4(Cin) != Cs:
            raise ValueError("conv_transpose2x2: weight must be (Cin, Cout, 2, 2) matching the input channels")
        ldy = ceil4(Cout)
        wp = packs.get(weight, "ct_fwd", (4, Cout, 1, Cin, Cs, 1, 4, 0, Cout * 4, 0))
        y = _empty((B, 2 * H, 2 * W, ldy), x)
        _conv_launch(x, wp, bias, y, None, B, H, W, Cs, H, W, ldy, 4 * Cout, Cout, 1, 1, 1, 0, shuffle=1, cin=Cin)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.slots = (_slot(weight), _slot(bias))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = _req(dy, "dy")
        B, H, W, Cs = x.shape
        Cin, Cout = weight.shape[0], weight.shape[1]
        ldy = dy.shape[3]
        dx = dw = db = None
        fork = side.mark()
        if ctx.needs_input_grad[0]:  # a 2x2 / stride-2 conv over dy
            wd = packs.get(weight, "ct_bwd", (1, Cin, 4, Cout, ldy, 0, Cout * 4, 1, 4, 0))
            dx = _empty((B, H, W, Cs), x)
            _conv_launch(dy, wd, None, dx, None, B, 2 * H, 2 * W, ldy, H, W, Cs, Cin, Cin, 2, 2, 2, 0, cin=Cout)
        if ctx.needs_input_grad[1]:  # weight gradient of that same conv, with x in the role of its output gradient
            with side.branch(ctx.slots[0] is not None, B * H * W, fork, x, dy):
                slabs, ns = _wgrad(dy, x, B, 2 * H, 2 * W, ldy, H, W, Cs, Cin, 2, 2, 2, 0,
                                   2.0 * B * H * W * Cin * 4 * Cout)
                dw = unpack(slabs, weight.shape, 1, Cin, 4, Cout, ldy, 0, Cout * 4, 1, 4, out=ctx.slots[0], nslabs=ns)
            if ctx.slots[0] is not None:
                dw = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            with side.branch(ctx.slots[1] is not None, B * 4 * H * W, fork, dy):
                db = _colsum(dy, None, B * 4 * H * W, Cout, ldy, out=ctx.slots[1])
            if ctx.slots[1] is not None:
                db = None
        return dx, dw, db


def conv_transpose2x2(x, weight, bias=None):
    return _ConvT2x2.apply(x, weight, bias)


# ----------------------------------------------------------------------------- depthwise conv
class _DwConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, stride, pad):
        x = _req(x, "x")
        weight = _req(weight, "weight")
        B, H, W, Cs = x.shape
        C, one, K, K2 = weight.shape
        if one != 1 or K != K2 or ceil4(C) != Cs:
            raise ValueError("dwconv: weight must be (C, 1, K, K) matching the input channels")
        Ho = (H + 2 * pad - K) // stride + 1
        Wo = (W + 2 * pad - K) // stride + 1
        wp = packs.get(weight, "dw", (1, 1, K * K, C, Cs, 0, 0, 1, K * K, 0))
        y = _empty((B, Ho, Wo, Cs), x)
        _k("vmtl_dwconv_fwd", x=x, wp=wp, y=y, B=B, H=H, W=W, Cs=Cs, Ho=Ho, Wo=Wo, K=K, stride=stride, pad=pad)
        ctx.save_for_backward(x, weight, wp)
        ctx.cfg = (stride, pad)
        ctx.slot = _slot(weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, wp = ctx.saved_tensors
        stride, pad = ctx.cfg
        dy = _req(dy, "dy")
        B, H, W, Cs = x.shape
        C, _, K, _ = weight.shape
        _, Ho, Wo, _ = dy.shape
        dx = dw = None
        stamp(f"main dw M={B * Ho * Wo} C={C}")
        fork = side.mark()
        if ctx.needs_input_grad[0]:
            dx = _empty(x.shape, x)
            _k("vmtl_dwconv_bwd_data", dy=dy, wp=wp, dx=dx, B=B, H=H, W=W, Cs=Cs, Ho=Ho, Wo=Wo, K=K, stride=stride,
               pad=pad)
        if ctx.needs_input_grad[1]:
            with side.branch(ctx.slot is not None, B * Ho * Wo, fork, x, dy):
                partial = _empty((lib().raw("vmtl_dwconv_bwd_weight_rows")(B * Ho * Wo, Cs), K * K, Cs), x)
                dw = _empty(weight.shape, x) if ctx.slot is None else ctx.slot
                _k("vmtl_dwconv_bwd_weight", x=x, dy=dy, partial=partial, dw=dw, B=B, H=H, W=W, C=C, Cs=Cs, Ho=Ho,
                   Wo=Wo, K=K, stride=stride, pad=pad)
                stamp(f"side dw M={B * Ho * Wo} C={C}")
            if ctx.slot is not None:
                dw = None
        return dx, dw, None, None


class _BNActDw(torch.autograd.Function):
    """(y, stats) = dwconv(act(BN(x))): timm InvertedResidual's conv_pw -> bn1 -> act -> conv_dw boundary (reference
    utils/model_utils.py:25-34 [3P]) as a pre-activation node.  x is the RAW pointwise-conv output with its BatchNorm
    partial rows; normalise + activation run while the depthwise taps are loaded and the output's own BatchNorm
    partial rows come from the same pass (vmtl_dwconv_bn_fwd): two launches fewer per block than apply / dwconv /
    statistics sweep.  The activated input is written back once (a) for the weight gradient."""

    @staticmethod
    def forward(ctx, x, stats, rpb, gamma, beta, rm, rv, nbt, weight, cfg):
        C, training, momentum, eps, act, stride, pad, want_stats, return_act = cfg
        x, weight = _req(x, "x"), _req(weight, "weight")
        B, H, W, Cs = x.shape
        Cw, one, K, K2 = weight.shape
        if one != 1 or K != K2 or Cw != C or ceil4(C) != Cs or pad != (K - 1) // 2:
            raise ValueError("bn_act_dwconv: weight must be (C, 1, K, K) over x's channels with pad (K-1)//2")
        Ho = (H + 2 * pad - K) // stride + 1
        Wo = (W + 2 * pad - K) // stride + 1
        mean, invstd, ca, cc = _bn_fwd_coef(x, stats, rpb, gamma, beta, rm, rv, nbt, C, training, momentum, eps)
        wp = packs.get(weight, "dw", (1, 1, K * K, C, Cs, 0, 0, 1, K * K, 0))
        need_bwd = any(ctx.needs_input_grad)
        a = _empty(x.shape, x) if need_bwd or return_act else None
        y = _empty((B, Ho, Wo, Cs), x)
        ostats = None
        if want_stats:
            ostats = _empty((lib().raw("vmtl_dwconv_bn_stats_rows")(B * Ho * Wo, Cs), 2, Cs), x)
        _k("vmtl_dwconv_bn_fwd", x=x, coef_a=ca, coef_c=cc, act=act, wp=wp, y=y, a_out=a, partial=ostats, B=B, H=H, W=W,
           Cs=Cs, Ho=Ho, Wo=Wo, K=K, stride=stride, pad=pad)
        ctx.save_for_backward(x, a, weight, wp, mean, invstd, gamma, beta)
        ctx.cfg = (C, training, act, stride, pad)
        ctx.slots = (_slot(gamma), _slot(beta), _slot(weight))
        ctx.set_materialize_grads(False)
        if ostats is not None:
            ctx.mark_non_differentiable(ostats)
        if return_act:  # a = act(BN(x)) as a second differentiable output (the residual branch of the block reads it)
            return y, ostats, a
        return y, ostats

    @staticmethod
    def backward(ctx, dy, _dstats, d_a=None):
        x, a, weight, wp, mean, invstd, gamma, beta = ctx.saved_tensors
        C, training, act, stride, pad = ctx.cfg
        sg, sb, sw = ctx.slots
        if dy is None:
            if d_a is not None:
                raise NotImplementedError("bn_act_dwconv: gradient through the activation output only")
            return (None,) * 10
        dy = _req(dy, "dy")
        B, H, W, Cs = x.shape
        K = weight.shape[2]
        _, Ho, Wo, _ = dy.shape
        M = B * H * W
        stamp(f"main bndw M={B * Ho * Wo} C={C}")
        fork = side.mark()
        da = _empty(x.shape, x)
        if d_a is not None:  # + the gradient that reached a through its other consumer, added on the way out
            _k("vmtl_dwconv_bwd_data_add", dy=dy, wp=wp, addend=_req(d_a, "d_a"), dx=da, B=B, H=H, W=W, Cs=Cs, Ho=Ho, Wo=Wo,
               K=K, stride=stride, pad=pad)
        else:
            _k("vmtl_dwconv_bwd_data", dy=dy, wp=wp, dx=da, B=B, H=H, W=W, Cs=Cs, Ho=Ho, Wo=Wo, K=K, stride=stride, pad=pad)
        dgamma = _empty((C,), x) if sg is None else sg
        dbeta = _empty((C,), x) if sb is None else sb
        dx = _empty(x.shape, x)
        part = _empty((_reduce_rows(M), 2, Cs), x)
        _k("vmtl_bn_bwd", x=x, dy=da, mean=mean, invstd=invstd, gamma=gamma, beta=beta, mul=None, dmul=None, partial=part,
           sum_dz=dbeta, sum_dzx=dgamma, dx=dx, M=M, C=C, Cs=Cs, act=act, training=1 if training else 0)
        dw = _empty(weight.shape, x) if sw is None else sw
        with side.branch(sw is not None, B * Ho * Wo, fork, a, dy):
            partial = _empty((lib().raw("vmtl_dwconv_bwd_weight_rows")(B * Ho * Wo, Cs), K * K, Cs), x)
            _k("vmtl_dwconv_bwd_weight", x=a, dy=dy, partial=partial, dw=dw, B=B, H=H, W=W, C=C, Cs=Cs, Ho=Ho, Wo=Wo, K=K,
               stride=stride, pad=pad)
            stamp(f"side bndw C={C}")
        nif = lambda g, slot: None if slot is not None else g
        return dx, None, None, nif(dgamma, sg), nif(dbeta, sb), None, None, None, nif(dw, sw), None


def bn_act_dwconv(x, stats, rpb, bn, C, act, weight, stride=1, pad=1, want_stats=True, return_act=False):
    """(y_raw, stats, rows_per_block[, a]) = dwconv(act(bn(x_raw))); bn = the nn.BatchNorm2d container of x's layer;
    return_act: also hand back a = act(bn(x_raw)) (differentiable) for a second consumer of the activation."""
    if bn.momentum is None:
        raise NotImplementedError("BatchNorm2d(momentum=None) (cumulative moving average) is not implemented")
    cfg = (C, bn.training, float(bn.momentum), bn.eps, act, stride, pad, bool(want_stats), bool(return_act))
    out = _BNActDw.apply(x, stats, rpb, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                         bn.num_batches_tracked, weight, cfg)
    y, ostats = out[0], out[1]
    orpb = lib().raw("vmtl_dwconv_bn_stats_block")(y.shape[0] * y.shape[1] * y.shape[2], y.shape[3]) if ostats is not None else 0
    return (y, ostats, orpb, out[2]) if return_act else (y, ostats, orpb)


def dwconv(x, weight, stride=1, pad=1):
    return _DwConv.apply(x, weight, stride, pad)


# ----------------------------------------------------------------------------- BatchNorm + act (+ gate, + residual)
# statistics rows up to which BatchNorm finalize is folded into the apply launch (VMTL_BN_FUSE_ROWS overrides;
# the library accepts at most vmtl_bn_fuse_max_rows())
_BN_FUSE_ROWS = int(os.environ.get("VMTL_BN_FUSE_ROWS", "16"))  # measured: 16 rows neutral, 64 rows +0.5 ms/step (redundant fp64 merges)


class _BNAct(torch.autograd.Function):
    """y = act(BN(x)) [* mul] [+ res].  gamma/beta None -> plain activation (no normalisation)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, nbt, mul, res, stats, C, training, momentum, eps,
                act, stats_rpb=0):
        x = _req(x, "x")
        B, H, W, Cs = x.shape
        M = B * H * W
        mean = invstd = None
        if mul is not None:
            mul = _req(mul, "mul")
        if res is not None:
            res = _req(res, "res")
        y = _empty(x.shape, x)
        fused = False
        if gamma is not None:
            mean, invstd = _empty((Cs,), x), _empty((Cs,), x)
            if training:
                if stats is not None:
                    partial, nblk = stats, stats.shape[0]
                    rpb = stats_rpb if stats_rpb else lib().raw("vmtl_conv2d_stats_block")(B, H, W, Cs)
                    if nblk <= _BN_FUSE_ROWS:
                        # few statistics rows: every thread merges them itself, no separate finalize launch
                        _k("vmtl_bn_apply_fused", x=x, partial=partial, nblk=nblk, rows_per_blk=rpb, eps=eps,
                           momentum=momentum, running_mean=running_mean, running_var=running_var,
                           num_batches_tracked=nbt, save_mean=mean, save_invstd=invstd, gamma=gamma, beta=beta, mul=mul,
                           res=res, y=y, M=M, C=C, Cs=Cs, act=act)
                        fused = True
                else:
                    partial, nblk, rpb = _empty((_reduce_rows(M), 2, Cs), x), 0, 0
                if not fused:
                    _k("vmtl_bn_stats", x=x, M=M, C=C, Cs=Cs, partial=partial, nblk_from_conv=nblk,
                       rows_per_blk_from_conv=rpb, eps=eps,
                       momentum=momentum, running_mean=running_mean, running_var=running_var, num_batches_tracked=nbt,
                       save_mean=mean, save_invstd=invstd)
            else:
                _k("vmtl_bn_eval_stats", running_mean=running_mean, running_var=running_var, C=C, Cs=Cs, eps=eps,
                   save_mean=mean, save_invstd=invstd)
        if not fused:
            _k("vmtl_bn_apply", x=x, mean=mean, invstd=invstd, gamma=gamma, beta=beta, mul=mul, res=res, y=y, M=M, C=C,
               Cs=Cs, act=act)
        ctx.save_for_backward(x, gamma, beta, mean, invstd, mul)
        ctx.cfg = (C, training, act, res is not None)
        ctx.slots = (_slot(gamma), _slot(beta))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, invstd, mul = ctx.saved_tensors
        C, training, act, has_res = ctx.cfg
        dy = _req(dy, "dy")
        B, H, W, Cs = x.shape
        M = B * H * W
        need_sums = gamma is not None
        dmul = _empty(x.shape, x) if (mul is not None and ctx.needs_input_grad[6]) else None
        partial = sum_dz = sum_dzx = None
        if need_sums or dmul is not None:
            partial = _empty((_reduce_rows(M), 2, Cs), x)
        if need_sums:  # the kernels write exactly C entries: [dbeta | dgamma] may be arena slots
            sum_dzx = _empty((C,), x) if ctx.slots[0] is None else ctx.slots[0]
            sum_dz = _empty((C,), x) if ctx.slots[1] is None else ctx.slots[1]
        dx = _empty(x.shape, x)
        _k("vmtl_bn_bwd", x=x, dy=dy, mean=mean, invstd=invstd, gamma=gamma, beta=beta, mul=mul, dmul=dmul,
           partial=partial, sum_dz=sum_dz, sum_dzx=sum_dzx, dx=dx, M=M, C=C, Cs=Cs, act=act,
           training=1 if training else 0)
        dgamma = sum_dzx if (need_sums and ctx.slots[0] is None) else None
        dbeta = sum_dz if (need_sums and ctx.slots[1] is None) else None
        return (dx, dgamma, dbeta, None, None, None, dmul, dy if has_res else None, None, None, None, None, None,
                None, None)


def bn_act(x, gamma, beta, running_mean, running_var, nbt, C, training, momentum=0.1, eps=1e-5, act=ACT_NONE,
           mul=None, res=None, stats=None, stats_rpb=0):
    """stats: BatchNorm partial rows from the producing conv's epilogue, stats_rpb pixels each (0: taken from the
    `_vmtl_rpb` tag ops.conv2d puts on them, else the implicit-GEMM kernel's row block for this shape)."""
    if stats is not None and not stats_rpb:
        stats_rpb = getattr(stats, "_vmtl_rpb", 0)
    return _BNAct.apply(x, gamma, beta, running_mean, running_var, nbt, mul, res, stats, C, training, momentum, eps,
                        act, stats_rpb)


class _BNActPool(torch.autograd.Function):
    """maxpool2(act(BN(x))) as one node: the full-resolution activation and the full-resolution gradient of the pool never exist
    (vmtl_bn_act_pool2_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, nbt, stats, C, training, momentum, eps, act, stats_rpb):
        x = _req(x, "x")
        B, H, W, Cs = x.shape
        M = B * H * W
        mean, invstd = _empty((Cs,), x), _empty((Cs,), x)
        if training:
            if stats is not None:
                partial, nblk = stats, stats.shape[0]
                rpb = stats_rpb if stats_rpb else lib().raw("vmtl_conv2d_stats_block")(B, H, W, Cs)
            else:
                partial, nblk, rpb = _empty((_reduce_rows(M), 2, Cs), x), 0, 0
            _k("vmtl_bn_stats", x=x, M=M, C=C, Cs=Cs, partial=partial, nblk_from_conv=nblk, rows_per_blk_from_conv=rpb,
               eps=eps, momentum=momentum, running_mean=running_mean, running_var=running_var, num_batches_tracked=nbt,
               save_mean=mean, save_invstd=invstd)
        else:
            _k("vmtl_bn_eval_stats", running_mean=running_mean, running_var=running_var, C=C, Cs=Cs, eps=eps,
               save_mean=mean, save_invstd=invstd)
        y = _empty((B, H // 2, W // 2, Cs), x)
        _k("vmtl_bn_act_pool2_fwd", x=x, mean=mean, invstd=invstd, gamma=gamma, beta=beta, y=y, B=B, H=H, W=W, C=C, Cs=Cs,
           act=act)
        ctx.save_for_backward(x, gamma, beta, mean, invstd)
        ctx.cfg = (C, training, act)
        ctx.slots = (_slot(gamma), _slot(beta))
        return y

    @staticmethod
    def backward(ctx, dyp):
        x, gamma, beta, mean, invstd = ctx.saved_tensors
        C, training, act = ctx.cfg
        dyp = _req(dyp, "dy")
        B, H, W, Cs = x.shape
        partial = _empty((_reduce_rows(B * (H // 2) * (W // 2)), 2, Cs), x)
        sum_dzx = _empty((C,), x) if ctx.slots[0] is None else ctx.slots[0]  # exactly C entries: may be arena slots
        sum_dz = _empty((C,), x) if ctx.slots[1] is None else ctx.slots[1]
        dx = _empty(x.shape, x)
        _k("vmtl_bn_act_pool2_bwd", x=x, dyp=dyp, mean=mean, invstd=invstd, gamma=gamma, beta=beta, partial=partial,
           sum_dz=sum_dz, sum_dzx=sum_dzx, dx=dx, B=B, H=H, W=W, C=C, Cs=Cs, act=act, training=1 if training else 0)
        dgamma = sum_dzx if ctx.slots[0] is None else None
        dbeta = sum_dz if ctx.slots[1] is None else None
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None


FUSE_BN_POOL = os.environ.get("VMTL_FUSE_BN_POOL", "1") != "0"  # MTAN encoder attention: BN + ReLU + MaxPool2d as one node


def bn_act_pool2(x, gamma, beta, running_mean, running_var, nbt, C, training, momentum=0.1, eps=1e-5, act=ACT_NONE,
                 stats=None, stats_rpb=0):
    """maxpool2(act(BN(x))); falls back to the two separate nodes for odd extents (MaxPool2d floors) or when switched off."""
    B, H, W, Cs = x.shape
    if stats is not None and not stats_rpb:
        stats_rpb = getattr(stats, "_vmtl_rpb", 0)
    if not FUSE_BN_POOL or (H & 1) or (W & 1) or H < 2 or W < 2:
        return maxpool2(bn_act(x, gamma, beta, running_mean, running_var, nbt, C, training, momentum, eps, act, stats=stats,
                               stats_rpb=stats_rpb))
    return _BNActPool.apply(x, gamma, beta, running_mean, running_var, nbt, stats, C, training, momentum, eps, act, stats_rpb)


def activation(x, act, C, mul=None):
    """Plain activation (optionally times a gate operand) through the same fused kernel."""
    return _BNAct.apply(x, None, None, None, None, None, mul, None, None, C, False, 0.0, 0.0, act, 0)


# ----------------------------------------------------------------------------- concat / upsample / pad
class _Concat2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, Ca, Cb, up_a, up_b, H, W, off_a, off_b):
        a = _req(a, "a")
        B, Ha, Wa, Csa = a.shape
        if b is not None:
            b = _req(b, "b")
            _, Hb, Wb, Csb = b.shape
        else:
            Hb = Wb = Csb = 0
            Cb = 0
        Cd = ceil4(Ca + Cb)
        y = _empty((B, H, W, Cd), a)
        _k("vmtl_concat2", a=a, Ha=Ha, Wa=Wa, Ca=Ca, Csa=Csa, upa=up_a, oha=off_a[0], owa=off_a[1], b=b, Hb=Hb, Wb=Wb,
           Cb=Cb, Csb=Csb, upb=up_b, ohb=off_b[0], owb=off_b[1], y=y, B=B, H=H, W=W, Cd=Cd)
        ctx.cfg = (a.shape, None if b is None else b.shape, Ca, Cb, up_a, up_b, H, W, off_a, off_b)
        return y

    @staticmethod
    def backward(ctx, dy):
        ashape, bshape, Ca, Cb, up_a, up_b, H, W, off_a, off_b = ctx.cfg
        dy = _req(dy, "dy")
        B, _, _, Cd = dy.shape
        da = db = None
        if ctx.needs_input_grad[0]:
            da = _empty(ashape, dy)
            _k("vmtl_concat2_bwd", dy=dy, dx=da, B=B, H=H, W=W, Cd=Cd, c_off=0, Hs=ashape[1], Ws=ashape[2], C=Ca,
               Cs=ashape[3], up=up_a, oh=off_a[0], ow=off_a[1])
        if bshape is not None and ctx.needs_input_grad[1]:
            db = _empty(bshape, dy)
            _k("vmtl_concat2_bwd", dy=dy, dx=db, B=B, H=H, W=W, Cd=Cd, c_off=Ca, Hs=bshape[1], Ws=bshape[2], C=Cb,
               Cs=bshape[3], up=up_b, oh=off_b[0], ow=off_b[1])
        return da, db, None, None, None, None, None, None, None, None


def concat2(a, Ca, b=None, Cb=0, up_a=1, up_b=1, out_hw=None, off_a=(0, 0), off_b=(0, 0)):
    """Channel concat [a | b] into an (H, W) canvas; each source may be nearest-x2 upsampled
    and/or placed at an offset (zeros elsewhere).  b=None: pure upsample / pad of a."""
    if out_hw is None:
        out_hw = (a.shape[1] * up_a, a.shape[2] * up_a)
    return _Concat2.apply(a, b, Ca, Cb, up_a, up_b, out_hw[0], out_hw[1], tuple(off_a), tuple(off_b))


# ----------------------------------------------------------------------------- pooling / bilinear
class _MaxPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _req(x, "x")
        B, H, W, Cs = x.shape
        y = _empty((B, H // 2, W // 2, Cs), x)
        _k("vmtl_maxpool2_fwd", x=x, y=y, B=B, H=H, W=W, Cs=Cs)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _req(dy, "dy")
        B, H, W, Cs = x.shape
        dx = _empty(x.shape, x)
        _k("vmtl_maxpool2_bwd", x=x, dy=dy, dx=dx, B=B, H=H, W=W, Cs=Cs)
        return dx


def maxpool2(x):
    return _MaxPool2.apply(x)


class _BilinearUp2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _req(x, "x")
        B, H, W, Cs = x.shape
        y = _empty((B, 2 * H, 2 * W, Cs), x)
        _k("vmtl_bilinear_up2_fwd", x=x, y=y, B=B, H=H, W=W, Cs=Cs)
        ctx.shape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _req(dy, "dy")
        B, H, W, Cs = ctx.shape
        dx = _empty(ctx.shape, dy)
        _k("vmtl_bilinear_up2_bwd", dy=dy, dx=dx, B=B, H=H, W=W, Cs=Cs)
        return dx


def bilinear_up2(x):
    return _BilinearUp2.apply(x)


# ----------------------------------------------------------------------------- squeeze-excite pieces
class _SpatialMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _req(x, "x")
        B, H, W, Cs = x.shape
        y = _empty((B, 1, 1, Cs), x)
        _k("vmtl_spatial_mean", x=x, y=y, B=B, HW=H * W, Cs=Cs)
        ctx.shape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _req(dy, "dy")
        B, H, W, Cs = ctx.shape
        dx = _empty(ctx.shape, dy)
        _k("vmtl_channel_bcast", x=None, s=dy, y=dx, B=B, HW=H * W, Cs=Cs, mode=1)
        return dx


def spatial_mean(x):
    return _SpatialMean.apply(x)


class _ChannelScale(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s):
        x, s = _req(x, "x"), _req(s, "s")
        B, H, W, Cs = x.shape
        y = _empty(x.shape, x)
        _k("vmtl_channel_bcast", x=x, s=s, y=y, B=B, HW=H * W, Cs=Cs, mode=0)
        ctx.save_for_backward(x, s)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, s = ctx.saved_tensors
        dy = _req(dy, "dy")
        B, H, W, Cs = x.shape
        dx = ds = None
        if ctx.needs_input_grad[0]:
            dx = _empty(x.shape, x)
            _k("vmtl_channel_bcast", x=dy, s=s, y=dx, B=B, HW=H * W, Cs=Cs, mode=0)
        if ctx.needs_input_grad[1]:
            ds = _empty(s.shape, x)
            _k("vmtl_channel_scale_bwd_s", x=x, dy=dy, ds=ds, B=B, HW=H * W, Cs=Cs)
        return dx, ds


def channel_scale(x, s):
    return _ChannelScale.apply(x, s)


# ----------------------------------------------------------------------------- cross-stitch (diagonal scale)
class _SqueezeExcite(torch.autograd.Function):
    """y = x * act2(W_e act1(W_r mean_hw(x) + b_r) + b_e) - timm SqueezeExcite (ReLU / hard-sigmoid in
    MobileNetV3), the gate of the `basic` encoder's inverted-residual blocks - as four launches: per-image
    partial sums over HW, two batch-sized GEMMs (vmtl_fc_fwd: the first one finishes the mean while it loads
    its operand), one scale pass.  Backward: partial sums of dy*x, two GEMMs with the activation backward
    applied on load, one pass dx = dy*g + dmean/HW; the four parameter gradients come from two small
    launches in the torch layout (side stream when they go to arena slots).  Weights are the torch
    (R, C, 1, 1) / (C, R, 1, 1) conv parameters."""

    @staticmethod
    def forward(ctx, x, wr, br, we, be, act1, act2):
        x, wr, we = _req(x, "x"), _req(wr, "reduce weight"), _req(we, "expand weight")
        B, H, W, Cs = x.shape
        HW = H * W
        R, C = wr.shape[0], wr.shape[1]
        if ceil4(C) != Cs or tuple(we.shape[:2]) != (C, R) or br is None or be is None:
            raise ValueError("squeeze_excite: weights must be (R, C, 1, 1) / (C, R, 1, 1) with biases, matching x")
        if B > lib().raw("vmtl_fc_max_rows")():
            raise ValueError("squeeze_excite: batch too large for the batch-sized GEMM kernels")
        Rs = ceil4(R)
        S = lib().raw("vmtl_hw_reduce_parts")(B, HW, Cs)
        parts = _empty((S, B, Cs), x)
        _k("vmtl_hw_reduce", x=x, y=None, part=parts, B=B, HW=HW, Cs=Cs)
        z1, h = _empty((B, Rs), x), _empty((B, Rs), x)
        pooled = _empty((B, Cs), x)  # finished mean (a_out): every [b][c < Cs] is written since Cs <= ceil16(C)
        _k("vmtl_fc_fwd", a=parts, a_parts=S, a_part_stride=B * Cs, a_scale=1.0 / HW, a_z=None, a_act=0, a_out=pooled,
           w=wr, bias=br, z=z1, y=h, M=B, K=C, N=R, lda=Cs, ldw=C, ldy=Rs, act=act1)
        z2, g = _empty((B, Cs), x), _empty((B, Cs), x)
        _k("vmtl_fc_fwd", a=h, a_parts=1, a_part_stride=0, a_scale=1.0, a_z=None, a_act=0, a_out=None, w=we, bias=be,
           z=z2, y=g, M=B, K=R, N=C, lda=Rs, ldw=R, ldy=Cs, act=act2)
        y = _empty(x.shape, x)
        _k("vmtl_channel_scale_add", x=x, s=g, t=None, t_scale=0.0, y=y, B=B, HW=HW, Cs=Cs)
        ctx.save_for_backward(x, pooled, z1, h, z2, g, wr, we)
        ctx.S = S
        ctx.acts = (act1, act2)
        ctx.slots = (_slot(wr), _slot(br), _slot(we), _slot(be))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, pooled, z1, h, z2, g, wr, we = ctx.saved_tensors
        act1, act2 = ctx.acts
        dy = _req(dy, "dy")
        B, H, W, Cs = x.shape
        HW = H * W
        R, C = wr.shape[0], wr.shape[1]
        Rs, S = ceil4(R), ctx.S
        # dg[b][c] = sum_hw dy*x, left as per-slice partial sums for the GEMM to finish
        dparts = _empty((S, B, Cs), x)
        _k("vmtl_hw_reduce", x=dy, y=x, part=dparts, B=B, HW=HW, Cs=Cs)
        weT = packs.get(we, "dgrad", (1, R, 1, C, Cs, 0, 1, 1, R, 1))   # [R][Cs]
        dh = _empty((B, Rs), x)
        dg = _empty((B, Cs), x)  # finished sum_hw dy*x (a_out), for the weight gradient
        _k("vmtl_fc_fwd", a=dparts, a_parts=S, a_part_stride=B * Cs, a_scale=1.0, a_z=z2, a_act=act2, a_out=dg, w=weT,
           bias=None, z=None, y=dh, M=B, K=C, N=R, lda=Cs, ldw=Cs, ldy=Rs, act=0)
        fork = side.mark()
        dx = None
        if ctx.needs_input_grad[0]:
            wrT = packs.get(wr, "dgrad", (1, C, 1, R, Rs, 0, 1, 1, C, 1))  # [C][Rs]
            dmean = _empty((B, Cs), x)
            _k("vmtl_fc_fwd", a=dh, a_parts=1, a_part_stride=0, a_scale=1.0, a_z=z1, a_act=act1, a_out=None, w=wrT,
               bias=None, z=None, y=dmean, M=B, K=R, N=C, lda=Rs, ldw=Rs, ldy=Cs, act=0)
            dx = _empty(x.shape, x)
            _k("vmtl_channel_scale_add", x=dy, s=g, t=dmean, t_scale=1.0 / HW, y=dx, B=B, HW=HW, Cs=Cs)
        slots = ctx.slots
        all_slots = all(s is not None for s in slots)
        with side.branch(all_slots, B, fork, dg, dh, pooled, z1, h, z2):
            dwr = _empty(wr.shape, x) if slots[0] is None else slots[0]
            dbr = _empty((R,), x) if slots[1] is None else slots[1]
            dwe = _empty(we.shape, x) if slots[2] is None else slots[2]
            dbe = _empty((C,), x) if slots[3] is None else slots[3]
            _k("vmtl_fc_wgrad", x=h, x_parts=1, x_part_stride=0, x_scale=1.0, dyo=dg, dy_parts=1, dy_part_stride=0,
               zo=z2, dw=dwe, db=dbe, M=B, K=R, N=C, lda=Rs, ldn=Cs, act=act2)
            _k("vmtl_fc_wgrad", x=pooled, x_parts=1, x_part_stride=0, x_scale=1.0, dyo=dh, dy_parts=1,
               dy_part_stride=0, zo=z1, dw=dwr, db=dbr, M=B, K=C, N=R, lda=Cs, ldn=Rs, act=act1)
        ret = [None if sl is not None else t for t, sl in zip((dwr, dbr, dwe, dbe), slots)]
        return dx, ret[0], ret[1], ret[2], ret[3], None, None


def squeeze_excite(x, w_reduce, b_reduce, w_expand, b_expand, act1=ACT_RELU, act2=ACT_HSIGMOID):
    return _SqueezeExcite.apply(x, w_reduce, b_reduce, w_expand, b_expand, act1, act2)


def squeeze_excite_max_batch() -> int:
    return lib().raw("vmtl_fc_max_rows")()


class _Stitch(torch.autograd.Function):
    """y = w[a, a, (c)] * x for task a; `weights` is the full (T,T[,C]) parameter
    (reference models/cross_stitch_model.py:21-37).  Off-diagonal entries get zero gradient."""

    @staticmethod
    def forward(ctx, x, weights, task, C):
        x = _req(x, "x")
        weights = _req(weights, "weights")
        B, H, W, Cs = x.shape
        T = weights.shape[0]
        channel_wise = weights.dim() == 3
        # element offset of w[task, task, 0] and the stride between channels
        off = (task * T + task) * (weights.shape[2] if channel_wise else 1)
        wview = weights.view(-1)[off:]
        y = _empty(x.shape, x)
        # _flop slot = algorithmic BYTES here (8 B per element: read + write) for bench.py's HBM roofline
        _k("vmtl_stitch", _flop=8.0 * B * H * W * C, x=x, w=wview, y=y, M=B * H * W, C=C, Cs=Cs, wstride=1 if channel_wise else 0)
        ctx.save_for_backward(x, weights)
        ctx.cfg = (task, C, channel_wise, off)
        ctx.slot = _slot(weights)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weights = ctx.saved_tensors
        task, C, channel_wise, off = ctx.cfg
        dy = _req(dy, "dy")
        B, H, W, Cs = x.shape
        M = B * H * W
        dx = dw = None
        wv = weights.view(-1)[off:]
        ws = 1 if channel_wise else 0
        if ctx.needs_input_grad[1]:
            n = C if channel_wise else 1
            if ctx.slot is not None:  # arena slot (off-diagonal entries stay at their initial zero)
                out = ctx.slot.view(-1)[off:off + n]
            else:
                dw = torch.zeros_like(weights)
                out = dw.view(-1)[off:off + n]
            dx = _empty(x.shape, x) if ctx.needs_input_grad[0] else None
            # one sweep: dx = w * dy and the column sums of x * dy
            _k("vmtl_stitch_bwd", x=x, dy=dy, w=wv, dx=dx, partial=_empty((_reduce_rows(M) + 1, Cs), x), dw=out, M=M, C=C, Cs=Cs,
               wstride=ws, reduce_all=0 if channel_wise else 1)
        elif ctx.needs_input_grad[0]:
            dx = _empty(x.shape, x)
            _k("vmtl_stitch", x=dy, w=wv, y=dx, M=M, C=C, Cs=Cs, wstride=ws)
        return dx, dw, None, None


def stitch(x, weights, task, C):
    return _Stitch.apply(x, weights, task, C)


# ----------------------------------------------------------------------------- boundary layout + postprocess
class _ToNHWC(torch.autograd.Function):
    """(B,C,H,W) contiguous -> internal [B,H,W,Cs]."""

    @staticmethod
    def forward(ctx, x):
        x = _req(x, "x")
        B, C, H, W = x.shape
        Cs = ceil4(C)
        y = _empty((B, H, W, Cs), x)
        _k("vmtl_nchw_to_nhwc", x=x, y=y, B=B, C=C, HW=H * W, Cs=Cs, Cw=Cs)
        ctx.C = C
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _req(dy, "dy")
        B, H, W, Cs = dy.shape
        dx = _empty((B, ctx.C, H, W), dy)
        _k("vmtl_nhwc_to_nchw", x=dy, y=dx, B=B, C=ctx.C, HW=H * W, Cs=Cs)
        return dx


def to_nhwc(x):
    st = getattr(x, "_vmtl_nhwc", None)
    if st is not None and st.shape[0] == x.shape[0] and tuple(st.shape[1:3]) == tuple(x.shape[2:]) and not x.requires_grad:
        return st  # produced by hwc_to_model_input: already the internal storage, no relayout
    return _ToNHWC.apply(x)


def hwc_to_model_input(x_hwc, scale: float = 1.0):
    """(B,H,W,C) device tensor in dataset sample layout -> the reference's (B,C,H,W) input tensor, backed by the
    model's internal NHWC storage (zero pad channels) which to_nhwc() picks up without a second relayout."""
    x_hwc = _req(x_hwc, "img")
    B, H, W, C = x_hwc.shape
    Cs = ceil4(C)
    st = _empty((B, H, W, Cs), x_hwc)
    _k("vmtl_hwc_to_nhwc_pad", x=x_hwc, y=st, P=B * H * W, C=C, Cs=Cs, scale=scale)
    view = st[..., :C].permute(0, 3, 1, 2)
    view._vmtl_nhwc = st
    return view


class _ToNCHW(torch.autograd.Function):
    """internal [B,H,W,Cs] -> (B,C,H,W) contiguous, the reference's output layout."""

    @staticmethod
    def forward(ctx, x, C):
        x = _req(x, "x")
        B, H, W, Cs = x.shape
        y = _empty((B, C, H, W), x)
        _k("vmtl_nhwc_to_nchw", x=x, y=y, B=B, C=C, HW=H * W, Cs=Cs)
        ctx.Cs = Cs
        _remember_mode(ctx)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W = dy.shape
        ld = ctx.Cs
        _restore_mode(ctx)  # first backward node of a (task) network
        if (dy.is_cuda and dy.dtype == torch.float32 and dy.stride() == (H * W * ld, 1, W * ld, ld) and dy.storage_offset() == 0
                and dy.untyped_storage().nbytes() == 4 * B * H * W * ld and getattr(dy, "_vmtl_nhwc", None) is not None):
            # the cross-entropy backward already wrote this gradient as NHWC rows of exactly our width, pad lanes zero
            return torch.as_strided(dy, (B, H, W, ld), (H * W * ld, W * ld, ld, 1)), None
        dy = _req(dy, "dy")
        dx = _empty((B, H, W, ctx.Cs), dy)
        _k("vmtl_nchw_to_nhwc", x=dy, y=dx, B=B, C=C, HW=H * W, Cs=ctx.Cs, Cw=ctx.Cs)
        return dx, None


def to_nchw(x, C):
    return _ToNCHW.apply(x, C)


# ----------------------------------------------------------------------------- narrow full-resolution tail
def conv3x3_small_supported(Cs: int, Nw: int) -> bool:
    return bool(lib().raw("vmtl_conv3x3_small_supported")(Cs, Nw))


def _small(x, wp, y, B, H, W, Cs, ldy, Nw, Cout, flop, x2=None, pa=None, pb=None, pc=None, act_in=ACT_NONE, a_out=None,
           bias=None, yb=None, Ca=0, stats=None, ep_mode=0, ez=None):
    """One vmtl_conv3x3_small launch (csrc/conv_small.hip); ez = (x, mean, invstd, gamma, beta, act) for ep_mode 2."""
    ez = ez or (None, None, None, None, None, ACT_NONE)
    _k("vmtl_conv3x3_small", _flop=flop, x=x, x2=x2, pa=pa, pb=pb, pc=pc, act_in=act_in, a_out=a_out, wp=wp, bias=bias,
       y=y, yb=yb, Ca=Ca, stats=stats, ep_mode=ep_mode, ez_x=ez[0], ez_mean=ez[1], ez_invstd=ez[2], ez_gamma=ez[3],
       ez_beta=ez[4], ez_act=ez[5], B=B, H=H, W=W, Cs=Cs, ldy=ldy, Nw=Nw, Cout=Cout)


def _bn_fwd_coef(x, stats, rpb, gamma, beta, rm, rv, nbt, C, training, momentum, eps):
    """BatchNorm statistics (from conv-epilogue partial rows `stats` of `rpb` pixels each, or a sweep of x when
    stats is None; running buffers in eval mode) -> (mean, invstd, coef_a, coef_c) with
    BN(x) = coef_a * x + coef_c per channel (zeros on pad channels)."""
    B, H, W, Cs = x.shape
    M = B * H * W
    mean, invstd, ca, cc = (_empty((Cs,), x) for _ in range(4))
    if training:
        if stats is not None:
            partial, nblk = stats, stats.shape[0]
        else:
            partial, nblk, rpb = _empty((_reduce_rows(M), 2, Cs), x), 0, 0
        _k("vmtl_bn_stats_coef", x=x, M=M, C=C, Cs=Cs, partial=partial, nblk_from_conv=nblk, rows_per_blk_from_conv=rpb,
           eps=eps, momentum=momentum, running_mean=rm, running_var=rv, num_batches_tracked=nbt, save_mean=mean,
           save_invstd=invstd, gamma=gamma, beta=beta, coef_a=ca, coef_c=cc)
    else:
        _k("vmtl_bn_eval_stats_coef", running_mean=rm, running_var=rv, C=C, Cs=Cs, eps=eps, save_mean=mean,
           save_invstd=invstd, gamma=gamma, beta=beta, coef_a=ca, coef_c=cc)
    return mean, invstd, ca, cc


class _DecoderTail(torch.autograd.Function):
    """relu(BN1(x1)) -> conv2 3x3 -> relu(BN2(.)) -> {head a, head b} 3x3 (+bias), returned as the reference's two
    NCHW maps: the narrow full-resolution tail of `basic` (smp DecoderBlock conv2 of the last decoder block via
    reference utils/model_utils.py:25-34, segm_head / depth_head of models/basic_model.py:30-51) on
    vmtl_conv3x3_small.  x1 is the RAW output of the block's first conv (with its BatchNorm partial rows stats1):
    both normalise+ReLU passes run as the consumer conv's prologue, both BatchNorm-backward reductions as the
    producing data-gradient's epilogue, BN2's backward-apply as the prologue of conv2's data gradient."""

    @staticmethod
    def forward(ctx, x1, stats1, rpb1, g1, b1, rm1, rv1, nbt1, w2, g2, b2, rm2, rv2, nbt2, wa, ba, wb, bb, cfg):
        tr1, mom1, eps1, tr2, mom2, eps2 = cfg
        x1, w2, wa, wb = _req(x1, "x1"), _req(w2, "conv2 weight"), _req(wa, "head a weight"), _req(wb, "head b weight")
        B, H, W, Cs1 = x1.shape
        C2, C1 = w2.shape[0], w2.shape[1]
        Ca, Cb = wa.shape[0], wb.shape[0]
        N = Ca + Cb
        ldy2, ldyh = ceil4(C2), ceil4(N)
        if (ceil4(C1) != Cs1 or tuple(w2.shape[2:]) != (3, 3) or tuple(wa.shape[1:]) != (C2, 3, 3)
                or tuple(wb.shape[1:]) != (C2, 3, 3) or ba is None or bb is None):
            raise ValueError("decoder_tail: conv2 must be 3x3 over x1's channels and both heads 3x3 over conv2's, with biases")
        if not (conv3x3_small_supported(Cs1, C2) and conv3x3_small_supported(ldy2, N) and conv3x3_small_supported(ldyh, C2)
                and H % 4 == 0 and W % 32 == 0):
            raise ValueError("decoder_tail: shape not covered by vmtl_conv3x3_small (use decoder_tail_supported())")
        need_bwd = any(ctx.needs_input_grad)
        M = B * H * W
        tiles = lib().raw("vmtl_conv3x3_small_stat_rows")(B, H, W)  # statistics rows ...
        rpb = lib().raw("vmtl_conv3x3_small_stat_block")(B, H, W)   # ... of this many pixels each
        mean1, invstd1, pa1, pc1 = _bn_fwd_coef(x1, stats1, rpb1, g1, b1, rm1, rv1, nbt1, C1, tr1, mom1, eps1)
        wp2 = packs.get(w2, "fwd", (1, C2, 9, C1, Cs1, 0, C1 * 9, 1, 9, 0))
        a1 = _empty(x1.shape, x1) if need_bwd else None
        x2 = _empty((B, H, W, ldy2), x1)
        stats2 = _empty((tiles, 2, ldy2), x1) if tr2 else None
        _small(x1, wp2, x2, B, H, W, Cs1, ldy2, C2, C2, 2.0 * M * C2 * 9 * C1, pa=pa1, pc=pc1, act_in=ACT_RELU, a_out=a1,
               stats=stats2, ep_mode=1 if tr2 else 0)
        mean2, invstd2, pa2, pc2 = _bn_fwd_coef(x2, stats2, rpb, g2, b2, rm2, rv2, nbt2, C2, tr2, mom2, eps2)
        # both heads as ONE GEMM operand / bias vector, kept packed by the step's batched packing launch
        wph = packs.shared(wa, "heads_fwd", (N, 9 * ldy2))
        packs.get(wa, "heads_fwd", (1, Ca, 9, C2, ldy2, 0, C2 * 9, 1, 9, 0), out=wph[:Ca])
        packs.get(wb, "heads_fwd", (1, Cb, 9, C2, ldy2, 0, C2 * 9, 1, 9, 0), out=wph[Ca:])
        bias = packs.shared(wa, "heads_bias", (N,))
        packs.get(ba, "heads_bias", (1, 1, 1, Ca, Ca, 0, 0, 0, 1, 0), out=bias[:Ca])
        packs.get(bb, "heads_bias", (1, 1, 1, Cb, Cb, 0, 0, 0, 1, 0), out=bias[Ca:])
        a2 = _empty(x2.shape, x1) if need_bwd else None
        oa, ob = _empty((B, Ca, H, W), x1), _empty((B, Cb, H, W), x1)
        _small(x2, wph, oa, B, H, W, ldy2, ldyh, N, N, 2.0 * M * N * 9 * C2, pa=pa2, pc=pc2, act_in=ACT_RELU, a_out=a2,
               bias=bias, yb=ob, Ca=Ca)
        ctx.save_for_backward(x1, a1, x2, a2, mean1, invstd1, mean2, invstd2, g1, b1, g2, b2, w2, wa, wb)
        ctx.cfg = (tr1, tr2)
        ctx.slots = tuple(_slot(t) for t in (g1, b1, w2, g2, b2, wa, ba, wb, bb))
        _remember_mode(ctx)
        return oa, ob

    @staticmethod
    def backward(ctx, ga, gb):
        x1, a1, x2, a2, mean1, invstd1, mean2, invstd2, g1, b1, g2, b2, w2, wa, wb = ctx.saved_tensors
        tr1, tr2 = ctx.cfg
        _restore_mode(ctx)
        sg1, sb1, sw2, sg2, sb2, swa, sba, swb, sbb = ctx.slots
        B, H, W, Cs1 = x1.shape
        C2, C1 = w2.shape[0], w2.shape[1]
        Ca, Cb = wa.shape[0], wb.shape[0]
        N = Ca + Cb
        ldy2, ldyh = ceil4(C2), ceil4(N)
        M = B * H * W
        tiles = lib().raw("vmtl_conv3x3_small_stat_rows")(B, H, W)
        gb = torch.zeros((B, Cb, H, W), device=x1.device) if gb is None else _req(gb, "grad b")
        # head a's gradient may already sit in NHWC storage of the right width (the cross-entropy backward writes it so)
        dy = None
        if ga is not None and ga.stride() == (H * W * ldyh, 1, W * ldyh, ldyh) and ga.storage_offset() == 0 \
                and ga.dtype == torch.float32 and ga.untyped_storage().nbytes() == 4 * B * H * W * ldyh:
            dy = torch.as_strided(ga, (B, H, W, ldyh), (H * W * ldyh, W * ldyh, ldyh, 1))
        if dy is None:
            ga = torch.zeros((B, Ca, H, W), device=x1.device) if ga is None else _req(ga, "grad a")
            dy = _empty((B, H, W, ldyh), x1)
            _k("vmtl_nchw_to_nhwc", x=ga, y=dy.view(-1), B=B, C=Ca, HW=H * W, Cs=ldyh, Cw=Ca)
        dyf = dy.view(-1)
        _k("vmtl_nchw_to_nhwc", x=gb, y=dyf[Ca:], B=B, C=Cb, HW=H * W, Cs=ldyh, Cw=ldyh - Ca)  # also zeroes pad lanes
        stamp("main tail heads")
        fork = side.mark()
        # heads' data gradient; epilogue = ReLU + BatchNorm-2 backward reduction (dz2 and its per-tile column sums)
        wb_ref = weakref.ref(wb)  # the cache entry must not keep a parameter (and through it the whole model) alive

        def pack_heads_dgrad(w, dst):  # both heads' flipped weights as ONE dgrad operand [C2][9][ldyh]
            pack(w, 1, C2, 9, Ca, ldyh, 0, 9, 1, C2 * 9, flip=1, out=dst)
            _k("vmtl_pack_weights_slice", src=wb_ref(), dst=dst.view(-1)[Ca:], R0=C2, T=9, C=Cb, group=ldyh, sr0=9, st=1,
               sc=C2 * 9, flip=1)

        wd = packs.get_custom(wa, "heads_dgrad", (C2, 9 * ldyh), pack_heads_dgrad, deps=(wb,))
        dz2, part2 = _empty((B, H, W, ldy2), x1), _empty((tiles, 2, ldy2), x1)
        _small(dy, wd, dz2, B, H, W, ldyh, ldy2, C2, C2, 2.0 * M * C2 * 9 * N, stats=part2, ep_mode=2,
               ez=(x2, mean2, invstd2, g2, b2, ACT_RELU))
        with side.branch(all(s is not None for s in (swa, sba, swb, sbb)), M, fork, a2, dy):
            slabs, ns = _wgrad(a2, dy, B, H, W, ldy2, H, W, ldyh, N, 3, 3, 1, 1, 2.0 * M * N * 9 * C2)
            stride = N * 9 * ldy2
            dwa = unpack(slabs, wa.shape, 1, Ca, 9, C2, ldy2, 0, C2 * 9, 1, 9, out=swa, nslabs=ns, slab_stride=stride)
            dwb = unpack(slabs.view(-1)[Ca * 9 * ldy2:], wb.shape, 1, Cb, 9, C2, ldy2, 0, C2 * 9, 1, 9, out=swb, nslabs=ns,
                         slab_stride=stride)
            dbias = _colsum(dy, None, M, N, ldyh)
            dba = _empty((Ca,), x1) if sba is None else sba
            dbb = _empty((Cb,), x1) if sbb is None else sbb
            _copy_vec(dbias, dba, Ca)
            _copy_vec(dbias[Ca:], dbb, Cb)
            stamp("side tail heads")
        # BatchNorm-2 parameter gradients + its backward-apply as affine coefficients of (dz2, x2)
        dbeta2 = _empty((C2,), x1) if sb2 is None else sb2
        dgamma2 = _empty((C2,), x1) if sg2 is None else sg2
        cA, cB, cC = _empty((ldy2,), x1), _empty((ldy2,), x1), _empty((ldy2,), x1)
        _k("vmtl_bn_bwd_finalize", partial=part2, nblk=tiles, M=M, C=C2, Cs=ldy2, sum_dz=dbeta2, sum_dzx=dgamma2, mean=mean2,
           invstd=invstd2, gamma=g2, training=1 if tr2 else 0, coef_a=cA, coef_b=cB, coef_c=cC)
        stamp("main tail conv2")
        # conv2's data gradient: prologue dx2 = cA*dz2 + cB*x2 + cC (kept in dx2 for the weight gradient),
        # epilogue = ReLU + BatchNorm-1 backward reduction
        wd2 = packs.get(w2, "dgrad", (1, C1, 9, C2, ldy2, 0, 9, 1, C1 * 9, 1))
        dx2 = _empty(x2.shape, x1)
        dz1, part1 = _empty(x1.shape, x1), _empty((tiles, 2, Cs1), x1)
        _small(dz2, wd2, dz1, B, H, W, ldy2, Cs1, C1, C1, 2.0 * M * C1 * 9 * C2, x2=x2, pa=cA, pb=cB, pc=cC, a_out=dx2,
               stats=part1, ep_mode=2, ez=(x1, mean1, invstd1, g1, b1, ACT_RELU))
        fork = side.mark()  # AFTER the launch above: the weight gradient reads the dx2 it wrote
        dbeta1 = _empty((C1,), x1) if sb1 is None else sb1
        dgamma1 = _empty((C1,), x1) if sg1 is None else sg1
        _k("vmtl_bn_bwd_finalize", partial=part1, nblk=tiles, M=M, C=C1, Cs=Cs1, sum_dz=dbeta1, sum_dzx=dgamma1, mean=None,
           invstd=None, gamma=None, training=1 if tr1 else 0, coef_a=None, coef_b=None, coef_c=None)
        dx1 = None
        if ctx.needs_input_grad[0]:
            dx1 = _empty(x1.shape, x1)
            _k("vmtl_bn_bwd_apply", x=x1, dz=dz1, mean=mean1, invstd=invstd1, gamma=g1, sum_dz=dbeta1, sum_dzx=dgamma1,
               dx=dx1, M=M, C=C1, Cs=Cs1, training=1 if tr1 else 0)
        with side.branch(sw2 is not None, M, fork, a1, dx2):
            slabs, ns = _wgrad(a1, dx2, B, H, W, Cs1, H, W, ldy2, C2, 3, 3, 1, 1, 2.0 * M * C2 * 9 * C1)
            dw2 = unpack(slabs, w2.shape, 1, C2, 9, C1, Cs1, 0, C1 * 9, 1, 9, out=sw2, nslabs=ns)
            stamp("side tail conv2")
        nif = lambda g, slot: None if slot is not None else g
        return (dx1, None, None, nif(dgamma1, sg1), nif(dbeta1, sb1), None, None, None, nif(dw2, sw2), nif(dgamma2, sg2),
                nif(dbeta2, sb2), None, None, None, nif(dwa, swa), nif(dba, sba), nif(dwb, swb), nif(dbb, sbb), None)


def decoder_tail_supported(x1_shape, C1, C2, N) -> bool:
    B, H, W, Cs1 = x1_shape
    return (ceil4(C1) == Cs1 and conv3x3_small_supported(Cs1, C2) and conv3x3_small_supported(ceil4(C2), N)
            and conv3x3_small_supported(ceil4(N), C2) and H % 4 == 0 and W % 32 == 0
            and os.environ.get("VMTL_SMALL_TAIL", "1") != "0")


def decoder_tail(x1, stats1, rpb1, bn1, conv2_weight, bn2, wa, ba, wb, bb):
    """(head_a, head_b) NCHW = heads(relu(bn2(conv2(relu(bn1(x1)))))); bn1 / bn2 are nn.BatchNorm2d parameter
    containers, x1 the raw conv output feeding bn1 (stats1 = its conv-epilogue partial rows of rpb1 pixels, or None)."""
    def mom(bn):
        if bn.momentum is None:
            raise NotImplementedError("BatchNorm2d(momentum=None) (cumulative moving average) is not implemented")
        return float(bn.momentum)

    cfg = (bn1.training, mom(bn1), bn1.eps, bn2.training, mom(bn2), bn2.eps)
    return _DecoderTail.apply(x1, stats1, rpb1, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var,
                              bn1.num_batches_tracked, conv2_weight, bn2.weight, bn2.bias, bn2.running_mean,
                              bn2.running_var, bn2.num_batches_tracked, wa, ba, wb, bb, cfg)


def _copy_vec(src, dst, n):
    """dst[:n] = src[:n] for small per-channel vectors (the pack kernel in its degenerate 1x1x1 form)."""
    _k("vmtl_pack_weights", src=src, dst=dst, R1=1, R0=1, T=1, C=n, Cs=n, sr1=0, sr0=0, st=0, sc=1, flip=0)


class _DualHead(torch.autograd.Function):
    """Two KxK heads reading the same feature map (reference models/basic_model.py:30-51: segm_head and
    depth_head) as ONE implicit GEMM with N = Ca + Cb output channels.  Parameters stay separate torch
    tensors; outputs are the two contiguous NCHW maps the reference returns."""

    @staticmethod
    def forward(ctx, x, wa, ba, wb, bb, pad):
        x, wa, wb = _req(x, "x"), _req(wa, "weight a"), _req(wb, "weight b")
        B, H, W, Cs = x.shape
        Ca, Cin, KH, KW = wa.shape
        Cb = wb.shape[0]
        if tuple(wb.shape[1:]) != (Cin, KH, KW) or ceil4(Cin) != Cs or ba is None or bb is None:
            raise ValueError("dual_head: both heads must share input channels / kernel size and have a bias")
        N, KK = Ca + Cb, KH * KW
        ldy, Ktot = ceil4(N), KK * Cs
        wp = _empty((N, Ktot), x)
        pack(wa, 1, Ca, KK, Cin, Cs, 0, Cin * KK, 1, KK, out=wp[:Ca])
        pack(wb, 1, Cb, KK, Cin, Cs, 0, Cin * KK, 1, KK, out=wp[Ca:])
        bias = _empty((N,), x)
        _copy_vec(ba, bias, Ca)
        _copy_vec(bb, bias[Ca:], Cb)
        y = _empty((B, H, W, ldy), x)
        _conv_launch(x, wp, bias, y, None, B, H, W, Cs, H, W, ldy, N, N, KH, KW, 1, pad, cin=Cin)
        oa, ob = _empty((B, Ca, H, W), x), _empty((B, Cb, H, W), x)
        yf = y.view(-1)
        _k("vmtl_nhwc_to_nchw", x=yf, y=oa, B=B, C=Ca, HW=H * W, Cs=ldy)
        _k("vmtl_nhwc_to_nchw", x=yf[Ca:], y=ob, B=B, C=Cb, HW=H * W, Cs=ldy)
        ctx.save_for_backward(x, wa, wb)
        ctx.cfg = (pad, ldy)
        ctx.slots = (_slot(wa), _slot(ba), _slot(wb), _slot(bb))
        _remember_mode(ctx)
        return oa, ob

    @staticmethod
    def backward(ctx, ga, gb):
        x, wa, wb = ctx.saved_tensors
        pad, ldy = ctx.cfg
        _restore_mode(ctx)
        B, H, W, Cs = x.shape
        Ca, Cin, KH, KW = wa.shape
        Cb = wb.shape[0]
        N, KK = Ca + Cb, KH * KW
        ga = torch.zeros((B, Ca, H, W), device=x.device) if ga is None else _req(ga, "grad a")
        gb = torch.zeros((B, Cb, H, W), device=x.device) if gb is None else _req(gb, "grad b")
        dy = _empty((B, H, W, ldy), x)
        dyf = dy.view(-1)
        _k("vmtl_nchw_to_nhwc", x=ga, y=dyf, B=B, C=Ca, HW=H * W, Cs=ldy, Cw=Ca)
        _k("vmtl_nchw_to_nhwc", x=gb, y=dyf[Ca:], B=B, C=Cb, HW=H * W, Cs=ldy, Cw=ldy - Ca)  # also zeroes pad lanes
        dx = None
        fork = side.mark()
        if ctx.needs_input_grad[0]:
            # one tap-flipped, transposed operand [ci][tap'][co]: head a fills co < Ca and zeroes the rest of
            # every ldy-wide group, head b then fills co in [Ca, Ca+Cb)
            wd = pack(wa, 1, Cin, KK, Ca, ldy, 0, KK, 1, Cin * KK, flip=1)
            _k("vmtl_pack_weights_slice", src=wb, dst=wd.view(-1)[Ca:], R0=Cin, T=KK, C=Cb, group=ldy, sr0=KK, st=1,
               sc=Cin * KK, flip=1)
            dx = _empty((B, H, W, Cs), x)
            _conv_launch(dy, wd, None, dx, None, B, H, W, ldy, H, W, Cs, Cin, Cin, KH, KW, 1, KH - 1 - pad, cin=N)
        with side.branch(all(s is not None for s in ctx.slots), B * H * W, fork, x, dy):
            slabs, ns = _wgrad(x, dy, B, H, W, Cs, H, W, ldy, N, KH, KW, 1, pad, 2.0 * B * H * W * N * KK * Cin)
            stride = N * KK * Cs
            dwa = unpack(slabs, wa.shape, 1, Ca, KK, Cin, Cs, 0, Cin * KK, 1, KK, out=ctx.slots[0], nslabs=ns,
                         slab_stride=stride)
            dwb = unpack(slabs.view(-1)[Ca * KK * Cs:], wb.shape, 1, Cb, KK, Cin, Cs, 0, Cin * KK, 1, KK,
                         out=ctx.slots[2], nslabs=ns, slab_stride=stride)
            db = _colsum(dy, None, B * H * W, N, ldy)
            dba = _empty((Ca,), x) if ctx.slots[1] is None else ctx.slots[1]
            dbb = _empty((Cb,), x) if ctx.slots[3] is None else ctx.slots[3]
            _copy_vec(db, dba, Ca)
            _copy_vec(db[Ca:], dbb, Cb)
            stamp("side head")
        none_if = lambda g, slot: None if slot is not None else g
        return (dx, none_if(dwa, ctx.slots[0]), none_if(dba, ctx.slots[1]), none_if(dwb, ctx.slots[2]),
                none_if(dbb, ctx.slots[3]), None)


def dual_head(x, wa, ba, wb, bb, pad=1):
    return _DualHead.apply(x, wa, ba, wb, bb, pad)


class _Sigmoid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _req(x, "x")
        y = _empty(x.shape, x)
        _k("vmtl_eltwise", a=x, b=None, y=y, mode=1, total=x.numel())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _req(dy, "dy")
        dx = _empty(y.shape, y)
        _k("vmtl_eltwise", a=y, b=dy, y=dx, mode=2, total=y.numel())
        return dx


def sigmoid(x):
    return _Sigmoid.apply(x)


class _Fork(torch.autograd.Function):
    """n handles on one tensor for its n consumers (a block's residual branch and its first conv, an encoder feature and
    the next stage; MTAN's shared features feed the shared path AND both task attention modules: reference
    models/mtan_model.py:364-399).  Autograd would sum the gradients with n-1 ATen kernels; here the sum is ONE launch
    of this library (vmtl_add_n, up to 4 operands per launch), so the step runs no kernel the C ABI does not export."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)  # an unused handle contributes None, not a zero tensor to add
        return tuple(x.detach() for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [_req(g, "grad") for g in gs if g is not None]
        if not gs:
            return None, None
        while len(gs) > 1:
            k = min(4, len(gs))
            ops4 = gs[:k] + [None] * (4 - k)
            out = _empty(gs[0].shape, gs[0])
            _k("vmtl_add_n", a=ops4[0], b=ops4[1], c=ops4[2], d=ops4[3], y=out, total=out.numel())
            gs = [out] + gs[k:]
        return gs[0], None


def fork(x, n=2):
    """n independent handles on x (gradients summed by this library's kernel)."""
    return _Fork.apply(x, n)


class _AddScalars(torch.autograd.Function):
    """wa * a + wb * b for two scalar losses (reference lit_module.py:127-129) as one launch of this library."""

    @staticmethod
    def forward(ctx, a, b, wa, wb):
        a, b = _req(a, "a"), _req(b, "b")
        out = _empty(a.shape, a)
        _k("vmtl_axpby", a=a, b=b, y=out, wa=wa, wb=wb, total=a.numel())
        ctx.w = (wa, wb)
        return out

    @staticmethod
    def backward(ctx, g):
        wa, wb = ctx.w
        g = _req(g, "grad")
        if wa == 1.0 and wb == 1.0:
            return g, g, None, None
        ga, gb = _empty(g.shape, g), _empty(g.shape, g)
        zero = g  # b operand unused when its weight is 0
        _k("vmtl_axpby", a=g, b=zero, y=ga, wa=wa, wb=0.0, total=g.numel())
        _k("vmtl_axpby", a=g, b=zero, y=gb, wa=wb, wb=0.0, total=g.numel())
        return ga, gb, None, None


def add_losses(a, b, wa=1.0, wb=1.0):
    return _AddScalars.apply(a, b, float(wa), float(wb))


def argmax_channels(logits):
    """argmax over dim 1 of (B,C,H,W) logits -> int64 (B,H,W); reads NCHW or channels-last strides in place."""
    if not logits.is_cuda:
        raise RuntimeError("argmax_channels: tensor is not on the GPU (no CPU fallback on the hot path)")
    B, C, H, W = logits.shape
    z = logits.detach()
    if not (z.stride(3) * W == z.stride(2) and z.dtype == torch.float32):
        z = z.float().contiguous()
    out = torch.empty((B, H, W), dtype=torch.int64, device=z.device)
    _k("vmtl_argmax_channels", z=z, out=out, B=B, HW=H * W, C=C, sb=z.stride(0), sc=z.stride(1), sp=z.stride(3))
    return out


# ----------------------------------------------------------------------------- losses
class _CrossEntropy(torch.autograd.Function):
    """mean_{b,h,w} -log_softmax(logits)[target]; logits (B,C,H,W) NCHW-contiguous."""

    @staticmethod
    def forward(ctx, logits, target, want_argmax=False):
        logits = _req(logits, "logits")
        if target.dtype != torch.int64:
            raise TypeError("cross_entropy: target must be int64 class indices")
        target = target.contiguous()
        B, C, H, W = logits.shape
        if tuple(target.shape) != (B, H, W):
            raise ValueError(f"cross_entropy: target shape {tuple(target.shape)} does not match logits {(B, H, W)}")
        P = B * H * W
        loss = _empty((), logits)
        ws = torch.empty((lib().raw("vmtl_ce_workspace_bytes")(P) // 8,), dtype=torch.float64, device=logits.device)
        ctx.save_for_backward(logits, target)
        if want_argmax:
            pred = torch.empty((B, H, W), dtype=torch.int64, device=logits.device)
            _k("vmtl_ce_fwd_argmax", logits=logits, target=target, loss=loss, workspace=ws, argmax=pred, B=B, HW=H * W,
               C=C, sb=C * H * W, sc=H * W, sp=1)
            ctx.mark_non_differentiable(pred)
            return loss, pred
        _k("vmtl_ce_fwd", logits=logits, target=target, loss=loss, workspace=ws, B=B, HW=H * W, C=C,
           sb=C * H * W, sc=H * W, sp=1)
        return loss

    @staticmethod
    def backward(ctx, g, _gpred=None):
        logits, target = ctx.saved_tensors
        B, C, H, W = logits.shape
        g = _req(g, "grad_output")
        # the gradient is laid out NHWC with room for one more head ([B][H][W][ceil4(C+1)]) and returned as its
        # (B,C,H,W) view: a head conv that consumes it (ops.decoder_tail) takes the storage as its dY operand without
        # the NCHW -> NHWC relayout of this 80 MB tensor; any other consumer sees an ordinary strided tensor
        ld = ceil4(C + 1)
        st = _empty((B, H, W, ld), logits)
        _k("vmtl_ce_bwd_strided", logits=logits, target=target, grad_out=g, dlogits=st, B=B, HW=H * W, C=C,
           sb=C * H * W, sc=H * W, sp=1, dsb=H * W * ld, dsc=1, dsp=ld)
        dl = st[..., :C].permute(0, 3, 1, 2)
        dl._vmtl_nhwc = st
        return dl, None, None


def cross_entropy(logits, target):
    return _CrossEntropy.apply(logits, target, False)


def cross_entropy_with_argmax(logits, target):
    """(loss, argmax over the class axis) from one pass over the logits (reference lit_module.py:123 + 137-138)."""
    return _CrossEntropy.apply(logits, target, True)


class _SILog(torch.autograd.Function):
    """reference losses.py:29-36 on predictions in (0,1); pred and target hold the same number of elements."""

    @staticmethod
    def forward(ctx, pred, target, min_depth):
        pred, target = _req(pred, "pred"), _req(target, "target")
        if pred.numel() != target.numel():
            raise ValueError("silog: pred and target must have the same number of elements")
        P = pred.numel()
        loss, stats = _empty((), pred), _empty((3,), pred)
        ws = torch.empty((lib().raw("vmtl_silog_workspace_bytes")(P) // 8,), dtype=torch.float64, device=pred.device)
        _k("vmtl_silog_fwd", pred=pred, target=target, min_depth=min_depth, loss=loss, stats=stats, workspace=ws, P=P)
        ctx.save_for_backward(pred, target, stats)
        ctx.min_depth = min_depth
        return loss

    @staticmethod
    def backward(ctx, g):
        pred, target, stats = ctx.saved_tensors
        g = _req(g, "grad_output")
        dp = _empty(pred.shape, pred)
        _k("vmtl_silog_bwd", pred=pred, target=target, stats=stats, grad_out=g, min_depth=ctx.min_depth, dpred=dp,
           P=pred.numel())
        return dp, None, None


def silog(pred, target, min_depth=1e-3):
    return _SILog.apply(pred, target, float(min_depth))


class _L1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        pred, target = _req(pred, "pred"), _req(target, "target")
        if pred.numel() != target.numel():
            raise ValueError("l1: pred and target must have the same number of elements")
        P = pred.numel()
        loss = _empty((), pred)
        ws = torch.empty((lib().raw("vmtl_silog_workspace_bytes")(P) // 8,), dtype=torch.float64, device=pred.device)
        _k("vmtl_l1_fwd", pred=pred, target=target, loss=loss, workspace=ws, P=P)
        ctx.save_for_backward(pred, target)
        return loss

    @staticmethod
    def backward(ctx, g):
        pred, target = ctx.saved_tensors
        g = _req(g, "grad_output")
        dp = _empty(pred.shape, pred)
        _k("vmtl_l1_bwd", pred=pred, target=target, grad_out=g, dpred=dp, P=pred.numel())
        return dp, None


def l1_loss(pred, target):
    return _L1.apply(pred, target)


# ----------------------------------------------------------------------------- optimizer
def adam_step(p, g, m, v, step_t, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
    """Fused torch.optim.Adam update over flat fp32 buffers; step_t is a 1-element device
    tensor holding the (already incremented) step count."""
    _k("vmtl_adam_step", p=p, g=g, m=m, v=v, step_ptr=step_t, lr=lr, b1=betas[0], b2=betas[1], eps=eps,
       weight_decay=weight_decay, grad_scale=grad_scale, n=p.numel())
