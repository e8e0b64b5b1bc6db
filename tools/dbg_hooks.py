"""Tuning aid, NOT product code: launch-dropping ablations around vision_mtl_amd.ops._k.

    import tools.dbg_hooks as h; h.install(skip_side=True)              # main chain alone (WRONG gradients)
    h.install(skip={"vmtl_bn_bwd_finalize"})                            # upper bound of removing those launches
    h.install(extra_after=("vmtl_bn_stats", "vmtl_bn_stats_coef"), extra=4)  # marginal cost of a launch on the chain

Every mode makes the step compute WRONG results (that is the point: an upper bound of what removing the
launches would gain) - which is why these switches live here and not in the library.  Keep the data finite when
ablating: all-NaN operands toggle fewer bits, the chip clocks higher and the step looks faster than it is."""
import torch

from vision_mtl_amd import ops
from vision_mtl_amd._lib import lib

_orig = None


def install(skip_side=False, skip=(), extra_after=(), extra=0):
    global _orig
    if _orig is None:
        _orig = ops._k
    skip, extra_after = frozenset(skip), frozenset(extra_after)
    buf = torch.zeros(64, device="cuda") if extra else None

    def _k(name, _flop=None, _xflop=None, **kw):
        if ops._RECORD is not None and _flop is not None:
            ops._RECORD.append((name, dict(kw), float(_flop), float(_flop if _xflop is None else _xflop)))
        if (skip_side and ops.side.in_branch) or name in skip:
            return
        lib().callk(name, stream=ops._stream(), **kw)
        if extra and name in extra_after:
            for _ in range(extra):
                lib().callk("vmtl_fill_zero", p=buf, n=4, stream=ops._stream())

    ops._k = _k


def uninstall():
    global _orig
    if _orig is not None:
        ops._k, _orig = _orig, None
