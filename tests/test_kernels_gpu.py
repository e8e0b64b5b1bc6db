"""-m gpu: every HIP kernel family against plain PyTorch fp32 on the CPU (forward values and
all gradients), on shapes that include the odd channel counts of the `basic` decoder
(540/270/135/67/33), non-power-of-two maps and M/N/K tails.  Tolerance: 1e-4 of the
reference's max magnitude (BASELINE.json north_star: "within 1e-4 rel fp32")."""
import pytest
import torch
import torch.nn.functional as F

from tests.util import assert_close, from_dev_nhwc, to_dev_nhwc

pytestmark = pytest.mark.gpu


def _ops():
    from vision_mtl_amd import ops

    return ops


CONV_CASES = [
    # B, Cin, H, W, Cout, K, stride, pad, bias
    (2, 3, 16, 24, 16, 3, 2, 1, False),     # stem-like, Cin=3 -> Cs=4, stride 2
    (2, 67, 12, 20, 33, 3, 1, 1, False),    # decoder block 4 conv1 channels
    (1, 33, 9, 7, 33, 3, 1, 1, True),       # odd map, M tail
    (2, 151, 8, 8, 67, 3, 1, 1, False),
    (1, 294, 4, 8, 135, 3, 1, 1, False),
    (1, 580, 4, 4, 270, 3, 1, 1, False),
    (1, 1072, 2, 4, 540, 3, 1, 1, False),
    (3, 33, 16, 16, 19, 3, 1, 1, True),     # segm head
    (3, 33, 16, 16, 1, 3, 1, 1, True),      # depth head
    (2, 192, 8, 8, 128, 1, 1, 0, True),     # MTAN attention 1x1
    (4, 16, 6, 10, 64, 1, 1, 0, False),     # pointwise expand
    (5, 960, 1, 1, 240, 1, 1, 0, True),     # SE reduce on a (B,1,1,C) map
    # enough row tiles for the 128-row configurations with VALU tail columns (33 = 32+1, 20 = 16+4, 67 = 64+3)
    (4, 33, 112, 112, 33, 3, 1, 1, True),
    (4, 19, 112, 112, 20, 3, 1, 1, False),
    (2, 37, 160, 160, 67, 3, 1, 1, False),
    # 1x1 at M = 65536+: the persistent large-M pointwise kernel (forward 64x128 tiles, data gradient 64x96 tiles)
    (1, 192, 256, 256, 128, 1, 1, 0, True),
    (1, 128, 300, 300, 32, 1, 1, 0, True),   # M = 90000 (ragged tile), N = 32: 256x32 tiles... needs M*ldy >= 2^23: no -> old kernel
    (4, 128, 256, 256, 32, 1, 1, 0, False),  # M = 262144, N = 32: 256x32 tiles
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd_bwd(dev, case):
    ops = _ops()
    B, Cin, H, W, Cout, K, stride, pad, bias = case
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, generator=g) if bias else None
    need_dx = stride == 1
    xr = x.clone().requires_grad_(need_dx)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, stride=stride, padding=pad)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)

    xd = to_dev_nhwc(x, dev).requires_grad_(need_dx)
    wd = w.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True) if bias else None
    y, stats = ops.conv2d(xd, wd, bd, stride=stride, pad=pad, want_stats=True)
    assert_close(from_dev_nhwc(y, Cout), yr.detach(), what="conv fwd")
    # pad channels must be exactly zero
    assert y[..., Cout:].abs().max().item() == 0.0 if y.shape[-1] > Cout else True
    # fused BatchNorm partials: per row block (mean_b, M2_b); merged (Chan) they give the column mean / variance
    from vision_mtl_amd._lib import lib

    Ho, Wo = yr.shape[2], yr.shape[3]
    M = B * Ho * Wo
    if stats is None:  # a tile-starved shape runs split-K: no statistics epilogue, the BatchNorm sweeps y itself
        assert ops.conv_ksplit(B, Ho, Wo, xd.shape[-1], y.shape[-1], K, K, stride, pad) > 1
    else:
        rpb = stats._vmtl_rpb  # pixels per statistics row (the launch's row block: implicit GEMM or pointwise kernel)
        st = stats.double().cpu()
        nb = torch.tensor([max(0, min(rpb, M - b * rpb)) for b in range(st.shape[0])], dtype=torch.float64)[:, None]
        mean = (nb * st[:, 0]).sum(0) / M
        var = (st[:, 1] + nb * (st[:, 0] - mean) ** 2).sum(0) / M
        yo = yr.detach().double()
        assert_close(mean[:Cout], yo.mean((0, 2, 3)), tol=1e-5, atol=1e-6, what="conv stats mean")
        assert_close(var[:Cout], yo.var((0, 2, 3), unbiased=False), tol=1e-4, what="conv stats var")
    y.backward(to_dev_nhwc(gy, dev))
    assert_close(wd.grad.cpu(), wr.grad, what="conv wgrad")
    if need_dx:
        assert_close(from_dev_nhwc(xd.grad, Cin), xr.grad, what="conv dgrad")
        if xd.grad.shape[-1] > Cin:
            assert xd.grad[..., Cin:].abs().max().item() == 0.0
    if bias:
        assert_close(bd.grad.cpu(), br.grad, what="conv bias grad")


@pytest.mark.parametrize("tile", list(range(16)))
def test_conv2d_every_tile_config(dev, tile, vmtl_env):
    """Each implicit-GEMM tile configuration (ids in conv_igemm.hip, incl. the tail-column ones) forced
    through the tuning override on one ragged shape: values, pad zeros, BatchNorm partials, data gradient."""
    ops = _ops()
    from vision_mtl_amd._lib import lib

    vmtl_env("VMTL_FORCE_TILE", str(tile))
    B, Cin, H, W, Cout = 3, 37, 21, 19, 70
    g = torch.Generator().manual_seed(900 + tile)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    xr = x.clone().requires_grad_(True)
    yr = F.conv2d(xr, w, None, padding=1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = to_dev_nhwc(x, dev).requires_grad_(True)
    y, stats = ops.conv2d(xd, w.to(dev), None, stride=1, pad=1, want_stats=True)
    assert_close(from_dev_nhwc(y, Cout), yr.detach(), what=f"tile {tile} fwd")
    assert y[..., Cout:].abs().max().item() == 0.0
    M = B * H * W
    rpb = lib().raw("vmtl_conv2d_stats_block")(B, H, W, y.shape[-1])
    st = stats.double().cpu()
    nb = torch.tensor([max(0, min(rpb, M - b * rpb)) for b in range(st.shape[0])], dtype=torch.float64)[:, None]
    mean = (nb * st[:, 0]).sum(0) / M
    var = (st[:, 1] + nb * (st[:, 0] - mean) ** 2).sum(0) / M
    yo = yr.detach().double()
    assert_close(mean[:Cout], yo.mean((0, 2, 3)), tol=1e-5, atol=1e-6, what=f"tile {tile} stats mean")
    assert_close(var[:Cout], yo.var((0, 2, 3), unbiased=False), tol=1e-4, what=f"tile {tile} stats var")
    y.backward(to_dev_nhwc(gy, dev))
    assert_close(from_dev_nhwc(xd.grad, Cin), xr.grad, what=f"tile {tile} dgrad")
    assert xd.grad[..., Cin:].abs().max().item() == 0.0


@pytest.mark.parametrize("tile", [3, 4, 5, 6, 7, 10, 11])
def test_conv2d_bf16x3_operand_split(dev, tile, vmtl_env):
    """Opt-in VMTL_BF16X3=1: fp32 operands split exactly into three bf16 planes, six bf16 MFMAs per product.
    Held to a TIGHTER bar than the fp32-MFMA path (2e-6 of the output magnitude): it is an fp32-accurate
    formulation, not a reduced-precision one.  Forward values, BatchNorm partials, data gradient."""
    ops = _ops()
    vmtl_env("VMTL_FORCE_TILE", str(tile))
    vmtl_env("VMTL_KSPLIT_BLOCKS", "1")  # no split-K here: the statistics epilogue is part of what is compared
    B, Cin, H, W, Cout = 2, 120, 17, 23, 150  # K = 9*120 = 1080: above the K >= 768 gate of the variant
    g = torch.Generator().manual_seed(1900 + tile)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    yr = F.conv2d(x.double(), w.double(), None, padding=1)
    gy = torch.randn(yr.shape, generator=g)
    dxr = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), padding=1)
    out = {}
    for mode in ("0", "1"):
        vmtl_env("VMTL_BF16X3", mode)
        xd = to_dev_nhwc(x, dev).requires_grad_(True)
        y, stats = ops.conv2d(xd, w.to(dev), None, stride=1, pad=1, want_stats=True)
        y.backward(to_dev_nhwc(gy, dev))
        out[mode] = (from_dev_nhwc(y, Cout).double(), from_dev_nhwc(xd.grad, Cin).double(), stats.cpu())
    for mode, (y, dx, _) in out.items():
        ey = float((y.detach() - yr).abs().max() / yr.abs().max())
        ex = float((dx - dxr).abs().max() / dxr.abs().max())
        assert ey < 2e-6 and ex < 2e-6, f"VMTL_BF16X3={mode}: fwd err {ey:.2e}, dgrad err {ex:.2e}"
    assert not torch.equal(out["0"][0], out["1"][0])  # the switch really selected another kernel
    assert_close(out["1"][2], out["0"][2], tol=1e-5, atol=1e-6, what="BatchNorm partials under bf16x3")


@pytest.mark.parametrize("rows", [16, 32, 48, 64, 80, 144, 20, 36, 68, 128])
def test_conv2d_wgrad_every_row_config(dev, rows, vmtl_env):
    """Each weight-gradient tile height (incl. the VALU tail-row ones) forced through the tuning override."""
    ops = _ops()
    vmtl_env("VMTL_FORCE_WG_ROWS", str(rows))
    B, Cin, H, W, Cout = 2, 21, 19, 23, 70
    g = torch.Generator().manual_seed(700 + rows)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    wr = w.clone().requires_grad_(True)
    yr = F.conv2d(x, wr, None, padding=1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    wd = w.to(dev).requires_grad_(True)
    y = ops.conv2d(to_dev_nhwc(x, dev), wd, None, stride=1, pad=1)
    y.backward(to_dev_nhwc(gy, dev))
    assert_close(wd.grad.cpu(), wr.grad, what=f"wgrad rows {rows}")


def _wgrad_pair(dev, vmtl_env, B, Cin, H, W, Cout, K, stride, pad, seed):
    """Weight gradient through the product path with the scalar-chunk loader (default) and with the general loader."""
    ops = _ops()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    wr = w.clone().requires_grad_(True)
    yr = F.conv2d(x, wr, None, stride=stride, padding=pad)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    got = []
    for fast in ("1", "0"):
        vmtl_env("VMTL_WG_FAST", fast)
        wd = w.to(dev).requires_grad_(True)
        y = ops.conv2d(to_dev_nhwc(x, dev), wd, None, stride=stride, pad=pad)
        y.backward(to_dev_nhwc(gy, dev))
        got.append(wd.grad.cpu())
    return got, wr.grad


@pytest.mark.parametrize("rows", [16, 32, 48, 64, 80, 144, 20, 36, 68, 128])
def test_conv2d_wgrad_scalar_chunk_loader_every_row_config(dev, rows, vmtl_env):
    """Output width a multiple of 32 (whole-row chunks): the scalar-chunk loader of every tile height must reproduce the
    general loader bit for bit (same chunks, same MFMA order) and match torch; 3 images so slices cross image borders."""
    vmtl_env("VMTL_FORCE_WG_ROWS", str(rows))
    (fast, general), ref = _wgrad_pair(dev, vmtl_env, 3, 21, 5, 64, 70, 3, 1, 1, 900 + rows)
    assert torch.equal(fast, general)
    assert_close(fast, ref, what=f"wgrad rows {rows}, scalar-chunk loader")


@pytest.mark.parametrize("case", [(2, 3, 32, 32, 64, 3, 1, 1),    # first conv of MTAN at 32x32: Cs = 4, one row per chunk
                                  (2, 8, 16, 64, 24, 3, 2, 1),    # stride 2: Wo = 32
                                  (2, 12, 9, 96, 19, 3, 1, 1),    # Wo = 96: three chunks per row, 20-row tile
                                  (1, 32, 64, 128, 32, 3, 1, 1),  # Ktot = 288: the last kk tile has one live wave
                                  (3, 16, 8, 12, 40, 1, 1, 0),    # pointwise, M = 288 = 9 chunks: the flat form
                                  (2, 40, 7, 9, 24, 1, 1, 0)])    # pointwise, M = 126: not a multiple of 32 -> general loader
def test_conv2d_wgrad_scalar_chunk_loader_shapes(dev, case, vmtl_env):
    B, Cin, H, W, Cout, K, stride, pad = case
    (fast, general), ref = _wgrad_pair(dev, vmtl_env, B, Cin, H, W, Cout, K, stride, pad, 77)
    assert torch.equal(fast, general)
    assert_close(fast, ref, what=f"wgrad {case}")


@pytest.mark.parametrize("case", [(2, 32, 16, 64, 32),   # MTAN's 32 -> 32: 18 kk tiles, 2 co tiles
                                  (2, 33, 9, 32, 30),    # Cs = 36 (kk tiles straddle taps, 21 of them), one strip
                                  (1, 33, 70, 96, 20),   # heads: ldy = 20; 70 rows: several bands with a ragged last one
                                  (3, 16, 12, 64, 16),   # csnet
                                  (2, 32, 12, 64, 16),
                                  (2, 16, 10, 32, 1),    # depth head: one output channel
                                  (2, 3, 40, 64, 24),    # first conv: Cs = 4, 3 kk tiles
                                  (1, 20, 5, 128, 36)])   # 3 co tiles
def test_conv3x3_wgrad_small(dev, case, vmtl_env):
    """The strip-walking halo weight gradient against torch and against conv_wgrad_kernel (VMTL_WGRAD_SMALL=0), through the
    product path (ops.conv2d backward) and directly through the C ABI with the band geometry forced small."""
    from vision_mtl_amd._lib import lib

    ops = _ops()
    B, Cin, H, W, Cout = case
    Cs, ldy = (Cin + 3) // 4 * 4, (Cout + 3) // 4 * 4
    assert lib().raw("vmtl_conv3x3_wgrad_small_supported")(Cs, ldy, W) == 1
    g = torch.Generator().manual_seed(4321)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    wr = w.clone().requires_grad_(True)
    yr = F.conv2d(x, wr, None, padding=1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    got = {}
    for mode, target in (("small", None), ("small_bands", 4096), ("general", None)):
        vmtl_env("VMTL_WGRAD_SMALL", "0" if mode == "general" else "1")
        if target:
            vmtl_env("VMTL_WS_TARGET", str(target))  # many short strip segments: every band boundary / halo row is exercised
        wd = w.to(dev).requires_grad_(True)
        y = ops.conv2d(to_dev_nhwc(x, dev), wd, None, stride=1, pad=1)
        y.backward(to_dev_nhwc(gy, dev))
        got[mode] = wd.grad.cpu()
    for mode in ("small", "small_bands", "general"):
        assert_close(got[mode], wr.grad, what=f"wgrad {mode} {case}")
    assert not torch.equal(got["small"], got["general"]) or Cout * Cin < 64  # different kernels (summation orders) really ran


@pytest.mark.parametrize("case", [(2, 16, 8, 32, 16, False), (2, 32, 12, 64, 16, False), (1, 16, 9, 40, 32, True),
                                  (2, 16, 8, 32, 19, True), (2, 16, 6, 36, 1, True), (1, 33, 8, 64, 33, True),
                                  (2, 32, 8, 32, 32, False)])
def test_plain_narrow_conv_on_the_halo_tile_kernel(dev, case, monkeypatch):
    """Launches without a statistics epilogue of narrow 3x3 layers (data gradients, heads) are routed to
    vmtl_conv3x3_small above _SMALL_MIN_ROWS pixels: forced here at test sizes, forward (+ bias) and data gradient vs torch."""
    ops = _ops()
    recorded = []
    orig_k = ops._k

    def spy(name, *a, **kw):
        recorded.append(name)
        return orig_k(name, *a, **kw)

    monkeypatch.setattr(ops, "_SMALL_MIN_ROWS", 1)
    monkeypatch.setattr(ops, "_k", spy)
    B, Cin, H, W, Cout, bias = case
    g = torch.Generator().manual_seed(99)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g) if bias else None
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, padding=1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = to_dev_nhwc(x, dev).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True) if bias else None
    y = ops.conv2d(xd, wd, bd, stride=1, pad=1)  # no statistics requested
    assert_close(from_dev_nhwc(y, Cout), yr.detach(), what="conv fwd (halo-tile kernel)")
    if y.shape[-1] > Cout:
        assert y[..., Cout:].abs().max().item() == 0.0
    y.backward(to_dev_nhwc(gy, dev))
    # forward: the input's channel storage must be one of the kernel's widths; data gradient: dy's
    want = sum(((c + 3) // 4 * 4) in (16, 20, 32, 36) for c in (Cin, Cout))
    assert recorded.count("vmtl_conv3x3_small") == want and want >= 1, recorded
    assert_close(from_dev_nhwc(xd.grad, Cin), xr.grad, what="conv dgrad (halo-tile kernel)")
    if xd.grad.shape[-1] > Cin:
        assert xd.grad[..., Cin:].abs().max().item() == 0.0
    assert_close(wd.grad.cpu(), wr.grad, what="conv wgrad")
    if bias:
        assert_close(bd.grad.cpu(), br.grad, what="conv bias grad")


@pytest.mark.parametrize("case", [(2, 32, 8, 32, 32, True), (2, 16, 12, 64, 16, False), (1, 32, 8, 64, 16, False),
                                  (2, 16, 8, 32, 19, True)])
def test_narrow_conv_with_statistics_on_the_halo_tile_kernel(dev, case, monkeypatch):
    """Narrow 3x3 launches WITH the BatchNorm statistics epilogue (and a bias, as MTAN's attention convs have) on
    vmtl_conv3x3_small: values, the (mean, M2) partial rows under the geometry ops.conv_stats_geometry reports, gradients."""
    ops = _ops()
    recorded = []
    orig_k = ops._k
    monkeypatch.setattr(ops, "_SMALL_MIN_ROWS", 1)
    monkeypatch.setattr(ops, "_k", lambda name, *a, **kw: (recorded.append(name), orig_k(name, *a, **kw))[1])
    B, Cin, H, W, Cout, bias = case
    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g) if bias else None
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, b, padding=1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = to_dev_nhwc(x, dev).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    y, stats = ops.conv2d(xd, wd, None if b is None else b.to(dev), stride=1, pad=1, want_stats=True)
    assert "vmtl_conv3x3_small" in recorded and "vmtl_conv2d_fwd" not in recorded
    assert_close(from_dev_nhwc(y, Cout), yr.detach(), what="conv fwd")
    M, rpb = B * H * W, stats._vmtl_rpb
    assert stats.shape[0] * rpb == M
    st = stats.double().cpu()
    mean = st[:, 0].mean(0)
    var = (st[:, 1] + rpb * (st[:, 0] - mean) ** 2).sum(0) / M
    yo = yr.detach().double()
    assert_close(mean[:Cout], yo.mean((0, 2, 3)), tol=1e-5, atol=1e-6, what="stats mean")
    assert_close(var[:Cout], yo.var((0, 2, 3), unbiased=False), tol=1e-4, what="stats var")
    y.backward(to_dev_nhwc(gy, dev))
    assert_close(from_dev_nhwc(xd.grad, Cin), xr.grad, what="conv dgrad")
    assert_close(wd.grad.cpu(), wr.grad, what="conv wgrad")


@pytest.mark.parametrize("case", [(2, 64, 5, 7, 32, True), (1, 512, 4, 4, 256, True), (3, 8, 3, 3, 5, False)])
def test_conv_transpose2x2(dev, case):
    ops = _ops()
    B, Cin, H, W, Cout, bias = case
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cin, Cout, 2, 2, generator=g) / Cin ** 0.5
    b = torch.randn(Cout, generator=g) if bias else None
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    yr = F.conv_transpose2d(xr, wr, br, stride=2)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = to_dev_nhwc(x, dev).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True) if bias else None
    y = ops.conv_transpose2x2(xd, wd, bd)
    assert_close(from_dev_nhwc(y, Cout), yr.detach(), what="convT fwd")
    if y.shape[-1] > Cout:
        assert y[..., Cout:].abs().max().item() == 0.0
    y.backward(to_dev_nhwc(gy, dev))
    assert_close(from_dev_nhwc(xd.grad, Cin), xr.grad, what="convT dgrad")
    assert_close(wd.grad.cpu(), wr.grad, what="convT wgrad")
    if bias:
        assert_close(bd.grad.cpu(), br.grad, what="convT bias grad")


@pytest.mark.parametrize("case", [(2, 16, 12, 20, 3, 1), (2, 64, 12, 20, 3, 2), (2, 72, 9, 11, 5, 2),
                                  (1, 120, 8, 8, 5, 1), (2, 200, 4, 6, 3, 1), (1, 672, 4, 8, 5, 2)])
def test_dwconv(dev, case):
    ops = _ops()
    B, C, H, W, K, stride = case
    pad = (K - 1) // 2
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, 1, K, K, generator=g) / K
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride=stride, padding=pad, groups=C)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = to_dev_nhwc(x, dev).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    y = ops.dwconv(xd, wd, stride=stride, pad=pad)
    assert_close(from_dev_nhwc(y, C), yr.detach(), what="dwconv fwd")
    y.backward(to_dev_nhwc(gy, dev))
    assert_close(from_dev_nhwc(xd.grad, C), xr.grad, what="dwconv dgrad")
    assert_close(wd.grad.cpu(), wr.grad, what="dwconv wgrad")


@pytest.mark.parametrize("case", [(2, 135, 16, 4, 6, 67), (1, 960, 112, 3, 5, 540), (2, 67, 0, 8, 8, 33),
                                  (3, 270, 24, 5, 3, 135), (2, 20, 7, 1, 1, 9),
                                  (2, 40, 8, 80, 80, 33), (1, 24, 6, 112, 112, 67)])  # tail-column tiles
def test_up2_conv(dev, case):
    _check_up2(dev, case)


@pytest.mark.parametrize("tile", [0, 3, 5, 9, 12, 13, 14, 15])
def test_up2_conv_forced_tile(dev, tile, vmtl_env):
    vmtl_env("VMTL_FORCE_TILE", str(tile))
    _check_up2(dev, (2, 24, 6, 10, 12, 35))


def _check_up2(dev, case):
    """Phase-decomposed decoder-block entry == conv3x3(cat[nearest_x2(x), skip]) (values, fused BatchNorm
    partials, and the gradients of x, skip and the weight)."""
    ops = _ops()
    B, C0, C1, H2, W2, Cout = case
    g = torch.Generator().manual_seed(47)
    x = torch.randn(B, C0, H2, W2, generator=g)
    sk = torch.randn(B, C1, 2 * H2, 2 * W2, generator=g) if C1 else None
    w = torch.randn(Cout, C0 + C1, 3, 3, generator=g) / ((C0 + C1) * 9) ** 0.5
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    skr = sk.clone().requires_grad_(True) if C1 else None
    up = F.interpolate(xr, scale_factor=2, mode="nearest")
    yr = F.conv2d(torch.cat([up, skr], 1) if C1 else up, wr, None, padding=1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = to_dev_nhwc(x, dev).requires_grad_(True)
    skd = to_dev_nhwc(sk, dev).requires_grad_(True) if C1 else None
    wd = w.to(dev).requires_grad_(True)
    y, stats = ops.up2_conv(xd, C0, skd, wd, want_stats=True)
    assert_close(from_dev_nhwc(y, Cout), yr.detach(), what="up2 fwd")
    if y.shape[-1] > Cout:
        assert y[..., Cout:].abs().max().item() == 0.0
    if stats is not None:
        from vision_mtl_amd._lib import lib

        Mq = B * H2 * W2
        rpb = lib().raw("vmtl_conv2d_up2_stats_block")(B, H2, W2, y.shape[-1])
        assert Mq % rpb == 0 and stats.shape[0] == 4 * (Mq // rpb)
        st = stats.double().cpu()
        mean = st[:, 0].mean(0)  # equal-sized blocks
        var = (st[:, 1] + rpb * (st[:, 0] - mean) ** 2).sum(0) / (4 * Mq)
        yo = yr.detach().double()
        assert_close(mean[:Cout], yo.mean((0, 2, 3)), tol=1e-5, atol=1e-6, what="up2 stats mean")
        assert_close(var[:Cout], yo.var((0, 2, 3), unbiased=False), tol=1e-4, what="up2 stats var")
    y.backward(to_dev_nhwc(gy, dev))
    assert_close(from_dev_nhwc(xd.grad, C0), xr.grad, what="up2 dx")
    assert_close(wd.grad.cpu(), wr.grad, what="up2 dw")
    if C1:
        assert_close(from_dev_nhwc(skd.grad, C1), skr.grad, what="up2 dskip")


@pytest.mark.parametrize("bias", [False, True])
def test_conv2d_forward_split_k(dev, bias):
    """Tile-starved forward convs (decoder blocks 0-1 at small batch) run as K slices + a slab sum that also adds the
    bias; they hand back no statistics rows.  Values, pad zeros and all gradients against torch."""
    ops = _ops()
    B, Cin, H, W, Cout = 2, 130, 12, 12, 70
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g) if bias else None
    ref = [t.clone().requires_grad_(True) for t in (x, w)] + ([b.clone().requires_grad_(True)] if bias else [])
    yr = F.conv2d(ref[0], ref[1], ref[2] if bias else None, padding=1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = to_dev_nhwc(x, dev).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True) if bias else None
    assert ops.conv_ksplit(B, H, W, xd.shape[-1], 72, 3, 3, 1, 1) > 1
    y, stats = ops.conv2d(xd, wd, bd, stride=1, pad=1, want_stats=True)
    assert stats is None
    assert_close(from_dev_nhwc(y, Cout), yr.detach(), what="split-K fwd")
    assert y[..., Cout:].abs().max().item() == 0.0
    y.backward(to_dev_nhwc(gy, dev))
    assert_close(from_dev_nhwc(xd.grad, Cin), ref[0].grad, what="split-K dgrad")
    assert_close(wd.grad.cpu(), ref[1].grad, what="split-K wgrad")
    if bias:
        assert_close(bd.grad.cpu(), ref[2].grad, what="split-K bias grad")


def test_up2_conv_split_k(dev):
    """The phase-decomposed decoder-block entry with its K loop split over workgroups (deep blocks at small batch)."""
    from vision_mtl_amd._lib import lib

    case = (1, 240, 40, 6, 8, 70)
    assert lib().raw("vmtl_conv2d_up2_ksplit")(1, 6, 8, 72, 4 * 240 + 9 * 40) > 1
    _check_up2(dev, case)


def test_dual_head(dev):
    """Fused segm+depth heads == two separate F.conv2d heads (values and all gradients)."""
    ops = _ops()
    g = torch.Generator().manual_seed(43)
    B, Cin, H, W, Ca, Cb = 2, 33, 12, 20, 19, 1
    x = torch.randn(B, Cin, H, W, generator=g)
    wa, wb = torch.randn(Ca, Cin, 3, 3, generator=g) * 0.1, torch.randn(Cb, Cin, 3, 3, generator=g) * 0.1
    ba, bb = torch.randn(Ca, generator=g), torch.randn(Cb, generator=g)
    ga, gb = torch.randn(B, Ca, H, W, generator=g), torch.randn(B, Cb, H, W, generator=g)
    ref = [t.clone().requires_grad_(True) for t in (x, wa, ba, wb, bb)]
    ya, yb = F.conv2d(ref[0], ref[1], ref[2], padding=1), F.conv2d(ref[0], ref[3], ref[4], padding=1)
    torch.autograd.backward([ya, yb], [ga, gb])
    xd = to_dev_nhwc(x, dev).requires_grad_(True)
    d = [t.to(dev).requires_grad_(True) for t in (wa, ba, wb, bb)]
    oa, ob = ops.dual_head(xd, d[0], d[1], d[2], d[3], pad=1)
    assert oa.is_contiguous() and ob.is_contiguous() and oa.shape == ya.shape and ob.shape == yb.shape
    assert_close(oa.detach().cpu(), ya.detach(), what="dual head a")
    assert_close(ob.detach().cpu(), yb.detach(), what="dual head b")
    torch.autograd.backward([oa, ob], [ga.to(dev), gb.to(dev)])
    assert_close(from_dev_nhwc(xd.grad, Cin), ref[0].grad, what="dual head dx")
    for i, name in enumerate(["wa", "ba", "wb", "bb"]):
        assert_close(d[i].grad.cpu(), ref[i + 1].grad, what=f"dual head d{name}")


ACTS = {"none": lambda t: t, "relu": F.relu, "hardswish": F.hardswish, "hardsigmoid": F.hardsigmoid,
        "sigmoid": torch.sigmoid}


@pytest.mark.parametrize("act", list(ACTS))
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("shape", [(4, 33, 9, 13), (2, 128, 8, 8), (3, 1072, 2, 3)])
def test_bn_act(dev, act, training, shape):
    ops = _ops()
    B, C, H, W = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, C, H, W, generator=g) * 2 + 0.5
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.3
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    use_mul = act == "sigmoid"       # MTAN gate
    use_res = act == "none"          # inverted-residual skip
    mul = torch.randn(B, C, H, W, generator=g) if use_mul else None
    res = torch.randn(B, C, H, W, generator=g) if use_res else None

    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rmr, rvr = rm.clone(), rv.clone()
    mulr = mul.clone().requires_grad_(True) if use_mul else None
    resr = res.clone().requires_grad_(True) if use_res else None
    yr = ACTS[act](F.batch_norm(xr, rmr, rvr, gr, br, training=training, momentum=0.1, eps=1e-5))
    if use_mul:
        yr = mulr * yr
    if use_res:
        yr = yr + resr
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)

    xd = to_dev_nhwc(x, dev).requires_grad_(True)
    gd, bd = gamma.to(dev).requires_grad_(True), beta.to(dev).requires_grad_(True)
    rmd, rvd = rm.to(dev), rv.to(dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    muld = to_dev_nhwc(mul, dev).requires_grad_(True) if use_mul else None
    resd = to_dev_nhwc(res, dev).requires_grad_(True) if use_res else None
    y = ops.bn_act(xd, gd, bd, rmd, rvd, nbt, C, training, 0.1, 1e-5, ops.ACT_CODES[act], mul=muld, res=resd)
    assert_close(from_dev_nhwc(y, C), yr.detach(), what="bn fwd")
    if y.shape[-1] > C and not use_res:
        assert y[..., C:].abs().max().item() == 0.0
    y.backward(to_dev_nhwc(gy, dev))
    assert_close(from_dev_nhwc(xd.grad, C), xr.grad, tol=2e-4, what="bn dx")
    assert_close(gd.grad.cpu(), gr.grad, tol=2e-4, what="bn dgamma")
    assert_close(bd.grad.cpu(), br.grad, tol=2e-4, what="bn dbeta")
    if use_mul:
        assert_close(from_dev_nhwc(muld.grad, C), mulr.grad, what="gate dmul")
    if use_res:
        assert_close(from_dev_nhwc(resd.grad, C), resr.grad, what="residual grad")
    if training:
        assert_close(rmd.cpu(), rmr, what="running_mean")
        assert_close(rvd.cpu(), rvr, what="running_var")
        assert int(nbt.item()) == 1


@pytest.mark.parametrize("fused", [True, False])
def test_bn_large_mean_small_std(dev, fused):
    """|mean| >> std (values ~300, std ~0.05): E[x^2]-E[x]^2 in fp32 would lose the variance entirely
    (x^2 ~ 9e4 has an ulp of 8e-3, the variance is 2e-3); the (mean, M2) partials merged with Chan's
    formula must match two-pass statistics.  Reference = fp64 BatchNorm of the SAME conv output, no
    activation (with a mean stored in fp32 the normalised value carries ~3e-4 sigma of representation
    error, which legitimately flips ReLU masks of values that close to zero).
    fused=True takes the partials from the conv epilogue, fused=False from the stand-alone sweep."""
    ops = _ops()
    g = torch.Generator().manual_seed(41)
    B, C, H, W = 4, 20, 24, 40
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, C, 1, 1, generator=g) * 0.01
    bias = 300.0 + torch.randn(C, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    gy = torch.randn(B, C, H, W, generator=g)
    xd = to_dev_nhwc(x, dev)
    if fused:
        z, stats = ops.conv2d(xd, w.to(dev), bias.to(dev), 1, 0, want_stats=True)
    else:
        z, stats = ops.conv2d(xd, w.to(dev), bias.to(dev), 1, 0), None
    z = z.detach().requires_grad_(True)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    y = ops.bn_act(z, gamma.to(dev), beta.to(dev), rm, rv, nbt, C, True, 0.1, 1e-5, ops.ACT_NONE, stats=stats)
    y.backward(to_dev_nhwc(gy, dev))
    zr = from_dev_nhwc(z.detach(), C).double().requires_grad_(True)
    rmr, rvr = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
    yr = F.batch_norm(zr, rmr, rvr, gamma.double(), beta.double(), training=True, eps=1e-5)
    yr.backward(gy.double())
    assert_close(rm.cpu(), rmr, tol=1e-6, what="running_mean")
    assert_close(rv.cpu(), rvr, tol=1e-4, what="running_var (the variance itself)")
    assert_close(from_dev_nhwc(y, C), yr.detach(), tol=5e-4, what="bn(large mean) fwd")
    assert_close(from_dev_nhwc(z.grad, C), zr.grad, tol=5e-3, what="bn(large mean) dz")


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("shape", [(2, 33, 8, 12), (3, 64, 6, 10), (1, 128, 2, 2)])
def test_bn_act_pool2_fused(dev, shape, training):
    """BatchNorm + ReLU + MaxPool2d(2) as one node (MTAN's encoder attention tail) against the three torch modules: values,
    dx, dgamma, dbeta, running buffers - and against the two separate nodes of this library (VMTL_FUSE_BN_POOL=0 path)."""
    ops = _ops()
    B, C, H, W = shape
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, C, H, W, generator=g) * 1.5 + 0.3
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.randn(C, generator=g) * 0.5 + 1.0)  # some negative scales: the arg-max is taken AFTER the BatchNorm
        bn.bias.copy_(torch.randn(C, generator=g) * 0.2)
        bn.running_mean.copy_(torch.randn(C, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    bn.train(training)
    sd = {k: v.clone() for k, v in bn.state_dict().items()}
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(F.relu(bn(xr)), 2)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)

    def run(fused):
        bd = torch.nn.BatchNorm2d(C)
        bd.load_state_dict(sd)
        bd = bd.to(dev).train(training)
        xd = to_dev_nhwc(x, dev).requires_grad_(True)
        args = (xd, bd.weight, bd.bias, bd.running_mean, bd.running_var, bd.num_batches_tracked, C, training, 0.1, bd.eps,
                ops.ACT_RELU)
        y = ops.bn_act_pool2(*args) if fused else ops.maxpool2(ops.bn_act(*args))
        y.backward(to_dev_nhwc(gy, dev))
        return y, xd.grad, bd
    y, dx, bd = run(True)
    assert_close(from_dev_nhwc(y, C), yr.detach(), what="bn+relu+pool fwd")
    if y.shape[-1] > C:
        assert y[..., C:].abs().max().item() == 0.0 and dx[..., C:].abs().max().item() == 0.0
    assert_close(from_dev_nhwc(dx, C), xr.grad, what="bn+relu+pool dx")
    assert_close(bd.weight.grad.cpu(), bn.weight.grad, what="dgamma")
    assert_close(bd.bias.grad.cpu(), bn.bias.grad, what="dbeta")
    assert_close(bd.running_mean.cpu(), bn.running_mean, tol=1e-5, what="running_mean")
    assert_close(bd.running_var.cpu(), bn.running_var, tol=1e-5, what="running_var")
    y2, dx2, bd2 = run(False)
    assert_close(y.cpu(), y2.cpu(), tol=1e-6, what="fused vs separate nodes")
    assert_close(dx.cpu(), dx2.cpu(), tol=1e-5, what="fused vs separate nodes dx")


def test_plain_activation(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 30, 5, 7, generator=g) * 3
    for act in ["relu", "hardswish", "hardsigmoid"]:
        xr = x.clone().requires_grad_(True)
        yr = ACTS[act](xr)
        gy = torch.randn(yr.shape, generator=g)
        yr.backward(gy)
        xd = to_dev_nhwc(x, dev).requires_grad_(True)
        y = ops.activation(xd, ops.ACT_CODES[act], 30)
        assert_close(from_dev_nhwc(y, 30), yr.detach(), what=act)
        y.backward(to_dev_nhwc(gy, dev))
        assert_close(from_dev_nhwc(xd.grad, 30), xr.grad, what=act + " grad")


def test_concat_up_pad(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    # smp decoder block: cat[nearest2(x), skip]
    a, b = torch.randn(2, 135, 4, 6, generator=g), torch.randn(2, 24, 8, 12, generator=g)
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = torch.cat([F.interpolate(ar, scale_factor=2, mode="nearest"), br], 1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    ad, bd = to_dev_nhwc(a, dev).requires_grad_(True), to_dev_nhwc(b, dev).requires_grad_(True)
    y = ops.concat2(ad, 135, bd, 24, up_a=2)
    assert_close(from_dev_nhwc(y, 159), yr.detach(), what="concat up fwd")
    assert y[..., 159:].abs().max().item() == 0.0
    y.backward(to_dev_nhwc(gy, dev))
    assert_close(from_dev_nhwc(ad.grad, 135), ar.grad, what="concat up grad a")
    assert_close(from_dev_nhwc(bd.grad, 24), br.grad, what="concat grad b")
    # reference utils/model_utils.py:46-58: zero-pad x1 into x2's canvas (odd differences), cat[x2, x1]
    x1, x2 = torch.randn(2, 10, 5, 4, generator=g), torch.randn(2, 7, 8, 9, generator=g)
    x1r, x2r = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
    dY, dX = 3, 5
    yr = torch.cat([x2r, F.pad(x1r, [dX // 2, dX - dX // 2, dY // 2, dY - dY // 2])], 1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    x1d, x2d = to_dev_nhwc(x1, dev).requires_grad_(True), to_dev_nhwc(x2, dev).requires_grad_(True)
    y = ops.concat2(x2d, 7, x1d, 10, out_hw=(8, 9), off_b=(dY // 2, dX // 2))
    assert_close(from_dev_nhwc(y, 17), yr.detach(), what="pad concat fwd")
    y.backward(to_dev_nhwc(gy, dev))
    assert_close(from_dev_nhwc(x1d.grad, 10), x1r.grad, what="pad concat grad x1")
    assert_close(from_dev_nhwc(x2d.grad, 7), x2r.grad, what="pad concat grad x2")


def test_maxpool_bilinear(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(13)
    x = F.relu(torch.randn(2, 20, 8, 12, generator=g))  # many exact ties at 0, as after ReLU
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 2)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = to_dev_nhwc(x, dev).requires_grad_(True)
    y = ops.maxpool2(xd)
    assert torch.equal(from_dev_nhwc(y, 20), yr.detach())
    y.backward(to_dev_nhwc(gy, dev))
    assert torch.equal(from_dev_nhwc(xd.grad, 20), xr.grad)
    for shape in [(2, 12, 5, 7), (1, 128, 16, 16), (2, 4, 1, 3)]:
        x = torch.randn(*shape, generator=g)
        xr = x.clone().requires_grad_(True)
        yr = F.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=True)
        gy = torch.randn(yr.shape, generator=g)
        yr.backward(gy)
        xd = to_dev_nhwc(x, dev).requires_grad_(True)
        y = ops.bilinear_up2(xd)
        assert_close(from_dev_nhwc(y, shape[1]), yr.detach(), tol=1e-5, what="bilinear fwd")
        y.backward(to_dev_nhwc(gy, dev))
        assert_close(from_dev_nhwc(xd.grad, shape[1]), xr.grad, tol=1e-5, what="bilinear bwd")


def test_squeeze_excite_pieces(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(17)
    x = torch.randn(3, 72, 6, 10, generator=g)
    s = torch.rand(3, 72, 1, 1, generator=g)
    xr, sr = x.clone().requires_grad_(True), s.clone().requires_grad_(True)
    yr = xr * sr + xr.mean((2, 3), keepdim=True)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd, sd = to_dev_nhwc(x, dev).requires_grad_(True), to_dev_nhwc(s, dev).requires_grad_(True)
    m = ops.spatial_mean(xd)
    y = ops.channel_scale(xd, sd)
    assert_close(from_dev_nhwc(m, 72), x.mean((2, 3), keepdim=True), what="spatial mean")
    assert_close(from_dev_nhwc(y, 72), (x * s), what="channel scale")
    gyd = to_dev_nhwc(gy, dev)
    torch.autograd.backward([y, m], [gyd, gyd.sum((1, 2), keepdim=True)])
    assert_close(from_dev_nhwc(xd.grad, 72), xr.grad, what="se dx")
    assert_close(from_dev_nhwc(sd.grad, 72), sr.grad, what="se ds")


@pytest.mark.parametrize("case", [(32, 72, 24, 16, 32), (3, 960, 240, 4, 8), (5, 120, 32, 9, 7), (64, 40, 10, 2, 2),
                                  (17, 18, 7, 1, 1)])
def test_squeeze_excite_fused(dev, case):
    """ops.squeeze_excite (partial spatial sums -> two batch-sized GEMMs -> scale) == timm SqueezeExcite in
    torch: x * hardsigmoid(conv_expand(relu(conv_reduce(mean_hw(x))))); values and all five gradients."""
    ops = _ops()
    B, C, R, H, W = case
    g = torch.Generator().manual_seed(29)
    x = torch.randn(B, C, H, W, generator=g)
    wr, br = torch.randn(R, C, 1, 1, generator=g) / C ** 0.5, torch.randn(R, generator=g) * 0.5
    we, be = torch.randn(C, R, 1, 1, generator=g) * (2.0 / R ** 0.5), torch.randn(C, generator=g)
    ref = [t.clone().requires_grad_(True) for t in (x, wr, br, we, be)]
    s = F.conv2d(F.relu(F.conv2d(ref[0].mean((2, 3), keepdim=True), ref[1], ref[2])), ref[3], ref[4])
    yr = ref[0] * F.hardsigmoid(s)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = to_dev_nhwc(x, dev).requires_grad_(True)
    d = [t.to(dev).requires_grad_(True) for t in (wr, br, we, be)]
    y = ops.squeeze_excite(xd, d[0], d[1], d[2], d[3])
    assert_close(from_dev_nhwc(y, C), yr.detach(), what="se fwd")
    if y.shape[-1] > C:
        assert y[..., C:].abs().max().item() == 0.0
    y.backward(to_dev_nhwc(gy, dev))
    assert_close(from_dev_nhwc(xd.grad, C), ref[0].grad, what="se dx")
    for i, name in enumerate(["w_reduce", "b_reduce", "w_expand", "b_expand"]):
        assert_close(d[i].grad.cpu(), ref[i + 1].grad, what=f"se d{name}")


@pytest.mark.parametrize("B", [3, 66])
def test_squeeze_excite_module_both_paths(dev, B):
    """The encoder's SqueezeExcite module: batch <= 64 takes the fused batch-sized-GEMM path, larger batches
    the conv path (spatial mean -> 1x1 convs -> scale); both must equal the timm formula."""
    from vision_mtl_amd import layers as L
    from vision_mtl_amd.models.unet_mobilenetv3 import SqueezeExcite

    ops = _ops()
    assert (B <= ops.squeeze_excite_max_batch()) == (B == 3)
    torch.manual_seed(31)
    se = SqueezeExcite(40, 10)
    x = torch.randn(B, 40, 4, 6)
    xr = x.clone().requires_grad_(True)
    s = F.conv2d(F.relu(F.conv2d(xr.mean((2, 3), keepdim=True), se.conv_reduce.weight, se.conv_reduce.bias)),
                 se.conv_expand.weight, se.conv_expand.bias)
    yr = xr * F.hardsigmoid(s)
    gy = torch.randn(yr.shape)
    yr.backward(gy)
    ref_grads = [p.grad.clone() for p in se.parameters()]
    for p in se.parameters():
        p.grad = None
    se = se.to(dev)
    xd = to_dev_nhwc(x, dev).requires_grad_(True)
    y = se.run(L.Act(xd, 40))
    assert_close(from_dev_nhwc(y.t, 40), yr.detach(), what="SE module fwd")
    y.t.backward(to_dev_nhwc(gy, dev))
    assert_close(from_dev_nhwc(xd.grad, 40), xr.grad, what="SE module dx")
    for p, g in zip(se.parameters(), ref_grads):
        assert_close(p.grad.cpu(), g, what="SE module parameter gradient")


@pytest.mark.parametrize("channel_wise", [True, False])
def test_stitch(dev, channel_wise):
    ops = _ops()
    g = torch.Generator().manual_seed(19)
    T, B, C, H, W = 2, 3, 22, 5, 6
    w = torch.rand(T, T, C, generator=g) if channel_wise else torch.rand(T, T, generator=g)
    x = torch.randn(T, B, C, H, W, generator=g)
    wr, xr = w.clone().requires_grad_(True), x.clone().requires_grad_(True)
    eq = "aac,abcij->abcij" if channel_wise else "aa,abcij->abcij"
    yr = torch.einsum(eq, wr, xr)  # reference models/cross_stitch_model.py:34,36
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    wd = w.to(dev).requires_grad_(True)
    xs = [to_dev_nhwc(x[a], dev).requires_grad_(True) for a in range(T)]
    ys = [ops.stitch(xs[a], wd, a, C) for a in range(T)]
    torch.autograd.backward(ys, [to_dev_nhwc(gy[a], dev) for a in range(T)])
    for a in range(T):
        assert_close(from_dev_nhwc(ys[a], C), yr[a].detach(), what="stitch fwd")
        assert_close(from_dev_nhwc(xs[a].grad, C), xr.grad[a], what="stitch dx")
    assert_close(wd.grad.cpu(), wr.grad, what="stitch dw")
    off = wd.grad.cpu()[0, 1]
    assert float(off.abs().max()) == 0.0  # off-diagonal weights receive exactly zero gradient


@pytest.mark.parametrize("channel_wise", [True, False])
@pytest.mark.parametrize("case", [(2, 40, 9, 11, 24, 1, 0), (2, 37, 8, 12, 20, 3, 1)])
def test_conv_with_folded_stitch_scale(dev, case, channel_wise):
    """ops.conv2d(stitch=(w, task)) == conv(w[task, task, (c)] * x) (reference models/cross_stitch_model.py:32-37 followed
    by the next conv of the CSNet walk): output, dx, the conv's weight gradient and the stitch layer's weight gradient
    (diagonal entry of this task only; every other entry exactly zero) - the scale is folded into the packed operands
    and its gradient comes out of the conv's weight-gradient slabs."""
    ops = _ops()
    B, Cin, H, W, Cout, K, pad = case
    T, task = 2, 1
    g = torch.Generator().manual_seed(321)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    sw = torch.rand((T, T, Cin) if channel_wise else (T, T), generator=g) + 0.25
    xr, wr, swr = x.clone().requires_grad_(True), w.clone().requires_grad_(True), sw.clone().requires_grad_(True)
    scale = swr[task, task].view(1, Cin, 1, 1) if channel_wise else swr[task, task]
    yr = F.conv2d(scale * xr, wr, None, padding=pad)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = to_dev_nhwc(x, dev).requires_grad_(True)
    wd, swd = w.to(dev).requires_grad_(True), sw.to(dev).requires_grad_(True)
    y = ops.conv2d(xd, wd, None, stride=1, pad=pad, stitch=(swd, task))
    assert_close(from_dev_nhwc(y, Cout), yr.detach(), what="stitched conv fwd")
    y.backward(to_dev_nhwc(gy, dev))
    assert_close(from_dev_nhwc(xd.grad, Cin), xr.grad, what="stitched conv dx")
    assert_close(wd.grad.cpu(), wr.grad, what="stitched conv dW")
    assert_close(swd.grad.cpu(), swr.grad, tol=2e-4, what="stitch weight gradient")
    off = swd.grad.cpu().clone()
    off[task, task] = 0
    assert off.abs().max().item() == 0.0  # other tasks' / off-diagonal entries: exactly zero
    # a changed stitch weight reaches the packed operand of the next call (pack-cache dependency on the factor)
    with torch.no_grad():
        swd.mul_(0.5)
    y2 = ops.conv2d(xd.detach(), wd.detach(), None, stride=1, pad=pad, stitch=(swd, task))
    assert_close(y2, 0.5 * y.detach(), tol=1e-5, what="stitched conv after an in-place update of the stitch weights")


@pytest.mark.parametrize("C", [19, 14, 5])
def test_cross_entropy(dev, C):
    ops = _ops()
    g = torch.Generator().manual_seed(23)
    B, H, W = 3, 17, 29
    z = torch.randn(B, C, H, W, generator=g) * 3
    t = torch.randint(0, C, (B, H, W), generator=g)
    zr = z.clone().requires_grad_(True)
    lr = F.cross_entropy(zr, t)
    (lr * 1.7).backward()
    zd = z.to(dev).requires_grad_(True)
    l = ops.cross_entropy(zd, t.to(dev))
    (l * 1.7).backward()
    assert abs(l.item() - lr.item()) <= 1e-5 * abs(lr.item())
    assert_close(zd.grad.cpu(), zr.grad, tol=1e-5, what="CE grad")
    assert torch.equal(ops.argmax_channels(zd).cpu(), z.argmax(1))


def test_cross_entropy_large_logits(dev):
    """Logits around +-100 (a net without BatchNorm produces them): the gradient must stay accurate to
    fp32 rounding of the PROBABILITIES, which needs (z - max) - log(sum), not z - (max + log(sum))."""
    ops = _ops()
    g = torch.Generator().manual_seed(37)
    B, C, H, W = 2, 19, 16, 24
    z = torch.randn(B, C, H, W, generator=g) * 2 + 120.0 * torch.randn(B, 1, H, W, generator=g)
    t = torch.randint(0, C, (B, H, W), generator=g)
    zr = z.double().requires_grad_(True)
    lr = F.cross_entropy(zr, t)
    lr.backward()
    zd = z.to(dev).requires_grad_(True)
    l = ops.cross_entropy(zd, t.to(dev))
    l.backward()
    assert abs(l.item() - lr.item()) <= 2e-6 * abs(lr.item())
    assert_close(zd.grad.cpu(), zr.grad, tol=2e-6, what="CE grad (large logits)")


@pytest.mark.parametrize("masked", [False, True])
def test_silog_l1_sigmoid(dev, masked):
    ops = _ops()
    g = torch.Generator().manual_seed(29)
    B, H, W = 2, 16, 24
    zl = torch.randn(B, 1, H, W, generator=g)
    t = 0.002 + 0.498 * torch.rand(B, H, W, 1, generator=g)
    if masked:
        t[torch.rand(B, H, W, 1, generator=g) < 0.1] = 0.0
    zr = zl.clone().requires_grad_(True)
    pr = torch.sigmoid(zr).permute(0, 2, 3, 1)
    m = t > 1e-3
    gg = torch.log(pr[m]) - torch.log(t[m])
    lr = 10 * torch.sqrt(torch.var(gg) + 0.15 * torch.mean(gg) ** 2)   # reference losses.py:29-36
    lr.backward()
    zd = zl.to(dev).requires_grad_(True)
    pd = ops.sigmoid(zd).permute(0, 2, 3, 1)
    l = ops.silog(pd, t.to(dev))
    l.backward()
    assert abs(l.item() - lr.item()) <= 1e-5 * abs(lr.item())
    assert_close(zd.grad.cpu(), zr.grad, tol=1e-4, what="silog grad")
    zr.grad = None
    zd.grad = None
    l1r = (torch.sigmoid(zr).permute(0, 2, 3, 1) - t).abs().mean()
    l1r.backward()
    l1 = ops.l1_loss(ops.sigmoid(zd).permute(0, 2, 3, 1), t.to(dev))
    l1.backward()
    assert abs(l1.item() - l1r.item()) <= 1e-5 * abs(l1r.item())
    assert_close(zd.grad.cpu(), zr.grad, tol=1e-4, what="l1 grad")


def test_layout_roundtrip_and_adam(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, 3, 8, 12, generator=g)
    xd = x.to(dev).requires_grad_(True)
    y = ops.to_nhwc(xd)
    assert torch.equal(from_dev_nhwc(y, 3), x) and y.shape[-1] == 4
    back = ops.to_nchw(y, 3)
    assert torch.equal(back.cpu(), x)
    back.backward(torch.ones_like(back))
    assert torch.equal(xd.grad.cpu(), torch.ones_like(x))
    # fused Adam == torch.optim.Adam over 3 steps
    p = torch.randn(1000, generator=g)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=5e-3)
    pd, m, v = p.to(dev), torch.zeros(1000, device=dev), torch.zeros(1000, device=dev)
    step = torch.zeros(1, device=dev)
    for i in range(3):
        gr = torch.randn(1000, generator=g)
        pr.grad = gr.clone()
        opt.step()
        step += 1
        ops.adam_step(pd, gr.to(dev), m, v, step, 5e-3)
    assert_close(pd.cpu(), pr.detach(), tol=1e-6, what="adam")


@pytest.mark.parametrize("case", [(2, 16, 6, 9, 7, 10, True), (1, 64, 128, 16, 24, 128, True), (3, 8, 3, 5, 11, 33, False),
                                  (2, 128, 64, 8, 8, 64, True),
                                  # large-M kernel (csrc/conv_pw.hip pw_big_kernel): 64x128 tiles K=192 two-source, data
                                  # gradient 64x96 tiles two-destination; 64x64 tiles K=256 (two column tiles), data
                                  # gradient 64x128 x 2 column tiles; K=64; a ragged last row tile
                                  (1, 64, 128, 256, 256, 128, True), (1, 128, 128, 256, 256, 128, True),
                                  (2, 32, 32, 256, 256, 128, False), (1, 64, 128, 257, 257, 100, True)])
def test_conv1x1_cat_matches_conv_on_concat(dev, case):
    """ops.conv1x1_cat(xa, xb) == conv1x1(cat[xa, xb]) (MTAN attention conv1, reference models/mtan_model.py:57-59,
    139-141): output, BatchNorm partial rows, both input gradients, weight and bias gradients - the concat, the
    gradient split and the second read of the concat by the weight gradient never materialise."""
    ops = _ops()
    B, Ca, Cb, H, W, Cout, bias = case
    g = torch.Generator().manual_seed(123)
    xa, xb = torch.randn(B, Ca, H, W, generator=g), torch.randn(B, Cb, H, W, generator=g)
    w = torch.randn(Cout, Ca + Cb, 1, 1, generator=g) / (Ca + Cb) ** 0.5
    b = torch.randn(Cout, generator=g) if bias else None
    xar, xbr, wr = xa.clone().requires_grad_(True), xb.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    yr = F.conv2d(torch.cat([xar, xbr], 1), wr, br)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xad, xbd = to_dev_nhwc(xa, dev).requires_grad_(True), to_dev_nhwc(xb, dev).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True) if bias else None
    assert ops.conv1x1_cat_supported(xad, Ca, xbd)
    y, stats = ops.conv1x1_cat(xad, xbd, Cb, wd, bd, want_stats=True)
    assert_close(from_dev_nhwc(y, Cout), yr.detach(), what="cat conv fwd")
    if y.shape[-1] > Cout:
        assert y[..., Cout:].abs().max().item() == 0.0
    M, rpb = B * H * W, stats._vmtl_rpb
    st = stats.double().cpu()
    cnt = torch.tensor([max(0, min(rpb, M - i * rpb)) for i in range(st.shape[0])], dtype=torch.float64).view(-1, 1)
    mean = (st[:, 0, :Cout] * cnt).sum(0) / M
    var = ((st[:, 1, :Cout] + cnt * (st[:, 0, :Cout] - mean) ** 2) * (cnt > 0)).sum(0) / M
    yo = yr.detach().double()
    assert_close(mean, yo.mean((0, 2, 3)), tol=1e-5, atol=1e-6, what="cat conv stats mean")
    assert_close(var, yo.var((0, 2, 3), unbiased=False), tol=1e-4, what="cat conv stats var")
    y.backward(to_dev_nhwc(gy, dev))
    assert_close(from_dev_nhwc(xad.grad, Ca), xar.grad, what="cat conv dxa")
    assert_close(from_dev_nhwc(xbd.grad, Cb), xbr.grad, what="cat conv dxb")
    if xbd.grad.shape[-1] > Cb:
        assert xbd.grad[..., Cb:].abs().max().item() == 0.0
    assert_close(wd.grad.cpu(), wr.grad, what="cat conv dw")
    if bias:
        assert_close(bd.grad.cpu(), br.grad, what="cat conv db")
