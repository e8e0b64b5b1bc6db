"""-m gpu: the halo-tile 3x3 kernel for the narrow full-resolution layers (csrc/conv_small.hip) and the fused
decoder tail built on it (ops.decoder_tail), against plain PyTorch fp32 on the CPU.  The reference composition is
smp DecoderBlock.conv2 (Conv-BN-ReLU) + the two 3x3 heads of reference vision_mtl/models/basic_model.py:30-51.
Tolerance 1e-4 of the reference's max magnitude (BASELINE.json north_star)."""
import pytest
import torch
import torch.nn.functional as F

from tests.util import assert_close, ceil4, from_dev_nhwc, to_dev_nhwc

pytestmark = pytest.mark.gpu


def _small(ops, x, wp, y, Cs, ldy, Nw, Cout, **kw):
    B, H, W, _ = x.shape
    ops._small(x, wp, y, B, H, W, Cs, ldy, Nw, Cout, 0.0, **kw)


def _pack_fwd(ops, w, dev):
    Cout, Cin = w.shape[:2]
    return ops.pack(w.to(dev), 1, Cout, 9, Cin, ceil4(Cin), 0, Cin * 9, 1, 9)


# B, Cin, Cout, H, W
SMALL_CASES = [(2, 33, 33, 8, 64), (1, 33, 20, 12, 32), (2, 20, 33, 4, 96), (1, 32, 32, 8, 32), (2, 16, 14, 8, 32),
               (1, 33, 33, 7, 45)]  # the last one: partial tiles (plain mode only)


@pytest.mark.parametrize("case", SMALL_CASES)
def test_conv3x3_small_plain_and_prologue(dev, case):
    from vision_mtl_amd import ops

    B, Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    bias = torch.randn(Cout, generator=g)
    Cs, ldy = ceil4(Cin), ceil4(Cout)
    xd, wp = to_dev_nhwc(x, dev), _pack_fwd(ops, w, dev)
    # plain + bias
    y = torch.full((B, H, W, ldy), float("nan"), device=dev)
    _small(ops, xd, wp, y, Cs, ldy, Cout, Cout, bias=bias.to(dev))
    assert_close(from_dev_nhwc(y, Cout), F.conv2d(x, w, bias, padding=1), what="small conv plain")
    assert float(y[..., Cout:].abs().sum()) == 0.0, "pad channels must be zero"
    # prologue relu(a*x + c), transformed input written back
    pa, pc = torch.randn(Cin, generator=g), torch.randn(Cin, generator=g)
    pad = lambda v: torch.cat([v, torch.zeros(Cs - Cin)]).to(dev)
    a_out = torch.full_like(xd, float("nan"))
    _small(ops, xd, wp, y, Cs, ldy, Cout, Cout, pa=pad(pa), pc=pad(pc), act_in=ops.ACT_RELU, a_out=a_out)
    a_ref = F.relu(x * pa.view(1, -1, 1, 1) + pc.view(1, -1, 1, 1))
    assert_close(from_dev_nhwc(a_out, Cin), a_ref, what="small conv a_out")
    assert float(a_out[..., Cin:].abs().sum()) == 0.0
    assert_close(from_dev_nhwc(y, Cout), F.conv2d(a_ref, w, None, padding=1), what="small conv prologue")
    # two-operand affine prologue, no activation
    x2 = torch.randn(B, Cin, H, W, generator=g)
    pb = torch.randn(Cin, generator=g)
    _small(ops, xd, wp, y, Cs, ldy, Cout, Cout, x2=to_dev_nhwc(x2, dev), pa=pad(pa), pb=pad(pb), pc=pad(pc), a_out=a_out)
    v = lambda t: t.view(1, -1, 1, 1)
    a_ref = x * v(pa) + x2 * v(pb) + v(pc)
    assert_close(from_dev_nhwc(a_out, Cin), a_ref, what="small conv 2-operand a_out")
    assert_close(from_dev_nhwc(y, Cout), F.conv2d(a_ref, w, None, padding=1), what="small conv 2-operand prologue")


@pytest.mark.parametrize("case", SMALL_CASES[:5])
def test_conv3x3_small_epilogues(dev, case):
    from vision_mtl_amd import ops
    from vision_mtl_amd._lib import lib

    B, Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(8)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    Cs, ldy = ceil4(Cin), ceil4(Cout)
    xd, wp = to_dev_nhwc(x, dev), _pack_fwd(ops, w, dev)
    yr = F.conv2d(x, w, None, padding=1)
    tiles = lib().raw("vmtl_conv3x3_small_tiles")(B, H, W)
    assert tiles == B * (H // 4) * (W // 32)
    # few tiles: one statistics row per tile (= per workgroup), 128 pixels each
    assert lib().raw("vmtl_conv3x3_small_stat_rows")(B, H, W) == tiles
    assert lib().raw("vmtl_conv3x3_small_stat_block")(B, H, W) == 128

    def per_tile(t):  # (B,C,H,W) -> (tiles, C, 128) in the kernel's tile order (image, tile row, tile column)
        return t.view(B, -1, H // 4, 4, W // 32, 32).permute(0, 2, 4, 1, 3, 5).reshape(tiles, t.shape[1], 128)

    # mode 1: per-tile (mean, M2)
    y = torch.empty((B, H, W, ldy), device=dev)
    stats = torch.full((tiles, 2, ldy), float("nan"), device=dev)
    _small(ops, xd, wp, y, Cs, ldy, Cout, Cout, stats=stats, ep_mode=1)
    assert_close(from_dev_nhwc(y, Cout), yr, what="mode 1 values")
    pt = per_tile(yr).double()
    assert_close(stats[:, 0, :Cout].cpu(), pt.mean(-1), tol=1e-5, atol=1e-6, what="tile mean")
    assert_close(stats[:, 1, :Cout].cpu(), ((pt - pt.mean(-1, keepdim=True)) ** 2).sum(-1), tol=1e-4, what="tile M2")
    assert float(stats[:, :, Cout:].abs().sum()) == 0.0
    # mode 2: dz = conv * relu'(gamma * xhat + beta), per-tile (sum dz, sum dz*xhat)
    xz = torch.randn(B, Cout, H, W, generator=g)
    mean, invstd = torch.randn(Cout, generator=g) * 0.1, torch.rand(Cout, generator=g) + 0.5
    gamma, beta = torch.randn(Cout, generator=g), torch.randn(Cout, generator=g) * 0.3
    v = lambda t: t.view(1, -1, 1, 1)
    xhat = (xz - v(mean)) * v(invstd)
    dz_ref = yr * ((v(gamma) * xhat + v(beta)) > 0).float()
    pad = lambda t: torch.cat([t, torch.zeros(ldy - Cout)]).to(dev)
    _small(ops, xd, wp, y, Cs, ldy, Cout, Cout, stats=stats, ep_mode=2,
           ez=(to_dev_nhwc(xz, dev), pad(mean), pad(invstd), pad(gamma), pad(beta), ops.ACT_RELU))
    assert_close(from_dev_nhwc(y, Cout), dz_ref, what="mode 2 dz")
    assert float(y[..., Cout:].abs().sum()) == 0.0
    scale = float(per_tile(dz_ref.abs()).sum(-1).max())
    assert_close(stats[:, 0, :Cout].cpu(), per_tile(dz_ref).double().sum(-1), tol=1e-5, atol=1e-5 * scale, what="sum dz")
    assert_close(stats[:, 1, :Cout].cpu(), per_tile(dz_ref * xhat).double().sum(-1), tol=1e-5, atol=1e-5 * scale,
                 what="sum dz*xhat")


def test_conv3x3_small_statistics_accumulated_per_workgroup(dev):
    """More tiles than workgroup slots: every persistent workgroup folds its tiles into ONE statistics row
    ((mean, M2) merged with Chan's formula, plain sums added); merged over rows they must give the tensor's statistics."""
    from vision_mtl_amd import ops
    from vision_mtl_amd._lib import lib

    B, Cin, Cout, H, W = 8, 33, 33, 64, 256
    g = torch.Generator().manual_seed(12)
    x = torch.randn(B, Cin, H, W, generator=g) + 0.5
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    Cs, ldy = ceil4(Cin), ceil4(Cout)
    rows, blk = lib().raw("vmtl_conv3x3_small_stat_rows")(B, H, W), lib().raw("vmtl_conv3x3_small_stat_block")(B, H, W)
    tiles = lib().raw("vmtl_conv3x3_small_tiles")(B, H, W)
    assert rows < tiles and rows * blk == B * H * W, (rows, blk, tiles)
    xd, wp = to_dev_nhwc(x, dev), _pack_fwd(ops, w, dev)
    yr = F.conv2d(x, w, None, padding=1).double()
    y = torch.empty((B, H, W, ldy), device=dev)
    stats = torch.full((rows, 2, ldy), float("nan"), device=dev)
    _small(ops, xd, wp, y, Cs, ldy, Cout, Cout, stats=stats, ep_mode=1)
    st = stats.cpu().double()
    mean = st[:, 0, :Cout].mean(0)  # equal-sized rows
    m2 = st[:, 1, :Cout].sum(0) + blk * ((st[:, 0, :Cout] - mean) ** 2).sum(0)
    assert_close(mean, yr.mean((0, 2, 3)), tol=1e-5, atol=1e-6, what="merged mean")
    assert_close(m2 / (B * H * W), yr.var((0, 2, 3), unbiased=False), tol=1e-4, what="merged variance")
    xz = torch.randn(B, Cout, H, W, generator=g)
    vec = [torch.rand(Cout, generator=g) + 0.5 for _ in range(4)]
    v = lambda t: t.view(1, -1, 1, 1)
    xhat = (xz - v(vec[0])) * v(vec[1])
    dz = yr.float() * ((v(vec[2]) * xhat + v(vec[3])) > 0).float()
    pad = lambda t: torch.cat([t, torch.zeros(ldy - Cout)]).to(dev)
    _small(ops, xd, wp, y, Cs, ldy, Cout, Cout, stats=stats, ep_mode=2,
           ez=(to_dev_nhwc(xz, dev), pad(vec[0]), pad(vec[1]), pad(vec[2]), pad(vec[3]), ops.ACT_RELU))
    st = stats.cpu().double()
    scale = float(dz.abs().double().sum((0, 2, 3)).max())
    assert_close(st[:, 0, :Cout].sum(0), dz.double().sum((0, 2, 3)), tol=1e-5, atol=1e-6 * scale, what="total sum dz")
    assert_close(st[:, 1, :Cout].sum(0), (dz * xhat).double().sum((0, 2, 3)), tol=1e-5, atol=1e-6 * scale, what="total sum dz*xhat")


def test_conv3x3_small_nchw_split(dev):
    from vision_mtl_amd import ops

    g = torch.Generator().manual_seed(9)
    B, Cin, Ca, Cb, H, W = 2, 33, 19, 1, 8, 64
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Ca + Cb, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    bias = torch.randn(Ca + Cb, generator=g)
    oa, ob = torch.empty((B, Ca, H, W), device=dev), torch.empty((B, Cb, H, W), device=dev)
    _small(ops, to_dev_nhwc(x, dev), _pack_fwd(ops, w, dev), oa, ceil4(Cin), ceil4(Ca + Cb), Ca + Cb, Ca + Cb,
           bias=bias.to(dev), yb=ob, Ca=Ca)
    yr = F.conv2d(x, w, bias, padding=1)
    assert_close(oa.cpu(), yr[:, :Ca], what="split store head a")
    assert_close(ob.cpu(), yr[:, Ca:], what="split store head b")


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("shape", [(2, 33, 33, 19, 8, 64), (1, 16, 16, 14, 12, 32)])
def test_decoder_tail_matches_torch(dev, training, shape):
    """heads(relu(bn2(conv2(relu(bn1(x1)))))) and every gradient == the torch composition, in train mode (batch
    statistics from the conv epilogues, running buffers updated) and in eval mode."""
    from vision_mtl_amd import ops

    B, C1, C2, Ca, H, W = shape
    Cb = 1
    g = torch.Generator().manual_seed(10)
    x1 = torch.randn(B, C1, H, W, generator=g) * 2 + 0.5
    bn1, bn2 = torch.nn.BatchNorm2d(C1), torch.nn.BatchNorm2d(C2)
    for bn in (bn1, bn2):
        bn.weight.data = torch.rand(bn.num_features, generator=g) + 0.5
        bn.bias.data = torch.randn(bn.num_features, generator=g) * 0.2
        bn.running_mean.data = torch.randn(bn.num_features, generator=g) * 0.1
        bn.running_var.data = torch.rand(bn.num_features, generator=g) + 0.5
        bn.train(training)
    w2 = torch.randn(C2, C1, 3, 3, generator=g) / (C1 * 9) ** 0.5
    wa, wb = torch.randn(Ca, C2, 3, 3, generator=g) * 0.1, torch.randn(Cb, C2, 3, 3, generator=g) * 0.1
    ba, bb = torch.randn(Ca, generator=g), torch.randn(Cb, generator=g)
    ga, gb = torch.randn(B, Ca, H, W, generator=g), torch.randn(B, Cb, H, W, generator=g)
    import copy

    bn1d, bn2d = copy.deepcopy(bn1).to(dev), copy.deepcopy(bn2).to(dev)
    ref = [t.clone().requires_grad_(True) for t in (x1, w2, wa, ba, wb, bb)]
    a2 = F.relu(bn2(F.conv2d(F.relu(bn1(ref[0])), ref[1], None, padding=1)))
    ya, yb = F.conv2d(a2, ref[2], ref[3], padding=1), F.conv2d(a2, ref[4], ref[5], padding=1)
    torch.autograd.backward([ya, yb], [ga, gb])

    xd = to_dev_nhwc(x1, dev).requires_grad_(True)
    d = [t.to(dev).requires_grad_(True) for t in (w2, wa, ba, wb, bb)]
    assert ops.decoder_tail_supported(xd.shape, C1, C2, Ca + Cb)
    oa, ob = ops.decoder_tail(xd, None, 0, bn1d, d[0], bn2d, d[1], d[2], d[3], d[4])
    assert oa.is_contiguous() and ob.is_contiguous()
    assert_close(oa.detach().cpu(), ya.detach(), what="tail head a")
    assert_close(ob.detach().cpu(), yb.detach(), what="tail head b")
    torch.autograd.backward([oa, ob], [ga.to(dev), gb.to(dev)])
    assert_close(from_dev_nhwc(xd.grad, C1), ref[0].grad, tol=2e-4, what="tail dx1")
    for i, name in enumerate(["w2", "wa", "ba", "wb", "bb"]):
        assert_close(d[i].grad.cpu(), ref[i + 1].grad, tol=2e-4, what=f"tail d{name}")
    for nm, a, b in (("bn1", bn1d, bn1), ("bn2", bn2d, bn2)):
        assert_close(a.weight.grad.cpu(), b.weight.grad, tol=2e-4, what=f"tail d{nm}.weight")
        assert_close(a.bias.grad.cpu(), b.bias.grad, tol=2e-4, what=f"tail d{nm}.bias")
        assert_close(a.running_mean.cpu(), b.running_mean, tol=1e-5, what=f"{nm}.running_mean")
        assert_close(a.running_var.cpu(), b.running_var, tol=1e-5, what=f"{nm}.running_var")
        assert int(a.num_batches_tracked) == int(b.num_batches_tracked)


# B, C (channels of x), Cout, C1 (skip channels, up2 only), H, W, up2
BNCONV_CASES = [(2, 33, 33, 0, 16, 24, False), (2, 67, 33, 0, 12, 20, True), (1, 135, 67, 16, 8, 12, True),
                (2, 20, 67, 24, 9, 7, True), (1, 64, 128, 0, 8, 8, False)]  # last: split-K data gradient -> unfused fallback


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("case", BNCONV_CASES)
def test_bn_act_conv_matches_torch(dev, case, training):
    """ops.bn_act_conv (pre-activation node: BatchNorm + ReLU + the consuming conv, with the BatchNorm-backward
    reduction in the data gradient's epilogue) == the torch composition, values and every gradient."""
    import copy

    from vision_mtl_amd import ops

    B, C, Cout, C1, H, W, up2 = case
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, C, H, W, generator=g) * 1.5 + 0.3
    bn = torch.nn.BatchNorm2d(C)
    bn.weight.data = torch.rand(C, generator=g) + 0.5
    bn.bias.data = torch.randn(C, generator=g) * 0.2
    bn.running_mean.data = torch.randn(C, generator=g) * 0.1
    bn.running_var.data = torch.rand(C, generator=g) + 0.5
    bn.train(training)
    bnd = copy.deepcopy(bn).to(dev)
    w = torch.randn(Cout, C + C1, 3, 3, generator=g) / ((C + C1) * 9) ** 0.5
    skip = torch.randn(B, C1, 2 * H, 2 * W, generator=g) if C1 else None
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    skr = skip.clone().requires_grad_(True) if C1 else None
    a = F.relu(bn(xr))
    if up2:
        a = F.interpolate(a, scale_factor=2, mode="nearest")
        a = torch.cat([a, skr], 1) if C1 else a
    yr = F.conv2d(a, wr, None, padding=1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)

    xd = to_dev_nhwc(x, dev).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    skd = to_dev_nhwc(skip, dev).requires_grad_(True) if C1 else None
    y, stats, rpb = ops.bn_act_conv(xd, None, 0, bnd, C, ops.ACT_RELU, wd, skip=skd, up2=up2, want_stats=True)
    assert_close(from_dev_nhwc(y, Cout), yr.detach(), what="bn_act_conv fwd")
    if stats is not None:  # the raw output's own BatchNorm partial rows (for the NEXT node)
        st = stats.cpu().double()
        M = yr.shape[0] * yr.shape[2] * yr.shape[3]
        nrows = [min(rpb, M - i * rpb) for i in range(st.shape[0])] if not up2 else [rpb] * st.shape[0]
        cnt = torch.tensor(nrows, dtype=torch.float64).view(-1, 1)
        mean = (st[:, 0, :Cout] * cnt).sum(0) / M
        assert_close(mean, yr.detach().double().mean((0, 2, 3)), tol=1e-5, atol=1e-6, what="output stats mean")
    y.backward(to_dev_nhwc(gy, dev))
    assert_close(from_dev_nhwc(xd.grad, C), xr.grad, tol=2e-4, what="bn_act_conv dx")
    assert_close(wd.grad.cpu(), wr.grad, tol=2e-4, what="bn_act_conv dw")
    assert_close(bnd.weight.grad.cpu(), bn.weight.grad, tol=2e-4, what="bn_act_conv dgamma")
    assert_close(bnd.bias.grad.cpu(), bn.bias.grad, tol=2e-4, what="bn_act_conv dbeta")
    if C1:
        assert_close(from_dev_nhwc(skd.grad, C1), skr.grad, tol=2e-4, what="bn_act_conv dskip")
    assert_close(bnd.running_mean.cpu(), bn.running_mean, tol=1e-5, what="running_mean")
    assert_close(bnd.running_var.cpu(), bn.running_var, tol=1e-5, what="running_var")


# B, C, H, W, K, stride, act
BNDW_CASES = [(2, 64, 16, 24, 3, 2, "relu"), (2, 72, 12, 20, 5, 2, "relu"), (1, 120, 9, 7, 5, 1, "relu"),
              (2, 200, 8, 8, 3, 1, "hardswish"), (1, 960, 4, 8, 5, 1, "hardswish"), (3, 24, 15, 17, 3, 2, "relu")]


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("case", BNDW_CASES)
def test_bn_act_dwconv_matches_torch(dev, case, training):
    """ops.bn_act_dwconv (BatchNorm + activation applied while the depthwise taps load, output statistics from the
    same pass) == depthwise_conv(act(bn(x))) in torch: values, output statistics, every gradient."""
    import copy

    from vision_mtl_amd import ops

    B, C, H, W, K, stride, act = case
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, C, H, W, generator=g) * 1.5 + 0.3
    bn = torch.nn.BatchNorm2d(C)
    bn.weight.data = torch.rand(C, generator=g) + 0.5
    bn.bias.data = torch.randn(C, generator=g) * 0.2
    bn.running_mean.data = torch.randn(C, generator=g) * 0.1
    bn.running_var.data = torch.rand(C, generator=g) + 0.5
    bn.train(training)
    bnd = copy.deepcopy(bn).to(dev)
    w = torch.randn(C, 1, K, K, generator=g) / K
    fact = {"relu": F.relu, "hardswish": F.hardswish}[act]
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(fact(bn(xr)), wr, None, stride=stride, padding=(K - 1) // 2, groups=C)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = to_dev_nhwc(x, dev).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    y, stats, rpb = ops.bn_act_dwconv(xd, None, 0, bnd, C, ops.ACT_CODES[act], wd, stride, (K - 1) // 2, want_stats=True)
    assert_close(from_dev_nhwc(y, C), yr.detach(), what="bn_act_dwconv fwd")
    assert float(y[..., C:].abs().sum()) == 0.0 if y.shape[-1] > C else True
    st = stats.cpu().double()
    M = yr.shape[0] * yr.shape[2] * yr.shape[3]
    cnt = torch.tensor([max(0, min(rpb, M - i * rpb)) for i in range(st.shape[0])], dtype=torch.float64).view(-1, 1)
    mean = (st[:, 0, :C] * cnt).sum(0) / M
    var = ((st[:, 1, :C] + cnt * (st[:, 0, :C] - mean) ** 2) * (cnt > 0)).sum(0) / M
    yo = yr.detach().double()
    assert_close(mean, yo.mean((0, 2, 3)), tol=1e-5, atol=1e-6, what="dw output stats mean")
    assert_close(var, yo.var((0, 2, 3), unbiased=False), tol=1e-4, what="dw output stats var")
    y.backward(to_dev_nhwc(gy, dev))
    assert_close(from_dev_nhwc(xd.grad, C), xr.grad, tol=2e-4, what="bn_act_dwconv dx")
    assert_close(wd.grad.cpu(), wr.grad, tol=2e-4, what="bn_act_dwconv dw")
    assert_close(bnd.weight.grad.cpu(), bn.weight.grad, tol=2e-4, what="bn_act_dwconv dgamma")
    assert_close(bnd.bias.grad.cpu(), bn.bias.grad, tol=2e-4, what="bn_act_dwconv dbeta")
    assert_close(bnd.running_mean.cpu(), bn.running_mean, tol=1e-5, what="running_mean")
    assert_close(bnd.running_var.cpu(), bn.running_var, tol=1e-5, what="running_var")


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(2, 16, 12, 20, 3, 1, "hardswish"), (1, 72, 8, 12, 5, 2, "relu"), (2, 24, 9, 7, 3, 1, "relu")])
def test_bn_act_dwconv_returns_activation_for_a_second_consumer(dev, case):
    """return_act=True: a = act(bn(x)) comes back as a differentiable output (the residual branch of the block) and the
    gradient that reaches it is added inside the depthwise data gradient (vmtl_dwconv_bwd_data_add):
    loss = <y, gy> + <a, ga> must give torch's gradients for x, the depthwise weight and the BatchNorm parameters."""
    import copy

    from vision_mtl_amd import ops

    B, C, H, W, K, stride, act = case
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, C, H, W, generator=g) * 1.2 - 0.2
    bn = torch.nn.BatchNorm2d(C)
    bn.weight.data = torch.rand(C, generator=g) + 0.5
    bn.bias.data = torch.randn(C, generator=g) * 0.2
    bnd = copy.deepcopy(bn).to(dev)
    w = torch.randn(C, 1, K, K, generator=g) / K
    fact = {"relu": F.relu, "hardswish": F.hardswish}[act]
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ar = fact(bn(xr))
    yr = F.conv2d(ar, wr, None, stride=stride, padding=(K - 1) // 2, groups=C)
    gy, ga = torch.randn(yr.shape, generator=g), torch.randn(ar.shape, generator=g)
    ((yr * gy).sum() + (ar * ga).sum()).backward()
    xd = to_dev_nhwc(x, dev).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    y, stats, rpb, a = ops.bn_act_dwconv(xd, None, 0, bnd, C, ops.ACT_CODES[act], wd, stride, (K - 1) // 2,
                                         want_stats=True, return_act=True)
    assert_close(from_dev_nhwc(a, C), ar.detach(), what="returned activation")
    assert_close(from_dev_nhwc(y, C), yr.detach(), what="fwd")
    ((y * to_dev_nhwc(gy, dev)).sum() + (a * to_dev_nhwc(ga, dev)).sum()).backward()
    assert_close(from_dev_nhwc(xd.grad, C), xr.grad, tol=2e-4, what="dx with the second consumer's gradient")
    assert_close(wd.grad.cpu(), wr.grad, tol=2e-4, what="dw")
    assert_close(bnd.weight.grad.cpu(), bn.weight.grad, tol=2e-4, what="dgamma")
    assert_close(bnd.bias.grad.cpu(), bn.bias.grad, tol=2e-4, what="dbeta")


# B, C, H, W, Cout, act, bias
BNPW_CASES = [(2, 16, 12, 20, 24, "relu", True), (1, 72, 8, 8, 24, "relu", False), (2, 128, 16, 16, 64, "relu", True),
              (1, 240, 4, 8, 40, "hardswish", False), (2, 64, 9, 7, 130, "relu", True),
              # large-M kernel with the staging prologue: 64x128, 256x32 and 64x64 tiles
              (1, 128, 256, 256, 128, "relu", True), (4, 128, 256, 256, 32, "relu", True), (2, 128, 256, 256, 64, "relu", False)]


@pytest.mark.gpu
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("case", BNPW_CASES)
def test_bn_act_conv1x1_matches_torch(dev, case, training):
    """ops.bn_act_conv1x1 (BatchNorm + activation on the pointwise GEMM's operand fragments, BatchNorm-backward
    reduction in the data gradient's epilogue) == conv1x1(act(bn(x))) in torch: values, output statistics, every
    gradient, running buffers; K-split and plain tilings, 32- and 64-column tiles."""
    import copy

    from vision_mtl_amd import ops

    B, C, H, W, Cout, act, bias = case
    g = torch.Generator().manual_seed(55)
    x = torch.randn(B, C, H, W, generator=g) * 1.3 + 0.2
    bn = torch.nn.BatchNorm2d(C)
    bn.weight.data = torch.rand(C, generator=g) + 0.5
    bn.bias.data = torch.randn(C, generator=g) * 0.2
    bn.running_mean.data = torch.randn(C, generator=g) * 0.1
    bn.running_var.data = torch.rand(C, generator=g) + 0.5
    bn.train(training)
    bnd = copy.deepcopy(bn).to(dev)
    w = torch.randn(Cout, C, 1, 1, generator=g) / C ** 0.5
    b = torch.randn(Cout, generator=g) if bias else None
    fact = {"relu": F.relu, "hardswish": F.hardswish}[act]
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    yr = F.conv2d(fact(bn(xr)), wr, br)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = to_dev_nhwc(x, dev).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True) if bias else None
    assert ops.bn_act_conv1x1_supported(xd, ops.ACT_CODES[act])
    y, stats, rpb = ops.bn_act_conv1x1(xd, None, 0, bnd, C, ops.ACT_CODES[act], wd, bd, want_stats=True)
    assert_close(from_dev_nhwc(y, Cout), yr.detach(), what="bn_act_conv1x1 fwd")
    if y.shape[-1] > Cout:
        assert y[..., Cout:].abs().max().item() == 0.0
    M = B * H * W
    st = stats.cpu().double()
    cnt = torch.tensor([max(0, min(rpb, M - i * rpb)) for i in range(st.shape[0])], dtype=torch.float64).view(-1, 1)
    mean = (st[:, 0, :Cout] * cnt).sum(0) / M
    var = ((st[:, 1, :Cout] + cnt * (st[:, 0, :Cout] - mean) ** 2) * (cnt > 0)).sum(0) / M
    yo = yr.detach().double()
    assert_close(mean, yo.mean((0, 2, 3)), tol=1e-5, atol=1e-6, what="pw output stats mean")
    assert_close(var, yo.var((0, 2, 3), unbiased=False), tol=1e-4, what="pw output stats var")
    y.backward(to_dev_nhwc(gy, dev))
    assert_close(from_dev_nhwc(xd.grad, C), xr.grad, tol=2e-4, what="bn_act_conv1x1 dx")
    assert_close(wd.grad.cpu(), wr.grad, tol=2e-4, what="bn_act_conv1x1 dw")
    assert_close(bnd.weight.grad.cpu(), bn.weight.grad, tol=2e-4, what="bn_act_conv1x1 dgamma")
    assert_close(bnd.bias.grad.cpu(), bn.bias.grad, tol=2e-4, what="bn_act_conv1x1 dbeta")
    if bias:
        assert_close(bd.grad.cpu(), br.grad, tol=2e-4, what="bn_act_conv1x1 dbias")
    assert_close(bnd.running_mean.cpu(), bn.running_mean, tol=1e-5, what="running_mean")
    assert_close(bnd.running_var.cpu(), bn.running_var, tol=1e-5, what="running_var")


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(2, 24, 12, 20, 72, True), (1, 160, 4, 8, 960, True), (2, 40, 9, 7, 120, False)])
def test_bn_act_conv1x1_with_residual_and_returned_activation(dev, case):
    """The inverted-residual hand-over: a = bn3(x) + res feeds the next block's expand conv (on its operand fragments),
    comes back as a differentiable output for its other consumers, and their gradient is added inside the data gradient
    before the BatchNorm backward: loss = <y, gy> + <a, ga> must give torch's gradients for x, res, BN and the weight."""
    import copy

    from vision_mtl_amd import ops

    B, C, H, W, Cout, use_res = case
    g = torch.Generator().manual_seed(91)
    x = torch.randn(B, C, H, W, generator=g) * 1.1 + 0.1
    r = torch.randn(B, C, H, W, generator=g)
    bn = torch.nn.BatchNorm2d(C)
    bn.weight.data = torch.rand(C, generator=g) + 0.5
    bn.bias.data = torch.randn(C, generator=g) * 0.2
    bnd = copy.deepcopy(bn).to(dev)
    w = torch.randn(Cout, C, 1, 1, generator=g) / C ** 0.5
    xr, rr, wr = x.clone().requires_grad_(True), r.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ar = bn(xr) + rr if use_res else bn(xr)
    yr = F.conv2d(ar, wr)
    gy, ga = torch.randn(yr.shape, generator=g), torch.randn(ar.shape, generator=g)
    ((yr * gy).sum() + (ar * ga).sum()).backward()
    xd, rd = to_dev_nhwc(x, dev).requires_grad_(True), to_dev_nhwc(r, dev).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    y, stats, rpb, a = ops.bn_act_conv1x1(xd, None, 0, bnd, C, ops.ACT_NONE, wd, None, want_stats=True,
                                          res=rd if use_res else None, return_act=True)
    assert_close(from_dev_nhwc(a, C), ar.detach(), what="returned activation")
    assert_close(from_dev_nhwc(y, Cout), yr.detach(), what="fwd")
    ((y * to_dev_nhwc(gy, dev)).sum() + (a * to_dev_nhwc(ga, dev)).sum()).backward()
    assert_close(from_dev_nhwc(xd.grad, C), xr.grad, tol=2e-4, what="dx")
    if use_res:
        assert_close(from_dev_nhwc(rd.grad, C), rr.grad, tol=2e-4, what="dres")
    assert_close(wd.grad.cpu(), wr.grad, tol=2e-4, what="dw")
    assert_close(bnd.weight.grad.cpu(), bn.weight.grad, tol=2e-4, what="dgamma")
    assert_close(bnd.bias.grad.cpu(), bn.bias.grad, tol=2e-4, what="dbeta")
