"""Diagnostic (GPU box): one CSNet decoder-stage entry in isolation, HIP vs CPU fp64."""
import sys

import torch
import torch.nn.functional as F

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import from_dev_nhwc, to_dev_nhwc
from vision_mtl_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


for task in (0, 1):
    for (Cx, Hx, Cs_, Hs, Cout) in [(128, 16, 24, 32, 64), (256, 8, 40, 16, 128), (960, 4, 112, 8, 256)]:
        B = 2
        x1 = torch.randn(B, Cx, Hx, Hx, generator=g)
        sk = torch.randn(B, Cs_, Hs, Hs, generator=g)
        Ct = Cx + Cs_
        wst = torch.rand(2, 2, Ct, generator=g)
        wc = torch.randn(Cout, Ct, 3, 3, generator=g) / (Ct * 9) ** 0.5
        gamma, beta = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
        gy = torch.randn(B, Cout, Hs, Hs, generator=g)
        # CPU fp64
        x1r, skr, wstr, wcr = (t.double().requires_grad_(True) for t in (x1, sk, wst, wc))
        gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
        d = Hs - Hx
        canvas = torch.cat([skr, F.pad(x1r, [d // 2, d - d // 2, d // 2, d - d // 2])], 1)
        st = canvas * wstr[task, task][None, :, None, None]
        z = F.conv2d(st, wcr, None, padding=1)
        y = F.relu(F.batch_norm(z, None, None, gr, br, training=True, eps=1e-5))
        y.backward(gy.double())
        # HIP
        x1d, skd = to_dev_nhwc(x1, dev).requires_grad_(True), to_dev_nhwc(sk, dev).requires_grad_(True)
        wstd, wcd = wst.to(dev).requires_grad_(True), wc.to(dev).requires_grad_(True)
        gd, bd = gamma.to(dev).requires_grad_(True), beta.to(dev).requires_grad_(True)
        cv = ops.concat2(skd, Cs_, x1d, Cx, out_hw=(Hs, Hs), off_b=(d // 2, d // 2))
        sd = ops.stitch(cv, wstd, task, Ct)
        zz, stats = ops.conv2d(sd, wcd, None, 1, 1, want_stats=True)
        nbt = torch.zeros((), dtype=torch.int64, device=dev)
        yy = ops.bn_act(zz, gd, bd, torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev), nbt, Cout, True, 0.1,
                        1e-5, ops.ACT_RELU, stats=stats)
        yy.backward(to_dev_nhwc(gy, dev))
        print(f"task {task} Cx {Cx}: y {rel(from_dev_nhwc(yy, Cout), y.detach()):.1e} dx1 "
              f"{rel(from_dev_nhwc(x1d.grad, Cx), x1r.grad):.1e} dskip {rel(from_dev_nhwc(skd.grad, Cs_), skr.grad):.1e} "
              f"dwc {rel(wcd.grad.cpu(), wcr.grad):.1e} dwst {rel(wstd.grad.cpu(), wstr.grad):.1e} "
              f"dgamma {rel(gd.grad.cpu(), gr.grad):.1e} dbeta {rel(bd.grad.cpu(), br.grad):.1e}")
