"""Micro-benchmark (GPU box) of vmtl_conv3x3_small on the `basic` tail shapes at bs 32, 128x256, through the C ABI,
next to the implicit-GEMM kernel on the same shapes."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision_mtl_amd._lib import lib

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--height", type=int, default=128)
ap.add_argument("--width", type=int, default=256)
ap.add_argument("--only", default="")
args = ap.parse_args()
dev = torch.device("cuda:0")
L = lib()
st = torch.cuda.current_stream().cuda_stream
B, H, W = args.batch, args.height, args.width


def c4(c):
    return (c + 3) // 4 * 4


def timeit(fn):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / args.reps * 1e3


tiles = L.raw("vmtl_conv3x3_small_stat_rows")(B, H, W)
for name, Cin, Cout in [("conv2 33->33", 33, 33), ("heads 33->20", 33, 20), ("heads dgrad 20->33", 20, 33),
                        ("mtan 32->32", 32, 32), ("cs 32->16", 32, 16), ("cs 16->32", 16, 32), ("cs 16->16", 16, 16),
                        ("cs 16->19", 16, 19), ("cs 16->1", 16, 1)]:
    if args.only and args.only not in name:
        continue
    Cs, ldy = c4(Cin), c4(Cout)
    x, x2 = torch.randn(B, H, W, Cs, device=dev), torch.randn(B, H, W, Cs, device=dev)
    a_out = torch.empty_like(x)
    wp = torch.randn(Cout, 9 * Cs, device=dev) * 0.05
    y = torch.empty(B, H, W, ldy, device=dev)
    xz = torch.randn(B, H, W, ldy, device=dev)
    stats = torch.empty(tiles, 2, ldy, device=dev)
    pa, pb, pc = (torch.randn(Cs, device=dev) for _ in range(3))
    vec = [torch.rand(ldy, device=dev) + 0.5 for _ in range(4)]
    flop = 2.0 * B * H * W * Cout * 9 * Cin

    def small(**kw):
        a = dict(x=x, x2=None, pa=None, pb=None, pc=None, act_in=0, a_out=None, wp=wp, bias=None, y=y, yb=None, Ca=0,
                 stats=None, ep_mode=0, ez_x=None, ez_mean=None, ez_invstd=None, ez_gamma=None, ez_beta=None, ez_act=1,
                 B=B, H=H, W=W, Cs=Cs, ldy=ldy, Nw=Cout, Cout=Cout, stream=st)
        a.update(kw)
        return lambda: L.callk("vmtl_conv3x3_small", **a)

    ez = dict(ez_x=xz, ez_mean=vec[0], ez_invstd=vec[1], ez_gamma=vec[2], ez_beta=vec[3])
    variants = [("plain", small()),
                ("stats only (mode 1)", small(stats=stats, ep_mode=1)),
                ("prologue relu + a_out", small(pa=pa, pc=pc, act_in=1, a_out=a_out)),
                ("prologue + stats (mode 1)", small(pa=pa, pc=pc, act_in=1, a_out=a_out, stats=stats, ep_mode=1)),
                ("mode 2 (BN bwd epilogue)", small(stats=stats, ep_mode=2, **ez)),
                ("2-op prologue + a_out + mode 2", small(x2=x2, pa=pa, pb=pb, pc=pc, a_out=a_out, stats=stats, ep_mode=2, **ez))]
    for vn, fn in variants:
        us = timeit(fn)
        print(f"small  {name:20s} {vn:34s} {us:8.1f} us  {flop / us / 1e6:6.1f} TF", flush=True)
    ig = lambda: L.callk("vmtl_conv2d_fwd", x=x, wp=wp, bias=None, y=y, stats=None, B=B, H=H, W=W, Cs=Cs, Ho=H, Wo=W,
                         ldy=ldy, Nw=Cout, Cout=Cout, KH=3, KW=3, stride=1, pad=1, act=0, shuffle=0, stream=st)
    us = timeit(ig)
    print(f"igemm  {name:20s} {'plain':34s} {us:8.1f} us  {flop / us / 1e6:6.1f} TF", flush=True)
    st_ig = torch.empty(L.raw("vmtl_conv2d_stats_rows")(B, H, W, ldy) + 1, 2, ldy, device=dev)
    igs = lambda: L.callk("vmtl_conv2d_fwd", x=x, wp=wp, bias=None, y=y, stats=st_ig, B=B, H=H, W=W, Cs=Cs, Ho=H, Wo=W,
                          ldy=ldy, Nw=Cout, Cout=Cout, KH=3, KW=3, stride=1, pad=1, act=0, shuffle=0, stream=st)
    us = timeit(igs)
    print(f"igemm  {name:20s} {'with the statistics epilogue':34s} {us:8.1f} us  {flop / us / 1e6:6.1f} TF", flush=True)
