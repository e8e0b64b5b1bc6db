"""Micro-benchmark (GPU box) of vmtl_conv2d_up2_fwd on the two narrow decoder shapes of `basic` (bs 32, 128x256), one
tile configuration per process: VMTL_FORCE_TILE=<id> python tools/bench_up2.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision_mtl_amd._lib import lib

dev = torch.device("cuda:0")
L = lib()
st = torch.cuda.current_stream().cuda_stream
for name, B, H2, W2, C0, C1, Cout in [("block5 conv1 67->33", 32, 64, 128, 67, 0, 33), ("block4 conv1 135+16->67", 32, 32, 64, 135, 16, 67)]:
    c4 = lambda c: (c + 3) // 4 * 4
    C0s, C1s, ldy = c4(C0), c4(C1), c4(Cout)
    xl = torch.randn(B, H2, W2, C0s, device=dev)
    skip = torch.randn(B, 2 * H2, 2 * W2, C1s, device=dev) if C1 else None
    Ktot = 4 * C0s + 9 * C1s
    wp = torch.randn(4, Cout, Ktot, device=dev) * 0.05
    y = torch.empty(B, 2 * H2, 2 * W2, ldy, device=dev)
    bm = L.raw("vmtl_conv2d_up2_stats_block")(B, H2, W2, ldy)
    M = B * H2 * W2
    stats = torch.empty(4 * (M // bm), 2, ldy, device=dev) if M % bm == 0 and not os.environ.get("UP2_NO_STATS") else None
    fn = lambda: L.callk("vmtl_conv2d_up2_fwd", xl=xl, skip=skip, wp_eff=wp, y=y, stats=stats, B=B, H2=H2, W2=W2, C0s=C0s,
                         C1s=C1s, ldy=ldy, Cout=Cout, stream=st)
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    xflop = 2.0 * 4 * M * Cout * (4 * C0 + 9 * C1)
    print(f"tile {os.environ.get('VMTL_FORCE_TILE', 'auto'):>4s}  {name:26s} {us:8.1f} us  executed {xflop / us / 1e6:6.1f} TF", flush=True)

# the data gradient of the same two layers w.r.t. the low-res input: a 4x4 stride-2 conv over dY
for name, B, H2, W2, C0, Cout in [("block5 conv1 dgrad 33->67", 32, 64, 128, 67, 33), ("block4 conv1 dgrad 67->135", 32, 32, 64, 135, 67)]:
    c4 = lambda c: (c + 3) // 4 * 4
    Cs, ldy = c4(Cout), c4(C0)
    dy = torch.randn(B, 2 * H2, 2 * W2, Cs, device=dev)
    wd = torch.randn(C0, 16 * Cs, device=dev) * 0.05
    dx = torch.empty(B, H2, W2, ldy, device=dev)
    fn = lambda: L.callk("vmtl_conv2d_fwd", x=dy, wp=wd, bias=None, y=dx, stats=None, B=B, H=2 * H2, W=2 * W2, Cs=Cs, Ho=H2,
                         Wo=W2, ldy=ldy, Nw=C0, Cout=C0, KH=4, KW=4, stride=2, pad=1, act=0, shuffle=0, stream=st)
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    xflop = 2.0 * B * H2 * W2 * C0 * 16 * Cout
    print(f"tile {os.environ.get('VMTL_FORCE_TILE', 'auto'):>4s}  {name:26s} {us:8.1f} us  executed {xflop / us / 1e6:6.1f} TF", flush=True)
