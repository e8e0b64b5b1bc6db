"""Micro-benchmark (GPU box) of the implicit-GEMM kernels on the `basic` decoder shapes at bs 32,
called through the C ABI.  VMTL_FORCE_TILE=<id> / VMTL_FORCE_WG_SPLITS=<n> override the host heuristics."""
import argparse
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision_mtl_amd._lib import lib

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--only", default="")
ap.add_argument("--stats", action="store_true", help="forward launches also write the BatchNorm partial rows (as in a training step)")
args = ap.parse_args()
dev = torch.device("cuda:0")
L = lib()
st = torch.cuda.current_stream().cuda_stream


def c4(c):
    return (c + 3) // 4 * 4


# (name, B, H, W, Cin, Cout, K)
LAYERS = [("blk0.c1", 32, 8, 16, 1072, 540, 3), ("blk0.c2", 32, 8, 16, 540, 540, 3),
          ("blk1.c1", 32, 16, 32, 580, 270, 3), ("blk1.c2", 32, 16, 32, 270, 270, 3),
          ("blk2.c1", 32, 32, 64, 294, 135, 3), ("blk2.c2", 32, 32, 64, 135, 135, 3),
          ("blk3.c1", 32, 64, 128, 151, 67, 3), ("blk3.c2", 32, 64, 128, 67, 67, 3),
          ("blk4.c1", 32, 128, 256, 67, 33, 3), ("blk4.c2", 32, 128, 256, 33, 33, 3),
          ("head20", 32, 128, 256, 33, 20, 3),
          # same GEMM shapes as blk2.c2 / blk3.c2 without tap re-reads (1x1): isolates the im2col operand traffic
          ("pw.blk2c2", 32, 32, 64, 1215, 135, 1), ("pw.blk3c2", 32, 64, 128, 603, 67, 1),
          # MTAN attention 1x1 convs at full resolution (bs 16, 256x256): output-heavy, 4-6 K steps
          ("pw.mtan192", 16, 256, 256, 128, 192, 1), ("pw.mtan128", 16, 256, 256, 192, 128, 1),
          # MTAN (bs 16, 256x256, first encoder width 32): the C -> C 3x3 convs of the four resolutions
          ("mtan.s0", 16, 256, 256, 32, 32, 3), ("mtan.s1", 16, 128, 128, 64, 64, 3),
          ("mtan.s2", 16, 64, 64, 128, 128, 3), ("mtan.s3", 16, 32, 32, 256, 256, 3),
          # csnet's full-resolution decoder tail (bs 32, 128x256) and MTAN's first conv
          ("cs.32-16", 32, 128, 256, 32, 16, 3), ("cs.16-16", 32, 128, 256, 16, 16, 3), ("cs.16-19", 32, 128, 256, 16, 19, 3),
          ("cs.16-1", 32, 128, 256, 16, 1, 3), ("cs.80-32", 32, 64, 128, 80, 32, 3), ("mtan.c0", 16, 256, 256, 3, 32, 3)]


def timeit(fn):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / args.reps


tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
flops = 0.0
for name, B, H, W, Cin, Cout, K in LAYERS:
    if args.only and args.only not in name:
        continue
    Cs, ldy, KK = c4(Cin), c4(Cout), K * K
    x = torch.randn(B, H, W, Cs, device=dev)
    dy = torch.randn(B, H, W, ldy, device=dev)
    wp = torch.randn(Cout, KK * Cs, device=dev) * 0.01
    wd = torch.randn(Cin, KK * ldy, device=dev) * 0.01
    y = torch.empty(B, H, W, ldy, device=dev)
    dx = torch.empty(B, H, W, Cs, device=dev)
    M = B * H * W
    fl = 2.0 * M * Cout * Cin * KK
    stats = None
    if args.stats:
        stats_t = torch.empty(L.raw("vmtl_conv2d_stats_rows")(B, H, W, ldy) + 1, 2, ldy, device=dev)
        stats = stats_t.data_ptr()
    t_f = timeit(lambda: L.call("vmtl_conv2d_fwd", x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), stats, B, H, W, Cs, H, W,
                                ldy, Cout, Cout, K, K, 1, K // 2, 0, 0, st))
    t_d = timeit(lambda: L.call("vmtl_conv2d_fwd", dy.data_ptr(), wd.data_ptr(), None, dx.data_ptr(), None, B, H, W, ldy, H,
                                W, Cs, Cin, Cin, K, K, 1, K // 2, 0, 0, st))
    # the data gradient with the producer's BatchNorm + ReLU backward in its epilogue (vmtl_conv2d_bnbwd)
    ezs = torch.empty(L.raw("vmtl_conv2d_stats_rows")(B, H, W, Cs) + 1, 2, Cs, device=dev)
    ezx = torch.randn(B, H, W, Cs, device=dev)
    vec = [torch.rand(Cs, device=dev) + 0.5 for _ in range(4)]
    t_z = timeit(lambda: L.call("vmtl_conv2d_bnbwd", dy.data_ptr(), wd.data_ptr(), dx.data_ptr(), ezs.data_ptr(), ezx.data_ptr(),
                                vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(), vec[3].data_ptr(), 1, B, H, W, ldy, H,
                                W, Cs, Cin, Cin, K, K, 1, K // 2, st))
    S = L.raw("vmtl_conv2d_wgrad_splits")(M, Cout, KK * Cs)
    slabs = torch.empty(S, Cout, KK * Cs, device=dev)
    t_w = timeit(lambda: L.call("vmtl_conv2d_wgrad", x.data_ptr(), dy.data_ptr(), slabs.data_ptr(), S, B, H, W, Cs, H, W, ldy,
                                Cout, K, K, 1, K // 2, st))
    t_ws = None
    if K == 3 and L.raw("vmtl_conv3x3_wgrad_small_supported")(Cs, ldy, W):  # the strip-walking halo kernel on the same layer
        ns = L.raw("vmtl_conv3x3_wgrad_small_slabs")(B, H, W)
        slabs_s = torch.empty(ns, Cout, KK * Cs, device=dev)
        t_ws = timeit(lambda: L.call("vmtl_conv3x3_wgrad_small", x.data_ptr(), dy.data_ptr(), slabs_s.data_ptr(), ns, B, H, W, Cs,
                                     ldy, Cout, st))
    tot["fwd"] += t_f
    tot["dgrad"] += t_d
    tot["wgrad"] += t_w
    flops += fl
    print(f"{name:8s} M={M:8d} Cin={Cin:5d} Cout={Cout:4d}  fwd {t_f * 1e3:7.1f} us {fl / t_f / 1e9:6.1f} TF | "
          f"dgrad {t_d * 1e3:7.1f} us {fl / t_d / 1e9:6.1f} TF (+bnbwd {t_z * 1e3:6.1f}) | wgrad(S={S:3d}) {t_w * 1e3:7.1f} us {fl / t_w / 1e9:6.1f} TF"
          + (f" | halo wgrad {t_ws * 1e3:7.1f} us {fl / t_ws / 1e9:6.1f} TF" if t_ws else ""))
print(f"TOTAL fwd {tot['fwd']:.3f} ms ({flops / tot['fwd'] / 1e9:.1f} TF)  dgrad {tot['dgrad']:.3f} ms ({flops / tot['dgrad'] / 1e9:.1f} TF)"
      f"  wgrad {tot['wgrad']:.3f} ms ({flops / tot['wgrad'] / 1e9:.1f} TF)  sum {sum(tot.values()):.3f} ms")
