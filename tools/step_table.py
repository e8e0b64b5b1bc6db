"""Tuning aid (GPU box): one eager training step with EVERY C-ABI launch bracketed by HIP events, printed as a
per-launch-shape table (entry point, phase, main dimensions, launches, total us, TF for the GEMM kernels).

    VMTL_SIDE_STREAM=0 python tools/step_table.py --model mtan --batch 16 --height 256 --width 256 --classes 14

The step is issued behind a device-side sleep, so the host runs ahead of the GPU and the event pairs time the kernels,
not the Python launch gaps.  Side stream off: one stream, isolated durations (the same convention as profiles/)."""
import argparse
import collections
import os
import sys

os.environ.setdefault("VMTL_SIDE_STREAM", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from vision_mtl_amd import ops
from vision_mtl_amd._lib import lib

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="basic")
ap.add_argument("--stitch", default="layer")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--height", type=int, default=128)
ap.add_argument("--width", type=int, default=256)
ap.add_argument("--classes", type=int, default=19)
ap.add_argument("--top", type=int, default=70)
ap.add_argument("--per-launch", action="store_true", help="also list every launch in issue order")
args = ap.parse_args()
dev = torch.device("cuda:0")
model, module = bench.build(args, dev)
from vision_mtl_amd import dp

arena = dp.FlatArena(model)
batch = bench.make_batch(args, dev, 0)
for _ in range(2):
    module.training_step(batch, 0).backward()
    module.step_outputs["train"]["loss"].clear()
torch.cuda.synchronize()

rec = []
phase = ["fwd"]
DIMS = ("B", "H", "W", "Ho", "Wo", "H2", "W2", "M", "Cs", "C0s", "C1s", "Ks", "K1", "K2s", "ldy", "Nw", "Cout", "C", "KH", "stride",
        "n", "rows", "shuffle", "act", "mode")


def _k(name, _flop=None, _xflop=None, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib().callk(name, stream=ops._stream(), **kw)
    e1.record()
    dims = " ".join(f"{k}={kw[k]}" for k in DIMS if k in kw and isinstance(kw[k], int))
    rec.append((name[5:], phase[0], dims, _flop, e0, e1))


orig, ops._k = ops._k, _k
torch.cuda._sleep(int(2.0e8))  # ~0.1 s of backlog
loss = module.training_step(batch, 0)
phase[0] = "bwd"
loss.backward()
torch.cuda.synchronize()
ops._k = orig

agg = collections.OrderedDict()
tot = 0.0
for name, ph, dims, flop, e0, e1 in rec:
    us = e0.elapsed_time(e1) * 1e3
    tot += us
    a = agg.setdefault((name, ph, dims), [0, 0.0, 0.0])
    a[0] += 1
    a[1] += us
    a[2] += flop or 0.0
    if args.per_launch:
        print(f"{ph} {name:28s} {us:8.1f} us  {dims}")
print(f"# {len(rec)} launches, {tot / 1e3:.3f} ms inside event pairs")
fam = collections.Counter()
for (name, ph, dims), (n, us, flop) in agg.items():
    fam[name] += us
print("# by entry point (ms): " + ", ".join(f"{k} {v / 1e3:.2f}" for k, v in fam.most_common(25)))
for (name, ph, dims), (n, us, flop) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:args.top]:
    tf = f"{flop / us / 1e6:6.1f} TF" if flop else "         "
    print(f"{us:9.1f} us  x{n:<3d} {tf}  {ph} {name:26s} {dims}")
