#!/usr/bin/env python
"""Headline benchmark: images/sec of the full multi-task training step graph
(model forward -> CrossEntropy + SILog -> backward to every parameter [+ one RCCL gradient
all-reduce when N > 1]) on synthetic data, fp32, model in train mode — BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one process per GPU over RCCL: either the driver starts the ranks itself (python -m torch.distributed.run
... bench.py --gpus N, WORLD_SIZE set) or - plain `python bench.py --gpus N` - this process starts them: before anything
touches the GPU it spawns N fresh child ranks through torch.distributed.run, relays rank 0's JSON line and exits with
the children's status (the parent never initialises HIP and nothing that has is ever re-executed).

Prints ONE JSON line on rank 0.  The step is the PRODUCT's captured step (vision_mtl_amd.graphed.GraphedStep: fwd + CE +
SILog + bwd replayed from a hipGraph); config.ms_per_step_eager is the same step issued launch by launch from Python,
the way the reference's loop drives it.  config.backend / ranks_seen / loss_per_rank prove that N ranks took part.
Besides the driver contract the line carries
  roofline     : the forward + data-gradient conv launches of one step (conv_igemm_kernel and its two
                 specialisations conv3x3_small_kernel / pw_gemm_kernel) timed launch-by-launch with HIP
                 events on the launch stream; achieved = algorithmic FLOPs per launch / average launch
                 duration, peak = 157.3 TF (exact-fp32 MFMA, MI355X_MICROARCH.md)
  configs      : (default 1-GPU invocation only) the same measurement for the other BASELINE.json
                 configurations: basic bs=8, basic 256x256, csnet (+ the cross-stitch kernel's HBM
                 roofline), mtan 256x256 bs=16 (+ algorithmic HBM GB/s)
  cpu_baseline : the oracle (CPU restatement of the same step graph, oracle/) timed on the host
                 cores of this box on a bounded sample; a reported baseline, not the target.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS_FP32_MFMA = 157.3  # MI355X_MICROARCH.md, chip-level parameters
# algorithmic GFLOP per image, fwd + bwd; used only to quote whole-step TFLOP/s
STEP_GFLOP_PER_IMG = {("basic", 128, 256): 33.07, ("basic", 256, 256): 66.14, ("mtan", 256, 256): 201.2}


_T0 = time.perf_counter()


def log(msg):
    print(f"[bench +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def build(args, device):
    from vision_mtl_amd.lit_module import MTLModule
    from vision_mtl_amd.utils.pipeline_utils import build_model

    torch.manual_seed(11)  # reference cfg.py:194
    ns = argparse.Namespace(model_name=args.model, backbone_weights=None, channel_wise_stitching=args.stitch == "channel")
    model = build_model(ns, argparse.Namespace(num_classes=args.classes)).to(device).train()
    module = MTLModule(model, num_classes=args.classes, device=str(device))
    module.compute_metrics = False  # the metric is fwd + losses + bwd ("no metrics, no logging")
    return model, module


def make_batch(args, device, rank):
    from vision_mtl_amd.data import synthetic_batch

    b = synthetic_batch(args.batch, args.height, args.width, args.classes, seed=11 + rank)
    return {k: v.to(device) for k, v in b.items()}


def time_conv_kernels(module, batch, reps=3):
    """Replay every implicit-GEMM launch of one training step on its own, bracketed by HIP events on
    the launch stream.  Returns per-family (igemm = fwd+dgrad kernel, wgrad) totals."""
    from vision_mtl_amd import ops
    from vision_mtl_amd._lib import lib

    ops._RECORD = []
    try:
        loss = module.training_step(batch, 0)
        loss.backward()
    finally:
        rec, ops._RECORD = ops._RECORD, None
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    fam = {"vmtl_conv2d_fwd": dict(flop=0.0, ms=0.0, launches=0), "vmtl_conv2d_wgrad": dict(flop=0.0, ms=0.0, launches=0),
           "vmtl_stitch": dict(flop=0.0, ms=0.0, launches=0, bytes=0.0)}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, kw, flop, xflop in rec:  # flop = algorithmic (reference-formulation) FLOPs, xflop = executed
        lib().callk(name, stream=stream, **kw)  # warm
        e0.record()
        for _ in range(reps):
            lib().callk(name, stream=stream, **kw)
        e1.record()
        e1.synchronize()
        if name in ("vmtl_conv1x1_cat_wgrad", "vmtl_conv3x3_wgrad_small"):
            name = "vmtl_conv2d_wgrad"
        f = fam[name if name in ("vmtl_conv2d_wgrad", "vmtl_stitch") else "vmtl_conv2d_fwd"]
        ms = e0.elapsed_time(e1) / reps
        if name == "vmtl_stitch":  # HBM-bound: flop slot carries algorithmic bytes (8 B per element)
            f["bytes"] += flop
        f["flop"] += flop
        f["xflop"] = f.get("xflop", 0.0) + xflop
        f["ms"] += ms
        f["launches"] += 1
        if os.environ.get("VMTL_CONV_TABLE") and name != "vmtl_stitch":
            if "Ho" in kw:
                M, K, tag = kw["B"] * kw["Ho"] * kw["Wo"], kw["KH"] * kw["KW"] * kw["Cs"], f"k{kw['KH']}s{kw['stride']}"
            elif "Ks" in kw:
                M, K, tag = kw["M"], kw["Ks"], "k1s1"
            elif "K1" in kw:
                M, K, tag = kw["M"], kw["K1"] + kw["K2s"], "k1cat"
                kw = dict(kw, Nw=kw.get("Nw", 0))
            elif "H2" in kw:
                M, K, tag = 4 * kw["B"] * kw["H2"] * kw["W2"], 4 * kw["C0s"] + 9 * kw["C1s"], "up2 "
            else:
                M, K, tag = kw["B"] * kw["H"] * kw["W"], 9 * kw.get("Cs", 0), "k3s1"
            log(f"{name[5:]:14s} M={M:8d} N={kw.get('Nw', kw.get('Cout', 0)):5d} K={K:6d} {tag} {ms * 1e3:9.1f} us "
                f"{flop / ms / 1e9:7.1f} TF (executed {xflop / ms / 1e9:6.1f})")
    return fam


def cpu_baseline(args):
    """Oracle step (fwd + CE + SILog + bwd, train mode) on the host cores; bounded sample."""
    from oracle.losses import step_losses, synthetic_batch

    # the GPU box gives one GPU's share of the host (16 cores); os.cpu_count() reports the whole machine
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(avail, 16))
    torch.set_num_threads(threads)
    log(f"cpu baseline: {threads} threads (affinity {avail}, cpu_count {os.cpu_count()})")
    bs = 8  # BASELINE.json configs[0]: the reference's own CPU-runnable case
    from vision_mtl_amd.utils.pipeline_utils import build_model

    torch.manual_seed(11)
    ns = argparse.Namespace(model_name=args.model, backbone_weights=None, channel_wise_stitching=args.stitch == "channel")
    sd = build_model(ns, argparse.Namespace(num_classes=args.classes)).state_dict()
    sd = {k: v.clone() for k, v in sd.items()}
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    batch = synthetic_batch(bs, args.height, args.width, args.classes, seed=11)
    if args.model == "basic":
        from oracle.unet_mobilenetv3 import basic_forward as fwd

        run = lambda: fwd(sd, batch["img"], True)
    elif args.model == "mtan":
        from oracle.mtan import mtan_forward

        run = lambda: mtan_forward(sd, batch["img"], ["depth", "segm"], 4, True)
    else:
        from oracle.cross_stitch import csnet_forward

        run = lambda: csnet_forward(sd, batch["img"], ["depth", "segm"], True)

    def step():
        for v in sd.values():
            if v.requires_grad:
                v.grad = None
        step_losses(run(), batch["mask"], batch["depth"])["loss"].backward()

    t0 = time.perf_counter()
    step()  # warm-up
    log(f"cpu baseline warm-up step: {time.perf_counter() - t0:.1f}s")
    n, t0 = 0, time.perf_counter()
    while n < 2 or (time.perf_counter() - t0 < 10.0 and n < 20):
        step()
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(bs * n / dt, 3), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle {args.model} {args.height}x{args.width} bs={bs} fwd+CE+SILog+bwd, {n} steps in {dt:.1f}s"}


def measure(args, device, rank, world, extras=False):
    """One configuration: build, warm up, capture, time exactly args.steps steps; returns the result object
    (rank 0; None elsewhere).  extras: add the roofline (per-launch conv timing) and, at N = 1, the CPU baseline."""
    import torch.distributed as dist

    import gc

    from vision_mtl_amd import dp, ops

    # a previous configuration of the same process (the `configs` array): drop its model, captured graph and packed
    # operands first - left to the cyclic collector they stayed alive and csnet measured 21 ms/step instead of 17.7
    gc.collect()
    ops.packs.purge()
    torch.cuda.empty_cache()
    model, module = build(args, device)
    arena = dp.FlatArena(model)
    batch = make_batch(args, device, rank)
    log(f"{args.model} {args.height}x{args.width} bs={args.batch}: model + batch resident")

    def step():
        ops.stamp("step start")
        loss = module.training_step(batch, 0)
        ops.stamp("forward done")
        loss.backward()
        ops.stamp("backward joined")
        return loss

    def after_step():
        scale = arena.all_reduce_mean()  # one RCCL all-reduce of the flat gradient (no-op at N=1)
        if args.adam:
            arena.adam_step(lr=5e-4, grad_scale=scale)

    # eager warm-up (also primes allocator pools and kernel code objects)
    for i in range(2):
        step()
        after_step()
        module.step_outputs["train"]["loss"].clear()
        torch.cuda.synchronize()
        log(f"eager warm-up step {i} done")

    # the same step driven the way the reference's loop drives it (training_lit.py:81-98): launch by launch from Python
    n = 5
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(n):
        step()
        after_step()
    torch.cuda.synchronize()
    ms_eager = (time.perf_counter() - t1) / n * 1e3
    module.step_outputs["train"]["loss"].clear()
    log(f"eager: {ms_eager:.2f} ms/step")

    if os.environ.get("VMTL_STAMPS") == "1":  # two-stream timeline of one replayed step (tuning aid)
        ops._STAMPS = []
    gstep = None
    if not args.no_graph:
        from vision_mtl_amd.graphed import GraphedStep

        gstep = GraphedStep(module, batch, arena=arena, warmup=1)  # the PRODUCT's captured step (fwd + losses + bwd)
        log("hipGraph captured (vision_mtl_amd.graphed.GraphedStep)")
    graph = gstep.graph if gstep is not None else None

    def run_step():
        if graph is not None:
            graph.replay()  # the batch is already resident in the step's static input buffers (metric: inputs in HBM)
        else:
            step()
        after_step()

    for _ in range(args.warmup):
        run_step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if ops._STAMPS is not None and graph is not None:
        stamps, ops._STAMPS = ops._STAMPS, None
        n = len(stamps) // 3  # GraphedStep's warm-up step, its rehearsal step and the captured step all appended
        graph.replay()
        torch.cuda.synchronize()
        vals = [(int(t.item()), tag) for tag, t in stamps[-n:]]
        t0 = min(v for v, _ in vals)
        for v, tag in sorted(vals):
            log(f"stamp {(v - t0) / 100.0:10.1f} us  {tag}")
    loss_t = (gstep._loss if gstep is not None else step().detach()).reshape(1).float()
    loss_val = float(loss_t.item())
    # proof that `world` ranks took part: every rank contributes a one and its loss (each rank has its own shard)
    ranks_seen, losses, backend = 1, [round(loss_val, 5)], "none"
    if world > 1:
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)
        ranks_seen = int(round(float(ones.item())))
        gathered = [torch.zeros_like(loss_t) for _ in range(world)]
        dist.all_gather(gathered, loss_t)
        losses = [round(float(g.item()), 5) for g in gathered]
        backend = dist.get_backend()
    log(f"timed region done: {dt / args.steps * 1e3:.3f} ms/step")
    # "report Adam-inclusive step time separately" - the same steps followed by the fused Adam
    # update over the flat arena (one launch); measured after the headline so it cannot touch `value`
    ms_adam = None
    if extras and not args.adam:
        n = min(args.steps, 10)
        run_step()
        arena.adam_step(lr=5e-4, grad_scale=1.0)  # untimed: allocates the moment buffers
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n):
            run_step()
            arena.adam_step(lr=5e-4, grad_scale=1.0)
        torch.cuda.synchronize()
        ms_adam = (time.perf_counter() - t1) / n * 1e3
        if world > 1:
            t = torch.tensor([ms_adam], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms_adam = float(t.item())
    if rank != 0:
        return None

    ms = dt / args.steps * 1e3
    value = args.batch * world * args.steps / dt
    out = {
        "metric": f"images/sec fwd+bwd, {args.model} model {args.height}x{args.width}",
        "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.model} {args.height}x{args.width} C={args.classes} bs={args.batch}/GPU, "
                               f"train-mode fwd + CE + SILog + bwd" + (" + Adam" if args.adam else "")
                               + (f" + 1 grad all-reduce ({'RCCL' if backend == 'nccl' else backend})" if world > 1 else ""),
                   "global_batch": args.batch * world, "parallelism": f"dp{world}",
                   "launch": "eager" if graph is None else "hipGraph replay (vision_mtl_amd.graphed.GraphedStep)",
                   "loss": round(loss_val, 5), "backend": backend, "ranks_seen": ranks_seen, "loss_per_rank": losses},
    }
    if args.model == "csnet":
        out["config"]["workload"] += f", {args.stitch}-wise stitching"
        out["config"]["channel_wise_stitching"] = args.stitch == "channel"
    out["config"]["ms_per_step_eager"] = round(ms_eager, 3)
    if ms_adam is not None:
        out["config"]["ms_per_step_with_adam"] = round(ms_adam, 4)
    gf = STEP_GFLOP_PER_IMG.get((args.model, args.height, args.width))
    if gf:
        tf = gf * value / world / 1e3
        out["config"]["step_tflops_per_gpu"] = round(tf, 2)
        out["config"]["step_frac_of_fp32_mfma_peak"] = round(tf / PEAK_TFLOPS_FP32_MFMA, 4)
    if args.model == "mtan" and (args.height, args.width) == (256, 256):
        # ~3.2 GB of minimum activation traffic per image and step (conv outputs written once +
        # read once forward, ~3x that backward); quoted next to the MFMA figure because MTAN sits near the ridge
        gbps = 3.2 * value / world
        out["config"]["algorithmic_hbm_GBps"] = round(gbps, 1)
        out["config"]["algorithmic_hbm_frac_of_8TBps"] = round(gbps / 8000.0, 4)
    if not args.no_roofline:
        fam = time_conv_kernels(module, batch)
        log("per-launch conv timing done")
        ig = fam["vmtl_conv2d_fwd"]
        ach = ig["flop"] / (ig["ms"] * 1e-3) / 1e12 if ig["ms"] > 0 else 0.0
        out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_TFLOPS_FP32_MFMA,
                           "unit": "TFLOP/s", "frac": round(ach / PEAK_TFLOPS_FP32_MFMA, 4), "traffic": None,
                           "kernel": "forward + data-gradient conv launches of one step (conv_igemm_kernel, "
                                     "conv3x3_small_kernel, pw_gemm_kernel)",
                           "launches_per_step": ig["launches"],
                           "avg_launch_us": round(ig["ms"] * 1e3 / max(ig["launches"], 1), 2),
                           "flop_per_launch": round(ig["flop"] / max(ig["launches"], 1)),
                           "executed_tflops": round(ig.get("xflop", ig["flop"]) / (ig["ms"] * 1e-3) / 1e12, 2),
                           "ms_per_step_in_kernel": round(ig["ms"], 3)}
        # HBM bytes per launch from the hardware counters: cannot be collected inside this process, so the
        # committed rocprofv3 --pmc summary of this same workload is quoted (profiles/, tools/profile_configs.sh)
        tag = {("basic", 128, 256, 32): "basic", ("basic", 128, 256, 8): "basic_bs8", ("basic", 256, 256, 32): "basic_256",
               ("csnet", 128, 256, 32): "csnet", ("mtan", 256, 256, 16): "mtan"}.get((args.model, args.height, args.width, args.batch))
        if tag == "csnet" and args.stitch == "layer":
            tag = "csnet_layer"
        prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
        pmc = next((os.path.join(prof, f"{r}_{tag}_pmc.json") for r in ("r03", "r02")
                    if tag and os.path.exists(os.path.join(prof, f"{r}_{tag}_pmc.json"))), "")
        if tag and os.path.exists(pmc):
            with open(pmc) as f:
                ks = json.load(f)["kernels"]
            fams = [ks[k] for k in ("conv_igemm_kernel", "conv3x3_small_kernel", "pw_gemm_kernel") if k in ks and "hbm_bytes_per_launch" in ks[k]]
            if fams:
                tot = sum(k["hbm_bytes_per_launch"] * k["launches"] for k in fams)
                out["roofline"]["traffic"] = round(tot / sum(k["launches"] for k in fams))
                out["roofline"]["traffic_unit"] = "HBM bytes per launch over the same kernels (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE)"
                # NOT measured in this run: counters cannot be collected from inside the process
                out["roofline"]["traffic_source"] = f"committed profile profiles/{os.path.basename(pmc)}"
        wg = fam["vmtl_conv2d_wgrad"]
        if wg["ms"] > 0:
            wach = wg["flop"] / (wg["ms"] * 1e-3) / 1e12
            out["roofline"]["wgrad_kernel"] = {"achieved": round(wach, 2), "frac": round(wach / PEAK_TFLOPS_FP32_MFMA, 4),
                                               "launches_per_step": wg["launches"],
                                               "ms_per_step_in_kernel": round(wg["ms"], 3)}
        st = fam.get("vmtl_stitch")
        if st and st["ms"] > 0:
            # the cross-stitch kernel is HBM-bound: 8 B per element (read + write), SURVEY.md section 8(d)
            gb = st["bytes"] / (st["ms"] * 1e-3) / 1e9
            out["roofline"]["stitch_kernel"] = {"bound": "hbm", "achieved": round(gb, 1), "peak": 8000.0, "unit": "GB/s",
                                                "frac": round(gb / 8000.0, 4), "launches_per_step": st["launches"],
                                                "bytes_per_launch": round(st["bytes"] / st["launches"]),
                                                "ms_per_step_in_kernel": round(st["ms"], 3)}
    if world == 1 and extras and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    # drop this configuration's device state before the next one is built
    del graph, gstep, arena, module, model, batch
    ops.packs.refresh()
    torch.cuda.empty_cache()
    return out


# the other BASELINE.json configurations, timed after the headline by the default 1-GPU invocation
_REHEARSAL_STREAM = None

EXTRA_CONFIGS = [dict(model="basic", batch=8, height=128, width=256, classes=19),    # the metric string's literal bs=8
                 dict(model="basic", batch=32, height=256, width=256, classes=19),   # north_star: "and 256x256 batches"
                 # configs[2], both stitch modes (the reference CLI default is layer-wise, utils/utils.py:27)
                 dict(model="csnet", batch=32, height=128, width=256, classes=19, stitch="layer"),
                 dict(model="csnet", batch=32, height=128, width=256, classes=19, stitch="channel"),
                 dict(model="mtan", batch=16, height=256, width=256, classes=14)]    # configs[3]


def spawn_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` without a torch.distributed.run environment: start N fresh ranks (one process per
    GPU) and relay their output.  Runs BEFORE anything in this process touches the GPU; the parent only waits."""
    import socket
    import subprocess

    with socket.socket() as sock:  # a free rendezvous port
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    log(f"spawning {n} ranks: {' '.join(cmd)}")
    proc = subprocess.run(cmd, env=env)  # stdout / stderr are inherited: rank 0's JSON line goes straight through
    return proc.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="basic", choices=["basic", "mtan", "csnet"])
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--height", type=int, default=128)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--classes", type=int, default=19)
    ap.add_argument("--stitch", default="layer", choices=["layer", "channel"],
                    help="csnet: --channel_wise_stitching of the reference CLI (default off = layer-wise, utils/utils.py:27)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--adam", action="store_true", help="include the fused Adam update in the timed step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--only-headline", action="store_true", help="skip the `configs` array of the other BASELINE configurations")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="only the multi-rank plumbing (rendezvous, one all-reduce, rank 0's JSON line): runs on CPU over gloo")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))  # nothing above this line touches the GPU

    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:  # checked BEFORE the rendezvous (which would wait for the others)
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}: launch with "
                         f"torch.distributed.run --nproc-per-node {args.gpus}, or drop the WORLD_SIZE variable")
    from vision_mtl_amd import dp

    rank, world, local_rank = dp.init_distributed()
    if args.launcher_selftest:
        import torch.distributed as dist

        ones = torch.ones(1, device="cuda" if torch.cuda.is_available() and dist.get_backend() == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(ones)
        if rank == 0:
            print(json.dumps({"selftest": True, "n_gpus": world, "ranks_seen": int(ones.item()),
                              "backend": dist.get_backend() if world > 1 else "none"}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    device = torch.device("cuda", local_rank)
    log(f"rank {rank}/{world} on {torch.cuda.get_device_name(device)}; building {args.model}")
    out = measure(args, device, rank, world, extras=True)
    default_headline = (args.model, args.batch, args.height, args.width) == ("basic", 32, 128, 256)
    if default_headline and not args.only_headline and not args.no_graph:
        # driver-visible lines for the other BASELINE.json configurations (same measurement, fewer steps).  At N > 1 every
        # rank runs them too (north_star: 128x256 and 256x256 batches at 1/2/4/8 GPUs), without the per-launch roofline
        if rank == 0:
            out["configs"] = []
        for c in EXTRA_CONFIGS:
            a = argparse.Namespace(**{**vars(args), **c, "steps": min(args.steps, 10), "warmup": min(args.warmup, 3),
                                      "no_roofline": args.no_roofline or world > 1})
            r = measure(a, device, rank, world)
            if rank != 0:
                continue
            keep = {k: r[k] for k in ("metric", "value", "unit", "n_gpus", "ms_per_step", "steps", "warmup", "dtype") if k in r}
            keep["config"] = r["config"]
            if "roofline" in r:
                keep["roofline"] = r["roofline"]
            out["configs"].append(keep)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
