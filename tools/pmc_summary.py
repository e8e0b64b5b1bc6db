"""Summarise the rocprofv3 --pmc passes tools/profile_configs.sh wrote under <dir>/pmc*/ into one JSON on stdout:
per kernel family the launch count and the sum of every counter over the family's dispatches, plus the derived
figures DESIGN.md quotes (MFMA utilisation = share of SIMD cycles with the matrix pipe busy, LDS bank-conflict share, L2 hit rate, HBM bytes per launch with the
gfx950 corrections of MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are reported in KiB, and FETCH_SIZE tallies
128-byte requests at 64 bytes -> doubled)."""
import csv
import glob
import json
import re
import sys

d, cmd = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
# optional: the kernel-trace statistics CSV and the bench line of the SAME configuration (tools/profile_configs.sh passes
# them): per-family ms/step and the conv roofline recomputed from the profiler's own durations
stats_csv = sys.argv[3] if len(sys.argv) > 3 else None
bench_line = sys.argv[4] if len(sys.argv) > 4 else None
FAM = [("conv_igemm_kernel", "conv_igemm_kernel"), ("conv_wgrad_kernel", "conv_wgrad_kernel"),
       ("wgrad_small_kernel", "wgrad_small_kernel"),
       ("conv3x3_small_kernel", "conv3x3_small_kernel"), ("pw_gemm_kernel", "pw_gemm_kernel"), ("pw_big_kernel", "pw_big_kernel"),
       ("stitch", "stitch_kernel"),
       ("bn_", "bn_* (BatchNorm family)"), ("dwconv", "dwconv_*"), ("pack", "pack / unpack"), ("at::native", "ATen"),
       ("rocclr", "rocclr copy / fill")]


def family(name):
    for key, fam in FAM:
        if key in name:
            return fam
    return "other"


out = {}
for f in sorted(glob.glob(f"{d}/pmc*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        fam = out.setdefault(family(r["Kernel_Name"]), {"launches": {}, "counters": {}})
        c = r["Counter_Name"]
        fam["counters"][c] = fam["counters"].get(c, 0.0) + float(r["Counter_Value"])
        fam["launches"][c] = fam["launches"].get(c, 0) + 1
res = {"command": f"tools/profile_configs.sh: rocprofv3 --kernel-trace --pmc <set> --output-format csv -- python3 bench.py {cmd} "
                  "--steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph (VMTL_SIDE_STREAM=0; one pass per counter set)",
       "note": "sums over all dispatches of the run; FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; hbm_bytes_per_launch = "
               "(2 * FETCH_SIZE + WRITE_SIZE) * 1024 / launches (gfx950: 128-B fetches are tallied at 64 B)",
       "kernels": {}}
for fam, v in out.items():
    c, n = v["counters"], v["launches"]
    k = {"launches": max(n.values()) if n else 0}
    k.update({name: val for name, val in sorted(c.items())})
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
        # SQ_VALU_MFMA_BUSY_CYCLES = 32 cycles per v_mfma_f32_16x16x4_f32, summed over the 1024 SIMDs;
        # GRBM_GUI_ACTIVE = kernel cycles summed over the 8 XCDs (MI355X_MICROARCH.md, "DVFS give-back")
        k["mfma_util"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    if c.get("SQ_INSTS_MFMA"):
        k["valu_per_mfma"] = c.get("SQ_INSTS_VALU", 0.0) / c["SQ_INSTS_MFMA"]
    if c.get("SQ_LDS_IDX_ACTIVE"):
        k["lds_bank_conflict_share"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    if "TCC_HIT_sum" in c:
        tot = c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0.0)
        k["l2_hit_rate"] = c["TCC_HIT_sum"] / tot if tot else None
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c and n.get("FETCH_SIZE"):
        k["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / n["FETCH_SIZE"]
    res["kernels"][fam] = k
if stats_csv and bench_line:
    rows = list(csv.DictReader(open(stats_csv)))
    line = json.loads([l for l in open(bench_line) if l.startswith("{")][-1])
    rf = line.get("roofline", {})
    # whole-step executions of the traced run = calls of a kernel that runs exactly once per step
    steps = max([int(r["Calls"]) for r in rows if "silog_fwd" in r["Name"]] or [0])
    # every conv launch of one recorded step is replayed 1 + 3 times by bench.py's per-launch roofline timing
    replays = 4
    fam_ms, fam_calls = {}, {}
    for r in rows:
        f = family(r["Name"])
        fam_ms[f] = fam_ms.get(f, 0.0) + float(r["TotalDurationNs"]) * 1e-6
        fam_calls[f] = fam_calls.get(f, 0) + int(r["Calls"])
    conv_fams = ("conv_igemm_kernel", "conv_wgrad_kernel", "wgrad_small_kernel", "conv3x3_small_kernel", "pw_gemm_kernel", "pw_big_kernel", "stitch_kernel")
    per_step = {f: (ms / (steps + replays) if f in conv_fams else ms / max(steps, 1)) for f, ms in fam_ms.items()}
    trace = {"source": "rocprofv3 --kernel-trace --stats of the same configuration (VMTL_SIDE_STREAM=0: isolated durations)",
             "step_executions": steps, "conv_replays_per_launch": replays,
             "ms_per_step_by_family": {f: round(v, 4) for f, v in sorted(per_step.items(), key=lambda kv: -kv[1])},
             "launches_per_step_by_family": {f: round(c / ((steps + replays) if f in conv_fams else max(steps, 1)), 1)
                                             for f, c in sorted(fam_calls.items())}}
    if rf and steps:
        fd = sum(per_step.get(f, 0.0) for f in ("conv_igemm_kernel", "conv3x3_small_kernel", "pw_gemm_kernel", "pw_big_kernel"))
        flop = rf["flop_per_launch"] * rf["launches_per_step"]
        tf = flop / (fd * 1e-3) / 1e12 if fd else 0.0
        trace["conv_fwd_dgrad_recomputed"] = {"ms_per_step_in_kernel": round(fd, 3), "algorithmic_flop_per_step": flop,
                                              "achieved_tflops": round(tf, 2), "frac_of_157.3": round(tf / 157.3, 4),
                                              "bench_line_frac_unprofiled": rf.get("frac")}
        wg = rf.get("wgrad_kernel")
        wms = per_step.get("conv_wgrad_kernel", 0.0) + per_step.get("wgrad_small_kernel", 0.0)  # both weight-gradient kernels
        if wg and wms:
            wflop = wg["achieved"] * 1e12 * wg["ms_per_step_in_kernel"] * 1e-3
            wtf = wflop / (wms * 1e-3) / 1e12
            trace["conv_wgrad_recomputed"] = {"ms_per_step_in_kernel": round(wms, 3),
                                              "achieved_tflops": round(wtf, 2), "frac_of_157.3": round(wtf / 157.3, 4),
                                              "bench_line_frac_unprofiled": wg.get("frac")}
    res["kernel_trace"] = trace
print(json.dumps(res, indent=1))
