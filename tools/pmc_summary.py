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
FAM = [("conv_igemm_kernel", "conv_igemm_kernel"), ("conv_wgrad_kernel", "conv_wgrad_kernel"),
       ("conv3x3_small_kernel", "conv3x3_small_kernel"), ("pw_gemm_kernel", "pw_gemm_kernel"), ("stitch", "stitch_kernel"),
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
print(json.dumps(res, indent=1))
