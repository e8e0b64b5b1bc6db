"""Sum a rocprofv3 --pmc counter_collection CSV per kernel name (first 60 chars): python tools/pmc_quick.py <dir>"""
import csv
import glob
import sys
from collections import defaultdict

acc, n = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
for f in glob.glob(f"{sys.argv[1]}/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k][r["Counter_Name"]] += 1
for k, c in sorted(acc.items()):
    d = dict(c)
    extra = []
    if d.get("SQ_LDS_IDX_ACTIVE"):
        extra.append(f"lds_conflict_share={d.get('SQ_LDS_BANK_CONFLICT', 0) / d['SQ_LDS_IDX_ACTIVE']:.3f}")
    if d.get("SQ_INSTS_MFMA"):
        extra.append(f"valu_per_mfma={d.get('SQ_INSTS_VALU', 0) / d['SQ_INSTS_MFMA']:.2f} lds_per_mfma={d.get('SQ_INSTS_LDS', 0) / d['SQ_INSTS_MFMA']:.2f}")
    if d.get("SQ_BUSY_CYCLES") and d.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        extra.append(f"mfma_busy/sq_busy={d['SQ_VALU_MFMA_BUSY_CYCLES'] / d['SQ_BUSY_CYCLES']:.3f}")
    print(f"{k:70s} x{max(n[k].values()):4d} " + " ".join(extra) + " | " + " ".join(f"{a}={v:.3g}" for a, v in sorted(d.items())))
