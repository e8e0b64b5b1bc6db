#!/bin/bash
# SQ wait / issue breakdown of vmtl_conv3x3_small on the conv2 shape (tools/bench_small.py); run on the GPU box.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_small
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/bench_small.py --only conv2 --reps 2 > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv,glob
acc={}
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "small" not in r["Kernel_Name"]: continue
        k=(r["Kernel_Name"][:48], r["Counter_Name"])
        acc.setdefault(k,[0,0]); acc[k][0]+=float(r["Counter_Value"]); acc[k][1]+=1
names=sorted(set(k[0] for k in acc))
for n in names:
    print(n)
    for (kn,c),(v,cnt) in sorted(acc.items()):
        if kn==n: print(f"   {c:28s} {v/cnt:16.0f} per dispatch ({cnt} dispatches)")
PY
rm -rf $OUT/p[0-9]
