#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/test.log 2>&1; echo "pytest rc=$?" | tee -a $O/test.log
MT="--model mtan --batch 16 --height 256 --width 256 --classes 14 --only-headline --no-cpu-baseline --no-roofline"
VMTL_EXP_LEGACY_CAPTURE=1 python bench.py $MT > $O/mtan_legacy.json 2> $O/mtan_legacy.err; echo legacy
VMTL_EXP_NO_FORK=1 python bench.py $MT > $O/mtan_nofork.json 2> $O/mtan_nofork.err; echo nofork
VMTL_EXP_ALWAYS_FILL=1 python bench.py $MT > $O/mtan_fill.json 2> $O/mtan_fill.err; echo fill
VMTL_EXP_ALWAYS_FILL=1 VMTL_EXP_NO_FORK=1 python bench.py $MT > $O/mtan_fill_nofork.json 2> $O/mtan_fill_nofork.err; echo fill_nofork
grep -h '"value"' $O/mtan_*.json | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'], d['config'].get('ms_per_step_eager'))"
cd /tmp && export TMPDIR=/tmp
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  n=$(echo $c | cut -c1-12 | tr ' ' '_')
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$n -- python3 $R/tools/bench_conv.py --only blk2.c2 --reps 2 > $O/pmc_$n.log 2>&1 || exit 1
  python3 $R/tools/pmc_quick.py $O/pmc_$n > $O/pmc_$n.txt; rm -rf $O/pmc_$n
done
echo "all done"
