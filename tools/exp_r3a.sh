#!/bin/bash
# round-3 experiment batch A (GPU box): full GPU tests, wgrad LDS-stride A/B, MTAN graph vs eager, per-launch tables
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3a; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/test.log 2>&1; echo "pytest rc=$?" | tee -a $O/test.log
for pad in 0 1 2; do VMTL_WG_PAD=$pad python tools/bench_conv.py --only blk --reps 5 > $O/conv_pad$pad.log 2>&1; done
echo "conv A/B done"
cd /tmp && export TMPDIR=/tmp
for pad in 0 1; do
  VMTL_WG_PAD=$pad timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_pad$pad -- python3 $R/tools/bench_conv.py --only blk2.c2 --reps 2 > $O/pmc_pad$pad.log 2>&1 || exit 1
  python3 $R/tools/pmc_quick.py $O/pmc_pad$pad > $O/pmc_pad$pad.txt; rm -rf $O/pmc_pad$pad
done
echo "pmc done"
cd $R
MT="--model mtan --batch 16 --height 256 --width 256 --classes 14 --only-headline --no-cpu-baseline"
VMTL_SIDE_STREAM=0 python bench.py $MT --no-roofline > $O/mtan_side0.json 2> $O/mtan_side0.err
VMTL_SIDE_STREAM=1 python bench.py $MT --no-roofline > $O/mtan_side1.json 2> $O/mtan_side1.err
VMTL_CONV_TABLE=1 python bench.py $MT > $O/mtan_table.json 2> $O/mtan_table.err
VMTL_CONV_TABLE=1 python bench.py --batch 8 --only-headline --no-cpu-baseline > $O/bs8_table.json 2> $O/bs8_table.err
VMTL_CONV_TABLE=1 python bench.py --only-headline --no-cpu-baseline > $O/bs32_table.json 2> $O/bs32_table.err
echo "all done"
