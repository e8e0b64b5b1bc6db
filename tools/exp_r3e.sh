#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3e; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/test.log 2>&1; echo "pytest rc=$?" | tee -a $O/test.log
MT="--model mtan --batch 16 --height 256 --width 256 --classes 14 --only-headline --no-cpu-baseline --no-roofline"
CS="--model csnet --only-headline --no-cpu-baseline --no-roofline"
for v in 0 1; do
  VMTL_EXP_BRANCH_FILL=$v python bench.py $MT > $O/mtan_bf$v.json 2> $O/mtan_bf$v.err
  VMTL_EXP_BRANCH_FILL=$v python bench.py --batch 8 --only-headline --no-cpu-baseline --no-roofline > $O/basic8_bf$v.json 2> $O/basic8_bf$v.err
  VMTL_EXP_BRANCH_FILL=$v python bench.py --batch 32 --only-headline --no-cpu-baseline --no-roofline > $O/basic32_bf$v.json 2> $O/basic32_bf$v.err
  VMTL_EXP_BRANCH_FILL=$v python bench.py --batch 32 --height 256 --only-headline --no-cpu-baseline --no-roofline > $O/basic256_bf$v.json 2> $O/basic256_bf$v.err
  VMTL_EXP_BRANCH_FILL=$v python bench.py $CS --stitch layer > $O/csl_bf$v.json 2> $O/csl_bf$v.err
  VMTL_EXP_BRANCH_FILL=$v python bench.py $CS --stitch channel > $O/csc_bf$v.json 2> $O/csc_bf$v.err
done
for f in $O/*.json; do python3 -c "
import sys,json
d=json.load(open('$f')); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['config'].get('ms_per_step_eager'))"; done
echo "all done"
