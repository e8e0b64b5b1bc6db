#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3n; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_conv_small_gpu.py -m gpu -x -q -k "conv1x1 or conv2d_fwd_bwd" > $O/test_pw.log 2>&1; rc=$?; echo "pytest pw rc=$rc"; tail -3 $O/test_pw.log
[ $rc -eq 0 ] || exit 1
MT="--model mtan --batch 16 --height 256 --width 256 --classes 14 --only-headline --no-cpu-baseline"
for occ in 2 1; do
  VMTL_PW_BIG_OCC=$occ VMTL_CONV_TABLE=1 python bench.py $MT > $O/mtan_occ$occ.json 2> $O/mtan_occ$occ.err
  python3 -c "
import json
d=json.load(open('$O/mtan_occ$occ.json')); print('occ $occ', d['value'], d['ms_per_step'], d['config'].get('ms_per_step_eager'))"
  grep " us " $O/mtan_occ$occ.err | grep "conv1x1" | grep "M= 1048576\|M=  262144" | cut -c18-150 | sort | uniq -c | sort -k8 -n -r | awk '{print}' | head -14
done
