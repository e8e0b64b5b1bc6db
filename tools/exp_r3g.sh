#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3g; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_conv_small_gpu.py -m gpu -x -q -k "conv1x1 or conv2d_fwd_bwd" > $O/test_pw.log 2>&1; rc=$?; echo "pytest pw rc=$rc" | tee -a $O/test_pw.log
tail -30 $O/test_pw.log
[ $rc -eq 0 ] || exit 1
MT="--model mtan --batch 16 --height 256 --width 256 --classes 14 --only-headline --no-cpu-baseline"
VMTL_CONV_TABLE=1 python bench.py $MT > $O/mtan_big.json 2> $O/mtan_big.err; echo "mtan big rc=$?"
VMTL_PW_BIG=0 python bench.py $MT --no-roofline > $O/mtan_nobig.json 2> $O/mtan_nobig.err
python3 -c "
import json
for f in ['mtan_big','mtan_nobig']:
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d['config'].get('ms_per_step_eager'))"
grep " us " $O/mtan_big.err | grep "conv1x1" | cut -c18-150 | sort | uniq -c | sort -k8 -n -r | head -40
