#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3t; mkdir -p $O
cd $R
python -m pytest tests/test_kernels_gpu.py tests/test_mtan_gpu.py tests/test_csnet_gpu.py tests/test_basic_gpu.py tests/test_tight_grads_gpu.py -m gpu -x -q > $O/test.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/test.log
[ $rc -eq 0 ] || { grep -E "Error|assert |FAILED|mismatch" $O/test.log | head -20; exit 1; }
for v in 0 1 0 1; do
  VMTL_SMALL_STATS=$v python bench.py --no-cpu-baseline --only-headline --model mtan --batch 16 --height 256 --width 256 --classes 14 > $O/m_$v.json 2> $O/m_$v.err
  VMTL_SMALL_STATS=$v python bench.py --no-cpu-baseline --only-headline --model csnet > $O/c_$v.json 2> $O/c_$v.err
  python3 -c "
import json
for f in ('$O/m_$v.json','$O/c_$v.json'):
    r=json.load(open(f)); print('small_stats=$v', r['config']['workload'][:14], r['value'], r['ms_per_step'])"
done
