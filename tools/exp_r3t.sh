#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3t; mkdir -p $O
cd $R
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "halo_tile or conv2d_fwd_bwd" > $O/test.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/test.log
[ $rc -eq 0 ] || { grep -E "Error|assert |FAILED|mismatch" $O/test.log | head -20; exit 1; }
for v in 1073741824 65536; do
  VMTL_SMALL_MIN_ROWS=$v python bench.py --no-cpu-baseline --only-headline --model csnet > $O/b_$v.json 2> $O/b_$v.err
  VMTL_SMALL_MIN_ROWS=$v python bench.py --no-cpu-baseline --only-headline --model mtan --batch 16 --height 256 --width 256 --classes 14 > $O/m_$v.json 2> $O/m_$v.err
  python3 -c "
import json
for f in ('$O/b_$v.json','$O/m_$v.json'):
    r=json.load(open(f)); print('$v', r['config']['workload'][:30], r['value'], r['ms_per_step'], 'conv', r['roofline'].get('achieved'))"
done
