#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3t; mkdir -p $O
cd $R
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q > $O/test.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 $O/test.log
[ $rc -eq 0 ] || exit 1
python tools/bench_conv.py --only blk > $O/conv4.log 2>&1
cut -c1-118 $O/conv4.log | tail -12
python bench.py --no-cpu-baseline --only-headline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 -c "
import json
r=json.load(open('$O/bench.json')); c=r['config']; rf=r.get('roofline',{})
print(c['workload'][:70], r['value'], r['ms_per_step'], 'eager', c.get('ms_per_step_eager'), 'conv', rf.get('achieved'), 'wgrad', rf.get('wgrad_kernel',{}).get('achieved'))"
