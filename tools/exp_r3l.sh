#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3l; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q -s > $O/test.log 2>&1; rc=$?; echo "pytest rc=$rc"
grep -E "gradient bar|worst gradient|passed|failed" $O/test.log | tail -20
[ $rc -eq 0 ] || { tail -30 $O/test.log; exit 1; }
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 -c "
import json
d=json.load(open('$O/bench.json'))
for r in [d]+d['configs']:
    c=r['config']; rf=r.get('roofline',{})
    print(c['workload'][:70], r['value'], r['ms_per_step'], 'eager', c.get('ms_per_step_eager'), 'conv', rf.get('achieved'), 'wgrad', rf.get('wgrad_kernel',{}).get('achieved'))"
