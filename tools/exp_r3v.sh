#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3v; mkdir -p $O
cd $R
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 -c "
import json
d=json.load(open('$O/bench.json'))
for r in [d]+d['configs']:
    c=r['config']; rf=r.get('roofline',{})
    print(c['workload'][:70], r['value'], r['ms_per_step'], 'eager', c.get('ms_per_step_eager'), 'conv', rf.get('achieved'), 'wgrad', rf.get('wgrad_kernel',{}).get('achieved'))"
