#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3t; mkdir -p $O
cd $R
for pr in 0 -1 0 -1; do
  VMTL_MAIN_PRIORITY=$pr python bench.py --no-cpu-baseline --only-headline > $O/p_$pr.json 2> $O/p_$pr.err
  python3 -c "
import json
r=json.load(open('$O/p_$pr.json')); print('main prio $pr', r['value'], r['ms_per_step'])" || tail -3 $O/p_$pr.err
done
VMTL_MAIN_PRIORITY=-1 python bench.py --no-cpu-baseline --only-headline --model mtan --batch 16 --height 256 --width 256 --classes 14 > $O/m.json 2> $O/m.err; python3 -c "
import json
r=json.load(open('$O/m.json')); print('mtan main prio -1', r['value'], r['ms_per_step'])"
