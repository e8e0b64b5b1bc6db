#!/bin/bash
# A/B of library builds (ab/libvmtl_<name>.so vs the in-tree one) on one bench configuration: same box, back to back, twice each
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3w; mkdir -p $O
cd $R
VARIANTS=${VARIANTS:-head cur}
for rep in 1 2; do
for v in $VARIANTS; do
  if [ $v = cur ]; then unset VMTL_LIB; else export VMTL_LIB=$R/ab/libvmtl_$v.so; fi
  python bench.py --no-cpu-baseline --only-headline "$@" > $O/b_$v$rep.json 2> $O/b_$v$rep.err
  python3 -c "
import json
r=json.load(open('$O/b_$v$rep.json')); c=r['config']; rf=r.get('roofline',{})
print('$v$rep', r['value'], r['ms_per_step'], 'conv', rf.get('achieved'), 'wgrad', rf.get('wgrad_kernel',{}).get('achieved'))"
done
done
