#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3m; mkdir -p $O
cd $R
for v in base prio base prio; do
  L=""; [ $v = prio ] && L=$R/vision_mtl_amd/csrc/libvmtl_prio.so
  VMTL_LIB=$L python tools/bench_conv.py --only blk --reps 5 > $O/conv_$v.log 2>&1; echo "$v: $(tail -1 $O/conv_$v.log)"
done
for v in base prio; do
  L=""; [ $v = prio ] && L=$R/vision_mtl_amd/csrc/libvmtl_prio.so
  VMTL_LIB=$L python bench.py --only-headline --no-cpu-baseline --no-roofline > $O/bench_$v.json 2> $O/bench_$v.err
  VMTL_LIB=$L python bench.py --model mtan --batch 16 --height 256 --width 256 --classes 14 --only-headline --no-cpu-baseline --no-roofline > $O/mtan_$v.json 2> $O/mtan_$v.err
  python3 -c "
import json
for f in ['bench_$v','mtan_$v']:
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'])"
done
