#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3d; mkdir -p $O
cd $R
MT="--model mtan --batch 16 --height 256 --width 256 --classes 14 --only-headline --no-cpu-baseline --no-roofline"
VMTL_EXP_ONE_FILL=1 python bench.py $MT > $O/mtan_onefill.json 2> $O/mtan_onefill.err; echo onefill
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python bench.py $MT > $O/mtan_nopc.json 2> $O/mtan_nopc.err; echo nopc
for bs in 8 32; do
  B="--batch $bs --only-headline --no-cpu-baseline --no-roofline"
  python bench.py $B > $O/basic${bs}_default.json 2> $O/basic${bs}_default.err
  DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python bench.py $B > $O/basic${bs}_nopc.json 2> $O/basic${bs}_nopc.err
  VMTL_EXP_ONE_FILL=1 python bench.py $B > $O/basic${bs}_onefill.json 2> $O/basic${bs}_onefill.err
  VMTL_SIDE_STREAM=0 python bench.py $B > $O/basic${bs}_side0.json 2> $O/basic${bs}_side0.err
  VMTL_SIDE_STREAM=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python bench.py $B > $O/basic${bs}_side0_nopc.json 2> $O/basic${bs}_side0_nopc.err
done
VMTL_STAMPS=1 python bench.py --batch 8 --only-headline --no-cpu-baseline --no-roofline > $O/basic8_stamps.json 2> $O/basic8_stamps.err
for f in $O/*.json; do python3 -c "
import sys,json
d=json.load(open('$f')); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['config'].get('ms_per_step_eager'))"; done
echo "all done"
