#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3h; mkdir -p $O
cd $R
for g in 0 2 3; do VMTL_GLDS=$g python tools/bench_conv.py --only blk --reps 5 > $O/conv_glds$g.log 2>&1; echo "glds $g: $(tail -1 $O/conv_glds$g.log)"; done
