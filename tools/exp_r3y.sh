#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3y; mkdir -p $O
cd $R
for t in 512 768 1024 2048; do
  VMTL_WS_TARGET=$t python tools/bench_conv.py > $O/c_$t.log 2>&1
  echo "== target $t"; grep "halo" $O/c_$t.log | sed 's/M=.*| wgrad/| wgrad/' | cut -c1-120
done
