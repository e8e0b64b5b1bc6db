#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3t; mkdir -p $O
cd $R
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q > $O/test.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/test.log
[ $rc -eq 0 ] || exit 1
VMTL_WG_FAST=0 python tools/bench_conv.py > $O/conv_fast0.log 2>&1 && \
VMTL_WG_FAST=1 python tools/bench_conv.py > $O/conv_fast1.log 2>&1
paste <(awk '{print $1, $NF-4, $(NF-3), $(NF-2), $(NF-1), $NF}' $O/conv_fast0.log | sed 's/.*wgrad/wgrad/') <(sed 's/.*wgrad/wgrad/' $O/conv_fast1.log) | cut -c1-150
