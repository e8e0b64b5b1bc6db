#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3t; mkdir -p $O
cd $R
python -m pytest tests/test_kernels_gpu.py tests/test_tight_grads_gpu.py tests/test_basic_gpu.py -m gpu -x -q > $O/test.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 $O/test.log
[ $rc -eq 0 ] || exit 1
VMTL_LIB=$R/ab/libvmtl_head.so python tools/bench_conv.py > $O/conv_head.log 2>&1
python tools/bench_conv.py > $O/conv_cur.log 2>&1
paste <(awk '{print $1}' $O/conv_head.log) <(grep -o "dgrad.*bnbwd[^)]*)" $O/conv_head.log) <(grep -o "dgrad.*bnbwd[^)]*)" $O/conv_cur.log) | tail -20
