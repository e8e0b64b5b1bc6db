#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3t; mkdir -p $O
cd $R
python -m pytest tests/test_kernels_gpu.py tests/test_conv_small_gpu.py -m gpu -x -q > $O/test.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/test.log
[ $rc -eq 0 ] || { grep -E "Error|assert|FAILED" $O/test.log | head -20; exit 1; }
python tools/bench_conv.py --stats > $O/conv_stats.log 2>&1
cut -c1-110 $O/conv_stats.log | tail -20
python tools/bench_small.py --batch 16 --height 256 --width 256 --only mtan 2>&1 | grep -v amdgpu.ids
