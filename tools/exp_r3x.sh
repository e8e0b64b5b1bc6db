#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3x; mkdir -p $O
cd $R
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q > $O/test.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 $O/test.log
[ $rc -eq 0 ] || exit 1
bash tools/exp_r3v.sh
