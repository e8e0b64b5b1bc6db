#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3t; mkdir -p $O
cd $R
python -m pytest tests/test_kernels_gpu.py tests/test_basic_gpu.py tests/test_csnet_gpu.py tests/test_arena_gpu.py -m gpu -x -q > $O/test.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 $O/test.log
[ $rc -eq 0 ] || { grep -E "Error|assert |FAILED" $O/test.log | head; exit 1; }
python tools/step_table.py --model basic --batch 32 --top 400 2>/dev/null | grep -E "pack_weights_batch|^# [0-9]"
python tools/step_table.py --model csnet --batch 32 --top 600 2>/dev/null | grep -E "pack_weights_batch|^# [0-9]"
python bench.py --no-cpu-baseline --only-headline > $O/bench.json 2> $O/bench.err; python3 -c "
import json
r=json.load(open('$O/bench.json')); print(r['value'], r['ms_per_step'])"
