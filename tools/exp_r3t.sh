#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3t; mkdir -p $O
cd $R
python -m pytest tests/test_kernels_gpu.py tests/test_mtan_gpu.py -m gpu -x -q -k "maxpool or bilinear or concat or mtan" > $O/test.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/test.log
[ $rc -eq 0 ] || { grep -E "Error|assert |FAILED|mismatch" $O/test.log | head -20; exit 1; }
python tools/step_table.py --model mtan --batch 16 --height 256 --width 256 --classes 14 --top 400 2>/dev/null | grep -E "bilinear|maxpool|concat2|^# [0-9]" | head -20
python bench.py --no-cpu-baseline --only-headline --model mtan --batch 16 --height 256 --width 256 --classes 14 > $O/m.json 2> $O/m.err
python3 -c "
import json
r=json.load(open('$O/m.json')); print(r['value'], r['ms_per_step'])"
