#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3s; mkdir -p $O
cd $R
python tools/step_table.py --model mtan --batch 16 --height 256 --width 256 --classes 14 --top 120 > $O/mtan.txt 2> $O/mtan.err && \
python tools/step_table.py --model basic --batch 8 --top 80 > $O/bs8.txt 2> $O/bs8.err && \
python tools/step_table.py --model basic --batch 32 --top 80 > $O/basic.txt 2> $O/basic.err && \
python tools/step_table.py --model csnet --batch 32 --top 80 > $O/csnet.txt 2> $O/csnet.err
echo rc=$?
head -3 $O/mtan.txt
