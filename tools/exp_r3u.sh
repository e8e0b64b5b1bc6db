#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3u; mkdir -p $O
cd $R
timeout -k 10 300 python tools/diag_tight.py > $O/diag_intree.txt 2>&1
echo "intree: $(grep -c '<<<<' $O/diag_intree.txt) flagged; $(grep 'enc_layers.3.dconv.double_conv.3.weight' $O/diag_intree.txt)"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/test_full.log 2>&1; echo "pytest rc=$?"; tail -3 $O/test_full.log
