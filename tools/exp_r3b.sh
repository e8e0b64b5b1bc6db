#!/bin/bash
# round-3 experiment batch B: tests; split-K threshold A/B; MTAN graph regression hunt (r02 worktree vs now)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3b; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/test.log 2>&1; echo "pytest rc=$?" | tee -a $O/test.log
for kb in 256 512; do
  VMTL_KSPLIT_BLOCKS=$kb python bench.py --no-cpu-baseline > $O/bench_kb$kb.json 2> $O/bench_kb$kb.err; echo "bench kb$kb rc=$?"
done
MT="--model mtan --batch 16 --height 256 --width 256 --classes 14 --only-headline --no-cpu-baseline --no-roofline"
(cd _r02 && python bench.py $MT > $O/mtan_r02.json 2> $O/mtan_r02.err; echo "r02 mtan rc=$?")
(cd _r02 && python bench.py $MT --no-graph > $O/mtan_r02_eager.json 2> $O/mtan_r02_eager.err; echo "r02 mtan eager rc=$?")
VMTL_STAMPS=1 python bench.py $MT > $O/mtan_stamps.json 2> $O/mtan_stamps.err
echo "all done"
