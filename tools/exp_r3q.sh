#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3q; mkdir -p $O
cd $R
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "up2 or conv2d" 2>&1 | tail -2
for v in base new base new; do
  L=""; [ $v = base ] && L=$R/vision_mtl_amd/csrc/libvmtl_base.so
  VMTL_LIB=$L python tools/bench_up2.py > $O/up2_$v.log 2>&1; echo "== $v"; grep -v amdgpu $O/up2_$v.log | tail -8
done
for v in base new; do
  L=""; [ $v = base ] && L=$R/vision_mtl_amd/csrc/libvmtl_base.so
  VMTL_LIB=$L python tools/bench_conv.py --only blk --reps 5 > $O/conv_$v.log 2>&1; echo "$v: $(tail -1 $O/conv_$v.log)"
done
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python3 -c "
import json
d=json.load(open('$O/bench.json'))
for r in [d]+d['configs']:
    c=r['config']; rf=r.get('roofline',{})
    print(c['workload'][:66], r['value'], r['ms_per_step'], c.get('ms_per_step_eager'), rf.get('achieved'), rf.get('wgrad_kernel',{}).get('achieved'))"
