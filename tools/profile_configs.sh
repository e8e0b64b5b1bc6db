#!/bin/bash
# Reproduces every file under profiles/ for one round (run on the GPU box from the repo root):
#   tools/profile_configs.sh r03 [config ...]        configs: basic basic_bs8 basic_256 csnet csnet_layer mtan (default: all)
# Per config: one rocprofv3 --kernel-trace --stats run (kernel_stats CSV + the bench line under the profiler) and
# separate --pmc passes (counter sets that fit the gfx950 PMC slots, MI355X_MICROARCH.md "rocprofv3 PMC slots";
# FETCH_SIZE and WRITE_SIZE cannot share a pass), summarised by tools/pmc_summary.py into ONE json per config.
# The program itself follows `--` (no env / bash -c hop: the profiler's library initialises the GPU before exec).
set -e
ROUND=${1:-r03}; shift || true
CONFIGS=${@:-basic basic_bs8 basic_256 csnet csnet_layer mtan}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$ROUND
mkdir -p $OUT $ROOT/profiles
cd /tmp && export TMPDIR=/tmp
args_of() {
  case $1 in
    basic)     echo "--model basic --batch 32 --height 128 --width 256 --classes 19" ;;
    basic_bs8) echo "--model basic --batch 8 --height 128 --width 256 --classes 19" ;;
    basic_256) echo "--model basic --batch 32 --height 256 --width 256 --classes 19" ;;
    csnet)     echo "--model csnet --stitch channel --batch 32 --height 128 --width 256 --classes 19" ;;
    csnet_layer) echo "--model csnet --stitch layer --batch 32 --height 128 --width 256 --classes 19" ;;
    mtan)      echo "--model mtan --batch 16 --height 256 --width 256 --classes 14" ;;
  esac
}
for cfg in $CONFIGS; do
  A=$(args_of $cfg)
  mkdir -p $OUT/$cfg
  echo "== $cfg: kernel trace" >&2
  VMTL_SIDE_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$cfg/trace -- \
    python3 $ROOT/bench.py $A --steps 5 --warmup 2 --no-cpu-baseline --only-headline \
    > $OUT/$cfg/bench_line_under_rocprof.json 2> $OUT/$cfg/trace.err
  cp $OUT/$cfg/trace/*/*_kernel_stats.csv $ROOT/profiles/${ROUND}_${cfg}_kernel_stats_side_stream_off.csv
  cp $OUT/$cfg/bench_line_under_rocprof.json $ROOT/profiles/${ROUND}_${cfg}_bench_line_under_rocprof.json
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    echo "== $cfg: pmc pass $i ($set)" >&2
    VMTL_SIDE_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$cfg/pmc$i -- \
      python3 $ROOT/bench.py $A --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph --only-headline \
      > $OUT/$cfg/pmc$i.json 2> $OUT/$cfg/pmc$i.err
  done
  python3 $ROOT/tools/pmc_summary.py $OUT/$cfg "$A" $ROOT/profiles/${ROUND}_${cfg}_kernel_stats_side_stream_off.csv \
    $ROOT/profiles/${ROUND}_${cfg}_bench_line_under_rocprof.json > $ROOT/profiles/${ROUND}_${cfg}_pmc.json
  rm -rf $OUT/$cfg/trace $OUT/$cfg/pmc[0-9]  # raw per-dispatch CSVs: tens of MB per config (gpurun returns <= 64 MiB)
  echo "== $cfg done" >&2
done
# gpurun only brings gpurun_out/ back: keep a copy of the summaries there
mkdir -p $ROOT/gpurun_out/profiles_$ROUND && cp $ROOT/profiles/${ROUND}_*.* $ROOT/gpurun_out/profiles_$ROUND/
