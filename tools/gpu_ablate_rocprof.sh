#!/bin/bash
# per-kernel GPU durations (rocprofv3 kernel trace) under the DSVC_TG_DEBUG ablation knobs.
# bash tools/gpu_ablate_rocprof.sh <tag> <B> <steps> <precision> "<dbg values>"
TAG=$1; B=$2; N=$3; PREC=$4; DBGS=${5:-"0 1 2 3 4 7 8 15"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for d in $DBGS; do
  DSVC_TG_DEBUG=$d timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_abl_$d -o kt -- python $ROOT/tools/prof_sampler.py $B $N $PREC graph > $OUT/${TAG}_abl_$d.log 2>&1
  echo "== dbg=$d $(tail -1 $OUT/${TAG}_abl_$d.log)"
  python $ROOT/tools/rocprof_stats.py $OUT/${TAG}_abl_$d | grep -E "tgemm|k_add|kernel " | head -7
  rm -rf $OUT/${TAG}_abl_$d
done
