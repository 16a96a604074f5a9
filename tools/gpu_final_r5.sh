#!/bin/bash
# Round-5 final measurement visit on the final tree: bash tools/gpu_final_r5.sh <tag> "<parts>"
#   tests   full GPU suite -> <tag>_pytest_gpu.txt, then profiles/b32_error_fit.json from its batch-of-32 lines (tools/b32_error_fit.py)
#   bench   python bench.py (the driver's default line) -> <tag>_bench.json
#   prof1 / prof32   rocprofv3 --kernel-trace --stats of the single-clip / 32-clip bench command -> <tag>_kernel_stats[_b32].csv
#   profvoc / proftrain   the same for the vocoder alone (one clip) and for bench.py --train
#   pmc     PMC traffic passes (tools/gpu_pmc_r5.sh): gate / res-skip kernels at B = 1, fused layer at 32 clips, the training step
TAG=${1:-r5Z}
PARTS=${2:-"tests bench prof1 prof32 profvoc proftrain pmc"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
for part in $PARTS; do
cd $ROOT
case $part in
tests)
  timeout 2400 python -m pytest tests -m gpu -q -rP --durations=12 > $OUT/${TAG}_pytest_gpu.txt 2>&1
  echo "pytest rc=$?" >> $OUT/${TAG}_pytest_gpu.txt
  grep -E "passed|failed|error" $OUT/${TAG}_pytest_gpu.txt | tail -3
  python tools/b32_error_fit.py $OUT/${TAG}_pytest_gpu.txt && cp profiles/b32_error_fit.json $OUT/${TAG}_b32_error_fit.json ;;
bench)
  timeout 1200 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
  echo "bench rc=$?"; cut -c1-600 $OUT/${TAG}_bench.json; tail -2 $OUT/${TAG}_bench.err ;;
prof1)
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python $ROOT/bench.py --steps 2 --warmup 1 --no-batched --no-cpu-baseline > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.err
  F=$(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && head -30 "$F" > $OUT/${TAG}_kernel_stats.csv && head -6 $OUT/${TAG}_kernel_stats.csv | cut -c1-200
  rm -rf $OUT/${TAG}_prof ;;
prof32)
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof32 -o bench -- python $ROOT/bench.py --clips-per-gpu 32 --steps 1 --warmup 1 --no-batched --no-cpu-baseline > $OUT/${TAG}_prof32_bench.json 2> $OUT/${TAG}_prof32.err
  F=$(find $OUT/${TAG}_prof32 -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && head -30 "$F" > $OUT/${TAG}_kernel_stats_b32.csv && head -6 $OUT/${TAG}_kernel_stats_b32.csv | cut -c1-200
  rm -rf $OUT/${TAG}_prof32 ;;
profvoc)
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_profv -o voc -- python $ROOT/tools/prof_vocoder.py 1 20 > $OUT/${TAG}_voc_time.txt 2> $OUT/${TAG}_profv.err
  F=$(find $OUT/${TAG}_profv -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && head -30 "$F" > $OUT/${TAG}_kernel_stats_voc.csv && head -8 $OUT/${TAG}_kernel_stats_voc.csv | cut -c1-200
  cat $OUT/${TAG}_voc_time.txt; rm -rf $OUT/${TAG}_profv ;;
proftrain)
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_proft -o tr -- python $ROOT/bench.py --train --steps 5 --warmup 2 > $OUT/${TAG}_train_bench.json 2> $OUT/${TAG}_proft.err
  F=$(find $OUT/${TAG}_proft -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && head -30 "$F" > $OUT/${TAG}_kernel_stats_train.csv && head -8 $OUT/${TAG}_kernel_stats_train.csv | cut -c1-200
  cut -c1-300 $OUT/${TAG}_train_bench.json; rm -rf $OUT/${TAG}_proft ;;
pmc)
  bash $ROOT/tools/gpu_pmc_r5.sh $TAG
  ls $OUT | grep ${TAG}_.*traffic ;;
esac
done
