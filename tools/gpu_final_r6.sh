#!/bin/bash
# Round-6 final measurement visit on the final tree: bash tools/gpu_final_r6.sh <tag> "<parts>"
#   tests   full GPU suite -> <tag>_pytest_gpu.txt, then profiles/b32_error_fit.json from its batch-of-32 lines (tools/b32_error_fit.py)
#   bench   python bench.py (the driver's default line) -> <tag>_bench.json
#   prof1 / prof32   rocprofv3 --kernel-trace --stats of the single-clip / 32-clip bench command -> <tag>_kernel_stats[_b32].csv
#   profvoc / proftrain   the same for the vocoder alone (one clip) and for bench.py --train
#   pmc     PMC traffic passes (tools/gpu_pmc_r5.sh): gate / res-skip kernels at B = 1, fused layer at 32 clips, the training step
#   trainroof  profiles/train_kernels.json from <tag>_kernel_stats_train.csv + <tag>_train_traffic.json (tools/train_roofline.py; after proftrain and pmc)
#   ablate  the profiling-build ablations of the fused layer kernel (tools/gpu_r6_ablate.py) -> <tag>_layer_ablations.txt
#   job     python bench.py --gpus 1 --job-clips 256 [--clips-per-batch 32|36] -> <tag>_job256*.json
TAG=${1:-r6Z}
PARTS=${2:-"tests prof1 prof32 profvoc vocroof proftrain pmc trainroof bench"}
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
  [ -n "$F" ] && head -60 "$F" > $OUT/${TAG}_kernel_stats_train.csv && head -8 $OUT/${TAG}_kernel_stats_train.csv | cut -c1-200
  cut -c1-300 $OUT/${TAG}_train_bench.json; rm -rf $OUT/${TAG}_proft ;;
pmc)
  bash $ROOT/tools/gpu_pmc_r5.sh $TAG
  for f in gate_traffic resskip_traffic layer_traffic_b32 train_traffic; do [ -s $OUT/${TAG}_$f.json ] && cp $OUT/${TAG}_$f.json $ROOT/profiles/$f.json; done
  ls $OUT | grep ${TAG}_.*traffic ;;
vocroof)
  python tools/voc_roofline.py $OUT/${TAG}_kernel_stats_voc.csv 21 > $OUT/${TAG}_voc_roofline.txt 2>&1 && cp $ROOT/profiles/voc_kernels.json $OUT/${TAG}_voc_kernels.json
  cat $OUT/${TAG}_voc_roofline.txt ;;
trainroof)
  cp $OUT/${TAG}_train_traffic.json $ROOT/profiles/train_traffic.json
  python tools/train_roofline.py $OUT/${TAG}_kernel_stats_train.csv 7 > $OUT/${TAG}_train_roofline.txt 2>&1 && cp $ROOT/profiles/train_kernels.json $OUT/${TAG}_train_kernels.json
  head -12 $OUT/${TAG}_train_roofline.txt ;;
ablate)
  timeout 300 python tools/gpu_r6_ablate.py 32 > $OUT/${TAG}_layer_ablations.txt 2>&1; grep abl= $OUT/${TAG}_layer_ablations.txt ;;
job)
  timeout 300 python bench.py --gpus 1 --job-clips 256 > $OUT/${TAG}_job256.json 2>/dev/null; cut -c1-300 $OUT/${TAG}_job256.json
  timeout 300 python bench.py --gpus 1 --job-clips 252 --clips-per-batch 36 > $OUT/${TAG}_job252_cpb36.json 2>/dev/null; cut -c1-300 $OUT/${TAG}_job252_cpb36.json ;;
esac
done
