#!/bin/bash
# Round 5, weight gradients from the frame-major planes (wgrad.h: wgrad_fm_kernel): bash tools/gpu_wgrad_fm.sh <tag> "<parts>"
#   micro   tools/micro/wgrad_fm_test.hip: the transposing LDS read's lane mapping, the kernel against float64, its time beside wgrad_nt_kernel
#   tests   tests/test_gpu_train.py (every gradient golden)
#   bench   python bench.py --train, then rocprofv3 kernel stats of the same command
#   pmc     whole-step HBM traffic by PMC -> <tag>_train_traffic.json
TAG=${1:-r5w}
PARTS=${2:-"micro tests bench"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
for part in $PARTS; do
cd $ROOT
case $part in
micro)
  cd $ROOT/tools/micro
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o /tmp/wgrad_fm_test wgrad_fm_test.hip ../../diff-svc_amd/csrc/common.hip 2>&1 | grep -i "error" | head
  timeout 300 /tmp/wgrad_fm_test > $OUT/${TAG}_wgrad_fm_micro.txt 2>&1; echo "micro rc=$?"; cat $OUT/${TAG}_wgrad_fm_micro.txt ;;
tests)
  timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -rP -x > $OUT/${TAG}_pytest_train.txt 2>&1
  echo "pytest rc=$?"; tail -3 $OUT/${TAG}_pytest_train.txt; grep -E "^(train|optimizer)" $OUT/${TAG}_pytest_train.txt | tail -30 ;;
bench)
  timeout 600 python bench.py --train --steps 10 --warmup 3 > $OUT/${TAG}_train_bench.json 2> $OUT/${TAG}_train_bench.err
  echo "bench rc=$?"; cut -c1-330 $OUT/${TAG}_train_bench.json; tail -2 $OUT/${TAG}_train_bench.err
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_proft -o tr -- python $ROOT/bench.py --train --steps 5 --warmup 2 > /dev/null 2> $OUT/${TAG}_proft.err
  F=$(find $OUT/${TAG}_proft -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && head -30 "$F" > $OUT/${TAG}_kernel_stats_train.csv && head -12 $OUT/${TAG}_kernel_stats_train.csv | cut -c1-200
  rm -rf $OUT/${TAG}_proft ;;
pmc)
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_pmct_$c -o pmc -- python $ROOT/bench.py --train --steps 3 --warmup 1 > $OUT/${TAG}_pmct_$c.log 2>&1
  done
  python $ROOT/tools/rocprof_traffic.py $OUT/${TAG}_pmct_FETCH_SIZE $OUT/${TAG}_pmct_WRITE_SIZE "@step" $OUT/${TAG}_train_traffic.json "bench.py --train --steps 3 --warmup 1" 4
  rm -rf $OUT/${TAG}_pmct_FETCH_SIZE $OUT/${TAG}_pmct_WRITE_SIZE ;;
esac
done
