#!/bin/bash
# Round-5 GPU visit: bash tools/gpu_r5.sh <tag> "<parts>"   parts: mid midtests + every part of tools/gpu_r4.sh (tests bench prof1 prof32 pmc32 train ...)
TAG=${1:-r5a}
PARTS=${2:-"midtests mid"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
for part in $PARTS; do
case $part in
mid)
  timeout 900 python tools/gpu_mid_sweep.py ${MIDPREC:-f16_w6} > $OUT/${TAG}_mid_sweep.txt 2>&1; cat $OUT/${TAG}_mid_sweep.txt ;;
midtests)
  timeout 900 python -m pytest tests/test_gpu_headline.py -q -rP -k "mid_ or 24k_architecture" > $OUT/${TAG}_mid_tests.txt 2>&1
  grep -E "passed|failed|error|^mid batch|^fused layer kernel|^24 kHz" $OUT/${TAG}_mid_tests.txt ;;
*)
  bash tools/gpu_r4.sh $TAG "$part" ;;
esac
done
