#!/bin/bash
# A/B of a tuning env knob on the sampler (graph replay):  bash tools/gpu_ab4.sh <tag> <ENVVAR> "<values>" "<batches>"
TAG=${1:-ab4}; VAR=${2:-DSVC_TG_DEEP_RING}; VALS=${3:-"1 0"}; BATCHES=${4:-"1 4"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
{
for rep in 1 2; do for b in $BATCHES; do for v in $VALS; do
  steps=300; [ $b -gt 8 ] && steps=60
  echo -n "$VAR=$v "; env $VAR=$v python tools/prof_sampler.py $b $steps f16_d64 graph 2>/dev/null | tail -1
done; done; done
} > $OUT/${TAG}_ab.txt 2>&1
cat $OUT/${TAG}_ab.txt
timeout 600 python -m pytest tests -m gpu -q -x > $OUT/${TAG}_pytest_gpu.txt 2>&1; tail -6 $OUT/${TAG}_pytest_gpu.txt
