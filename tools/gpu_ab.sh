#!/bin/bash
# A/B timings of tuning knobs (sampler-only, graph replay).  bash tools/gpu_ab.sh <tag>
TAG=${1:-ab}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
{
for pf in 0 64 128 256; do
  echo -n "B=1 d64 prefetch_wgs=$pf: "; DSVC_TG_PREFETCH_WGS=$pf python tools/prof_sampler.py 1 300 f16_d64 graph | tail -1
done
for w in 0 1; do
  echo -n "B=32 d64 waves12=$w: "; DSVC_TG_WAVES12=$w python tools/prof_sampler.py 32 40 f16_d64 graph | tail -1
done
echo -n "B=8 d64: "; python tools/prof_sampler.py 8 60 f16_d64 graph | tail -1
echo -n "B=8 d64 waves12: "; DSVC_TG_WAVES12=1 python tools/prof_sampler.py 8 60 f16_d64 graph | tail -1
} > $OUT/${TAG}_ab.txt 2>&1
cat $OUT/${TAG}_ab.txt
timeout 600 python -m pytest tests -m gpu -q -x > $OUT/${TAG}_pytest_gpu.txt 2>&1; tail -4 $OUT/${TAG}_pytest_gpu.txt
DSVC_TG_WAVES12=1 timeout 600 python -m pytest tests/test_gpu_diffnet.py -m gpu -q -x -k "batched or full_size" > $OUT/${TAG}_pytest_w12.txt 2>&1; tail -3 $OUT/${TAG}_pytest_w12.txt
