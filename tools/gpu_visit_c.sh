#!/bin/bash
# A/B: 64-frame x 4-wave tiling (2 workgroups per CU) vs the 128-frame tiling, unfused and fused; new PLMS / taps / clip-id tests
TAG=${1:-r2c}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -rP -k "plms or sampler_vs_reference or taps or clip_ids or ragged or seam" > $OUT/${TAG}_pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/${TAG}_pytest_gpu.txt
grep -E "passed|failed|error" $OUT/${TAG}_pytest_gpu.txt | tail -3
grep -E "^(sampler golden|PLMS|tgemm taps)" $OUT/${TAG}_pytest_gpu.txt
grep -E "^(FAILED|ERROR)" $OUT/${TAG}_pytest_gpu.txt | head
{
for e in "X=1" "DSVC_NO_FUSED_LAYER=1" "DSVC_NO_FUSED_LAYER=1 DSVC_TG_TILE64=1"; do
  echo "== $e"; env $e timeout 300 python tools/prof_sampler.py 32 64 f16_d64 graph | tail -1
done
} > $OUT/${TAG}_ab.txt 2>&1
cat $OUT/${TAG}_ab.txt
cd /tmp
DSVC_NO_FUSED_LAYER=1 DSVC_TG_TILE64=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof64 -o s -- python $ROOT/tools/prof_sampler.py 32 30 f16_d64 graph > $OUT/${TAG}_prof64.log 2>&1
F=$(find $OUT/${TAG}_prof64 -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && head -8 "$F" > $OUT/${TAG}_kernel_stats_tile64.csv && head -6 $OUT/${TAG}_kernel_stats_tile64.csv
rm -rf $OUT/${TAG}_prof64
