#!/bin/bash
# two-branch graph: equality test + A/B; trainer tests after the optimizer split
TAG=${1:-r2g}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_headline.py tests/test_gpu_train.py tests/test_gpu_diffnet.py -m gpu -q -rP -k "two_branch or train or optimizer or period_aligned or full_chain" > $OUT/${TAG}_pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/${TAG}_pytest_gpu.txt
grep -E "passed|failed|error" $OUT/${TAG}_pytest_gpu.txt | tail -3
grep -E "^(train|optimizer|throughput)" $OUT/${TAG}_pytest_gpu.txt
grep -E "^(FAILED|ERROR)|Error" $OUT/${TAG}_pytest_gpu.txt | head -20
{
for e in "DSVC_SPLIT2=1" "DSVC_SPLIT2=0"; do
  echo "== $e"; env $e timeout 300 python tools/prof_sampler.py 32 192 f16_d64 graph | tail -1
done
} > $OUT/${TAG}_ab.txt 2>&1
cat $OUT/${TAG}_ab.txt
