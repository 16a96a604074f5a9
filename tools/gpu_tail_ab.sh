#!/bin/bash
# A/B: single fp16 plane (DSVC_TAIL_HI=1) vs hi|lo planes for the skip sum / relu(skip proj) operands of the two tail projections
TAG=${1:-tailab}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
R=$OUT/${TAG}_ab.txt; : > $R
for v in 0 1; do
  echo "== DSVC_TAIL_HI=$v" >> $R
  DSVC_TAIL_HI=$v timeout 900 python -m pytest tests/test_gpu_headline.py tests/test_gpu_diffnet.py -m gpu -q -rP 2>&1 | grep -E "passed|failed|^(headline|throughput tiling f16_d64, 1000|end to end|plms)" >> $R
  DSVC_TAIL_HI=$v timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('B=1 ms', d['ms_per_step'], 'batched s', d['batched']['s_per_batch'], 'x', d['batched']['value'], 'plms ms', d['plms_50']['ms_per_clip'])" >> $R
done
cat $R
