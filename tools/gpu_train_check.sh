#!/bin/bash
# training step: parity tests + bench line + rocprofv3 kernel stats
TAG=${1:-train}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -rP > $OUT/${TAG}_pytest.txt 2>&1
tail -2 $OUT/${TAG}_pytest.txt; grep -E "^(train|optimizer)" $OUT/${TAG}_pytest.txt
timeout 600 python bench.py --train --steps 5 --warmup 2 > $OUT/${TAG}_train_bench.json 2> $OUT/${TAG}_train_bench.err
cat $OUT/${TAG}_train_bench.json | cut -c1-400
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o tr -- python $ROOT/bench.py --train --steps 3 --warmup 1 > /dev/null 2>&1
F=$(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && head -24 "$F" > $OUT/${TAG}_kernel_stats_train.csv
rm -rf $OUT/${TAG}_prof
