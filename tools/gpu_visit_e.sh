#!/bin/bash
# training-step first contact: its parity tests, then the timing
TAG=${1:-r2e}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_vocoder.py -m gpu -q -rP > $OUT/${TAG}_pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/${TAG}_pytest_gpu.txt
grep -E "passed|failed|error" $OUT/${TAG}_pytest_gpu.txt | tail -3
grep -E "^(train step|optimizer|hifigan)" $OUT/${TAG}_pytest_gpu.txt
grep -E "^(FAILED|ERROR)|Error|error" $OUT/${TAG}_pytest_gpu.txt | head -20
timeout 600 python bench.py --train --steps 5 --warmup 2 > $OUT/${TAG}_train_bench.json 2> $OUT/${TAG}_train_bench.err
echo "train bench rc=$?"; cat $OUT/${TAG}_train_bench.json; tail -5 $OUT/${TAG}_train_bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_proft -o s -- python $ROOT/bench.py --train --steps 3 --warmup 1 > $OUT/${TAG}_proft.log 2>&1
F=$(find $OUT/${TAG}_proft -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && head -16 "$F" > $OUT/${TAG}_kernel_stats_train.csv && head -14 $OUT/${TAG}_kernel_stats_train.csv
rm -rf $OUT/${TAG}_proft
