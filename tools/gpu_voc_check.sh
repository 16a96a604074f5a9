#!/bin/bash
# vocoder parity tests + timing + rocprofv3 kernel stats (32 clips and 1 clip)
TAG=${1:-voc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_vocoder.py tests/test_gpu_pipeline.py -m gpu -q > $OUT/${TAG}_pytest.txt 2>&1
tail -3 $OUT/${TAG}_pytest.txt
python tools/prof_vocoder.py 32 3; python tools/prof_vocoder.py 1 10
cd /tmp
for B in 32 1; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof$B -o voc -- python $ROOT/tools/prof_vocoder.py $B 3 > /dev/null 2>&1
F=$(find $OUT/${TAG}_prof$B -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && head -16 "$F" > $OUT/${TAG}_voc${B}_stats.csv
rm -rf $OUT/${TAG}_prof$B
done
