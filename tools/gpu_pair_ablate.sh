#!/bin/bash
# k_pair_mfma phase ablation: per-kernel rocprofv3 averages at 32 clips with DSVC_PAIR_DBG = 0 (full), 1 (no MFMA loops), 2 (no epilogue), 3 (staging + LDS park only)
TAG=${1:-pair}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_vocoder.py tests/test_gpu_pipeline.py -m gpu -q > $OUT/${TAG}_pytest.txt 2>&1
tail -2 $OUT/${TAG}_pytest.txt
python tools/prof_vocoder.py 32 3; python tools/prof_vocoder.py 1 10
cd /tmp
R=$OUT/${TAG}_ablate.txt; : > $R
for D in 0 1 2 3; do
DSVC_PAIR_DBG=$D timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_p$D -o voc -- python $ROOT/tools/prof_vocoder.py 32 2 > /dev/null 2>&1
F=$(find $OUT/${TAG}_p$D -name "*kernel_stats.csv" | head -1)
echo "== DSVC_PAIR_DBG=$D" >> $R
[ -n "$F" ] && grep "k_pair_mfma" "$F" | awk -F, '{print $1, "avg_ns", $4}' >> $R
rm -rf $OUT/${TAG}_p$D
done
cat $R
