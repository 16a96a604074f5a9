#!/bin/bash
TAG=${1:-ab2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
{
for st in 0 1; do
  echo -n "B=32 d64 stream=$st: "; DSVC_TG_STREAM=$st python tools/prof_sampler.py 32 40 f16_d64 graph | tail -1
done
} > $OUT/${TAG}_ab.txt 2>&1
cat $OUT/${TAG}_ab.txt
cd /tmp; export TMPDIR=/tmp
for st in 0 1; do
DSVC_TG_STREAM=$st timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_kt$st -o kt -- python $ROOT/tools/prof_sampler.py 32 20 f16_d64 graph > $OUT/${TAG}_kt$st.log 2>&1
python $ROOT/tools/rocprof_stats.py $OUT/${TAG}_kt$st gaps > $OUT/${TAG}_kt$st.txt 2>&1
rm -rf $OUT/${TAG}_kt$st
echo "== stream=$st"; grep -E "tgemm|gap|->" $OUT/${TAG}_kt$st.txt | head -16
done
