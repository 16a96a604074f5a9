#!/bin/bash
# A/B of vocoder conv tilings (DSVC_VOC_T64 / _T128 / _T256 knobs in vocoder.hip) at B=32 and B=1
TAG=${1:-vocab}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
R=$OUT/${TAG}_ab.txt; : > $R
run() { echo "== $1" >> $R; env $1 timeout 120 python tools/prof_vocoder.py 32 3 2>&1 | grep "ms per call" >> $R; env $1 timeout 120 python tools/prof_vocoder.py 1 10 2>&1 | grep "ms per call" >> $R; }
run "DSVC_X=0"
for v in 2 5 6 7; do run "DSVC_VOC_T64=$v"; done
for v in 2 4 5 6; do run "DSVC_VOC_T128=$v"; done
for v in 2 3 5 6 7; do run "DSVC_VOC_T256=$v"; done
run "DSVC_VOC_T64=2 DSVC_VOC_T128=2 DSVC_VOC_T256=2"
cat $R
