#!/bin/bash
# same-box A/B of prebuilt libraries (ab/libA.so, ab/libB.so; env LIBS="A C D" for others) on a command that prints a timing line:
#   bash tools/gpu_lib_ab.sh <tag> "<python command>"      e.g. "python tools/prof_sampler.py 32 60 f16_w2 graph"
TAG=${1:-ab}; CMD=${2:-"python tools/prof_sampler.py 32 60 f16_w2 graph"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
cp diff-svc_amd/libdsvc_hip.so /tmp/lib_orig.so
for round in 1 2 3; do
  for v in ${LIBS:-A B}; do
    cp ab/lib$v.so diff-svc_amd/libdsvc_hip.so
    echo "round $round lib$v: $(timeout 300 $CMD 2>/dev/null | tail -1)" | tee -a $OUT/${TAG}_lib_ab.txt
  done
done
cp /tmp/lib_orig.so diff-svc_amd/libdsvc_hip.so
