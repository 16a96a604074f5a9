#!/bin/bash
# same-box A/B of two prebuilt libraries (ab/libA.so, ab/libB.so) on the training step: bash tools/gpu_train_ab.sh <tag>
TAG=${1:-ab}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
cp diff-svc_amd/libdsvc_hip.so /tmp/lib_orig.so
for round in 1 2 3; do
  for v in A B; do
    cp ab/lib$v.so diff-svc_amd/libdsvc_hip.so
    ms=$(timeout 300 python bench.py --train --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "round $round lib$v: $ms ms per step" | tee -a $OUT/${TAG}_train_ab.txt
  done
done
cp /tmp/lib_orig.so diff-svc_amd/libdsvc_hip.so
