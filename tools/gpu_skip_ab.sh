#!/bin/bash
# Round 6 (third session): same-box A/B of the padded-tile skip on the two-launch tilings (TGemmArgs::skip_rowclip).
#   ab/libA.so = HEAD without it, ab/libB.so = with it.  bash tools/gpu_skip_ab.sh <tag>
#   1. the single-clip headline (the kernels it touches): ms per DDPM step by graph replay, three rounds each
#   2. ragged groups through SvcPipeline.infer_chunks on both libraries: PLMS-50 on the seven chunks, DDPM on small groups
TAG=${1:-r6ab}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
bash tools/gpu_lib_ab.sh ${TAG}_b1 "python tools/prof_sampler.py 1 1000 f16_x3t graph"
bash tools/gpu_lib_ab.sh ${TAG}_b3 "python tools/prof_sampler.py 3 500 f16_x3t graph"
cp diff-svc_amd/libdsvc_hip.so /tmp/lib_orig.so
for v in A B; do
  cp ab/lib$v.so diff-svc_amd/libdsvc_hip.so
  {
  echo "=== lib$v: PLMS-50 430,700,861,1200,1600,2100,2600"; timeout 300 python tools/gpu_chunks_probe.py 1000 430,700,861,1200,1600,2100,2600 20 2>&1 | tail -8
  echo "=== lib$v: PLMS-50 200,350,500,640,861"; timeout 300 python tools/gpu_chunks_probe.py 1000 200,350,500,640,861 20 2>&1 | tail -6
  echo "=== lib$v: DDPM 200,350,500,640,861"; timeout 300 python tools/gpu_chunks_probe.py 1000 200,350,500,640,861 2>&1 | tail -6
  echo "=== lib$v: DDPM 300,861,1500"; timeout 300 python tools/gpu_chunks_probe.py 1000 300,861,1500 2>&1 | tail -6
  } >> $OUT/${TAG}_chunks.txt
done
cp /tmp/lib_orig.so diff-svc_amd/libdsvc_hip.so
cat $OUT/${TAG}_chunks.txt
