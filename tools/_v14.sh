cd $GRAFT_REPO_ROOT
cp diff-svc_amd/libdsvc_hip.so /tmp/lib_orig.so
for v in N R N R; do
  cp ab/lib$v.so diff-svc_amd/libdsvc_hip.so
  for T in 1200 1600; do echo "lib$v: $(python tools/prof_sampler.py 1 300 f16_x3t graph $T 2>/dev/null | tail -1)"; done
  echo "lib$v: $(python tools/gpu_chunks_probe.py 1000 430,700,861,1200,1600,2100,2600 2>/dev/null | grep 'one by one')"
done > gpurun_out/r6au_check.txt
cp /tmp/lib_orig.so diff-svc_amd/libdsvc_hip.so
cat gpurun_out/r6au_check.txt
