#!/bin/bash
# PMC traffic passes only (own rocprofv3 runs, kernel-trace + one counter, eager launches): B=1 gate kernel and the 32-clip fused layer.
TAG=${1:-pmc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_pmc_$c -o pmc -- python $ROOT/tools/prof_sampler.py 1 60 f16_m64 > $OUT/${TAG}_pmc_$c.log 2>&1
done
python $ROOT/tools/rocprof_traffic.py $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE "TEpiGate" $OUT/${TAG}_gate_traffic.json "tools/prof_sampler.py 1 60 f16_m64 (eager launches)"
rm -rf $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE
bash $ROOT/tools/gpu_traffic_b32.sh $TAG
