#!/bin/bash
# HBM traffic of the fused residual-layer kernel (and, with DSVC_NO_FUSED_LAYER=1, of the two kernels it replaces) in the 32-clip tiling (separate rocprofv3 --pmc passes, eager launches).
# bash tools/gpu_traffic_b32.sh <tag>
TAG=${1:-t32}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_pmc_$c -o pmc -- python $ROOT/tools/prof_sampler.py 32 4 f16_m64 > $OUT/${TAG}_pmc_$c.log 2>&1
done
python $ROOT/tools/rocprof_traffic.py $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE "tlayer_kernel" $OUT/${TAG}_layer_traffic_b32.json "tools/prof_sampler.py 32 4 f16_m64 (eager launches)"
[ -z "$DSVC_NO_FUSED_LAYER" ] || python $ROOT/tools/rocprof_traffic.py $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE "8, 1, dsvc::TEpiGate" $OUT/${TAG}_gate_traffic_b32.json "tools/prof_sampler.py 32 4 f16_m64 (eager launches)"
[ -z "$DSVC_NO_FUSED_LAYER" ] || python $ROOT/tools/rocprof_traffic.py $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE "8, 1, dsvc::TEpiResSkip" $OUT/${TAG}_out_traffic_b32.json "tools/prof_sampler.py 32 4 f16_m64 (eager launches)"
rm -rf $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE
