#!/bin/bash
# Round-5 PMC traffic passes (own rocprofv3 runs: kernel-trace + ONE counter, eager launches -- --pmc on hipGraph replays segfaults on this stack):
#   B = 1 at f16_x3t: BOTH kernels of the two-launch layer (gate, res/skip)   -> gate_traffic.json, resskip_traffic.json
#   B = 32 at f16_w6: the fused layer kernel                                   -> layer_traffic_b32.json
# bash tools/gpu_pmc_r5.sh <tag>
TAG=${1:-r5pmc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_pmc1_$c -o pmc -- python $ROOT/tools/prof_sampler.py 1 60 f16_x3t > $OUT/${TAG}_pmc1_$c.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_pmc32_$c -o pmc -- python $ROOT/tools/prof_sampler.py 32 4 f16_w6 > $OUT/${TAG}_pmc32_$c.log 2>&1
done
python $ROOT/tools/rocprof_traffic.py $OUT/${TAG}_pmc1_FETCH_SIZE $OUT/${TAG}_pmc1_WRITE_SIZE "TEpiGate" $OUT/${TAG}_gate_traffic.json "tools/prof_sampler.py 1 60 f16_x3t (eager launches)"
python $ROOT/tools/rocprof_traffic.py $OUT/${TAG}_pmc1_FETCH_SIZE $OUT/${TAG}_pmc1_WRITE_SIZE "TEpiResSkip" $OUT/${TAG}_resskip_traffic.json "tools/prof_sampler.py 1 60 f16_x3t (eager launches)"
python $ROOT/tools/rocprof_traffic.py $OUT/${TAG}_pmc32_FETCH_SIZE $OUT/${TAG}_pmc32_WRITE_SIZE "tlayer_kernel" $OUT/${TAG}_layer_traffic_b32.json "tools/prof_sampler.py 32 4 f16_w6 (eager launches)"
rm -rf $OUT/${TAG}_pmc1_FETCH_SIZE $OUT/${TAG}_pmc1_WRITE_SIZE $OUT/${TAG}_pmc32_FETCH_SIZE $OUT/${TAG}_pmc32_WRITE_SIZE
# the training step (BASELINE configs[4]): whole-step HBM traffic, every per-step kernel summed                 -> train_traffic.json
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_pmct_$c -o pmc -- python $ROOT/bench.py --train --steps 3 --warmup 1 > $OUT/${TAG}_pmct_$c.log 2>&1
done
python $ROOT/tools/rocprof_traffic.py $OUT/${TAG}_pmct_FETCH_SIZE $OUT/${TAG}_pmct_WRITE_SIZE "@step" $OUT/${TAG}_train_traffic.json "bench.py --train --steps 3 --warmup 1" 4
rm -rf $OUT/${TAG}_pmct_FETCH_SIZE $OUT/${TAG}_pmct_WRITE_SIZE
