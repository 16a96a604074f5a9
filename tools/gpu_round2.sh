#!/bin/bash
# Round-2 GPU visit: parity tests (all, no -x), the bench line, rocprofv3 kernel stats for the single-clip AND the 32-clip bench
# command, PMC traffic passes for both tilings.  bash tools/gpu_round2.sh <tag> [skip-pmc]
TAG=${1:-r2a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rP --durations=15 > $OUT/${TAG}_pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/${TAG}_pytest_gpu.txt
grep -E "passed|failed|error" $OUT/${TAG}_pytest_gpu.txt | tail -5
grep -E "^(headline|throughput tiling|tgemm taps|end to end)" $OUT/${TAG}_pytest_gpu.txt
timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench rc=$?"; cat $OUT/${TAG}_bench.json; tail -3 $OUT/${TAG}_bench.err
cd /tmp
# kernel stats of the single-clip bench command (BASELINE configs[1]) ...
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python $ROOT/bench.py --steps 2 --warmup 1 --no-batched --no-cpu-baseline > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.err
echo "rocprof B1 rc=$?"
F=$(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && head -30 "$F" > $OUT/${TAG}_kernel_stats.csv && head -8 $OUT/${TAG}_kernel_stats.csv
rm -rf $OUT/${TAG}_prof
# ... and of the 32-clip batch (the per-GPU share of configs[3]): same bench.py, --clips-per-gpu 32
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof32 -o bench -- python $ROOT/bench.py --clips-per-gpu 32 --steps 1 --warmup 1 --no-batched --no-cpu-baseline > $OUT/${TAG}_prof32_bench.json 2> $OUT/${TAG}_prof32.err
echo "rocprof B32 rc=$?"
F=$(find $OUT/${TAG}_prof32 -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && head -30 "$F" > $OUT/${TAG}_kernel_stats_b32.csv && head -8 $OUT/${TAG}_kernel_stats_b32.csv
rm -rf $OUT/${TAG}_prof32
cat $OUT/${TAG}_prof32_bench.json
[ "$2" = "skip-pmc" ] && exit 0
# PMC passes (own runs, kernel-trace only, eager launches): single clip ...
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_pmc_$c -o pmc -- python $ROOT/tools/prof_sampler.py 1 60 f16_m64 > $OUT/${TAG}_pmc_$c.log 2>&1
done
python $ROOT/tools/rocprof_traffic.py $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE "TEpiGate" $OUT/${TAG}_gate_traffic.json "tools/prof_sampler.py 1 60 f16_m64 (eager launches)"
rm -rf $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE
# ... and the 32-clip tiling
bash $ROOT/tools/gpu_traffic_b32.sh $TAG
