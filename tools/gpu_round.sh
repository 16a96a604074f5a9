#!/bin/bash
# One GPU-box visit: parity tests, the bench line, and a rocprofv3 kernel trace of the bench command.
# Usage (from the repo root on the box):  bash tools/gpu_round.sh [tag]
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/${TAG}_pytest_gpu.txt
tail -5 $OUT/${TAG}_pytest_gpu.txt
timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench rc=$?"; cat $OUT/${TAG}_bench.json; tail -3 $OUT/${TAG}_bench.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python $ROOT/bench.py --steps 2 --warmup 1 --no-batched --no-cpu-baseline > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.err
echo "rocprof rc=$?"
find $OUT/${TAG}_prof -name "*stats*" | head
F=$(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && head -30 "$F" > $OUT/${TAG}_kernel_stats.csv && cat $OUT/${TAG}_kernel_stats.csv
# PMC passes (own runs, no other trace domains) for the gate kernel's HBM traffic
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_pmc_$c -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --no-batched --no-cpu-baseline --no-graph > $OUT/${TAG}_pmc_$c.log 2>&1
done
python $ROOT/tools/rocprof_traffic.py $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE "TEpiGate" $OUT/${TAG}_gate_traffic.json
rm -rf $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE
# keep the pulled directory small: the raw trace is not needed
find $OUT/${TAG}_prof -name "*kernel_trace.csv" -size +20M -delete
