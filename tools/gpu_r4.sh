#!/bin/bash
# Round-4 GPU visit: bash tools/gpu_r4.sh <tag> [parts]   parts: any of  overlap tests bench prof1 prof32 pmc32  (default: overlap tests bench prof32)
TAG=${1:-r3a}
PARTS=${2:-"overlap tests bench prof32"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
PREC32=${PREC32:-auto}
for part in $PARTS; do
case $part in
phase)
  for p in f16_w2 f16_m64; do timeout 300 python tools/gpu_phase_offset.py $p 128 >> $OUT/${TAG}_phase_offset.txt 2>&1; done; cat $OUT/${TAG}_phase_offset.txt ;;
tailcheck)
  timeout 300 python tools/gpu_tail_check.py f16_w2 > $OUT/${TAG}_tail_check.txt 2>&1; cat $OUT/${TAG}_tail_check.txt ;;
stamps2)
  for q in 1 2; do timeout 300 python tools/gpu_layer_stamps.py f16_w2 $q >> $OUT/${TAG}_layer_stamps.txt 2>&1; done; cat $OUT/${TAG}_layer_stamps.txt ;;
stamps)
  for p in f16_w2 f16_m64; do timeout 300 python tools/gpu_layer_stamps.py $p >> $OUT/${TAG}_layer_stamps.txt 2>&1; done; cat $OUT/${TAG}_layer_stamps.txt ;;
prio)
  for p in f16_w2 f16_m64; do timeout 300 python tools/gpu_layer_prio.py $p 128 >> $OUT/${TAG}_layer_prio.txt 2>&1; done; cat $OUT/${TAG}_layer_prio.txt ;;
x3t)
  timeout 600 python tools/gpu_x3t_time.py > $OUT/${TAG}_x3t_time.txt 2>&1; cat $OUT/${TAG}_x3t_time.txt
  timeout 900 python -m pytest tests/test_gpu_diffnet.py tests/test_gpu_headline.py -q -k "x3t" -rP > $OUT/${TAG}_x3t_tests.txt 2>&1; grep -E "passed|failed|^sampler golden|^PLMS|^end to end" $OUT/${TAG}_x3t_tests.txt ;;
bisect)
  for a in tiny 44k; do timeout 300 python tools/gpu_x3t_bisect.py $a >> $OUT/${TAG}_bisect.txt 2>&1; done; cat $OUT/${TAG}_bisect.txt ;;
defer)
  for p in f16_w2 f16_m64; do timeout 300 python tools/gpu_defer_ab.py $p 128 >> $OUT/${TAG}_defer_ab.txt 2>&1; done; cat $OUT/${TAG}_defer_ab.txt ;;
spread)
  timeout 900 python -m pytest tests/test_gpu_headline.py -q -s -k "spread" > $OUT/${TAG}_spread.txt 2>&1; grep -E "^spread|passed|failed" $OUT/${TAG}_spread.txt ;;
train)
  timeout 600 python -m pytest tests/test_gpu_train.py -q -rP > $OUT/${TAG}_train_tests.txt 2>&1; grep -E "passed|failed|^train|^optimizer" $OUT/${TAG}_train_tests.txt
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_proft -o bench -- python $ROOT/bench.py --train --steps 5 --warmup 2 > $OUT/${TAG}_train_bench.json 2> $OUT/${TAG}_train.err
  F=$(find $OUT/${TAG}_proft -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && head -40 "$F" > $OUT/${TAG}_kernel_stats_train.csv && cut -c1-200 $OUT/${TAG}_kernel_stats_train.csv | head -24
  rm -rf $OUT/${TAG}_proft; cat $OUT/${TAG}_train_bench.json; cd $ROOT ;;
sweep)
  timeout 600 python tools/gpu_auto_sweep.py > $OUT/${TAG}_auto_sweep.txt 2>&1; cat $OUT/${TAG}_auto_sweep.txt ;;
overlap)
  timeout 120 tools/micro/overlap 4 > $OUT/${TAG}_overlap.txt 2>&1; echo "overlap rc=$?"; cat $OUT/${TAG}_overlap.txt ;;
tests)
  DSVC_PARTIAL_GOLDENS=${DSVC_PARTIAL_GOLDENS:-0} timeout 2400 python -m pytest tests -m gpu -q -rP --durations=12 > $OUT/${TAG}_pytest_gpu.txt 2>&1
  echo "pytest rc=$?" >> $OUT/${TAG}_pytest_gpu.txt
  grep -E "passed|failed|error" $OUT/${TAG}_pytest_gpu.txt | tail -5
  grep -E "^(spread|batch of 32|headline|throughput tiling|end to end|train step|train traj|optimizer)" $OUT/${TAG}_pytest_gpu.txt ;;
bench)
  timeout 1200 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
  echo "bench rc=$?"; cat $OUT/${TAG}_bench.json; tail -3 $OUT/${TAG}_bench.err ;;
prof1)
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python $ROOT/bench.py --steps 2 --warmup 1 --no-batched --no-cpu-baseline > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.err
  echo "rocprof B1 rc=$?"
  F=$(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && head -30 "$F" > $OUT/${TAG}_kernel_stats.csv && head -8 $OUT/${TAG}_kernel_stats.csv
  rm -rf $OUT/${TAG}_prof; cd $ROOT ;;
prof32)
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof32 -o bench -- python $ROOT/bench.py --clips-per-gpu 32 --steps 1 --warmup 1 --no-batched --no-cpu-baseline --precision $PREC32 > $OUT/${TAG}_prof32_bench.json 2> $OUT/${TAG}_prof32.err
  echo "rocprof B32 rc=$?"
  F=$(find $OUT/${TAG}_prof32 -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && head -30 "$F" > $OUT/${TAG}_kernel_stats_b32.csv && head -8 $OUT/${TAG}_kernel_stats_b32.csv
  rm -rf $OUT/${TAG}_prof32; cat $OUT/${TAG}_prof32_bench.json; cd $ROOT ;;
pmc1)
  cd /tmp
  P=${PMCPREC1:-f16_x3t}
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_pmc1_$c -o pmc -- python $ROOT/tools/prof_sampler.py 1 60 $P > $OUT/${TAG}_pmc1_$c.log 2>&1
  done
  python $ROOT/tools/rocprof_traffic.py $OUT/${TAG}_pmc1_FETCH_SIZE $OUT/${TAG}_pmc1_WRITE_SIZE "TEpiGate" $OUT/${TAG}_gate_traffic.json "tools/prof_sampler.py 1 60 $P (eager launches)"
  rm -rf $OUT/${TAG}_pmc1_FETCH_SIZE $OUT/${TAG}_pmc1_WRITE_SIZE; cd $ROOT ;;
pmc32)
  cd /tmp
  P=${PMCPREC32:-f16_w6}
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_pmc32_$c -o pmc -- python $ROOT/tools/prof_sampler.py 32 12 $P > $OUT/${TAG}_pmc32_$c.log 2>&1
  done
  python $ROOT/tools/rocprof_traffic.py $OUT/${TAG}_pmc32_FETCH_SIZE $OUT/${TAG}_pmc32_WRITE_SIZE "tlayer_kernel" $OUT/${TAG}_layer_traffic_b32.json "tools/prof_sampler.py 32 12 $P (eager launches)"
  rm -rf $OUT/${TAG}_pmc32_FETCH_SIZE $OUT/${TAG}_pmc32_WRITE_SIZE; cd $ROOT ;;
pmctrain)
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_pmct_$c -o pmc -- python $ROOT/bench.py --train --steps 3 --warmup 1 > $OUT/${TAG}_pmct_$c.log 2>&1
  done
  for k in TEpiDxT TEpiGateT TEpiGateBwdT TEpiResSkipT wgrad_nt_kernel; do
    python $ROOT/tools/rocprof_traffic.py $OUT/${TAG}_pmct_FETCH_SIZE $OUT/${TAG}_pmct_WRITE_SIZE "$k" $OUT/${TAG}_train_traffic_$k.json "bench.py --train --steps 3 --warmup 1"
  done
  rm -rf $OUT/${TAG}_pmct_FETCH_SIZE $OUT/${TAG}_pmct_WRITE_SIZE; cd $ROOT ;;
esac
done
