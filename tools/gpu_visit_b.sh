#!/bin/bash
# first contact of the fused layer kernel: parity tests, position diagnostic, A/B timings, kernel stats, bench line
TAG=${1:-r2b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -rP --durations=8 > $OUT/${TAG}_pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/${TAG}_pytest_gpu.txt
grep -E "passed|failed|error" $OUT/${TAG}_pytest_gpu.txt | tail -3
grep -E "^(headline|throughput tiling|tgemm taps|end to end|fused)" $OUT/${TAG}_pytest_gpu.txt
grep -E "^(FAILED|ERROR)" $OUT/${TAG}_pytest_gpu.txt | head
timeout 300 python tools/diag_position.py > $OUT/${TAG}_diag_position.txt 2>&1; cat $OUT/${TAG}_diag_position.txt | tail -12
{
for e in "X=1" "DSVC_NO_FUSED_LAYER=1" "DSVC_FUSED_PF=1"; do
  echo "== $e"; env $e timeout 300 python tools/prof_sampler.py 32 64 f16_d64 graph | tail -1
  env $e timeout 300 python tools/prof_sampler.py 32 64 f16_w2 graph | tail -1
done
} > $OUT/${TAG}_ab.txt 2>&1
cat $OUT/${TAG}_ab.txt
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof32 -o s -- python $ROOT/tools/prof_sampler.py 32 30 f16_d64 graph > $OUT/${TAG}_prof32.log 2>&1
F=$(find $OUT/${TAG}_prof32 -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && head -12 "$F" > $OUT/${TAG}_kernel_stats_b32.csv && head -6 $OUT/${TAG}_kernel_stats_b32.csv
rm -rf $OUT/${TAG}_prof32
cd $ROOT
timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench rc=$?"; cat $OUT/${TAG}_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['batched']['value'], d['batched']['roofline'], d['plms_50']['value'])"
