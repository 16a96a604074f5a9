#!/bin/bash
# kernel-trace + PMC passes over the sampler-only workload.  bash tools/gpu_prof.sh <tag> <precision>
TAG=${1:-p}; PREC=${2:-f16_d64}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for cfg in "1 200" "32 30"; do
  set -- $cfg; B=$1; N=$2
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_kt_B$B -o kt -- python $ROOT/tools/prof_sampler.py $B $N $PREC > $OUT/${TAG}_kt_B$B.log 2>&1
  python $ROOT/tools/rocprof_stats.py $OUT/${TAG}_kt_B$B > $OUT/${TAG}_kt_B$B.txt 2>&1
  find $OUT/${TAG}_kt_B$B -name "*.csv" -size +8M -delete
  cat $OUT/${TAG}_kt_B$B.log | tail -1; head -14 $OUT/${TAG}_kt_B$B.txt
done
# PMC passes (their own runs, kernel-trace only): B=32, few steps
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/${TAG}_pmc$i -o pmc -- python $ROOT/tools/prof_sampler.py 32 6 $PREC > $OUT/${TAG}_pmc$i.log 2>&1
  python $ROOT/tools/rocprof_stats.py $OUT/${TAG}_pmc$i pmc > $OUT/${TAG}_pmc$i.txt 2>&1
  find $OUT/${TAG}_pmc$i -name "*.csv" -size +8M -delete
  tail -2 $OUT/${TAG}_pmc$i.log
done
grep -A12 "TEpiGate" $OUT/${TAG}_pmc1.txt | head -30
