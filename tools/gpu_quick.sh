#!/bin/bash
# quick GPU visit: diagnostics of the tgemm path, then the parity tests.  bash tools/gpu_quick.sh <tag>
TAG=${1:-q}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python tools/gpu_diag2.py > $OUT/${TAG}_diag2.txt 2>&1
echo "diag rc=$?"; tail -50 $OUT/${TAG}_diag2.txt
timeout 1200 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_diffnet.py::test_plain_f16_fails_the_bar_dither_is_needed > $OUT/${TAG}_pytest_gpu.txt 2>&1
echo "pytest rc=$?"; tail -30 $OUT/${TAG}_pytest_gpu.txt
