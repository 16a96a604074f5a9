#!/bin/bash
# In-kernel phase timeline of the single clip's kernels (profiling build, DSVC_TG_STAMPS): bash tools/gpu_b1_stamps.sh <tag>
TAG=${1:-r6ah}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
DSVC_TG_STAMPS=/tmp/st_b1 timeout 300 python - <<'PY' > gpurun_out/${TAG}_b1_stamps_run.txt 2>&1
import sys, runpy
sys.path.insert(0, ".")
import diffsvc_amd
from diffsvc_amd import _lib
_lib.use_profiling_build()
sys.argv = ["prof_sampler.py", "1", "10", "f16_x3t"]
runpy.run_path("tools/prof_sampler.py", run_name="__main__")
PY
python tools/stamps_report.py /tmp/st_b1 86 > gpurun_out/${TAG}_b1_stamps.txt 2>&1
head -50 gpurun_out/${TAG}_b1_stamps.txt
