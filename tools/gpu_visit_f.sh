#!/bin/bash
# HuBERT-soft first contact + the optimizer test after the StepLR fix
TAG=${1:-r2f}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hubert.py tests/test_gpu_train.py -m gpu -q -rP > $OUT/${TAG}_pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/${TAG}_pytest_gpu.txt
grep -E "passed|failed|error" $OUT/${TAG}_pytest_gpu.txt | tail -3
grep -E "^(hubert|train step|optimizer)" $OUT/${TAG}_pytest_gpu.txt
grep -E "^(FAILED|ERROR)|Error" $OUT/${TAG}_pytest_gpu.txt | head -20
timeout 300 python - > $OUT/${TAG}_hubert_time.txt 2>&1 <<'PY'
import sys, time
sys.path[:0] = ["."]
import torch, diffsvc_amd
from diffsvc_amd import synth
from diffsvc_amd.hubert import HubertSoftHip
hb = HubertSoftHip(synth.hubert_state(11))
wav = torch.from_numpy(synth.speech_like_wav(1, 160000)).cuda()
for _ in range(2): hb.units(wav)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): hb.units(wav)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("hubert-soft units, 10 s clip (160000 samples -> 500 frames): %.2f ms per clip = %.0fx real time" % (dt * 1e3, 10.0 / dt))
PY
cat $OUT/${TAG}_hubert_time.txt | tail -3
