#!/bin/bash
# PitchExtractor first contact (+ HuBERT again: k_layernorm moved to rowops.h)
TAG=${1:-r2i}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pe.py tests/test_gpu_hubert.py -m gpu -q -rP > $OUT/${TAG}_pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/${TAG}_pytest_gpu.txt
grep -E "passed|failed|error" $OUT/${TAG}_pytest_gpu.txt | tail -3
grep -E "^(hubert|pe )" $OUT/${TAG}_pytest_gpu.txt
grep -E "^(FAILED|ERROR)|Error|^E  " $OUT/${TAG}_pytest_gpu.txt | head -30
timeout 300 python - > $OUT/${TAG}_pe_time.txt 2>&1 <<'PY'
import sys, time
sys.path[:0] = ["."]
import torch, diffsvc_amd
from diffsvc_amd import synth
from diffsvc_amd.pe import PitchExtractorHip
hp = dict(synth.HPARAMS_24K)
pe = PitchExtractorHip(hparams=hp).cuda()
pe.load_state_dict(synth.pe_state(hp, 5)); pe.eval()
for B, T in ((1, 1875), (32, 1875)):          # 10 s at 24 kHz / hop 128
    mel = torch.from_numpy(synth.mel_like(1, B, T, 80, (0,))).cuda()
    for _ in range(2): pe(mel)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): pe(mel)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("pitch extractor, %d x 10 s clips (%d frames each): %.2f ms per call = %.0fx real time" % (B, T, dt * 1e3, B * 10.0 / dt))
PY
cat $OUT/${TAG}_pe_time.txt | tail -4
