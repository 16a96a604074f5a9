cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pe.py tests/test_gpu_infer.py -q -rP 2>&1 | grep -E "passed|failed|^pe |^configs|Error|assert" | cut -c1-330 | tee gpurun_out/r5e_pe_tests.txt
timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -rP -k "pe" 2>&1 | grep -E "passed|failed|Error|assert" | cut -c1-330 | tee -a gpurun_out/r5e_pe_tests.txt
python - <<'PY'
import sys, time, torch
sys.path.insert(0,'.')
import diffsvc_amd
from diffsvc_amd import synth
from diffsvc_amd.pe import PitchExtractorHip
hp24 = dict(synth.HPARAMS_24K)
pe = PitchExtractorHip(hparams=hp24).cuda(); pe.load_state_dict(synth.pe_state(hp24, 5))
for B in (1, 4, 32):
    mel = torch.from_numpy(synth.mel_like(1, B, 1875, 80)).cuda()
    pe(mel); torch.cuda.synchronize(); t=time.time()
    for _ in range(10): pe(mel)
    torch.cuda.synchronize(); print("pe B=%d T=1875: %.3f ms" % (B, (time.time()-t)/10*1e3))
PY
