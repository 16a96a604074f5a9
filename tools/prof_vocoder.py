"""Vocoder-only workload for rocprofv3: python tools/prof_vocoder.py <B> <reps> [precision]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import diffsvc_amd
if os.environ.get("DSVC_USE_PROF"):      # A/B against another build parked as libdsvc_hip_prof.so
    from diffsvc_amd import _lib
    _lib.use_profiling_build()
from diffsvc_amd import synth
from diffsvc_amd.engine import VocoderHandle
B, reps = int(sys.argv[1]), int(sys.argv[2])
prec = sys.argv[3] if len(sys.argv) > 3 else "f16_x3"
h = dict(synth.VOCODER_44K)
voc = VocoderHandle(synth.vocoder_state(h, 1), h, precision=prec)
mel = (torch.randn(B, 861, 128, device="cuda") * 0.5 - 2.5).clamp(-6, 1.5)
f0 = torch.full((B, 861), 220.0, device="cuda")
voc.vocode(mel, f0, seed=1); torch.cuda.synchronize(); t0 = time.time()
for i in range(reps): voc.vocode(mel, f0, seed=2 + i)
torch.cuda.synchronize()
print("B=%d %s: %.2f ms per call" % (B, prec, (time.time() - t0) / reps * 1e3))
