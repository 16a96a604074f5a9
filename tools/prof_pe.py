"""The 24 kHz pitch extractor alone on 1875 frames (10 s at hop 128): python tools/prof_pe.py [reps]  (for rocprofv3 --kernel-trace --stats)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import diffsvc_amd
from diffsvc_amd import synth
from diffsvc_amd.pe import PitchExtractorHip
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
hp24 = dict(synth.HPARAMS_24K)
pe = PitchExtractorHip(hparams=hp24).cuda()
pe.load_state_dict(synth.pe_state(hp24, 5))
mel24 = torch.from_numpy(synth.mel_like(1, 1, 1875, 80)).cuda()
for _ in range(3):
    out = pe(mel24)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    out = pe(mel24)
torch.cuda.synchronize()
print("pitch extractor: %.3f ms per 1875 frames" % ((time.perf_counter() - t0) / reps * 1e3))
