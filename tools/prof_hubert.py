"""HuBERT-soft alone on one 10 s utterance (160 000 samples at 16 kHz): python tools/prof_hubert.py [reps]  (for rocprofv3 --kernel-trace --stats)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import diffsvc_amd
from diffsvc_amd import synth
from diffsvc_amd.hubert import HubertSoftHip
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
h = HubertSoftHip(synth.hubert_state(11))
g = torch.Generator().manual_seed(3)
w16 = (torch.rand(160000, generator=g) * 2 - 1).mul(0.3).cuda()
for _ in range(3):
    u = h.units(w16)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    u = h.units(w16)
torch.cuda.synchronize()
print("HuBERT-soft: %.2f ms per 10 s utterance, units %s finite %s" % ((time.perf_counter() - t0) / reps * 1e3, tuple(u.shape), bool(torch.isfinite(u).all())))
