"""PLMS-only workload: python tools/prof_plms.py <B> <T> <precision> [reps]  -> ms per 50-iteration chain (pndm_speedup 20, K_step 1000), graph replay"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import diffsvc_amd
from diffsvc_amd import synth
from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
B, T, prec = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
hp = dict(synth.HPARAMS_44K)
sd = synth.acoustic_state(hp, 0)
den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=prec, prefix="denoise_fn.")
smp = SamplerHandle(den, sd)
cond = torch.randn(B, 256, T, device="cuda") * 0.5
for _ in range(3):
    smp.sample(cond, 1000, speedup=20, seed=1, use_graph=True)
torch.cuda.synchronize(); t0 = time.time()
for i in range(reps):
    smp.sample(cond, 1000, speedup=20, seed=2 + i, use_graph=True)
torch.cuda.synchronize(); dt = (time.time() - t0) / reps
print("PLMS-50 B=%d T=%d %s: %.3f ms per chain" % (B, T, prec, dt * 1e3))
