"""Where precision="auto" should hand a DDPM call from f16_x3t (fp32-class) to f16_w2: ms per DDPM step of both over the clip count.
   python tools/gpu_auto_sweep.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import diffsvc_amd
from diffsvc_amd import _lib as _dsvc_lib
_dsvc_lib.hooks_build().__enter__()      # this tool sets dsvc_*_debug_set keys: they exist in the test-hooks build only (round 6)
from diffsvc_amd import synth
from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
hp = dict(synth.HPARAMS_44K)
sd = synth.acoustic_state_conditioned(hp, 0, 1.5, 0.07)
BS = (1, 2, 4, 6, 8, 10, 12, 14, 16, 20, 24, 32)
res = {}
for prec in ("f16_w2", "f16_w2/two-launch", "f16_x3t"):
    den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=prec.split("/")[0], prefix="denoise_fn.")
    if "/" in prec:
        den.debug_set("two_launch_layer", 1)       # the layer as gate + output launches (channel passes over blockIdx.y) at every size
    smp = SamplerHandle(den, sd)
    for B in BS:
        steps = max(24, 240 // B)
        cond = torch.randn(B, 256, 861, device="cuda") * 0.5
        smp.sample(cond, 25, seed=1, use_graph=True)
        best = 1e9
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.time()
            smp.sample(cond, steps, seed=2 + rep, use_graph=True)
            torch.cuda.synchronize(); best = min(best, (time.time() - t0) / steps * 1e3)
        res[(prec, B)] = best
    del smp, den
    torch.cuda.empty_cache()
print("clips  frames   f16_w2 ms/step  w2 two-launch  f16_x3t ms/step  x3t/min(w2)")
for B in BS:
    a, a2, b = res[("f16_w2", B)], res[("f16_w2/two-launch", B)], res[("f16_x3t", B)]
    print("%5d %7d %14.3f %14.3f %16.3f %9.2f" % (B, B * 861, a, a2, b, b / min(a, a2)), flush=True)
