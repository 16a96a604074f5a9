"""Where the fused step tail (csrc/ttail.h) spends its time: shader-clock stamps at the phase boundaries, per wave (needs the DSVC_PROFILING build).
   python -m diffsvc_amd.build --profiling && python tools/gpu_tail_stamps.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from diffsvc_amd import _lib
_lib.use_profiling_build()
from diffsvc_amd import synth
from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
hp = dict(synth.HPARAMS_44K)
sd = synth.acoustic_state(hp, 0)
den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision="f16_w6", prefix="denoise_fn.")
smp = SamplerHandle(den, sd)
NAMES = ["tile landed", "skip proj done", "s2 parked", "out proj done", "posterior done", "state visible", "in proj done"]
for B in (32, 8):
    cond = torch.randn(B, 256, 861, device="cuda") * 0.5
    for mode in (2, 3):
        den.debug_set("fused_tail", 10 + mode)
        smp.sample(cond, 6, seed=1, use_graph=False)
        torch.cuda.synchronize()
        e = den.debug_buffer("eps").flatten()
        waves = 8 if mode == 2 else 4
        wgs = den.debug_buffer("eps").shape[0] // (64 if mode == 2 else 32)
        st = e[: wgs * waves * 8].view(wgs, waves, 8)[:, :, :7]
        print("B=%d mode=%d: %d workgroups x %d waves; shader clocks since the wave started (mean over workgroups | min | max):" % (B, mode, wgs, waves))
        for i, n in enumerate(NAMES):
            v = st[:, :, i]
            print("   %-16s mean %8.0f  min %8.0f  max %8.0f   (wave 0: %8.0f, last wave: %8.0f)" % (n, v.mean(), v.min(), v.max(), v[:, 0].mean(), v[:, -1].mean()))
