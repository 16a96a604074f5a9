"""The fused step tail (csrc/ttail.h: skip projection -> output projection + posterior step -> the next evaluation's input projection in ONE launch)
against the three tgemm launches: state after 1 / 2 / 20 DDPM steps (fp32 summation order only) and ms per step at 32 / 16 / 8 clips.
   python tools/gpu_fused_tail.py"""
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
sd = synth.acoustic_state(hp, 0)
den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision="f16_w6", prefix="denoise_fn.")
smp = SamplerHandle(den, sd)
for B in (32, 16, 8):
    cond = torch.randn(B, 256, 861, device="cuda") * 0.5
    for n in (1, 2, 20):
        out = {}
        for f in (0, 2, 3):
            den.debug_set("fused_tail", f)
            out[f] = smp.sample(cond, n, seed=1, use_graph=False, return_x=True)[1]
        d = (out[2] - out[0]).abs().max().item()
        print("B=%d, %2d steps: fused tail vs three launches, max |diff| of the state %.2e (finite %s); 32-frame form == 64-frame form: %s" % (
            B, n, d, bool(torch.isfinite(out[2]).all()), bool(torch.equal(out[2], out[3]))), flush=True)
    for rnd in range(2):
        for f in (0, 2, 3):
            den.debug_set("fused_tail", f)
            smp.sample(cond, 130, seed=1, use_graph=True)
            torch.cuda.synchronize(); t0 = time.time()
            smp.sample(cond, 256, seed=2, use_graph=True)
            torch.cuda.synchronize()
            print("B=%d fused_tail=%d: %.3f ms/step" % (B, f, (time.time() - t0) / 256 * 1e3), flush=True)
