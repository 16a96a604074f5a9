"""Tiling of the three small projections of a DDPM step (input projection, skip projection + ReLU, output projection + posterior step) at 32 clips:
ms per step for every value of the "tail_tiling" knob (two bits each: in | skip << 2 | out << 4; 0 = the shipped 64- / 128-frame tiles, 1 = 32-frame tiles
with one workgroup per output pass, 2 = 32-frame tiles, one workgroup walking all passes).   python tools/gpu_tail_tiling.py [precision] [clips]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import diffsvc_amd
from diffsvc_amd import _lib
_lib.use_profiling_build()        # the knob this tool turns exists in libdsvc_hip_prof.so only (python -m diffsvc_amd.build --profiling)
from diffsvc_amd import synth
from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
prec = sys.argv[1] if len(sys.argv) > 1 else "f16_w6"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
hp = dict(synth.HPARAMS_44K)
sd = synth.acoustic_state(hp, 0)
den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=prec, prefix="denoise_fn.")
smp = SamplerHandle(den, sd)
cond = torch.randn(B, 256, 861, device="cuda") * 0.5
ref = None
for knob in (0, 21, 42, 1, 2, 4, 8, 16, 32, 0):
    den.debug_set("tail_tiling", knob)
    x = smp.sample(cond, 130, seed=1, use_graph=True)
    if ref is None:
        ref = x.clone()
    same = bool(torch.equal(x, ref))
    best = 1e9
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        smp.sample(cond, 256, seed=2 + rep, use_graph=True)
        torch.cuda.synchronize(); best = min(best, (time.time() - t0) / 256 * 1e3)
    print("tail_tiling %2d (in %d, skip %d, out %d): %.3f ms/step   130-step mel %s the default tiling's" % (knob, knob & 3, (knob >> 2) & 3, (knob >> 4) & 3, best,
          "==" if same else "max |diff| %.2e from" % (x - ref).abs().max().item()), flush=True)
