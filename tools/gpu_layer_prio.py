"""A/B of the fused layer kernel's pipe-sharing schemes (tlayer.h PRIOV 0..2) at 32 clips: ms per DDPM step (graph replay) and bit-equality
with scheme 0 (priorities do not change the arithmetic).   python tools/gpu_layer_prio.py <precision> <steps>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import diffsvc_amd
from diffsvc_amd import _lib
_lib.use_profiling_build()        # the knob this tool turns exists in libdsvc_hip_prof.so only (python -m diffsvc_amd.build --profiling)
from diffsvc_amd import synth
from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
prec = sys.argv[1] if len(sys.argv) > 1 else "f16_w2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 128
hp = dict(synth.HPARAMS_44K)
sd = synth.acoustic_state(hp, 0)
den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=prec, prefix="denoise_fn.")
smp = SamplerHandle(den, sd)
cond = torch.randn(32, 256, 861, device="cuda") * 0.5
ref = None
for m in (0, 1, 2, 0, 1, 2):
    den.debug_set("layer_prio", m)
    smp.sample(cond, 70, seed=1, use_graph=True)
    torch.cuda.synchronize(); t0 = time.time()
    mel = smp.sample(cond, steps, seed=2, use_graph=True)
    torch.cuda.synchronize(); dt = (time.time() - t0) / steps * 1e3
    if ref is None:
        ref = mel.clone()
    print("%s layer_prio %d: %.3f ms/step (%.1f us per layer incl. tail), bit-equal to scheme 0: %s" % (prec, m, dt, dt * 50, bool(torch.equal(mel, ref))), flush=True)
