"""A/B at 32 clips: the fused layer kernel with the deferred skip contraction (tskip.h: default) against its in-layer skip accumulation
(debug_set defer_skip 0) -- ms per DDPM step (graph replay) and the difference of the sampled mel.
    python tools/gpu_defer_ab.py <precision> <steps>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import diffsvc_amd
from diffsvc_amd import _lib as _dsvc_lib
_dsvc_lib.hooks_build().__enter__()      # this tool sets dsvc_*_debug_set keys: they exist in the test-hooks build only (round 6)
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
for mode in (1, 0, 1, 0):
    den.debug_set("defer_skip", mode)
    smp.sample(cond, 70, seed=1, use_graph=True)
    torch.cuda.synchronize(); t0 = time.time()
    mel = smp.sample(cond, steps, seed=2, use_graph=True)
    torch.cuda.synchronize(); dt = (time.time() - t0) / steps * 1e3
    if ref is None:
        ref = mel.clone()
    print("%s defer_skip %d: %.3f ms/step (%.1f us per layer incl. tail), max |mel diff| vs the first run %.2e, finite %s"
          % (prec, mode, dt, dt * 50, (mel - ref).abs().max().item(), bool(torch.isfinite(mel).all())), flush=True)
