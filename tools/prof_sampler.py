"""Sampler-only workload for rocprofv3: python tools/prof_sampler.py <B> <steps> <precision> [graph] [T]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import diffsvc_amd
if os.environ.get("DSVC_PROF_KNOBS"):       # knobs are dsvc_denoiser_debug_set keys: they exist in the test-hooks build only (round 6); without
    from diffsvc_amd import _lib as _dsvc_lib     # them this workload runs on the product library (what the PMC traffic passes profile)
    _dsvc_lib.hooks_build().__enter__()
from diffsvc_amd import synth
from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
B, steps, prec = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
graph = len(sys.argv) > 4 and sys.argv[4] == "graph"
hp = dict(synth.HPARAMS_44K)
sd = synth.acoustic_state(hp, 0)
den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=prec, prefix="denoise_fn.")
smp = SamplerHandle(den, sd)
for kv in os.environ.get("DSVC_PROF_KNOBS", "").split(","):          # e.g. DSVC_PROF_KNOBS=fused_tail=1 (dsvc_denoiser_debug_set keys)
    if "=" in kv:
        den.debug_set(kv.split("=")[0], int(kv.split("=")[1]))
T = int(sys.argv[5]) if len(sys.argv) > 5 else 861
cond = torch.randn(B, 256, T, device="cuda") * 0.5
smp.sample(cond, 130 if graph else 25, seed=1, use_graph=graph)      # (graph: two dither periods, so that the capture happens here)
torch.cuda.synchronize(); t0 = time.time()
smp.sample(cond, steps, seed=2, use_graph=graph)
torch.cuda.synchronize(); dt = time.time() - t0
print("B=%d %s steps=%d graph=%d%s: %.3f ms/step" % (B, prec, steps, graph, "" if T == 861 else " T=%d" % T, dt / steps * 1e3))
