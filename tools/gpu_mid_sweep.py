"""Mid-size batches (7 ... 20 ten-second clips) at f16_w6: ms per DDPM step of the fused layer kernel on 32- / 64- / 128-frame tiles
(debug_set fused_nt), of the two-launch layer (f16_w2 arithmetic: what these batches ran until round 4) and of f16_x3t.
   python tools/gpu_mid_sweep.py [precision]"""
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
BS = (7, 8, 10, 12, 14, 16, 18, 20, 24)
PREC = sys.argv[1] if len(sys.argv) > 1 else "f16_w6"
res = {}


def time_it(smp, B):
    steps = 64                                            # one dither period: the captured graph
    cond = torch.randn(B, 256, 861, device="cuda") * 0.5
    smp.sample(cond, 130, seed=1, use_graph=True)
    best = 1e9
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        smp.sample(cond, 2 * steps, seed=2 + rep, use_graph=True)
        torch.cuda.synchronize(); best = min(best, (time.time() - t0) / (2 * steps) * 1e3)
    return best


den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=PREC, prefix="denoise_fn.")
smp = SamplerHandle(den, sd)
for mode in ("nt1", "nt2", "nt4", "two-launch", "auto"):
    den.debug_set("two_launch_layer", 1 if mode == "two-launch" else 0)
    den.debug_set("fused_nt", int(mode[2]) if mode.startswith("nt") else 0)
    for B in BS:
        res[(mode, B)] = time_it(smp, B)
        if mode == "auto":
            res[("kind", B)] = smp.profile_gate_kernel(B, 861, 1)[2]
del smp, den
torch.cuda.empty_cache()
den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision="f16_x3t", prefix="denoise_fn.")
smp = SamplerHandle(den, sd)
for B in BS:
    res[("x3t", B)] = time_it(smp, B)
print("%s: ms per DDPM step" % PREC)
print("clips  tiles128   32-frame   64-frame  128-frame  two-launch(w2)   auto (N-tiles)    f16_x3t")
for B in BS:
    print("%5d %9d %10.3f %10.3f %10.3f %15.3f %10.3f (%d) %12.3f" % (B, (B * 896 + 127) // 128, res[("nt1", B)], res[("nt2", B)], res[("nt4", B)],
                                                                   res[("two-launch", B)], res[("auto", B)], res[("kind", B)], res[("x3t", B)]), flush=True)
