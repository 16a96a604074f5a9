"""Round 4, f16_w6 (the dilated conv's w_lo * x correction on the block-scaled 6-bit MFMA inside the fused layer kernel): correctness and time
against f16_w2 at 32 clips.
    python tools/gpu_w6_check.py [steps]
1. one denoiser evaluation: eps of f16_w6 against f16_w2 (exact fp16 lo plane) and of plain f16 (no lo plane at all) against f16_w2 -- the
   6-bit product is right if |w6 - w2| is a few per cent of |f16 - w2| (the lo term's own quantisation), garbage shows as >= 100 %;
   the w6 handle with `w6_off` must equal the w2 handle bit for bit (same kernel, same planes);
2. ms per DDPM step (graph replay) of both, alternating;
3. a 1000-step chain of both from the same noise: the difference between them (two members of the same error class).
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import diffsvc_amd
from diffsvc_amd import _lib as _dsvc_lib
_dsvc_lib.hooks_build().__enter__()      # this tool sets dsvc_*_debug_set keys: they exist in the test-hooks build only (round 6)
from diffsvc_amd import synth
from diffsvc_amd.engine import DenoiserHandle, SamplerHandle

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B, T = 32, 861
hp = dict(synth.HPARAMS_44K)
sd = synth.acoustic_state(hp, 0)
g = torch.Generator().manual_seed(5)
spec = torch.randn(B, 1, 128, T, generator=g).cuda()
cond = (torch.randn(B, 256, T, generator=g) * 0.5).cuda()
t = torch.randint(0, 1000, (B,), generator=g).cuda()

outs = {}
for prec in ("f16_w2", "f16_w6d1", "f16_w6", "f16"):
    den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=prec, prefix="denoise_fn.")
    if prec == "f16":
        den.debug_set("two_launch_layer", 0)
    outs[prec] = den.forward(spec, t, cond).clone()
    if prec == "f16_w6":
        den.debug_set("w6_off", 1)
        outs["w6_off"] = den.forward(spec, t, cond).clone()
        den.debug_set("w6_off", 0)
    assert torch.isfinite(outs[prec]).all(), prec
    del den
ref = outs["f16_w2"]
scale = ref.abs().max().item()
for k in ("f16_w6d1", "f16_w6", "f16", "w6_off"):
    d = (outs[k] - ref)
    print("eps %-9s vs f16_w2: max |diff| %.3e  rms %.3e   (|eps| max %.2f)" % (k, d.abs().max().item(), d.pow(2).mean().sqrt().item(), scale), flush=True)
print("w6 handle with w6_off == w2 handle bit for bit:", bool(torch.equal(outs["w6_off"], ref)), flush=True)
r6 = (outs["f16_w6"] - ref).pow(2).mean().sqrt().item(); r1 = (outs["f16"] - ref).pow(2).mean().sqrt().item()
print("rms(w6 - w2) / rms(f16 - w2) = %.3f   (the 6-bit product carries the lo term if this is << 1)" % (r6 / max(r1, 1e-30)), flush=True)

handles = {}
for name, prec, g6off in (("f16_w2", "f16_w2", 0), ("f16_w6 (no g_lo)", "f16_w6", 1), ("f16_w6", "f16_w6", 0)):
    den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=prec, prefix="denoise_fn.")
    if g6off:
        den.debug_set("g6_off", 1)
    handles[name] = (den, SamplerHandle(den, sd))
for rep in range(2):
    for name in handles:
        den, smp = handles[name]
        smp.sample(cond, 130, seed=1, use_graph=True)
        torch.cuda.synchronize(); t0 = time.time()
        mel = smp.sample(cond, 2 * steps, seed=2, use_graph=True)
        torch.cuda.synchronize(); dt = (time.time() - t0) / (2 * steps) * 1e3
        print("%-18s %.3f ms/step (%.1f us per layer incl. the step tail), finite %s" % (name, dt, dt * 50, bool(torch.isfinite(mel).all())), flush=True)
