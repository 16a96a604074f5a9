"""f16_x3t (split activations on the tgemm engine) against f16_x3 (conv_gemm engine) and f16_w2: ms per DDPM step at 1 / 8 / 32 clips and the
50-iteration PLMS clip (51 evaluations, graph).   python tools/gpu_x3t_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import diffsvc_amd
from diffsvc_amd import synth
from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
hp = dict(synth.HPARAMS_44K)
sd = synth.acoustic_state_conditioned(hp, 0, 1.5, 0.07)
for prec in ("f16_w2", "f16_x3t", "f16_x3"):
    den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=prec, prefix="denoise_fn.")
    smp = SamplerHandle(den, sd)
    line = prec + ":"
    for B, steps in ((1, 200), (8, 60), (32, 30)):
        cond = torch.randn(B, 256, 861, device="cuda") * 0.5
        smp.sample(cond, 25, seed=1, use_graph=True)
        torch.cuda.synchronize(); t0 = time.time()
        smp.sample(cond, steps, seed=2, use_graph=True)
        torch.cuda.synchronize(); line += "  B=%d %.3f ms/step" % (B, (time.time() - t0) / steps * 1e3)
    cond = torch.randn(1, 256, 861, device="cuda") * 0.5
    smp.sample(cond, 1000, speedup=20, seed=1, use_graph=True)
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(5):
        mel = smp.sample(cond, 1000, speedup=20, seed=2 + i, use_graph=True)
    torch.cuda.synchronize(); line += "  | PLMS-50 (51 evaluations, sampler only) %.2f ms per clip, finite %s" % ((time.time() - t0) / 5 * 1e3, bool(torch.isfinite(mel).all()))
    print(line, flush=True)
    del smp, den
    torch.cuda.empty_cache()
