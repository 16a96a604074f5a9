"""Does running the batch as several independent half-batches on separate HIP streams de-phase the HBM-bound and
MFMA-bound kernel phases?  python tools/gpu_multistream.py <precision> <steps>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import diffsvc_amd
from diffsvc_amd import synth
from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
prec = sys.argv[1] if len(sys.argv) > 1 else "f16_d16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
hp = dict(synth.HPARAMS_44K)
sd = synth.acoustic_state(hp, 0)
for total, parts in ((32, 1), (32, 2), (32, 4), (64, 2), (64, 4), (48, 3)):
    B = total // parts
    hs = []
    for i in range(parts):
        den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=prec, prefix="denoise_fn.")
        hs.append((den, SamplerHandle(den, sd), torch.cuda.Stream(), torch.randn(B, 256, 861, device="cuda") * 0.5))
    for den, smp, st, cond in hs:                       # warm-up + graph capture
        with torch.cuda.stream(st):
            smp.sample(cond, 25, seed=1, use_graph=True)
    torch.cuda.synchronize(); t0 = time.time()
    for den, smp, st, cond in hs:
        with torch.cuda.stream(st):
            smp.sample(cond, steps, seed=2, use_graph=True)
    torch.cuda.synchronize(); dt = time.time() - t0
    print("%d clips as %d x %d on %d stream(s): %.3f ms/step  -> %.1fx RT @1000 steps" % (
        total, parts, B, parts, dt / steps * 1e3, 10.0 * total / (dt / steps * 1000)), flush=True)
    del hs
    torch.cuda.empty_cache()
