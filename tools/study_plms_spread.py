"""PLMS-50 (pndm_speedup 20 over the 1000-step schedule, 51 evaluations, T=861) on the conditioned synthetic checkpoint: HIP path vs the
oracle for several (clip, noise) pairs at f16_w2 (the precision 'auto' picks) -- how much margin under the 1e-3 mel bar?
    python tools/study_plms_spread.py 0 1 2 3 ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import diffsvc_amd
from diffsvc_amd import synth
from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
from util import oracle_sample

clips = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3]
hp = dict(synth.HPARAMS_44K)
sd = synth.acoustic_state_conditioned(hp, 0, 1.5, 0.07)
torch.set_num_threads(min(32, os.cpu_count() or 1))
H = {}
for prec in ("f16_w2", "f16_m64"):
    den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=prec, prefix="denoise_fn.")
    H[prec] = SamplerHandle(den, sd)
for c in clips:
    seed = 2000 + c
    t0 = time.time()
    with torch.no_grad():
        r = oracle_sample(hp, sd, [c], 861, 500, 20, seed, 1000)
    line = "clip %d seed %d (oracle %.0f s, mel %.2f..%.2f):" % (c, seed, time.time() - t0, r["mel_out"].min().item(), r["mel_out"].max().item())
    for prec, smp in H.items():
        mel = smp.sample(r["cond_t"].cuda(), 1000, speedup=20, seed=seed, first_clip=c, mel2ph=r["mel2ph"].cuda())
        d = mel.cpu() - r["mel_out"]
        line += "  %s max %.2e rms %.2e" % (prec, d.abs().max().item(), d.pow(2).mean().sqrt().item())
    print(line, flush=True)
