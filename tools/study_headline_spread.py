"""How much margin does the benchmarked configuration have under the 1e-3 mel bar?  The committed golden pins two clips against the real
reference; this study runs the oracle (the pinned restatement) for further clips / seeds at the same configuration -- one 10 s clip,
T=861, all 1000 DDPM steps -- and prints the HIP path's max-abs mel error at f16_d64 and f16_w2.  ~70 s of host time per clip.
    python tools/study_headline_spread.py 2 3 4 5        (clip ids)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import diffsvc_amd
from diffsvc_amd import synth
from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
from util import oracle_sample
import dsvc_oracle as O

clips = [int(a) for a in sys.argv[1:]] or [2, 3]
hp = dict(synth.HPARAMS_44K)
sd = synth.acoustic_state(hp, 0)
torch.set_num_threads(min(32, os.cpu_count() or 1))
handles = {}
for prec in ("f16_m64", "f16_d64", "f16_w2"):
    den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=prec, prefix="denoise_fn.")
    handles[prec] = SamplerHandle(den, sd)
for c in clips:
    seed = 1000 + c
    t0 = time.time()
    with torch.no_grad():
        r = oracle_sample(hp, sd, [c], 861, 500, 1, seed, 1000)
    t_or = time.time() - t0
    line = "clip %d seed %d (oracle %.0f s):" % (c, seed, t_or)
    for prec, smp in handles.items():
        mel = smp.sample(r["cond_t"].cuda(), 1000, seed=seed, first_clip=c, mel2ph=r["mel2ph"].cuda())
        err = (mel.cpu() - r["mel_out"]).abs().max().item()
        line += "  %s %.2e" % (prec, err)
    print(line, flush=True)
