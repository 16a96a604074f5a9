"""Follow-up of tools/study_headline_spread.py: which cheap precision schedule keeps the 1000-step mel error robustly under 1e-3?
   python tools/study_hybrid.py <clip ids...>
Schedules (DDPM steps 999..0): d64 = dithered single-plane weights; mix = dithered dilated conv + exact (hi+lo) output 1x1;
w2 = exact weights everywhere.  'A>=s|B' runs A for t >= s and B for t < s (state handed over through x_init / return_x)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import diffsvc_amd
from diffsvc_amd import synth
from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
from util import oracle_sample

clips = [int(a) for a in sys.argv[1:]] or [4, 6]
hp = dict(synth.HPARAMS_44K)
sd = synth.acoustic_state(hp, 0)
torch.set_num_threads(min(32, os.cpu_count() or 1))

def handle(prec, out_w2):
    os.environ["DSVC_OUT_W2"] = "1" if out_w2 else "0"
    den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=prec, prefix="denoise_fn.")
    return SamplerHandle(den, sd)

H = {"d64": handle("f16_d64", False), "mix": handle("f16_d64", True), "w2": handle("f16_w2", False)}
SCHED = [("d64", None, None), ("mix", None, None), ("d64", 200, "w2"), ("d64", 100, "w2"), ("mix", 100, "w2"), ("mix", 50, "w2"), ("d64", 200, "mix"), ("w2", None, None)]
for c in clips:
    seed = 1000 + c
    with torch.no_grad():
        r = oracle_sample(hp, sd, [c], 861, 500, 1, seed, 1000)
    cond, m2p = r["cond_t"].cuda(), r["mel2ph"].cuda()
    line = "clip %d:" % c
    for a, s, b in SCHED:
        if s is None:
            mel = H[a].sample(cond, 1000, seed=seed, first_clip=c, mel2ph=m2p)
            name = a
        else:
            _, x = H[a].sample(cond, 1000, seed=seed, first_clip=c, mel2ph=m2p, t_stop=s, return_x=True)
            mel = H[b].sample(cond, s, seed=seed, first_clip=c, mel2ph=m2p, x_init=x)
            name = "%s>=%d|%s" % (a, s, b)
        line += "  %s %.2e" % (name, (mel.cpu() - r["mel_out"]).abs().max().item())
    print(line, flush=True)
