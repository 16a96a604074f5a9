"""Round 4: the single-clip precision f16_x3t with its w_lo * x_hi term on the 6-bit MFMA (tgemm W6 kernels) against the same handle with the fp16
lo plane (debug_set x3t_w6_off 1): ms per DDPM step (graph replay, one 10 s clip) and the 1000-step mel error against the real-reference golden.
    python tools/gpu_x3t6_check.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import diffsvc_amd
from diffsvc_amd import _lib as _dsvc_lib
_dsvc_lib.hooks_build().__enter__()      # this tool sets dsvc_*_debug_set keys: they exist in the test-hooks build only (round 6)
from diffsvc_amd import synth
from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
import dsvc_oracle as O
from util import clip_batch, load_golden

hp = dict(synth.HPARAMS_44K)
sd = synth.acoustic_state(hp, 0)
g = load_golden("e2e_44k_T861_k1000")
clips = [int(c) for c in g["clips"]]
hub, m2p, f0 = clip_batch(hp, clips, 861, 500)
cond, _, _ = O.build_cond(sd, hub, m2p, f0.clone(), hp)
cond = cond.transpose(1, 2).contiguous().cuda()
handles = {}
for name, off in (("f16_x3t fp16 lo plane", 1), ("f16_x3t 6-bit lo codes", 0)):
    den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision="f16_x3t", prefix="denoise_fn.")
    den.debug_set("x3t_w6_off", off)
    handles[name] = (den, SamplerHandle(den, sd))
for rep in range(2):
    for name, (den, smp) in handles.items():
        smp.sample(cond[:1], 200, mel2ph=m2p[:1].cuda(), seed=1, first_clip=0, use_graph=True)
        torch.cuda.synchronize(); t0 = time.time()
        smp.sample(cond[:1], 1000, mel2ph=m2p[:1].cuda(), seed=2, first_clip=0, use_graph=True)
        torch.cuda.synchronize(); dt = (time.time() - t0)
        print("%-24s %.4f ms/step (%.1f ms per clip: %.1fx RT for the sampler alone)" % (name, dt, dt * 1e3, 10.0 / dt), flush=True)
for name, (den, smp) in handles.items():
    errs = []
    for i, c in enumerate(clips):
        mel = smp.sample(cond[i:i + 1], 1000, mel2ph=m2p[i:i + 1].cuda(), seed=int(g["seed"]), first_clip=c, use_graph=True)
        errs.append((mel[0].cpu() - torch.from_numpy(g["mel_out"][i])).abs().max().item())
    print("%-24s 1000-step mel max-abs error vs the real reference: %s" % (name, ["%.2e" % e for e in errs]), flush=True)
