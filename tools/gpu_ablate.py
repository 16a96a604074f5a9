"""Ablation timings of the two per-layer kernels (DSVC_TG_DEBUG knobs): python tools/gpu_ablate.py [precision]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import diffsvc_amd
from diffsvc_amd import synth
from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
prec = sys.argv[1] if len(sys.argv) > 1 else "f16_d16"
hp = dict(synth.HPARAMS_44K)
sd = synth.acoustic_state(hp, 0)
den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=prec, prefix="denoise_fn.")
smp = SamplerHandle(den, sd)
names = {32: "L2-hot weights (all tiles = tile 0)", 39: "mainloop only, L2-hot weights", 64: "no pass rotation", 71: "mainloop only, no rotation", 0: "full", 1: "no acc-init loads", 2: "no epilogue", 3: "no init, no epilogue", 4: "no tile DMA", 7: "mainloop only (no init/epi/DMA)",
         8: "no MFMA loop", 15: "empty (launch + barrier)", 16: "no priority split", 11: "DMA only", 14: "init loads only", 13: "epilogue only"}
for B in (32,):
    cond = torch.randn(B, 256, 861, device="cuda") * 0.5
    smp.sample(cond, 3, seed=1, use_graph=False)
    for which in ("gate", "out"):
        os.environ["DSVC_PROFILE_KERNEL"] = which
        for dbg in (0, 64, 16, 32, 1, 2, 3, 4, 7, 71, 39, 8, 11, 14, 13, 15):
            os.environ["DSVC_TG_DEBUG"] = str(dbg)
            us, rows = smp.profile_gate_kernel(B, 861, 3)
            print("B=%-2d %-4s dbg=%-2d %-34s %8.1f us" % (B, which, dbg, names[dbg], us), flush=True)
os.environ.pop("DSVC_TG_DEBUG"); os.environ.pop("DSVC_PROFILE_KERNEL")
