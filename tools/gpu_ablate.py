"""Ablation timings of the two per-layer kernels (DSVC_TG_DEBUG knobs): python tools/gpu_ablate.py [precision] [B] [dbg,dbg,...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import diffsvc_amd
from diffsvc_amd import synth
from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
prec = sys.argv[1] if len(sys.argv) > 1 else "f16_d64"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dbgs = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 7, 4103, 4096, 39, 15]
hp = dict(synth.HPARAMS_44K)
sd = synth.acoustic_state(hp, 0)
den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=prec, prefix="denoise_fn.")
smp = SamplerHandle(den, sd)
names = {0: "full", 1: "no acc-init loads", 2: "no epilogue", 4: "no tile DMA", 7: "mainloop only (no init/epi/DMA)", 8: "no MFMA loop",
         15: "empty (launch + barrier)", 32: "L2-hot weights (all tiles = tile 0)", 39: "mainloop only, L2-hot weights",
         4096: "L1-hot weights (ring re-reads group 0)", 4103: "mainloop only, L1-hot weights", 256: "waves 4-7 skip MFMAs", 263: "mainloop only, waves 4-7 idle"}
cond = torch.randn(B, 256, 861, device="cuda") * 0.5
smp.sample(cond, 3, seed=1, use_graph=False)
for which in ("gate", "out"):
    os.environ["DSVC_PROFILE_KERNEL"] = which
    for dbg in dbgs:
        os.environ["DSVC_TG_DEBUG"] = str(dbg)
        us, rows = smp.profile_gate_kernel(B, 861, 5)
        print("B=%-2d %-4s dbg=%-4d %-40s %8.1f us" % (B, which, dbg, names.get(dbg, ""), us), flush=True)
os.environ.pop("DSVC_TG_DEBUG"); os.environ.pop("DSVC_PROFILE_KERNEL")
