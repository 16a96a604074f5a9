"""Do the MFMA-bound gate phase and the HBM-bound output phase of the fused layer kernel overlap when two independent half-batches run
on separate HIP streams with a DELIBERATE phase offset?  (Clips are independent, so a batch of 32 can run as two chains of 16 whose
layer kernels -- 112 workgroups each -- share the chip; started together they stay in phase, so stream 2 is held back by `offset` us.)
    python tools/gpu_phase_offset.py <precision> <steps>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import diffsvc_amd
from diffsvc_amd import synth
from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
prec = sys.argv[1] if len(sys.argv) > 1 else "f16_w2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 128
hp = dict(synth.HPARAMS_44K)
sd = synth.acoustic_state(hp, 0)


def build(B):
    den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=prec, prefix="denoise_fn.")
    return den, SamplerHandle(den, sd), torch.cuda.Stream(), torch.randn(B, 256, 861, device="cuda") * 0.5


one = build(32)
with torch.cuda.stream(one[2]):
    one[1].sample(one[3], 70, seed=1, use_graph=True)
torch.cuda.synchronize(); t0 = time.time()
with torch.cuda.stream(one[2]):
    one[1].sample(one[3], steps, seed=2, use_graph=True)
torch.cuda.synchronize(); base = (time.time() - t0) / steps * 1e3
print("%s: 32 clips in one batch: %.3f ms/step (%.1f us per layer incl. the step tail)" % (prec, base, base * 1e3 / 20), flush=True)
del one
torch.cuda.empty_cache()
hs = [build(16), build(16)]
for den, smp, st, cond in hs:
    with torch.cuda.stream(st):
        smp.sample(cond, 70, seed=1, use_graph=True)
torch.cuda.synchronize()
clock_mhz = 100.0                                      # torch.cuda._sleep counts cycles of the fixed-rate counter on ROCm builds; calibrated below
t0 = time.time(); torch.cuda._sleep(10_000_000); torch.cuda.synchronize(); per_cycle_us = (time.time() - t0) * 1e6 / 10_000_000
print("torch.cuda._sleep: %.4f us per cycle" % per_cycle_us, flush=True)
for off_us in (0, 20, 40, 60, 80, 100):
    torch.cuda.synchronize(); t0 = time.time()
    for i, (den, smp, st, cond) in enumerate(hs):
        with torch.cuda.stream(st):
            if i == 1 and off_us:
                torch.cuda._sleep(int(off_us / per_cycle_us))
            smp.sample(cond, steps, seed=2, use_graph=True)
    torch.cuda.synchronize(); dt = (time.time() - t0) / steps * 1e3
    print("%s: 2 x 16 clips on two streams, stream 2 held back %3d us: %.3f ms/step (%+.1f %% vs one batch)" % (prec, off_us, dt, 100 * (dt / base - 1)), flush=True)
