"""Round 4 experiment: the fused layer kernel runs a matrix-pipe-bound gate phase and then an HBM-bound output phase, and with one workgroup
per CU and one wave of tiles every CU is in the same phase at the same time.  Two half batches on two streams, half a layer out of phase,
would overlap one half's output phase (HBM) with the other half's gate phase (MFMA).
    python tools/gpu_dephase.py [steps] [precision]
Measures ms per DDPM step for (a) one 32-clip sampler, (b) two 16-clip samplers on two streams from two host threads, started together,
(c) the same with the second stream delayed by a spin of D microseconds before every replay.
"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import diffsvc_amd
from diffsvc_amd import _lib as _dsvc_lib
_dsvc_lib.hooks_build().__enter__()      # this tool sets dsvc_*_debug_set keys: they exist in the test-hooks build only (round 6)
from diffsvc_amd import synth
from diffsvc_amd.engine import DenoiserHandle, SamplerHandle

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prec = sys.argv[2] if len(sys.argv) > 2 else "f16_w6"
T = 861
hp = dict(synth.HPARAMS_44K)
sd = synth.acoustic_state(hp, 0)
g = torch.Generator().manual_seed(5)
cond32 = (torch.randn(32, 256, T, generator=g) * 0.5).cuda()


def make(n):
    den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=prec, prefix="denoise_fn.")
    if n < 32:
        den.debug_set("two_launch_layer", -1)      # the fused layer kernel also below the 120-tile threshold
    return den, SamplerHandle(den, sd)


den32, smp32 = make(32)
halves = [make(16), make(16)]
conds = [cond32[:16].contiguous(), cond32[16:].contiguous()]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run32():
    smp32.sample(cond32, 130, seed=1, use_graph=True)
    torch.cuda.synchronize(); t0 = time.time()
    smp32.sample(cond32, steps, seed=2, use_graph=True)
    torch.cuda.synchronize()
    return (time.time() - t0) / steps * 1e3


def run_halves(delay_cycles):
    def work(i, n, seed):
        with torch.cuda.stream(streams[i]):
            if i == 1 and delay_cycles:
                torch.cuda._sleep(delay_cycles)
            halves[i][1].sample(conds[i], n, seed=seed, use_graph=True)
    for i in range(2):
        work(i, 130, 1)
    torch.cuda.synchronize()
    ths = [threading.Thread(target=work, args=(i, steps, 2)) for i in range(2)]
    t0 = time.time()
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    torch.cuda.synchronize()
    return (time.time() - t0) / steps * 1e3


for rep in range(2):
    print("%s  one 32-clip batch                      %.3f ms/step" % (prec, run32()), flush=True)
    for d_us in (0, 20, 40, 60, 90):
        print("%s  two 16-clip halves, 2 streams, delay %3d us  %.3f ms/step (both halves)" % (prec, d_us, run_halves(int(d_us * 2100))), flush=True)
# one half alone: what a 16-clip batch costs by itself
with torch.cuda.stream(streams[0]):
    halves[0][1].sample(conds[0], 130, seed=1, use_graph=True)
    torch.cuda.synchronize(); t0 = time.time()
    halves[0][1].sample(conds[0], steps, seed=2, use_graph=True)
    torch.cuda.synchronize()
    print("%s  one 16-clip half alone                 %.3f ms/step" % (prec, (time.time() - t0) / steps * 1e3), flush=True)
