"""One-by-one serving over MANY chunk lengths (the reference's loop over a long song): 18 buckets of 128 rows cover the slicer's 5 ... 30 s chunks.
python tools/gpu_lru_probe.py [speedup] -> per pass: seconds, captures, buckets built.  A pass after the first must build and capture nothing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import diffsvc_amd
from diffsvc_amd import synth
from diffsvc_amd.pipeline import SvcPipeline
speedup = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda")
hp = dict(synth.HPARAMS_44K, K_step=1000 if speedup > 1 else 200)
h = dict(synth.VOCODER_44K)
pipe = SvcPipeline(hp, synth.acoustic_state(hp, 0), synth.vocoder_state(h, 1), h, precision="auto", vocoder_precision="f16_x3")
rng = np.random.default_rng(1)
Ts = [int(t) for t in rng.permutation(np.arange(430, 2601, 128))] + [int(t) for t in rng.integers(430, 2601, size=12)]      # 17 + 12 chunks, 17+ buckets
chunks = []
for i, T in enumerate(Ts):
    a, b, c, _ = synth.clip_inputs(100 + i, T=T, n_units=max(2, T * 500 // 861), H=256)
    chunks.append(tuple(torch.from_numpy(v[None]).to(dev) for v in (a, b, c)))
audio = sum(Ts) * 512 / 44100.0
smp = pipe.model._handle("plms" if speedup > 1 else "ddpm", speedup, frames=Ts[0], clips=1)
for p in range(3):
    s0 = smp.stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i, ch in enumerate(chunks):
        pipe.infer(*ch, speedup=speedup, seed=40 + i, first_clip=100 + i, full_length=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s1 = smp.stats()
    print("pass %d: %d chunks (%d buckets), %.0f s of audio in %.3f s = %.1fx RT; captures %d, buckets built %d, graphs alive %d" % (
        p, len(Ts), len({(T + 8 + 127) // 128 for T in Ts}), audio, dt, audio / dt,
        s1["capture_ddpm"] + s1["capture_plms"] - s0["capture_ddpm"] - s0["capture_plms"], s1["buckets_allocated"] - s0["buckets_allocated"], s1["graphs_alive"]), flush=True)
