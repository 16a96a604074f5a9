"""Where the time of a batched ragged PLMS call goes (round 6, third session): python tools/gpu_chunks_stages.py [speedup]
The seven chunks of bench.py's `ragged` as ONE padded batch through SvcPipeline.infer: acoustic model (cond builder + sampler), the ragged vocoder
(one call per distinct kept length), glue -- each bracketed by a device synchronisation; and the vocoder on the same chunks as equal-length batches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import diffsvc_amd
from diffsvc_amd import synth
from diffsvc_amd.pipeline import SvcPipeline
SPEEDUP = int(sys.argv[1]) if len(sys.argv) > 1 else 20
LENS = (430, 700, 861, 1200, 1600, 2100, 2600)
dev = torch.device("cuda")
hp = dict(synth.HPARAMS_44K, K_step=1000)
h = dict(synth.VOCODER_44K)
pipe = SvcPipeline(hp, synth.acoustic_state(hp, 0), synth.vocoder_state(h, 1), h, precision="auto", vocoder_precision="f16_x3")
chunks = []
for i, T in enumerate(LENS):
    a, b, c, _ = synth.clip_inputs(100 + i, T=T, n_units=max(2, T * 500 // 861), H=256)
    chunks.append(tuple(torch.from_numpy(v).to(dev) for v in (a, b, c)))
pipe.plan_chunks = lambda lengths, speedup=1: [sorted(range(len(LENS)), key=lambda i: -LENS[i])]
marks = {}
def timed(name, fn):
    def wrap(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); marks[name] = marks.get(name, 0.0) + time.perf_counter() - t0
        return r
    return wrap
pipe.model.forward = timed("acoustic model (cond + sampler)", pipe.model.forward)
pipe._vocode_ragged = timed("ragged vocoder incl. its glue", pipe._vocode_ragged)
voc = pipe.vocoder.vocode
pipe.vocoder.vocode = timed("  vocoder calls alone", voc)
for rep in range(3):
    marks.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = pipe.infer_chunks(chunks, seed=3, first_clip=100, speedup=SPEEDUP)
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
    print("pass %d: total %.1f ms; %s" % (rep, tot * 1e3, "; ".join("%s %.1f ms" % (k, v * 1e3) for k, v in marks.items())), flush=True)
# the vocoder on equal-length batches: what a ragged vocoder batch could cost at best
for T, B in ((861, 1), (861, 7), (2600, 1), (2600, 4), (1356, 7)):
    mel = torch.randn(B, T, 128, device=dev).clamp(-4, 1); f0 = torch.full((B, T), 220.0, device=dev)
    voc(mel, f0, seed=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        voc(mel, f0, seed=0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("vocoder B=%d T=%d: %.2f ms = %.3f ms per second of audio" % (B, T, dt * 1e3, dt * 1e3 / (B * T * 512 / 44100.0)), flush=True)
