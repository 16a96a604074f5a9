"""Round 6 (second session): the chunks of one utterance as padded batches (SvcPipeline.infer_chunks) against the reference's one-by-one loop.
python tools/gpu_chunks_probe.py  -> the plan, modelled and measured time per grouping, 1000-step DDPM, the seven chunk lengths of bench.py's `ragged`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import diffsvc_amd
from diffsvc_amd import synth
from diffsvc_amd.pipeline import SvcPipeline
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
RAGGED_T = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (430, 700, 861, 1200, 1600, 2100, 2600)
SPEEDUP = int(sys.argv[3]) if len(sys.argv) > 3 else 1                 # > 1: PLMS with that interval
dev = torch.device("cuda")
hp = dict(synth.HPARAMS_44K, K_step=STEPS)
h = dict(synth.VOCODER_44K)
pipe = SvcPipeline(hp, synth.acoustic_state(hp, 0), synth.vocoder_state(h, 1), h, precision="auto", vocoder_precision="f16_x3")
chunks = []
for i, T in enumerate(RAGGED_T):
    a, b, c, _ = synth.clip_inputs(100 + i, T=T, n_units=max(2, T * 500 // 861), H=256)
    chunks.append(tuple(torch.from_numpy(v).to(dev) for v in (a, b, c)))
audio = sum(RAGGED_T) * 512 / 44100.0
plan = pipe.plan_chunks(list(RAGGED_T), SPEEDUP)
print("plan:", [[RAGGED_T[i] for i in g] for g in plan], flush=True)
def run(batch, groups=None):
    if groups is not None:
        pipe.plan_chunks = lambda lengths, speedup=1: groups
    torch.cuda.synchronize(); t1 = time.perf_counter()
    out = pipe.infer_chunks(chunks, seed=3, first_clip=100, batch=batch, speedup=SPEEDUP)
    torch.cuda.synchronize(); print("   (first call %.3f s)" % (time.perf_counter() - t1), flush=True)
    t0 = time.perf_counter()
    out = pipe.infer_chunks(chunks, seed=4, first_clip=100, batch=batch, speedup=SPEEDUP)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, out
orig = pipe.plan_chunks
t_seq, o_seq = run(False)
print("one by one: %.3f s = %.1fx RT" % (t_seq, audio / t_seq), flush=True)
n = len(RAGGED_T)
desc = sorted(range(n), key=lambda i: -RAGGED_T[i])
alts = [("planned", None), ("all in one batch", [desc])]
if n == 7:
    alts += [("{2600,2100,1600} {1200,861,700,430}", [[6, 5, 4], [3, 2, 1, 0]]), ("{2600,2100,1600} {1200,861} {700,430}", [[6, 5, 4], [3, 2], [1, 0]]),
             ("{2600,2100} {1600,1200,861,700,430}", [[6, 5], [4, 3, 2, 1, 0]])]
elif n >= 4:
    alts += [("two halves", [desc[:n // 2], desc[n // 2:]])]
for name, groups in alts:
    pipe.plan_chunks = orig
    g = groups if groups is not None else plan
    model = sum(pipe._chunk_group_cost(sorted((RAGGED_T[i] for i in x), reverse=True), SPEEDUP) for x in g) * (STEPS if SPEEDUP <= 1 else STEPS // SPEEDUP) * 1e-6
    t, o = run(True, groups)
    err = max(float((a - b).abs().max()) for a, b in zip(o, o_seq))
    print("%-48s %.3f s = %5.1fx RT (modelled sampler time %.3f s); max |PCM - one-by-one| %.2e" % (name, t, audio / t, model, err), flush=True)
