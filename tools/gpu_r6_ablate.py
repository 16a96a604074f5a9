"""Round 6: what the two byte / fixed-cost levers of VERDICT r5 "next 1" could buy on the batched fused layer kernel, measured as ablations of the
REAL kernel in the profiling build (python -m diffsvc_amd.build --profiling; env DSVC_TL_ABL, csrc/tlayer.h):

  abl 1  the gate phase's accumulator init reads HALF of cproj (1536 instead of 3072 B per frame; results WRONG): an upper bound on what cproj as
         fp16 hi + 6-bit lo codes (2112 B per frame) can buy -- it removes MORE bytes and decodes nothing;
  abl 2  the neighbour hand-off a persistent per-evaluation launch needs (poll tiles i-1 / i+1, agent acquire; drain, agent release, flag), as pure
         overhead inside the real kernel (results unchanged): the COST side of that design; its BENEFIT side is bounded by the 1.75 us graph-node
         boundary + the part of the ~9 us prologue that does not depend on the neighbours (profiles/r3g_layer_stamps.txt).

python tools/gpu_r6_ablate.py [clips] -> us per fused layer launch (HIP events over 20 back-to-back layers) and ms per DDPM step (graph replay)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import diffsvc_amd
from diffsvc_amd import _lib
_lib.use_profiling_build()
from diffsvc_amd import synth
from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
hp = dict(synth.HPARAMS_44K)
sd = synth.acoustic_state(hp, 0)
cond = torch.randn(B, 256, 861, device="cuda") * 0.5
names = {0: "product kernel", 1: "half of cproj's bytes", 2: "neighbour hand-off protocol as overhead", 3: "both",
         10: "residual / skip tiles with plain (cacheable) accesses", 13: "3/4 of the residual / skip bytes (3 B per element)"}
# second session of round 6 (python tools/gpu_r6_ablate.py 32 stream): DSVC_TL_STREAM = 0 plain loads / stores for the fp32 residual and skip tiles
# (can the 256 MB Infinity Cache hold the 132 MB a 32-clip batch re-touches every layer?), 3 = a pure byte ablation, 3 of the 4 dwordx4 per lane and tile
# (WRONG results): an upper bound on what 3-byte residual / skip elements (fp32 rounded to 16 significant bits) could buy
variants = (0, 10, 13) if (len(sys.argv) > 2 and sys.argv[2] == "stream") else (0, 1, 2, 3)
# python tools/gpu_r6_ablate.py 32 dephase: DSVC_TL_DEPHASE = n -> after "g complete" waves 4-7 (n > 0) / 0-3 (n < 0) sleep |n| x 1024 clocks (results unchanged):
# do the two waves of a SIMD, out of step, hide each other's store / accumulator-init latency in the output phase?  (100 + n encodes n; 113 = with the byte ablation)
if len(sys.argv) > 2 and sys.argv[2] == "dephase":
    variants = (0, 104, 108, 112, 96, 92, 213)
    names.update({104: "dephase +4", 108: "dephase +8", 112: "dephase +12", 96: "dephase -4", 92: "dephase -8", 213: "dephase +8 and 3/4 of the residual / skip bytes"})
for rep in range(2):
    for abl in variants:
        os.environ["DSVC_TL_ABL"] = str(abl if abl < 10 else 0)
        os.environ.pop("DSVC_TL_STREAM", None); os.environ.pop("DSVC_TL_DEPHASE", None)
        if 10 <= abl < 50: os.environ["DSVC_TL_STREAM"] = str(abl - 10)
        if 50 <= abl < 200: os.environ["DSVC_TL_DEPHASE"] = str(abl - 100)
        if abl == 213: os.environ["DSVC_TL_DEPHASE"] = "8"; os.environ["DSVC_TL_STREAM"] = "3"
        den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision="f16_w6", prefix="denoise_fn.")      # (a fresh handle: the captured graph bakes the knob)
        smp = SamplerHandle(den, sd)
        us, rows, kind = smp.profile_gate_kernel(B, 861, 3)
        smp.sample(cond, 1000, seed=1, t_stop=1000 - 140, use_graph=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        smp.sample(cond, 1000, seed=2, t_stop=1000 - 330, use_graph=True)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 330 * 1e3
        print("B=%d abl=%d %-42s fused layer %7.1f us per launch (tile width %d), %.3f ms per DDPM step" % (B, abl, names[abl], us, 32 * kind, ms), flush=True)
        del smp, den
