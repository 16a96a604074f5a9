"""Does a clip's output depend on its position in the batch / on what ran before?  (tiny architecture, tests' pipeline)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from diffsvc_amd import synth
from diffsvc_amd.pipeline import SvcPipeline
from util import clip_batch
hp = synth.tiny_hparams(K=20); h = synth.tiny_vocoder(num_mels=16)
sd, vs = synth.acoustic_state(hp, 3), synth.vocoder_state(h, 5)
for prec in ("f16_w2", "f16_x3"):
    pipe = SvcPipeline(hp, sd, vs, h, precision=prec, vocoder_precision="f16_x3")
    clips = [9, 2, 5]
    hub, m2p, f0 = clip_batch(hp, clips, 40, 23)
    ids = torch.tensor(clips, dtype=torch.int32, device="cuda")
    perm = [2, 0, 1]
    for graph in (True, False):
        w1, m1 = pipe.infer(hub.cuda(), m2p.cuda(), f0.cuda(), seed=3, clip_ids=ids, return_mel=True, use_graph=graph)
        w2, m2 = pipe.infer(hub.cuda(), m2p.cuda(), f0.cuda(), seed=3, clip_ids=ids, return_mel=True, use_graph=graph)
        ws, ms = pipe.infer(hub[perm].cuda(), m2p[perm].cuda(), f0[perm].cuda(), seed=3, clip_ids=ids[perm], return_mel=True, use_graph=graph)
        print(prec, "graph", graph, "| rerun: mel", (m1 - m2).abs().max().item(), "wav", (w1 - w2).abs().max().item(),
              "| permuted: mel", (ms - m1[perm]).abs().max().item(), "wav", (ws - w1[perm]).abs().max().item())
        # vocoder alone on the same mel, permuted
        f0d = pipe.model.fs2(hub.cuda(), m2p.cuda(), None, None, f0.clone().cuda(), None, None, infer=True)["f0_denorm"]
        melc = torch.clamp(m1, hp["mel_vmin"], hp["mel_vmax"])
        va = pipe.vocoder.vocode(melc, f0d, seed=3, clip_ids=ids)
        vb = pipe.vocoder.vocode(melc[perm].contiguous(), f0d[perm].contiguous(), seed=3, clip_ids=ids[perm])
        print("   vocoder alone permuted:", (vb - va[perm]).abs().max().item(), "per clip", [(vb[i] - va[perm[i]]).abs().max().item() for i in range(3)])
