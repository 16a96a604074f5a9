"""Vocoder operand schemes (cfg.precision of dsvc_vocoder: f16_x3 = split activations and weights, f16_w2 = hi + lo weights with fp16 activations,
f16 = single planes): PCM error against the real-reference goldens and time per 10 s clip.
    python tools/gpu_voc_prec.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import diffsvc_amd
from diffsvc_amd import synth
from diffsvc_amd.engine import VocoderHandle

h = dict(synth.VOCODER_44K)
g = np.load(os.path.join(ROOT, "tests", "golden", "vocoder_44k.npz"))
z = np.load(os.path.join(ROOT, "tests", "golden", "e2e_44k_T861_k1000.npz"))          # the headline golden: reference mel -> reference wav of clip 0
for prec in ("f16_x3", "f16_w2", "f16"):
    voc = VocoderHandle(synth.vocoder_state(h, int(g["wseed"])), h, precision=prec)
    clips = [int(c) for c in g["clips"]]
    errs = []
    for i, c in enumerate(clips):
        w = voc.vocode(torch.from_numpy(g["mel"][i:i + 1]).cuda(), torch.from_numpy(g["f0"][i:i + 1]).cuda(), seed=int(g["seed"]), first_clip=c).cpu()
        ref = torch.from_numpy(g["wav"][i:i + 1])
        errs.append(((w - ref).pow(2).mean().sqrt().item(), (w - ref).abs().max().item(), ref.pow(2).mean().sqrt().item()))
    # 10 s clip timing (861 frames)
    gen = torch.Generator().manual_seed(1)
    mel = (torch.randn(1, 861, 128, generator=gen) * 0.8 - 2.5).cuda()
    f0 = torch.full((1, 861), 220.0).cuda()
    for _ in range(3):
        voc.vocode(mel, f0, seed=1, first_clip=0)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10):
        voc.vocode(mel, f0, seed=1, first_clip=0)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 10 * 1e3
    # the two headline goldens' own vocoder stage: mel_out (reference) -> wav (reference)
    vs = VocoderHandle(synth.vocoder_state(h, int(z["vseed"])), h, precision=prec)
    w = vs.vocode(torch.from_numpy(z["mel_out"][0:1]).cuda(), torch.from_numpy(z["f0_denorm"][0:1]).cuda(), seed=int(z["seed"]), first_clip=int(z["clips"][0])).cpu().numpy()
    e2 = [float(np.sqrt(np.mean((w.reshape(-1) - z["wav0"].reshape(-1)) ** 2)))]
    print("%-7s golden clips: rms err %s  max %s (signal rms %.3f) | %.2f ms per 861-frame clip | e2e goldens' vocoder stage rms %s" % (
        prec, ["%.2e" % e[0] for e in errs], ["%.2e" % e[1] for e in errs], errs[0][2], dt, ["%.2e" % v for v in e2]), flush=True)
