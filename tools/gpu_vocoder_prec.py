"""Vocoder operand precision: waveform RMS against the reference golden and time per 10 s clip.  python tools/gpu_vocoder_prec.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import diffsvc_amd
from diffsvc_amd import synth
from diffsvc_amd.engine import VocoderHandle
from util import load_golden
for prec in ("f16_x3", "f16_w2", "f16"):
    for name in ("vocoder_tiny", "vocoder_44k"):
        g = load_golden(name)
        h = synth.tiny_vocoder() if "tiny" in name else dict(synth.VOCODER_44K)
        voc = VocoderHandle(synth.vocoder_state(h, int(g["wseed"])), h, precision=prec)
        wavs = [voc.vocode(torch.from_numpy(g["mel"][i:i + 1]).cuda(), torch.from_numpy(g["f0"][i:i + 1]).cuda(), seed=int(g["seed"]), first_clip=int(c)).cpu()
                for i, c in enumerate(g["clips"])]
        wav = torch.cat(wavs); ref = torch.from_numpy(g["wav"])
        print("%-7s %-13s rms %.3e  max %.3e  (ref rms %.3f)" % (prec, name, (wav - ref).pow(2).mean().sqrt().item(), (wav - ref).abs().max().item(), ref.pow(2).mean().sqrt().item()), flush=True)
    h = dict(synth.VOCODER_44K)
    voc = VocoderHandle(synth.vocoder_state(h, 1), h, precision=prec)
    for B in (1, 32):
        mel = (torch.randn(B, 861, 128, device="cuda") * 0.5 - 2.5).clamp(-6, 1.5)
        f0 = torch.full((B, 861), 220.0, device="cuda")
        voc.vocode(mel, f0, seed=1); torch.cuda.synchronize(); t0 = time.time()
        for i in range(3): voc.vocode(mel, f0, seed=2 + i)
        torch.cuda.synchronize()
        print("%-7s B=%-2d %.2f ms per call" % (prec, B, (time.time() - t0) / 3 * 1e3), flush=True)
