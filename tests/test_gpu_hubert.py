"""GPU parity of the content encoder in front of the hot path: HubertSoft.units on the HIP kernels
(dsvc_hubert_*) against the REAL reference module's output (tests/golden/hubert_units.npz) and against the oracle at other lengths."""
import io
import wave

import numpy as np
import pytest
import torch

from diffsvc_amd import synth
import dsvc_oracle as O
from util import load_golden

pytestmark = pytest.mark.gpu

UNIT_TOL = 1e-3           # units are O(1) (std 1.05, max 3.5): the same max-abs bar as the mel


@pytest.fixture(scope="module")
def hubert():
    from diffsvc_amd.hubert import HubertSoftHip
    g = load_golden("hubert_units")
    sd = synth.hubert_state(int(g["wseed"]))
    return HubertSoftHip(sd), sd, g


def test_hubert_units_vs_reference_golden(hubert):
    hb, sd, g = hubert
    for i, n in enumerate(g["lengths"]):
        wav = torch.from_numpy(synth.speech_like_wav(100 + i, int(n))).cuda()
        u = hb.units(wav)
        ref = torch.from_numpy(g["units%d" % i])
        assert tuple(u.shape) == (1,) + tuple(ref.shape)
        err = (u[0].cpu() - ref).abs().max().item()
        print("hubert units, %d samples -> %d frames: max-abs err %.2e" % (int(n), ref.shape[0], err))
        assert err < UNIT_TOL, err


@pytest.mark.parametrize("n", [320, 400, 1999, 24001, 160000])
def test_hubert_units_lengths_vs_oracle(hubert, n):
    """Edge lengths: the shortest clip that yields a frame, odd sample counts (odd / even frame counts through the seven strided
    convs), and the benchmark's 10 s clip (500 frames)."""
    hb, sd, _ = hubert
    wav = synth.speech_like_wav(7, n)
    u = hb.units(torch.from_numpy(wav).cuda())[0].cpu()
    with torch.no_grad():
        ref = O.hubert_units(sd, torch.from_numpy(wav)[None, None])[0]
    assert u.shape == ref.shape and u.shape[0] == hb.frames(n)
    err = (u - ref).abs().max().item()
    print("hubert units, %d samples -> %d frames: max-abs err vs oracle %.2e" % (n, ref.shape[0], err))
    assert err < UNIT_TOL, err
    again = hb.units(torch.from_numpy(wav).cuda())[0].cpu()         # a shorter call after a longer one must not see stale workspace rows
    assert torch.equal(again, u)
    if n <= 24001:                                                  # and a third call at the same length reads its own samples
        wav2 = synth.speech_like_wav(8, n)
        u2 = hb.units(torch.from_numpy(wav2).cuda())[0].cpu()
        with torch.no_grad():
            ref2 = O.hubert_units(sd, torch.from_numpy(wav2)[None, None])[0]
        assert (u2 - ref2).abs().max().item() < UNIT_TOL and not torch.equal(u2, u)


def test_hubertencoder_plugin_contract(tmp_path, hubert):
    """HubertencoderHip as the reference uses Hubertencoder (hubertinfer.py:13-42): first *.pt beside pt_path, encode(path | BytesIO)
    -> np.float32 [T, 256], 22.05 kHz input resampled to 16 kHz, a cached .npy beside the file wins."""
    from diffsvc_amd.hparams import set_hparams
    from diffsvc_amd.hubert import HubertencoderHip
    hb, sd, _ = hubert
    set_hparams(dict(synth.HPARAMS_44K))
    d = tmp_path / "hubert"
    d.mkdir()
    torch.save({"module." + k: v for k, v in sd.items()}, str(d / "hubert_soft.pt"))       # 'module.' prefix as from a DDP-trained file
    enc = HubertencoderHip(str(d / "hubert_soft.pt"))
    sr, n = 16000, 8000
    pcm = (synth.speech_like_wav(3, n) * 32767).astype("<i2")
    path = str(tmp_path / "a.wav")
    with wave.open(path, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr); w.writeframes(pcm.tobytes())
    units = enc.encode(path)
    assert isinstance(units, np.ndarray) and units.dtype == np.float32 and units.shape == (hb.frames(n), 256)
    with torch.no_grad():
        ref = O.hubert_units(sd, torch.from_numpy(pcm.astype(np.float32) / 32768.0)[None, None])[0].numpy()
    assert np.abs(units - ref).max() < UNIT_TOL
    buf = io.BytesIO(open(path, "rb").read())
    assert np.array_equal(enc.encode(buf), units)
    np.save(str(tmp_path / "a.npy"), np.zeros((3, 256), np.float32))
    assert enc.encode(path).shape == (3, 256)
