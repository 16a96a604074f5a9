"""``HubertSoftHip`` / ``HubertencoderHip`` -- the content encoder in front of the hot path on the HIP kernels
(network/hubert/hubert_model.py:67-77 ``HubertSoft.units``; preprocessing/hubertinfer.py:13-42 ``Hubertencoder``).

``HubertencoderHip(pt_path)`` keeps the reference wrapper's contract: it picks the first ``*.pt`` beside ``pt_path``
(hubertinfer.py:22), loads the plain state dict (``module.`` prefix stripped, hubert_model.py:227-229) and
``encode(wav_path_or_BytesIO) -> np.float32 [T, 256]`` (a cached ``<wav>.npy`` beside the file wins, hubertinfer.py:35-38).
The wav is read as in nvSTFT / get_units: mono, resampled to 16 kHz."""
import ctypes
import os
from io import BytesIO
from pathlib import Path

import numpy as np
import torch

from ._lib import check, host_f32, lib, ptr, stream_ptr


class HubertSoftHip:
    """dsvc_hubert: ``units(wav)`` with wav a device tensor [N] / [1,N] / [1,1,N] at 16 kHz -> [1, T, 256] (as the reference)."""

    def __init__(self, state):
        if not torch.cuda.is_available():
            raise RuntimeError("HubertSoftHip needs a HIP device (there is no CPU path)")
        self._h = ctypes.c_void_p(0)
        check(lib().dsvc_hubert_create(ctypes.byref(self._h)))
        for k, v in state.items():
            if k.startswith("module."):
                k = k[len("module."):]
            h, p = host_f32(v)
            check(lib().dsvc_hubert_load_tensor(self._h, k.encode(), p, h.numel()))
        check(lib().dsvc_hubert_finalize(self._h))

    @staticmethod
    def frames(n_samples):
        t = ctypes.c_int32(0)
        check(lib().dsvc_hubert_frames(int(n_samples), ctypes.byref(t)))
        return t.value

    def units(self, wav):
        if not wav.is_cuda:
            raise RuntimeError("diffsvc_amd: the waveform must live on the HIP device; there is no CPU path")
        w = wav.reshape(-1).contiguous().float()
        T = self.frames(w.numel())
        out = torch.empty(1, T, 256, device=w.device, dtype=torch.float32)
        check(lib().dsvc_hubert_units(self._h, ptr(w), w.numel(), ptr(out), stream_ptr()))
        return out

    __call__ = units

    def __del__(self):
        try:
            if self._h:
                lib().dsvc_hubert_destroy(self._h)
                self._h = ctypes.c_void_p(0)
        except Exception:
            pass


class HubertencoderHip:
    def __init__(self, pt_path="checkpoints/hubert/hubert_soft.pt"):
        from .hparams import get_hparams
        if get_hparams().get("use_vec"):
            raise NotImplementedError("ContentVec (use_vec) goes through fairseq in the reference and is not part of this path")
        cands = list(Path(pt_path).parent.rglob("*.pt"))
        if not cands:
            raise FileNotFoundError("no *.pt beside %s" % pt_path)
        self.pt_path = str(cands[0])
        self.hbt_model = HubertSoftHip(torch.load(self.pt_path, map_location="cpu"))

    def encode(self, wav_path):
        from .vocoder import read_wav
        if isinstance(wav_path, BytesIO):
            wav_path.seek(0)
        else:
            npy = Path(wav_path).with_suffix(".npy")
            if os.path.exists(npy):
                return np.load(str(npy))
        wav16 = read_wav(wav_path, 16000, mono="mean")                           # librosa.load(path, sr=16000) (hubert_model.py:236)
        return self.hbt_model.units(torch.from_numpy(wav16).cuda())[0].cpu().numpy()
