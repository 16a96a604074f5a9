"""``NsfHifiGANHip`` -- drop-in vocoder plugin for ``network.vocoders.nsf_hifigan.NsfHifiGAN``
(nsf_hifigan.py:8-92).  Select it with one YAML line, no reference edits (base_vocoder.py:11-19):

    vocoder: diffsvc_amd.vocoder.NsfHifiGANHip

Contract kept: no-arg constructor reading ``hparams['vocoder_ckpt']`` (+ sibling config.json, models.py:14-21),
``spec2wav(mel[T,M] log10, f0=[T] Hz) -> np.float32[T*hop]``, static ``wav2spec(path) -> (wav, mel[T,M] log10)``,
and registration under the bare class name (infer_tool.py:244-247 looks it up through VOCODERS).
The generator, the harmonic source and the STFT/mel run in libdsvc_hip.so.
"""
import json
import os
import wave

import numpy as np
import torch

from .engine import MelspecHandle, VocoderHandle
from .hparams import get_hparams

try:                                                       # inside the reference tree: use its registry
    from network.vocoders.base_vocoder import BaseVocoder, register_vocoder
except Exception:                                          # standalone
    class BaseVocoder:                                     # noqa: D401
        pass

    def register_vocoder(cls):
        return cls

_melspec_cache = {}


def load_generator_checkpoint(model_path):
    """(generator state dict with weight-norm pairs, config dict) -- the format of models.py:14-26."""
    config_file = os.path.join(os.path.split(model_path)[0], "config.json")
    with open(config_file) as f:
        h = json.loads(f.read())
    cp = torch.load(model_path, map_location="cpu")
    return cp["generator"], h


def read_wav(path_or_file, target_sr, mono="first"):
    """int PCM -> [-1, 1) by the type's magnitude, resampled to the model rate.  ``mono='first'`` keeps channel 0 -- what the 44.1 kHz
    nvSTFT loader does (nvSTFT.py:14-44); ``mono='mean'`` averages the channels -- what ``librosa.load(mono=True)`` does on the 24 kHz
    front-end (data_gen_utils.py:106) and in ``get_units`` (hubert_model.py:234-247).  Stereo input differs between the two."""
    if mono not in ("first", "mean"):
        raise ValueError("mono must be 'first' or 'mean'")
    try:
        import soundfile as sf
        data, sr = sf.read(path_or_file, always_2d=True)
        data = data.astype(np.float32)
    except ImportError:
        with wave.open(path_or_file, "rb") as w:
            sr, nch, width, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
            raw = w.readframes(n)
        if width != 2:
            raise RuntimeError("only 16-bit PCM wav is supported without the soundfile package")
        data = np.frombuffer(raw, dtype="<i2").reshape(-1, nch).astype(np.float32) / 32768.0
    data = data[:, 0] if mono == "first" or data.shape[1] == 1 else data.mean(axis=1, dtype=np.float32)
    data = np.ascontiguousarray(data, dtype=np.float32)
    if sr != target_sr:
        data = resample(data, sr, target_sr)
    return data


def resample(data, sr, target_sr):
    """The reference resamples with ``librosa.resample`` (nvSTFT.py:38-40).  librosa when it is importable (identical to the
    reference then); otherwise a polyphase FIR resampler (scipy.signal.resample_poly) -- same band-limited signal, not the same
    samples bit for bit, so a one-line notice is printed."""
    try:
        import librosa
        return librosa.resample(data, orig_sr=sr, target_sr=target_sr).astype(np.float32)
    except ImportError:
        pass
    try:
        from math import gcd
        from scipy.signal import resample_poly
    except ImportError:
        raise RuntimeError("wav2spec: file is %d Hz, model expects %d Hz and neither librosa nor scipy is available to resample" % (sr, target_sr))
    g = gcd(int(sr), int(target_sr))
    print("| wav2spec: resampling %d -> %d Hz with scipy.signal.resample_poly (librosa, which the reference uses, is not installed)" % (sr, target_sr))
    return resample_poly(data.astype(np.float64), int(target_sr) // g, int(sr) // g).astype(np.float32)


@register_vocoder
class NsfHifiGANHip(BaseVocoder):
    def __init__(self, device=None, precision="f16_x3"):
        if not torch.cuda.is_available():
            raise RuntimeError("NsfHifiGANHip needs a HIP device (there is no CPU path)")
        self.device = device or "cuda"
        hp = get_hparams()
        model_path = hp["vocoder_ckpt"]
        if not os.path.exists(model_path):
            raise FileNotFoundError("HifiGAN model file is not found: %s" % model_path)
        print("| Load HifiGAN (HIP): ", model_path)
        state, self.h = load_generator_checkpoint(model_path)
        self.model = VocoderHandle(state, self.h, precision=precision)
        self.seed = 0

    def _warn_mismatch(self):
        hp = get_hparams()
        for a, b in (("sampling_rate", "audio_sample_rate"), ("num_mels", "audio_num_mel_bins"), ("n_fft", "fft_size"),
                     ("win_size", "win_size"), ("hop_size", "hop_size"), ("fmin", "fmin"), ("fmax", "fmax")):
            if a in self.h and b in hp and self.h[a] != hp[b]:
                print("Mismatch parameters: hparams['%s']=" % b, hp[b], "!=", self.h[a], "(vocoder)")

    def spec2wav_torch(self, mel, **kwargs):          # mel [B, T, bins] device tensor -> flat device tensor
        self._warn_mismatch()
        f0 = kwargs.get("f0")
        if f0 is None or not get_hparams().get("use_nsf"):
            raise NotImplementedError("the NSF generator needs f0 (use_nsf: true)")
        self.seed += 1
        return self.model.vocode(mel.to(self.device), f0.to(self.device), seed=kwargs.get("seed", self.seed)).view(-1)

    def spec2wav(self, mel, **kwargs):
        self._warn_mismatch()
        f0 = kwargs.get("f0")
        if f0 is None or not get_hparams().get("use_nsf"):
            raise NotImplementedError("the NSF generator needs f0 (use_nsf: true)")
        self.seed += 1
        c = torch.FloatTensor(np.asarray(mel)).unsqueeze(0).to(self.device)
        f = torch.FloatTensor(np.asarray(f0)[None, :]).to(self.device)
        y = self.model.vocode(c, f, seed=kwargs.get("seed", self.seed), first_clip=kwargs.get("first_clip", 0)).view(-1)
        return y.cpu().numpy()

    @staticmethod
    def wav2spec(inp_path, device=None):
        hp = get_hparams()
        key = tuple(hp[k] for k in ("audio_sample_rate", "fft_size", "win_size", "hop_size", "audio_num_mel_bins", "fmin", "fmax"))
        if key not in _melspec_cache:
            _melspec_cache[key] = MelspecHandle(*key)
        wav = read_wav(inp_path, hp["audio_sample_rate"])
        mel = _melspec_cache[key].mel(torch.from_numpy(wav)[None].cuda())[0]
        return wav, mel.cpu().numpy()


def _load_yaml_chain(path, _seen=None):
    """A checkpoint directory's config.yaml with its ``base_config`` chain resolved the way utils/hparams.py:40-60 does."""
    import yaml
    _seen = _seen or set()
    with open(path, encoding="utf-8") as f:
        cfg = yaml.safe_load(f) or {}
    out = {}
    base = cfg.get("base_config")
    if base:
        for b in (base if isinstance(base, list) else [base]):
            if b.startswith("."):
                b = os.path.normpath(os.path.join(os.path.dirname(path), b))
            if b not in _seen and os.path.exists(b):
                _seen.add(b)
                out.update(_load_yaml_chain(b, _seen))
    out.update({k: v for k, v in cfg.items() if k != "base_config"})
    return out


@register_vocoder
class HifiGANHip(BaseVocoder):
    """Drop-in for ``network.vocoders.hifigan.HifiGAN`` (hifigan.py:46-76), the 24 kHz vocoder of the demo config
    (training/config.yaml:342-343): ``vocoder: diffsvc_amd.vocoder.HifiGANHip``.  ``hparams['vocoder_ckpt']`` is a DIRECTORY holding
    ``config.yaml`` + ``model_ckpt_steps_<N>.ckpt`` (``['state_dict']['model_gen']``, highest N wins) or ``config.json`` +
    ``generator_v1`` (``['generator']``).  ``HifiGanGenerator`` (modules/hifigan/hifigan.py:104-178) is the NSF-HiFiGAN network with
    80 mel bins, natural-log mels fed unscaled, and the harmonic source only when the config says ``use_pitch_embed`` AND the call
    carries an f0 with ``hparams['use_nsf']`` (hifigan.py:66-71) -- it runs on the same kernels (dsvc_vocoder, mel_scale 1)."""

    def __init__(self, device=None, precision="f16_x3"):
        import glob
        import re
        if not torch.cuda.is_available():
            raise RuntimeError("HifiGANHip needs a HIP device (there is no CPU path)")
        self.device = device or "cuda"
        self.precision = precision
        base_dir = get_hparams()["vocoder_ckpt"]
        yml, jsn = os.path.join(base_dir, "config.yaml"), os.path.join(base_dir, "config.json")
        if os.path.exists(yml):
            files = [f for f in glob.glob(os.path.join(base_dir, "model_ckpt_steps_*.*")) if re.search(r"model_ckpt_steps_(\d+)", f)]
            if not files:
                raise FileNotFoundError("no model_ckpt_steps_*.ckpt in %s" % base_dir)
            file_path = sorted(files, key=lambda x: int(re.findall(r"model_ckpt_steps_(\d+)", x.replace("\\", "/"))[-1]))[-1]
            print("| load HifiGAN (HIP): ", file_path)
            if os.path.splitext(file_path)[-1] != ".ckpt":
                raise NotImplementedError("only .ckpt state-dict checkpoints are supported (a pickled .pth module cannot be repacked)")
            self.config = _load_yaml_chain(yml)
            self.state = torch.load(file_path, map_location="cpu")["state_dict"]["model_gen"]
        elif os.path.exists(jsn):
            with open(jsn, encoding="utf-8") as f:
                self.config = json.load(f)
            self.state = torch.load(os.path.join(base_dir, "generator_v1"), map_location="cpu")["generator"]
        else:
            raise FileNotFoundError("HifiGAN config not found in %s (config.yaml or config.json)" % base_dir)
        self.h = dict(self.config, num_mels=80)                                  # Conv1d(80, ...) is hard-coded (hifigan.py:117)
        self.has_source = bool(self.config.get("use_pitch_embed"))
        self._handles = {}
        self.seed = 0

    def _handle(self, with_source):
        if with_source not in self._handles:
            self._handles[with_source] = VocoderHandle(self.state, self.h, precision=self.precision, mel_scale=1.0, use_source=with_source)
        return self._handles[with_source]

    def spec2wav(self, mel, **kwargs):
        hp = get_hparams()
        if hp.get("vocoder_denoise_c", 0.0) > 0:
            raise NotImplementedError("vocoder_denoise_c > 0 (spectral-subtraction denoiser, vocoder_utils.py:7-15) is not part of this path")
        f0 = kwargs.get("f0")
        with_source = f0 is not None and bool(hp.get("use_nsf")) and self.has_source
        if f0 is not None and hp.get("use_nsf") and not self.has_source:
            raise RuntimeError("use_nsf with an f0, but this generator was built without use_pitch_embed")
        self.seed += 1
        c = torch.FloatTensor(np.asarray(mel)).unsqueeze(0).to(self.device)      # natural-log mel, NOT rescaled (hifigan.py:64)
        f = torch.FloatTensor(np.asarray(f0)[None, :]).to(self.device) if with_source else None
        y = self._handle(with_source).vocode(c, f, seed=kwargs.get("seed", self.seed), first_clip=kwargs.get("first_clip", 0)).view(-1)
        return y.cpu().numpy()

    @staticmethod
    def wav2spec(wav_fn, return_linear=False):
        """PWG.wav2spec (network/vocoders/pwg.py:106-122) -> process_utterance (preprocessing/data_gen_utils.py:96-145): the file at the
        model's rate, a centred zero-padded STFT, |X| through the mel filterbank, log10(max(eps, .)); the waveform comes back
        zero-padded to frames * hop.  Returns (wav [T*hop], mel [T, num_mels]) -- and, ``return_linear=True``, the normalised dB spectrogram
        [T, fft_size / 2 + 1] as third element (data_gen_utils.py:144-149: audio.normalize(audio.amp_to_db(|X|)) with hparams['min_level_db'])."""
        hp = get_hparams()
        sr, hop = hp["audio_sample_rate"], hp["hop_size"]
        fmin = 0 if hp["fmin"] == -1 else hp["fmin"]
        fmax = sr / 2 if hp["fmax"] == -1 else hp["fmax"]
        eps = float(hp.get("wav2spec_eps", 1e-10))
        key = ("pwg", sr, hp["fft_size"], hp["win_size"], hop, hp["audio_num_mel_bins"], fmin, fmax, eps)
        if key not in _melspec_cache:
            _melspec_cache[key] = MelspecHandle(sr, hp["fft_size"], hp["win_size"], hop, hp["audio_num_mel_bins"], fmin, fmax, clip_val=eps, mode=1)
        wav = read_wav(wav_fn, sr, mono="mean")                                  # librosa.core.load(wav_path, sr=...) averages channels
        if hp.get("loud_norm"):                                                  # data_gen_utils.py:117-122: to -22 LUFS (BS.1770 as pyloudnorm 0.1.0
            from .loudness import loud_norm                                      # implements it, restated in loudness.py: the package is absent here)
            wav = loud_norm(wav, sr, -22.0).astype(np.float32)
        lin = None
        if return_linear:
            mel, lin = _melspec_cache[key].mel_and_linear(torch.from_numpy(wav)[None].cuda(), hp.get("min_level_db", -100))      # (process_utterance's default)
            mel, lin = mel[0].cpu().numpy(), lin[0].cpu().numpy()
        else:
            mel = _melspec_cache[key].mel(torch.from_numpy(wav)[None].cuda())[0].cpu().numpy()
        n = mel.shape[0] * hop                                                   # librosa_pad_lr(..., 1) then wav[:T * hop]
        wav = np.pad(wav, (0, max(0, n - len(wav))))[:n]
        return (wav, mel, lin) if return_linear else (wav, mel)
