"""``SvcHip`` / ``run_clip`` -- the reference's inference driver (infer_tools/infer_tool.py:104-278 ``Svc``, infer.py:17-71 ``run_clip``)
over the native components: slicer -> per chunk [wav2spec, HuBERT-soft units, f0, get_align] -> GaussianDiffusionHip -> (PitchExtractorHip)
-> after_infer -> vocoder -> the chunks stitched back to the input's length.

This is host glue, kept close to the reference so that its knobs mean the same (key, acc, use_pe, use_gt_mel, add_noise_step, thre,
slice_db).  What it does NOT bring is a pitch tracker: the reference calls torchcrepe or parselmouth (third-party, absent here), so
``f0_fn(wav, mel, hparams) -> f0_hz [T]`` is a constructor argument; without one, and with a pitch-extractor checkpoint loaded, the
f0 is read off the INPUT mel by ``PitchExtractorHip`` (not what the reference does -- stated, not hidden).  The md5-keyed JSON caches
of chunks and f0 (infer_tool.py:31-52, infer.py:31-39) are a convenience of the reference's CLI and are not reproduced."""
import io
import os
import re
import wave
from pathlib import Path

import numpy as np
import torch

from .hparams import get_hparams
from .slicer import chunks_of, cut_samples
from .synth import align_units


def load_ckpt(cur_model, ckpt_base_dir, prefix_in_ckpt="model", force=True, strict=True):
    """utils/__init__.py:178-209: a file, or a directory whose highest ``model_ckpt_steps_<N>.ckpt`` wins; the ``state_dict`` entries
    under ``<prefix>.`` are loaded (strictly by default); a missing checkpoint is an assertion when ``force``."""
    if os.path.isfile(ckpt_base_dir):
        path = ckpt_base_dir
    else:
        found = [p for p in Path(ckpt_base_dir).glob("model_ckpt_steps_*.ckpt") if re.search(r"steps_(\d+)\.ckpt$", p.name)]
        path = str(max(found, key=lambda p: int(re.search(r"steps_(\d+)\.ckpt$", p.name).group(1)))) if found else None
    if path is None:
        msg = "| ckpt not found in %s." % ckpt_base_dir
        assert not force, msg
        print(msg)
        return None
    sd = torch.load(path, map_location="cpu")["state_dict"]
    pre = prefix_in_ckpt + "."
    cur_model.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}, strict=strict)
    print("| load '%s' from '%s'." % (prefix_in_ckpt, path))
    return path


def norm_interp_f0(f0_hz, hp):
    """utils/pitch_utils.py:45-60 (+ norm_f0 :33-42, pitch_norm 'log'): log2, unvoiced frames interpolated; returns (f0, uv) tensors."""
    f0 = np.asarray(f0_hz, dtype=np.float64 if np.asarray(f0_hz).dtype == np.float64 else np.float32).copy()
    uv = f0 == 0
    if hp["pitch_norm"] != "log":
        raise NotImplementedError("pitch_norm must be 'log'")
    with np.errstate(divide="ignore"):
        f0 = np.log2(f0)
    if hp.get("use_uv"):
        f0[uv] = 0
    if uv.sum() == len(f0):
        f0[uv] = 0
    elif uv.sum() > 0:
        f0[uv] = np.interp(np.where(uv)[0], np.where(~uv)[0], f0[~uv])
    return torch.FloatTensor(f0), torch.FloatTensor(uv)


def _wav_bytes(samples, sr):
    """What run_clip hands a chunk to ``Svc.infer`` as: an in-memory PCM-16 wav (soundfile.write(..., format='wav'), infer.py:49-50)."""
    pcm = np.clip(np.rint(np.asarray(samples, dtype=np.float64) * 32767.0), -32768, 32767).astype("<i2")
    buf = io.BytesIO()
    with wave.open(buf, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(int(sr))
        w.writeframes(pcm.tobytes())
    buf.seek(0)
    return buf


class SvcHip:
    """``model``: a loaded GaussianDiffusionHip; ``vocoder``: NsfHifiGANHip / HifiGANHip (``wav2spec`` + ``spec2wav``); ``hubert``: an
    object with ``encode(path | BytesIO) -> [T_h, 256]`` (HubertencoderHip); ``pe``: a loaded PitchExtractorHip or None."""

    def __init__(self, project_name, model, vocoder, hubert, pe=None, f0_fn=None, hparams=None):
        self.project_name = project_name
        self.hp = hparams if hparams is not None else get_hparams()
        self.model, self.vocoder, self.hubert, self.pe, self.f0_fn = model, vocoder, hubert, pe, f0_fn
        self.mel_bins = self.hp["audio_num_mel_bins"]

    # ---- Svc.pre -> temporary_dict2processed_input -> getitem -> processed_input2batch (infer_tool.py:203-335), one item ----
    def pre(self, wav_fn, accelerate):
        hp = self.hp
        wav, mel = self.vocoder.wav2spec(wav_fn)
        if isinstance(wav_fn, io.BytesIO):
            wav_fn.seek(0)
        mel = np.asarray(mel)[:hp["max_frames"]]
        if self.f0_fn is not None:
            f0_hz = np.asarray(self.f0_fn(wav, mel, hp), dtype=np.float32)[:len(mel)]
        elif self.pe is not None:
            self.pe.hp = hp
            f0_hz = self.pe(torch.from_numpy(mel)[None].cuda())["f0_denorm_pred"][0].cpu().numpy()
        else:
            raise RuntimeError("SvcHip needs an f0 tracker (f0_fn=...) or a pitch-extractor checkpoint: torchcrepe / parselmouth, which the "
                               "reference calls here, are not part of this package")
        units = np.asarray(self.hubert.encode(wav_fn))[:hp["max_input_tokens"]]
        mel2ph = align_units(mel.shape[0], units.shape[0])                       # get_align (infer_tool.py:231-242)
        f0, uv = norm_interp_f0(f0_hz, hp)
        hp["pndm_speedup"] = accelerate
        return {"hubert": torch.from_numpy(units).float()[None], "mels": torch.from_numpy(mel).float()[None], "f0": f0[None], "uv": uv[None],
                "mel2ph": torch.from_numpy(mel2ph)[None]}

    # ---- Svc.infer + after_infer (infer_tool.py:143-200) ----
    @torch.no_grad()
    def infer(self, in_path, key, acc, use_pe=True, use_crepe=True, thre=0.05, singer=False, **kwargs):
        hp = self.hp
        batch = self.pre(in_path, acc)
        f0 = batch["f0"] + (key / 12)
        f0[f0 > np.log2(hp["f0_max"])] = 0
        self.model.hp = hp
        self.model.fs2.hp = hp
        out = self.model(batch["hubert"].cuda(), mel2ph=batch["mel2ph"].cuda(), f0=f0.clone().cuda(), uv=batch["uv"].cuda(),
                         ref_mels=batch["mels"].cuda(), infer=True, **kwargs)
        mel_out = self.model.out2mel(out["mel_out"])
        f0_gt = 2 ** f0                                                          # denorm_f0(batch['f0'], batch['uv'], hparams)
        if hp.get("use_uv"):
            f0_gt[batch["uv"] > 0] = 0
        if use_pe:
            if self.pe is None:
                raise RuntimeError("use_pe=True needs a pitch-extractor checkpoint")
            self.pe.hp = hp
            f0_pred = self.pe(out["mel_out"])["f0_denorm_pred"]
        else:
            f0_pred = out["f0_denorm"]
        mel_gt = batch["mels"][0].numpy()
        mel_pred = mel_out[0].cpu().numpy()
        gt_mask = np.abs(mel_gt).sum(-1) > 0
        pred_mask = np.abs(mel_pred).sum(-1) > 0
        mel_pred = np.clip(mel_pred[pred_mask], hp["mel_vmin"], hp["mel_vmax"])
        f0_gt = f0_gt[0].cpu().numpy()[gt_mask]
        f0_pred = f0_pred[0].cpu().numpy()
        f0_pred = f0_pred[:len(pred_mask)][pred_mask]
        if singer:
            from .formats import save_singer_features
            save_singer_features(in_path, mel_pred, f0_pred)
        wav_pred = self.vocoder.spec2wav(mel_pred, f0=f0_pred)
        return f0_gt, f0_pred, wav_pred


def run_clip(svc_model, key, acc, use_pe, use_crepe, thre, use_gt_mel, add_noise_step, file_path=None, audio=None, sr=None, out_path=None,
             slice_db=-40, **kwargs):
    """infer.py:17-71: slice at silences, convert every voiced chunk, keep silent chunks as zeros, force every chunk back to its input
    duration (pad with the chunk's mean / cut from the front), write PCM-16.  ``audio``/``sr`` may be given instead of ``file_path``."""
    hp = svc_model.hp
    use_pe = use_pe if hp["audio_sample_rate"] == 24000 else False
    if audio is None:
        from .vocoder import read_wav
        with wave.open(file_path, "rb") as w:
            sr = w.getframerate()
        audio = read_wav(file_path, sr)
    chunks = cut_samples(audio, sr, db_thresh=slice_db)
    f0_tst, f0_pred, out = [], [], []
    model_sr, hop = hp["audio_sample_rate"], hp["hop_size"]
    for slice_tag, data in chunks_of(chunks, np.asarray(audio)):
        length = int(np.ceil(len(data) / sr * model_sr))
        if slice_tag:
            n = int(np.ceil(length / hop))
            _f0_tst, _f0_pred, _audio = np.zeros(n), np.zeros(n), np.zeros(length)
        else:
            _f0_tst, _f0_pred, _audio = svc_model.infer(_wav_bytes(data, sr), key=key, acc=acc, use_pe=use_pe, use_crepe=use_crepe, thre=thre,
                                                        use_gt_mel=use_gt_mel, add_noise_step=add_noise_step)
        fix = np.zeros(length)
        fix[:] = np.mean(_audio)
        fix[:len(_audio)] = _audio[0 if len(_audio) < len(fix) else len(_audio) - len(fix):]
        f0_tst.extend(_f0_tst); f0_pred.extend(_f0_pred); out.extend(list(fix))
    if out_path is not None:
        pcm = np.clip(np.rint(np.asarray(out, dtype=np.float64) * 32767.0), -32768, 32767).astype("<i2")
        with wave.open(out_path, "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(int(model_sr))
            w.writeframes(pcm.tobytes())
    return np.array(f0_tst), np.array(f0_pred), out
