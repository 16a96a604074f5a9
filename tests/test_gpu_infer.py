"""GPU: the native components chained wav in -> wav out (mel front-end, HuBERT-soft, cond builder, sampler, pitch extractor, vocoder) on
synthetic checkpoints.  Parity of every component is pinned elsewhere (and the reference's own driver is exercised over the drop-ins in
tests/test_reference_seams.py, where the reference tree exists); this file only checks that the pieces fit together on the device.
The driver below is test-local: the product package ships the components and the three plugin seams, not a copy of the reference's
``Svc`` / ``run_clip`` host glue (kept as-is in the reference)."""
import io
import wave

import numpy as np
import pytest
import torch

from diffsvc_amd import synth

pytestmark = pytest.mark.gpu


def wav_bytes(samples, sr):
    buf = io.BytesIO()
    with wave.open(buf, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(int(sr))
        w.writeframes(np.clip(np.rint(samples * 32767.0), -32768, 32767).astype("<i2").tobytes())
    buf.seek(0)
    return buf


def convert(parts, wav_file, f0_hz_of, speedup, seed, pe=None, **model_kw):
    """One utterance through the chain.  parts = (hparams, model, vocoder, hubert); f0_hz_of(n_frames) -> the input pitch track."""
    hp, model, vocoder, hubert = parts
    wav, mel = vocoder.wav2spec(wav_file)
    wav_file.seek(0)
    from diffsvc_amd.vocoder import read_wav
    units = hubert.units(torch.from_numpy(read_wav(wav_file, 16000, mono="mean")).cuda())[0]
    f0_hz = np.asarray(f0_hz_of(len(mel)), np.float32)
    voiced = f0_hz > 0
    f0 = np.where(voiced, np.log2(np.maximum(f0_hz, 1e-3)), 0.0).astype(np.float32)
    if voiced.any() and not voiced.all():
        f0[~voiced] = np.interp(np.where(~voiced)[0], np.where(voiced)[0], f0[voiced])
    m2p = torch.from_numpy(synth.align_units(len(mel), units.shape[0]))[None].cuda()
    hp["pndm_speedup"] = speedup
    out = model(units[None], mel2ph=m2p, f0=torch.from_numpy(f0)[None].cuda(), ref_mels=torch.from_numpy(mel)[None].cuda(),
                infer=True, seed=seed, **model_kw)
    mel_out = out["mel_out"][0].clamp(hp["mel_vmin"], hp["mel_vmax"])
    f0_voc = pe(out["mel_out"])["f0_denorm_pred"][0] if pe is not None else out["f0_denorm"][0]
    return mel, mel_out.cpu().numpy(), f0_voc.cpu().numpy(), vocoder.spec2wav(mel_out.cpu().numpy(), f0=f0_voc.cpu().numpy(), seed=seed)


@pytest.fixture(scope="module")
def tiny_44k(tmp_path_factory):
    from diffsvc_amd.denoiser import DiffNetHip
    from diffsvc_amd.hparams import set_hparams
    from diffsvc_amd.hubert import HubertSoftHip
    from diffsvc_amd.sampler import GaussianDiffusionHip
    from diffsvc_amd.vocoder import NsfHifiGANHip
    d = tmp_path_factory.mktemp("svc")
    h = synth.tiny_vocoder(num_mels=16)                       # hop 16, 44.1 kHz
    hp = set_hparams(dict(synth.tiny_hparams(M=16, H=256, C=64, L=4, K=40), hop_size=16, fft_size=64, win_size=64, fmin=40, fmax=16000,
                          vocoder_ckpt=str(d / "voc" / "model")))
    synth.save_vocoder_ckpt(str(d / "voc"), dict(h, n_fft=64, win_size=64, hop_size=16), 5)
    model = GaussianDiffusionHip(None, 16, DiffNetHip(16, hparams=hp, precision="f16_x3"), timesteps=hp["timesteps"], K_step=hp["K_step"],
                                 loss_type="l2", spec_min=hp["spec_min"], spec_max=hp["spec_max"], hparams=hp)
    model.load_state_dict(synth.acoustic_state(hp, 3), strict=True)
    return hp, model.cuda(), NsfHifiGANHip(), HubertSoftHip(synth.hubert_state(11))


def f0_track(n):
    return np.where(np.arange(n) % 50 < 44, 200.0 * 2.0 ** (0.2 * np.sin(np.arange(n) / 30.0)), 0.0)


def test_wav_to_wav_chain_44k(tiny_44k):
    """Lengths follow the mel front-end, the result is reproducible for a fixed seed, DDPM and PLMS both run and differ."""
    hp = tiny_44k[0]
    sr = hp["audio_sample_rate"]
    wav = synth.speech_like_wav(1, int(0.9 * sr), sr)
    mel_in, mel_a, f0_a, y_a = convert(tiny_44k, wav_bytes(wav, sr), f0_track, 1, 5)
    n = len(mel_in)
    assert abs(n - len(wav) / 16) <= 2 and mel_a.shape == (n, 16) and len(y_a) == n * 16 and len(f0_a) == n
    assert np.isfinite(y_a).all() and np.abs(y_a).max() <= 1.0 and np.std(y_a) > 1e-3
    assert np.allclose(f0_a[f0_track(n) > 0], f0_track(n)[f0_track(n) > 0], rtol=1e-5)
    _, mel_b, _, y_b = convert(tiny_44k, wav_bytes(wav, sr), f0_track, 10, 5)
    _, mel_c, _, y_c = convert(tiny_44k, wav_bytes(wav, sr), f0_track, 10, 5)
    assert np.array_equal(y_b, y_c) and np.array_equal(mel_b, mel_c) and not np.allclose(mel_a, mel_b, atol=1e-3)


def test_use_gt_mel_starts_from_the_noised_input_mel(tiny_44k):
    """use_gt_mel / add_noise_step (diffusion.py:255-261): a shallow run (few noise steps from the input's own mel) stays closer to the
    input mel than a full run from noise."""
    hp = tiny_44k[0]
    sr = hp["audio_sample_rate"]
    wav = synth.speech_like_wav(4, int(0.7 * sr), sr)
    mel_in, shallow, _, _ = convert(tiny_44k, wav_bytes(wav, sr), f0_track, 1, 9, use_gt_mel=True, add_noise_step=3)
    _, full, _, _ = convert(tiny_44k, wav_bytes(wav, sr), f0_track, 1, 9)
    ref = np.clip(mel_in, hp["mel_vmin"], hp["mel_vmax"])
    assert np.abs(shallow - ref).mean() < 0.5 * np.abs(full - ref).mean()


def test_config_b_chain_24k_with_pitch_extractor(tmp_path):
    """BASELINE configs[0] shapes on the device: 22.05 kHz input -> [PWG-style wav2spec (24 kHz, 80 bins), HuBERT-soft, get_align] ->
    20-iteration PLMS (pndm_speedup 50 over the 1000-step schedule) -> PitchExtractor(mel_out) drives the 24 kHz HiFi-GAN (use_pe)."""
    from diffsvc_amd.denoiser import DiffNetHip
    from diffsvc_amd.hparams import set_hparams
    from diffsvc_amd.hubert import HubertSoftHip
    from diffsvc_amd.pe import PitchExtractorHip
    from diffsvc_amd.sampler import GaussianDiffusionHip
    from diffsvc_amd.vocoder import HifiGANHip
    hp = set_hparams(dict(synth.HPARAMS_24K, residual_layers=4, wav2spec_eps=1e-6, loud_norm=False, use_nsf=True,
                          vocoder_ckpt=str(tmp_path / "hifigan")))
    synth.save_hifigan_ckpt(str(tmp_path / "hifigan"), dict(synth.VOCODER_24K), 4)
    model = GaussianDiffusionHip(None, 80, DiffNetHip(80, hparams=hp), timesteps=1000, K_step=1000, loss_type="l2",
                                 spec_min=hp["spec_min"], spec_max=hp["spec_max"], hparams=hp)
    model.load_state_dict(synth.acoustic_state_conditioned(hp, 2, 1.35, 0.05), strict=True)
    pe = PitchExtractorHip(hparams=hp).cuda()
    pe.load_state_dict(synth.pe_state(hp, 5), strict=True)
    parts = (hp, model.cuda(), HifiGANHip(), HubertSoftHip(synth.hubert_state(11)))
    sr = 22050
    wav = synth.speech_like_wav(5, int(2.5 * sr), sr)
    mel_in, mel_out, f0_voc, y = convert(parts, wav_bytes(wav, sr), lambda n: np.full(n, 180.0), 50, 3, pe=pe.eval())
    assert abs(len(y) - len(wav) / sr * 24000) <= 128 and np.isfinite(y).all() and np.abs(y).max() <= 1.0 and np.std(y) > 1e-4
    assert mel_out.shape == mel_in.shape and (f0_voc > 0).any() and not np.allclose(f0_voc[:50], 180.0)     # f0 came from the extractor
