"""GPU parity: NSF-HiFiGAN generator and the STFT/mel front-end (HIP, through the C ABI) against the golden
vectors minted from the real reference and against the oracle."""
import numpy as np
import pytest
import torch

from diffsvc_amd import synth
import dsvc_oracle as O
from util import load_golden

pytestmark = pytest.mark.gpu

WAV_RMS_TOL = 1e-4          # north_star: waveform within 1e-4 RMS
MEL_TOL = 1e-3              # north_star: mel within 1e-3 max-abs


def vocoder_for(h, wseed, precision="f16_x3"):
    from diffsvc_amd.engine import VocoderHandle
    return VocoderHandle(synth.vocoder_state(h, wseed), h, precision=precision)


@pytest.mark.parametrize("name", ["vocoder_tiny", "vocoder_44k", "vocoder_tiny_rb2", "vocoder_44k_rb2", "vocoder_tiny_rb2_d3"])
def test_vocoder_vs_reference_golden(name):
    g = load_golden(name)
    h = synth.tiny_vocoder() if "tiny" in name else dict(synth.VOCODER_44K)
    if name.endswith("_rb2"):            # generators built from ResBlock2 (models.py:73-91, :337): one conv per residual step, two dilations
        h = dict(h, resblock="2", resblock_dilation_sizes=[[1, 3]] * len(h["resblock_kernel_sizes"]))
    if name.endswith("_rb2_d3"):         # ... under a config whose lists hold THREE dilations: the reference builds two convs from the first two (models.py:77-82)
        h = dict(h, resblock="2", resblock_dilation_sizes=[[1, 3, 5], [2, 4, 7]])
    voc = vocoder_for(h, int(g["wseed"]))
    clips = [int(c) for c in g["clips"]]
    wavs = []
    for i, c in enumerate(clips):
        w = voc.vocode(torch.from_numpy(g["mel"][i:i + 1]).cuda(), torch.from_numpy(g["f0"][i:i + 1]).cuda(),
                       seed=int(g["seed"]), first_clip=c)
        wavs.append(w.cpu())
    wav = torch.cat(wavs)
    ref = torch.from_numpy(g["wav"])
    rms = (wav - ref).pow(2).mean().sqrt().item()
    assert rms < WAV_RMS_TOL, rms
    assert (wav - ref).abs().max().item() < 2e-3


def test_vocoder_batch_equals_per_clip_and_oracle_long_clip():
    """A longer clip than the golden (several voiced/unvoiced transitions, phase wraps of every harmonic) and
    a batch of two: batched == per clip bit for bit, and both within the RMS bar of the oracle."""
    h = dict(synth.VOCODER_44K)
    voc = vocoder_for(h, 1)
    T, hop = 40, 512
    g = np.random.Generator(np.random.PCG64(3))
    mel = torch.from_numpy((g.standard_normal((2, T, 128)) * 0.8 - 2.5).astype(np.float32))
    f0 = np.stack([synth.clip_inputs(c, T=T, n_units=23)[3] for c in (4, 9)])
    f0[0, 5:9] = 0.0
    f0[1, 30:] = 0.0
    f0 = torch.from_numpy(f0)
    full = voc.vocode(mel.cuda(), f0.cuda(), seed=77, first_clip=4)
    one = voc.vocode(mel[1:2].cuda(), f0[1:2].cuda(), seed=77, first_clip=5)
    assert torch.equal(full[1], one[0])
    ini, nz = O.vocoder_rng(77, [4, 5], T * hop)
    gw = O.fold_weight_norm(synth.vocoder_state(h, 1))
    with torch.no_grad():
        ref = O.generator_forward(gw, h, 2.30259 * mel.transpose(2, 1), f0, ini, nz).reshape(2, -1)
    rms = (full.cpu() - ref).pow(2).mean().sqrt().item()
    assert rms < WAV_RMS_TOL, rms


def test_vocoder_all_unvoiced_and_constant_pitch():
    """Edge cases of the harmonic source: f0 == 0 everywhere (noise excitation only) and a constant pitch whose
    phase accumulates over every frame (the closed-form cumsum must track torch's sequential one)."""
    h = synth.tiny_vocoder()
    voc = vocoder_for(h, 5)
    T = 64
    hop = int(np.prod(h["upsample_rates"]))
    g = np.random.Generator(np.random.PCG64(8))
    mel = torch.from_numpy((g.standard_normal((2, T, h["num_mels"])) * 0.8 - 2.5).astype(np.float32))
    f0 = torch.zeros(2, T)
    f0[1] = 437.3
    wav = voc.vocode(mel.cuda(), f0.cuda(), seed=3, first_clip=0).cpu()
    ini, nz = O.vocoder_rng(3, [0, 1], T * hop)
    gw = O.fold_weight_norm(synth.vocoder_state(h, 5))
    with torch.no_grad():
        ref = O.generator_forward(gw, h, 2.30259 * mel.transpose(2, 1), f0, ini, nz).reshape(2, -1)
    assert (wav - ref).pow(2).mean().sqrt().item() < WAV_RMS_TOL


def test_vocoder_rejects_bad_shapes():
    h = synth.tiny_vocoder()
    voc = vocoder_for(h, 5)
    with pytest.raises(ValueError):
        voc.vocode(torch.zeros(1, 8, h["num_mels"] + 1).cuda(), torch.zeros(1, 8).cuda())
    with pytest.raises(RuntimeError):
        voc.vocode(torch.zeros(1, 8, h["num_mels"]), torch.zeros(1, 8))          # CPU tensors: no CPU path


@pytest.mark.parametrize("name", ["melspec_44k", "melspec_24k"])
def test_melspec_vs_reference_golden(name):
    from diffsvc_amd.engine import MelspecHandle
    g = load_golden(name)
    sr, n_fft, win, hop, n_mels, fmin, fmax = [int(v) for v in g["cfg"]]
    ms = MelspecHandle(sr, n_fft, win, hop, n_mels, fmin, fmax)
    wav = torch.from_numpy(g["wav"])[None]
    assert ms.frames(wav.shape[1]) == g["mel"].shape[0]
    mel = ms.mel(wav.cuda())[0].cpu().numpy()
    err = np.abs(mel - g["mel"]).max()
    assert err < MEL_TOL, err
    assert err < 2e-4, err          # fp32 FFT: far inside the bar


def test_melspec_full_clip_batch_and_silence():
    """BASELINE-size input (10 s @ 44.1 kHz -> 861 frames), a batch of 2, one clip digital silence: the frame
    count matches nvSTFT.py:92-96, silence lands exactly on log10(clip_val), rows are independent."""
    from diffsvc_amd.engine import MelspecHandle
    ms = MelspecHandle(44100, 2048, 2048, 512, 128, 40, 16000)
    N = 441000
    g = np.random.Generator(np.random.PCG64(1))
    wav = torch.zeros(2, N)
    wav[0] = torch.from_numpy((g.standard_normal(N) * 0.1).astype(np.float32))
    assert ms.frames(N) == 861
    mel = ms.mel(wav.cuda()).cpu()
    assert mel.shape == (2, 861, 128)
    sil = 0.434294 * np.log(1e-5)       # |X| = sqrt(1e-9) everywhere -> every filter sums below clip_val
    assert torch.allclose(mel[1], torch.full_like(mel[1], float(sil)), atol=1e-6)
    part = ms.mel(wav[:1, :20000].cuda()).cpu()            # a prefix shares its interior frames with the full clip
    assert torch.equal(part[0, :30], mel[0, :30])
    ref = O.mel_spectrogram(wav[:1, :20000], 44100, 2048, 2048, 512, 128, 40, 16000)[0]
    assert (part[0] - ref).abs().max().item() < 2e-4


@pytest.mark.parametrize("with_source", [True, False])
def test_hifigan_24k_generator_vs_reference_golden(with_source):
    """The 24 kHz demo-config vocoder (modules/hifigan/hifigan.py:104-178, HifiGanGenerator) on the same kernels: 80 natural-log mel
    bins fed unscaled (mel_scale 1), hop 128, the harmonic source only when an f0 is given -- against the REAL generator."""
    from diffsvc_amd.engine import VocoderHandle
    g = load_golden("hifigan_24k")
    h = dict(synth.VOCODER_24K)
    voc = VocoderHandle(synth.vocoder_state(h, int(g["wseed"])), h, precision="f16_x3", mel_scale=1.0, use_source=with_source)
    clips = [int(c) for c in g["clips"]]
    wavs = []
    for i, c in enumerate(clips):
        f0 = torch.from_numpy(g["f0"][i:i + 1]).cuda() if with_source else None
        wavs.append(voc.vocode(torch.from_numpy(g["mel"][i:i + 1]).cuda(), f0, seed=int(g["seed"]), first_clip=c).cpu())
    ref = torch.from_numpy(g["wav_src" if with_source else "wav_plain"])
    rms = (torch.cat(wavs) - ref).pow(2).mean().sqrt().item()
    print("hifigan 24k (source=%s): wav RMS err %.2e" % (with_source, rms))
    assert rms < WAV_RMS_TOL, rms


def test_hifigan_24k_plugin_contract(tmp_path):
    """HifiGANHip through the reference's 24 kHz vocoder contract (network/vocoders/hifigan.py:46-76): hparams['vocoder_ckpt'] is a
    directory with config.yaml + model_ckpt_steps_<N>.ckpt (['state_dict']['model_gen'], highest N wins); spec2wav(numpy mel[T,80]
    natural log, f0=...) -> numpy; the source is used only with an f0 AND hparams['use_nsf']."""
    from diffsvc_amd.hparams import set_hparams
    from diffsvc_amd.vocoder import HifiGANHip
    h = dict(synth.VOCODER_24K)
    d = str(tmp_path / "hifigan")
    synth.save_hifigan_ckpt(d, h, seed=3, steps=900)                 # an older step count: must lose against ...
    vs = synth.save_hifigan_ckpt(d, h, seed=7, steps=12000)           # ... this one
    set_hparams(dict(synth.HPARAMS_24K, vocoder_ckpt=d, use_nsf=True))
    voc = HifiGANHip()
    g = load_golden("hifigan_24k")
    mel, f0 = g["mel"][0], g["f0"][0]
    hop = int(np.prod(h["upsample_rates"]))
    ini, nz = O.vocoder_rng(5, [0], mel.shape[0] * hop)
    gw = O.fold_weight_norm(vs)
    wav = voc.spec2wav(mel, f0=f0, seed=5)
    assert isinstance(wav, np.ndarray) and wav.dtype == np.float32 and wav.shape == (mel.shape[0] * hop,)
    with torch.no_grad():
        c = torch.from_numpy(mel)[None].transpose(2, 1)
        ref_src = O.generator_forward(gw, h, c, torch.from_numpy(f0)[None], ini, nz).reshape(-1).numpy()
        ref_plain = O.generator_forward(gw, h, c, None, ini, nz).reshape(-1).numpy()
    assert np.sqrt(np.mean((wav - ref_src) ** 2)) < WAV_RMS_TOL
    assert np.sqrt(np.mean((voc.spec2wav(mel) - ref_plain) ** 2)) < WAV_RMS_TOL            # no f0: plain HiFi-GAN (hifigan.py:66-71)
    set_hparams(dict(synth.HPARAMS_24K, vocoder_ckpt=d, use_nsf=False))
    assert np.sqrt(np.mean((voc.spec2wav(mel, f0=f0) - ref_plain) ** 2)) < WAV_RMS_TOL     # f0 ignored without use_nsf
    with pytest.raises(FileNotFoundError):
        set_hparams(dict(synth.HPARAMS_24K, vocoder_ckpt=str(tmp_path / "nothing")))
        HifiGANHip()


def test_pwg_front_end_wav2spec_24k(tmp_path):
    """HifiGANHip.wav2spec = PWG.wav2spec -> process_utterance (preprocessing/data_gen_utils.py:124-145), the 24 kHz demo config's mel
    front-end (fft 512, hop 128, 80 bins, fmin 30, fmax 12000, eps 1e-6): centred zero-padded STFT, |X|, mel, log10(max(eps, .)) -- against
    the oracle restatement (unpinned at librosa, like the filterbank), incl. a 22.05 kHz file that is resampled first, lengths that are
    and are not multiples of the hop, and the waveform zero-padded to frames * hop."""
    import wave
    from diffsvc_amd.hparams import set_hparams
    from diffsvc_amd.vocoder import HifiGANHip, read_wav
    hp = set_hparams(dict(synth.HPARAMS_24K, wav2spec_eps=1e-6, loud_norm=False))
    for sr, n in ((24000, 24000), (24000, 12345), (22050, 30000)):
        pcm = (synth.speech_like_wav(n, n, sr) * 32767).astype("<i2")
        path = str(tmp_path / ("a%d_%d.wav" % (sr, n)))
        with wave.open(path, "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr)
            w.writeframes(pcm.tobytes())
        wav, mel = HifiGANHip.wav2spec(path)
        src = read_wav(path, 24000)
        T = 1 + len(src) // 128
        assert mel.shape == (T, 80) and wav.shape == (T * 128,)
        assert np.array_equal(wav[:len(src)], src[:len(wav)]) and (wav[len(src):] == 0).all()
        ref = O.process_utterance_mel(torch.from_numpy(src)[None], 24000, 512, 512, 128, 80, 30, 12000, eps=1e-6)[0].numpy()
        err = np.abs(mel - ref).max()
        print("pwg wav2spec %d Hz, %d samples -> %d frames: max-abs err %.2e (mel range %.2f..%.2f)" % (sr, n, T, err, ref.min(), ref.max()))
        assert err < 2e-4, err
        # return_linear=True (round 6): the normalised dB spectrogram beside the same wav / mel
        set_hparams(dict(hp, min_level_db=-120), clear=False)
        wav3, mel3, lin = HifiGANHip.wav2spec(path, return_linear=True)
        assert np.array_equal(wav3, wav) and np.array_equal(mel3, mel) and lin.shape == (T, 257)
        lref = O.process_utterance_linear(torch.from_numpy(src)[None], 512, 512, 128, -120.0)[0].numpy()
        lerr = np.abs(lin - lref).max()
        strong = lref >= 0.5                                   # bins above -60 dB: an fp32 FFT's rounding noise (~1e-6 absolute on this signal) is
        serr = np.abs(lin - lref)[strong].max()                # invisible there; at -90 dB it is 0.3 % of |X| = 2e-4 of the normalised dB scale -- in
        print("pwg wav2spec return_linear: max-abs err %.2e, %.2e on the bins above -60 dB (range %.3f..%.3f)" % (lerr, serr, lref.min(), lref.max()))
        assert serr < 5e-5 and lerr < 1e-3 and 0.0 <= lref.min() and lin.min() >= 0.0, (lerr, serr)      # torch.stft's own fp32 arithmetic just the same
    # loud_norm (round 6; data_gen_utils.py:117-122): the waveform is brought to -22 LUFS before the STFT -- the returned wav is the normalised one
    # (BS.1770 as pyloudnorm implements it, restated on the host: diffsvc_amd/loudness.py), the mel is the mel OF that waveform
    from diffsvc_amd.loudness import integrated_loudness
    set_hparams(dict(hp, loud_norm=True), clear=False)
    wav_n, mel_n = HifiGANHip.wav2spec(path)
    src = read_wav(path, 24000)
    assert abs(integrated_loudness(wav_n[:len(src)], 24000) - (-22.0)) < 0.02 and np.abs(wav_n).max() <= 1.0
    gain = np.abs(wav_n).max() / np.abs(src).max()
    ref_n = O.process_utterance_mel(torch.from_numpy((src * gain).astype(np.float32))[None], 24000, 512, 512, 128, 80, 30, 12000, eps=1e-6)[0].numpy()
    assert np.abs(mel_n - ref_n).max() < 2e-4
    set_hparams(dict(hp, loud_norm=False), clear=False)
