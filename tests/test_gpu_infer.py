"""GPU: the inference driver (diffsvc_amd.infer: SvcHip / run_clip, mirroring infer_tools/infer_tool.py:104-278 and infer.py:17-71) over
the native components end to end -- wav in, wav out -- on synthetic checkpoints (BASELINE configs[0] is this plumbing on the reference's
demo wav).  Parity of every component is pinned elsewhere; here the glue is checked: chunking, key shift, masks / clip, durations."""
import io
import wave

import numpy as np
import pytest
import torch

from diffsvc_amd import synth
from util import ROOT  # noqa: F401

pytestmark = pytest.mark.gpu


class _Units:
    """Hubertencoder.encode on the HIP encoder without a checkpoint file on disk (preprocessing/hubertinfer.py:30-42)."""

    def __init__(self, hb):
        self.hb = hb

    def encode(self, wav_fn):
        from diffsvc_amd.vocoder import read_wav
        if isinstance(wav_fn, io.BytesIO):
            wav_fn.seek(0)
        return self.hb.units(torch.from_numpy(read_wav(wav_fn, 16000)).cuda())[0].cpu().numpy()


@pytest.fixture(scope="module")
def svc(tmp_path_factory):
    from diffsvc_amd.denoiser import DiffNetHip
    from diffsvc_amd.hparams import set_hparams
    from diffsvc_amd.hubert import HubertSoftHip
    from diffsvc_amd.infer import SvcHip, load_ckpt
    from diffsvc_amd.sampler import GaussianDiffusionHip
    from diffsvc_amd.vocoder import NsfHifiGANHip
    d = tmp_path_factory.mktemp("svc")
    h = synth.tiny_vocoder(num_mels=16)                       # hop 16, 44.1 kHz
    hp = dict(synth.tiny_hparams(M=16, H=256, C=64, L=4, K=40), hop_size=16, fft_size=64, win_size=64, fmin=40, fmax=16000,
              vocoder_ckpt=str(d / "voc" / "model"), max_frames=42000, max_input_tokens=60000)
    hp = set_hparams(hp)
    synth.save_vocoder_ckpt(str(d / "voc"), dict(h, n_fft=64, win_size=64, hop_size=16), 5)
    sd = synth.acoustic_state(hp, 3)
    (d / "ckpt").mkdir()
    torch.save({"state_dict": {"model." + k: v for k, v in sd.items()}}, str(d / "ckpt" / "model_ckpt_steps_100.ckpt"))
    torch.save({"state_dict": {"model." + k: v * 0 for k, v in sd.items()}}, str(d / "ckpt" / "model_ckpt_steps_20.ckpt"))   # an older one: must lose
    model = GaussianDiffusionHip(None, 16, DiffNetHip(16, hparams=hp, precision="f16_x3"), timesteps=hp["timesteps"], K_step=hp["K_step"],
                                 loss_type="l2", spec_min=hp["spec_min"], spec_max=hp["spec_max"], hparams=hp)
    assert load_ckpt(model, str(d / "ckpt"), "model", strict=True).endswith("steps_100.ckpt")
    model.cuda()
    f0_fn = lambda wav, mel, hp_: np.where(np.arange(len(mel)) % 50 < 44, 200.0 * 2.0 ** (0.2 * np.sin(np.arange(len(mel)) / 30.0)), 0.0)
    s = SvcHip("test", model, NsfHifiGANHip(), _Units(HubertSoftHip(synth.hubert_state(11))), f0_fn=f0_fn, hparams=hp)
    return s, hp, sd


def _speech(seed, seconds, sr):
    return synth.speech_like_wav(seed, int(seconds * sr), sr)


def test_infer_one_segment(svc):
    """Svc.infer on an in-memory wav: lengths follow the mel front-end, f0_gt carries the key shift (one octave = x2), the result is
    reproducible for a fixed sampler seed, and DDPM (acc 1) and PLMS (acc 10) both run."""
    from diffsvc_amd.infer import _wav_bytes
    s, hp, _ = svc
    sr = hp["audio_sample_rate"]
    wav = _speech(1, 0.9, sr)
    f0_a, f0p_a, y_a = s.infer(_wav_bytes(wav, sr), key=0, acc=1, use_pe=False, seed=5)
    f0_b, f0p_b, y_b = s.infer(_wav_bytes(wav, sr), key=12, acc=1, use_pe=False, seed=5)
    n_frames = len(f0_a)
    assert abs(n_frames - len(wav) / 16) <= 2 and len(y_a) == n_frames * 16 and len(f0p_a) == n_frames
    voiced = f0_a > 0
    assert voiced.any() and np.allclose(f0_b[voiced], 2 * f0_a[voiced], rtol=1e-5)
    assert np.allclose(f0p_a, f0_a, rtol=1e-5)                                # use_pe=False: the vocoder is driven by the input f0
    assert np.isfinite(y_a).all() and np.abs(y_a).max() <= 1.0 and np.std(y_a) > 1e-3
    s.vocoder.seed = 0
    _, _, y1 = s.infer(_wav_bytes(wav, sr), key=0, acc=10, use_pe=False, seed=5)
    s.vocoder.seed = 0
    _, _, y2 = s.infer(_wav_bytes(wav, sr), key=0, acc=10, use_pe=False, seed=5)
    assert np.array_equal(y1, y2) and not np.allclose(y1[:len(y_a)], y_a[:len(y1)], atol=1e-3)
    with pytest.raises(RuntimeError, match="pitch-extractor"):
        s.infer(_wav_bytes(wav, sr), key=0, acc=10, use_pe=True, seed=5)


def test_run_clip_stitches_chunks_back_to_the_input_duration(svc, tmp_path):
    """run_clip: a 22.05 kHz input with a long silence in the middle is sliced, the silent chunk comes back as zeros, every chunk is
    forced to its own input duration at the model's rate, and the PCM-16 file has exactly ceil-summed length."""
    from diffsvc_amd.infer import run_clip
    from diffsvc_amd.slicer import chunks_of, cut_samples
    s, hp, _ = svc
    sr = 22050
    a, b = _speech(2, 3.0, sr), _speech(3, 3.0, sr)
    audio = np.concatenate([a, np.zeros(int(1.5 * sr), np.float32), b]).astype(np.float32)
    chunks = chunks_of(cut_samples(audio, sr, db_thresh=-40), audio)
    assert len(chunks) >= 3 and any(tag for tag, _ in chunks)
    out_path = str(tmp_path / "out.wav")
    f0_t, f0_p, out = run_clip(s, key=0, acc=10, use_pe=True, use_crepe=False, thre=0.05, use_gt_mel=False, add_noise_step=500,
                               audio=audio, sr=sr, out_path=out_path, slice_db=-40, seed=7)
    out = np.asarray(out)
    want = [int(np.ceil(len(d) / sr * hp["audio_sample_rate"])) for _, d in chunks]
    assert len(out) == sum(want)
    pos = 0
    for (tag, _), n in zip(chunks, want):
        seg = out[pos:pos + n]
        assert (seg == 0).all() if tag else np.std(seg) > 1e-4
        pos += n
    with wave.open(out_path, "rb") as w:
        assert (w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()) == (hp["audio_sample_rate"], 1, 2, len(out))
    assert len(f0_t) == len(f0_p)


def test_use_gt_mel_starts_from_the_noised_input_mel(svc):
    """use_gt_mel / add_noise_step (diffusion.py:255-261): a shallow run (few noise steps from the input's own mel) stays closer to the
    input mel than a full run from noise -- the q_sample path is wired through the driver."""
    from diffsvc_amd.infer import _wav_bytes
    s, hp, _ = svc
    sr = hp["audio_sample_rate"]
    wav = _speech(4, 0.7, sr)
    _, mel_in = s.vocoder.wav2spec(_wav_bytes(wav, sr))
    got = {}
    orig = s.vocoder.spec2wav
    try:
        def capture(mel, **kw):
            got["mel"] = np.asarray(mel)
            return np.zeros(len(mel) * 16)
        s.vocoder.spec2wav = capture
        s.infer(_wav_bytes(wav, sr), key=0, acc=1, use_pe=False, use_gt_mel=True, add_noise_step=3, seed=9)
        shallow = got["mel"]
        s.infer(_wav_bytes(wav, sr), key=0, acc=1, use_pe=False, seed=9)
        full = got["mel"]
    finally:
        s.vocoder.spec2wav = orig
    ref = np.clip(mel_in[:len(shallow)], hp["mel_vmin"], hp["mel_vmax"])
    assert np.abs(shallow - ref).mean() < 0.5 * np.abs(full[:len(ref)] - ref).mean()


def test_config_b_demo_chain_24k_with_pitch_extractor(tmp_path):
    """BASELINE configs[0] shapes end to end without the reference tree: 22.05 kHz input -> slicer -> [PWG wav2spec (24 kHz, 80 bins),
    HuBERT-soft, f0 read off the input mel by the pitch extractor (no crepe / parselmouth here), get_align] -> 20-iteration PLMS
    (acc 50 over a 1000-step schedule) -> PitchExtractor(mel_out) drives the 24 kHz HiFi-GAN (use_pe) -> stitched PCM."""
    from diffsvc_amd.denoiser import DiffNetHip
    from diffsvc_amd.hparams import set_hparams
    from diffsvc_amd.hubert import HubertSoftHip
    from diffsvc_amd.infer import SvcHip, run_clip
    from diffsvc_amd.pe import PitchExtractorHip
    from diffsvc_amd.sampler import GaussianDiffusionHip
    from diffsvc_amd.vocoder import HifiGANHip
    hp = set_hparams(dict(synth.HPARAMS_24K, residual_layers=4, wav2spec_eps=1e-6, loud_norm=False, use_nsf=True,
                          vocoder_ckpt=str(tmp_path / "hifigan"), max_frames=42000, max_input_tokens=60000))
    synth.save_hifigan_ckpt(str(tmp_path / "hifigan"), dict(synth.VOCODER_24K), 4)
    sd = synth.acoustic_state_conditioned(hp, 2, 1.35, 0.05)
    model = GaussianDiffusionHip(None, 80, DiffNetHip(80, hparams=hp), timesteps=1000, K_step=1000, loss_type="l2",
                                 spec_min=hp["spec_min"], spec_max=hp["spec_max"], hparams=hp)
    model.load_state_dict(sd, strict=True)
    model.cuda()
    pe = PitchExtractorHip(hparams=hp).cuda()
    pe.load_state_dict(synth.pe_state(hp, 5), strict=True)
    s = SvcHip("demo", model, HifiGANHip(), _Units(HubertSoftHip(synth.hubert_state(11))), pe=pe.eval(), hparams=hp)
    sr = 22050
    audio = np.concatenate([_speech(5, 2.5, sr), np.zeros(int(1.2 * sr), np.float32), _speech(6, 3.5, sr)]).astype(np.float32)
    f0_t, f0_p, out = run_clip(s, key=2, acc=50, use_pe=True, use_crepe=False, thre=0.05, use_gt_mel=False, add_noise_step=500,
                               audio=audio, sr=sr, out_path=str(tmp_path / "o.wav"), seed=3)
    out = np.asarray(out)
    assert abs(len(out) - len(audio) / sr * 24000) <= 4 and np.isfinite(out).all() and np.abs(out).max() <= 1.0
    assert len(f0_t) == len(f0_p) and (np.asarray(f0_p) > 0).any()
    assert not np.allclose(np.asarray(f0_p)[:50], np.asarray(f0_t)[:50])        # the vocoder's f0 came from the extractor, not the input track
