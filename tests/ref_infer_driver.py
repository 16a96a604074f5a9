"""Run INSIDE a subprocess by tests/test_reference_seams.py: BASELINE configs[0] ("infer.py plumbing, no GPU") -- the REAL reference
driver (infer_tools/infer_tool.py:104-345: Svc.infer / pre / temporary_dict2processed_input / getitem / processed_input2batch /
after_infer, unmodified) over the drop-in classes, wav in -> wav out.

There is no GPU in the build container and the product has no CPU path, so the four C-ABI handle wrappers (sampler, denoiser,
vocoder, mel front-end) are replaced here by stand-ins that compute with the oracle.  What this exercises is therefore every line of
HOST code between the reference's driver and the drop-ins: class contracts, keyword sets, returned dict keys, tensor layouts,
hparams sharing, checkpoint loading through the reference's loader, the vocoder registry lookup by bare class name.  The expected
output is computed independently from the same inputs with plain oracle calls.  TEST INFRASTRUCTURE (container only)."""
import io
import json
import os
import sys
import tempfile
import wave

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
work = tempfile.mkdtemp(prefix="dsvc_ref_infer_")
import atexit, shutil  # noqa: E402
atexit.register(shutil.rmtree, work, ignore_errors=True)
os.makedirs(os.path.join(work, "infer_tools"))            # infer_tool.py:52 opens ./infer_tools/f0_temp.json relative to the CWD
os.chdir(work)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import refshim  # noqa: E402

refshim.install()
import infer_tools.infer_tool as IT  # noqa: E402  (the reference's driver module)
sys.modules["soundfile"] = None                            # the stub module has no read(): make read_wav take its stdlib branch

import diffsvc_amd  # noqa: E402,F401
from diffsvc_amd import synth  # noqa: E402
import dsvc_oracle as O  # noqa: E402

# ---- no device here: .cuda() / .to('cuda') are identity, the availability check passes ----
torch.cuda.is_available = lambda: True
torch.Tensor.cuda = lambda self, *a, **k: self
_to = torch.Tensor.to


def _to_host(self, *a, **k):
    a = tuple(x for x in a if not ((isinstance(x, str) and x.startswith("cuda")) or (isinstance(x, torch.device) and x.type == "cuda")))
    if isinstance(k.get("device"), str) and k["device"].startswith("cuda"):
        k.pop("device")
    return _to(self, *a, **k) if (a or k) else self


torch.Tensor.to = _to_host


# ---- oracle-backed stand-ins for the C-ABI handle wrappers (diffsvc_amd/engine.py) ----
class FakeDenoiser:
    def __init__(self, state, mel_bins, hidden, channels, layers, dilation_cycle, max_steps, precision="auto", prefix=""):
        self.state = {"denoise_fn." + k[len(prefix):]: v.detach().float().cpu() for k, v in state.items() if k.startswith(prefix)}
        self.mel_bins, self.hidden, self.cyc = mel_bins, hidden, dilation_cycle


class FakeSampler:
    calls = []

    def __init__(self, den, state):
        self.den = den
        self.sd = dict(den.state)
        self.sd.update({k: v.detach().float().cpu() for k, v in state.items()})

    def sample(self, cond, t_start, speedup=1, x_init=None, mel2ph=None, seed=0, first_clip=0, t_stop=0, use_graph=True,
               return_x=False, ref_mel=None, clip_ids=None, clip_lens=None, clip_lens_host=None):
        B, H, T = cond.shape
        M = self.den.mel_bins
        clips = [int(v) for v in clip_ids.tolist()] if clip_ids is not None else [first_clip + b for b in range(B)]
        FakeSampler.calls.append(dict(t_start=t_start, speedup=speedup, seed=seed, T=T))
        assert ref_mel is None and x_init is None
        if clip_lens is not None or B > 1:
            # dsvc_sample_args.clip_lens: frames beyond a clip's length are the convs' zero padding, exactly as if the clip had run alone
            lens = [int(v) for v in clip_lens.tolist()] if clip_lens is not None else [T] * B
            assert clip_lens_host is None or list(clip_lens_host) == lens
            out = torch.zeros(B, T, M)
            for b in range(B):
                n = lens[b]
                out[b:b + 1, :n] = self.sample(cond[b:b + 1, :, :n].contiguous(), t_start, speedup, mel2ph=None if mel2ph is None else mel2ph[b:b + 1, :n],
                                               seed=seed, first_clip=clips[b])
                FakeSampler.calls.pop()
            return out
        x = O.ddpm_noise_ref_layout(seed, clips, 0, T, M, O.PURPOSE_X_INIT)
        if speedup > 1:
            x = O.sample_plms(self.sd, cond, x, speedup, self.den.cyc, t_start=t_start)
        else:
            x = O.sample_ddpm(self.sd, cond, x, lambda i: O.ddpm_noise_ref_layout(seed, clips, i, T, M), self.den.cyc, t_start=t_start)
        return O.finish_mel(self.sd, x, mel2ph)


class FakeVocoder:
    def __init__(self, state, h, precision="f16_x3", mel_scale=2.30259, use_source=True):
        self.gw, self.h, self.mel_scale = O.fold_weight_norm(state), h, mel_scale
        self.hop = int(np.prod(h["upsample_rates"]))

    def vocode(self, mel, f0, seed=0, first_clip=0, clip_ids=None):
        B, T, _ = mel.shape
        ini, nz = O.vocoder_rng(seed, [first_clip + b for b in range(B)], T * self.hop)
        with torch.no_grad():
            return O.generator_forward(self.gw, self.h, self.mel_scale * mel.transpose(2, 1), f0, ini, nz).reshape(B, -1)


class FakeMelspec:
    def __init__(self, sr, n_fft, win_size, hop, n_mels, fmin, fmax, clip_val=1e-5, mode=0):
        self.a = (sr, n_fft, win_size, hop, n_mels, fmin, fmax)

    def mel(self, wav):
        return O.mel_spectrogram(wav, *self.a)


import diffsvc_amd.denoiser as DN  # noqa: E402
import diffsvc_amd.sampler as SM  # noqa: E402
import diffsvc_amd.vocoder as VC  # noqa: E402
DN.DenoiserHandle, SM.SamplerHandle, VC.VocoderHandle, VC.MelspecHandle = FakeDenoiser, FakeSampler, FakeVocoder, FakeMelspec

# ---- hparams: the tiny architecture at 44.1 kHz with the keys the reference driver reads ----
voc_cfg = dict(synth.tiny_vocoder(num_mels=16), n_fft=64, win_size=64, hop_size=16)
hp = dict(synth.tiny_hparams(M=16, H=32, C=64, L=4, K=40), hop_size=16, fft_size=64, win_size=64, fmin=40, fmax=16000,
          vocoder_ckpt=os.path.join(work, "voc", "model"), max_frames=42000, max_input_tokens=60000, debug=False,
          binarization_args=dict(with_f0=True, with_hubert=True, with_align=True), hubert_gpu=False)
refshim.set_hparams(hp)
from utils.hparams import hparams as ref_hp  # noqa: E402
import utils  # noqa: E402  (the reference's package)
vstate = synth.save_vocoder_ckpt(os.path.join(work, "voc"), voc_cfg, 5)
ckpt = os.path.join(work, "model_ckpt_steps_100.ckpt")
sd = synth.save_acoustic_ckpt(ckpt, hp, seed=3)

# ---- the wiring of INTEGRATION.md section 1, applied to a Svc built without its GPU-only constructor ----
svc = IT.Svc.__new__(IT.Svc)
svc.project_name = "demo"
svc.DIFF_DECODERS = {"wavenet": lambda h_: DN.DiffNetHip(h_["audio_num_mel_bins"])}
svc.mel_bins = ref_hp["audio_num_mel_bins"]
svc.model = SM.GaussianDiffusionHip(phone_encoder=None, out_dims=svc.mel_bins, denoise_fn=svc.DIFF_DECODERS[ref_hp["diff_decoder_type"]](ref_hp),
                                    timesteps=ref_hp["timesteps"], K_step=ref_hp["K_step"], loss_type=ref_hp["diff_loss_type"],
                                    spec_min=ref_hp["spec_min"], spec_max=ref_hp["spec_max"])
svc.model_path = ckpt
svc.load_ckpt()                                              # utils.load_ckpt(strict=True) of the reference
svc.vocoder = IT.get_vocoder_cls(ref_hp)()                   # the dotted class path of the YAML `vocoder:` key

sr = ref_hp["audio_sample_rate"]
wav = synth.speech_like_wav(1, int(0.5 * sr), sr)
n_units = 17
g = np.random.Generator(np.random.PCG64(9))
units = (g.standard_normal((n_units, ref_hp["hidden_size"])) * 0.5).astype(np.float32)


class Units:
    def encode(self, wav_fn):
        return units


svc.hubert = Units()
f0_track = lambda n: np.where(np.arange(n) % 30 < 25, 190.0 * 2.0 ** (0.2 * np.sin(np.arange(n) / 11.0)), 0.0).astype(np.float32)
IT.get_pitch_crepe = lambda wav_, mel_, hp_, thre: (f0_track(len(mel_)), np.ones(len(mel_), np.int64))   # torchcrepe is third-party, absent


def wav_file():
    buf = io.BytesIO()
    with wave.open(buf, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr)
        w.writeframes(np.clip(np.rint(wav * 32767.0), -32768, 32767).astype("<i2").tobytes())
    buf.seek(0)
    return buf


out = {}
key, acc, seed = 2, 10, 5
svc.vocoder.seed = 0
f0_gt, f0_pred, wav_pred = svc.infer(wav_file(), key=key, acc=acc, use_pe=False, use_crepe=True, seed=seed)
out["sampler_calls"] = FakeSampler.calls
out["pndm_speedup_set_by_pre"] = ref_hp["pndm_speedup"] == acc

# ---- the same conversion with plain oracle calls ----
pcm = np.clip(np.rint(wav * 32767.0), -32768, 32767).astype("<i2").astype(np.float32) / 32768.0
mel_in = O.mel_spectrogram(torch.from_numpy(pcm)[None], sr, 64, 64, 16, 16, 40, 16000)[0].numpy()
T = mel_in.shape[0]
f0_hz = f0_track(T)
from utils.pitch_utils import norm_interp_f0  # noqa: E402  (the reference's own)
f0, uv = norm_interp_f0(f0_hz, ref_hp)
f0 = f0 + key / 12
f0[f0 > np.log2(ref_hp["f0_max"])] = 0
m2p = torch.from_numpy(O.get_align(T, n_units))[None]
full = {k: v for k, v in sd.items()}
cond, f0_denorm, _ = O.build_cond(full, torch.from_numpy(units)[None], m2p, f0[None].clone(), ref_hp)
x = O.ddpm_noise_ref_layout(seed, [0], 0, T, 16, O.PURPOSE_X_INIT)
x = O.sample_plms(full, cond.transpose(1, 2).contiguous(), x, acc, ref_hp["dilation_cycle_length"], t_start=ref_hp["K_step"])
mel_ref = O.finish_mel(full, x, m2p)[0].numpy()
mel_c, f0_c = O.after_infer_mel(mel_ref, f0_denorm[0].numpy(), ref_hp)
ini, nz = O.vocoder_rng(1, [0], len(mel_c) * 16)            # NsfHifiGANHip counts its calls: the first spec2wav uses seed 1
wav_ref = O.spec2wav(O.fold_weight_norm(vstate), voc_cfg, mel_c, f0_c, ini, nz).numpy()
out["frames"] = int(T)
out["wav_len_ok"] = len(wav_pred) == len(mel_c) * 16 == len(wav_ref)
out["wav_max_abs_diff"] = float(np.abs(np.asarray(wav_pred) - wav_ref).max())
out["f0_pred_max_abs_diff"] = float(np.abs(np.asarray(f0_pred) - f0_c).max())
out["f0_gt_is_shifted_input"] = bool(np.allclose(np.asarray(f0_gt)[f0_hz > 0], f0_hz[f0_hz > 0] * 2.0 ** (key / 12), rtol=1e-5))
out["wav_rms"] = float(np.sqrt((wav_ref ** 2).mean()))

# ---- the chunks of one utterance through diffsvc_amd.svc_chunks.infer_chunks (the reference's Svc object, its collate functions, ONE model call
#      per planned group) against the reference's loop of Svc.infer calls: the glue must hand every chunk what the loop hands it ----
from diffsvc_amd.svc_chunks import infer_chunks as svc_infer_chunks  # noqa: E402


def wav_file_of(seconds, s_):
    w_ = synth.speech_like_wav(s_, int(seconds * sr), sr)
    buf = io.BytesIO()
    with wave.open(buf, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr)
        w.writeframes(np.clip(np.rint(w_ * 32767.0), -32768, 32767).astype("<i2").tobytes())
    buf.seek(0)
    return buf


secs = [0.31, 0.5, 0.22, 0.5]
# (the reference's driver caches every f0 track it extracts in a JSON-able dict, infer_tool.py:205-217: a second pass over the same audio would read
#  float64 lists back where the first pass had float32 arrays -- cleared between the passes so that all three see the same inputs)
IT.f0_dict.clear()
svc.vocoder.seed = 0
loop = [svc.infer(wav_file_of(t, 7 + i), key=key, acc=acc, use_pe=False, use_crepe=True, seed=seed, first_clip=i) for i, t in enumerate(secs)]
n_calls = len(FakeSampler.calls)
IT.f0_dict.clear()
svc.vocoder.seed = 0
got = svc_infer_chunks(svc, [wav_file_of(t, 7 + i) for i, t in enumerate(secs)], key=key, acc=acc, use_pe=False, use_crepe=True, seed=seed)
out["chunks_model_calls"] = len(FakeSampler.calls) - n_calls
# (f0_pred and the waveform: exact.  f0_gt is 2 ** f0 evaluated by torch on the host over the padded batch: its vectorised loop and its scalar tail
#  differ in the last bit, so a value's bits depend on where it sits in the tensor -- one ulp, 1.5e-5 Hz)
out["chunks_equal_loop"] = bool(all(len(a) == 3 and np.array_equal(np.asarray(a[1]), np.asarray(b[1])) and np.array_equal(np.asarray(a[2]), np.asarray(b[2]))
                                    and np.asarray(a[0]).shape == np.asarray(b[0]).shape and np.allclose(np.asarray(a[0]), np.asarray(b[0]), rtol=3e-7, atol=0)
                                    for a, b in zip(got, loop)))
out["chunks_lens"] = [int(len(a[2])) for a in got]
out["chunks_diffs"] = [[float(np.abs(np.asarray(x, dtype=np.float64) - np.asarray(y, dtype=np.float64)).max()) if np.asarray(x).shape == np.asarray(y).shape else [list(np.asarray(x).shape), list(np.asarray(y).shape)] for x, y in zip(a, b)] for a, b in zip(got, loop)]
IT.f0_dict.clear()
svc.vocoder.seed = 0
one = svc_infer_chunks(svc, [wav_file_of(t, 7 + i) for i, t in enumerate(secs)], key=key, acc=acc, use_pe=False, use_crepe=True, seed=seed, batch=False)
out["chunks_unbatched_equal_loop"] = bool(all(all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(a, b)) for a, b in zip(one, loop)))
print("RESULT " + json.dumps(out))
