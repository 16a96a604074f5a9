"""Thin object wrappers over the C ABI handles (include/dsvc.h).  Device memory, streams and tensors come
from PyTorch-ROCm; every FLOP on the hot path happens inside libdsvc_hip.so."""
import ctypes

import torch

from . import _lib
from ._lib import PRECISIONS, check, host_f32, lib, parse_precision, ptr, stream_ptr

SCHEDULE_KEYS = ("alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                 "posterior_mean_coef1", "posterior_mean_coef2", "posterior_log_variance_clipped",
                 "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "spec_min", "spec_max")


def _need_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("diffsvc_amd: tensors must live on the HIP device (got %s); there is no CPU path" % t.device)


def _prec(p):
    if isinstance(p, str):
        if p not in PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(PRECISIONS))
        return PRECISIONS[p]
    return int(p)


class DenoiserHandle:
    """dsvc_denoiser: DiffNet weights packed for the MFMA kernels + per-step FiLM tables."""

    def __init__(self, state, mel_bins, hidden, channels, layers, dilation_cycle, max_steps,
                 precision="f16_d64", prefix=""):
        self._h = ctypes.c_void_p(0)
        self._L = lib()                                  # the library this handle lives in (product, or the test-hooks build: _lib.hooks_build)
        prec, variants = parse_precision(precision)
        self.cfg = _lib.DenoiserCfg(mel_bins, hidden, channels, layers, dilation_cycle, max_steps, prec, variants)
        self.mel_bins, self.hidden = mel_bins, hidden
        self._ck(self._L.dsvc_denoiser_create(ctypes.byref(self.cfg), ctypes.byref(self._h)))
        n = 0
        for k, v in state.items():
            if not k.startswith(prefix):
                continue
            h, p = host_f32(v)
            self._ck(self._L.dsvc_denoiser_load_tensor(self._h, k[len(prefix):].encode(), p, h.numel()))
            n += 1
        if n == 0:
            raise RuntimeError("no tensors with prefix '%s' in the state dict" % prefix)
        self._ck(self._L.dsvc_denoiser_finalize(self._h))

    def _ck(self, rc):
        check(rc, self._L)

    def forward(self, spec, t, cond, cond_changed=True):
        """DiffNet.forward: spec [B,1,M,T], t [B] (any int dtype), cond [B,H,T] -> [B,1,M,T]."""
        _need_cuda(spec, t, cond)
        B, _, M, T = spec.shape
        if M != self.mel_bins or cond.shape != (B, self.hidden, T):
            raise ValueError("shape mismatch: spec %s cond %s" % (tuple(spec.shape), tuple(cond.shape)))
        spec = spec.contiguous().float()
        cond = cond.contiguous().float()
        t32 = t.to(torch.int32).contiguous()
        if t32.numel() != B:
            raise ValueError("diffusion_step must hold one step per clip (%d), got %d" % (B, t32.numel()))
        # the step embedding / FiLM path is TABULATED for the integer steps 0 .. timesteps-1 (net.py:32-44,99-103 evaluated at load).
        # Steps outside the table are clamped ON THE DEVICE and raise a sticky flag (no device-to-host check on this 1000-calls-per-
        # clip seam): check() raises (later, valid calls are executed normally).  (The sampler never leaves the range; this guards direct callers.)
        out = torch.empty_like(spec)
        self._ck(self._L.dsvc_denoiser_forward(self._h, ptr(spec), ptr(t32), ptr(cond), ptr(out), B, T,
                                          1 if cond_changed else 0, stream_ptr()))
        return out

    def check(self):
        """Wait for the current stream and raise if any forward() since the last check saw a diffusion step outside the schedule."""
        self._ck(self._L.dsvc_denoiser_check(self._h, stream_ptr()))

    def debug_set(self, key, value):
        """Test support: 'stop_after_layers' (n >= 0, -1 = off) / 'two_launch_layer' (0 / 1) -- see include/dsvc.h."""
        self._ck(self._L.dsvc_denoiser_debug_set(self._h, key.encode(), int(value)))

    def debug_buffer(self, name):
        """Copy of an internal frame-major buffer as a [rows, ld] tensor (parity-test aid)."""
        rows, ld = ctypes.c_int32(0), ctypes.c_int32(0)
        self._ck(self._L.dsvc_denoiser_debug_buffer(self._h, name.encode(), None, 0, ctypes.byref(rows), ctypes.byref(ld)))
        out = torch.empty(rows.value, ld.value, device="cuda", dtype=torch.float32)
        self._ck(self._L.dsvc_denoiser_debug_buffer(self._h, name.encode(), ptr(out), out.numel(), None, None))
        return out

    def __del__(self):
        try:
            if self._h:
                self._L.dsvc_denoiser_destroy(self._h)
                self._h = ctypes.c_void_p(0)
        except Exception:
            pass


class SamplerHandle:
    """dsvc_sampler: schedule tables + DDPM / PLMS loops driving a DenoiserHandle."""

    def __init__(self, denoiser, state):
        self._h = ctypes.c_void_p(0)
        self._L = denoiser._L
        self.den = denoiser
        self._ck(self._L.dsvc_sampler_create(denoiser._h, ctypes.byref(self._h)))
        for k in SCHEDULE_KEYS:
            if k not in state:
                raise KeyError("schedule buffer '%s' missing from the checkpoint state dict" % k)
            h, p = host_f32(state[k].reshape(-1))
            self._ck(self._L.dsvc_sampler_load_tensor(self._h, k.encode(), p, h.numel()))
        self._ck(self._L.dsvc_sampler_finalize(self._h))

    def _ck(self, rc):
        check(rc, self._L)

    def stats(self):
        """dsvc_sampler_stats: {'capture_ddpm', 'capture_plms', 'graph_launches', 'buckets_allocated', 'bucket_reuses', 'graphs_alive'} since the
        handle was created -- how often a variable-length call sequence had to build a workspace bucket or capture a chain."""
        out = (ctypes.c_int64 * 6)()
        self._ck(self._L.dsvc_sampler_stats(self._h, out, 6))
        return dict(zip(("capture_ddpm", "capture_plms", "graph_launches", "buckets_allocated", "bucket_reuses", "graphs_alive"), [int(v) for v in out]))

    def phase_timing(self, enable=True):
        """Measurement aid (include/dsvc_debug.h: dsvc_sampler_phase_times): record HIP events at the phase boundaries of every later sample()."""
        self._ck(self._L.dsvc_sampler_phase_times(self._h, 1 if enable else 0, None))

    def phase_times(self):
        """{'prepare', 'init', 'chain', 'finish'} in ms of the last sample() call made with phase_timing on (waits for it)."""
        out = (ctypes.c_float * 4)()
        self._ck(self._L.dsvc_sampler_phase_times(self._h, 1, out))
        return dict(zip(("prepare", "init", "chain", "finish"), [float(v) for v in out]))

    def sample(self, cond, t_start, speedup=1, x_init=None, mel2ph=None, seed=0, first_clip=0, t_stop=0,
               use_graph=True, return_x=False, ref_mel=None, clip_ids=None, clip_lens=None, clip_lens_host=None):
        """cond [B,H,T] -> mel_out [B,T,M] (denormalised, masked).  See dsvc_sample_args.
        ref_mel [B,T,M]: use_gt_mel start (q_sample at t_start-1); clip_ids [B]: explicit Philox clip ids; clip_lens [B]: valid
        frames per clip (frames beyond are the convs' zero padding, as if the clip ran alone); clip_lens_host: the same lengths as a host
        sequence when the caller has them (scheduling only: the tile width of a ragged batch is then chosen by the tiles that have work)."""
        _need_cuda(cond, x_init, mel2ph, ref_mel, clip_ids, clip_lens)
        if cond.dim() != 3:
            raise RuntimeError("cond must be [B, hidden, T], got %s" % (tuple(cond.shape),))
        B, H, T = cond.shape
        M = self.den.mel_bins
        if H != self.den.hidden:
            raise RuntimeError("cond has %d channels, the denoiser was built for hidden_size %d" % (H, self.den.hidden))
        if x_init is not None and tuple(x_init.shape) != (B, 1, M, T):
            raise RuntimeError("x_init must be [B,1,M,T] = %s, got %s" % ((B, 1, M, T), tuple(x_init.shape)))
        if mel2ph is not None and tuple(mel2ph.shape) != (B, T):
            raise RuntimeError("mel2ph must be [B,T] = %s, got %s" % ((B, T), tuple(mel2ph.shape)))
        cond = cond.contiguous().float()
        mel = torch.empty(B, T, M, device=cond.device, dtype=torch.float32)
        xo = torch.empty(B, 1, M, T, device=cond.device, dtype=torch.float32) if return_x else None
        xi = x_init.contiguous().float() if x_init is not None else None
        m2p = mel2ph.to(torch.int32).contiguous() if mel2ph is not None else None
        if ref_mel is not None and tuple(ref_mel.shape) != (B, T, M):
            raise RuntimeError("ref_mel must be [B,T,M] = %s, got %s" % ((B, T, M), tuple(ref_mel.shape)))
        rm = ref_mel.contiguous().float() if ref_mel is not None else None
        ids = lens = None
        if clip_ids is not None:
            ids = clip_ids.to(torch.int32).contiguous()
            if ids.numel() != B:
                raise RuntimeError("clip_ids must hold %d entries" % B)
        if clip_lens is not None:
            lens = clip_lens.to(torch.int32).contiguous()
            if lens.numel() != B:
                raise RuntimeError("clip_lens must hold %d entries" % B)
        lens_h = None
        if clip_lens_host is not None and lens is not None:
            if len(clip_lens_host) != B:
                raise RuntimeError("clip_lens_host must hold %d entries" % B)
            lens_h = (ctypes.c_int32 * B)(*[int(v) for v in clip_lens_host])
        a = _lib.SampleArgs(B, T, cond.data_ptr(), xi.data_ptr() if xi is not None else None,
                            rm.data_ptr() if rm is not None else None,
                            m2p.data_ptr() if m2p is not None else None, seed, first_clip,
                            ids.data_ptr() if ids is not None else None, lens.data_ptr() if lens is not None else None,
                            ctypes.cast(lens_h, ctypes.c_void_p) if lens_h is not None else None,
                            t_start, t_stop,
                            int(speedup), 1 if use_graph else 0, mel.data_ptr(), xo.data_ptr() if xo is not None else None)
        self._ck(self._L.dsvc_sample(self._h, ctypes.byref(a), stream_ptr()))
        return (mel, xo) if return_x else mel

    def profile_gate_kernel(self, B, T, iters=5):
        """(average launch time in us, rows per launch, kind) of the dominant kernel at this batch size; kind 0 = the gate
        kernel of the two-launch layer, 1 = the fused residual-layer kernel."""
        us = ctypes.c_float(0)
        rows = ctypes.c_int64(0)
        kind = ctypes.c_int32(0)
        self._ck(self._L.dsvc_sampler_profile_gate_kernel(self._h, B, T, iters, ctypes.byref(us), ctypes.byref(rows), ctypes.byref(kind), stream_ptr()))
        return us.value, rows.value, kind.value

    def __del__(self):
        try:
            if self._h:
                self._L.dsvc_sampler_destroy(self._h)
                self._h = ctypes.c_void_p(0)
        except Exception:
            pass


class VocoderHandle:
    """dsvc_vocoder: NSF-HiFiGAN generator.  ``state`` is the checkpoint's 'generator' dict (weight-norm pairs
    included), ``h`` its config.json (modules/nsf_hifigan/models.py:14-30).  The 24 kHz HifiGanGenerator
    (modules/hifigan/hifigan.py:104-178) is the same network: ``mel_scale=1.0`` (its wrapper feeds natural-log mels unscaled),
    ``use_source`` = the config's ``use_pitch_embed``, ``sampling_rate`` from ``audio_sample_rate``."""

    def __init__(self, state, h, precision="f16_x3", mel_scale=2.30259, use_source=True):
        self._h = ctypes.c_void_p(0)
        rb = str(h.get("resblock", "1"))
        if rb not in ("1", "2"):
            raise ValueError("resblock must be '1' or '2' (modules/nsf_hifigan/models.py:337)")
        rates, ksz = list(h["upsample_rates"]), list(h["upsample_kernel_sizes"])
        rks, rds = list(h["resblock_kernel_sizes"]), [list(d) for d in h["resblock_dilation_sizes"]]
        # ResBlock1 builds convs1 / convs2 from dilation[0..2] (models.py:36-55), ResBlock2 exactly two convs from dilation[0] and dilation[1]
        # (models.py:77-82) -- whatever the length of the list: longer lists are accepted and their tail ignored, shorter ones raise an
        # IndexError in the reference's constructor (here: ValueError)
        ndil = 3 if rb == "1" else 2
        if len(rates) > 8 or len(rks) > 4 or not rds:
            raise ValueError("unsupported generator geometry")
        if any(len(d) < ndil for d in rds):
            raise ValueError("resblock '%s' needs %d dilations per kernel size, got %s (modules/nsf_hifigan/models.py:%s)"
                             % (rb, ndil, rds, "36-55" if rb == "1" else "77-82"))
        rds = [d[:ndil] for d in rds]
        cfg = _lib.VocoderCfg()
        sr = h["sampling_rate"] if "sampling_rate" in h else h["audio_sample_rate"]
        cfg.num_mels, cfg.upsample_initial_channel, cfg.sampling_rate = h["num_mels"], h["upsample_initial_channel"], sr
        cfg.mel_scale, cfg.use_source = float(mel_scale), 1 if use_source else 0
        self.use_source = bool(use_source)
        cfg.n_ups = len(rates)
        for i, (u, k) in enumerate(zip(rates, ksz)):
            cfg.upsample_rates[i], cfg.upsample_kernel_sizes[i] = u, k
        cfg.n_kernels = len(rks)
        for j, (k, ds) in enumerate(zip(rks, rds)):
            cfg.resblock_kernel_sizes[j] = k
            for m, d in enumerate(ds):
                cfg.resblock_dilations[j][m] = d
        cfg.resblock, cfg.n_dilations = int(rb), ndil
        cfg.harmonics = 8                                # harmonic_num=8, models.py:334
        cfg.precision = _prec(precision)
        self.cfg = cfg
        self.num_mels = h["num_mels"]
        self.hop = 1
        for u in rates:
            self.hop *= u
        check(lib().dsvc_vocoder_create(ctypes.byref(cfg), ctypes.byref(self._h)))
        for k, v in state.items():
            hbuf, p = host_f32(v)
            check(lib().dsvc_vocoder_load_tensor(self._h, k.encode(), p, hbuf.numel()))
        check(lib().dsvc_vocoder_finalize(self._h))

    def vocode(self, mel, f0, seed=0, first_clip=0, clip_ids=None):
        """mel [B,T,M] log10, f0 [B,T] Hz (0 = unvoiced) -> wav [B, T*hop].  clip_ids [B]: explicit Philox clip ids."""
        _need_cuda(mel, f0, clip_ids)
        B, T, M = mel.shape
        if self.use_source and f0 is None:
            raise ValueError("this generator has a harmonic source: f0 is required")
        if not self.use_source:
            f0 = None
        if M != self.num_mels or (f0 is not None and f0.shape != (B, T)):
            raise ValueError("shape mismatch: mel %s f0 %s" % (tuple(mel.shape), None if f0 is None else tuple(f0.shape)))
        mel = mel.contiguous().float()
        f0 = f0.contiguous().float() if f0 is not None else None
        wav = torch.empty(B, T * self.hop, device=mel.device, dtype=torch.float32)
        ids = clip_ids.to(torch.int32).contiguous() if clip_ids is not None else None
        if ids is not None and ids.numel() != B:
            raise ValueError("clip_ids must hold %d entries" % B)
        check(lib().dsvc_vocode(self._h, ptr(mel), ptr(f0), ptr(wav), B, T, seed, first_clip, ptr(ids), stream_ptr()))
        return wav

    def __del__(self):
        try:
            if self._h:
                lib().dsvc_vocoder_destroy(self._h)
                self._h = ctypes.c_void_p(0)
        except Exception:
            pass


class MelspecHandle:
    """dsvc_melspec: STFT -> mel -> log10.  mode 0: modules/nsf_hifigan/nvSTFT.py:72-104 + nsf_hifigan.py:86-91; mode 1: the centred,
    zero-padded front-end of process_utterance (preprocessing/data_gen_utils.py:124-136) with ``clip_val`` = its eps."""

    def __init__(self, sr, n_fft, win_size, hop, n_mels, fmin, fmax, clip_val=1e-5, mode=0):
        from .melfb import mel_filterbank
        self._h = ctypes.c_void_p(0)
        self.n_mels = n_mels
        self.n_fft = n_fft
        basis = torch.from_numpy(mel_filterbank(sr, n_fft, n_mels, fmin, fmax)).contiguous()
        cfg = _lib.MelspecCfg(n_fft, win_size, hop, n_mels, clip_val, mode)
        check(lib().dsvc_melspec_create(ctypes.byref(cfg), ctypes.c_void_p(basis.data_ptr()), ctypes.byref(self._h)))

    def frames(self, n_samples):
        t = ctypes.c_int32(0)
        check(lib().dsvc_melspec_frames(self._h, n_samples, ctypes.byref(t)))
        return t.value

    def mel(self, wav):
        """wav [B,N] in [-1,1] -> mel [B,T,n_mels] log10."""
        _need_cuda(wav)
        wav = wav.contiguous().float()
        B, N = wav.shape
        out = torch.empty(B, self.frames(N), self.n_mels, device=wav.device, dtype=torch.float32)
        check(lib().dsvc_melspec_run(self._h, ptr(wav), ptr(out), B, N, stream_ptr()))
        return out

    def mel_and_linear(self, wav, min_level_db):
        """wav [B,N] -> (mel [B,T,n_mels] log10, linear [B,T,n_fft/2+1]): process_utterance(return_linear=True)'s normalised dB spectrogram."""
        _need_cuda(wav)
        wav = wav.contiguous().float()
        B, N = wav.shape
        T = self.frames(N)
        mel = torch.empty(B, T, self.n_mels, device=wav.device, dtype=torch.float32)
        lin = torch.empty(B, T, self.n_fft // 2 + 1, device=wav.device, dtype=torch.float32)
        check(lib().dsvc_melspec_run_linear(self._h, ptr(wav), ptr(mel), ptr(lin), float(min_level_db), B, N, stream_ptr()))
        return mel, lin

    def __del__(self):
        try:
            if self._h:
                lib().dsvc_melspec_destroy(self._h)
                self._h = ctypes.c_void_p(0)
        except Exception:
            pass
