"""ctypes binding of libdsvc_hip.so (include/dsvc.h).  There is NO fallback: if the library is missing or
does not load, every entry point raises."""
import ctypes
import os

import torch  # noqa: F401  -- must be imported first so the HIP runtime torch ships is the one we bind to

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libdsvc_hip.so")

PREC_F16, PREC_F16_W2, PREC_F16_X3 = 0, 1, 2
PREC_F16_MIX = 3
PREC_F16_X3T = 4
PREC_F16_W6 = 5
PREC_F16_W6N = 6
PRECISIONS = {"f16": PREC_F16, "f16_w2": PREC_F16_W2, "f16_x3": PREC_F16_X3, "f16_x3t": PREC_F16_X3T}
ABI_VERSION = 9


def parse_precision(p):
    """'f16' | 'f16_w2' | 'f16_x3' | 'f16_x3t' (f16_x3's operand scheme on the tgemm engine) | 'f16_w6' / 'f16_w6dN' (f16_w2 with fp6 w_lo codes, N dithered roundings, default 64) | 'f16_dN' (fp16 operands, N time-dithered weight roundings) | 'f16_mN' (the same for the dilated
    conv, exact hi+lo weights for the output 1x1) -> (enum, variants)."""
    if isinstance(p, (tuple, list)):
        return int(p[0]), int(p[1])
    if isinstance(p, int):
        return p, -1                                     # (DSVC_VARIANTS_DEFAULT: the scheme's default number of dithered roundings, chosen by the library: 64 for the 6-bit schemes)
    if p == "f16_x3t":                                   # (64 dithered roundings of the fp6 w_lo codes its small tilings use since round 4)
        return PREC_F16_X3T, 64
    if p in PRECISIONS:
        return PRECISIONS[p], 1
    if p.startswith("f16_d") and p[5:].isdigit() and int(p[5:]) >= 1:
        return PREC_F16, int(p[5:])
    if p.startswith("f16_m") and p[5:].isdigit() and int(p[5:]) >= 1:
        return PREC_F16_MIX, int(p[5:])
    if p == "f16_w6":                                    # f16_w2 with the w_lo * x correction of the dilated conv on the 6-bit matrix instruction
        return PREC_F16_W6, 64                           # (fused layer kernel; w_lo as 64 time-dithered fp6 roundings)
    if p.startswith("f16_w6d") and p[7:].isdigit() and int(p[7:]) >= 1:
        return PREC_F16_W6, int(p[7:])
    if p == "f16_w6n":                                   # ... without the output projection's g_lo correction (f16_w2's error class, faster)
        return PREC_F16_W6N, 64
    raise ValueError("precision must be one of %s, 'f16_dN' or 'f16_mN'" % sorted(PRECISIONS))

c_f32p = ctypes.POINTER(ctypes.c_float)
c_i32p = ctypes.POINTER(ctypes.c_int32)


class DenoiserCfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ("mel_bins", "hidden", "channels", "layers", "dilation_cycle", "max_steps", "precision", "weight_variants")]


class SampleArgs(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("T", ctypes.c_int32), ("cond", ctypes.c_void_p), ("x_init", ctypes.c_void_p),
                ("ref_mel", ctypes.c_void_p), ("mel2ph", ctypes.c_void_p), ("seed", ctypes.c_uint64),
                ("first_clip", ctypes.c_int32), ("clip_ids", ctypes.c_void_p), ("clip_lens", ctypes.c_void_p), ("clip_lens_host", ctypes.c_void_p),
                ("t_start", ctypes.c_int32), ("t_stop", ctypes.c_int32),
                ("speedup", ctypes.c_int32), ("use_graph", ctypes.c_int32), ("mel_out", ctypes.c_void_p),
                ("x_out", ctypes.c_void_p)]


class VocoderCfg(ctypes.Structure):
    _fields_ = [("num_mels", ctypes.c_int32), ("upsample_initial_channel", ctypes.c_int32),
                ("sampling_rate", ctypes.c_int32), ("n_ups", ctypes.c_int32),
                ("upsample_rates", ctypes.c_int32 * 8), ("upsample_kernel_sizes", ctypes.c_int32 * 8),
                ("n_kernels", ctypes.c_int32), ("resblock_kernel_sizes", ctypes.c_int32 * 4),
                ("resblock_dilations", (ctypes.c_int32 * 3) * 4), ("harmonics", ctypes.c_int32),
                ("precision", ctypes.c_int32), ("mel_scale", ctypes.c_float), ("use_source", ctypes.c_int32),
                ("resblock", ctypes.c_int32), ("n_dilations", ctypes.c_int32)]


class MelspecCfg(ctypes.Structure):
    _fields_ = [("n_fft", ctypes.c_int32), ("win_size", ctypes.c_int32), ("hop", ctypes.c_int32),
                ("n_mels", ctypes.c_int32), ("clip_val", ctypes.c_float), ("mode", ctypes.c_int32)]


class TrainerCfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ("mel_bins", "hidden", "channels", "layers", "dilation_cycle", "timesteps", "loss_l1", "pitch_vocab")]


class PeCfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ("n_mel", "hidden", "predictor_hidden", "prenet_layers", "conv_layers", "predictor_layers", "kernel",
                 "predictor_kernel", "pitch_norm", "use_uv")] + [("f0_mean", ctypes.c_float), ("f0_std", ctypes.c_float)]


class TrainArgs(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("T", ctypes.c_int32), ("mel", ctypes.c_void_p), ("cond", ctypes.c_void_p), ("t", ctypes.c_void_p),
                ("pitch", ctypes.c_void_p), ("mel2ph", ctypes.c_void_p), ("seed", ctypes.c_uint64), ("first_clip", ctypes.c_int32),
                ("clip_ids", ctypes.c_void_p)]


# every symbol include/dsvc.h declares: (name, restype, argtypes)
_VP = ctypes.c_void_p
SYMBOLS = [
    ("dsvc_abi_version", ctypes.c_int, []),
    ("dsvc_last_error", ctypes.c_char_p, []),
    ("dsvc_probe_mfma", ctypes.c_int, [ctypes.c_int32, c_f32p, c_f32p, _VP]),
    ("dsvc_probe_mfma_detail", ctypes.c_int, [ctypes.c_int32, c_f32p, _VP]),
    ("dsvc_denoiser_create", ctypes.c_int, [ctypes.POINTER(DenoiserCfg), ctypes.POINTER(_VP)]),
    ("dsvc_denoiser_load_tensor", ctypes.c_int, [_VP, ctypes.c_char_p, _VP, ctypes.c_int64]),
    ("dsvc_denoiser_finalize", ctypes.c_int, [_VP]),
    ("dsvc_denoiser_destroy", None, [_VP]),
    ("dsvc_denoiser_forward", ctypes.c_int, [_VP, _VP, _VP, _VP, _VP, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _VP]),
    ("dsvc_denoiser_debug_buffer", ctypes.c_int, [_VP, ctypes.c_char_p, _VP, ctypes.c_int64, c_i32p, c_i32p]),
    ("dsvc_denoiser_check", ctypes.c_int, [_VP, _VP]),
    ("dsvc_denoiser_debug_set", ctypes.c_int, [_VP, ctypes.c_char_p, ctypes.c_int32]),
    ("dsvc_sampler_create", ctypes.c_int, [_VP, ctypes.POINTER(_VP)]),
    ("dsvc_sampler_load_tensor", ctypes.c_int, [_VP, ctypes.c_char_p, _VP, ctypes.c_int64]),
    ("dsvc_sampler_finalize", ctypes.c_int, [_VP]),
    ("dsvc_sampler_destroy", None, [_VP]),
    ("dsvc_sample", ctypes.c_int, [_VP, ctypes.POINTER(SampleArgs), _VP]),
    ("dsvc_sampler_stats", ctypes.c_int, [_VP, ctypes.POINTER(ctypes.c_int64), ctypes.c_int32]),
    ("dsvc_sampler_phase_times", ctypes.c_int, [_VP, ctypes.c_int32, c_f32p]),
    ("dsvc_sampler_profile_gate_kernel", ctypes.c_int,
     [_VP, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64), c_i32p, _VP]),
    ("dsvc_vocoder_create", ctypes.c_int, [ctypes.POINTER(VocoderCfg), ctypes.POINTER(_VP)]),
    ("dsvc_vocoder_load_tensor", ctypes.c_int, [_VP, ctypes.c_char_p, _VP, ctypes.c_int64]),
    ("dsvc_vocoder_finalize", ctypes.c_int, [_VP]),
    ("dsvc_vocoder_destroy", None, [_VP]),
    ("dsvc_vocode", ctypes.c_int, [_VP, _VP, _VP, _VP, ctypes.c_int32, ctypes.c_int32, ctypes.c_uint64, ctypes.c_int32, _VP, _VP]),
    ("dsvc_melspec_create", ctypes.c_int, [ctypes.POINTER(MelspecCfg), _VP, ctypes.POINTER(_VP)]),
    ("dsvc_melspec_destroy", None, [_VP]),
    ("dsvc_melspec_frames", ctypes.c_int, [_VP, ctypes.c_int64, ctypes.POINTER(ctypes.c_int32)]),
    ("dsvc_melspec_run", ctypes.c_int, [_VP, _VP, _VP, ctypes.c_int32, ctypes.c_int64, _VP]),
    ("dsvc_melspec_run_linear", ctypes.c_int, [_VP, _VP, _VP, _VP, ctypes.c_float, ctypes.c_int32, ctypes.c_int64, _VP]),
    ("dsvc_pitch_coarse", ctypes.c_int, [_VP, _VP, _VP, _VP, ctypes.c_int32, ctypes.c_int64, _VP, _VP, _VP]),
    ("dsvc_cond_build", ctypes.c_int, [_VP, _VP, _VP, _VP, _VP, ctypes.c_int32, _VP, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                       ctypes.c_int32, _VP, _VP, _VP, _VP, _VP]),
    ("dsvc_hubert_create", ctypes.c_int, [ctypes.POINTER(_VP)]),
    ("dsvc_hubert_load_tensor", ctypes.c_int, [_VP, ctypes.c_char_p, _VP, ctypes.c_int64]),
    ("dsvc_hubert_finalize", ctypes.c_int, [_VP]),
    ("dsvc_hubert_destroy", None, [_VP]),
    ("dsvc_hubert_frames", ctypes.c_int, [ctypes.c_int64, ctypes.POINTER(ctypes.c_int32)]),
    ("dsvc_hubert_units", ctypes.c_int, [_VP, _VP, ctypes.c_int64, _VP, _VP]),
    ("dsvc_pe_create", ctypes.c_int, [ctypes.POINTER(PeCfg), ctypes.POINTER(_VP)]),
    ("dsvc_pe_load_tensor", ctypes.c_int, [_VP, ctypes.c_char_p, _VP, ctypes.c_int64]),
    ("dsvc_pe_finalize", ctypes.c_int, [_VP]),
    ("dsvc_pe_set_positions", ctypes.c_int, [_VP, _VP, ctypes.c_int32]),
    ("dsvc_pe_destroy", None, [_VP]),
    ("dsvc_pe_run", ctypes.c_int, [_VP, _VP, ctypes.c_int32, ctypes.c_int32, _VP, _VP, _VP]),
    ("dsvc_trainer_create", ctypes.c_int, [ctypes.POINTER(TrainerCfg), ctypes.POINTER(_VP)]),
    ("dsvc_trainer_destroy", None, [_VP]),
    ("dsvc_trainer_param_count", ctypes.c_int, [_VP, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    ("dsvc_trainer_param_info", ctypes.c_int, [_VP, ctypes.c_int64, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_int64),
                                               ctypes.POINTER(ctypes.c_int64)]),
    ("dsvc_trainer_bind", ctypes.c_int, [_VP, _VP, _VP]),
    ("dsvc_trainer_set_schedule", ctypes.c_int, [_VP, _VP, _VP, ctypes.c_int32, _VP, _VP, ctypes.c_int32]),
    ("dsvc_trainer_step", ctypes.c_int, [_VP, ctypes.POINTER(TrainArgs), _VP, _VP]),
    ("dsvc_trainer_step_begin", ctypes.c_int, [_VP, ctypes.POINTER(TrainArgs), _VP]),
    ("dsvc_trainer_check", ctypes.c_int, [_VP, _VP]),
    ("dsvc_trainer_debug_set", ctypes.c_int, [_VP, ctypes.c_char_p, ctypes.c_int32]),
    ("dsvc_trainer_step_layers", ctypes.c_int, [_VP, ctypes.c_int32, ctypes.c_int32, _VP]),
    ("dsvc_trainer_step_end", ctypes.c_int, [_VP, _VP, _VP]),
    ("dsvc_adamw_step", ctypes.c_int, [_VP, _VP, _VP, _VP, ctypes.c_int64, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                       ctypes.c_float, ctypes.c_int64, _VP, ctypes.c_float, _VP]),
    ("dsvc_grad_clip_coef", ctypes.c_int, [_VP, ctypes.c_int64, ctypes.c_float, _VP, _VP, _VP]),
]

_libs = {}                      # path -> loaded library (product, test-hooks and profiling builds can coexist: every internal is hidden)
HOOKS_PATH = os.path.join(HERE, "libdsvc_hip_hooks.so")


def use_profiling_build():
    """Profiling tools only (tools/): bind to libdsvc_hip_prof.so, the -DDSVC_PROFILING build whose kernels carry the ablation knobs
    (``python -m diffsvc_amd.build --profiling``).  Must be called before the first ``lib()``."""
    global LIB_PATH
    if _libs:
        raise RuntimeError("the library is already loaded")
    LIB_PATH = os.path.join(HERE, "libdsvc_hip_prof.so")


def _load(path):
    handle = _libs.get(path)
    if handle is None:
        if not os.path.exists(path):
            raise RuntimeError("%s not found at %s: build it with `python -m diffsvc_amd.build` "
                               "(there is no CPU fallback)" % (os.path.basename(path), path))
        handle = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
        for name, res, args in SYMBOLS:
            fn = getattr(handle, name)          # AttributeError if the ABI symbol is missing
            fn.restype = res
            fn.argtypes = args
        if handle.dsvc_abi_version() != ABI_VERSION:
            raise RuntimeError("%s ABI version mismatch" % os.path.basename(path))
        _libs[path] = handle
    return handle


def lib():
    """Load the HIP library (once).  Raises if it is missing -- build it with diffsvc_amd.build.build()."""
    return _load(LIB_PATH)


class hooks_build:
    """``with _lib.hooks_build(): ...`` -- handles created inside bind to libdsvc_hip_hooks.so, the TEST-HOOKS build: the product library plus the
    ``dsvc_*_debug_set`` keys that change which kernel computes a result (per-layer taps, the A/B partners of the fused kernels and of the
    6-bit products).  Same sources, same kernels; only csrc/diffnet.hip and csrc/train.hip are compiled a second time with -DDSVC_TEST_HOOKS,
    which switches on branches of those two host functions and nothing else.  The product library refuses those keys.  A handle keeps the
    library it was created in for its lifetime (``DenoiserHandle._L``), so hooks handles and product handles coexist in one process."""

    def __enter__(self):
        global LIB_PATH
        self._prev = LIB_PATH
        if not LIB_PATH.endswith("_prof.so"):              # (the profiling build carries the hooks as well)
            LIB_PATH = HOOKS_PATH
        return lib()

    def __exit__(self, *exc):
        global LIB_PATH
        LIB_PATH = self._prev
        return False


def check(rc, L=None):
    if rc != 0:
        msg = (L or lib()).dsvc_last_error()
        raise RuntimeError("dsvc error %d: %s" % (rc, msg.decode() if msg else "?"))


def stream_ptr():
    """The caller's current HIP stream as void* (torch.cuda.current_stream)."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def host_f32(t):
    """Contiguous fp32 host copy of a checkpoint tensor + its pointer."""
    h = t.detach().to("cpu", torch.float32).contiguous()
    return h, ctypes.c_void_p(h.data_ptr())
