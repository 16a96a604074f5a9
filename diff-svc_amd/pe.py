"""``PitchExtractorHip`` -- the mel -> f0 network of the 24 kHz path on the HIP kernels (modules/fastspeech/pe.py:120-148).

It stands where the reference's ``PitchExtractor`` stands (infer_tools/infer_tool.py:134-136, training/task/tts.py:111-114):

    pe = PitchExtractorHip().cuda()
    pe.load_state_dict(ckpt_state, strict=True)       # what utils.load_ckpt(pe, hparams['pe_ckpt'], 'model', strict=True) ends in
    pe.eval()
    f0 = pe(outputs['mel_out'])['f0_denorm_pred']     # infer_tool.py:165-166

Inference only (eval-mode BatchNorm, no dropout); there is no CPU path."""
import ctypes
import math
from collections import OrderedDict

import torch

from ._lib import PeCfg, check, host_f32, lib, ptr, stream_ptr

_PITCH_NORM = {"log": 0, "standard": 1}


def expected_keys(n_mel, hidden, predictor_hidden, conv_layers, kernel=5, predictor_kernel=5):
    """name -> shape of PitchExtractor(n_mel_bins, conv_layers).state_dict()  (pe.py:121-134)."""
    H, P = hidden, predictor_hidden
    ks = OrderedDict()
    for l in range(3):
        q = "mel_prenet.layers.%d." % l
        ks[q + "0.weight"] = (H, n_mel if l == 0 else H, kernel)
        ks[q + "0.bias"] = (H,)
        for nm in ("weight", "bias", "running_mean", "running_var"):
            ks[q + "2." + nm] = (H,)
        ks[q + "2.num_batches_tracked"] = ()
    ks["mel_prenet.out_proj.weight"] = (H, H)
    ks["mel_prenet.out_proj.bias"] = (H,)
    if conv_layers > 0:
        ks["mel_encoder.in_proj.weight"] = (H, H)
        ks["mel_encoder.in_proj.bias"] = (H,)
        for l in range(conv_layers):
            q = "mel_encoder.conv.%d." % l
            ks[q + "conv.conv.weight"] = (H, H, kernel)
            ks[q + "conv.conv.bias"] = (H,)
            ks[q + "norm.weight"] = (H,)
            ks[q + "norm.bias"] = (H,)
        ks["mel_encoder.out_proj.weight"] = (H, H)
        ks["mel_encoder.out_proj.bias"] = (H,)
    for l in range(5):
        q = "pitch_predictor.conv.%d." % l
        ks[q + "1.weight"] = (P, H if l == 0 else P, predictor_kernel)
        ks[q + "1.bias"] = (P,)
        ks[q + "3.weight"] = (P,)
        ks[q + "3.bias"] = (P,)
    ks["pitch_predictor.linear.weight"] = (2, P)
    ks["pitch_predictor.linear.bias"] = (2,)
    ks["pitch_predictor.embed_positions._float_tensor"] = (1,)
    ks["pitch_predictor.pos_embed_alpha"] = (1,)
    return ks


def position_table(n_rows, dim):
    """The constant table of SinusoidalPositionalEmbedding(dim, padding_idx=0) (common_layers.py:105-122): row p holds
    sin(p * w_i) | cos(p * w_i) with w_i = 10000^(-i / (dim/2 - 1)); row 0 (padding) is zero.  Built with the same fp32 torch ops in
    the same shapes as the reference so the table is bit-identical."""
    half = dim // 2
    w = torch.exp(torch.arange(half, dtype=torch.float) * -(math.log(10000) / (half - 1)))
    phase = torch.arange(n_rows, dtype=torch.float).unsqueeze(1) * w.unsqueeze(0)
    cols = [torch.sin(phase), torch.cos(phase)]
    if dim % 2 == 1:
        cols.append(torch.zeros(n_rows, 1))
    t = torch.cat(cols, dim=1)
    t[0] = 0
    return t.contiguous()


class PitchExtractorHip:
    def __init__(self, n_mel_bins=80, conv_layers=2, hparams=None):
        from .hparams import get_hparams
        hp = hparams if hparams is not None else get_hparams()
        self.hp = hp
        self.n_mel_bins = n_mel_bins
        self.conv_layers = conv_layers
        self.hidden_size = hp["hidden_size"]
        self.predictor_hidden = hp["predictor_hidden"] if hp["predictor_hidden"] > 0 else self.hidden_size
        if hp["ffn_padding"] != "SAME":
            raise NotImplementedError("PitchExtractorHip: only ffn_padding 'SAME' (the shipped configs) is implemented")
        self._state = None
        self._h = None
        self._cfg_key = None
        self._table_rows = 0
        self.training = False

    # ---- the nn.Module surface the reference touches ----
    def cuda(self, *a, **k):
        if not torch.cuda.is_available():
            raise RuntimeError("PitchExtractorHip needs a HIP device (there is no CPU path)")
        return self

    to = cuda

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("PitchExtractorHip is inference-only (eval-mode BatchNorm, no dropout)")
        return self

    def state_dict(self):
        if self._state is None:
            raise RuntimeError("PitchExtractorHip: no weights loaded")
        return OrderedDict(self._state)

    def load_state_dict(self, state, strict=True):
        want = expected_keys(self.n_mel_bins, self.hidden_size, self.predictor_hidden, self.conv_layers,
                             predictor_kernel=self.hp["predictor_kernel"])
        missing = [k for k in want if k not in state]
        unexpected = [k for k in state if k not in want]
        if strict and (missing or unexpected):
            raise RuntimeError("Error(s) in loading state_dict for PitchExtractorHip: missing %s, unexpected %s" % (missing, unexpected))
        for k, shape in want.items():
            if k in state and tuple(state[k].shape) != tuple(shape):
                raise RuntimeError("size mismatch for %s: checkpoint %s, model %s" % (k, tuple(state[k].shape), tuple(shape)))
        needed = [k for k in missing if not k.endswith(("num_batches_tracked", "_float_tensor"))]
        if needed:
            raise RuntimeError("PitchExtractorHip cannot run without %s" % needed)
        self._state = OrderedDict((k, state[k].detach().cpu().clone()) for k in want if k in state)
        self._release()
        return self

    # ---- device handle ----
    def _cfg(self):
        hp = self.hp
        use_uv = hp["pitch_type"] == "frame" and bool(hp["use_uv"])                     # pe.py:145 + pitch_utils.py:72
        return PeCfg(self.n_mel_bins, self.hidden_size, self.predictor_hidden, 3, self.conv_layers, 5, 5, hp["predictor_kernel"],
                     _PITCH_NORM.get(hp["pitch_norm"], 2), int(use_uv), float(hp.get("f0_mean", 0.0)), float(hp.get("f0_std", 1.0)))

    def _handle(self, T):
        cfg = self._cfg()
        key = tuple(getattr(cfg, f) for f, _ in PeCfg._fields_)
        if self._h is None or key != self._cfg_key:                    # hparams are read at call time, as the reference does
            if self._state is None:
                raise RuntimeError("PitchExtractorHip: no weights loaded")
            self._release()
            h = ctypes.c_void_p(0)
            check(lib().dsvc_pe_create(ctypes.byref(cfg), ctypes.byref(h)))
            self._h = h
            for k, v in self._state.items():
                if k.endswith(("num_batches_tracked", "_float_tensor")):
                    continue
                t, p = host_f32(v)
                check(lib().dsvc_pe_load_tensor(h, k.encode(), p, t.numel()))
            check(lib().dsvc_pe_finalize(h))
            self._cfg_key = key
            self._table_rows = 0
        if T + 1 > self._table_rows:                                   # init_size 4096, regrown to exactly what a longer input needs
            rows = max(4096, T + 1)
            t, p = host_f32(position_table(rows, self.hidden_size))
            check(lib().dsvc_pe_set_positions(self._h, p, rows))
            self._table_rows = rows
        return self._h

    def _release(self):
        if self._h is not None:
            lib().dsvc_pe_destroy(self._h)
            self._h = None
            self._cfg_key = None
            self._table_rows = 0

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # ---- PitchExtractor.forward (pe.py:136-148) ----
    def forward(self, mel_input=None):
        if not mel_input.is_cuda:
            raise RuntimeError("diffsvc_amd: mel_input must live on the HIP device; there is no CPU path")
        if mel_input.dim() != 3 or mel_input.shape[-1] != self.n_mel_bins:
            raise ValueError("mel_input must be [B, T, %d], got %s" % (self.n_mel_bins, tuple(mel_input.shape)))
        mel = mel_input.contiguous().float()
        B, T, _ = mel.shape
        h = self._handle(T)
        pred = torch.empty(B, T, 2, device=mel.device, dtype=torch.float32)
        f0 = torch.empty(B, T, device=mel.device, dtype=torch.float32)
        check(lib().dsvc_pe_run(h, ptr(mel), B, T, ptr(pred), ptr(f0), stream_ptr()))
        return {"pitch_pred": pred, "f0_denorm_pred": f0}

    __call__ = forward
