"""``DiffNetHip`` -- drop-in for the reference denoiser ``network.diff.net.DiffNet`` (net.py:86-135).

Same constructor argument, same parameter names and shapes (so ``load_state_dict(strict=True)`` of a reference
checkpoint works), same ``forward(spec, diffusion_step, cond)`` contract; the arithmetic runs
in libdsvc_hip.so.  Register it in the reference's seam with

    DIFF_DECODERS['wavenet'] = lambda hp: DiffNetHip(hp['audio_num_mel_bins'])

(infer_tools/infer_tool.py:107-111).  ``precision`` selects the operand scheme of the two big per-layer contractions
(include/dsvc.h): ``"f16_w2"`` = exact weights as fp16 hi + lo planes (2 MFMAs per product), fp16 activations; ``"f16_dN"`` = one fp16
MFMA per product with N time-dithered weight roundings; ``"f16_mN"`` the same for the dilated conv with exact (hi + lo) weights for the
output 1x1; ``"f16_x3"`` splits the activations as well (3 MFMAs, fp32-class).  The default ``"auto"`` picks per sampler:

* DDPM, up to 6 ten-second clips per call (< ``BATCHED_FRAMES`` mel frames): ``f16_x3t`` -- fp32-class (hi + lo weights AND split
  activations, 3 MFMAs) on the tgemm engine.  The small-batch kernels are bound by the weight stream and by latency, not by MFMAs, so the
  third MFMA costs 14 ... 20 % there (0.418 ms per step for one clip against 0.366 at f16_w2) and buys 3.3e-5 ... 4.9e-5 mel error after 1000
  steps on every real-reference golden instead of 6e-4 ... 9e-4 (round 3; profiles/r3l_auto_sweep.txt, r3l_pytest_gpu.txt).  Round 4: inside
  the sampler's DDPM loop the w_lo * x_hi product of these tilings is one 6-bit MFMA per 64 input channels on time-dithered fp6 codes of w_lo
  (0.391 instead of 0.417 ms per step; 7e-5 ... 9e-5 on the 21 goldens, conditioned checkpoints included); PLMS and ``forward()`` keep the
  fp16 lo plane -- PLMS amplifies the 1e-5-relative weight error of a single dither variant to 4.7e-4 (measured).
* DDPM, larger calls: ``f16_w6`` (round 4) -- f16_w2's operand scheme (exact weights as fp16 hi + w_lo, fp16 activations) with every
  correction term on the block-scaled 6-bit matrix instruction inside the fused layer kernel (w_lo as time-dithered fp6 codes against
  bf6(x) converted in registers: 16 fp16 + 4 six-bit MFMAs per 64 input channels instead of 32 fp16 ones), plus a 6-bit correction of the
  gate output's own fp16 rounding in the output 1x1.  The maximum mel error of a 1000-step chain is a heavy-tailed statistic: on the 31 + n
  real-reference goldens of two 32-clip batches it is 4.0e-4 ... 5.7e-4 (Gumbel fit: P(a clip > 1e-3) ~ 5e-9), conditioned checkpoints
  1.3e-4 ... 1.7e-4; f16_w2 itself measures 6.2e-4 ... 9.1e-4 there (P ~ 1e-3 per clip -- one 256-clip job in four holds a clip over the
  bar on a random-init checkpoint) at 9 % more time per step.  Round 5: calls between the batched threshold (48 tiles of 128 rows = 7
  ten-second clips) and the 128-frame tiling's minimum (120 tiles = 18 clips) run the SAME kernel -- 6-bit corrections included -- on 64- or
  32-frame tiles (until round 4 they fell back to f16_w2 on the two-launch tilings: the scheme whose error tail failed the ship bar).
  ``f16_w6n`` = the same without the gate-output correction: f16_w2's error class, another 6 % faster.
* PLMS/PNDM, whose Adams-Bashforth extrapolation amplifies a single evaluation's rounding: ``f16_x3t`` (7e-6 on the 50-iteration golden
  at T=861; 26 ms per 10 s clip).  With fp16 activations even exact weights leave that chain at (8.2 +- 1.2)e-4 over ten (clip, noise)
  pairs, one of them at 1.08e-3 (profiles/r2w_precision_spread.txt).
``forward`` is inference only (no autograd through a single evaluation); training goes through the sampler module's ``infer=False`` branch
(``GaussianDiffusionHip.forward``: loss and gradients of every parameter of this module from ``dsvc_trainer_step``) or through
``diffsvc_amd.train.DiffusionTrainerHip``.
"""
import math

import torch
from torch import nn

from .engine import DenoiserHandle
from .hparams import get_hparams


class _Mish(nn.Module):                      # placeholder so the MLP's keys are mlp.0.* / mlp.2.* (net.py:99-103)
    def forward(self, x):
        return x * torch.tanh(nn.functional.softplus(x))


class _ResidualBlockParams(nn.Module):
    """Parameter container mirroring ResidualBlock (net.py:58-64)."""

    def __init__(self, encoder_hidden, residual_channels, dilation):
        super().__init__()
        self.dilated_conv = nn.Conv1d(residual_channels, 2 * residual_channels, 3, padding=dilation, dilation=dilation)
        nn.init.kaiming_normal_(self.dilated_conv.weight)
        self.diffusion_projection = nn.Linear(residual_channels, residual_channels)
        self.conditioner_projection = nn.Conv1d(encoder_hidden, 2 * residual_channels, 1)
        nn.init.kaiming_normal_(self.conditioner_projection.weight)
        self.output_projection = nn.Conv1d(residual_channels, 2 * residual_channels, 1)
        nn.init.kaiming_normal_(self.output_projection.weight)


class DiffNetHip(nn.Module):
    AUTO = {"ddpm": "f16_x3t", "ddpm_batched": "f16_w6", "plms": "f16_x3t", "plms_coarse": "f16_x3t", "forward": "f16_x3t"}
    BATCHED_FRAMES = 6000          # B * T from which a DDPM call takes the batched precision when only the frame count is known (7 ten-second
                                   # clips: f16_x3t costs +14 ... 20 % below that and +50 ... 90 % above, profiles/r3l_auto_sweep.txt)
    BATCHED_TILES = 48             # ... and, when the batch shape is known, the rule itself: the call's rows fill >= 48 tiles of 128 rows -- from
                                   # there the C library runs f16_w6 on its fused layer kernel (csrc/diffnet.hip: fused_nt), below it the handle
                                   # would fall back to f16_w2 on the two-launch tilings, so `auto` stays on f16_x3t

    def __init__(self, in_dims=80, hparams=None, precision="auto"):
        super().__init__()
        hp = hparams if hparams is not None else get_hparams()
        self.in_dims = in_dims
        self.encoder_hidden = hp["hidden_size"]
        self.n_layers = hp["residual_layers"]
        self.channels = C = hp["residual_channels"]
        self.dilation_cycle = hp["dilation_cycle_length"]
        self.max_steps = int(hp.get("timesteps", 1000))
        self.precision = precision
        self.input_projection = nn.Conv1d(in_dims, C, 1)
        nn.init.kaiming_normal_(self.input_projection.weight)
        self.mlp = nn.Sequential(nn.Linear(C, C * 4), _Mish(), nn.Linear(C * 4, C))
        self.residual_layers = nn.ModuleList([
            _ResidualBlockParams(self.encoder_hidden, C, 2 ** (i % self.dilation_cycle)) for i in range(self.n_layers)])
        self.skip_projection = nn.Conv1d(C, C, 1)
        nn.init.kaiming_normal_(self.skip_projection.weight)
        self.output_projection = nn.Conv1d(C, in_dims, 1)
        nn.init.zeros_(self.output_projection.weight)           # net.py:110
        self._handles = {}             # precision -> (DenoiserHandle, params key): packed device weights are derived state
        self._cond_ref = None          # the cond tensor whose hoisted projections the C handle holds (a strong reference, so the
        self._cond_ver = -1            # caching allocator cannot hand its address to a different tensor: no ABA on data_ptr)

    # -- C handle management: rebuilt whenever a parameter tensor changes (load_state_dict, .to(), in-place edits)
    def _params_key(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters()) + (self.precision,)

    def workspace_tiles(self, clips, T):
        """128-row tiles of the C workspace for a [clips, T] call (csrc/diffnet.hip: bucket_rows -- a clip occupies round_up(T + largest
        dilation, 128) rows: the BUCKET every T up to that size shares, workspace and captured graphs included)."""
        max_dil = 2 ** min(self.dilation_cycle - 1, self.n_layers - 1)
        return clips * ((T + max_dil + 127) // 128)

    def precision_for(self, use, speedup=1, frames=None, clips=None):
        """The operand precision used for ``use`` in {'ddpm', 'plms', 'forward'} (PLMS: also by its step interval; DDPM: also by the
        call's size, ``frames`` = B * T, with ``clips`` = B the exact rule of BATCHED_TILES)."""
        if self.precision != "auto":
            return self.precision
        if use == "plms" and speedup > 20:
            use = "plms_coarse"
        if use == "ddpm" and frames is not None:
            if self.channels not in (256, 384):
                batched = False                          # (the fused layer kernel is built for 2 or 3 channel blocks of 128: csrc/tlayer.h)
            elif clips is not None and clips > 0:
                batched = self.workspace_tiles(clips, frames // clips) >= self.BATCHED_TILES
            else:
                batched = frames >= self.BATCHED_FRAMES
            if batched:
                use = "ddpm_batched"
        return self.AUTO[use]

    def handle(self, use="forward", speedup=1, frames=None, clips=None):
        """The C handle for a use; rebuilt whenever a parameter tensor changes (load_state_dict, .to(), in-place edits)."""
        prec = self.precision_for(use, speedup, frames, clips)
        key = self._params_key()
        cur = self._handles.get(prec)
        if cur is None or cur[1] != key:
            # drop handles packed from older parameters first: a 64-variant handle is 3 GB of device memory
            self._handles = {p: hk for p, hk in self._handles.items() if hk[1] == key}
            cur = (DenoiserHandle(self.state_dict(), self.in_dims, self.encoder_hidden, self.channels, self.n_layers,
                                  self.dilation_cycle, self.max_steps, precision=prec), key)
            self._handles[prec] = cur
            self.invalidate_cond()
        return cur[0]

    def check(self):
        """Wait for the stream and raise IndexError if a forward() since the last check passed a diffusion step outside [0, timesteps) -- the
        reference's step embedding accepts any float, its sampler's extract() raises on such a t (diffusion.py:22-25); forward() itself runs
        on the clamped step rather than synchronise 1000 times per clip.  Call it wherever the result is read back anyway."""
        for h, _ in self._handles.values():
            try:
                h.check()
            except RuntimeError as ex:
                raise IndexError(str(ex))

    def invalidate_cond(self):
        """Forget which cond the C handle's hoisted conditioner projections belong to (the sampler path overwrites them)."""
        self._cond_ref, self._cond_ver = None, -1

    def forward(self, spec, diffusion_step, cond):
        """spec [B,1,M,T], diffusion_step [B] (long), cond [B,H,T] -> [B,1,M,T]   (net.py:112-135)"""
        h = self.handle("forward")
        # a sampler loop calls with the SAME cond tensor 1000 times: the hoisted conditioner projections are recomputed only
        # when the tensor object or its version counter changes (identity, not address: see __init__)
        changed = not (cond is self._cond_ref and cond._version == self._cond_ver)
        out = h.forward(spec, diffusion_step.reshape(-1), cond, cond_changed=changed)
        self._cond_ref, self._cond_ver = cond, cond._version
        return out
