"""Condition builder -- the ``no_fs2: true`` branch of ``FastSpeech2.forward`` (modules/fastspeech/fs2.py:94-154)
with ``add_pitch`` (fs2.py:185-238): cond = (gather(pad(hubert), mel2ph) + pitch_embed[coarse(2**f0)]) * (mel2ph>0).

Index work (f0 -> coarse pitch bin) is kept exact: for host tensors with the same fp32 torch-CPU ops the
reference CPU path uses; for device tensors by ``dsvc_pitch_coarse`` (csrc/cond.hip), a binary search over the fp32 thresholds
at which that very expression steps to the next bin (``coarse_thresholds``: bisection with the reference expression, once per
hparams) -- no device log/pow decides a bin and nothing leaves the device.  The gather / embedding / mask are exact data movement.
Registered under the attribute name ``fs2`` so a reference checkpoint's ``fs2.*`` keys load strictly; the
parameters the disabled FastSpeech2 branches own (mel_out, pitch_predictor) are accepted and kept as buffers.

``use_energy_embed`` (fs2.py:81-82,143-144,240-247; round 4): decoder_inp additionally gets energy_embed[clamp(energy * 256 // 4, max=255)]
before the mask -- an exact embedding lookup, done with torch ops behind the device builder (both shipped configs leave it off).  The
speaker branches (``use_spk_id`` / ``use_spk_embed``) stay rejected: the reference's own constructor has ``spk_embed_proj`` commented out
(fs2.py:32-39, "not used"), so its forward raises AttributeError on them and no checkpoint of this architecture carries such a table."""
import numpy as np
import torch
from torch import nn
import torch.nn.functional as F


def f0_to_coarse(f0, hp):
    """utils/pitch_utils.py:17-31 (torch branch), fp32, on the CPU."""
    f0_bin, f0_max, f0_min = hp["f0_bin"], hp["f0_max"], hp["f0_min"]
    mel_min = 1127 * np.log(1 + f0_min / 700)
    mel_max = 1127 * np.log(1 + f0_max / 700)
    m = 1127 * (1 + f0 / 700).log()
    m[m > 0] = (m[m > 0] - mel_min) * (f0_bin - 2) / (mel_max - mel_min) + 1
    m[m <= 1] = 1
    m[m > f0_bin - 1] = f0_bin - 1
    coarse = (m + 0.5).long()
    assert coarse.max() <= 255 and coarse.min() >= 1, (coarse.max(), coarse.min())
    return coarse


_THRESHOLDS = {}


def _coarse_of(x, hp):
    """The reference expression on ONE normalised pitch value: f0_to_coarse(2 ** x) as a [1, 1] fp32 tensor (scalar code path)."""
    return int(f0_to_coarse(2 ** torch.tensor([[x]], dtype=torch.float32), hp)[0, 0])


def coarse_thresholds(hp):
    """fp32 thresholds thr[0..f0_bin-3] of the normalised pitch x = log2(f0): f0_to_coarse(2**x) == 1 + #{k: x >= thr[k]}.
    thr[k] is the smallest fp32 x whose bin is >= k + 2, found by bisection over the (ordered) bit patterns of positive floats with
    the reference expression itself; the result is checked for a clean step at every threshold and on a random sample."""
    key = (hp["f0_bin"], float(hp["f0_min"]), float(hp["f0_max"]))
    if key in _THRESHOLDS:
        return _THRESHOLDS[key]
    as_bits = lambda v: int(np.float32(v).view(np.uint32))
    as_float = lambda b: float(np.uint32(b).view(np.float32))
    top = f0_to_coarse(torch.tensor([[float("inf")]]), hp).item()          # f0_bin - 1
    lo_b, hi_b = as_bits(0.0), as_bits(64.0)
    assert _coarse_of(as_float(lo_b), hp) == 1 and _coarse_of(as_float(hi_b), hp) == top
    thr, start = [], lo_b
    for k in range(2, top + 1):
        a, b = start, hi_b                                                 # coarse(a) < k <= coarse(b)
        while b - a > 1:
            m = (a + b) // 2
            if _coarse_of(as_float(m), hp) >= k:
                b = m
            else:
                a = m
        if not (_coarse_of(as_float(b), hp) >= k > _coarse_of(as_float(b - 1), hp)):
            raise RuntimeError("f0_to_coarse is not a clean step function around bin %d on this host" % k)
        thr.append(as_float(b))
        start = b - 1 if b > lo_b else b
    t = np.asarray(thr, dtype=np.float32)
    assert np.all(np.diff(t) >= 0)
    g = np.random.Generator(np.random.PCG64(7))
    for x in np.concatenate([g.uniform(4.0, 11.0, 192).astype(np.float32), t[::16], np.nextafter(t[::16], np.float32(-1))]):
        assert _coarse_of(float(x), hp) == 1 + int(np.searchsorted(t, x, side="right")), float(x)
    _THRESHOLDS[key] = torch.from_numpy(t)
    return _THRESHOLDS[key]


class CondBuilder(nn.Module):
    def __init__(self, hparams, out_dims=None):
        super().__init__()
        self.hp = hparams
        self.hidden_size = hparams["hidden_size"]
        self.padding_idx = 0
        self.pitch_embed = nn.Embedding(300, self.hidden_size, self.padding_idx)
        nn.init.normal_(self.pitch_embed.weight, mean=0, std=self.hidden_size ** -0.5)
        nn.init.constant_(self.pitch_embed.weight[self.padding_idx], 0)
        if hparams.get("use_energy_embed"):                       # fs2.py:81-82
            self.energy_embed = nn.Embedding(256, self.hidden_size, self.padding_idx)
            nn.init.normal_(self.energy_embed.weight, mean=0, std=self.hidden_size ** -0.5)
            nn.init.constant_(self.energy_embed.weight[self.padding_idx], 0)
        self._extras = {}          # checkpointed-but-unused fs2.* tensors (mel_out, pitch_predictor, ...)
        self._thr_dev = None       # (device, thresholds) of the device pitch path
        self._oob = None           # sticky device flag: a device-path call saw mel2ph outside [0, N] (check_alignment)

    # accept (and round-trip) the fs2.* tensors of the branches that are disabled by no_fs2 / use_pe=False
    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for k in list(state_dict.keys()):
            if k.startswith(prefix) and not k.startswith(prefix + "pitch_embed.") and not (
                    k.startswith(prefix + "energy_embed.") and hasattr(self, "energy_embed")):
                self._extras[k[len(prefix):]] = state_dict[k].detach().clone()
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)
        for k in list(unexpected_keys):
            if k.startswith(prefix) and k[len(prefix):] in self._extras:
                unexpected_keys.remove(k)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        for k, v in self._extras.items():
            destination[prefix + k] = v

    def check_alignment(self):
        """Raise IndexError if any device-path call since the last check was given a mel2ph entry outside [0, number of content units] -- what
        torch.gather raises immediately in the reference (fs2.py:100-102); here such frames silently got zero content.  Synchronises."""
        if self._oob is not None:
            bad, self._oob = bool(self._oob.item()), None
            if bad:
                raise IndexError("diffsvc_amd: mel2ph holds an index outside [0, hubert frames] (the reference's torch.gather raises here); "
                                 "those frames were built from the zero pad row")

    def _pitch_device(self, f0, mel2ph, uv):
        """f0_denorm and the coarse bin on the device (csrc/cond.hip): no host round trip, no synchronisation."""
        import ctypes
        from ._lib import check, lib, ptr, stream_ptr
        hp, dev = self.hp, f0.device
        if self._thr_dev is None or self._thr_dev[0] != dev:
            self._thr_dev = (dev, coarse_thresholds(hp).to(dev))
        thr = self._thr_dev[1]
        x = f0.detach().to(torch.float32).contiguous()
        m2p = mel2ph.to(torch.int64).contiguous()
        uv_t = uv.to(dev, torch.float32).contiguous() if (uv is not None and hp.get("use_uv")) else None
        f0_denorm = torch.empty(x.shape, device=dev, dtype=torch.float32)
        coarse = torch.empty(x.shape, device=dev, dtype=torch.int64)
        check(lib().dsvc_pitch_coarse(ptr(x), ptr(m2p), ptr(uv_t) if uv_t is not None else ctypes.c_void_p(0), ptr(thr), thr.numel(),
                                      x.numel(), ptr(f0_denorm), ptr(coarse), stream_ptr()))
        return f0_denorm, coarse

    def _build_device(self, hubert, mel2ph, f0, uv):
        """The whole builder in ONE launch (dsvc_cond_build, csrc/cond.hip): pitch bins, gather, embedding, mask, the [B, H, T] transpose
        and the reference's in-place ``f0[mel2ph == 0] = 0`` -- no torch launches, no synchronisation."""
        import ctypes
        from ._lib import check, lib, ptr, stream_ptr
        hp, dev = self.hp, hubert.device
        if self._thr_dev is None or self._thr_dev[0] != dev:
            self._thr_dev = (dev, coarse_thresholds(hp).to(dev))
        thr = self._thr_dev[1]
        B, N, H = hubert.shape
        T = mel2ph.shape[1]
        hub = hubert.detach().to(torch.float32).contiguous()
        m2p = mel2ph.to(torch.int64).contiguous()
        x = f0 if (f0.dtype == torch.float32 and f0.is_contiguous()) else f0.detach().to(torch.float32).contiguous()
        uv_t = uv.to(dev, torch.float32).contiguous() if (uv is not None and hp.get("use_uv")) else None
        emb = self.pitch_embed.weight.detach()
        emb = emb if (emb.dtype == torch.float32 and emb.is_contiguous()) else emb.to(torch.float32).contiguous()
        dec = torch.empty(B, T, H, device=dev, dtype=torch.float32)
        cond = torch.empty(B, H, T, device=dev, dtype=torch.float32)
        f0_denorm = torch.empty(B, T, device=dev, dtype=torch.float32)
        coarse = torch.empty(B, T, device=dev, dtype=torch.int64)
        check(lib().dsvc_cond_build(ptr(hub), ptr(m2p), ptr(x), ptr(uv_t) if uv_t is not None else ctypes.c_void_p(0), ptr(thr), thr.numel(),
                                    ptr(emb), B, N, T, H, ptr(dec), ptr(cond), ptr(f0_denorm), ptr(coarse), stream_ptr()))
        # the reference's torch.gather (fs2.py:100-102) raises on an alignment index outside [0, N]; the kernel reads the zero pad row instead
        # (memory-safe).  A sticky DEVICE flag keeps the difference visible without a synchronisation on this path: check_alignment() reports it.
        oob = ((m2p < 0) | (m2p > N)).any()
        self._oob = oob if self._oob is None or self._oob.device != dev else (self._oob | oob)
        if x is not f0:
            f0.copy_(x)                                                        # the reference mutates its argument (fs2.py:231)
        return dec, cond, f0_denorm, coarse

    def _pitch_host(self, f0, mel2ph, uv):
        hp = self.hp
        # ---- index work on the host, bit-exact with the reference CPU path (fs2.py:229-233, pitch_utils.py) ----
        # The reference runs every clip ALONE as a [1, T_clip] tensor (infer_tool.py:277), and torch's CPU kernels are not
        # position-independent in the last bit: a vectorised loop handles the last (numel mod 16) elements with the scalar libm
        # routine and the rest with the SIMD one.  So 2**f0 and the coarse bin are computed per clip on its own [1, T_clip]
        # slice (frames up to the last content frame), exactly the tensor the reference would have built for that clip --
        # a clip then gets the same f0_denorm / pitch bins alone, in a batch and at any batch position.
        f0_cpu = f0.detach().to("cpu", torch.float32)
        m2p_cpu = mel2ph.cpu()
        pad_cpu = m2p_cpu == 0
        B, T = m2p_cpu.shape
        uv_cpu = uv.cpu() if (uv is not None and hp.get("use_uv")) else None
        f0_denorm = torch.zeros(B, T, dtype=torch.float32)
        coarse = torch.ones(B, T, dtype=torch.long)                             # f0_to_coarse(0) == 1 on padded frames
        ar = torch.arange(1, T + 1)
        for b in range(B):
            n = int(((~pad_cpu[b]) * ar).max().item()) if (~pad_cpu[b]).any() else T   # trailing padding excluded (all-padded: whole row)
            d = 2 ** f0_cpu[b:b + 1, :n]
            if uv_cpu is not None:
                d[uv_cpu[b:b + 1, :n] > 0] = 0
            d[pad_cpu[b:b + 1, :n]] = 0
            f0_denorm[b, :n] = d[0]
            coarse[b, :n] = f0_to_coarse(d.clone(), hp)[0]
        return f0_denorm, coarse

    def forward(self, hubert, mel2ph=None, spk_embed=None, ref_mels=None, f0=None, uv=None, energy=None,
                skip_decoder=True, spk_embed_dur_id=None, spk_embed_f0_id=None, infer=False, **kwargs):
        hp = self.hp
        if hp.get("use_spk_embed") or hp.get("use_spk_id"):
            raise NotImplementedError("use_spk_id / use_spk_embed: the reference's FastSpeech2 never creates spk_embed_proj (fs2.py:32-39 is "
                                      "commented out), so its own forward fails on these configurations")
        if not hp.get("no_fs2", False):
            raise NotImplementedError("only the no_fs2 configuration of the reference is supported (the FFT encoder / decoder are out of scope)")
        ret = {"mel2ph": mel2ph}
        e_emb = None
        if hp.get("use_energy_embed"):                                        # add_energy (fs2.py:240-247)
            if energy is None:
                raise ValueError("use_energy_embed: forward() needs the frame energies (the reference indexes None here)")
            ret["energy_pred"] = energy
            e_idx = torch.clamp(energy.to(hubert.device) * 256 // 4, max=255).long()
            e_emb = self.energy_embed(e_idx)
        dev = hubert.device
        if hp.get("pitch_norm", "log") != "log":
            raise NotImplementedError("pitch_norm must be 'log'")
        if hubert.is_cuda:
            # NOT differentiable: the tensors below carry no autograd graph even with grad enabled (the drop-ins are inference modules; the
            # training branch, GaussianDiffusionHip.forward(infer=False), gets d pitch_embed from dsvc_trainer_step, not from a graph over this)
            dec, cond, f0_denorm, coarse = self._build_device(hubert, mel2ph, f0, uv)
            if e_emb is not None:                                             # (x + e) * mask = x * mask + e * mask: the mask is 0 / 1
                dec = dec + e_emb.detach() * (mel2ph > 0).to(dec.dtype)[:, :, None]
                cond = dec.transpose(1, 2).contiguous()
            ret.update(f0_denorm=f0_denorm, pitch_pred=coarse.unsqueeze(-1), decoder_inp=dec, cond_bht=cond)
            return ret
        padded = F.pad(hubert, [0, 0, 1, 0])
        idx = mel2ph[..., None].repeat([1, 1, hubert.shape[-1]])
        gathered = torch.gather(padded, 1, idx)                               # [B, T, H]
        nonpad = (mel2ph > 0).float()[:, :, None]
        if hp.get("pitch_norm", "log") != "log":
            raise NotImplementedError("pitch_norm must be 'log'")
        if hubert.is_cuda:
            f0_denorm, coarse = self._pitch_device(f0, mel2ph, uv)
        else:
            f0_denorm, coarse = self._pitch_host(f0, mel2ph, uv)
        f0[(mel2ph == 0)] = 0                                                  # the reference mutates its argument (fs2.py:231)
        ret["f0_denorm"] = f0_denorm.to(dev)
        ret["pitch_pred"] = coarse.unsqueeze(-1).to(dev)
        emb = self.pitch_embed(coarse.to(dev))
        ret["decoder_inp"] = ((gathered + emb) + e_emb) * nonpad if e_emb is not None else (gathered + emb) * nonpad
        return ret
