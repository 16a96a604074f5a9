"""CPU oracle for the diff-svc inference hot path  (TEST INFRASTRUCTURE -- NOT PRODUCT CODE).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this file.  The product path (``diff-svc_amd/``) never does: it fails loudly when the HIP library
is missing instead of falling back to anything in here.

What this is: a plain PyTorch-CPU / numpy fp32 restatement of the algorithm the reference
(prophesier/diff-svc, mounted read-only at /root/reference in the build container) runs on the
path  cond -> GaussianDiffusion sampler (DDPM / PLMS) -> DiffNet -> NSF-HiFiGAN (+ STFT/mel).
Every function cites the reference file:line it restates.  The arithmetic itself lives in the
third-party dependency ``torch==1.12.1+cu113`` (requirements.txt:90); here it runs on this
image's torch CPU kernels.

Pinning status: the reference ships NO tests, golden vectors or fixtures,
so the oracle is pinned against outputs of the reference ITSELF, produced in the build container
by ``oracle/make_golden.py`` (imports /root/reference behind ``oracle/refshim.py``) and committed
under ``tests/golden/``.  ``tests/test_oracle_golden.py`` checks every function here against
those vectors.  One boundary stays unpinned: the mel filterbank (``librosa.filters.mel``,
librosa==0.9.1 is not installable here); it is restated from the published Slaney algorithm and
its hash is committed.

The random numbers are NOT the reference's: torch.randn (diffusion.py:34-37,160; models.py:192,
271) cannot be reproduced on a GPU, so both this oracle and the HIP kernels use the same
counter-based Philox4x32-10 + Box-Muller generator defined below, and the goldens were minted by
injecting exactly that noise into the reference.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------------
# Counter-based RNG shared with the HIP kernels (diff-svc_amd/csrc/philox.h)
# ----------------------------------------------------------------------------------------------
PURPOSE_DDPM_NOISE = 1   # z in p_sample, counter = (quad, step, clip, 1)
PURPOSE_X_INIT = 2       # x_T,           counter = (quad, 0,    clip, 2)
PURPOSE_SINE_NOISE = 3   # SineGen noise, counter = (sample, j,  clip, 3), j in {0,1,2}
PURPOSE_SINE_PHASE = 4   # SineGen initial phases, counter = (j, 0, clip, 4)

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32(c0, c1, c2, c3, seed):
    """Philox4x32-10 (Salmon et al., SC'11).  Counters are uint32 arrays (broadcastable), the key
    is the 64-bit seed split in two words.  Returns four uint32 arrays."""
    c0, c1, c2, c3 = np.broadcast_arrays(*[np.asarray(c, dtype=np.uint64) & _MASK for c in (c0, c1, c2, c3)])
    c0, c1, c2, c3 = c0.copy(), c1.copy(), c2.copy(), c3.copy()
    k0 = int(seed) & 0xFFFFFFFF
    k1 = (int(seed) >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0, lo1, n2, lo0
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32))


def _u01_open_low(r):
    """(0,1] from the top 24 bits -- exact in fp32."""
    return ((r >> np.uint32(8)).astype(np.float64) + 1.0) * (1.0 / 16777216.0)


def _u01_open_high(r):
    """[0,1) from the top 24 bits -- exact in fp32."""
    return (r >> np.uint32(8)).astype(np.float64) * (1.0 / 16777216.0)


def philox_normal4(c0, c1, c2, c3, seed):
    """Four N(0,1) values per counter (two Box-Muller pairs), returned as float32 [..., 4]."""
    r0, r1, r2, r3 = philox4x32(c0, c1, c2, c3, seed)
    ra = np.sqrt(-2.0 * np.log(_u01_open_low(r0)))
    rb = np.sqrt(-2.0 * np.log(_u01_open_low(r2)))
    ta = 2.0 * np.pi * _u01_open_high(r1)
    tb = 2.0 * np.pi * _u01_open_high(r3)
    out = np.stack([ra * np.cos(ta), ra * np.sin(ta), rb * np.cos(tb), rb * np.sin(tb)], axis=-1)
    return out.astype(np.float32)


def philox_uniform4(c0, c1, c2, c3, seed):
    """Four U[0,1) values per counter as float32 [..., 4]."""
    rs = philox4x32(c0, c1, c2, c3, seed)
    return np.stack([_u01_open_high(r) for r in rs], axis=-1).astype(np.float32)


def frame_major_noise(seed, clip, step, T, M, purpose=PURPOSE_DDPM_NOISE):
    """Gaussian noise for one clip and one sampler step in the FRAME-MAJOR order the HIP kernels
    use: element (t, m) is lane (t*M+m)&3 of counter quad (t*M+m)>>2.  Returns float32 [T, M]."""
    assert (T * M) % 4 == 0
    quad = np.arange(T * M // 4, dtype=np.uint64)
    z = philox_normal4(quad, np.uint64(step), np.uint64(clip), np.uint64(purpose), seed)
    return z.reshape(T, M)


def ddpm_noise_ref_layout(seed, clips, step, T, M, purpose=PURPOSE_DDPM_NOISE):
    """Same noise in the reference's [B,1,M,T] layout (network/diff/diffusion.py:160,265-268)."""
    z = np.stack([frame_major_noise(seed, c, step, T, M, purpose).T for c in clips], 0)
    return torch.from_numpy(np.ascontiguousarray(z[:, None]))


# ----------------------------------------------------------------------------------------------
# Schedules  (network/diff/diffusion.py:40-58, 87-120)
# ----------------------------------------------------------------------------------------------
SCHEDULE_KEYS = (
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
    "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
    "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
    "posterior_mean_coef1", "posterior_mean_coef2",
)


def linear_betas(timesteps, max_beta):
    """diffusion.py:40-45.  NB the reference binds max_beta at import time."""
    return np.linspace(1e-4, max_beta, timesteps)


def cosine_betas(timesteps, s=0.008):
    """diffusion.py:48-58."""
    steps = timesteps + 1
    x = np.linspace(0, steps, steps)
    ac = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    ac = ac / ac[0]
    return np.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)


def schedule_tables(betas):
    """The 12 registered buffers: float64 numpy math, then cast to fp32 (diffusion.py:87-120)."""
    betas = np.asarray(betas, dtype=np.float64)
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    pv = betas * (1.0 - ac_prev) / (1.0 - ac)
    t = {
        "betas": betas,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "sqrt_alphas_cumprod": np.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": np.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": np.log(1.0 - ac),
        "sqrt_recip_alphas_cumprod": np.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": np.sqrt(1.0 / ac - 1),
        "posterior_variance": pv,
        "posterior_log_variance_clipped": np.log(np.maximum(pv, 1e-20)),
        "posterior_mean_coef1": betas * np.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac),
    }
    return {k: torch.tensor(v, dtype=torch.float32) for k, v in t.items()}


# ----------------------------------------------------------------------------------------------
# DiffNet denoiser  (network/diff/net.py)
# ----------------------------------------------------------------------------------------------
def mish(x):
    """modules/commons/common_layers.py:485-487."""
    return x * torch.tanh(F.softplus(x))


def step_embedding(sd, t, prefix="denoise_fn."):
    """SinusoidalPosEmb + 2-layer Mish MLP (net.py:32-44, 99-103, 124-125).  t: int64 [B]."""
    C = sd[prefix + "mlp.2.weight"].shape[0]
    half = C // 2
    scale = math.log(10000) / (half - 1)
    freqs = torch.exp(torch.arange(half) * -scale)
    ang = t[:, None] * freqs[None, :]
    emb = torch.cat((ang.sin(), ang.cos()), dim=-1).to(sd[prefix + "mlp.0.weight"].dtype)      # (a float64 state dict evaluates the net in fp64)
    h = F.linear(emb, sd[prefix + "mlp.0.weight"], sd[prefix + "mlp.0.bias"])
    return F.linear(mish(h), sd[prefix + "mlp.2.weight"], sd[prefix + "mlp.2.bias"])


def diffnet_layers(sd, prefix="denoise_fn."):
    n = 0
    while (prefix + "residual_layers.%d.dilated_conv.weight" % n) in sd:
        n += 1
    return n


def diffnet_forward(sd, spec, t, cond, dilation_cycle, prefix="denoise_fn.", taps=None):
    """DiffNet.forward (net.py:112-135) with ResidualBlock.forward (net.py:66-84) inlined.
    spec [B,1,M,T] f32, t [B] i64, cond [B,H,T] f32 -> [B,1,M,T].  ``taps`` (optional dict)
    receives per-layer activations for debugging the HIP kernels."""
    p = lambda k: sd[prefix + k]
    L = diffnet_layers(sd, prefix)
    C = p("input_projection.weight").shape[0]
    x = F.relu(F.conv1d(spec[:, 0], p("input_projection.weight"), p("input_projection.bias")))
    emb = step_embedding(sd, t, prefix)
    skip_sum = torch.zeros_like(x)
    if taps is not None:
        taps["x_in"] = x.clone()
        taps["emb"] = emb.clone()
    for l in range(L):
        q = lambda k: p("residual_layers.%d.%s" % (l, k))
        d = 2 ** (l % dilation_cycle)
        film = F.linear(emb, q("diffusion_projection.weight"), q("diffusion_projection.bias"))[:, :, None]
        c = F.conv1d(cond, q("conditioner_projection.weight"), q("conditioner_projection.bias"))
        y = F.conv1d(x + film, q("dilated_conv.weight"), q("dilated_conv.bias"), padding=d, dilation=d) + c
        z = torch.sigmoid(y[:, :C]) * torch.tanh(y[:, C:])          # first half = gate, second = filter
        o = F.conv1d(z, q("output_projection.weight"), q("output_projection.bias"))
        x = (x + o[:, :C]) / math.sqrt(2.0)
        skip_sum = skip_sum + o[:, C:]
        if taps is not None:
            taps["g%d" % l] = z.clone()
            taps["x%d" % l] = x.clone()
            taps["s%d" % l] = skip_sum.clone()          # running (unscaled) skip sum after layer l
    s = skip_sum / math.sqrt(L)
    if taps is not None:
        taps["skip"] = s.clone()
    s = F.relu(F.conv1d(s, p("skip_projection.weight"), p("skip_projection.bias")))
    out = F.conv1d(s, p("output_projection.weight"), p("output_projection.bias"))
    return out[:, None]


# ----------------------------------------------------------------------------------------------
# Sampler  (network/diff/diffusion.py)
# ----------------------------------------------------------------------------------------------
def _at(table, t):
    """``extract`` (diffusion.py:28-31) for 4-D x."""
    return table.gather(-1, t).reshape(-1, 1, 1, 1)


def ddpm_update(sd, x, eps, t, z):
    """p_sample after the denoiser call (diffusion.py:131-163): predict x0, clamp, posterior mean,
    add sigma*z unless t == 0."""
    x0 = _at(sd["sqrt_recip_alphas_cumprod"], t) * x - _at(sd["sqrt_recipm1_alphas_cumprod"], t) * eps
    x0 = x0.clamp(-1.0, 1.0)
    mean = _at(sd["posterior_mean_coef1"], t) * x0 + _at(sd["posterior_mean_coef2"], t) * x
    logvar = _at(sd["posterior_log_variance_clipped"], t)
    nonzero = (1 - (t == 0).float()).reshape(-1, 1, 1, 1)
    return mean + nonzero * (0.5 * logvar).exp() * z


def plms_x_pred(sd, x, eps, t, interval):
    """get_x_pred inside p_sample_plms (diffusion.py:171-179)."""
    a_t = _at(sd["alphas_cumprod"], t)
    a_p = _at(sd["alphas_cumprod"], torch.clamp(t - interval, min=0))
    a_t_sq, a_p_sq = a_t.sqrt(), a_p.sqrt()
    delta = (a_p - a_t) * ((1 / (a_t_sq * (a_t_sq + a_p_sq))) * x
                           - 1 / (a_t_sq * (((1 - a_p) * a_t).sqrt() + ((1 - a_t) * a_p).sqrt())) * eps)
    return x + delta


def plms_combine(eps, hist):
    """Adams-Bashforth combination of the stored predictions (diffusion.py:188-193).
    ``hist`` is newest-last, at most 3 entries are used."""
    n = len(hist)
    if n == 1:
        return (3 * eps - hist[-1]) / 2
    if n == 2:
        return (23 * eps - 16 * hist[-1] + 5 * hist[-2]) / 12
    return (55 * eps - 59 * hist[-1] + 37 * hist[-2] - 9 * hist[-3]) / 24


def sample_ddpm(sd, cond, x, noise_fn, dilation_cycle, t_start=None, t_end=0):
    """HOT LOOP B (diffusion.py:276-278): for i in reversed(range(t_start)).  ``noise_fn(i)`` returns
    z [B,1,M,T] for step i.  ``t_end`` lets tests run a window of the chain."""
    B = x.shape[0]
    K = int(sd["betas"].shape[0]) if t_start is None else t_start
    with torch.no_grad():
        for i in reversed(range(t_end, K)):
            t = torch.full((B,), i, dtype=torch.long)
            eps = diffnet_forward(sd, x, t, cond, dilation_cycle)
            x = ddpm_update(sd, x, eps, t, noise_fn(i))
    return x


def sample_plms(sd, cond, x, interval, dilation_cycle, t_start=None):
    """HOT LOOP A (diffusion.py:269-275) with p_sample_plms (diffusion.py:165-198).  B must be 1 in
    the reference (``max(t-interval, 0)`` on a tensor, diffusion.py:186); every clip of a batch
    shares t here, so the batched form is the per-clip loop."""
    B = x.shape[0]
    K = int(sd["betas"].shape[0]) if t_start is None else t_start
    hist = []
    with torch.no_grad():
        for i in reversed(range(0, K, interval)):
            t = torch.full((B,), i, dtype=torch.long)
            eps = diffnet_forward(sd, x, t, cond, dilation_cycle)
            if len(hist) == 0:
                x_pred = plms_x_pred(sd, x, eps, t, interval)
                t_prev = torch.clamp(t - interval, min=0)
                eps_prev = diffnet_forward(sd, x_pred, t_prev, cond, dilation_cycle)
                eps_prime = (eps + eps_prev) / 2
            else:
                eps_prime = plms_combine(eps, hist)
            x = plms_x_pred(sd, x, eps_prime, t, interval)
            hist.append(eps)
            hist = hist[-4:]
    return x


def q_sample(sd, x0, t, noise):
    """diffusion.py:200-205."""
    return _at(sd["sqrt_alphas_cumprod"], t) * x0 + _at(sd["sqrt_one_minus_alphas_cumprod"], t) * noise


PURPOSE_TRAIN_NOISE = 5


def p_losses(sd, x_start, t, cond, noise, dilation_cycle, loss_type="l2"):
    """diffusion.py:207-225 (the nonpadding-weighted l1 variant is commented out in the reference's call site,
    train_pipeline.py:236-238).  x_start [B,1,M,T] normalised, t [B] long, cond [B,H,T], noise like x_start."""
    x_noisy = q_sample(sd, x_start, t, noise)
    x_recon = diffnet_forward(sd, x_noisy, t, cond, dilation_cycle)
    if loss_type == "l1":
        return (noise - x_recon).abs().mean()
    return F.mse_loss(noise, x_recon)


def train_loss_and_grads(sd, hubert, mel2ph, f0, mels, t, noise, hp):
    """GaussianDiffusion.forward(infer=False) (diffusion.py:227-241 -> train_pipeline.py:222-238) with torch autograd: the loss and
    the gradient of every ``denoise_fn.*`` parameter and of ``fs2.pitch_embed.weight``.  mels [B,T,M] (log-mel targets)."""
    names = [k for k in sd if k.startswith("denoise_fn.")] + ["fs2.pitch_embed.weight"]
    p = dict(sd)
    for k in names:
        p[k] = sd[k].detach().clone().requires_grad_(True)
    cond, _, _ = build_cond(p, hubert, mel2ph, f0.clone(), hp)
    x0 = norm_spec(sd, mels).transpose(1, 2)[:, None, :, :]
    loss = p_losses(p, x0, t, cond.transpose(1, 2), noise, hp["dilation_cycle_length"], hp.get("diff_loss_type", "l2"))
    grads = torch.autograd.grad(loss, [p[k] for k in names], allow_unused=True)
    return loss.detach(), {k: (g if g is not None else torch.zeros_like(sd[k])) for k, g in zip(names, grads)}


def norm_spec(sd, mel):
    """diffusion.py:286-287.  mel [B,T,M]."""
    return (mel - sd["spec_min"]) / (sd["spec_max"] - sd["spec_min"]) * 2 - 1


def denorm_spec(sd, x):
    """diffusion.py:289-290."""
    return (x + 1) / 2 * (sd["spec_max"] - sd["spec_min"]) + sd["spec_min"]


def finish_mel(sd, x, mel2ph):
    """Tail of GaussianDiffusion.forward(infer=True) (diffusion.py:279-283): [B,1,M,T] -> [B,T,M]."""
    mel = denorm_spec(sd, x[:, 0].transpose(1, 2))
    if mel2ph is not None:
        mel = mel * (mel2ph > 0).float()[:, :, None]
    return mel


# ----------------------------------------------------------------------------------------------
# Condition builder  (modules/fastspeech/fs2.py:94-154,185-238; utils/pitch_utils.py:17-31,63-76)
# ----------------------------------------------------------------------------------------------
def f0_to_coarse(f0, hp):
    """utils/pitch_utils.py:17-31, torch branch: mel-scale quantisation to [1, f0_bin-1]."""
    f0_bin, f0_max, f0_min = hp["f0_bin"], hp["f0_max"], hp["f0_min"]
    mel_min = 1127 * np.log(1 + f0_min / 700)
    mel_max = 1127 * np.log(1 + f0_max / 700)
    m = 1127 * (1 + f0 / 700).log()
    pos = m > 0
    m = torch.where(pos, (m - mel_min) * (f0_bin - 2) / (mel_max - mel_min) + 1, m)
    m = torch.where(m <= 1, torch.ones_like(m), m)
    m = torch.where(m > f0_bin - 1, torch.full_like(m, f0_bin - 1), m)
    return (m + 0.5).long()


def build_cond(sd, hubert, mel2ph, f0, hp, energy=None):
    """FastSpeech2.forward, ``no_fs2: true`` branch (fs2.py:94-154) + add_pitch (fs2.py:185-238):
    cond = (gather(pad(hubert), mel2ph) + pitch_embed[coarse(2**f0)]) * (mel2ph > 0);
    with ``use_energy_embed`` (fs2.py:143-144, add_energy :240-247) + energy_embed[clamp(energy * 256 // 4, max=255)] inside the mask.
    Returns (decoder_inp [B,T,H], f0_denorm [B,T], coarse [B,T])."""
    padded = F.pad(hubert, [0, 0, 1, 0])
    idx = mel2ph[..., None].repeat([1, 1, hubert.shape[-1]])
    gathered = torch.gather(padded, 1, idx)
    nonpad = (mel2ph > 0).float()[:, :, None]
    f0_denorm = 2 ** f0                                   # pitch_norm == 'log' (pitch_utils.py:66-67)
    f0_denorm = torch.where(mel2ph == 0, torch.zeros_like(f0_denorm), f0_denorm)
    coarse = f0_to_coarse(f0_denorm, hp)
    emb = F.embedding(coarse, sd["fs2.pitch_embed.weight"])
    dec = gathered + emb
    if hp.get("use_energy_embed"):
        dec = dec + F.embedding(torch.clamp(energy * 256 // 4, max=255).long(), sd["fs2.energy_embed.weight"])
    return dec * nonpad, f0_denorm, coarse


def get_align(n_mel, n_units):
    """infer_tools/infer_tool.py:231-242 -- uniform stretch of the unit frames over the mel frames."""
    mel2ph = np.zeros([n_mel], int)
    start = 0
    dur = n_mel / n_units
    for i in range(n_units):
        end = int(i * dur + dur + 0.5)
        mel2ph[start:end + 1] = i + 1
        start = end + 1
    return mel2ph


# ----------------------------------------------------------------------------------------------
# NSF-HiFiGAN generator  (modules/nsf_hifigan/models.py)
# ----------------------------------------------------------------------------------------------
LRELU = 0.1


def fold_weight_norm(state):
    """remove_weight_norm (models.py:28,389-396): w = g * v / ||v||, norm over all dims but 0."""
    out = {}
    for k, v in state.items():
        if k.endswith(".weight_g"):
            base = k[:-len(".weight_g")]
            wv = state[base + ".weight_v"]
            norm = wv.reshape(wv.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (wv.dim() - 1)))
            out[base + ".weight"] = v * wv / norm
        elif k.endswith(".weight_v"):
            continue
        else:
            out[k] = v
    return out


def sine_source(f0_up, sr, rand_ini, noise, lin_w, lin_b, harmonics=8,
                sine_amp=0.1, noise_std=0.003, voiced_threshold=0.0):
    """SineGen.forward/_f02sine + SourceModuleHnNSF.forward (models.py:183-276, 310-323).
    f0_up [B,N] (already nearest-upsampled), rand_ini [B,dim] with column 0 == 0, noise [B,N,dim]
    ~ N(0,1).  Returns har_source [B,1,N].  torch CPU cumsum accumulates fp32 inputs in double
    (ATen acc_type<float,false>) and that is what the reference CPU path does."""
    dim = harmonics + 1
    mult = torch.arange(1, dim + 1, dtype=torch.float32)
    fn = f0_up[:, :, None] * mult[None, None, :]
    rad = (fn / sr) % 1
    rad[:, 0, :] = rad[:, 0, :] + rand_ini
    tmp = torch.cumsum(rad, 1) % 1
    over = (tmp[:, 1:, :] - tmp[:, :-1, :]) < 0
    shift = torch.zeros_like(rad)
    shift[:, 1:, :] = over * -1.0
    sines = torch.sin(torch.cumsum(rad + shift, dim=1) * 2 * np.pi) * sine_amp
    uv = (f0_up > voiced_threshold).float()[:, :, None]
    noise_amp = uv * noise_std + (1 - uv) * sine_amp / 3
    sines = sines * uv + noise_amp * noise
    merged = torch.tanh(F.linear(sines, lin_w, lin_b))
    return merged.transpose(1, 2)


def generator_forward(gw, h, mel, f0, rand_ini, noise, taps=None):
    """Generator.forward (models.py:361-387) on folded weights ``gw`` (see fold_weight_norm); also the 24 kHz
    HifiGanGenerator.forward (modules/hifigan/hifigan.py:146-170: the same network, ``f0`` optional).
    mel [B,M,T] natural-log, f0 [B,T] Hz or None -> wav [B,1,T*prod(rates)]."""
    rates = list(h["upsample_rates"])
    ksz = list(h["upsample_kernel_sizes"])
    rks = list(h["resblock_kernel_sizes"])
    rds = [list(d) for d in h["resblock_dilation_sizes"]]
    hop = int(np.prod(rates))
    har = None                               # f0 None: the plain HiFi-GAN path of HifiGanGenerator.forward (hifigan.py:150-162)
    if f0 is not None:
        f0_up = f0[:, :, None].repeat(1, 1, hop).reshape(f0.shape[0], -1)      # nn.Upsample nearest (models.py:331,363)
        har = sine_source(f0_up, h["sampling_rate"] if "sampling_rate" in h else h["audio_sample_rate"], rand_ini, noise,
                          gw["m_source.l_linear.weight"], gw["m_source.l_linear.bias"])
        if taps is not None:
            taps["har"] = har.clone()
    x = F.conv1d(mel, gw["conv_pre.weight"], gw["conv_pre.bias"], padding=3)
    nk = len(rks)
    for i, (u, k) in enumerate(zip(rates, ksz)):
        x = F.leaky_relu(x, LRELU)
        x = F.conv_transpose1d(x, gw["ups.%d.weight" % i], gw["ups.%d.bias" % i], stride=u, padding=(k - u) // 2)
        if har is not None:
            if i + 1 < len(rates):
                s = int(np.prod(rates[i + 1:]))
                xs = F.conv1d(har, gw["noise_convs.%d.weight" % i], gw["noise_convs.%d.bias" % i], stride=s, padding=s // 2)
            else:
                xs = F.conv1d(har, gw["noise_convs.%d.weight" % i], gw["noise_convs.%d.bias" % i])
            x = x + xs
        if taps is not None:
            taps["up%d" % i] = x.clone()
        acc = None
        for j in range(nk):
            rb = _resblock1 if str(h.get("resblock", "1")) == "1" else _resblock2      # models.py:337
            r = rb(gw, "resblocks.%d." % (i * nk + j), x, rks[j], rds[j])
            acc = r if acc is None else acc + r
        x = acc / nk
        if taps is not None:
            taps["mrf%d" % i] = x.clone()
    x = F.leaky_relu(x)                      # default slope 0.01 (models.py:383)
    x = F.conv1d(x, gw["conv_post.weight"], gw["conv_post.bias"], padding=3)
    return torch.tanh(x)


def _resblock2(gw, prefix, x, k, dils):
    """ResBlock2.forward (models.py:86-91): one dilated conv per residual step -- exactly two of them, built from dilation[0] and dilation[1]
    whatever the length of the config's list (models.py:77-82)."""
    for j, d in enumerate(dils[:2]):
        xt = F.leaky_relu(x, LRELU)
        xt = F.conv1d(xt, gw[prefix + "convs.%d.weight" % j], gw[prefix + "convs.%d.bias" % j], padding=(k * d - d) // 2, dilation=d)
        x = xt + x
    return x


def _resblock1(gw, prefix, x, k, dils):
    """ResBlock1.forward (models.py:57-64)."""
    for j, d in enumerate(dils[:3]):                       # (three pairs from dilation[0..2], models.py:36-55)
        xt = F.leaky_relu(x, LRELU)
        xt = F.conv1d(xt, gw[prefix + "convs1.%d.weight" % j], gw[prefix + "convs1.%d.bias" % j],
                      padding=(k * d - d) // 2, dilation=d)
        xt = F.leaky_relu(xt, LRELU)
        xt = F.conv1d(xt, gw[prefix + "convs2.%d.weight" % j], gw[prefix + "convs2.%d.bias" % j],
                      padding=(k - 1) // 2)
        x = xt + x
    return x


def vocoder_rng(seed, clips, n_samples, dim=9):
    """Philox-defined random inputs of the source module, shared with the HIP kernel:
    initial phases U[0,1) for harmonics 1.. (column 0 forced to 0, models.py:192-194) and the
    additive noise N(0,1) [B,N,dim] (models.py:271)."""
    ini, nz = [], []
    for c in clips:
        j = np.arange(3, dtype=np.uint64)
        u = philox_uniform4(j, np.uint64(0), np.uint64(c), np.uint64(PURPOSE_SINE_PHASE), seed).reshape(-1)[:dim].copy()
        u[0] = 0.0
        ini.append(u)
        s = np.arange(n_samples, dtype=np.uint64)[:, None]
        z = philox_normal4(s, j[None, :], np.uint64(c), np.uint64(PURPOSE_SINE_NOISE), seed)   # [N,3,4]
        nz.append(z.reshape(n_samples, 12)[:, :dim])
    return torch.from_numpy(np.stack(ini)), torch.from_numpy(np.stack(nz))


# ----------------------------------------------------------------------------------------------
# STFT -> mel front-end  (modules/nsf_hifigan/nvSTFT.py:72-104, network/vocoders/nsf_hifigan.py:75-92)
# ----------------------------------------------------------------------------------------------
def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """Restatement of ``librosa.filters.mel`` (librosa==0.9.1, requirements.txt; call site
    nvSTFT.py:88) with its defaults: Slaney mel scale (htk=False), norm='slaney', float32 output.
    PARITY UNPINNED at this boundary: librosa is not installable in the build image."""
    n_freqs = 1 + n_fft // 2
    fftfreqs = np.linspace(0, float(sr) / 2, n_freqs)
    mel_pts = np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2)
    mel_f = _mel_to_hz(mel_pts)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    w = np.zeros((n_mels, n_freqs), dtype=np.float32)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    w *= enorm[:, np.newaxis]            # float32 *= float64: numpy multiplies in double, rounds once (as librosa does)
    return w


def mel_spectrogram(wav, sr, n_fft, win_size, hop, n_mels, fmin, fmax, clip_val=1e-5, basis=None):
    """STFT.get_mel (nvSTFT.py:72-104) followed by the log -> log10 scale of wav2spec
    (nsf_hifigan.py:86-91).  wav [B,N] -> mel [B,T,n_mels] (log10).  torch>=2 needs
    return_complex=True; |.| is computed from the real view exactly as nvSTFT.py:98 does."""
    if basis is None:
        basis = torch.from_numpy(mel_filterbank(sr, n_fft, n_mels, fmin, fmax))
    pad = int((n_fft - hop) / 2)
    y = F.pad(wav.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    spec = torch.stft(y, n_fft, hop_length=hop, win_length=win_size, window=torch.hann_window(win_size),
                      center=False, pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    spec = torch.view_as_real(spec)
    mag = torch.sqrt(spec.pow(2).sum(-1) + 1e-9)
    mel = torch.matmul(basis, mag)
    mel = torch.log(torch.clamp(mel, min=clip_val))
    return (0.434294 * mel).transpose(1, 2)


def process_utterance_mel(wav, sr, n_fft, win_size, hop, n_mels, fmin, fmax, eps=1e-10, basis=None):
    """The mel of process_utterance (preprocessing/data_gen_utils.py:124-136), the 24 kHz front-end behind PWG.wav2spec / HifiGAN:
    librosa.stft(n_fft, hop, win_length, 'hann', pad_mode='constant') = a centred STFT over a zero-padded signal with a periodic hann
    window, |X|, librosa mel filterbank, log10(max(eps, .)).  wav [B,N] -> mel [B, 1 + N // hop, n_mels].
    PARITY UNPINNED at librosa (not installable here, like the filterbank): restated from librosa 0.9.1's published definition."""
    if basis is None:
        basis = torch.from_numpy(mel_filterbank(sr, n_fft, n_mels, fmin, fmax))
    spec = torch.stft(wav, n_fft, hop_length=hop, win_length=win_size, window=torch.hann_window(win_size), center=True,
                      pad_mode="constant", normalized=False, onesided=True, return_complex=True)
    mel = torch.matmul(basis, spec.abs())
    return torch.log10(torch.clamp(mel, min=eps)).transpose(1, 2)


def process_utterance_linear(wav, n_fft, win_size, hop, min_level_db):
    """The third return value of process_utterance(return_linear=True) (preprocessing/data_gen_utils.py:144-149): audio.normalize(audio.amp_to_db(
    |X|), {'min_level_db': ...}) = (20 log10(max(1e-5, |X|)) - min_level_db) / -min_level_db (utils/audio.py:51-56) of the same centred, zero-padded
    STFT; the wrappers transpose it to [T, n_bins] (network/vocoders/pwg.py:119-120).  wav [B,N] -> [B, 1 + N // hop, n_fft / 2 + 1]."""
    spec = torch.stft(wav, n_fft, hop_length=hop, win_length=win_size, window=torch.hann_window(win_size), center=True,
                      pad_mode="constant", normalized=False, onesided=True, return_complex=True).abs()
    db = 20.0 * torch.log10(torch.clamp(spec, min=1e-5))
    return ((db - min_level_db) / -min_level_db).transpose(1, 2)


# ----------------------------------------------------------------------------------------------
# Content encoder  (network/hubert/hubert_model.py)
# ----------------------------------------------------------------------------------------------
def hubert_units(sd, wav):
    """HubertSoft.units (hubert_model.py:67-77) in functional form: wav [1,1,N] at 16 kHz -> units [1,T,256].
    FeatureExtractor :82-102, FeatureProjection :105-116, PositionalConvEmbedding :119-137 (weight_norm over dim 2),
    nn.TransformerEncoderLayer(768, 12, 3072, gelu, batch_first) x 12 (post-LN), proj."""
    x = F.pad(wav, (40, 40))
    x = F.conv1d(x, sd["feature_extractor.conv0.weight"], stride=5)
    x = F.gelu(F.group_norm(x, 512, sd["feature_extractor.norm0.weight"], sd["feature_extractor.norm0.bias"]))
    for i in range(1, 7):
        x = F.gelu(F.conv1d(x, sd["feature_extractor.conv%d.weight" % i], stride=2))
    x = x.transpose(1, 2)
    x = F.layer_norm(x, (512,), sd["feature_projection.norm.weight"], sd["feature_projection.norm.bias"])
    x = F.linear(x, sd["feature_projection.projection.weight"], sd["feature_projection.projection.bias"])
    g, v = sd["positional_embedding.conv.weight_g"], sd["positional_embedding.conv.weight_v"]
    w = g * v / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()                       # torch._weight_norm(v, g, dim=2)
    p = F.conv1d(x.transpose(1, 2), w, sd["positional_embedding.conv.bias"], padding=64, groups=16)
    x = x + F.gelu(p[:, :, :-1]).transpose(1, 2)
    x = F.layer_norm(x, (768,), sd["norm.weight"], sd["norm.bias"])
    B, T, _ = x.shape
    for l in range(12):
        q = "encoder.layers.%d." % l
        qkv = F.linear(x, sd[q + "self_attn.in_proj_weight"], sd[q + "self_attn.in_proj_bias"])
        qh, kh, vh = (t.reshape(B, T, 12, 64).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
        att = torch.softmax(qh @ kh.transpose(-1, -2) / 8.0, dim=-1) @ vh
        att = att.transpose(1, 2).reshape(B, T, 768)
        x = F.layer_norm(x + F.linear(att, sd[q + "self_attn.out_proj.weight"], sd[q + "self_attn.out_proj.bias"]), (768,),
                         sd[q + "norm1.weight"], sd[q + "norm1.bias"])
        ff = F.linear(F.gelu(F.linear(x, sd[q + "linear1.weight"], sd[q + "linear1.bias"])), sd[q + "linear2.weight"], sd[q + "linear2.bias"])
        x = F.layer_norm(x + ff, (768,), sd[q + "norm2.weight"], sd[q + "norm2.bias"])
    return F.linear(x, sd["proj.weight"], sd["proj.bias"])


# ----------------------------------------------------------------------------------------------
# Host glue between the sampler and the vocoder  (infer_tools/infer_tool.py:171-200)
# ----------------------------------------------------------------------------------------------
def after_infer_mel(mel_pred, f0_pred, hp):
    """Drop all-zero (padded) frames, clip to [mel_vmin, mel_vmax] (infer_tool.py:177-191).
    numpy in, numpy out; mel [T,M], f0 [T]."""
    mask = np.abs(mel_pred).sum(-1) > 0
    mel = np.clip(mel_pred[mask], hp["mel_vmin"], hp["mel_vmax"])
    f0 = f0_pred[:len(mask)][mask]
    return mel, f0


def spec2wav(gw, h, mel_log10, f0, rand_ini, noise):
    """NsfHifiGAN.spec2wav (network/vocoders/nsf_hifigan.py:47-73): log10 -> ln, run the generator."""
    c = 2.30259 * torch.as_tensor(mel_log10, dtype=torch.float32).unsqueeze(0).transpose(2, 1)
    f = torch.as_tensor(f0, dtype=torch.float32)[None, :]
    with torch.no_grad():
        return generator_forward(gw, h, c, f, rand_ini, noise).view(-1)


# ----------------------------------------------------------------------------------------------
# Pitch extractor  (modules/fastspeech/pe.py)
# ----------------------------------------------------------------------------------------------
def sinusoid_table(n_rows, dim):
    """SinusoidalPositionalEmbedding.get_embedding(n_rows, dim, padding_idx=0)  (modules/commons/common_layers.py:105-122)."""
    half = dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float) * -(math.log(10000) / (half - 1)))
    ang = torch.arange(n_rows, dtype=torch.float).unsqueeze(1) * freq.unsqueeze(0)
    tab = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1).view(n_rows, -1)
    if dim % 2 == 1:
        tab = torch.cat([tab, torch.zeros(n_rows, 1)], dim=1)
    tab[0, :] = 0
    return tab


def pitch_extractor(sd, mel, hp, conv_layers=2):
    """PitchExtractor.forward in eval mode (pe.py:136-148), functional: mel [B,T,M] -> (pitch_pred [B,T,2], f0_denorm_pred [B,T]).
    Prenet :23-42 (Conv1d k5 -> ReLU -> BatchNorm1d running stats, masked), ConvStacks :98-117 (residual GroupNorm(C/16) blocks),
    PitchPredictor tts_modules.py:222-235 (positions from make_positions(xs[..., 0], 0), LayerNorm over channels eps 1e-12),
    denorm_f0 pitch_utils.py:63-76."""
    pad = mel.abs().sum(-1).eq(0)
    keep = 1 - pad.float()[:, None, :]
    x = mel.transpose(1, 2)
    for l in range(3):
        q = "mel_prenet.layers.%d." % l
        k = sd[q + "0.weight"].shape[-1]
        x = F.relu(F.conv1d(x, sd[q + "0.weight"], sd[q + "0.bias"], padding=k // 2))
        x = F.batch_norm(x, sd[q + "2.running_mean"], sd[q + "2.running_var"], sd[q + "2.weight"], sd[q + "2.bias"], False, 0.1, 1e-5)
        x = x * keep
    x = F.linear(x.transpose(1, 2), sd["mel_prenet.out_proj.weight"], sd["mel_prenet.out_proj.bias"]) * keep.transpose(1, 2)
    if conv_layers > 0:
        x = F.linear(x, sd["mel_encoder.in_proj.weight"], sd["mel_encoder.in_proj.bias"]).transpose(1, 2)
        for l in range(conv_layers):
            q = "mel_encoder.conv.%d." % l
            w = sd[q + "conv.conv.weight"]
            y = F.conv1d(x, w, sd[q + "conv.conv.bias"], padding=w.shape[-1] // 2)
            y = F.group_norm(y, w.shape[0] // 16, sd[q + "norm.weight"], sd[q + "norm.bias"], 1e-5)
            x = x + F.relu(y)
        x = F.linear(x.transpose(1, 2), sd["mel_encoder.out_proj.weight"], sd["mel_encoder.out_proj.bias"])
    B, T, H = x.shape
    m = x[..., 0].ne(0).int()
    positions = (torch.cumsum(m, dim=1).type_as(m) * m).long()
    tab = sinusoid_table(max(4096, T + 1), H)
    x = x + sd["pitch_predictor.pos_embed_alpha"] * tab.index_select(0, positions.view(-1)).view(B, T, -1)
    x = x.transpose(1, 2)
    for l in range(5):
        q = "pitch_predictor.conv.%d." % l
        w = sd[q + "1.weight"]
        k = w.shape[-1]
        x = F.relu(F.conv1d(F.pad(x, ((k - 1) // 2, (k - 1) // 2)), w, sd[q + "1.bias"]))
        x = F.layer_norm(x.transpose(1, 2), (w.shape[0],), sd[q + "3.weight"], sd[q + "3.bias"], 1e-12).transpose(1, 2)
    pred = F.linear(x.transpose(1, 2), sd["pitch_predictor.linear.weight"], sd["pitch_predictor.linear.bias"])
    f0 = pred[:, :, 0]
    if hp["pitch_norm"] == "standard":
        f0 = f0 * hp["f0_std"] + hp["f0_mean"]
    if hp["pitch_norm"] == "log":
        f0 = 2 ** f0
    else:
        f0 = f0.clone()
    if hp["pitch_type"] == "frame" and hp["use_uv"]:
        f0[pred[:, :, 1] > 0] = 0
    f0[pad] = 0
    return pred, f0
