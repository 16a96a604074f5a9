"""Mint tests/golden/*.npz by running the REAL reference (/root/reference, imported behind
oracle/refshim.py) on synthetic checkpoints and seeded inputs.  Container-only: the GPU box has no
/root/reference, it only sees the committed vectors.  TEST INFRASTRUCTURE.

    python oracle/make_golden.py

Random draws of the reference (torch.randn in diffusion.py:34-37,160,265-268 and models.py:192,271) are
replaced by the Philox streams of oracle/dsvc_oracle.py so that a GPU kernel can reproduce them.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import diffsvc_amd  # noqa: E402
from diffsvc_amd import synth  # noqa: E402
import dsvc_oracle as O  # noqa: E402
import refshim  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEED = 20260922


def build_reference_model(hp, sd):
    refshim.set_hparams(hp)
    from network.diff.diffusion import GaussianDiffusion
    from network.diff.net import DiffNet
    m = GaussianDiffusion(None, hp["audio_num_mel_bins"], DiffNet(hp["audio_num_mel_bins"]),
                          timesteps=hp["timesteps"], K_step=hp["K_step"], loss_type=hp["diff_loss_type"],
                          spec_min=hp["spec_min"], spec_max=hp["spec_max"])
    m.load_state_dict(sd, strict=True)          # the strict load the real tool does (utils/__init__.py:202)
    return m.eval()


def clip_batch(hp, clips, T, n_units):
    hub, m2p, f0 = [], [], []
    for c in clips:
        h, m, f, _ = synth.clip_inputs(c, T=T, n_units=n_units, H=hp["hidden_size"])
        hub.append(h); m2p.append(m); f0.append(f)
    return (torch.from_numpy(np.stack(hub)), torch.from_numpy(np.stack(m2p)), torch.from_numpy(np.stack(f0)))


def run_reference_sampler(model, hp, hub, m2p, f0, clips, speedup, seed):
    """GaussianDiffusion.forward(infer=True) with Philox noise injected."""
    import network.diff.diffusion as D
    B, T = m2p.shape
    M = hp["audio_num_mel_bins"]
    state = {"t": None}
    orig_noise_like, orig_p_sample, orig_randn = D.noise_like, model.p_sample, torch.randn

    def noise_like(shape, device, repeat=False):
        return O.ddpm_noise_ref_layout(seed, clips, state["t"], T, M)

    def p_sample(x, t, cond, **kw):
        state["t"] = int(t[0])
        return orig_p_sample(x, t, cond, **kw)

    def randn(*shape, **kw):
        shape = shape[0] if len(shape) == 1 and not isinstance(shape[0], int) else shape
        assert tuple(shape) == (B, 1, M, T), shape
        return O.ddpm_noise_ref_layout(seed, clips, 0, T, M, O.PURPOSE_X_INIT)

    D.noise_like = noise_like
    model.p_sample = p_sample
    torch.randn = randn
    refshim.set_hparams(dict(hp, pndm_speedup=speedup))
    try:
        with torch.no_grad():
            ret = model(hub.clone(), mel2ph=m2p.clone(), f0=f0.clone(), uv=None, energy=None, ref_mels=None, infer=True)
    finally:
        D.noise_like, model.p_sample, torch.randn = orig_noise_like, orig_p_sample, orig_randn
    return ret


def golden_diffnet(name, hp, wseed, B, T):
    sd = synth.acoustic_state(hp, wseed)
    model = build_reference_model(hp, sd)
    g = np.random.Generator(np.random.PCG64(SEED + T))
    M, H = hp["audio_num_mel_bins"], hp["hidden_size"]
    spec = torch.from_numpy(g.standard_normal((B, 1, M, T)).astype(np.float32))
    cond = torch.from_numpy((g.standard_normal((B, H, T)) * 0.5).astype(np.float32))
    t = torch.from_numpy(g.integers(0, hp["timesteps"], size=(B,)).astype(np.int64))
    with torch.no_grad():
        out = model.denoise_fn(spec, t, cond)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), spec=spec.numpy(), cond=cond.numpy(), t=t.numpy(),
                        out=out.numpy(), wseed=wseed)
    print(name, "out std %.3f" % out.std().item())


def golden_sampler(name, hp, wseed, clips, T, n_units, speedup, seed, conditioned=None, store_cond=True, slack=0.0):
    """conditioned = (lam, rho): synth.acoustic_state_conditioned -- a checkpoint whose noise prediction tracks its input like a
    trained model's, so that the unclamped PNDM chain contracts; the reference's own mel must then stay inside
    [spec_min, spec_max] (asserted here: a golden outside the data range is an ill-conditioned parity probe)."""
    sd = synth.acoustic_state_conditioned(hp, wseed, *conditioned) if conditioned else synth.acoustic_state(hp, wseed)
    model = build_reference_model(hp, sd)
    hub, m2p, f0 = clip_batch(hp, clips, T, n_units)
    ret = run_reference_sampler(model, hp, hub, m2p, f0, clips, speedup, seed)
    lo, hi = ret["mel_out"].min().item(), ret["mel_out"].max().item()
    if conditioned:     # (slack: a clip of thousands of frames reaches further into the tail of its own distribution -- T = 7000 leaves [-5, 0] by half a unit
        #  on a handful of frames in the reference itself; still a contracting chain)
        assert min(hp["spec_min"]) - slack <= lo and hi <= max(hp["spec_max"]) + slack, (name, lo, hi)
    extra = dict(decoder_inp=ret["decoder_inp"].numpy()) if store_cond else {}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), mel_out=ret["mel_out"].numpy(),
                        f0_denorm=ret["f0_denorm"].numpy(),
                        pitch=ret["pitch_pred"].numpy(), wseed=wseed, clips=np.array(clips), T=T, n_units=n_units,
                        speedup=speedup, seed=seed, K_step=hp["K_step"], conditioned=np.array(conditioned if conditioned else []), **extra)
    print(name, "mel range %.3f..%.3f" % (lo, hi))


def golden_cond_energy(name="cond_energy_tiny", clips=(0, 1, 2), T=40, n_units=23, wseed=3):
    """The ``use_energy_embed`` branch of the condition builder through the REAL FastSpeech2.forward (modules/fastspeech/fs2.py:81-82,143-144,
    240-247) with ``no_fs2: true``: decoder_inp = (gather + pitch_embed + energy_embed[clamp(energy * 256 // 4, max=255)]) * mask.  The
    GaussianDiffusion of the reference owns it as ``self.fs2``; only that module is run here (skip_decoder=True)."""
    hp = dict(synth.tiny_hparams(K=50), use_energy_embed=True)
    refshim.set_hparams(hp)
    from modules.fastspeech.fs2 import FastSpeech2
    fs2 = FastSpeech2(None, hp["audio_num_mel_bins"]).eval()
    H = hp["hidden_size"]
    g = np.random.Generator(np.random.PCG64(SEED + 31))
    pw = (g.standard_normal((300, H)) * H ** -0.5).astype(np.float32); pw[0] = 0
    ew = (g.standard_normal((256, H)) * H ** -0.5).astype(np.float32); ew[0] = 0
    with torch.no_grad():
        fs2.pitch_embed.weight.copy_(torch.from_numpy(pw))
        fs2.energy_embed.weight.copy_(torch.from_numpy(ew))
    hub, m2p, f0 = clip_batch(hp, clips, T, n_units)
    m2p[0, T - 6:] = 0                                     # padded frames
    energy = torch.from_numpy(g.uniform(0.0, 4.5, size=(len(clips), T)).astype(np.float32))     # beyond 4.0: the clamp at bin 255
    with torch.no_grad():
        ret = fs2(hub.clone(), mel2ph=m2p.clone(), f0=f0.clone(), uv=None, energy=energy.clone(), skip_decoder=True, infer=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), hubert=hub.numpy(), mel2ph=m2p.numpy(), f0=f0.numpy(), energy=energy.numpy(),
                        pitch_embed=pw, energy_embed=ew, decoder_inp=ret["decoder_inp"].numpy(), f0_denorm=ret["f0_denorm"].numpy(),
                        pitch=ret["pitch_pred"].numpy())
    print(name, "decoder_inp", tuple(ret["decoder_inp"].shape), "energy bins", int((energy * 256 // 4).clamp(max=255).min()), "..",
          int((energy * 256 // 4).clamp(max=255).max()))


def vocoder_inputs(h, clips, T):
    M = h["num_mels"]
    mels, f0s = [], []
    for c in clips:
        g = np.random.Generator(np.random.PCG64(SEED + 1000 + c))
        mels.append((g.standard_normal((T, M)) * 0.8 - 2.5).astype(np.float32))       # log10 mel, [T, M]
        _, _, _, f0_hz = synth.clip_inputs(c, T=T, n_units=max(2, T // 2), H=8)
        if c % 2 == 1:
            f0_hz = f0_hz * 2.7                                                         # exercise more phase wraps
        f0s.append(f0_hz)
    return np.stack(mels), np.stack(f0s)


def run_reference_vocoder(h, wseed, mel, f0, clips, seed):
    """Generator.forward of the REAL reference (weight-normed checkpoint -> load_state_dict ->
    remove_weight_norm, models.py:14-30) with the source module's torch.rand / randn_like replaced by the
    Philox streams.  mel [B,T,M] log10 numpy, f0 [B,T] Hz numpy -> wav [B, T*hop] numpy."""
    refshim.install()
    import modules.nsf_hifigan.models as NM
    from modules.nsf_hifigan.env import AttrDict
    sdw = synth.vocoder_state(h, wseed)
    gen = NM.Generator(AttrDict(h))
    gen.load_state_dict(sdw, strict=True)
    gen.eval()
    gen.remove_weight_norm()
    T = mel.shape[1]
    hop = int(np.prod(h["upsample_rates"]))
    ini, nz = O.vocoder_rng(seed, clips, T * hop)
    orig_rand, orig_randn_like = torch.rand, torch.randn_like

    def rand(*shape, **kw):
        assert tuple(shape) == tuple(ini.shape), shape
        return ini.clone()

    def randn_like(x, **kw):
        if x.shape[-1] == nz.shape[-1]:
            return nz.clone()
        return torch.zeros_like(x)              # SourceModuleHnNSF's noise branch is discarded by Generator (models.py:322,365)

    torch.rand, torch.randn_like = rand, randn_like
    try:
        with torch.no_grad():
            c = 2.30259 * torch.from_numpy(mel).transpose(2, 1)          # spec2wav (nsf_hifigan.py:63-65)
            wav = gen(c, torch.from_numpy(f0))
    finally:
        torch.rand, torch.randn_like = orig_rand, orig_randn_like
    return wav.numpy().reshape(len(clips), -1)


def golden_vocoder(name, h, wseed, clips, T, seed):
    mel, f0 = vocoder_inputs(h, clips, T)
    wav = run_reference_vocoder(h, wseed, mel, f0, clips, seed)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), mel=mel, f0=f0, wav=wav,
                        wseed=wseed, clips=np.array(clips), seed=seed)
    print(name, "wav rms %.4f max %.3f" % (float(np.sqrt((wav ** 2).mean())), float(np.abs(wav).max())))


def golden_vocoder_rb2():
    """Generators built from ResBlock2 (h.resblock = '2': modules/nsf_hifigan/models.py:73-91, selected at :337), the REAL Generator as above:
    the tiny architecture, and the public V2/V3-style shape of the 44.1 kHz config at reduced width (three kernel sizes x two dilations,
    channels 256 -> 16, so that every conv engine of the device path sees a ResBlock2 stage)."""
    tiny2 = dict(synth.tiny_vocoder(rds=((1, 3), (1, 3))), resblock="2")
    golden_vocoder("vocoder_tiny_rb2", tiny2, 6, clips=[0, 3], T=24, seed=93)
    wide2 = dict(synth.VOCODER_44K, resblock="2", resblock_dilation_sizes=[[1, 3], [1, 3], [1, 3]])
    golden_vocoder("vocoder_44k_rb2", wide2, 2, clips=[1], T=12, seed=94)
    # a valid reference config with THREE-entry dilation lists under resblock '2' (e.g. a V1 config switched to ResBlock2): the reference still
    # builds two convs per block, from dilation[0] and dilation[1] (models.py:77-82) -- ADVICE r4: the drop-in asked for a non-existent convs.2
    tiny2d3 = dict(synth.tiny_vocoder(rds=((1, 3, 5), (2, 4, 7))), resblock="2")
    golden_vocoder("vocoder_tiny_rb2_d3", tiny2d3, 7, clips=[2], T=24, seed=95)


def golden_headline(name="e2e_44k_T861_k1000", clips=(0, 1), T=861, n_units=500, seed=2026, wseed=0, vseed=1, K=1000, speedup=1, with_wav=True,
                    conditioned=None, assert_clamp=True):
    """The BENCHMARKED configuration (BASELINE configs[1]: 10 s clip, T=861, 44.1 kHz architecture, full 1000-step DDPM) through
    the REAL reference end to end: GaussianDiffusion.forward(infer=True) (diffusion.py:227-284) -> the host glue of
    Svc.after_infer (clip to [mel_vmin, mel_vmax], infer_tool.py:177-183) -> Generator.forward (models.py:361-387) for the first
    clip.  ~2 x 1000 reference denoiser evaluations at T=861: minutes on 8 cores.  Inputs are regenerated from synth.clip_inputs
    on the test side (the cond builder is pinned bit for bit elsewhere), so only the outputs are stored.
    conditioned = (lam, rho): the checkpoint is synth.acoustic_state_conditioned -- a denoiser whose noise prediction tracks its input
    (eps ~= lam * x + rho * random net), so that p_sample's clamp(x0, -1, 1) (diffusion.py:149-150) is NOT what decides the output:
    with a random-init DiffNet 36 % of the reference mel sits exactly on spec_min / spec_max, where any error is invisible; here the
    fraction on the clamp is asserted below 1 % when minting."""
    import time
    clips = list(clips)
    hp = dict(synth.HPARAMS_44K, K_step=K)
    sd = synth.acoustic_state_conditioned(hp, wseed, *conditioned) if conditioned else synth.acoustic_state(hp, wseed)
    model = build_reference_model(hp, sd)
    hub, m2p, f0 = clip_batch(hp, clips, T, n_units)
    t0 = time.time()
    ret = run_reference_sampler(model, hp, hub, m2p, f0, clips, speedup, seed)
    mel = ret["mel_out"].numpy()
    on_clamp = float(((mel <= min(hp["spec_min"])) | (mel >= max(hp["spec_max"]))).mean())
    print(name, "sampler %.0f s, mel range %.3f..%.3f, %.2f %% of the mel on spec_min/spec_max" % (time.time() - t0, mel.min(), mel.max(), 100 * on_clamp))
    if conditioned and assert_clamp:
        assert on_clamp < 0.01, (name, on_clamp)
    extra = {}
    if with_wav:
        h = dict(synth.VOCODER_44K)
        mel_c = np.clip(mel[:1], hp["mel_vmin"], hp["mel_vmax"])
        f0_hz = ret["f0_denorm"].numpy()[:1]
        wav = run_reference_vocoder(h, vseed, mel_c, f0_hz, clips[:1], seed)
        extra["wav0"] = wav[0]
        print(name, "wav rms %.4f max %.3f" % (float(np.sqrt((wav ** 2).mean())), float(np.abs(wav).max())))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), mel_out=mel, f0_denorm=ret["f0_denorm"].numpy(),
                        pitch=ret["pitch_pred"].numpy().astype(np.int16), wseed=wseed, vseed=vseed, clips=np.array(clips), T=T,
                        n_units=n_units, speedup=speedup, seed=seed, K_step=K, conditioned=np.array(conditioned if conditioned else []),
                        on_clamp=on_clamp, **extra)


# Round 3: the spread of the 1000-step error is a heavy-tailed statistic (profiles/r2w_precision_spread.txt), so the shipped precision is
# held to the bar on MANY real-reference realisations, not on four.  All of these share seed 2026 with the two clips of
# e2e_44k_T861_k1000, so that ONE batch of 32 clips (clips 0..31, seed 2026 -- the per-GPU share of BASELINE configs[3]) contains every
# one of them: tests/test_gpu_headline.py::test_batch_of_32_* checks each clip of the batch that has a golden.
SPREAD_SEED = 2026
SPREAD_CLIPS = (2, 3, 5, 7, 8, 9, 10, 11, 12, 13)                    # random-init checkpoint (wseed 0), seed 2026
SPREAD_COND = (("ca", (1.5, 0.07), (0, 1)),                            # the PLMS probes' conditioned checkpoint
               ("cb", (1.2, 0.15), (0, 1, 2, 3)))                      # twice the random-network share: 0.6 % of the mel on the clamp


# Round 4: EVERY clip of two per-GPU shares of BASELINE configs[3] (clips 0..31 and 32..63 of the 256-clip job, seed 2026) has a golden, so
# that tests/test_gpu_headline.py::test_batch_of_32_* checks all 64 clips of two batches and fits the tail of the error distribution on 64
# samples instead of 12.  Clips 0 and 1 are the two rows of e2e_44k_T861_k1000; SPREAD_CLIPS were minted in round 3.
SHARE_CLIPS = tuple(c for c in range(64) if c not in (0, 1) and c not in SPREAD_CLIPS)


def share_golden(clip):
    """(golden name, row) of clip `clip` (seed 2026, random-init checkpoint) of the 64-clip double share."""
    return ("e2e_44k_T861_k1000", clip) if clip in (0, 1) else ("e2e_44k_T861_k1000_s2026_c%d" % clip, 0)


def spread_names(share=False):
    out = [("e2e_44k_T861_k1000_s2026_c%d" % c, c, SPREAD_SEED, None) for c in SPREAD_CLIPS]
    if share:
        return [("e2e_44k_T861_k1000_s2026_c%d" % c, c, SPREAD_SEED, None) for c in SHARE_CLIPS]
    out.append(("e2e_44k_T861_k1000_c9", 9, 1009, None))               # the shipped round-2 precision's worst (clip, noise) pair
    for tag, cond, clips in SPREAD_COND:
        out += [("e2e_44k_T861_k1000_%s_c%d" % (tag, c), c, SPREAD_SEED, cond) for c in clips]
    return out


def golden_headline_spread(only=None):
    """17 more single-clip runs of the benchmarked configuration through the REAL reference (~80 s each on 8 cores); skips files that
    exist, so an interrupted run resumes."""
    for name, clip, seed, cond in spread_names() + spread_names(share=True):
        if only and name not in only:
            continue
        if os.path.exists(os.path.join(OUT, name + ".npz")) and "--force" not in sys.argv:
            print(name, "exists")
            continue
        golden_headline(name=name, clips=(clip,), seed=seed, with_wav=False, conditioned=cond)


# Round 6: the shipped batched precision f16_w6 measures 4.0e-4 ... 6.1e-4 on the random-init checkpoint and 1.3e-4 ... 1.6e-4
# on the two conditioned ones -- how does its 1000-step error move BETWEEN them?  Three more checkpoints on the line from `ca` (eps tracks x,
# like a trained model: lam 1.5, rho 0.07) to random-init (lam 0, rho 1), eight clips each (clips 0..7, seed 2026: one batch of 8 on the GPU).
# The fraction of the reference's own mel that sits on p_sample's clamp is stored with every golden (`on_clamp`), not asserted.
WSTAT_CKPTS = (("w1", (1.0, 0.3)), ("w2", (0.6, 0.55)), ("w3", (0.25, 0.8)))
WSTAT_CLIPS = tuple(range(8))


def golden_weight_statistics():
    for tag, par in WSTAT_CKPTS:
        for c in WSTAT_CLIPS:
            name = "e2e_44k_T861_k1000_%s_c%d" % (tag, c)
            if os.path.exists(os.path.join(OUT, name + ".npz")) and "--force" not in sys.argv:
                print(name, "exists")
                continue
            golden_headline(name=name, clips=(c,), seed=SPREAD_SEED, with_wav=False, conditioned=par, assert_clamp=False)


def golden_headline_extra():
    """Two more single-clip runs of the benchmarked configuration through the real reference: the (clip, seed) pairs on which a spread
    study of the HIP path's 1000-step error (tools/study_headline_spread.py, profiles/r2w_*) found its largest values."""
    golden_headline(name="e2e_44k_T861_k1000_c4", clips=(4,), seed=1004, with_wav=False)
    golden_headline(name="e2e_44k_T861_k1000_c6", clips=(6,), seed=1006, with_wav=False)


TRAIN_CASES = (  # name, arch, loss, clips, T, n_units, seed  (the batches of tests/test_gpu_train.py)
    ("tiny_l2", "tiny", "l2", (0, 1, 2), 40, 23, 5), ("tiny_l1", "tiny", "l1", (0, 1, 2), 40, 23, 5),
    ("44k_l2", "44k", "l2", (4, 9), 64, 37, 6), ("44k_l1", "44k", "l1", (4, 9), 64, 37, 6))


# Round 4: BASELINE configs[4] AT THE BENCHMARKED SIZE -- exactly the batch bench.py --train times on rank 0 (bench.train_batch: clips 0..63,
# 128 frames, 74 content units, PCG64(77)): 8 704 rows, where the many-row conv tilings, the XCD-sliced weight gradients, the 128-row pgemm
# tiles and the 2^14 loss scale are in play.  Its own file (the 44.1 kHz lattice of all 171 tensors is ~5 MB).
TRAIN_CASES_BENCH = (("bench64x128_l2", "44k", "l2", tuple(range(64)), 128, 74, 77),)


def golden_train(cases=TRAIN_CASES, fname="train_grads"):
    """Training parity pin (BASELINE configs[4]): the REAL GaussianDiffusion.forward(infer=False) ->
    Batch2Loss.module4 -> p_losses (diffusion.py:207-241, train_pipeline.py:222-238) with torch autograd, on the batches of
    tests/test_gpu_train.py.  The two random draws of the training forward are injected: ``torch.randint`` (the diffusion steps t,
    train_pipeline.py:233) returns the batch's t, ``torch.randn_like`` (the noise, diffusion.py:208) the Philox training stream.
    Stored: the loss, the L2 norm of every parameter gradient, and the gradients themselves (tiny architecture: all of them;
    44.1 kHz: small tensors whole, large ones on the lattice of synth.train_grad_slices)."""
    out = {}
    for name, arch, loss_type, clips, T, n_units, seed in cases:
        hp = dict(synth.tiny_hparams(K=50) if arch == "tiny" else synth.HPARAMS_44K, diff_loss_type=loss_type)
        sd = synth.acoustic_state(hp, 3)
        model = build_reference_model(hp, sd)
        model.train()                                   # (DiffNet has no dropout / batch norm: train() == eval() numerically)
        hub, m2p, f0, mels, t = (torch.from_numpy(v) for v in synth.train_batch_kat(hp, clips, T, n_units, seed))
        M = hp["audio_num_mel_bins"]
        noise = O.ddpm_noise_ref_layout(seed, list(clips), 0, T, M, O.PURPOSE_TRAIN_NOISE)
        orig_randint, orig_randn_like = torch.randint, torch.randn_like
        calls = {"randint": 0, "randn_like": 0}

        def randint(lo, hi, size, **kw):
            assert (lo, hi, tuple(size)) == (0, hp["K_step"], (len(clips),)), (lo, hi, size)
            calls["randint"] += 1
            return t.clone()

        def randn_like(x, **kw):
            assert tuple(x.shape) == tuple(noise.shape), x.shape
            calls["randn_like"] += 1
            return noise.clone()

        torch.randint, torch.randn_like = randint, randn_like
        try:
            ret = model(hub.clone(), mel2ph=m2p.clone(), f0=f0.clone(), uv=None, energy=None, ref_mels=mels.clone(), infer=False)
            loss = ret["diff_loss"]
            loss.backward()
        finally:
            torch.randint, torch.randn_like = orig_randint, orig_randn_like
        assert calls == {"randint": 1, "randn_like": 1}, calls
        out[name + "/loss"] = np.float64(loss.item())
        names, norms = [], []
        for k, p in model.named_parameters():
            if not (k.startswith("denoise_fn.") or k == "fs2.pitch_embed.weight"):
                continue
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            names.append(k); norms.append(float(g.double().norm().item()))
            sl = synth.train_grad_slices(tuple(g.shape)) if arch != "tiny" else tuple(slice(None) for _ in g.shape)
            out[name + "/grad/" + k] = g[sl].detach().numpy().copy()
        out[name + "/names"] = np.array(names)
        out[name + "/norms"] = np.array(norms, dtype=np.float64)
        print("train", name, "loss %.6f, %d gradient tensors, |g| %.4f" % (loss.item(), len(names), float(np.sqrt((np.array(norms) ** 2).sum()))))
    np.savez_compressed(os.path.join(OUT, fname + ".npz"), **out)


def golden_train_bench_f64():
    """The benchmarked training batch once more in FLOAT64 (the oracle's restatement, which tests/test_oracle_golden.py pins to the real
    reference at 2e-5, evaluated with a float64 state dict): the yardstick for the fp32 rounding of the reference ITSELF.  At 8 192 frames the
    conditioner-projection weight gradients are sums with heavy cancellation -- the real reference's fp32 autograd sits 2.2e-4 ... 2.7e-4 from
    this evaluation on those 20 tensors (asserted below), so a comparison with the fp32 golden alone cannot ask for 5e-5 there;
    tests/test_gpu_train.py holds the HIP step to the fp64 values instead.  Same lattice as train_grads_bench.npz."""
    (name, arch, loss_type, clips, T, n_units, seed), = TRAIN_CASES_BENCH
    hp = dict(synth.HPARAMS_44K, diff_loss_type=loss_type)
    sd = synth.acoustic_state(hp, 3)
    hub, m2p, f0, mels, t = (torch.from_numpy(v) for v in synth.train_batch_kat(hp, clips, T, n_units, seed))
    noise = O.ddpm_noise_ref_layout(seed, list(clips), 0, T, hp["audio_num_mel_bins"], O.PURPOSE_TRAIN_NOISE)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    loss, grads = O.train_loss_and_grads(sd64, hub.double(), m2p, f0.double(), mels.double(), t, noise.double(), hp)
    g32 = np.load(os.path.join(OUT, "train_grads_bench.npz"))
    out = {name + "/loss": np.float64(loss.item())}
    worst = {}
    for k in (str(n) for n in g32[name + "/names"]):
        part = grads[k][synth.train_grad_slices(tuple(grads[k].shape))]
        out[name + "/grad/" + k] = part.numpy().copy()
        ref = torch.from_numpy(g32[name + "/grad/" + k]).double()
        if part.norm().item() > 0:
            kind = k.split(".")[-2] if "residual_layers" in k else k
            worst[kind] = max(worst.get(kind, 0.0), (part - ref).norm().item() / part.norm().item())
    print("train f64: loss %.9f (fp32 reference %.9f); the fp32 reference's relative L2 distance from the fp64 values, worst per tensor kind:" % (loss.item(), float(g32[name + "/loss"])))
    for kind, e in sorted(worst.items(), key=lambda kv: -kv[1]):
        print("   %-40s %.2e" % (kind, e))
    assert 1e-4 < worst["conditioner_projection"] < 5e-4
    np.savez_compressed(os.path.join(OUT, "train_grads_bench_f64.npz"), **out)


def golden_train_bench_kinks():
    """Where the benchmarked training batch sits on a ReLU kink: the FLOAT64 pre-activations of the two ReLUs of DiffNet
    (input projection, net.py:120-123; skip projection, :132-133) on exactly the batch of train_grads_bench.npz -- per output channel the
    smallest |pre-activation| over all frames, and the frame it occurs at.  A weight-gradient ROW of these two convs jumps by one frame's
    whole term when the fp32 evaluation's sign of such a pre-activation differs from the fp64 one's; tests/test_gpu_train.py may set ONE row
    of these two tensors aside -- and only if this file says that row has a pre-activation within fp32 rounding of zero."""
    (name, arch, loss_type, clips, T, n_units, seed), = TRAIN_CASES_BENCH
    hp = dict(synth.HPARAMS_44K, diff_loss_type=loss_type)
    sd = synth.acoustic_state(hp, 3)
    hub, m2p, f0, mels, t = (torch.from_numpy(v) for v in synth.train_batch_kat(hp, clips, T, n_units, seed))
    noise = O.ddpm_noise_ref_layout(seed, list(clips), 0, T, hp["audio_num_mel_bins"], O.PURPOSE_TRAIN_NOISE)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    out = {}
    with torch.no_grad():
        cond, _, _ = O.build_cond(sd64, hub.double(), m2p, f0.double().clone(), hp)
        x0 = O.norm_spec(sd64, mels.double()).transpose(1, 2)[:, None, :, :]
        x_noisy = O.q_sample(sd64, x0, t, noise.double())
        taps = {}
        O.diffnet_forward(sd64, x_noisy, t, cond.transpose(1, 2), hp["dilation_cycle_length"], taps=taps)
        pre = {"denoise_fn.input_projection.weight": F.conv1d(x_noisy[:, 0], sd64["denoise_fn.input_projection.weight"], sd64["denoise_fn.input_projection.bias"]),
               "denoise_fn.skip_projection.weight": F.conv1d(taps["skip"], sd64["denoise_fn.skip_projection.weight"], sd64["denoise_fn.skip_projection.bias"])}
    for k, v in pre.items():
        a = v.abs().permute(1, 0, 2).reshape(v.shape[1], -1)              # [channel][clip * T + frame]
        mn, at = a.min(1)
        out[k + "/min_abs_preact"] = mn.numpy()
        out[k + "/at"] = at.numpy()
        order = mn.argsort()[:3]
        print("%s: rows nearest a ReLU kink %s, |pre-activation| %s (median row %.2e)" % (k, order.tolist(), ["%.1e" % mn[i].item() for i in order], mn.median().item()))
    np.savez_compressed(os.path.join(OUT, "train_grads_bench_kinks.npz"), **out)


def golden_melspec(name, sr, n_fft, win, hop, n_mels, fmin, fmax, n_samples):
    """STFT.get_mel of the REAL reference (nvSTFT.py:72-104).  Two shims are unavoidable on this image:
    librosa's mel filterbank is replaced by the oracle's restatement (librosa is not installed), and
    torch.stft gets return_complex=True + view_as_real (the legacy real-view output was removed in torch 2)."""
    refshim.install()
    import modules.nsf_hifigan.nvSTFT as NV
    NV.librosa_mel_fn = lambda sr, n_fft, n_mels, fmin, fmax: O.mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    g = np.random.Generator(np.random.PCG64(SEED + n_samples))
    t = np.arange(n_samples) / sr
    wav = (0.3 * np.sin(2 * np.pi * 220 * t * (1 + 0.1 * np.sin(2 * np.pi * 3 * t))) + 0.05 * g.standard_normal(n_samples)
           + 0.2 * np.sin(2 * np.pi * 3100 * t)).astype(np.float32)
    wav[n_samples // 3: n_samples // 3 + 2 * n_fft] *= 1e-4                         # a near-silent stretch
    orig_stft = torch.stft

    def stft(*a, **kw):
        kw["return_complex"] = True
        return torch.view_as_real(orig_stft(*a, **kw))

    torch.stft = stft
    try:
        st = NV.STFT(sr, n_mels, n_fft, win, hop, fmin, fmax)
        with torch.no_grad():
            mel = st.get_mel(torch.from_numpy(wav)[None])[0].T * 0.434294          # wav2spec's log10 scale (nsf_hifigan.py:89-91)
    finally:
        torch.stft = orig_stft
    np.savez_compressed(os.path.join(OUT, name + ".npz"), wav=wav, mel=mel.numpy(), cfg=np.array([sr, n_fft, win, hop, n_mels, fmin, fmax]))
    print(name, "mel", tuple(mel.shape), "range %.2f..%.2f" % (mel.min().item(), mel.max().item()))


def golden_state_keys():
    """Names and shapes of the state dicts the REAL reference modules own (the strict-load contract of
    utils/__init__.py:178-209 and models.py:14-30), for the two acoustic configs and the vocoder."""
    import json
    out = {}
    for tag, hp in (("tiny", synth.tiny_hparams()), ("44k", dict(synth.HPARAMS_44K))):
        model = build_reference_model(hp, synth.acoustic_state(hp, 0))
        out["acoustic_" + tag] = {k: list(v.shape) for k, v in model.state_dict().items()}
    refshim.install()
    import modules.nsf_hifigan.models as NM
    from modules.nsf_hifigan.env import AttrDict
    for tag, h in (("tiny", synth.tiny_vocoder()), ("44k", dict(synth.VOCODER_44K))):
        gen = NM.Generator(AttrDict(h))
        out["vocoder_" + tag] = {k: list(v.shape) for k, v in gen.state_dict().items()}
    with open(os.path.join(OUT, "state_keys.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("state_keys", {k: len(v) for k, v in out.items()})


def main():
    os.makedirs(OUT, exist_ok=True)
    if "--keys-only" in sys.argv:
        return golden_state_keys()
    if "--24k-only" in sys.argv:
        return golden_24k()
    if "--slicer-only" in sys.argv:
        golden_slicer_demo_input()
        return golden_slicer()
    if "--schedule-only" in sys.argv:
        return golden_schedule()
    if "--train-kinks" in sys.argv:
        return golden_train_bench_kinks()
    if "--rb2-only" in sys.argv:
        return golden_vocoder_rb2()
    if "--cond-energy" in sys.argv:
        return golden_cond_energy()
    if "--headline-only" in sys.argv:
        return golden_headline()
    if "--headline-extra" in sys.argv:
        return golden_headline_extra()
    if "--headline-spread" in sys.argv:
        return golden_headline_spread()
    if "--train-only" in sys.argv:
        return golden_train()
    if "--train-bench" in sys.argv:
        golden_train(TRAIN_CASES_BENCH, "train_grads_bench")
        return golden_train_bench_f64()
    if "--train-bench-f64" in sys.argv:
        return golden_train_bench_f64()
    if "--plms-only" in sys.argv:
        return golden_plms_conditioned()
    if "--plms-t861-more" in sys.argv:
        return golden_plms_t861_more()
    if "--long-only" in sys.argv:
        return golden_long()
    if "--long-1000" in sys.argv:
        return golden_long_1000()
    if "--weight-statistics" in sys.argv:
        return golden_weight_statistics()
    if "--hifigan-only" in sys.argv:
        return golden_hifigan_24k()
    if "--hubert-only" in sys.argv:
        return golden_hubert()
    if "--pe-only" in sys.argv:
        return golden_pe()
    golden_state_keys()
    golden_vocoder("vocoder_tiny", synth.tiny_vocoder(), 5, clips=[0, 3], T=24, seed=90)
    golden_vocoder("vocoder_44k", dict(synth.VOCODER_44K), 1, clips=[1], T=12, seed=91)
    golden_vocoder_rb2()
    golden_melspec("melspec_44k", 44100, 2048, 2048, 512, 128, 40, 16000, 20000)
    golden_melspec("melspec_24k", 24000, 512, 512, 128, 80, 30, 12000, 6000)
    tiny = synth.tiny_hparams()
    full = dict(synth.HPARAMS_44K)
    golden_diffnet("diffnet_tiny", tiny, 3, B=2, T=40)
    golden_diffnet("diffnet_44k", full, 0, B=2, T=48)
    golden_sampler("ddpm_tiny", tiny, 3, clips=[0, 5], T=40, n_units=23, speedup=1, seed=77)
    golden_sampler("plms_tiny_s10", tiny, 3, clips=[2], T=40, n_units=23, speedup=10, seed=78)
    golden_sampler("plms_tiny_s5", tiny, 3, clips=[1], T=52, n_units=30, speedup=5, seed=79)
    golden_sampler("ddpm_44k_k20", dict(full, K_step=20), 0, clips=[0], T=32, n_units=19, speedup=1, seed=80)
    golden_sampler("plms_44k_k100_s20", dict(full, K_step=100), 0, clips=[4], T=32, n_units=19, speedup=20, seed=81)
    golden_24k()
    golden_hifigan_24k()
    golden_hubert()
    golden_pe()
    golden_plms_conditioned()
    golden_headline_extra()
    golden_headline_spread()
    golden_train()
    golden_train(TRAIN_CASES_BENCH, "train_grads_bench")
    golden_train_bench_f64()
    golden_slicer()
    golden_slicer_demo_input()
    golden_schedule()
    golden_cond_energy()
    golden_plms_t861_more()
    golden_long()
    golden_long_1000()
    golden_weight_statistics()


# Round 6: BASELINE configs[2] at the benchmarked size rested on ONE (clip, noise) pair.  Five more, same conditioned
# checkpoint, 51 evaluations of the REAL reference each.
PLMS_T861_MORE = ((1, 85), (2, 86), (3, 87), (4, 88), (7, 89))         # (clip, seed); (5, 88) left the data range on the reference itself (mel max +0.078): not a well-conditioned probe


def golden_plms_t861_more():
    full = dict(synth.HPARAMS_44K)
    for clip, seed in PLMS_T861_MORE:
        golden_sampler("plmsc_44k_T861_s20_c%d" % clip, full, 0, clips=[clip], T=861, n_units=500, speedup=20, seed=seed,
                       conditioned=(1.5, 0.07), store_cond=False)


# Round 6: the sampler beyond T = 861.  The reference accepts max_frames 42000 (training/config_nsf.yaml:82) and
# its slicer hands out chunks of 5 ... 30 s and more; T = 2600 (30 s) is 82 frame tiles of 32 on the small tilings, T = 7000 (81 s) crosses
# the fused-layer threshold with ONE clip.  20-step DDPM (K_step 20 of the 1000-step schedule) and 50-iteration PLMS of the
# REAL reference at both lengths, both architectures (PLMS: pndm_speedup 20 -- at 50 the conditioned synthetic checkpoint's own chain leaves the data range); three clips of different lengths run one by one (the reference is B = 1) that the
# drop-in runs as ONE ragged batch.
LONG_RAGGED = ((0, 2000, 1161), (1, 1500, 871), (2, 1111, 645))         # (clip, T, n_units)


def golden_long():
    full, k24 = dict(synth.HPARAMS_44K), dict(synth.HPARAMS_24K)

    def once(name, *a, **kw):
        if os.path.exists(os.path.join(OUT, name + ".npz")) and "--force" not in sys.argv:
            print(name, "exists")
            return
        golden_sampler(name, *a, **kw)

    for T, nu in ((2600, 1510), (7000, 4065)):
        once("ddpm_44k_k20_T%d" % T, dict(full, K_step=20), 0, clips=[0], T=T, n_units=nu, speedup=1, seed=101, store_cond=False)
        once("plmsc_44k_s20_T%d" % T, full, 0, clips=[1], T=T, n_units=nu, speedup=20, seed=102, conditioned=(1.5, 0.07), store_cond=False, slack=0.75)
        once("ddpm_24k_k20_T%d" % T, dict(k24, K_step=20), 2, clips=[2], T=T, n_units=nu, speedup=1, seed=103, store_cond=False)
    once("plmsc_24k_s20_T2600", k24, 2, clips=[3], T=2600, n_units=1510, speedup=20, seed=104, conditioned=(1.35, 0.05), store_cond=False)
    if os.path.exists(os.path.join(OUT, "ddpm_44k_k20_ragged3.npz")) and "--force" not in sys.argv:
        return
    hp = dict(full, K_step=20)
    sd = synth.acoustic_state(hp, 0)
    model = build_reference_model(hp, sd)
    out = {}
    for clip, T, nu in LONG_RAGGED:
        hub, m2p, f0 = clip_batch(hp, [clip], T, nu)
        ret = run_reference_sampler(model, hp, hub, m2p, f0, [clip], 1, 105)
        out["mel_c%d" % clip] = ret["mel_out"].numpy()[0]
    np.savez_compressed(os.path.join(OUT, "ddpm_44k_k20_ragged3.npz"), wseed=0, seed=105, K_step=20, clips=np.array([c for c, _, _ in LONG_RAGGED]),
                        T=np.array([t for _, t, _ in LONG_RAGGED]), n_units=np.array([n for _, _, n in LONG_RAGGED]), **out)
    print("ddpm_44k_k20_ragged3", {k: v.shape for k, v in out.items()})


def golden_long_1000():
    """The full 1000-step DDPM chain of the REAL reference on ONE clip of T = 2600 (f16_x3t on 82 small tiles) and ONE of T = 7000 (the fused
    f16_w6 layer kernel with a single clip): ~4 and ~11 minutes on 8 cores."""
    for T, nu, clip in ((2600, 1510, 0), (7000, 4065, 1)):
        name = "e2e_44k_T%d_k1000" % T
        if os.path.exists(os.path.join(OUT, name + ".npz")) and "--force" not in sys.argv:
            print(name, "exists")
            continue
        golden_headline(name=name, clips=(clip,), T=T, n_units=nu, seed=2027, with_wav=False)


def golden_plms_conditioned():
    """PLMS/PNDM parity probes on conditioned checkpoints (see golden_sampler): the 44.1 kHz architecture with the full 1000-step
    schedule at pndm_speedup=20 (50 iterations / 51 evaluations -- BASELINE configs[2]) at a small frame count and at the
    benchmarked T=861, the 24 kHz demo architecture at pndm_speedup=50 (BASELINE configs[0]), and the tiny architecture."""
    tiny = synth.tiny_hparams(K=100)
    full = dict(synth.HPARAMS_44K)
    golden_sampler("plmsc_tiny_s10", tiny, 3, clips=[2], T=40, n_units=23, speedup=10, seed=78, conditioned=(2.0, 0.05))
    golden_sampler("plmsc_tiny_s5", tiny, 3, clips=[1], T=52, n_units=30, speedup=5, seed=79, conditioned=(2.0, 0.05))
    golden_sampler("plmsc_44k_s20", full, 0, clips=[4], T=32, n_units=19, speedup=20, seed=81, conditioned=(1.5, 0.07))
    golden_sampler("plmsc_24k_s50", dict(synth.HPARAMS_24K), 2, clips=[3], T=36, n_units=21, speedup=50, seed=82, conditioned=(1.35, 0.05))
    golden_sampler("plmsc_44k_T861_s20", full, 0, clips=[0], T=861, n_units=500, speedup=20, seed=84, conditioned=(1.5, 0.07), store_cond=False)


def golden_schedule():
    """The 12 schedule buffers as the REAL GaussianDiffusion.__init__ computes them (diffusion.py:87-120), for the linear
    (max_beta 0.02, 44.1 kHz config) and the cosine schedule -- nothing loaded on top."""
    refshim.set_hparams(dict(synth.HPARAMS_44K))
    from network.diff.diffusion import GaussianDiffusion
    from network.diff.net import DiffNet
    out = {}
    for tag, extra in (("linear", dict(schedule_type="linear", max_beta=0.02)), ("cosine", dict(schedule_type="cosine"))):
        hp = dict(synth.HPARAMS_44K, **extra)
        refshim.set_hparams(hp)
        m = GaussianDiffusion(None, 128, DiffNet(128), timesteps=1000, K_step=1000, loss_type="l2", spec_min=hp["spec_min"], spec_max=hp["spec_max"])
        for k in O.SCHEDULE_KEYS:
            out[tag + "_" + k] = getattr(m, k).detach().numpy().copy()
        print("schedule", tag, float(out[tag + "_betas"][0]), float(out[tag + "_betas"][-1]))
    np.savez_compressed(os.path.join(OUT, "schedule.npz"), **out)


def golden_slicer():
    """Slicer KATs: the real ``infer_tools.slicer.Slicer`` on the synthetic signals of synth.SLICER_CASES."""
    import contextlib, io
    refshim.install()
    from infer_tools.slicer import Slicer
    out = []
    for case in synth.SLICER_CASES:
        audio = synth.slicer_audio(case)
        with contextlib.redirect_stdout(io.StringIO()):
            chunks = Slicer(sr=case["sr"], **case["args"]).slice(audio)
        out.append({"case": case, "n_samples": int(audio.shape[0]), "chunks": chunks})
        print("slicer seed", case["seed"], len(chunks), "chunks", [v["split_time"] for v in chunks.values()][:6])
    with open(os.path.join(OUT, "slicer_kat.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)


def golden_slicer_demo_input():
    """The real Slicer on the reference's shipped demo input raw/test_input.wav at six parameter sets."""
    import contextlib, io, wave
    refshim.install()
    from infer_tools.slicer import Slicer
    with wave.open(os.path.join(refshim.REF_ROOT, "raw", "test_input.wav"), "rb") as w:
        sr, n = w.getframerate(), w.getnframes()
        audio = np.frombuffer(w.readframes(n), dtype="<i2").astype(np.float32) / 32768.0
    cases = []
    for args in (dict(db_threshold=-40), dict(db_threshold=-30), dict(db_threshold=-40, min_length=3000),
                 dict(db_threshold=-35, win_l=400, win_s=30, max_silence_kept=800), dict(db_threshold=-50, min_length=8000),
                 dict(db_threshold=-25, win_l=200, win_s=10, max_silence_kept=300)):
        with contextlib.redirect_stdout(io.StringIO()):
            chunks = Slicer(sr=sr, **args).slice(audio)
        cases.append({"args": args, "chunks": chunks})
        print("slicer test_input.wav", args, [v["split_time"] for v in chunks.values()][:8])
    with open(os.path.join(OUT, "slicer_test_input.json"), "w") as f:
        json.dump({"sr": sr, "n_samples": n, "cases": cases}, f, indent=0, sort_keys=True)


def golden_hifigan_24k(name="hifigan_24k", clips=(1, 2), T=10, seed=92, wseed=7):
    """The 24 kHz generator of the demo config: the REAL modules/hifigan/hifigan.py HifiGanGenerator (weight-normed checkpoint ->
    strict load -> remove_weight_norm, as network/vocoders/hifigan.py:30-37 does) with and without an f0, source-module random
    draws replaced by the Philox streams.  Natural-log mel, fed unscaled (hifigan.py:64)."""
    refshim.install()
    import modules.hifigan.hifigan as HG
    clips = list(clips)
    h = dict(synth.VOCODER_24K)
    sdw = synth.vocoder_state(h, wseed)
    gen = HG.HifiGanGenerator(h)
    gen.load_state_dict(sdw, strict=True)
    gen.eval()
    gen.remove_weight_norm()
    mel, f0 = vocoder_inputs(h, clips, T)
    hop = int(np.prod(h["upsample_rates"]))
    ini, nz = O.vocoder_rng(seed, clips, T * hop)
    orig_rand, orig_randn_like = torch.rand, torch.randn_like

    def rand(*shape, **kw):
        assert tuple(shape) == tuple(ini.shape), shape
        return ini.clone()

    def randn_like(x, **kw):
        return nz.clone() if x.shape[-1] == nz.shape[-1] else torch.zeros_like(x)

    torch.rand, torch.randn_like = rand, randn_like
    try:
        with torch.no_grad():
            c = torch.from_numpy(mel).transpose(2, 1)
            wav_src = gen(c, torch.from_numpy(f0)).numpy().reshape(len(clips), -1)
            wav_plain = gen(c).numpy().reshape(len(clips), -1)
    finally:
        torch.rand, torch.randn_like = orig_rand, orig_randn_like
    np.savez_compressed(os.path.join(OUT, name + ".npz"), mel=mel, f0=f0, wav_src=wav_src, wav_plain=wav_plain, wseed=wseed,
                        clips=np.array(clips), seed=seed)
    print(name, "wav rms with source %.4f, plain %.4f" % (float(np.sqrt((wav_src ** 2).mean())), float(np.sqrt((wav_plain ** 2).mean()))))
    import json as _json
    with open(os.path.join(OUT, "state_keys.json")) as f:
        keys = _json.load(f)
    keys["vocoder_24k"] = {k: list(v.shape) for k, v in HG.HifiGanGenerator(h).state_dict().items()}
    with open(os.path.join(OUT, "state_keys.json"), "w") as f:
        _json.dump(keys, f, indent=0, sort_keys=True)


def golden_hubert(name="hubert_units", lengths=(16000, 33333), wseed=11):
    """HubertSoft.units of the REAL reference (network/hubert/hubert_model.py) on a synthetic checkpoint (strict load, eval) for two
    utterance lengths (an even and an odd frame-count chain through the seven strided convs)."""
    refshim.set_hparams(dict(synth.HPARAMS_44K))
    from network.hubert.hubert_model import HubertSoft
    sd = synth.hubert_state(wseed)
    m = HubertSoft()
    m.load_state_dict(sd, strict=True)
    m.eval()
    out = {"wseed": wseed, "lengths": np.array(lengths)}
    for i, n in enumerate(lengths):
        wav = synth.speech_like_wav(100 + i, n)
        with torch.no_grad():
            u = m.units(torch.from_numpy(wav)[None, None])
        out["units%d" % i] = u[0].numpy()
        print(name, n, "samples ->", tuple(u.shape), "units std %.3f max %.3f" % (u.std().item(), u.abs().max().item()))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    import json as _json
    with open(os.path.join(OUT, "state_keys.json")) as f:
        keys = _json.load(f)
    keys["hubert_soft"] = {k: list(v.shape) for k, v in HubertSoft().state_dict().items()}
    with open(os.path.join(OUT, "state_keys.json"), "w") as f:
        _json.dump(keys, f, indent=0, sort_keys=True)


PE_CASES = (  # (B, T, zero tail frames per clip, use_uv)
    (2, 50, (0, 7), False), (1, 300, (0,), False), (2, 20, (3, 0), False), (3, 33, (0, 5, 33), True))


def golden_pe(name="pe_24k", wseed=5):
    """PitchExtractor of the REAL reference (modules/fastspeech/pe.py) on a synthetic checkpoint (strict load, eval) at the 24 kHz
    demo shapes (80 mel bins, hidden 256): ragged padding tails, a clip shorter than 32 frames, an all-padding clip, and use_uv."""
    hp = dict(synth.HPARAMS_24K)
    refshim.set_hparams(hp)
    from modules.fastspeech.pe import PitchExtractor
    from utils.hparams import hparams as ref_hp
    sd = synth.pe_state(hp, wseed)
    m = PitchExtractor()
    m.load_state_dict(sd, strict=True)
    m.eval()
    out = {"wseed": wseed}
    for i, (B, T, tails, use_uv) in enumerate(PE_CASES):
        ref_hp["use_uv"] = use_uv
        mel = torch.from_numpy(synth.mel_like(40 + i, B, T, 80, tails))
        with torch.no_grad():
            r = m(mel)
        out["pitch_pred%d" % i] = r["pitch_pred"].numpy()
        out["f0_%d" % i] = r["f0_denorm_pred"].numpy()
        f = r["f0_denorm_pred"]
        print(name, (B, T, tails, use_uv), "f0 range %.1f..%.1f Hz, zeros %d" % (f[f > 0].min().item(), f.max().item(), int((f == 0).sum())))
    ref_hp["use_uv"] = hp["use_uv"]
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    import json as _json
    with open(os.path.join(OUT, "state_keys.json")) as f:
        keys = _json.load(f)
    keys["pitch_extractor"] = {k: list(v.shape) for k, v in PitchExtractor().state_dict().items()}
    with open(os.path.join(OUT, "state_keys.json"), "w") as f:
        _json.dump(keys, f, indent=0, sort_keys=True)


def golden_24k():
    """BASELINE configs[0] shapes: the 24 kHz demo architecture (training/config.yaml: M=80, C=256) with the full
    1000-step schedule and pndm_speedup=50 (20 PLMS iterations / 21 denoiser evaluations), plus one DDPM tail."""
    b = dict(synth.HPARAMS_24K)
    golden_diffnet("diffnet_24k", b, 2, B=2, T=37)
    golden_sampler("plms_24k_s50", b, 2, clips=[3], T=36, n_units=21, speedup=50, seed=82)
    golden_sampler("ddpm_24k_k30", dict(b, K_step=30), 2, clips=[6], T=36, n_units=21, speedup=1, seed=83)


if __name__ == "__main__":
    main()
