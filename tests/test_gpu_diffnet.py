"""GPU parity: DiffNet denoiser + DDPM / PLMS sampler (HIP, through the C ABI) against the oracle and
against the golden vectors minted from the real reference."""
import numpy as np
import pytest
import torch

from diffsvc_amd import synth
import dsvc_oracle as O
from util import golden_state, hp_for, load_golden, oracle_sample, clip_batch

pytestmark = pytest.mark.gpu

# max-abs tolerance on a single denoiser output (O(1) values) per operand precision
FWD_TOL = {"f16": 2e-2, "f16_w2": 6e-3, "f16_x3": 3e-4, "f16_x3t": 3e-4}


_CHAIN_CACHE = {}


def oracle_chain_1000():
    """The 1000-step oracle chain the full-chain tests share (44.1 kHz architecture, T=64): computed once per session."""
    if "r" not in _CHAIN_CACHE:
        hp = dict(synth.HPARAMS_44K)
        _CHAIN_CACHE["r"] = oracle_sample(hp, synth.acoustic_state(hp, 0), [0], 64, 37, 1, 2024, 1000)
    return _CHAIN_CACHE["r"]


def make_handles(hp, wseed, precision, sd=None):
    from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
    sd = sd if sd is not None else synth.acoustic_state(hp, wseed)
    den = DenoiserHandle(sd, hp["audio_num_mel_bins"], hp["hidden_size"], hp["residual_channels"],
                         hp["residual_layers"], hp["dilation_cycle_length"], hp["timesteps"],
                         precision=precision, prefix="denoise_fn.")
    return sd, den, SamplerHandle(den, sd)


@pytest.mark.parametrize("precision", ["f16_x3", "f16_x3t", "f16_w2", "f16"])
@pytest.mark.parametrize("name", ["diffnet_tiny", "diffnet_44k", "diffnet_24k"])
def test_denoiser_forward_vs_reference_golden(name, precision):
    g = load_golden(name)
    hp = hp_for(name)
    sd, den, _ = make_handles(hp, int(g["wseed"]), precision)
    out = den.forward(torch.from_numpy(g["spec"]).cuda(), torch.from_numpy(g["t"]).cuda(), torch.from_numpy(g["cond"]).cuda())
    err = (out.cpu() - torch.from_numpy(g["out"])).abs().max().item()
    assert err < FWD_TOL[precision], err


def test_denoiser_layer_taps_vs_oracle():
    """Per-layer activations (x after each residual block, gate output) against the oracle's taps."""
    hp = dict(synth.HPARAMS_44K)
    sd, den, _ = make_handles(hp, 0, "f16_x3")
    B, T, M, H, C = 1, 45, 128, 256, 384
    g = np.random.Generator(np.random.PCG64(5))
    spec = torch.from_numpy(g.standard_normal((B, 1, M, T)).astype(np.float32))
    cond = torch.from_numpy((g.standard_normal((B, H, T)) * 0.5).astype(np.float32))
    t = torch.tensor([731])
    taps = {}
    with torch.no_grad():
        ref = O.diffnet_forward(sd, spec, t, cond, 4, taps=taps)
    out = den.forward(spec.cuda(), t.cuda(), cond.cuda()).cpu()
    x_last = den.debug_buffer("xres")[:T].cpu()              # residual stream after the last layer, frame-major
    g_last = den.debug_buffer("g")[:T].cpu()
    skip = den.debug_buffer("skip")[:T].cpu()
    assert (x_last - taps["x19"][0].T).abs().max() < 2e-4
    assert (g_last - taps["g19"][0].T).abs().max() < 2e-4
    assert (skip / (20 ** 0.5) - taps["skip"][0].T).abs().max() < 2e-4
    assert (out - ref).abs().max() < 3e-4


def test_denoiser_batched_throughput_tiling_matches_oracle():
    """B=8 x T=861 (7168 frames) takes the 128x256 throughput tiling; per-clip steps differ."""
    hp = dict(synth.HPARAMS_44K)
    sd, den, _ = make_handles(hp, 0, "f16_x3")
    B, T = 8, 861
    g = np.random.Generator(np.random.PCG64(11))
    spec = torch.from_numpy(g.standard_normal((B, 1, 128, T)).astype(np.float32))
    cond = torch.from_numpy((g.standard_normal((B, 256, T)) * 0.5).astype(np.float32))
    t = torch.from_numpy(g.integers(0, 1000, size=(B,)))
    out = den.forward(spec.cuda(), t.cuda(), cond.cuda()).cpu()
    with torch.no_grad():
        for b in (0, 3, 7):
            ref = O.diffnet_forward(sd, spec[b:b + 1], t[b:b + 1], cond[b:b + 1], 4)
            assert (out[b:b + 1] - ref).abs().max() < 3e-4, b


@pytest.mark.parametrize("precision", ["f16_x3", "f16_x3t", "f16_w2", "f16_d64"])
@pytest.mark.parametrize("name", ["ddpm_tiny", "plmsc_tiny_s10", "plmsc_tiny_s5", "ddpm_44k_k20", "plmsc_44k_s20",
                                  "ddpm_24k_k30", "plmsc_24k_s50"])
def test_sampler_vs_reference_golden(name, precision):
    """mel within 1e-3 max-abs of the REAL reference (north_star) at every shipped precision, DDPM and PLMS/PNDM alike.  The
    PLMS goldens (plmsc_*) are minted on conditioned checkpoints whose noise prediction tracks its input like a trained model's
    (synth.acoustic_state_conditioned), so the unclamped PNDM chain contracts and the reference's own mel stays inside
    [spec_min, spec_max]: the 44.1 kHz architecture over the full 1000-step schedule at pndm_speedup=20 (BASELINE configs[2]),
    the 24 kHz demo architecture at pndm_speedup=50 (configs[0]).  f16_x3 runs on the conv_gemm engine, the others on tgemm
    (f16_x3t: the same fp32-class operand scheme as f16_x3 -- split activations, hi + lo weights, 3 MFMAs -- on the tgemm engine)."""
    g = load_golden(name)
    hp = dict(hp_for(name), K_step=int(g["K_step"]))
    if "tiny" in name:
        hp["timesteps"] = int(g["K_step"])
    tol = 1e-3
    if "ddpm" in name and not precision.startswith("f16_x3"):
        tol = 2e-3          # 20-30 coarse DDPM steps from t = K_step-1 of a short schedule amplify one fp16 rounding more than the
                            # 1000-step chain does (tests/test_gpu_headline.py holds THAT to 1e-3 at the benchmarked size)
    if "plms" in name and not precision.startswith("f16_x3"):
        # PLMS extrapolates from single evaluations (Adams-Bashforth weights 55/24, -59/24, ...): per-evaluation rounding is amplified,
        # the more the coarser the schedule.  The product picks its PLMS precision accordingly (DiffNetHip.precision_for):
        # f16_w2 up to pndm_speedup 20 -- held to 1e-3 on the 44.1 kHz architecture here and at T=861 in tests/test_gpu_headline.py --
        # and f16_x3 beyond; the one-MFMA f16_d64 is NOT a PLMS precision.  Measured bars for the other combinations:
        measured = {("plmsc_44k_s20", "f16_w2"): 1e-3, ("plmsc_tiny_s5", "f16_w2"): 1e-3, ("plmsc_tiny_s5", "f16_d64"): 1e-3,
                    ("plmsc_tiny_s10", "f16_w2"): 2e-3, ("plmsc_tiny_s10", "f16_d64"): 3e-3, ("plmsc_44k_s20", "f16_d64"): 3e-3,
                    ("plmsc_24k_s50", "f16_w2"): 1.5e-2, ("plmsc_24k_s50", "f16_d64"): 2.5e-2}
        tol = measured[(name, precision)]
    sd = golden_state(g, hp)
    _, den, smp = make_handles(hp, int(g["wseed"]), precision, sd=sd)
    clips = [int(c) for c in g["clips"]]
    hub, m2p, f0 = clip_batch(hp, clips, int(g["T"]), int(g["n_units"]))
    cond, _, _ = O.build_cond(sd, hub, m2p, f0.clone(), hp)
    if "decoder_inp" in g:
        assert np.array_equal(cond.numpy(), g["decoder_inp"])
    cond = cond.transpose(1, 2).contiguous().cuda()
    mels = []
    for i, c in enumerate(clips):            # one call per clip: Philox clip ids need not be contiguous
        mel = smp.sample(cond[i:i + 1], int(g["K_step"]), speedup=int(g["speedup"]), mel2ph=m2p[i:i + 1].cuda(),
                         seed=int(g["seed"]), first_clip=c, use_graph=False)
        mels.append(mel.cpu())
    err = (torch.cat(mels) - torch.from_numpy(g["mel_out"])).abs().max().item()
    print("sampler golden %s %s: mel max-abs err %.2e" % (name, precision, err))
    assert err < tol, err


def test_ddpm_graph_replay_equals_eager_and_oracle():
    hp = synth.tiny_hparams(K=50)
    sd, den, smp = make_handles(hp, 3, "f16_x3")
    r = oracle_sample(hp, sd, [0, 1, 2], 40, 23, 1, 123, 50)
    cond = r["cond_t"].cuda()
    m2p = r["mel2ph"].cuda()
    eager, xe = smp.sample(cond, 50, mel2ph=m2p, seed=123, first_clip=0, use_graph=False, return_x=True)
    graph, xg = smp.sample(cond, 50, mel2ph=m2p, seed=123, first_clip=0, use_graph=True, return_x=True)
    assert torch.equal(eager, graph)                      # same kernels, same order: bit-identical
    assert (xe.cpu() - r["x"]).abs().max() < 5e-4
    assert (eager.cpu() - r["mel_out"]).abs().max() < 1e-3


@pytest.mark.parametrize("precision", ["f16_x3", "f16_d16"])
def test_plms_graph_replay_equals_eager_and_oracle(precision):
    """PLMS with the per-iteration body replayed from a hipGraph (t and the history count live on the device):
    bit-identical to eager launches, and within the PLMS bar of the oracle."""
    hp = synth.tiny_hparams(K=100)
    sd, den, smp = make_handles(hp, 3, precision)
    r = oracle_sample(hp, sd, [0, 1], 40, 23, 5, 321, 100)
    cond, m2p = r["cond_t"].cuda(), r["mel2ph"].cuda()
    eager = smp.sample(cond, 100, speedup=5, mel2ph=m2p, seed=321, first_clip=0, use_graph=False)
    graph = smp.sample(cond, 100, speedup=5, mel2ph=m2p, seed=321, first_clip=0, use_graph=True)
    again = smp.sample(cond, 100, speedup=5, mel2ph=m2p, seed=321, first_clip=0, use_graph=True)     # cached graph
    assert torch.equal(eager, graph) and torch.equal(graph, again)
    assert (eager.cpu() - r["mel_out"]).abs().max().item() < (2e-3 if precision == "f16_x3" else 6e-3)


def test_batch_equals_per_clip():
    """Clips of a batch are independent: a batched call reproduces the B=1 calls bit for bit
    when the tiling is the same."""
    hp = synth.tiny_hparams(K=50)
    sd, den, smp = make_handles(hp, 3, "f16_w2")
    hub, m2p, f0 = clip_batch(hp, [0, 1, 2], 40, 23)
    cond, _, _ = O.build_cond(sd, hub, m2p, f0.clone(), hp)
    cond = cond.transpose(1, 2).contiguous().cuda()
    full = smp.sample(cond, 50, seed=9, first_clip=0, use_graph=False)
    for b in range(3):
        one = smp.sample(cond[b:b + 1], 50, seed=9, first_clip=b, use_graph=False)
        assert torch.equal(one[0], full[b]), b


def test_full_chain_1000_steps_w2_within_mel_bar():
    """The headline configuration's parity claim: 1000-step DDPM, 44.1 kHz architecture, default precision
    (f16_w2), mel within 1e-3 max-abs of the oracle -- at a frame count the oracle finishes in ~20 s."""
    hp = dict(synth.HPARAMS_44K)
    sd, den, smp = make_handles(hp, 0, "f16_w2")
    r = oracle_chain_1000()
    mel = smp.sample(r["cond_t"].cuda(), 1000, mel2ph=r["mel2ph"].cuda(), seed=2024, first_clip=0, use_graph=True)
    err = (mel.cpu() - r["mel_out"]).abs().max().item()
    assert err < 1e-3, err


def test_full_chain_1000_steps_dithered_f16_within_mel_bar():
    """One MFMA per product: fp16 weights with 64 time-dithered roundings (step t uses copy t % 64).  Same chain
    as above, same bar."""
    hp = dict(synth.HPARAMS_44K)
    sd, den, smp = make_handles(hp, 0, "f16_d64")
    r = oracle_chain_1000()
    mel = smp.sample(r["cond_t"].cuda(), 1000, mel2ph=r["mel2ph"].cuda(), seed=2024, first_clip=0, use_graph=True)
    err = (mel.cpu() - r["mel_out"]).abs().max().item()
    assert err < 1e-3, err


def test_plain_f16_fails_the_bar_dither_is_needed():
    """Documents WHY the dither exists: round-to-nearest fp16 weights alone miss the 1e-3 bar on a 1000-step chain
    (systematic rounding error), so 'f16' is not a shippable precision for the headline configuration."""
    hp = dict(synth.HPARAMS_44K)
    sd, den, smp = make_handles(hp, 0, "f16")
    r = oracle_chain_1000()
    mel = smp.sample(r["cond_t"].cuda(), 1000, mel2ph=r["mel2ph"].cuda(), seed=2024, first_clip=0, use_graph=True)
    err = (mel.cpu() - r["mel_out"]).abs().max().item()
    assert 1e-3 < err < 5e-2, err


@pytest.mark.parametrize("B,T", [(1, 1), (1, 5), (3, 31), (2, 33), (5, 64), (1, 97)])
def test_ragged_and_tiny_clip_lengths_vs_oracle(B, T):
    """Edge sizes: a single frame, lengths around the 32-frame tile, an odd batch -- 12-step DDPM on the tiny architecture
    against the oracle (gap rows between clips must behave as the convs' zero padding at every length)."""
    hp = synth.tiny_hparams(K=12)
    sd, den, smp = make_handles(hp, 4, "f16_x3")
    clips = list(range(B))
    n_units = max(1, T // 2)
    r = oracle_sample(hp, sd, clips, T, n_units, 1, 31, 12)
    _, m2p, _ = clip_batch(hp, clips, T, n_units)
    for graph in (False, True):
        mel = smp.sample(r["cond_t"].cuda(), 12, mel2ph=m2p.cuda(), seed=31, first_clip=0, use_graph=graph).cpu()
        assert mel.shape == r["mel_out"].shape
        err = (mel - r["mel_out"]).abs().max().item()
        assert err < 1e-3, (B, T, graph, err)


def test_sampler_rejects_bad_arguments():
    """Error behaviour of the C ABI: codes, never a crash."""
    hp = synth.tiny_hparams(K=12)
    sd, den, smp = make_handles(hp, 4, "f16_x3")
    cond = torch.zeros(1, hp["hidden_size"], 8, device="cuda")
    with pytest.raises(RuntimeError):
        smp.sample(cond, 13, seed=1)                    # more steps than the schedule has
    with pytest.raises(RuntimeError):
        smp.sample(cond, 0, seed=1)
    with pytest.raises(RuntimeError):
        smp.sample(cond.cpu(), 4, seed=1)               # host pointer
    with pytest.raises(RuntimeError):
        smp.sample(torch.zeros(1, hp["hidden_size"] + 1, 8, device="cuda"), 4, seed=1)


@pytest.mark.parametrize("precision,steps", [("f16_d64", 210), ("f16_d16", 70)])
def test_period_aligned_ddpm_graph_equals_eager(precision, steps):
    """The dithered DDPM graph is one period of the rounding schedule with every kernel's weight variant passed by value, replayed
    from a period-aligned step after an eager walk to the boundary; seed and clip id reach the captured kernels through device
    memory.  Same kernels, same variants, same noise as the eager loop: bit-identical, also for a second call with another seed
    and clip id on the SAME captured graph, and from an unaligned start."""
    hp = dict(synth.HPARAMS_44K)
    sd, den, smp = make_handles(hp, 0, precision)
    g = np.random.Generator(np.random.PCG64(3))
    cond = torch.from_numpy((g.standard_normal((2, 256, 40)) * 0.5).astype(np.float32)).cuda()
    for t_start, seed, clip0 in ((1000, 5, 0), (1000 - 7, 6, 3)):
        eager = smp.sample(cond, t_start, seed=seed, first_clip=clip0, t_stop=t_start - steps, use_graph=False, return_x=True)[1]
        graph = smp.sample(cond, t_start, seed=seed, first_clip=clip0, t_stop=t_start - steps, use_graph=True, return_x=True)[1]
        assert torch.isfinite(graph).all()
        assert torch.equal(eager, graph), (precision, t_start)
