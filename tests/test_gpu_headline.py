"""GPU parity for EXACTLY the configurations bench.py measures (BASELINE configs[1] and the per-GPU share of configs[3]):

  * one 10 s clip, T=861, 44.1 kHz architecture, the full 1000-step DDPM at the shipped precision, against a golden minted by
    running the REAL reference end to end at that size (oracle/make_golden.py::golden_headline -> e2e_44k_T861_k1000.npz);
  * the 128-frame throughput tiling of the tgemm engine (>= 6144 rows: what the batched number runs on): single evaluations at
    every tgemm precision, a short chain and the FULL 1000-step chain in a batch, per-layer taps;
  * the waveform bar end to end: reference-path PCM (reference sampler -> after_infer clip -> reference generator) against the
    HIP path's cond -> PCM, both from the same inputs (nothing of the HIP path is fed to the checker).
"""
import os

import numpy as np
import pytest
import torch

from diffsvc_amd import synth
import dsvc_oracle as O
from util import GOLD, clip_batch, golden_state, load_golden, oracle_sample

pytestmark = pytest.mark.gpu

MEL_BAR = 1e-3          # north_star: mel within 1e-3 max-abs
WAV_BAR = 1e-4          # north_star: waveform within 1e-4 RMS
# max-abs tolerance on ONE denoiser evaluation (O(1) outputs) per operand precision: a single evaluation carries the whole
# fp16 operand rounding; the chain tests above are the ones held to the mel bar
FWD_TOL = {"f16": 2e-2, "f16_d64": 2e-2, "f16_m64": 2e-2, "f16_w2": 6e-3}


def make_handles(hp, wseed, precision):
    from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
    sd = synth.acoustic_state(hp, wseed)
    den = DenoiserHandle(sd, hp["audio_num_mel_bins"], hp["hidden_size"], hp["residual_channels"],
                         hp["residual_layers"], hp["dilation_cycle_length"], hp["timesteps"],
                         precision=precision, prefix="denoise_fn.")
    return sd, den, SamplerHandle(den, sd)


_HEAD = {}


def headline():
    """golden + the regenerated inputs (the cond builder's index work is pinned bit for bit in tests/test_host.py)."""
    if not _HEAD:
        g = load_golden("e2e_44k_T861_k1000")
        hp = dict(synth.HPARAMS_44K, K_step=int(g["K_step"]))
        sd = synth.acoustic_state(hp, int(g["wseed"]))
        clips = [int(c) for c in g["clips"]]
        hub, m2p, f0 = clip_batch(hp, clips, int(g["T"]), int(g["n_units"]))
        cond, f0_denorm, pitch = O.build_cond(sd, hub, m2p, f0.clone(), hp)
        assert np.array_equal(pitch.numpy(), g["pitch"].astype(np.int64).reshape(pitch.shape)) and np.array_equal(f0_denorm.numpy(), g["f0_denorm"])
        _HEAD.update(g=g, hp=hp, sd=sd, clips=clips, hub=hub, m2p=m2p, f0=f0, cond_t=cond.transpose(1, 2).contiguous())
    return _HEAD


@pytest.mark.parametrize("precision", ["f16_x3t", "f16_m64", "f16_d64", "f16_w2"])
def test_headline_single_clip_T861_1000_steps_vs_reference(precision):
    """BENCH config: B=1, T=861, 1000 steps, graph replay -- mel within 1e-3 of the REAL reference, two clips / noise streams -- at the shipped
    single-clip precision (f16_x3t: DiffNetHip.AUTO['ddpm'], also held to 9.0e-4 on 21 goldens by test_spread_*[shipped]) and at the three
    fp16-activation schemes of rounds 1-3, which are MEASURED here, not shipped (f16_d64 -- every weight dithered -- passes on these two clips
    but not robustly: see the spread test)."""
    H = headline()
    g = H["g"]
    sd, den, smp = make_handles(H["hp"], int(g["wseed"]), precision)
    errs = []
    for i, c in enumerate(H["clips"]):
        mel = smp.sample(H["cond_t"][i:i + 1].cuda(), int(g["K_step"]), mel2ph=H["m2p"][i:i + 1].cuda(), seed=int(g["seed"]),
                         first_clip=c, use_graph=True)
        errs.append((mel[0].cpu() - torch.from_numpy(g["mel_out"][i])).abs().max().item())
    print("headline %s: mel max-abs err per clip %s" % (precision, ["%.2e" % e for e in errs]))
    assert max(errs) < MEL_BAR, errs


@pytest.mark.parametrize("precision", ["f16_m64", "f16_w2"])
def test_headline_worst_found_realisations_vs_reference(precision):
    """Two more single-clip runs of the REAL reference at the benchmarked configuration (e2e_44k_T861_k1000_c4 / _c6): the (clip,
    noise) pairs on which a spread study over eight realisations found the all-dithered f16_d64 at 1.02e-3 and 1.27e-3 -- over the
    bar (profiles/r2w_precision_spread.txt).  The shipped f16_m64 (dithered dilated conv, exact output 1x1) and f16_w2 stay inside."""
    from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
    hp = dict(synth.HPARAMS_44K)
    sd = synth.acoustic_state(hp, 0)
    den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=precision, prefix="denoise_fn.")
    smp = SamplerHandle(den, sd)
    errs = []
    for name in ("e2e_44k_T861_k1000_c4", "e2e_44k_T861_k1000_c6"):
        g = load_golden(name)
        clips = [int(c) for c in g["clips"]]
        hub, m2p, f0 = clip_batch(hp, clips, int(g["T"]), int(g["n_units"]))
        cond, f0_denorm, _ = O.build_cond(sd, hub, m2p, f0.clone(), hp)
        assert np.array_equal(f0_denorm.numpy(), g["f0_denorm"])
        mel = smp.sample(cond.transpose(1, 2).contiguous().cuda(), 1000, mel2ph=m2p.cuda(), seed=int(g["seed"]), first_clip=clips[0])
        errs.append((mel.cpu() - torch.from_numpy(g["mel_out"])).abs().max().item())
    print("headline extra %s: mel max-abs err (clip 4, clip 6) %s" % (precision, ["%.2e" % e for e in errs]))
    assert max(errs) < MEL_BAR, errs


@pytest.mark.parametrize("precision", ["f16_x3", "f16_x3t", "f16_d64", "f16_w2"])
def test_plms_50_iterations_T861_vs_reference(precision):
    """BASELINE configs[2] at the benchmarked size: one 10 s clip (T=861), 44.1 kHz architecture, the full 1000-step schedule at
    pndm_speedup=20 (50 PLMS iterations, 51 denoiser evaluations), the captured-graph path bench.py times -- mel within 1e-3 of the
    REAL reference on a conditioned checkpoint (plmsc_44k_T861_s20: the reference's own mel stays in [spec_min, spec_max])."""
    g = load_golden("plmsc_44k_T861_s20")
    hp = dict(synth.HPARAMS_44K, K_step=int(g["K_step"]))
    sd = golden_state(g, hp)
    from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
    den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=precision, prefix="denoise_fn.")
    smp = SamplerHandle(den, sd)
    clips = [int(c) for c in g["clips"]]
    hub, m2p, f0 = clip_batch(hp, clips, int(g["T"]), int(g["n_units"]))
    cond, f0_denorm, _ = O.build_cond(sd, hub, m2p, f0.clone(), hp)
    assert np.array_equal(f0_denorm.numpy(), g["f0_denorm"])
    cond = cond.transpose(1, 2).contiguous().cuda()
    errs = []
    for graph in (False, True, True):                       # eager, captured, replay of the cached graph
        mel = smp.sample(cond, int(g["K_step"]), speedup=int(g["speedup"]), mel2ph=m2p.cuda(), seed=int(g["seed"]), first_clip=clips[0],
                         use_graph=graph)
        errs.append((mel.cpu() - torch.from_numpy(g["mel_out"])).abs().max().item())
    print("PLMS-50 T=861 %s: mel max-abs err eager/graph/replay %s" % (precision, ["%.2e" % e for e in errs]))
    assert errs[0] == errs[1] == errs[2]                    # the captured graph is the eager loop
    if precision == "f16_d64":
        # documents WHY the drop-in does not run PLMS on dithered single-plane weights (DiffNetHip.precision_for): fine over a
        # 1000-step DDPM chain, well over the bar when 51 evaluations are extrapolated
        assert MEL_BAR < max(errs) < 1e-2, errs
    elif precision.startswith("f16_x3"):
        assert max(errs) < 5e-5, errs                       # the shipped PLMS precision: fp32-class (x3: conv_gemm engine, x3t: tgemm)
    else:
        assert max(errs) < MEL_BAR, errs                    # f16_w2 passes on THIS pair (7.7e-4); over ten pairs it is (8.2 +- 1.2)e-4
                                                            # with one at 1.08e-3 (profiles/r2w_precision_spread.txt): not shipped


def test_plms_50_iterations_T861_six_goldens_at_the_shipped_precision():
    """BASELINE configs[2] at the benchmarked size rested on ONE (clip, noise) pair.  Five more runs of the REAL reference
    (oracle/make_golden.py::golden_plms_t861_more: 51 evaluations each, same conditioned checkpoint, clips 1, 2, 3, 4, 7 with their own noise)
    beside the first; the shipped PLMS precision (DiffNetHip.AUTO['plms'] = f16_x3t), the captured-graph path bench.py times, ONE sampler for
    all six (the chain's graph is captured once): every one within 1e-4 of the reference."""
    from diffsvc_amd.denoiser import DiffNetHip
    from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
    from make_golden import PLMS_T861_MORE
    hp = dict(synth.HPARAMS_44K)
    names = ["plmsc_44k_T861_s20"] + ["plmsc_44k_T861_s20_c%d" % c for c, _ in PLMS_T861_MORE]
    precision = DiffNetHip.AUTO["plms"]
    smp, sd, errs = None, None, []
    for name in names:
        g = load_golden(name)
        if smp is None:
            sd = golden_state(g, hp)
            den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=precision, prefix="denoise_fn.")
            smp = SamplerHandle(den, sd)
        clips = [int(c) for c in g["clips"]]
        hub, m2p, f0 = clip_batch(hp, clips, int(g["T"]), int(g["n_units"]))
        cond, f0_denorm, _ = O.build_cond(sd, hub, m2p, f0.clone(), hp)
        assert np.array_equal(f0_denorm.numpy(), g["f0_denorm"])
        mel = smp.sample(cond.transpose(1, 2).contiguous().cuda(), int(g["K_step"]), speedup=int(g["speedup"]), mel2ph=m2p.cuda(),
                         seed=int(g["seed"]), first_clip=clips[0], use_graph=True)
        errs.append((clips[0], (mel.cpu() - torch.from_numpy(g["mel_out"])).abs().max().item()))
    print("PLMS-50 T=861 %s, six (clip, noise) pairs: mel max-abs err %s" % (precision, ["%d: %.2e" % e for e in errs]))
    assert smp.stats()["capture_plms"] == 1
    assert max(e[1] for e in errs) < 1e-4, errs


@pytest.mark.parametrize("precision", ["f16_m64", "f16_d64"])
def test_throughput_tiling_full_chain_vs_reference(precision):
    """The batched number's kernels over the FULL chain: 8 clips x T=861 (7168 rows -> the 128-frame tgemm tiling with the
    non-temporal residual/skip stream), 1000 steps; clips 0 and 1 of the batch are the reference golden's clips."""
    H = headline()
    g = H["g"]
    hp = H["hp"]
    sd, den, smp = make_handles(hp, int(g["wseed"]), precision)
    clips = list(range(8))
    hub, m2p, f0 = clip_batch(hp, clips, int(g["T"]), int(g["n_units"]))
    cond, _, _ = O.build_cond(sd, hub, m2p, f0.clone(), hp)
    mel = smp.sample(cond.transpose(1, 2).contiguous().cuda(), int(g["K_step"]), mel2ph=m2p.cuda(), seed=int(g["seed"]),
                     first_clip=0, use_graph=True).cpu()
    assert H["clips"] == [0, 1]
    errs = [(mel[i] - torch.from_numpy(g["mel_out"][i])).abs().max().item() for i in range(2)]
    print("throughput tiling %s, 1000 steps: mel max-abs err %s" % (precision, ["%.2e" % e for e in errs]))
    assert torch.isfinite(mel).all()
    assert max(errs) < MEL_BAR, errs


@pytest.mark.parametrize("precision", ["f16", "f16_w2", "f16_d64", "f16_m64"])
def test_throughput_tiling_forward_vs_oracle(precision):
    """dsvc_denoiser_forward at B=8 x T=861 on the tgemm engine (tgemm_kernel<4,8,...>), per-clip steps differ."""
    hp = dict(synth.HPARAMS_44K)
    sd, den, _ = make_handles(hp, 0, precision)
    B, T = 8, 861
    g = np.random.Generator(np.random.PCG64(11))
    spec = torch.from_numpy(g.standard_normal((B, 1, 128, T)).astype(np.float32))
    cond = torch.from_numpy((g.standard_normal((B, 256, T)) * 0.5).astype(np.float32))
    t = torch.from_numpy(g.integers(0, 1000, size=(B,)))
    out = den.forward(spec.cuda(), t.cuda(), cond.cuda()).cpu()
    errs = []
    with torch.no_grad():
        for b in (0, 3, 7):
            ref = O.diffnet_forward(sd, spec[b:b + 1], t[b:b + 1], cond[b:b + 1], 4)
            errs.append((out[b:b + 1] - ref).abs().max().item())
    print("throughput tiling forward %s: max-abs err %s" % (precision, ["%.2e" % e for e in errs]))
    assert max(errs) < FWD_TOL[precision], errs


@pytest.mark.parametrize("precision", ["f16_w2", "f16_d64", "f16_m64"])
def test_throughput_tiling_ddpm_20_steps_vs_oracle(precision):
    """20-step DDPM at B=8 x T=861 on the throughput tiling; two clips of the batch against B=1 oracle chains."""
    hp = dict(synth.HPARAMS_44K, K_step=20)
    sd, den, smp = make_handles(hp, 0, precision)
    clips, T, n_units, seed = list(range(8)), 861, 500, 41
    hub, m2p, f0 = clip_batch(hp, clips, T, n_units)
    cond, _, _ = O.build_cond(sd, hub, m2p, f0.clone(), hp)
    mel = smp.sample(cond.transpose(1, 2).contiguous().cuda(), 20, mel2ph=m2p.cuda(), seed=seed, first_clip=0, use_graph=False).cpu()
    errs = []
    for c in (2, 7):
        r = oracle_sample(hp, sd, [c], T, n_units, 1, seed, 20)
        errs.append((mel[c] - r["mel_out"][0]).abs().max().item())
    print("throughput tiling ddpm-20 %s: mel max-abs err %s" % (precision, ["%.2e" % e for e in errs]))
    assert max(errs) < MEL_BAR, errs


def _tap_errors(den, sd, spec, t, cond, rows_of):
    """Run the first n layers for n = 1..L (dsvc_denoiser_debug_set 'stop_after_layers') and compare the residual stream, the gate output and
    the running skip sum with the oracle's taps.  rows_of(b) -> slice of the frame-major debug buffers holding clip b."""
    L = O.diffnet_layers(sd)
    taps = {}
    with torch.no_grad():
        O.diffnet_forward(sd, spec, t, cond, 4, taps=taps)
    worst = {"x": (0.0, -1), "g": (0.0, -1), "s": (0.0, -1)}
    try:
        for n in range(1, L + 1):
            den.debug_set("stop_after_layers", n)
            den.forward(spec.cuda(), t.cuda(), cond.cuda())
            bufs = {"x": den.debug_buffer("xres").cpu(), "g": den.debug_buffer("g").cpu(), "s": den.debug_buffer("skip").cpu()}
            for k in bufs:
                for b in range(spec.shape[0]):
                    ref = taps["%s%d" % (k, n - 1)][b].T
                    e = (bufs[k][rows_of(b)] - ref).abs().max().item()
                    if e > worst[k][0]:
                        worst[k] = (e, n - 1)
    finally:
        den.debug_set("stop_after_layers", -1)
    return worst


@pytest.mark.parametrize("precision,B,T,fused", [("f16_w2", 1, 45, True), ("f16_d64", 1, 45, True), ("f16_d64", 2, 861, True),
                                                 ("f16_w2", 8, 861, True), ("f16_d64", 8, 861, True), ("f16_d64", 8, 861, False),
                                                 ("f16_m64", 1, 45, True), ("f16_m64", 8, 861, True),
                                                 ("f16_x3t", 1, 430, True), ("f16_x3t", 1, 861, True), ("f16_x3t", 1, 1100, True), ("f16_x3t", 1, 1600, True), ("f16_x3t", 2, 861, True),
                                                 ("f16_x3t", 1, 2100, True)])
def test_tgemm_engine_layer_taps_vs_oracle(precision, B, T, fused, hooks):
    """Per-layer localisation on the PRODUCT engine (tgemm): after every residual block the residual stream x_l, the gate
    output g_l and the running skip sum are compared with the oracle -- for the split-K single-clip tiling (T=45 and T=861), the
    small-batch tiling (B=2) and the 128-frame throughput tiling (B=8 x 861).  The throughput tiling runs a layer as ONE fused
    kernel whose gate output never leaves the CU (tlayer.h), so g is tapped on its two-launch form (fused=False); x and the skip
    sum are tapped on both.  The shipped single-clip precision f16_x3t (split operands, fp32-class) on each of its split-K tilings: 3 output tiles x 4 / 3
    K slices (T = 861: up to 32 frame tiles), 2 x 4 / 3 (T = 430: up to 21), 4 x 3 (T = 1100: up to 42), 6 x 2 (T = 1600 and two clips of 861: up to 64), and the plain tiling beyond."""
    hp = dict(synth.HPARAMS_44K)
    sd, den, _ = make_handles(hp, 0, precision)
    g = np.random.Generator(np.random.PCG64(5 + B))
    spec = torch.from_numpy(g.standard_normal((B, 1, 128, T)).astype(np.float32))
    cond = torch.from_numpy((g.standard_normal((B, 256, T)) * 0.5).astype(np.float32))
    t = torch.from_numpy(g.integers(0, 1000, size=(B,)))
    Tp = (T + 8 + 127) // 128 * 128          # csrc/diffnet.hip: bucket_rows
    den.debug_set("two_launch_layer", -1 if fused else 1)        # -1: the fused kernel wherever it is supported (auto takes it from ~18 clips)
    worst = _tap_errors(den, sd, spec, t, cond, lambda b: slice(b * Tp, b * Tp + T))
    print("tgemm taps %s B=%d T=%d fused=%d: worst |err| x %.2e (layer %d), g %.2e (layer %d), skip-sum %.2e (layer %d)"
          % (precision, B, T, fused, worst["x"][0], worst["x"][1], worst["g"][0], worst["g"][1], worst["s"][0], worst["s"][1]))
    # fp16 activations: one rounding of an O(1..4) value is 2^-11 relative; 20 layers of it stay well under these bars, a
    # mis-indexed tile or a wrong dilation does not
    tol = 1.5e-2 if precision != "f16_w2" else 1e-2
    if precision == "f16_x3t":
        tol = 2e-4                                           # (hi + lo weights, hi | lo activations: fp32-class; measured ~2e-5 after 20 layers)
    g_live = not (fused and B * Tp >= 6144)                  # the fused kernel keeps g in LDS: the debug buffer is not written
    g_tol = 1e-3 if precision == "f16_x3t" else tol          # (the tap reads the hi plane of g: one fp16 rounding of an O(1) value)
    assert worst["x"][0] < tol and (worst["g"][0] < g_tol or not g_live) and worst["s"][0] < 4 * tol, worst


@pytest.mark.parametrize("precision", ["f16_m64", "f16_d64", "f16_w2", "f16_x3", "f16_x3t"])
def test_end_to_end_waveform_vs_reference(precision):
    """cond -> 1000-step DDPM -> clip -> NSF-HiFiGAN through the HIP path against the REAL reference's PCM for the same inputs and
    noise streams (golden wav0: reference sampler -> after_infer clip -> reference generator).  The mel the vocoder sees is the
    HIP path's own -- this is the end-to-end number, not the vocoder in isolation."""
    from diffsvc_amd.pipeline import SvcPipeline
    H = headline()
    g = H["g"]
    hp = H["hp"]
    h = dict(synth.VOCODER_44K)
    pipe = SvcPipeline(hp, H["sd"], synth.vocoder_state(h, int(g["vseed"])), h, precision=precision, vocoder_precision="f16_x3")
    wav, mel = pipe.infer(H["hub"][:1].cuda(), H["m2p"][:1].cuda(), H["f0"][:1].cuda(), speedup=1, seed=int(g["seed"]),
                          first_clip=H["clips"][0], return_mel=True)
    ref = torch.from_numpy(g["wav0"])
    mel_err = (mel[0].cpu() - torch.from_numpy(g["mel_out"][0])).abs().max().item()
    rms = (wav[0].cpu() - ref).pow(2).mean().sqrt().item()
    print("end to end %s: mel max-abs err %.2e, wav RMS err %.2e (reference wav RMS %.3f)" % (precision, mel_err, rms, ref.pow(2).mean().sqrt().item()))
    assert wav.shape[1] == ref.shape[0]
    assert mel_err < MEL_BAR, mel_err
    assert rms < WAV_BAR, rms


@pytest.mark.parametrize("precision", ["f16_m64", "f16_d64", "f16_w2"])
def test_fused_layer_kernel_equals_the_two_launch_layer_bit_for_bit(precision, hooks):
    """The throughput tiling runs a residual layer as ONE kernel (tlayer.h: gate GEMM -> g in LDS -> output projection).  It issues
    the same MFMAs on the same operands in the same order as the two tgemm launches it replaces (debug_set two_launch_layer), so a
    20-step DDPM chain at B=8 x T=861 must come out bit-identical -- layer geometry, g hand-off through LDS, the alternating xh
    buffers and the pass rotation are all covered by one equality."""
    hp = dict(synth.HPARAMS_44K, K_step=20)
    sd, den, smp = make_handles(hp, 0, precision)
    clips, T, n_units, seed = list(range(8)), 861, 500, 77
    hub, m2p, f0 = clip_batch(hp, clips, T, n_units)
    cond, _, _ = O.build_cond(sd, hub, m2p, f0.clone(), hp)
    cond = cond.transpose(1, 2).contiguous().cuda()
    den.debug_set("two_launch_layer", -1)                        # at 8 clips the automatic choice is the two launches (faster there)
    fused = smp.sample(cond, 20, mel2ph=m2p.cuda(), seed=seed, first_clip=0, use_graph=False, return_x=True)[1]
    den.debug_set("two_launch_layer", 1)
    two = smp.sample(cond, 20, mel2ph=m2p.cuda(), seed=seed, first_clip=0, use_graph=False, return_x=True)[1]
    assert torch.isfinite(fused).all()
    d = (fused - two).abs().max().item()
    print("fused vs two-launch layer (%s): max |diff| %.3e" % (precision, d))
    assert torch.equal(fused, two), d


# ---- round 3: the 1000-step error is a heavy-tailed statistic, so the shipped precision is held to the bar on MANY realisations ----
SHIP_BAR = 9.0e-4       # the shipped DDPM precision must keep >= 10 % margin under the 1e-3 bar on every real-reference golden


def _spread():
    """(name, clip, seed, conditioned) of every single-clip real-reference golden of the benchmarked configuration."""
    from make_golden import spread_names
    out = [("e2e_44k_T861_k1000_c4", 4, 1004, None), ("e2e_44k_T861_k1000_c6", 6, 1006, None)]
    return out + list(spread_names())


def _shipped(batched=False):
    """What precision='auto' runs a DDPM call at: by size (DiffNetHip.precision_for)."""
    from diffsvc_amd.denoiser import DiffNetHip
    return DiffNetHip.AUTO["ddpm_batched" if batched else "ddpm"]


def _clip_cond(hp, sd, clip, g):
    hub, m2p, f0 = clip_batch(hp, [clip], int(g["T"]), int(g["n_units"]))
    cond, f0_denorm, _ = O.build_cond(sd, hub, m2p, f0.clone(), hp)
    assert np.array_equal(f0_denorm.numpy(), g["f0_denorm"])
    return cond.transpose(1, 2).contiguous().cuda(), m2p.cuda()


@pytest.mark.parametrize("which", ["shipped", "f16_m64", "f16_w2"])
def test_spread_of_single_clip_chains_vs_reference(which):
    """B=1, T=861, 1000 steps against NINETEEN further runs of the REAL reference (oracle/make_golden.py::golden_headline_spread + the two
    round-2 extras): twelve (clip, noise) pairs on the random-init checkpoint -- among them (9, 1009), the round-2 precision's worst --
    and six on CONDITIONED checkpoints whose reference mel has < 1 % of its values on spec_min / spec_max (p_sample's clamp,
    diffusion.py:149-150, cannot hide an error there; the random-init goldens have 36 % of their values on it).  The shipped DDPM
    precision (DiffNetHip.AUTO['ddpm'] -- since round 3 the fp32-class f16_x3t for calls of this size) must stay <= 9.0e-4 on EVERY one; the
    other schemes are measured beside it (f16_w2, the batched precision, is held to the 1e-3 bar here too)."""
    precision = _shipped() if which == "shipped" else which
    if which != "shipped" and precision == _shipped():
        pytest.skip("is the shipped precision")
    from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
    hp = dict(synth.HPARAMS_44K)
    handles, errs = {}, []
    for name, clip, seed, cond_par in _spread():
        g = load_golden(name)
        assert int(g["seed"]) == seed and [int(c) for c in g["clips"]] == [clip]
        key = tuple(cond_par) if cond_par else None
        if key not in handles:
            handles.clear()                                  # one packed weight set at a time (2.3 GB with 64 dither variants)
            sd = golden_state(g, hp)
            den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=precision, prefix="denoise_fn.")
            handles[key] = (sd, SamplerHandle(den, sd))
        sd, smp = handles[key]
        cond, m2p = _clip_cond(hp, sd, clip, g)
        mel = smp.sample(cond, 1000, mel2ph=m2p, seed=seed, first_clip=clip, use_graph=True).cpu()
        d = (mel - torch.from_numpy(g["mel_out"])).abs()
        errs.append((name.replace("e2e_44k_T861_k1000_", ""), d.max().item(), d.pow(2).mean().sqrt().item()))
    worst = max(e[1] for e in errs)
    print("spread %s (%s): worst %.2e | " % (which, precision, worst) + "  ".join("%s %.2e/%.1e" % e for e in errs))
    assert all(np.isfinite(e[1]) for e in errs)
    if which == "shipped":
        assert worst <= SHIP_BAR, errs
    elif precision == "f16_w2":
        assert worst < MEL_BAR, errs


def gumbel_fit(maxima):
    """Method-of-moments Gumbel fit of per-clip maximum errors: (mu, beta, P(one clip > 1e-3), P(a 256-clip job holds a clip > 1e-3))."""
    import math
    v = np.asarray(maxima, dtype=np.float64)
    beta = v.std(ddof=1) * math.sqrt(6.0) / math.pi
    mu = v.mean() - 0.5772156649 * beta
    p1 = 1.0 - math.exp(-math.exp(-(MEL_BAR - mu) / beta))
    return mu, beta, p1, 1.0 - (1.0 - p1) ** 256


@pytest.mark.parametrize("which", ["shipped", "f16_m64", "f16_w2", "f16_w6", "f16_x3t"])
@pytest.mark.parametrize("ckpt", ["random", "random2", "ca", "cb"])
def test_batch_of_32_full_chain_every_clip_with_a_golden(ckpt, which):
    """The per-GPU share of BASELINE configs[3] -- what bench.py's `batched.value` times: 32 clips x T=861 in ONE batch (28 672 rows,
    the fused layer kernel tlayer_kernel<3, .>), 1000 steps, at the shipped precision, seed 2026.  Round 4: EVERY clip of TWO shares has a
    real-reference golden -- "random" = clips 0..31 (rank 0's share of the 256-clip job), "random2" = clips 32..63 (rank 1's) on the random-init
    checkpoint, 32 of 32 checked each -- plus 2 + 4 clips on the two conditioned checkpoints.  The shipped precision must stay <= 9.0e-4 on all of
    them; the Gumbel fit of the per-clip maxima (the numbers bench.py's `batched.p_over_bar_per_256_clips` quotes) is printed.  The other operand
    schemes are measured beside the shipped one."""
    precision = _shipped(batched=True) if which == "shipped" else which
    if which != "shipped" and precision == _shipped(batched=True):
        pytest.skip("is the shipped precision")
    from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
    from make_golden import SPREAD_COND, SPREAD_SEED, share_golden
    hp = dict(synth.HPARAMS_44K)
    random_init = ckpt.startswith("random")
    first = 32 if ckpt == "random2" else 0
    cond_par = None if random_init else dict((t, c) for t, c, _ in SPREAD_COND)[ckpt]
    if random_init:
        have = {c: share_golden(c) for c in range(first, first + 32)}
        if os.environ.get("DSVC_PARTIAL_GOLDENS") == "1":   # (while oracle/make_golden.py --headline-spread is still minting: check what exists)
            have = {c: v for c, v in have.items() if os.path.exists(os.path.join(GOLD, v[0] + ".npz"))}
    else:
        have = {c: ("e2e_44k_T861_k1000_%s_c%d" % (ckpt, c), 0) for c in dict((t, cl) for t, _, cl in SPREAD_COND)[ckpt]}
    sd = synth.acoustic_state_conditioned(hp, 0, *cond_par) if cond_par else synth.acoustic_state(hp, 0)
    den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=precision, prefix="denoise_fn.")
    smp = SamplerHandle(den, sd)
    clips = list(range(first, first + 32))
    hub, m2p, f0 = clip_batch(hp, clips, 861, 500)
    cond, _, _ = O.build_cond(sd, hub, m2p, f0.clone(), hp)
    mel = smp.sample(cond.transpose(1, 2).contiguous().cuda(), 1000, mel2ph=m2p.cuda(), seed=SPREAD_SEED, first_clip=first, use_graph=True).cpu()
    assert torch.isfinite(mel).all()
    errs = []
    for c, (name, row) in sorted(have.items()):
        g = load_golden(name)
        assert int(g["seed"]) == SPREAD_SEED and int(g["clips"][row]) == c
        errs.append((c, (mel[c - first] - torch.from_numpy(g["mel_out"][row])).abs().max().item()))
    if random_init and os.environ.get("DSVC_PARTIAL_GOLDENS") != "1":
        assert len(errs) == 32                              # every clip of the share
    worst = max(e[1] for e in errs)
    print("batch of 32 (%s checkpoint, %s %s): worst %.2e; mel max-abs err per golden clip %s" % (ckpt, which, precision, worst, ["%d: %.2e" % e for e in errs]))
    if random_init and precision != "f16_x3t":
        mu, beta, p1, p256 = gumbel_fit([e[1] for e in errs])
        print("batch of 32 (%s, %s %s): Gumbel fit of the 32 per-clip maxima mu %.3e beta %.3e -> P(clip > 1e-3) %.2e, P(a 256-clip job holds one) %.2f"
              % (ckpt, which, precision, mu, beta, p1, p256))
    if which == "shipped":
        assert worst <= SHIP_BAR, errs
    elif precision in ("f16_w2", "f16_w6"):
        assert worst < MEL_BAR, errs
    elif precision == "f16_x3t":          # bench.py's `fp32_class.batched_value`: the 64-frame two-launch tiling of the split-activation scheme
        assert worst < 1e-4, errs


@pytest.mark.parametrize("B", [8, 12, 16])
def test_mid_batch_auto_every_clip_with_a_golden(B):
    """what precision='auto' REALLY runs between the batched threshold (7 ten-second clips) and the 128-frame tiling's
    minimum (18 clips).  Until round 4 that was f16_w2 on the two-launch tilings (the handle fell back silently) -- the scheme that failed
    the 9.0e-4 ship bar on the 64-golden study.  Round 5: the f16_w6 handle runs the fused layer kernel on 64- / 32-frame tiles there
    (csrc/diffnet.hip: fused_nt), 6-bit correction products and gate-output correction included.  8, 12 and 16 clips x T=861 in ONE batch,
    1000 steps, the precision DiffNetHip.precision_for picks, NO debug knob; every clip has a real-reference golden (clips 0..15 of rank 0's
    share) and must stay <= 9.0e-4; the kernel that ran is asserted to be the fused one."""
    from diffsvc_amd.denoiser import DiffNetHip
    from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
    from make_golden import SPREAD_SEED, share_golden
    hp = dict(synth.HPARAMS_44K)
    net = DiffNetHip(128, hparams=hp)
    precision = net.precision_for("ddpm", 1, frames=B * 861, clips=B)
    assert precision == _shipped(batched=True), precision
    sd = synth.acoustic_state(hp, 0)
    den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=precision, prefix="denoise_fn.")
    smp = SamplerHandle(den, sd)
    clips = list(range(B))
    hub, m2p, f0 = clip_batch(hp, clips, 861, 500)
    cond, _, _ = O.build_cond(sd, hub, m2p, f0.clone(), hp)
    mel = smp.sample(cond.transpose(1, 2).contiguous().cuda(), 1000, mel2ph=m2p.cuda(), seed=SPREAD_SEED, first_clip=0, use_graph=True).cpu()
    assert torch.isfinite(mel).all()
    errs = []
    for c in clips:
        name, row = share_golden(c)
        g = load_golden(name)
        assert int(g["seed"]) == SPREAD_SEED and int(g["clips"][row]) == c
        errs.append((c, (mel[c] - torch.from_numpy(g["mel_out"][row])).abs().max().item()))
    worst = max(e[1] for e in errs)
    nt = smp.profile_gate_kernel(B, 861, 1)[2]
    mu, beta, p1, p256 = gumbel_fit([e[1] for e in errs])
    print("mid batch of %d (auto -> %s, fused layer kernel on %d-frame tiles): worst %.2e; per clip %s" % (B, precision, 32 * nt, worst, ["%d: %.2e" % e for e in errs]))
    print("mid batch of %d: Gumbel fit of the %d per-clip maxima mu %.3e beta %.3e -> P(clip > 1e-3) %.2e" % (B, B, mu, beta, p1))
    assert nt in (1, 2), nt                              # the fused kernel's mid-size tilings, not the two-launch fallback (0)
    assert worst <= SHIP_BAR, errs


def test_f16_w6_at_dilation_16_runs_the_fused_kernel_on_the_tiles_that_fit():
    """with `dilation_cycle_length: 5` the largest dilation is 16, and the fused layer kernel's LDS plan on 128-frame tiles
    (time tile + g block + the 6-bit g_lo code block: 172 032 B) exceeds the CU's 160 KB.  Round 5 then failed inside launch_fused_layer although
    `auto` had picked f16_w6 for the call; `fused_nt()` now counts the code block when it chooses the tile width.  The 44.1 kHz architecture with
    a 5-cycle, 20 clips x T = 861 (140 tiles: where the 128-frame tiling would be the choice), 20 DDPM steps at the precision `auto` picks: runs,
    on 64- or 32-frame tiles, two clips within the coarse-chain bar of the oracle."""
    from diffsvc_amd.denoiser import DiffNetHip
    hp = dict(synth.HPARAMS_44K, dilation_cycle_length=5, K_step=20)
    B, T, n_units, seed = 20, 861, 500, 43
    precision = DiffNetHip(128, hparams=hp).precision_for("ddpm", 1, frames=B * T, clips=B)
    assert precision == "f16_w6"
    sd, den, smp = make_handles(hp, 0, precision)
    clips = list(range(B))
    hub, m2p, f0 = clip_batch(hp, clips, T, n_units)
    cond, _, _ = O.build_cond(sd, hub, m2p, f0.clone(), hp)
    mel = smp.sample(cond.transpose(1, 2).contiguous().cuda(), 20, mel2ph=m2p.cuda(), seed=seed, first_clip=0, use_graph=False).cpu()
    nt = smp.profile_gate_kernel(B, T, 1)[2]
    errs = []
    for c in (3, 17):
        r = oracle_sample(hp, sd, [c], T, n_units, 1, seed, 20)
        errs.append((mel[c] - r["mel_out"][0]).abs().max().item())
    print("f16_w6 at dilation 16 (cycle 5), 20 clips: fused kernel on %d-frame tiles, 20-step mel max-abs err %s" % (32 * nt, ["%.2e" % e for e in errs]))
    assert nt in (1, 2), nt
    assert torch.isfinite(mel).all() and max(errs) < 2e-3, errs


@pytest.mark.parametrize("tag", ["w1", "w2", "w3"])
def test_batched_precision_across_weight_statistics(tag):
    """the shipped batched precision (f16_w6) measures 4.0e-4 ... 6.1e-4 on the random-init checkpoint (1.65x of margin) and
    1.3e-4 ... 1.6e-4 on the two conditioned ones -- "a trained model is probably far safer" was the state.  Three more checkpoints on the line
    between them (oracle/make_golden.py::WSTAT_CKPTS: eps = lam x + rho * random network with (lam, rho) = (1.0, 0.3), (0.6, 0.55), (0.25, 0.8);
    `ca` is (1.5, 0.07), random-init (0, 1)), eight clips each through the REAL reference (1000 steps, T = 861, seed 2026), run here as ONE batch
    of 8 at the precision `auto` picks (f16_w6 on the fused kernel's 32-frame tiles).  Every clip <= the 9.0e-4 ship bar; the printed line
    (worst error, Gumbel fit, fraction of the reference's own mel on p_sample's clamp) is the data point of design/precision.md's table."""
    from diffsvc_amd.denoiser import DiffNetHip
    from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
    from make_golden import SPREAD_SEED, WSTAT_CKPTS, WSTAT_CLIPS
    par = dict(WSTAT_CKPTS)[tag]
    hp = dict(synth.HPARAMS_44K)
    clips = list(WSTAT_CLIPS)
    precision = DiffNetHip(128, hparams=hp).precision_for("ddpm", 1, frames=len(clips) * 861, clips=len(clips))
    assert precision == _shipped(batched=True), precision
    sd = synth.acoustic_state_conditioned(hp, 0, *par)
    den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=precision, prefix="denoise_fn.")
    smp = SamplerHandle(den, sd)
    hub, m2p, f0 = clip_batch(hp, clips, 861, 500)
    cond, _, _ = O.build_cond(sd, hub, m2p, f0.clone(), hp)
    mel = smp.sample(cond.transpose(1, 2).contiguous().cuda(), 1000, mel2ph=m2p.cuda(), seed=SPREAD_SEED, first_clip=0, use_graph=True).cpu()
    errs, clamp = [], []
    for c in clips:
        g = load_golden("e2e_44k_T861_k1000_%s_c%d" % (tag, c))
        assert int(g["seed"]) == SPREAD_SEED and [int(v) for v in g["clips"]] == [c] and tuple(float(v) for v in g["conditioned"]) == par
        errs.append((mel[c] - torch.from_numpy(g["mel_out"][0])).abs().max().item())
        clamp.append(float(g["on_clamp"]))
    mu, beta, p1, p256 = gumbel_fit(errs)
    print("weight statistics %s (lam %.2f, rho %.2f; %.1f %% of the reference mel on the clamp) at %s, 8 clips: worst %.2e, best %.2e, Gumbel mu %.2e beta %.1e "
          "-> P(clip > 1e-3) %.1e" % (tag, par[0], par[1], 100 * float(np.mean(clamp)), precision, max(errs), min(errs), mu, beta, p1))
    assert max(errs) <= SHIP_BAR, errs


@pytest.mark.parametrize("precision", ["f16_w6", "f16_w6n"])
def test_mid_size_tilings_of_the_fused_layer_kernel_equal_the_128_frame_tiling_bit_for_bit(precision, hooks):
    """The fused layer kernel on 64- and 32-frame tiles (round 5: tlayer_kernel<..., NT = 2 / 1>) issues the same products in the same order
    for every accumulator as on 128-frame tiles -- a frame's K loops do not depend on which workgroup holds it -- so a 20-step DDPM chain at
    8 x T=861 comes out bit-identical on all three (tile DMA and halo, g blocks and 6-bit code rows in LDS scaled by NT / 4, the output
    phase's prefetched accumulator sets that only the small tiles have registers for)."""
    hp = dict(synth.HPARAMS_44K, K_step=20)
    sd, den, smp = make_handles(hp, 0, precision)
    clips, T, n_units, seed = list(range(8)), 861, 500, 77
    hub, m2p, f0 = clip_batch(hp, clips, T, n_units)
    cond, _, _ = O.build_cond(sd, hub, m2p, f0.clone(), hp)
    cond = cond.transpose(1, 2).contiguous().cuda()
    out = {}
    for nt in (4, 2, 1):
        den.debug_set("fused_nt", nt)
        out[nt] = smp.sample(cond, 20, mel2ph=m2p.cuda(), seed=seed, first_clip=0, use_graph=False, return_x=True)[1]
        assert smp.profile_gate_kernel(8, T, 1)[2] == nt
    den.debug_set("fused_nt", 0)
    assert torch.isfinite(out[4]).all()
    d2, d1 = (out[2] - out[4]).abs().max().item(), (out[1] - out[4]).abs().max().item()
    print("fused layer kernel %s, 64- / 32-frame tiles vs 128-frame tiles: max |diff| %.3e / %.3e" % (precision, d2, d1))
    assert torch.equal(out[2], out[4]) and torch.equal(out[1], out[4]), (d2, d1)


@pytest.mark.parametrize("arch", ["44k", "24k"])
def test_fused_step_tail_vs_the_three_launches_and_across_graph_replays(arch, hooks):
    """The tail of a batched DDPM step as ONE kernel (round 5, csrc/ttail.h: skip projection -> output projection + posterior step -> the NEXT
    evaluation's input projection; the default of the f16_w6 fused-layer path) against the three tgemm launches it replaces (debug_set
    'fused_tail' 0).  Same operands and split scheme, but the fused kernel walks K in plain order and leaves out W_lo x_lo (2^-22 of a
    product): one step agrees to fp32 rounding, 20 steps to the chain's own sensitivity (the dither variants move a 20-step state by 3e-5
    too); its 64- and 32-frame tilings are bit-identical.  Then the host bookkeeping: the captured steps of a hipGraph replay expect their
    input projection done by the step before them, so a chain that STARTS on the period boundary with an already captured graph (the second
    call below) has to run that one projection eagerly -- eager, first (capturing) and second (replay-only) call must agree bit for bit.
    Both instantiations: C = 384 (44.1 kHz) and C = 256 (the 24 kHz architecture of BASELINE configs[0])."""
    hp = dict(synth.HPARAMS_44K if arch == "44k" else synth.HPARAMS_24K, K_step=192)
    sd, den, smp = make_handles(hp, 0 if arch == "44k" else 2, "f16_w6")
    den.debug_set("two_launch_layer", -1)                 # the fused layer kernel at 8 clips (the 24 kHz handle would choose it from 18 clips on)
    den.debug_set("fused_nt", 2)
    clips, T, n_units, seed = list(range(8)), 861, 500, 91
    hub, m2p, f0 = clip_batch(hp, clips, T, n_units)
    cond, _, _ = O.build_cond(sd, hub, m2p, f0.clone(), hp)
    cond = cond.transpose(1, 2).contiguous().cuda()
    run = lambda n, graph: smp.sample(cond, n, mel2ph=m2p.cuda(), seed=seed, first_clip=0, use_graph=graph, return_x=True)[1]
    for n, bar in ((1, 1e-6), (20, 2e-4)):
        out = {}
        for mode in (0, 2, 3):
            den.debug_set("fused_tail", mode)
            out[mode] = run(n, False)
        d = (out[2] - out[0]).abs().max().item()
        print("fused step tail (%s), %d steps at 8 x 861: max |diff| of the state vs the three launches %.2e; 32- vs 64-frame tiles identical: %s" % (
            arch, n, d, torch.equal(out[2], out[3])))
        assert torch.isfinite(out[2]).all() and 0.0 < d < bar, d
        assert torch.equal(out[2], out[3])
    den.debug_set("fused_tail", 1)
    eager = run(192, False)                               # t = 191: the chain starts on the 64-step period boundary
    first = run(192, True)                                # captures (one eager step first), then replays
    second = run(192, True)                               # replays only: the first graph launch opens the chain
    longer = run(150, True)                               # eager walk to the boundary, two replays, eager steps behind them
    assert torch.isfinite(eager).all()
    assert torch.equal(first, eager) and torch.equal(second, eager)
    assert torch.equal(longer, run(150, False))
    den.debug_set("fused_tail", 0)
    d = (run(192, True) - eager).abs().max().item()
    print("fused step tail (%s), 192 steps through graph replays: bit-identical to eager; vs the three launches %.2e" % (arch, d))
    assert 0.0 < d < 1e-3


@pytest.mark.parametrize("precision", ["f16_w2", "f16_m64"])
def test_deferred_skip_contraction_taps_and_equivalence(precision, hooks):
    """The deferred skip path (debug_set 'defer_skip' 1; not the default: time-neutral, design/tlayer.md): the layer kernels write the gate
    output g to HBM and compute only the residual half of the 1x1; ONE [C x L*C] contraction per evaluation (tskip.h) produces
    relu(skip_projection(sum of skips / sqrt(L))) from all layers' g with weights composed at load time.  Checked at 8 x 861: (1) per-layer residual stream x_l and gate output g_l against the oracle
    (g_l is now tappable in the fused form: it is in HBM), (2) the contraction's output against relu(skip_projection(.)) of the oracle,
    (3) a 20-step DDPM chain against the in-layer form of the same kernels -- equal up to fp32 summation order."""
    import torch.nn.functional as F
    hp = dict(synth.HPARAMS_44K, K_step=20)
    sd, den, smp = make_handles(hp, 0, precision)
    den.debug_set("two_launch_layer", -1)        # the fused layer kernel at 8 clips (the deferred form lives in it)
    den.debug_set("defer_skip", 1)
    B, T = 8, 861
    g = np.random.Generator(np.random.PCG64(23))
    spec = torch.from_numpy(g.standard_normal((B, 1, 128, T)).astype(np.float32))
    cond = torch.from_numpy((g.standard_normal((B, 256, T)) * 0.5).astype(np.float32))
    t = torch.from_numpy(g.integers(0, 1000, size=(B,)))
    Tp = (T + 8 + 127) // 128 * 128          # csrc/diffnet.hip: bucket_rows
    taps = {}
    with torch.no_grad():
        ref_out = O.diffnet_forward(sd, spec, t, cond, 4, taps=taps)
        s_ref = F.relu(F.conv1d(taps["skip"], sd["denoise_fn.skip_projection.weight"], sd["denoise_fn.skip_projection.bias"]))
    L = O.diffnet_layers(sd)
    worst = {"x": 0.0, "g": 0.0}
    try:
        for n in (1, 2, 7, L):
            den.debug_set("stop_after_layers", n)
            den.forward(spec.cuda(), t.cuda(), cond.cuda())
            bx, bg = den.debug_buffer("xres").cpu(), den.debug_buffer("g").cpu()
            for b in range(B):
                rows = slice(b * Tp, b * Tp + T)
                worst["x"] = max(worst["x"], (bx[rows] - taps["x%d" % (n - 1)][b].T).abs().max().item())
                worst["g"] = max(worst["g"], (bg[rows] - taps["g%d" % (n - 1)][b].T).abs().max().item())
    finally:
        den.debug_set("stop_after_layers", -1)
    out = den.forward(spec.cuda(), t.cuda(), cond.cuda()).cpu()
    s2 = den.debug_buffer("s2").cpu()                          # hi plane of relu(skip projection)
    es = max((s2[b * Tp:b * Tp + T] - s_ref[b].T).abs().max().item() for b in range(B))
    eo = (out - ref_out).abs().max().item()
    print("deferred skip path %s: worst |err| x %.2e, g %.2e, relu(skip proj) %.2e (fp16 hi plane), eps %.2e" % (precision, worst["x"], worst["g"], es, eo))
    tol = 1.5e-2 if precision != "f16_w2" else 1e-2
    assert worst["x"] < tol and worst["g"] < tol and es < 2 * tol and eo < FWD_TOL[precision]
    clips, n_units, seed = list(range(8)), 500, 77
    hub, m2p, f0 = clip_batch(hp, clips, T, n_units)
    cnd, _, _ = O.build_cond(sd, hub, m2p, f0.clone(), hp)
    cnd = cnd.transpose(1, 2).contiguous().cuda()
    a = smp.sample(cnd, 20, mel2ph=m2p.cuda(), seed=seed, first_clip=0, use_graph=False, return_x=True)[1]
    den.debug_set("defer_skip", 0)
    b = smp.sample(cnd, 20, mel2ph=m2p.cuda(), seed=seed, first_clip=0, use_graph=False, return_x=True)[1]
    d = (a - b).abs().max().item()
    print("deferred vs in-layer skip accumulation (%s), 20 DDPM steps at 8 x 861: max |diff| of the state %.2e" % (precision, d))
    assert torch.isfinite(a).all() and 0.0 < d < 2e-4


@pytest.mark.parametrize("precision", ["f16_w6", "f16_w6n", "f16_w2", "f16_w6/2", "f16_w6/1", "f16_w6n/1"])
def test_fused_layer_kernels_on_the_24k_architecture(precision, hooks):
    """The fused layer kernel's two-block instantiations (C = 256: the 24 kHz demo architecture, BASELINE configs[0]'s shapes) -- every other test of
    the batched path runs the three-block 44.1 kHz ones.  8 clips x T = 861 on the fused kernel (forced: the automatic choice needs >= 120 tiles), a
    30-step DDPM chain, every clip against the oracle's chain from the same Philox noise: f16_w6 / f16_w6n (the 6-bit correction products and the
    gate-output correction with NB = 2) and f16_w2 stay in the fp16-activation class (a wrong code layout shows at >= 1e-2)."""
    precision, _, nt = precision.partition("/")           # "/2", "/1": the 64- / 32-frame tiles of the mid-size batches (round 5)
    hp = dict(synth.HPARAMS_24K, K_step=30)
    sd, den, smp = make_handles(hp, 2, precision)
    den.debug_set("two_launch_layer", -1)
    if nt:
        den.debug_set("fused_nt", int(nt))
    clips, T, n_units, seed = list(range(8)), 861, 500, 41
    ref = oracle_sample(hp, sd, clips, T, n_units, 1, seed, 30)
    mel = smp.sample(ref["cond_t"].cuda(), 30, mel2ph=ref["mel2ph"].cuda(), seed=seed, first_clip=0, use_graph=False).cpu()
    errs = [(mel[b] - ref["mel_out"][b]).abs().max().item() for b in range(len(clips))]
    print("24 kHz architecture, fused layer kernel, %s: 30-step mel max-abs err per clip %s" % (precision, ["%.1e" % e for e in errs]))
    assert torch.isfinite(mel).all() and max(errs) < 2e-3, errs
