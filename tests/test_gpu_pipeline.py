"""GPU: the drop-in modules behind the reference's plugin seams, end to end (units + f0 -> cond -> sampler ->
NSF-HiFiGAN PCM), against the oracle; plus size-independent properties at the BASELINE clip size (T=861)."""
import os
import wave

import numpy as np
import pytest
import torch

from diffsvc_amd import synth
import dsvc_oracle as O
from util import clip_batch, oracle_sample

pytestmark = pytest.mark.gpu


def tiny_pipeline(K=30, precision="f16_x3"):
    from diffsvc_amd.pipeline import SvcPipeline
    hp = synth.tiny_hparams(K=K)
    h = synth.tiny_vocoder(num_mels=hp["audio_num_mel_bins"])
    sd, vs = synth.acoustic_state(hp, 3), synth.vocoder_state(h, 5)
    return SvcPipeline(hp, sd, vs, h, precision=precision, vocoder_precision="f16_x3"), hp, h, sd, vs


@pytest.mark.parametrize("precision", ["f16_x3", "auto"])          # auto: what ships (f16_x3t at this size, the tgemm engine)
@pytest.mark.parametrize("speedup", [1, 10])
def test_pipeline_end_to_end_vs_oracle(speedup, precision):
    pipe, hp, h, sd, vs = tiny_pipeline(precision=precision)
    T, n_units, clips, seed = 40, 23, [0, 1], 11
    hub, m2p, f0 = clip_batch(hp, clips, T, n_units)
    wav, mel = pipe.infer(hub.cuda(), m2p.cuda(), f0.cuda(), speedup=speedup, seed=seed, first_clip=0, return_mel=True)
    r = oracle_sample(hp, sd, clips, T, n_units, speedup, seed, hp["K_step"])
    assert (mel.cpu() - r["mel_out"]).abs().max().item() < (1e-3 if speedup == 1 else 2e-3)
    hop = int(np.prod(h["upsample_rates"]))
    ini, nz = O.vocoder_rng(seed, clips, T * hop)
    gw = O.fold_weight_norm(vs)
    with torch.no_grad():                                   # same mel in: isolates the vocoder from sampler rounding
        c = 2.30259 * torch.clamp(mel.cpu(), hp["mel_vmin"], hp["mel_vmax"]).transpose(2, 1)
        wav_ref = O.generator_forward(gw, h, c, r["f0_denorm"], ini, nz).reshape(len(clips), -1)
    assert wav.shape == (2, T * hop)
    assert (wav.cpu() - wav_ref).pow(2).mean().sqrt().item() < 1e-4


def test_denoiser_module_seam():
    """DiffNetHip as the reference uses DiffNet: construct from hparams, strict load, .cuda(), call
    denoise_fn(x, t, cond=cond) with an int64 step tensor (diffusion.py:147)."""
    from diffsvc_amd.denoiser import DiffNetHip
    hp = synth.tiny_hparams()
    sd = synth.acoustic_state(hp, 3)
    den = DiffNetHip(hp["audio_num_mel_bins"], hparams=hp, precision="f16_x3")
    den.load_state_dict({k[len("denoise_fn."):]: v for k, v in sd.items() if k.startswith("denoise_fn.")}, strict=True)
    den.cuda()
    g = np.random.Generator(np.random.PCG64(2))
    spec = torch.from_numpy(g.standard_normal((2, 1, 16, 33)).astype(np.float32))
    cond = torch.from_numpy((g.standard_normal((2, 32, 33)) * 0.5).astype(np.float32))
    t = torch.tensor([7, 41], dtype=torch.long)
    out = den(spec.cuda(), t.cuda(), cond=cond.cuda())
    with torch.no_grad():
        ref = O.diffnet_forward(sd, spec, t, cond, hp["dilation_cycle_length"])
    assert (out.cpu() - ref).abs().max().item() < 3e-4
    # an in-place weight edit is picked up (the packed device copy is derived state, never stale)
    with torch.no_grad():
        den.output_projection.weight.mul_(2.0)
        den.output_projection.bias.mul_(2.0)
    out2 = den(spec.cuda(), t.cuda(), cond=cond.cuda())
    assert (out2.cpu() - 2 * ref).abs().max().item() < 6e-4


@pytest.mark.parametrize("precision", ["f16_x3", "auto"])
def test_sampler_module_use_gt_mel_start(precision):
    """use_gt_mel / add_noise_step (diffusion.py:255-261): the chain starts from q_sample(norm_spec(ref_mel), t-1); norm_spec,
    q_sample and the noise draw run inside the C ABI (dsvc_sample_args.ref_mel)."""
    from diffsvc_amd.denoiser import DiffNetHip
    from diffsvc_amd.sampler import GaussianDiffusionHip
    hp = synth.tiny_hparams(K=50)
    sd = synth.acoustic_state(hp, 3)
    M = hp["audio_num_mel_bins"]
    model = GaussianDiffusionHip(None, M, DiffNetHip(M, hparams=hp, precision=precision), timesteps=50, K_step=50,
                                 loss_type="l2", spec_min=hp["spec_min"], spec_max=hp["spec_max"], hparams=hp)
    model.load_state_dict(sd, strict=True)
    model.cuda()
    model.hp = dict(hp, pndm_speedup=1)
    T, n_units, seed, steps = 40, 23, 5, 20
    hub, m2p, f0 = clip_batch(hp, [0], T, n_units)
    g = np.random.Generator(np.random.PCG64(4))
    ref_mel = torch.from_numpy((g.standard_normal((1, T, M)) * 0.7 - 2.5).astype(np.float32))
    ret = model(hub.cuda(), mel2ph=m2p.cuda(), f0=f0.clone().cuda(), ref_mels=ref_mel.cuda(), infer=True,
                use_gt_mel=True, add_noise_step=steps, seed=seed)
    cond, _, _ = O.build_cond(sd, hub, m2p, f0.clone(), hp)
    x0 = O.norm_spec(sd, ref_mel).transpose(1, 2)[:, None]
    noise = O.ddpm_noise_ref_layout(seed, [0], 0, T, M, O.PURPOSE_X_INIT)     # dsvc_sample draws q_sample's noise from the x_T stream
    x = O.q_sample(sd, x0, torch.tensor([steps - 1]), noise)
    x = O.sample_ddpm(sd, cond.transpose(1, 2).contiguous(), x, lambda i: O.ddpm_noise_ref_layout(seed, [0], i, T, M),
                      hp["dilation_cycle_length"], t_start=steps)
    assert (ret["mel_out"].cpu() - O.finish_mel(sd, x, m2p)).abs().max().item() < 1e-3


def test_vocoder_plugin_contract(tmp_path):
    """NsfHifiGANHip through the reference's vocoder contract (nsf_hifigan.py:8-92): no-arg constructor reading
    hparams['vocoder_ckpt'] + sibling config.json, spec2wav(numpy mel, f0=numpy) -> numpy, static wav2spec(path)."""
    from diffsvc_amd.hparams import set_hparams
    from diffsvc_amd.vocoder import NsfHifiGANHip
    h = synth.tiny_vocoder(num_mels=16)
    vs = synth.save_vocoder_ckpt(str(tmp_path / "voc"), h, seed=5)
    hp = dict(synth.tiny_hparams(), vocoder_ckpt=str(tmp_path / "voc" / "model"), audio_sample_rate=44100,
              fft_size=512, win_size=512, hop_size=128, fmin=40, fmax=16000)
    set_hparams(hp)
    voc = NsfHifiGANHip()
    T = 24
    hop = int(np.prod(h["upsample_rates"]))
    g = np.random.Generator(np.random.PCG64(6))
    mel = (g.standard_normal((T, 16)) * 0.8 - 2.5).astype(np.float32)
    f0 = synth.clip_inputs(2, T=T, n_units=12)[3]
    wav = voc.spec2wav(mel, f0=f0, seed=42)
    assert isinstance(wav, np.ndarray) and wav.dtype == np.float32 and wav.shape == (T * hop,)
    ini, nz = O.vocoder_rng(42, [0], T * hop)
    ref = O.spec2wav(O.fold_weight_norm(vs), h, mel, f0, ini, nz).numpy()
    assert np.sqrt(np.mean((wav - ref) ** 2)) < 1e-4
    # wav2spec: 16-bit PCM file -> (wav, mel[T, M] log10)
    sr, n = 44100, 6000
    t = np.arange(n) / sr
    pcm = (0.4 * np.sin(2 * np.pi * 330 * t) * 32767).astype("<i2")
    path = str(tmp_path / "a.wav")
    with wave.open(path, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr); w.writeframes(pcm.tobytes())
    wav_in, mel_out = NsfHifiGANHip.wav2spec(path)
    assert np.array_equal(wav_in, pcm.astype(np.float32) / 32768.0)
    ref_mel = O.mel_spectrogram(torch.from_numpy(wav_in)[None], sr, 512, 512, 128, 16, 40, 16000)[0].numpy()
    assert mel_out.shape == ref_mel.shape and np.abs(mel_out - ref_mel).max() < 1e-3
    with pytest.raises(FileNotFoundError):
        set_hparams(dict(hp, vocoder_ckpt=str(tmp_path / "missing" / "model")))
        NsfHifiGANHip()


def test_full_size_clip_properties():
    """BASELINE clip size (T=861, 44.1 kHz architecture), 24 DDPM steps -- properties that need no oracle run:
    (1) a chain split at an arbitrary step composes exactly (noise is keyed by (seed, clip, step));
    (2) a batch of clips equals the per-clip runs up to fp32 summation order (a single clip uses the split-K tiling, a
    batch does not; tests/test_gpu_diffnet.py::test_batch_equals_per_clip checks bit-equality under one tiling); (3) graph replay == eager; (4) output is finite
    and inside the denormalised range [spec_min, spec_max] (x0 is clamped to [-1, 1] at t=0)."""
    from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
    hp = dict(synth.HPARAMS_44K)
    sd = synth.acoustic_state(hp, 0)
    den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision="f16_w2", prefix="denoise_fn.")
    smp = SamplerHandle(den, sd)
    T, K = 861, 24
    hub, m2p, f0 = clip_batch(hp, [0, 1], T, 500)
    cond, _, _ = O.build_cond(sd, hub, m2p, f0.clone(), hp)
    cond = cond.transpose(1, 2).contiguous().cuda()
    full, xf = smp.sample(cond, K, seed=5, first_clip=0, use_graph=False, return_x=True)
    part, xp = smp.sample(cond, K, seed=5, first_clip=0, t_stop=9, use_graph=False, return_x=True)
    rest, xr = smp.sample(cond, 9, x_init=xp, seed=5, first_clip=0, use_graph=False, return_x=True)
    assert torch.equal(xr, xf) and torch.equal(rest, full)
    one = smp.sample(cond[1:2], K, seed=5, first_clip=1, use_graph=False)
    assert (one[0] - full[1]).abs().max().item() < 2e-4, (one[0] - full[1]).abs().max().item()
    graph = smp.sample(cond, K, seed=5, first_clip=0, use_graph=True)
    assert torch.equal(graph, full)
    assert torch.isfinite(full).all()
    assert full.min().item() >= hp["spec_min"][0] - 1e-5 and full.max().item() <= hp["spec_max"][0] + 1e-5


def test_full_size_vocoder_properties():
    """10 s clip through the 44.1 kHz generator: output length T*512, finite, |y| <= 1 (tanh), and the first
    frames equal a short-clip run wherever the receptive field allows (the net is fully convolutional)."""
    from diffsvc_amd.engine import VocoderHandle
    h = dict(synth.VOCODER_44K)
    voc = VocoderHandle(synth.vocoder_state(h, 1), h, precision="f16_x3")
    T = 861
    g = np.random.Generator(np.random.PCG64(12))
    mel = torch.from_numpy((g.standard_normal((1, T, 128)) * 0.8 - 2.5).astype(np.float32)).cuda()
    f0 = torch.from_numpy(synth.clip_inputs(0, T=T)[3])[None].cuda()
    wav = voc.vocode(mel, f0, seed=1, first_clip=0)
    assert wav.shape == (1, T * 512) and torch.isfinite(wav).all() and wav.abs().max().item() <= 1.0
    short = voc.vocode(mel[:, :64].contiguous(), f0[:, :64].contiguous(), seed=1, first_clip=0)
    n = (64 - 24) * 512                                     # receptive field of the generator < 24 frames
    assert (short[0, :n] - wav[0, :n]).abs().max().item() < 1e-5


def _run_bench(nproc, clips_per_gpu, extra_env=None, share_device=True, self_spawn=False, ddpm_steps=20, train=False):
    import json, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DSVC_BENCH_PCM_STATS="1", **(extra_env or {}))
    if share_device:
        env["DSVC_BENCH_SHARE_DEVICE"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what RCCL needs on this host driver
    tail = [os.path.join(root, "bench.py"), "--gpus", str(nproc), "--steps", "1", "--warmup", "1", "--ddpm-steps", str(ddpm_steps),
            "--clips-per-gpu", str(clips_per_gpu), "--no-cpu-baseline", "--no-batched"]
    if train:
        tail = [os.path.join(root, "bench.py"), "--gpus", str(nproc), "--steps", "4", "--warmup", "2", "--train"]
    if nproc > 1 and not self_spawn:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + tail
    else:
        cmd = [sys.executable] + tail
    if self_spawn:
        env = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_launch_contract_two_ranks_share_the_device():
    """The driver's N > 1 launch line (torch.distributed.run, one rank per GPU) exercised on a 1-GPU box: both ranks on
    cuda:0, gloo instead of RCCL.  Checks rendezvous, clip sharding, the gather, the max-over-ranks clock and the JSON line --
    and that the gathered PCM of the 2-rank job equals the 1-rank job clip for clip (noise streams are keyed by the global clip
    index; 2 clips per rank vs 4 clips on one rank run the same tiling family, so only the vocoder's batch-size dependent tiling
    can differ by summation order)."""
    d = _run_bench(2, 2)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["finite_output"] and d["value"] > 0
    assert d["config"]["parallelism"].startswith("utterance-sharded x2") and d["config"]["clips_per_gpu"] == 2
    assert d["rccl"]["world_size"] == 2
    # round 5: the line carries its own denominator -- rank 0 alone on its 2-clip share (no gather).  Two ranks SHARING one device cannot beat
    # one of them alone by much: the whole-job value over the solo value stays near 1 (a real 2-GPU job: near 2), never the 25x -> 900x
    # artefact of dividing a 32-clips-per-GPU line by the one-clip headline
    assert d["same_workload_1gpu"] > 0 and 0.5 < d["speedup_vs_1gpu_same_workload"] < 1.6, d["speedup_vs_1gpu_same_workload"]
    assert abs(d["scaling_efficiency"] - d["speedup_vs_1gpu_same_workload"] / 2) < 1e-9
    # round 4: the same job as plain `python bench.py --gpus 2` -- no launcher: bench.py spawns its ranks itself -- gives the same PCM
    d2 = _run_bench(2, 2, self_spawn=True)
    assert d2["n_gpus"] == 2 and d2["rccl"]["world_size"] == 2
    assert [s[:1] + [round(v, 6) for v in s[1:]] for s in d2["pcm_stats"]] == [s[:1] + [round(v, 6) for v in s[1:]] for s in d["pcm_stats"]]
    one = _run_bench(1, 4)
    assert [s[0] for s in d["pcm_stats"]] == [0, 1, 2, 3] == [s[0] for s in one["pcm_stats"]]
    for a, b in zip(d["pcm_stats"], one["pcm_stats"]):
        assert abs(a[1] - b[1]) <= 1e-3 + 1e-5 * abs(b[1]) and abs(a[2] - b[2]) <= 1e-5 * b[2], (a, b)


def test_bench_multi_gpu_line_carries_its_own_one_gpu_denominator():
    """`bench.py --gpus 1` times ONE clip per GPU (BASELINE configs[1]) and `--gpus N` 32 clips per GPU
    (configs[3]), so the driver's 1 -> N curve over the two `value`s would read the batch size as scaling.  Every N > 1 line therefore
    carries `same_workload_1gpu` -- rank 0 alone on its own 32-clip share, un-gathered, measured in the same run.  Here: two ranks x 32
    clips sharing the one device (a 200-step chain keeps it short; the ratio is what is checked): the solo leg must be the rate a plain
    one-rank `--clips-per-gpu 32` job measures (5 %), and two ranks time-slicing ONE device buy nothing over it (speed-up ~ 1, efficiency
    ~ 0.5 -- on two real GPUs: ~ 2 and ~ 1).  Same for --train."""
    d = _run_bench(2, 32, ddpm_steps=200)
    one = _run_bench(1, 32, ddpm_steps=200)
    print("bench --gpus 2 (shared device) x 32 clips: value %.1f, same_workload_1gpu %.1f, one-rank job %.1f, speed-up %.2f, efficiency %.2f"
          % (d["value"], d["same_workload_1gpu"], one["value"], d["speedup_vs_1gpu_same_workload"], d["scaling_efficiency"]))
    assert d["config"]["clips_per_gpu"] == 32 and one["config"]["clips_per_gpu"] == 32 and d["config"]["precision"] == one["config"]["precision"]
    assert abs(d["same_workload_1gpu"] / one["value"] - 1.0) < 0.05, (d["same_workload_1gpu"], one["value"])
    assert 0.8 < d["speedup_vs_1gpu_same_workload"] < 1.25 and abs(d["scaling_efficiency"] * 2 - d["speedup_vs_1gpu_same_workload"]) < 1e-9
    # round 6: the STRONG-scaling ratio north_star asks for -- the whole 64-clip job on one device (rank 0 alone, two batches of 32) against the
    # 2-rank job.  Two ranks time-slicing one device buy nothing: ~1 (on two real GPUs: ~2)
    ss = d["strong_scaling"]
    print("strong scaling: %d clips on one device %.2f s, on 2 ranks %.2f s, speed-up %.2f" % (ss["job_clips"], ss["one_gpu_s_per_job"], ss["n_gpu_s_per_job"], ss["speedup_vs_1gpu_whole_job"]))
    assert ss["job_clips"] == 64 and 0.8 < ss["speedup_vs_1gpu_whole_job"] < 1.25, ss
    assert abs(ss["one_gpu_value"] / one["value"] - 1.0) < 0.06, (ss["one_gpu_value"], one["value"])      # 2 x 32 back to back = the one-rank 32-clip rate
    t2 = _run_bench(2, 0, train=True)
    t1 = _run_bench(1, 0, train=True)
    print("bench --train --gpus 2 (shared device): value %.0f frames/s, same_workload_1gpu %.0f, one-rank job %.0f" % (t2["value"], t2["same_workload_1gpu"], t1["value"]))
    assert abs(t2["same_workload_1gpu"] / t1["value"] - 1.0) < 0.08, (t2["same_workload_1gpu"], t1["value"])
    assert 0.2 < t2["speedup_vs_1gpu_same_workload"] < 1.3      # (one device time-sliced by two ranks AND 128 MB of gradients all-reduced through the host by gloo: measured 0.43)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two HIP devices (RCCL over xGMI)")
@pytest.mark.parametrize("pcm16", [False, True])
def test_bench_two_gpus_over_rccl_equals_one_gpu(pcm16):
    """The real multi-GPU launch: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2`, one rank per GPU, backend
    'nccl' (= RCCL over xGMI), clip i on rank i % 2, one all_gather of the finished PCM -- as fp32 and as the 16-bit PCM the reference
    writes (infer.py:70).  The gathered PCM must equal the 1-GPU job clip for clip (global Philox clip ids)."""
    extra = {"DSVC_BENCH_PCM16": "1"} if pcm16 else {}
    d = _run_bench(2, 2, extra_env=extra, share_device=False)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["finite_output"] and d["value"] > 0
    assert d["config"]["gather"] == ("int16 PCM" if pcm16 else "fp32 PCM")
    one = _run_bench(1, 4, extra_env=extra)
    assert [s[0] for s in d["pcm_stats"]] == [0, 1, 2, 3] == [s[0] for s in one["pcm_stats"]]
    for a, b in zip(d["pcm_stats"], one["pcm_stats"]):
        assert abs(a[1] - b[1]) <= 1e-3 * (32767.0 if pcm16 else 1.0) + 1e-5 * abs(b[1]) and abs(a[2] - b[2]) <= 1e-5 * b[2], (a, b)


def _ragged_inputs(hp, clips, lens, n_units_of, T):
    """Clips of different lengths padded to a common T the way a batched caller pads them: mel2ph == 0, f0 = 0 and zero content
    rows beyond a clip's own frames."""
    N = max(n_units_of(l) for l in lens)
    hub = torch.zeros(len(clips), N, hp["hidden_size"])
    m2p = torch.zeros(len(clips), T, dtype=torch.long)
    f0 = torch.zeros(len(clips), T)
    per = []
    for i, (c, l) in enumerate(zip(clips, lens)):
        h, m, f, _ = synth.clip_inputs(int(c), T=l, n_units=n_units_of(l), H=hp["hidden_size"])
        hub[i, :h.shape[0]] = torch.from_numpy(h); m2p[i, :l] = torch.from_numpy(m); f0[i, :l] = torch.from_numpy(f)
        per.append((torch.from_numpy(h)[None], torch.from_numpy(m)[None], torch.from_numpy(f)[None]))
    return hub, m2p, f0, per


@pytest.mark.parametrize("precision", ["f16_x3", "auto"])
@pytest.mark.parametrize("speedup", [1, 10])
def test_ragged_batch_equals_per_clip_reference_runs(speedup, precision):
    """Variable-length clips in one padded batch (Svc.after_infer, infer_tool.py:177-191): the reference processes every clip
    ALONE at its own length, so the batch must reproduce those runs -- trailing padded frames act as the convs' zero padding
    inside the sampler, all-zero mel frames are dropped and f0 is cut with the same mask before the vocoder.  Checked against
    the oracle run per clip at the clip's own length: mel (1e-3) and PCM end to end (1e-4 RMS)."""
    pipe, hp, h, sd, vs = tiny_pipeline(K=30, precision=precision)
    clips, lens, T, seed = [4, 1, 7, 2], [40, 33, 40, 21], 40, 19
    n_units_of = lambda l: max(2, (l * 23) // 40)
    hub, m2p, f0, per = _ragged_inputs(hp, clips, lens, n_units_of, T)
    ids = torch.tensor(clips, dtype=torch.int32, device="cuda")
    wav, mel, wlens = pipe.infer(hub.cuda(), m2p.cuda(), f0.cuda(), speedup=speedup, seed=seed, clip_ids=ids, return_mel=True, return_lens=True)
    hop = int(np.prod(h["upsample_rates"]))
    gw = O.fold_weight_norm(vs)
    assert wlens.tolist() == [l * hop for l in lens]
    for i, (c, l) in enumerate(zip(clips, lens)):
        r = oracle_sample(hp, sd, [c], l, n_units_of(l), speedup, seed, hp["K_step"])
        err = (mel[i, :l].cpu() - r["mel_out"][0]).abs().max().item()
        assert err < (1e-3 if speedup == 1 else 2e-3), (i, err)
        assert (mel[i, l:] == 0).all() and (wav[i, l * hop:] == 0).all()
        mel_k, f0_k = O.after_infer_mel(r["mel_out"][0].numpy(), r["f0_denorm"][0].numpy(), hp)
        assert mel_k.shape[0] == l
        ini, nz = O.vocoder_rng(seed, [c], l * hop)
        ref = O.spec2wav(gw, h, mel_k, f0_k, ini, nz)
        rms = (wav[i, :l * hop].cpu() - ref).pow(2).mean().sqrt().item()
        print("ragged batch speedup=%d clip %d (len %d): mel err %.2e, wav RMS err %.2e" % (speedup, c, l, err, rms))
        assert rms < (1e-4 if speedup == 1 else 1e-3), (i, rms)     # (PLMS has no clamp: its mel bar is 2e-3 on this tiny schedule)


def test_clip_ids_keep_a_clips_noise_stream_wherever_it_is_placed():
    """Philox streams are keyed by the GLOBAL clip id (dsvc_sample_args.clip_ids / dsvc_vocode's clip_ids): a clip produces the
    same PCM alone, inside a batch, and at any batch position -- what makes an N-GPU sharded job equal the 1-GPU job per clip."""
    pipe, hp, h, sd, vs = tiny_pipeline(K=20, precision="f16_w2")
    T, n_units, seed = 40, 23, 3
    clips = [9, 2, 5]
    hub, m2p, f0 = clip_batch(hp, clips, T, n_units)
    ids = torch.tensor(clips, dtype=torch.int32, device="cuda")
    full = pipe.infer(hub.cuda(), m2p.cuda(), f0.cuda(), seed=seed, clip_ids=ids)
    perm = [2, 0, 1]
    shuf = pipe.infer(hub[perm].cuda(), m2p[perm].cuda(), f0[perm].cuda(), seed=seed, clip_ids=ids[perm])
    assert torch.equal(shuf, full[perm])
    for i, c in enumerate(clips):
        one = pipe.infer(hub[i:i + 1].cuda(), m2p[i:i + 1].cuda(), f0[i:i + 1].cuda(), seed=seed, first_clip=c)
        assert torch.equal(one[0], full[i]), c


def test_denoiser_seam_recomputes_cond_projections_for_a_new_tensor_at_a_recycled_address():
    """DiffNetHip caches the hoisted conditioner projections per cond TENSOR (identity + version), not per address: a fresh cond
    of the same shape that the caching allocator places at the freed address of the previous one must not reuse them."""
    from diffsvc_amd.denoiser import DiffNetHip
    hp = synth.tiny_hparams()
    sd = synth.acoustic_state(hp, 3)
    den = DiffNetHip(hp["audio_num_mel_bins"], hparams=hp, precision="f16_x3")
    den.load_state_dict({k[len("denoise_fn."):]: v for k, v in sd.items() if k.startswith("denoise_fn.")}, strict=True)
    den.cuda()
    g = np.random.Generator(np.random.PCG64(8))
    spec = torch.from_numpy(g.standard_normal((1, 1, 16, 33)).astype(np.float32))
    t = torch.tensor([11], dtype=torch.long)
    conds = [torch.from_numpy((g.standard_normal((1, 32, 33)) * 0.5).astype(np.float32)) for _ in range(2)]
    c0 = conds[0].cuda()
    p0 = c0.data_ptr()
    den(spec.cuda(), t.cuda(), cond=c0)
    del c0
    c1 = conds[1].cuda()                                    # same shape, freshly allocated: normally lands on p0
    recycled = c1.data_ptr() == p0
    out = den(spec.cuda(), t.cuda(), cond=c1).cpu()
    with torch.no_grad():
        ref = O.diffnet_forward(sd, spec, t, conds[1], hp["dilation_cycle_length"])
    assert (out - ref).abs().max().item() < 3e-4, recycled
    # in-place edit of the SAME tensor bumps its version: also recomputed
    c1.mul_(0.5)
    out = den(spec.cuda(), t.cuda(), cond=c1).cpu()
    with torch.no_grad():
        ref = O.diffnet_forward(sd, spec, t, conds[1] * 0.5, hp["dilation_cycle_length"])
    assert (out - ref).abs().max().item() < 3e-4
    # steps are tabulated for 0 .. timesteps-1 only: anything else is clamped on the device (no D2H check per call on this seam) and
    # reported by the handle's check() -- and only by it (round 4)
    h = den.handle("forward")
    h.check()
    edge = den(spec.cuda(), torch.tensor([hp["timesteps"] - 1]).cuda(), cond=c1)
    over = den(spec.cuda(), torch.tensor([hp["timesteps"]]).cuda(), cond=c1)
    assert torch.equal(over, edge)                           # memory-safe: computed at the clamped step
    with pytest.raises(RuntimeError, match="diffusion step outside"):
        h.check()
    h.check()                                                # the flag is consumed
    den(spec.cuda(), torch.tensor([-1]).cuda(), cond=c1)
    torch.cuda.synchronize()
    assert torch.equal(den(spec.cuda(), t.cuda(), cond=c1).cpu(), out)      # a later VALID call is executed, not rejected for its predecessor
    with pytest.raises(RuntimeError, match="diffusion step outside"):
        h.check()                                            # the flag is sticky until check() reports it
    h.check()


def test_use_pe_drives_the_vocoder_with_the_extracted_f0():
    """Svc.infer(use_pe=True) (infer_tool.py:165-166): the vocoder's f0 is PitchExtractor(mel_out)['f0_denorm_pred'], not the input f0.
    A full batch and a ragged one (each clip's extractor run sees only its own frames, as the reference's B=1 loop does): the f0 the
    oracle extracts from the SAME sampled mel, through the oracle vocoder, must give the pipeline's PCM."""
    from diffsvc_amd.pipeline import SvcPipeline
    hp = synth.tiny_hparams(K=20)
    h = synth.tiny_vocoder(num_mels=hp["audio_num_mel_bins"])
    sd, vs = synth.acoustic_state(hp, 3), synth.vocoder_state(h, 5)
    M = hp["audio_num_mel_bins"]
    ps = synth.pe_state(hp, 4, n_mel=M)
    pipe = SvcPipeline(hp, sd, vs, h, precision="f16_x3", vocoder_precision="f16_x3", pe_state=ps)
    hop = int(np.prod(h["upsample_rates"]))
    gw = O.fold_weight_norm(vs)
    clips, lens, T, seed = [3, 8, 5], [40, 40, 40], 40, 23
    for lens in ([40, 40, 40], [40, 26, 33]):
        n_units_of = lambda l: max(2, (l * 23) // 40)
        hub, m2p, f0, _ = _ragged_inputs(hp, clips, lens, n_units_of, T)
        ids = torch.tensor(clips, dtype=torch.int32, device="cuda")
        wav, mel = pipe.infer(hub.cuda(), m2p.cuda(), f0.cuda(), seed=seed, clip_ids=ids, return_mel=True, use_pe=True)
        plain = pipe.infer(hub.cuda(), m2p.cuda(), f0.cuda(), seed=seed, clip_ids=ids)
        assert not torch.allclose(wav, plain, atol=1e-3)              # the extracted f0 really is what drives the source
        for i, (c, l) in enumerate(zip(clips, lens)):
            m_i = mel[i, :l].cpu()
            with torch.no_grad():
                _, f0_ref = O.pitch_extractor(ps, m_i[None], hp)
            mel_k, f0_k = O.after_infer_mel(m_i.numpy(), f0_ref[0].numpy(), hp)
            ini, nz = O.vocoder_rng(seed, [c], l * hop)
            ref = O.spec2wav(gw, h, mel_k, f0_k, ini, nz)
            rms = (wav[i, :l * hop].cpu() - ref).pow(2).mean().sqrt().item()
            print("use_pe lens=%s clip %d: wav RMS err %.2e" % (lens, c, rms))
            assert rms < 1e-4, (lens, i, rms)
            assert (wav[i, l * hop:] == 0).all()
    bare = SvcPipeline(hp, sd, vs, h, precision="f16_x3")
    with pytest.raises(RuntimeError, match="pe_state"):
        bare.infer(hub.cuda(), m2p.cuda(), f0.cuda(), use_pe=True)


def test_cond_builder_device_pitch_path_equals_host_path():
    """CondBuilder on device tensors (dsvc_cond_build: pitch threshold search + gather + embedding + mask + transpose in one launch, no
    host round trip) against the host path (the reference's
    torch-CPU expression per clip): identical pitch bins and decoder_inp, f0_denorm to the last ulp of exp2 -- on the benchmark inputs,
    a ragged batch with an interior mel2ph == 0 gap and the use_uv branch.  Pitch values sitting exactly on a bin threshold (and one
    ulp below) are checked against the reference expression evaluated one value at a time: there torch's vectorised CPU loop and its
    scalar tail may themselves disagree in the last bit (profiles/r2b_diag_position.txt), the device follows the scalar one."""
    from diffsvc_amd.cond import CondBuilder, coarse_thresholds
    hp = dict(synth.HPARAMS_44K)
    sd = synth.acoustic_state(hp, 0)
    cb = CondBuilder(hp)
    cb.load_state_dict({k[len("fs2."):]: v for k, v in sd.items() if k.startswith("fs2.")}, strict=True)
    cbd = CondBuilder(hp).cuda()
    cbd.load_state_dict(cb.state_dict(), strict=True)
    hub, m2p, f0 = clip_batch(hp, [0, 3, 7], 861, 500)
    m2p = m2p.clone(); f0 = f0.clone()
    m2p[1, 700:] = 0; f0[1, 700:] = 0                        # trailing padding
    m2p[2, 100:110] = 0                                     # an interior gap
    uv = (torch.arange(861)[None].repeat(3, 1) % 97 == 0).float()
    for use_uv in (False, True):
        cb.hp = dict(hp, use_uv=use_uv); cbd.hp = cb.hp
        f0_host, f0_dev = f0.clone(), f0.clone().cuda()
        a = cb(hub, mel2ph=m2p, f0=f0_host, uv=uv)
        b = cbd(hub.cuda(), mel2ph=m2p.cuda(), f0=f0_dev, uv=uv.cuda())
        assert torch.equal(a["pitch_pred"], b["pitch_pred"].cpu())
        assert torch.equal(a["decoder_inp"], b["decoder_inp"].cpu())
        assert torch.equal(a["decoder_inp"].view(torch.int32), b["decoder_inp"].cpu().view(torch.int32))      # bit for bit, signed zeros included
        assert torch.equal(b["cond_bht"], b["decoder_inp"].transpose(1, 2))                                  # dsvc_cond_build's second layout
        assert torch.equal(f0_host, f0_dev.cpu()) and (f0_dev.cpu()[m2p == 0] == 0).all()                    # the in-place f0[mel2ph == 0] = 0 (fs2.py:231)
        fa, fb = a["f0_denorm"], b["f0_denorm"].cpu()
        assert torch.equal(fa == 0, fb == 0)
        assert ((fa - fb).abs() <= 2.4e-7 * fa.abs()).all()
        if use_uv:
            assert (b["pitch_pred"][:, ::97, 0] == 1).all() and (fb[:, ::97] == 0).all()
    thr = coarse_thresholds(hp)
    n = thr.numel()
    probe = torch.cat([thr, torch.from_numpy(np.nextafter(thr.numpy(), np.float32(0)))])[None]           # on / one ulp below every step
    cbd.hp = hp
    r = cbd(hub[:1, :, :].cuda(), mel2ph=torch.ones(1, 2 * n, dtype=torch.long).cuda(), f0=probe.cuda())
    assert (r["pitch_pred"][0, :n, 0].cpu() == torch.arange(2, n + 2)).all()
    assert (r["pitch_pred"][0, n:, 0].cpu() == torch.arange(1, n + 1)).all()
    # an alignment index past the content frames: the reference's torch.gather raises (fs2.py:100-102); the one-launch device path stays
    # memory-safe (zero pad row) and keeps a sticky flag that check_alignment() turns into the same IndexError, once
    cbd.check_alignment()
    bad = m2p.clone(); bad[0, 5] = hub.shape[1] + 3
    out_bad = cbd(hub.cuda(), mel2ph=bad.cuda(), f0=f0.clone().cuda())
    assert torch.equal(out_bad["decoder_inp"][0, 5], cbd.pitch_embed.weight[out_bad["pitch_pred"][0, 5, 0]])     # zero content + the frame's pitch embedding
    cbd(hub.cuda(), mel2ph=m2p.cuda(), f0=f0.clone().cuda())                   # a later valid call does not clear it
    with pytest.raises(IndexError, match="mel2ph holds an index outside"):
        cbd.check_alignment()
    cbd.check_alignment()
    with pytest.raises((IndexError, RuntimeError)):
        cb(hub, mel2ph=bad, f0=f0.clone())                                     # the host path IS torch.gather


@pytest.mark.gpu
def test_cond_builder_energy_branch_on_the_device_vs_real_fastspeech2():
    """use_energy_embed through the device builder (dsvc_cond_build + the energy lookup behind it) against the golden minted from the REAL
    FastSpeech2.forward (tests/golden/cond_energy_tiny.npz): decoder_inp bit for bit, and cond_bht is its transpose."""
    import os
    import numpy as np
    from diffsvc_amd import synth
    from diffsvc_amd.cond import CondBuilder
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cond_energy_tiny.npz"))
    hp = dict(synth.tiny_hparams(K=50), use_energy_embed=True)
    cb = CondBuilder(hp)
    cb.load_state_dict({"pitch_embed.weight": torch.from_numpy(g["pitch_embed"]), "energy_embed.weight": torch.from_numpy(g["energy_embed"])}, strict=True)
    cb = cb.cuda()
    hub, m2p, f0, en = (torch.from_numpy(g[k]).cuda() for k in ("hubert", "mel2ph", "f0", "energy"))
    with torch.no_grad():
        ret = cb(hub, mel2ph=m2p, f0=f0.clone(), energy=en, infer=True)
    assert np.array_equal(ret["decoder_inp"].cpu().numpy(), g["decoder_inp"])
    assert torch.equal(ret["cond_bht"], ret["decoder_inp"].transpose(1, 2))
    assert np.array_equal(ret["pitch_pred"].cpu().numpy(), g["pitch"])


@pytest.mark.parametrize("arch", ["tiny", "44k"])
def test_pipelined_job_equals_separate_batches_bit_for_bit(arch):
    """SvcPipeline.infer_job (round 6: north_star's 1-GPU denominator, the whole job on one device -- batch.py:25-43's loop as batches, the
    vocoder of batch k on a second stream under the sampler of batch k+1): the PCM equals what separate ``infer`` calls per batch return,
    bit for bit, overlapped or not, ragged last batch included.  44k: the batched precision on the fused layer kernel (8 clips per batch)."""
    if arch == "tiny":
        pipe, hp, h, sd, vs = tiny_pipeline(K=30, precision="auto")
        N, cpb, T, n_units = 10, 4, 40, 23
    else:
        from diffsvc_amd.pipeline import SvcPipeline
        hp = dict(synth.HPARAMS_44K, K_step=20)
        h = dict(synth.VOCODER_44K)
        sd, vs = synth.acoustic_state(hp, 0), synth.vocoder_state(h, 1)
        pipe = SvcPipeline(hp, sd, vs, h, precision="auto", vocoder_precision="f16_x3")
        N, cpb, T, n_units = 20, 8, 861, 500
    hub, m2p, f0 = (t.cuda() for t in clip_batch(hp, list(range(N)), T, n_units))
    ids = torch.arange(100, 100 + N, dtype=torch.int32, device="cuda")
    ref = torch.cat([pipe.infer(hub[lo:lo + cpb], m2p[lo:lo + cpb], f0[lo:lo + cpb], seed=9, clip_ids=ids[lo:lo + cpb], full_length=True)
                     for lo in range(0, N, cpb)])
    for overlap in (True, False, True):
        job = pipe.infer_job(hub, m2p, f0, clips_per_batch=cpb, seed=9, clip_ids=ids, overlap=overlap)
        torch.cuda.synchronize()
        assert torch.isfinite(job).all()
        assert torch.equal(job, ref), (arch, overlap, (job - ref).abs().max().item())


def _utterance_chunks(hp, lens, first):
    chunks = []
    for i, T in enumerate(lens):
        a, b, c, _ = synth.clip_inputs(first + i, T=T, n_units=max(2, T * 500 // 861), H=hp["hidden_size"])
        chunks.append(tuple(torch.from_numpy(v).cuda() for v in (a, b, c)))
    return chunks


def test_infer_chunks_one_by_one_is_the_per_chunk_infer_loop_and_batched_stays_inside_the_bars():
    """SvcPipeline.infer_chunks (round 6, second session): the chunks of one utterance (infer.py:44-67 hands them to the model one by one, each
    with its own T) as a few padded batches.  ``batch=False`` IS the reference's loop -- bit-identical to ``infer`` per chunk.  Batched, chunk i
    keeps the noise streams of clip ``first_clip + i``; a group whose rows fill the fused layer kernel runs at the batched precision (`auto`),
    so its PCM differs from the one-by-one pass by what separates the two shipped operand schemes after 60 steps of the chain -- far inside
    north_star's 1e-4 RMS -- and every chunk keeps its own length."""
    from diffsvc_amd.pipeline import SvcPipeline
    hp = dict(synth.HPARAMS_44K, K_step=60)
    h = dict(synth.VOCODER_44K)
    pipe = SvcPipeline(hp, synth.acoustic_state(hp, 0), synth.vocoder_state(h, 1), h, precision="auto", vocoder_precision="f16_x3")
    lens = [430, 700, 861, 1200, 1600, 2100, 2600]
    chunks = _utterance_chunks(hp, lens, 300)
    hop = pipe.vocoder.hop
    loop = [pipe.infer(c[0][None], c[1][None], c[2][None], seed=5, first_clip=300 + i, full_length=True)[0] for i, c in enumerate(chunks)]
    one = pipe.infer_chunks(chunks, seed=5, first_clip=300, batch=False)
    for i, (a, b) in enumerate(zip(one, loop)):
        assert a.shape == (lens[i] * hop,) and torch.equal(a, b), (i, lens[i])
    plan = pipe.plan_chunks(lens)
    assert sorted(i for g in plan for i in g) == list(range(len(lens))) and len(plan) < len(lens), plan
    assert any(pipe.model.denoise_fn.precision_for("ddpm", 1, frames=len(g) * lens[g[0]], clips=len(g)) == "f16_w6" for g in plan), plan
    got = pipe.infer_chunks(chunks, seed=5, first_clip=300)
    torch.cuda.synchronize()
    worst = 0.0
    for i, (a, b) in enumerate(zip(got, loop)):
        assert a.shape == b.shape and torch.isfinite(a).all(), (i, a.shape, b.shape)
        rms = float((a.double() - b.double()).pow(2).mean().sqrt())
        worst = max(worst, rms)
        assert rms < 2e-5, (i, lens[i], rms)             # measured 1e-6 ... 4e-6 after 60 steps (two operand schemes, two tilings)
    print("infer_chunks: plan %s, worst PCM rms against the one-by-one loop %.2e" % ([[lens[i] for i in g] for g in plan], worst))
    # PLMS chunks are batched too (the split-operand precision at any size: tilings only)
    hp_p = dict(hp, K_step=1000)
    pipe.hp = hp_p; pipe.model.K_step = 1000
    plms_loop = pipe.infer_chunks(chunks[:4], seed=6, first_clip=300, speedup=20, batch=False)
    plms_b = pipe.infer_chunks(chunks[:4], seed=6, first_clip=300, speedup=20)
    assert len(pipe.plan_chunks(lens[:4], 20)) < 4
    for i, (a, b) in enumerate(zip(plms_b, plms_loop)):
        assert a.shape == b.shape and float((a.double() - b.double()).pow(2).mean().sqrt()) < 1e-5, (i, float((a.double() - b.double()).pow(2).mean().sqrt()))
    pipe.hp = hp; pipe.model.K_step = 60
    # at a pinned precision the grouping changes the tiling only (fp32 summation order)
    pipe2 = SvcPipeline(hp, synth.acoustic_state(hp, 0), synth.vocoder_state(h, 1), h, precision="f16_x3t", vocoder_precision="f16_x3")
    sub = chunks[:3]
    a2 = pipe2.infer_chunks(sub, seed=5, first_clip=300, batch=False)
    pipe2.plan_chunks = lambda lengths, speedup=1: [[2, 1, 0]]
    b2 = pipe2.infer_chunks(sub, seed=5, first_clip=300)
    for i, (a, b) in enumerate(zip(a2, b2)):
        assert float((a.double() - b.double()).pow(2).mean().sqrt()) < 2e-6, i
