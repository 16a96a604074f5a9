"""CPU: host-side mirror of the reference seams (no GPU, no compute through the C ABI).

  * state-dict contract: the drop-in modules own exactly the keys/shapes the REAL reference modules own
    (tests/golden/state_keys.json, minted from /root/reference by oracle/make_golden.py), so the reference's
    strict checkpoint load (utils/__init__.py:178-209) works on them;
  * condition builder: index work bit-exact against the reference goldens;
  * utterance sharding + the one collective (all_gather of PCM) under gloo, world_size 2;
  * the product path refuses to run without a HIP device instead of falling back.
"""
import json
import os
import re
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from diffsvc_amd import synth
from diffsvc_amd.cond import CondBuilder
from diffsvc_amd.denoiser import DiffNetHip
from diffsvc_amd.pipeline import gather_pcm, shard_clips
from diffsvc_amd.sampler import GaussianDiffusionHip
from util import GOLD, clip_batch, load_golden


def _keys():
    with open(os.path.join(GOLD, "state_keys.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("tag,hp", [("tiny", synth.tiny_hparams()), ("44k", dict(synth.HPARAMS_44K))])
def test_acoustic_state_dict_contract(tag, hp):
    ref = _keys()["acoustic_" + tag]
    den = DiffNetHip(hp["audio_num_mel_bins"], hparams=hp)
    model = GaussianDiffusionHip(None, hp["audio_num_mel_bins"], den, timesteps=hp["timesteps"], K_step=hp["K_step"],
                                 loss_type=hp["diff_loss_type"], spec_min=hp["spec_min"], spec_max=hp["spec_max"], hparams=hp)
    sd = synth.acoustic_state(hp, 0)
    assert {k: list(v.shape) for k, v in sd.items()} == ref                 # the synthetic checkpoint has the reference's keys
    model.load_state_dict(sd, strict=True)                                  # ... and loads strictly into the drop-in
    own = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert own == ref, (set(own) ^ set(ref))
    # loaded buffers win over recomputed ones: corrupt one and see it survive the load
    sd2 = dict(sd, betas=sd["betas"] * 0.5)
    model.load_state_dict(sd2, strict=True)
    assert torch.equal(model.betas, sd2["betas"])


@pytest.mark.parametrize("tag,h", [("tiny", synth.tiny_vocoder()), ("44k", dict(synth.VOCODER_44K)), ("24k", dict(synth.VOCODER_24K))])
def test_vocoder_checkpoint_keys_match_reference(tag, h):
    ref = _keys()["vocoder_" + tag]
    sd = synth.vocoder_state(h, 1)
    assert {k: list(v.shape) for k, v in sd.items()} == ref


@pytest.mark.parametrize("name", ["ddpm_tiny", "plms_tiny_s5", "ddpm_44k_k20"])
def test_cond_builder_bit_exact_vs_reference(name):
    g = load_golden(name)
    hp = synth.tiny_hparams() if "tiny" in name else dict(synth.HPARAMS_44K)
    sd = synth.acoustic_state(hp, int(g["wseed"]))
    cb = CondBuilder(hp)
    cb.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("fs2.")}, strict=True)
    hub, m2p, f0 = clip_batch(hp, [int(c) for c in g["clips"]], int(g["T"]), int(g["n_units"]))
    f0_arg = f0.clone()
    ret = cb(hub, m2p, None, None, f0_arg, None, None, infer=True)
    assert np.array_equal(ret["pitch_pred"].numpy(), g["pitch"])            # index work: bit-exact
    assert np.array_equal(ret["f0_denorm"].numpy(), g["f0_denorm"])
    assert np.array_equal(ret["decoder_inp"].detach().numpy(), g["decoder_inp"])


def test_cond_builder_padding_and_unsupported_configs():
    hp = synth.tiny_hparams()
    cb = CondBuilder(hp)
    hub, m2p, f0 = clip_batch(hp, [0], 40, 23)
    m2p[0, 30:] = 0                                                         # padded frames
    f0_arg = f0.clone()
    ret = cb(hub, m2p, None, None, f0_arg, None, None, infer=True)
    assert (ret["decoder_inp"][0, 30:] == 0).all() and (ret["f0_denorm"][0, 30:] == 0).all()
    assert (f0_arg[0, 30:] == 0).all()                                      # the reference mutates its f0 argument (fs2.py:231)
    with pytest.raises(NotImplementedError):
        CondBuilder(dict(hp, no_fs2=False))(hub, m2p, None, None, f0.clone(), None, None)


def test_shard_clips_partition():
    for n, w in ((256, 8), (7, 2), (3, 4), (0, 2)):
        parts = [shard_clips(n, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gather_worker(rank, world, port, n_clips, L, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = shard_clips(n_clips, rank, world)
    local = torch.stack([torch.full((L,), float(i)) + torch.arange(L) * 1e-3 for i in ids])
    full = gather_pcm(local, ids, n_clips)
    full16 = gather_pcm(local / 8.0, ids, n_clips, as_int16=True)
    rooted = gather_pcm(local, ids, n_clips, root=0)           # north_star's "final gather": to rank 0 only
    rooted16 = gather_pcm(local / 8.0, ids, n_clips, as_int16=True, root=0)
    assert (rooted is None) == (rank != 0) and (rooted16 is None) == (rank != 0)
    dist.barrier()
    if rank == 0:
        assert torch.equal(rooted, full) and torch.equal(rooted16, full16)
        q.put((full.numpy(), full16.numpy()))
    dist.destroy_process_group()


def test_gather_pcm_world2_gloo():
    """The one collective of the sharded job (BASELINE configs[3]): clip i -> rank i % world, all_gather of the
    finished PCM, rank order undone.  gloo on CPU stands in for RCCL."""
    world, n_clips, L = 2, 6, 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, n_clips, L, q)) for r in range(world)]
    for p in procs:
        p.start()
    full, full16 = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.stack([np.full(L, float(i), np.float32) + np.arange(L, dtype=np.float32) * 1e-3 for i in range(n_clips)])
    assert np.array_equal(full, want)
    # the 16-bit form the reference writes (infer.py:70): round(x * 32767) clipped, converted before the exchange
    assert full16.dtype == np.int16
    assert np.array_equal(full16, np.clip(np.rint(want / np.float32(8.0) * np.float32(32767.0)), -32768, 32767).astype(np.int16))


def test_product_path_has_no_cpu_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from diffsvc_amd.pipeline import SvcPipeline
    hp = synth.tiny_hparams()
    with pytest.raises(RuntimeError, match="HIP device"):
        SvcPipeline(hp, synth.acoustic_state(hp, 0), {}, synth.tiny_vocoder())
    den = DiffNetHip(hp["audio_num_mel_bins"], hparams=hp)
    den.load_state_dict({k[len("denoise_fn."):]: v for k, v in synth.acoustic_state(hp, 0).items() if k.startswith("denoise_fn.")})
    with pytest.raises(RuntimeError):                                       # no device: the C ABI refuses, nothing falls back
        den(torch.zeros(1, 1, 16, 8), torch.zeros(1, dtype=torch.long), torch.zeros(1, 32, 8))


def test_product_sources_never_import_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "diff-svc_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert not re.search(r"^\s*(import|from)\s+(dsvc_oracle|refshim|oracle)\b", src, re.M), f
                assert "/root/reference" not in src, f


# ------------------------------------------------------------------------------------------------
# slicer (host-side integer work in front of the path: bit-exact bar)
def _slicer_kats():
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "slicer_kat.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("i", range(12))
def test_slicer_indices_bit_exact_vs_reference(i):
    from diffsvc_amd.slicer import Slicer
    kat = _slicer_kats()[i]
    case = kat["case"]
    audio = synth.slicer_audio(case)
    assert audio.shape[0] == kat["n_samples"]
    got = Slicer(sr=case["sr"], **case["args"]).slice(audio)
    assert got == kat["chunks"]


def test_slicer_chunks_tile_the_signal_and_reject_bad_windows():
    from diffsvc_amd.slicer import Slicer, cut_samples, chunks_of
    audio = synth.slicer_audio(synth.SLICER_CASES[3])
    chunks = cut_samples(np.stack([audio, audio]), 22050, db_thresh=-40)
    spans = [tuple(int(x) for x in v["split_time"].split(",")) for v in chunks.values()]
    assert spans[0][0] == 0 and spans[-1][1] == audio.shape[0]
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert sum(len(x) for _, x in chunks_of(chunks, audio)) == audio.shape[0]
    with pytest.raises(ValueError):
        Slicer(sr=22050, min_length=100, win_l=300)
    with pytest.raises(ValueError):
        Slicer(sr=22050, win_s=20, max_silence_kept=10)


def test_committed_bench_line_follows_the_driver_contract():
    """The last committed bench line (profiles/*_bench.json, produced by bench.py on the GPU box) carries every field the
    driver and the judge read: the metric contract, `roofline` and `cpu_baseline`."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(f for f in glob.glob(os.path.join(root, "profiles", "r*_bench.json")) if "prof" not in os.path.basename(f) and "train" not in os.path.basename(f))   # (*prof*_bench = lines printed under rocprofv3, *train* = bench.py --train)
    assert files, "no bench line committed under profiles/"
    with open(files[-1]) as f:
        d = json.loads(f.read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "audio-sec/wall-sec" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["n_gpus"] == 1 and "workload" in d["config"] and "model" not in d["config"] and d["data"] == "synthetic"
    r = d["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r) and r["bound"] in ("hbm", "mfma")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(c) and c["kind"] in ("port", "reference") and c["cores"] >= 1
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 10.0) < 0.05          # one 10 s clip per step


@pytest.mark.parametrize("tag,extra", [("linear", dict(schedule_type="linear", max_beta=0.02)), ("cosine", dict(schedule_type="cosine"))])
def test_drop_in_constructor_buffers_match_the_reference_constructor(tag, extra):
    """GaussianDiffusionHip.__init__ recomputes the 12 schedule buffers (float64 numpy, cast to fp32 last: diffusion.py:87-120) before a
    checkpoint overwrites them: value-checked bit for bit against the REAL constructor's buffers (tests/golden/schedule.npz)."""
    g = load_golden("schedule")
    hp = dict(synth.HPARAMS_44K, **extra)
    model = GaussianDiffusionHip(None, 128, DiffNetHip(128, hparams=hp), timesteps=1000, K_step=1000, loss_type="l2",
                                 spec_min=hp["spec_min"], spec_max=hp["spec_max"], hparams=hp)
    keys = [k[len(tag) + 1:] for k in g if k.startswith(tag + "_")]
    assert len(keys) == 12
    for k in keys:
        assert np.array_equal(getattr(model, k).numpy(), g[tag + "_" + k]), k


def _ddp_worker(rank, world, port, q):
    import torch.distributed as dist
    from diffsvc_amd.train import allreduce_mean_
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)            # this rank's "gradients"
    allreduce_mean_(flat)
    dist.barrier()
    if rank == 0:
        q.put(flat.numpy())
    dist.destroy_process_group()


def test_gradient_allreduce_mean_world2_gloo():
    """Data-parallel training's one collective (BASELINE configs[4]): the flat gradient buffer averaged over the ranks in place
    (what the reference's DDP reducer does, utils/pl_utils.py:187-221).  gloo on CPU stands in for RCCL."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(got, np.arange(1000, dtype=np.float32) * 1.5)


def test_hubert_checkpoint_keys_match_reference():
    """The synthetic HuBERT-soft checkpoint carries exactly the keys / shapes of the real HubertSoft.state_dict() (strict load)."""
    ref = _keys()["hubert_soft"]
    assert {k: list(v.shape) for k, v in synth.hubert_state(0).items()} == ref


def test_pe_checkpoint_keys_match_reference():
    """synth.pe_state mints exactly PitchExtractor().state_dict()'s keys and shapes (the reference loads it strictly, tts.py:113)."""
    ref = _keys()["pitch_extractor"]
    assert {k: list(v.shape) for k, v in synth.pe_state(dict(synth.HPARAMS_24K), 0).items()} == ref


def test_pe_expected_keys_and_position_table():
    """PitchExtractorHip's strict-load key table equals the real module's state_dict (names, shapes, order), and its sinusoid table is
    bit-identical to the oracle's restatement of SinusoidalPositionalEmbedding.get_embedding."""
    from diffsvc_amd.pe import expected_keys, position_table
    import dsvc_oracle as O
    ref = _keys()["pitch_extractor"]
    got = expected_keys(80, 256, 256, 2)
    assert {k: list(v) for k, v in got.items()} == ref
    assert list(got.keys()) == list(synth.pe_state(dict(synth.HPARAMS_24K), 0).keys())
    for rows, dim in ((4096, 256), (37, 32), (5000, 384)):
        assert torch.equal(position_table(rows, dim), O.sinusoid_table(rows, dim))


def test_coarse_pitch_thresholds_reproduce_f0_to_coarse():
    """cond.coarse_thresholds (what the device pitch kernel searches): 1 + #{k: x >= thr[k]} equals the reference expression
    f0_to_coarse(2**x) (utils/pitch_utils.py:17-31) at the thresholds themselves, one ulp below them, and on a dense random sample
    evaluated through torch's vectorised CPU path."""
    from diffsvc_amd.cond import coarse_thresholds, f0_to_coarse
    hp = dict(synth.HPARAMS_44K)
    thr = coarse_thresholds(hp).numpy()
    assert thr.shape == (hp["f0_bin"] - 2,) and np.all(np.diff(thr) > 0)
    for k in (0, 1, 100, 253):
        at = f0_to_coarse(2 ** torch.tensor([[float(thr[k])]]), hp).item()
        below = f0_to_coarse(2 ** torch.tensor([[float(np.nextafter(thr[k], np.float32(0)))]]), hp).item()
        assert (at, below) == (k + 2, k + 1)
    g = np.random.Generator(np.random.PCG64(3))
    x = torch.from_numpy(g.uniform(3.0, 11.0, 100003).astype(np.float32))[None]
    ref = f0_to_coarse(2 ** x, hp)[0].numpy()
    assert np.array_equal(ref, 1 + np.searchsorted(thr, x[0].numpy(), side="right"))


def test_formats_indexed_dataset_singer_features_and_wire_payload(tmp_path):
    """diffsvc_amd.formats: the binarised-dataset container round-trips (with its read cache), the singer-mode _mel.npy/_f0.npy pair
    lands where Svc.after_infer puts it, and the VST-bridge payload decodes / encodes (PCM-16 wav at the DAW's rate)."""
    import io
    import wave
    from diffsvc_amd import formats
    from diffsvc_amd.vocoder import read_wav
    path = str(tmp_path / "train")
    b = formats.IndexedDatasetBuilder(path)
    items = [{"i": i, "x": np.full((i + 1, 3), i, np.float32)} for i in range(5)]
    for it in items:
        b.add_item(it)
    b.finalize()
    ds = formats.IndexedDataset(path, num_cache=2)
    assert len(ds) == 5
    for i in (4, 4, 0, 2, 4, 1):
        assert ds[i]["i"] == i and np.array_equal(ds[i]["x"], items[i]["x"])
    assert len(ds._recent) == 2
    with pytest.raises(IndexError):
        ds[5]
    (tmp_path / "batch").mkdir()
    mel, f0 = np.ones((7, 4), np.float32), np.arange(7, dtype=np.float32)
    mp, fp = formats.save_singer_features(str(tmp_path / "batch" / "a.wav"), mel, f0)
    assert mp.endswith(os.path.join("singer_data", "a_mel.npy")) and np.array_equal(np.load(mp), mel) and np.array_equal(np.load(fp), f0)
    sr = 44100
    tone = (0.5 * np.sin(2 * np.pi * 440 * np.arange(sr // 10) / sr)).astype(np.float32)
    body = formats.encode_voice_change_response(tone, sr, sr)
    with wave.open(io.BytesIO(body.getvalue()), "rb") as w:
        assert (w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()) == (sr, 1, 2, tone.size)
    wav_io, key, daw_sr, spk = formats.decode_voice_change_request({"fPitchChange": "-3.0", "sampleRate": "48000.0", "sSpeakId": "0"}, body.getvalue())
    assert (key, daw_sr, spk) == (-3.0, 48000, 0)
    back = read_wav(wav_io, sr)
    assert back.shape == tone.shape and np.abs(back - tone).max() < 1.0 / 16384
    half = formats.encode_voice_change_response(tone, sr, 22050)
    with wave.open(io.BytesIO(half.getvalue()), "rb") as w:
        assert w.getframerate() == 22050 and abs(w.getnframes() - tone.size // 2) <= 1


def test_read_wav_mono_modes(tmp_path):
    """read_wav: 'first' keeps channel 0 (the 44.1 kHz nvSTFT loader, nvSTFT.py:14-44), 'mean' averages the channels
    (librosa.load(mono=True) on the 24 kHz front-end and in get_units); a mono file reads the same either way."""
    import wave
    from diffsvc_amd.vocoder import read_wav
    g = np.random.Generator(np.random.PCG64(3))
    st = (g.standard_normal((500, 2)) * 8000).astype("<i2")
    p2, p1 = str(tmp_path / "st.wav"), str(tmp_path / "mono.wav")
    for path, data, nch in ((p2, st, 2), (p1, st[:, :1], 1)):
        with wave.open(path, "wb") as w:
            w.setnchannels(nch); w.setsampwidth(2); w.setframerate(16000); w.writeframes(np.ascontiguousarray(data).tobytes())
    first, mean = read_wav(p2, 16000, mono="first"), read_wav(p2, 16000, mono="mean")
    assert first.dtype == np.float32 and first.shape == mean.shape == (500,)
    assert np.array_equal(first, st[:, 0].astype(np.float32) / 32768.0)
    assert np.allclose(mean, st.astype(np.float32).mean(axis=1) / 32768.0, atol=1e-7) and not np.allclose(first, mean)
    assert np.array_equal(read_wav(p1, 16000, mono="first"), read_wav(p1, 16000, mono="mean"))
    with pytest.raises(ValueError):
        read_wav(p1, 16000, mono="left")


def test_shard_clips_by_length_balances_and_is_deterministic():
    """Variable-length partitioning: every clip lands on exactly one rank, the heaviest rank is within one clip of the
    mean (LPT bound), equal-length clips reduce to an even split, and the result does not depend on who computes it."""
    from diffsvc_amd.pipeline import shard_clips, shard_clips_by_length
    g = np.random.Generator(np.random.PCG64(5))
    lens = g.integers(50, 2000, 257).tolist()
    for world in (1, 2, 8):
        parts = shard_clips_by_length(lens, world)
        assert sorted(i for p in parts for i in p) == list(range(len(lens)))
        loads = [sum(lens[i] for i in p) for p in parts]
        assert max(loads) - sum(lens) / world <= max(lens)
        assert parts == shard_clips_by_length(list(lens), world)
        for p in parts:
            assert [lens[i] for i in p] == sorted((lens[i] for i in p), reverse=True)
    even = shard_clips_by_length([861] * 256, 8)
    assert all(len(p) == 32 for p in even) and sorted(even[3]) == shard_clips(256, 3, 8)


def test_gradient_buckets_partition_the_flat_gradient_buffer():
    """diffsvc_amd.train.gradient_buckets: the slices the phased training step declares final (tail, groups of residual layers from the top
    down, head + pitch embedding) cover the flat buffer exactly once, in the order the C side finishes them."""
    from diffsvc_amd import synth
    from diffsvc_amd.train import gradient_buckets
    hp = synth.HPARAMS_44K
    M, H, C, L = hp["audio_num_mel_bins"], hp["hidden_size"], hp["residual_channels"], hp["residual_layers"]
    names = [("denoise_fn.input_projection.weight", C * M), ("denoise_fn.input_projection.bias", C), ("denoise_fn.mlp.0.weight", 4 * C * C),
             ("denoise_fn.mlp.0.bias", 4 * C), ("denoise_fn.mlp.2.weight", 4 * C * C), ("denoise_fn.mlp.2.bias", C)]
    for l in range(L):
        q = "denoise_fn.residual_layers.%d." % l
        names += [(q + "dilated_conv.weight", 2 * C * C * 3), (q + "dilated_conv.bias", 2 * C), (q + "diffusion_projection.weight", C * C),
                  (q + "diffusion_projection.bias", C), (q + "conditioner_projection.weight", 2 * C * H), (q + "conditioner_projection.bias", 2 * C),
                  (q + "output_projection.weight", 2 * C * C), (q + "output_projection.bias", 2 * C)]
    names += [("denoise_fn.skip_projection.weight", C * C), ("denoise_fn.skip_projection.bias", C), ("denoise_fn.output_projection.weight", M * C),
              ("denoise_fn.output_projection.bias", M), ("fs2.pitch_embed.weight", 300 * H)]
    layout, off = [], 0
    for n, k in names:
        layout.append((n, off, k)); off += k
    for per in (1, 5, 7, 20, 64):
        buckets = gradient_buckets(layout, L, per)
        assert buckets[0][0] == "begin" and buckets[-1][0] == "end"
        layer_calls = [(hi, lo) for ph, hi, lo, _ in buckets if ph == "layers"]
        assert layer_calls[0][0] == L and layer_calls[-1][1] == 0 and all(a[1] == b[0] for a, b in zip(layer_calls, layer_calls[1:]))
        cover = np.zeros(off, dtype=np.int32)
        for _, _, _, slices in buckets:
            for o, n in slices:
                cover[o:o + n] += 1
        assert (cover == 1).all(), per


@pytest.mark.parametrize("train", [False, True])
def test_bench_spawns_its_own_ranks_and_refuses_a_mismatched_world(train):
    """`python bench.py --gpus 2` with NO launcher around it must run two ranks (round 3 it silently measured one GPU and printed n_gpus: 1):
    bench.py re-executes itself under torch.distributed.run, as the reference spawns its DDP workers (utils/pl_utils.py:483-485).  Exercised
    without a GPU through the dry-run mode (DSVC_BENCH_DRY=1: rendezvous over gloo, the barrier-bracketed clock, max over ranks, one JSON line
    from rank 0 -- no hot path, value null).  A launcher that started another world size than --gpus asks for is refused with exit code 2."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["DSVC_BENCH_DRY"] = "1"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"] + (["--train"] if train else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] and d["train"] == train and d["rccl"]["world_size"] == 2 and d["rccl"]["backend"] == "gloo"
    assert d["ms_per_step"] >= 19.0                          # the slower rank's clock (rank 1 sleeps 20 ms per step), not rank 0's
    # round 5: an N > 1 line carries its own same-workload 1-GPU denominator (rank 0 alone while the others wait), so that nobody divides it
    # by the one-clip `--gpus 1` headline
    assert d["same_workload_1gpu"] > 0 and abs(d["scaling_efficiency"] * 2 - d["speedup_vs_1gpu_same_workload"]) < 1e-9
    bad = subprocess.run(cmd, capture_output=True, text=True, timeout=60, cwd=root, env=dict(env, WORLD_SIZE="3", RANK="0"))
    assert bad.returncode == 2 and "WORLD_SIZE=3" in bad.stderr and not [l for l in bad.stdout.splitlines() if l.startswith("{")]


def test_resblock_geometry_follows_the_reference_constructors():
    """ResBlock2 builds exactly two convs from dilation[0] and dilation[1] whatever the list's length (modules/nsf_hifigan/
    models.py:77-82), ResBlock1 three pairs from dilation[0..2] (:36-55).  A list that is too short raises in the reference's constructor
    (IndexError) -- here a ValueError before any device work; a longer list is accepted and its tail ignored: the synthetic checkpoint
    for a three-entry ResBlock2 config holds convs.0 / convs.1 only, like the real Generator's (tests/golden/vocoder_tiny_rb2_d3.npz runs it)."""
    from diffsvc_amd.engine import VocoderHandle
    h1 = dict(synth.tiny_vocoder(rds=((1,), (1,))), resblock="2")
    with pytest.raises(ValueError, match="needs 2 dilations"):
        VocoderHandle({}, h1)
    h2 = dict(synth.tiny_vocoder(rds=((1, 3), (1, 3))), resblock="1")
    with pytest.raises(ValueError, match="needs 3 dilations"):
        VocoderHandle({}, h2)
    sd = synth.vocoder_state(dict(synth.tiny_vocoder(rds=((1, 3, 5), (2, 4, 7))), resblock="2"), 7)
    assert not any(".convs.2." in k for k in sd) and any(".convs.1." in k for k in sd)


def test_auto_precision_follows_the_workspace_buckets():
    """`DiffNetHip.precision_for` mirrors the C library's layout rule (csrc/diffnet.hip: bucket_rows -- a clip occupies round_up(T + largest
    dilation, 128) rows since round 6) and its 48-tile threshold for the fused layer kernel: what `auto` picks for a call is what the handle
    will really run (f16_w6 only where the fused kernel takes the call).  Host logic only."""
    hp = dict(synth.HPARAMS_44K)
    net = DiffNetHip(128, hparams=hp)
    assert net.workspace_tiles(1, 861) == 7 and net.workspace_tiles(32, 861) == 224 and net.workspace_tiles(36, 861) == 252
    assert net.workspace_tiles(1, 888) == 7 and net.workspace_tiles(1, 889) == 8              # 888 + 8 = 896: the bucket's last frame
    assert net.precision_for("ddpm", 1, frames=861, clips=1) == "f16_x3t"
    assert net.precision_for("ddpm", 1, frames=6 * 861, clips=6) == "f16_x3t" and net.precision_for("ddpm", 1, frames=7 * 861, clips=7) == "f16_w6"
    assert net.precision_for("ddpm", 1, frames=2600, clips=1) == "f16_x3t"                    # 21 tiles
    assert net.precision_for("ddpm", 1, frames=6000, clips=1) == "f16_x3t" and net.precision_for("ddpm", 1, frames=6100, clips=1) == "f16_w6"   # 47 / 48 tiles
    assert net.precision_for("ddpm", 1, frames=3 * 2000, clips=3) == "f16_w6"                 # 3 x 16 tiles
    assert net.precision_for("plms", 20, frames=7000, clips=1) == "f16_x3t" and net.precision_for("forward") == "f16_x3t"
    tiny = DiffNetHip(16, hparams=synth.tiny_hparams())
    assert tiny.precision_for("ddpm", 1, frames=100000, clips=100) == "f16_x3t"               # 64 channels: no fused layer kernel for this architecture


def test_loud_norm_restatement_against_the_standards_conformance_point():
    """``loud_norm`` (preprocessing/data_gen_utils.py:117-122) calls pyloudnorm==0.1.0 (requirements.txt:67), which is absent here: its BS.1770
    integrated loudness is restated in diffsvc_amd/loudness.py -- parity unpinned at the dependency, pinned to the standard instead: a 997 Hz
    full-scale sine measures -3.01 LKFS (ITU-R BS.1770-4, conformance tolerance +-0.1 LU; the audio-EQ-cookbook biquads pyloudnorm designs for
    the signal's own rate give -3.05 ... -3.06), level changes map 1 : 1 to LU, silence below the absolute gate does not count, and normalising
    lands on the target."""
    from diffsvc_amd.loudness import integrated_loudness, loud_norm
    for rate in (48000, 44100, 24000):
        t = np.arange(rate * 5) / rate
        x = np.sin(2 * np.pi * 997 * t)
        L0 = integrated_loudness(x, rate)
        assert abs(L0 - (-3.01)) < 0.1, (rate, L0)
        assert abs(integrated_loudness(0.1 * x, rate) - (L0 - 20.0)) < 1e-6
        gated = np.concatenate([x, np.zeros(rate * 5)])                     # trailing silence: gated out, not averaged in (ungated: L0 - 3.01);
        assert L0 - 0.2 < integrated_loudness(gated, rate) <= L0            # the three blocks that straddle the edge pass the relative gate: -0.13 LU
        assert abs(integrated_loudness(np.stack([x, x], axis=1), rate) - (L0 + 10 * np.log10(2.0))) < 1e-6      # two channels: powers add
    y = loud_norm((0.05 * np.sin(2 * np.pi * 440 * np.arange(72000) / 24000)).astype(np.float32), 24000)
    assert y.dtype == np.float32 and abs(integrated_loudness(y, 24000) + 22.0) < 1e-3
    loud = loud_norm(np.sign(np.sin(2 * np.pi * 100 * np.arange(48000) / 24000)).astype(np.float32) * 0.001, 24000, target=-1.0)
    assert np.abs(loud).max() <= 1.0 + 1e-6                                 # a target that would clip: rescaled to full scale (data_gen_utils.py:121-122)
    with pytest.raises(ValueError):
        integrated_loudness(np.zeros(100), 24000)


def test_plan_chunks_groups_an_utterance_by_the_cost_model():
    """SvcPipeline.plan_chunks (host logic of infer_chunks): every chunk in exactly one group, groups contiguous in length order with the longest
    chunk first, no group beyond one round of 128-frame tiles, and the modelled time of the plan never above the one-by-one loop's."""
    import types
    from diffsvc_amd.pipeline import SvcPipeline
    from diffsvc_amd.denoiser import DiffNetHip
    from diffsvc_amd import synth
    hp = dict(synth.HPARAMS_44K)
    den = DiffNetHip(hp["audio_num_mel_bins"], hparams=hp, precision="auto")
    stub = types.SimpleNamespace(model=types.SimpleNamespace(denoise_fn=den), CHUNK_COST_FUSED=SvcPipeline.CHUNK_COST_FUSED,
                                 CHUNK_COST_SMALL=SvcPipeline.CHUNK_COST_SMALL, CHUNK_MAX_ROWS=SvcPipeline.CHUNK_MAX_ROWS)
    stub._chunk_group_cost = lambda lens, speedup=1: SvcPipeline._chunk_group_cost(stub, lens, speedup)
    rng = np.random.default_rng(5)
    cases = [[430, 700, 861, 1200, 1600, 2100, 2600], [861], [861, 861], [100, 7000], [2600] * 20] + [list(rng.integers(40, 3000, size=n)) for n in (3, 9, 17)]
    for lens in cases:
        plan = SvcPipeline.plan_chunks(stub, lens)
        assert sorted(i for g in plan for i in g) == list(range(len(lens))), (lens, plan)
        cost = 0.0
        for g in plan:
            assert all(lens[g[k]] >= lens[g[k + 1]] for k in range(len(g) - 1)), (lens, g)
            assert len(g) == 1 or den.workspace_tiles(len(g), int(lens[g[0]])) * 128 <= SvcPipeline.CHUNK_MAX_ROWS, (lens, g)
            cost += stub._chunk_group_cost([int(lens[i]) for i in g])
        assert cost <= sum(stub._chunk_group_cost([int(t)]) for t in lens) + 1e-6, (lens, plan)
    plan = SvcPipeline.plan_chunks(stub, cases[0])
    assert len(plan) < 7 and max(len(g) for g in plan) >= 3, plan        # the seven chunks of bench.py's `ragged`: batching pays
    big = SvcPipeline.plan_chunks(stub, [2600] * 20)                      # 21 active 128-frame tiles per chunk: no group beyond two rounds of workgroups
    assert all(len(g) * 21 <= 512 for g in big) and len(big) <= 3, big
