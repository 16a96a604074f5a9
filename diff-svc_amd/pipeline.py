"""Device-resident batched inference: content units + f0 -> cond -> 1000-step DDPM / PLMS -> NSF-HiFiGAN PCM.

This is the generalisation of the reference's sequential B=1 loop (batch.py:25-43, infer.py:45-67) that
BASELINE config 4 asks for.  Clips are independent, so a batch is sharded over ranks by
``clip i -> rank i % world`` with one collective at the very end: an all_gather of the finished PCM (RCCL over
xGMI under the 'nccl' backend, gloo in the CPU tests).  Nothing inside the hot loop communicates.
"""
import numpy as np
import torch

from .denoiser import DiffNetHip
from .engine import VocoderHandle
from .sampler import GaussianDiffusionHip


def shard_clips(n_clips, rank, world):
    """Static round-robin partition: the clip ids a rank owns."""
    return list(range(rank, n_clips, world))


def shard_clips_by_length(lengths, world):
    """Variable-length clips: longest-processing-time-first -- clips sorted by frame count, each handed to the rank with
    the least frames so far (ties: lowest rank).  Returns one list of clip ids per rank, each sorted by length so that a rank can cut its
    share into padded batches of similar clips; deterministic, so every rank computes the same partition without talking."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world
    parts = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        parts[r].append(i)
        load[r] += int(lengths[i])
    return parts


class SvcPipeline:
    """One process per GPU.  ``acoustic_state`` is a GaussianDiffusion state dict (no 'model.' prefix),
    ``vocoder_state``/``vocoder_cfg`` the NSF-HiFiGAN generator checkpoint and its config.json."""

    def __init__(self, hp, acoustic_state, vocoder_state, vocoder_cfg, precision="auto", vocoder_precision="f16_x3",
                 device="cuda", pe_state=None):
        if not torch.cuda.is_available():
            raise RuntimeError("SvcPipeline needs a HIP device (there is no CPU path)")
        self.hp = hp
        self.device = device
        den = DiffNetHip(hp["audio_num_mel_bins"], hparams=hp, precision=precision)
        self.model = GaussianDiffusionHip(None, hp["audio_num_mel_bins"], den, timesteps=hp["timesteps"], K_step=hp["K_step"],
                                          loss_type=hp.get("diff_loss_type", "l2"), spec_min=hp["spec_min"],
                                          spec_max=hp["spec_max"], hparams=hp)
        self.model.load_state_dict(acoustic_state, strict=True)
        self.model.to(device)
        self.vocoder = VocoderHandle(vocoder_state, vocoder_cfg, precision=vocoder_precision)
        self.pe = None
        if pe_state is not None:                                    # Svc.__init__: PitchExtractor().cuda() + strict load (infer_tool.py:134-136)
            from .pe import PitchExtractorHip
            self.pe = PitchExtractorHip(n_mel_bins=hp["audio_num_mel_bins"], hparams=hp).cuda()
            self.pe.load_state_dict(pe_state, strict=True)
            self.pe.eval()

    @torch.no_grad()
    def infer(self, hubert, mel2ph, f0, speedup=1, seed=0, first_clip=0, clip_ids=None, use_graph=True, return_mel=False,
              return_lens=False, use_pe=False, full_length=None):
        """hubert [B,N,H], mel2ph [B,T] long, f0 [B,T] log2 (interpolated) -- all device tensors.
        Returns PCM [B, T*hop] on the device (and mel [B,T,M] / the per-clip sample counts if asked).

        Clips of different lengths are padded to a common T with ``mel2ph == 0`` frames at the end.  The reference runs every
        clip alone (B=1, infer_tool.py:277), so a padded batch must equal the per-clip runs: trailing padded frames are the
        convs' ZERO PADDING inside the sampler (``clip_lens``), and the host glue of ``Svc.after_infer``
        (infer_tool.py:177-191) is reproduced per clip -- frames whose predicted mel row is all-zero are dropped (``mel_out``
        is masked by ``mel2ph > 0``, diffusion.py:280-281, so those are exactly the ``mel2ph == 0`` frames), the mel is clipped to
        [mel_vmin, mel_vmax], f0 is cut with the same mask -- before the vocoder sees it.  PCM rows are zero beyond a clip's own
        ``kept_frames * hop`` samples.

        ``use_pe``: drive the vocoder with the pitch extractor's f0 read off the sampled mel instead of the input f0
        (Svc.infer's ``use_pe``, infer_tool.py:165-168; needs ``pe_state`` at construction).

        ``full_length``: True = the caller guarantees that no clip carries padding frames (fixed-length batches): nothing is read back
        from the device before the work is enqueued.  None (default) = look (one 1-byte device-to-host read of ``(mel2ph > 0).all()``)."""
        if use_pe and self.pe is None:
            raise RuntimeError("use_pe=True needs a pitch-extractor checkpoint (SvcPipeline(..., pe_state=...))")
        hp = dict(self.hp, pndm_speedup=speedup)
        self.model.hp = hp
        self.model.fs2.hp = hp
        B, T = mel2ph.shape
        valid = mel2ph > 0
        ragged = False if full_length else not bool(valid.all().item())       # (one tiny D2H before anything is launched)
        clip_lens = lens_host = None
        if ragged:
            ar = torch.arange(1, T + 1, device=mel2ph.device)
            last = (valid * ar).amax(dim=1)                      # frames up to the last content frame (interior gaps stay inside)
            clip_lens = last.clamp(min=1).to(torch.int32)
            lens_host = clip_lens.tolist()                       # (the ragged path has read the device once already; the sampler schedules by these)
        ret = self.model(hubert, mel2ph=mel2ph, f0=f0.clone(), infer=True, seed=seed, first_clip=first_clip, clip_ids=clip_ids,
                         clip_lens=clip_lens, clip_lens_host=lens_host, use_graph=use_graph)
        mel = ret["mel_out"]
        mel_c = torch.clamp(mel, hp["mel_vmin"], hp["mel_vmax"])
        hop = self.vocoder.hop
        if use_pe:
            self.pe.hp = hp
            f0_hz = self._pe_f0(mel, clip_lens)
        else:
            f0_hz = ret["f0_denorm"]
        if not ragged:
            wav = self.vocoder.vocode(mel_c, f0_hz, seed=seed, first_clip=first_clip, clip_ids=clip_ids)
            lens = torch.full((B,), T * hop, dtype=torch.int64, device=wav.device)
        else:
            wav, lens = self._vocode_ragged(mel_c, f0_hz, valid, seed, first_clip, clip_ids)
        out = (wav,)
        if return_mel:
            out += (mel,)
        if return_lens:
            out += (lens,)
        return out if len(out) > 1 else wav

    @torch.no_grad()
    def infer_job(self, hubert, mel2ph, f0, clips_per_batch=32, speedup=1, seed=0, first_clip=0, clip_ids=None, use_graph=True, overlap=True):
        """A job of MANY fixed-length clips on ONE device -- the reference's sequential loop over its inputs (batch.py:25-43) as batches of
        ``clips_per_batch`` through the same sampler and vocoder: the 1-GPU side of north_star's "speed-up at 8 GPUs vs 1 GPU on the batched
        config" (the whole 256-clip job on one device, not one rank's share).  hubert [N,n,H], mel2ph [N,T], f0 [N,T] on the device, no padding
        frames (``full_length``); returns PCM [N, T*hop].

        ``overlap``: the vocoder of batch k runs on a second stream while the sampler of batch k+1 has the main stream -- a fused layer launch
        of 32 ten-second clips holds 224 of the 256 CUs for the whole DDPM loop, the generator's kernels fit beside it.  Every clip's noise
        streams are keyed by its global id, and neither the sampler nor the vocoder mixes clips or batches: the job is BIT-IDENTICAL to
        separate ``infer`` calls per batch (tests/test_gpu_pipeline.py), overlapped or not."""
        N, T = mel2ph.shape
        hop = self.vocoder.hop
        hp = dict(self.hp, pndm_speedup=speedup)
        self.model.hp = hp
        self.model.fs2.hp = hp
        if clip_ids is None:
            clip_ids = torch.arange(first_clip, first_clip + N, dtype=torch.int32, device=mel2ph.device)
        out = torch.empty(N, T * hop, device=mel2ph.device, dtype=torch.float32)
        main = torch.cuda.current_stream()
        if overlap and getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream()
        keep = []                                                    # tensors the side stream still reads
        for lo in range(0, N, clips_per_batch):
            hi = min(N, lo + clips_per_batch)
            ids = clip_ids[lo:hi].contiguous()
            ret = self.model(hubert[lo:hi], mel2ph=mel2ph[lo:hi], f0=f0[lo:hi].clone(), infer=True, seed=seed, clip_ids=ids, use_graph=use_graph)
            mel_c = torch.clamp(ret["mel_out"], hp["mel_vmin"], hp["mel_vmax"])
            f0_hz = ret["f0_denorm"]
            if not overlap:
                out[lo:hi] = self.vocoder.vocode(mel_c, f0_hz, seed=seed, clip_ids=ids)
                continue
            ready = torch.cuda.Event()
            ready.record(main)
            with torch.cuda.stream(self._side):
                self._side.wait_event(ready)
                out[lo:hi] = self.vocoder.vocode(mel_c, f0_hz, seed=seed, clip_ids=ids)      # (stream_ptr(): the side stream)
            keep.append((mel_c, f0_hz, ids))
        if overlap:
            main.wait_stream(self._side)
            for group in keep:
                for t in group:
                    t.record_stream(self._side)
        return out

    # ---- an utterance's chunks as a few padded batches (round 6, second session) ----
    # The reference's driver runs the chunks the slicer cut one after another at B = 1 (infer.py:44-67, infer_tool.py:155-159).  They are
    # independent, and a padded batch equals the per-clip runs (``infer``: trailing ``mel2ph == 0`` frames are the convs' zero padding, the host
    # glue is applied per clip), so a caller that has all the chunks in hand can trade the single clip's latency regime for the batched one.
    # Cost model of one DDPM evaluation, in us (csrc/diffnet.hip: fused_nt's table of the fused layer kernel -- 45 / 65 / 125 us per layer on
    # 32- / 64- / 128-frame tiles with one workgroup per CU -- and a straight line through measured small batches for the small tilings, whose
    # kernels are latency-bound at one clip: three ten-second clips in one call run at 51.8x RT, six at 63.2x, against 25.8x one by one, at
    # the SAME fp32-class operand scheme):
    CHUNK_COST_FUSED = ((45.0, 32), (65.0, 64), (125.0, 128))
    CHUNK_COST_SMALL = (280.0, 0.12)                      # a + b * rows of the active tiles: one clip of 861 frames 0.386 ms per step, three 0.57, six 0.93 (profiles/r6u_chunks_small.txt);
                                                          # ragged groups: profiles/r6ac_chunks.txt (the seven chunks as ONE PLMS batch 807x RT, as two 649x, one by one 495x)
    CHUNK_MAX_ROWS = 2 * 256 * 128                         # the padded rectangle of a group (its workspace); what it costs is its active tiles

    def _chunk_group_cost(self, lens, speedup=1):
        """One evaluation of the padded batch of chunks of these lengths (longest first), us.  PLMS (speedup > 1) stays on the split-operand
        tilings at any size (DiffNetHip.AUTO): the line of the small tilings, whose slope also fits the batched f16_x3t rate (0.17 us per row)."""
        den = self.model.denoise_fn
        B, T = len(lens), int(lens[0])
        rows = den.workspace_tiles(B, T) * 128
        if speedup <= 1 and rows // 128 >= den.BATCHED_TILES and den.precision_for("ddpm", 1, frames=B * T, clips=B) in ("f16_w6", "f16_w6n"):
            # the fused kernel's workgroups on tiles beyond a clip's length return at once: a launch costs the rounds of its ACTIVE tiles (csrc/tlayer.h)
            per_layer = min(c * -(-sum(-(-int(n) // w) for n in lens) // 256) for c, w in self.CHUNK_COST_FUSED)
            return den.n_layers * per_layer + 40.0 + 0.0016 * float(sum(lens))
        # the two-launch tilings skip frame tiles beyond a clip's length too (csrc/tgemm.h: TGemmArgs::skip_hi; 64-frame tiles from 48 tiles of
        # 128 rows, 32-frame tiles below): the line is walked with the rows of the ACTIVE tiles
        w = 64 if rows // 128 >= den.BATCHED_TILES else 32
        a, b = self.CHUNK_COST_SMALL
        return a + b * min(rows, sum(-(-int(n) // w) * w for n in lens))

    def plan_chunks(self, lengths, speedup=1):
        """Groups of chunk indices (each group = one padded batch, longest chunk first) that minimise the modelled time of one evaluation
        over the whole utterance: chunks sorted by length, contiguous groups, dynamic programme over the cut points."""
        order = sorted(range(len(lengths)), key=lambda i: -int(lengths[i]))
        n = len(order)
        best = [0.0] + [float("inf")] * n
        cut = [0] * (n + 1)
        for i in range(1, n + 1):
            for j in range(max(0, i - 64), i):               # group = order[j:i], padded to the length of order[j]; at most 64 chunks per batch
                B, T = i - j, int(lengths[order[j]])
                if B > 1 and self.model.denoise_fn.workspace_tiles(B, T) * 128 > self.CHUNK_MAX_ROWS:
                    continue
                c = best[j] + self._chunk_group_cost([int(lengths[k]) for k in order[j:i]], speedup)
                if c < best[i]:
                    best[i], cut[i] = c, j
        groups, i = [], n
        while i > 0:
            groups.append(order[cut[i]:i])
            i = cut[i]
        return groups[::-1]

    @torch.no_grad()
    def infer_chunks(self, chunks, speedup=1, seed=0, first_clip=0, use_graph=True, use_pe=False, batch=True):
        """The chunks of ONE utterance -- a list of (hubert [n_i, H], mel2ph [T_i], f0 [T_i]) device tensors, every chunk with its own length --
        to a list of PCM tensors [kept_frames_i * hop], in the order given.  Chunk i draws the noise streams of clip ``first_clip + i`` whatever
        batch it lands in, so the result does not depend on the grouping beyond the operand precision `auto` picks by call size.
        ``batch=False`` runs them one by one as the reference's loop does.  PLMS chunks are batched too: the reference's own PLMS loop only
        works for B = 1 (``max(t - interval, 0)`` on the step tensor, diffusion.py:186), the sampler here runs it per clip of a batch exactly
        as alone (tests/test_gpu_pipeline.py::test_ragged_batch_equals_per_clip_reference_runs), at the same split-operand precision."""
        n = len(chunks)
        lengths = [int(c[1].shape[-1]) for c in chunks]
        groups = self.plan_chunks(lengths, speedup) if (batch and n > 1) else [[i] for i in range(n)]
        out = [None] * n
        dev = chunks[0][1].device
        for g in groups:
            Tm = max(lengths[i] for i in g)
            Nm = max(int(chunks[i][0].shape[-2]) for i in g)
            H = int(chunks[g[0]][0].shape[-1])
            hub = torch.zeros(len(g), Nm, H, device=dev, dtype=torch.float32)
            m2p = torch.zeros(len(g), Tm, device=dev, dtype=torch.long)
            f0 = torch.zeros(len(g), Tm, device=dev, dtype=torch.float32)
            for b, i in enumerate(g):
                h, m, f = chunks[i]
                hub[b, :h.shape[-2]] = h.reshape(-1, H)
                m2p[b, :lengths[i]] = m.reshape(-1)
                f0[b, :lengths[i]] = f.reshape(-1)
            ids = torch.tensor([first_clip + i for i in g], dtype=torch.int32, device=dev)
            wav, lens = self.infer(hub, m2p, f0, speedup=speedup, seed=seed, clip_ids=ids, use_graph=use_graph, return_lens=True, use_pe=use_pe)
            lens = lens.tolist()
            for b, i in enumerate(g):
                out[i] = wav[b, :lens[b]]
        return out

    def check(self):
        """The deferred argument checks of the device path, at a point where the caller synchronises anyway (after the PCM has been read, at the
        end of a job): mel2ph entries outside [0, content frames] -- the reference's torch.gather raises an IndexError at once (fs2.py:100-102),
        the one-launch device builder keeps a sticky flag instead of a device-to-host read per call -- and diffusion steps outside the schedule
        on a DiffNetHip.forward() seam (extract() would raise, diffusion.py:22-25).  Synchronises; raises what the reference raises."""
        self.model.fs2.check_alignment()
        for h, _ in self.model.denoise_fn._handles.values():
            h.check()

    def _pe_f0(self, mel, clip_lens):
        """``self.pe(outputs['mel_out'])['f0_denorm_pred']`` per clip as the reference's B=1 loop sees it: a clip padded to the batch's T
        must not let the padding frames into its GroupNorm statistics, so clips run at their own length (grouped by length)."""
        if clip_lens is None:
            return self.pe(mel)["f0_denorm_pred"]
        B, T, _ = mel.shape
        f0 = torch.zeros(B, T, device=mel.device, dtype=torch.float32)
        lens = clip_lens.tolist()
        for n in sorted(set(lens)):
            members = torch.tensor([b for b in range(B) if lens[b] == n], device=mel.device)
            f0[members, :n] = self.pe(mel[members, :n].contiguous())["f0_denorm_pred"]
        return f0

    def _vocode_ragged(self, mel_c, f0_hz, valid, seed, first_clip, clip_ids):
        """after_infer's frame drop per clip, then one vocoder call per group of equal kept length."""
        B, T, M = mel_c.shape
        hop = self.vocoder.hop
        kept = valid.sum(dim=1).tolist()
        ids = clip_ids.tolist() if clip_ids is not None else [first_clip + b for b in range(B)]
        wav = torch.zeros(B, T * hop, device=mel_c.device, dtype=torch.float32)
        groups = {}
        for b, n in enumerate(kept):
            if n > 0:
                groups.setdefault(n, []).append(b)
        for n, members in sorted(groups.items()):
            mg = torch.stack([mel_c[b][valid[b]] for b in members])             # [g, n, M]: mel_pred[mel_pred_mask]
            fg = torch.stack([f0_hz[b][valid[b]] for b in members])             # f0_pred[mel_pred_mask]
            gid = torch.tensor([ids[b] for b in members], dtype=torch.int32, device=mel_c.device)
            w = self.vocoder.vocode(mg.contiguous(), fg.contiguous(), seed=seed, clip_ids=gid)
            wav[torch.tensor(members, device=wav.device), :n * hop] = w
        return wav, torch.tensor([n * hop for n in kept], dtype=torch.int64, device=wav.device)


def chunk_group_cost(den, lens, speedup=1, fused=SvcPipeline.CHUNK_COST_FUSED, small=SvcPipeline.CHUNK_COST_SMALL):
    """SvcPipeline._chunk_group_cost for a bare DiffNetHip (the reference-side helper diffsvc_amd.svc_chunks has no SvcPipeline)."""
    import types
    stub = types.SimpleNamespace(model=types.SimpleNamespace(denoise_fn=den), CHUNK_COST_FUSED=fused, CHUNK_COST_SMALL=small)
    return SvcPipeline._chunk_group_cost(stub, lens, speedup)


def plan_chunk_groups(den, lengths, speedup=1):
    """SvcPipeline.plan_chunks for a bare DiffNetHip."""
    import types
    stub = types.SimpleNamespace(model=types.SimpleNamespace(denoise_fn=den), CHUNK_COST_FUSED=SvcPipeline.CHUNK_COST_FUSED,
                                 CHUNK_COST_SMALL=SvcPipeline.CHUNK_COST_SMALL, CHUNK_MAX_ROWS=SvcPipeline.CHUNK_MAX_ROWS)
    stub._chunk_group_cost = lambda lens, sp=1: SvcPipeline._chunk_group_cost(stub, lens, sp)
    return SvcPipeline.plan_chunks(stub, lengths, speedup)


def pcm16(wav):
    """fp32 PCM in [-1, 1] -> int16 the way the reference's writer stores it (``soundfile.write(..., 'PCM_16')``, infer.py:70: libsndfile
    scales by 0x7FFF, rounds to nearest and clips)."""
    return torch.clamp(torch.round(wav * 32767.0), -32768.0, 32767.0).to(torch.int16)


def gather_pcm(local_wav, clip_ids, n_clips, group=None, as_int16=False, root=None):
    """The one collective of the sharded job: every rank contributes its finished PCM [n_local, L]; rank order
    is undone so the result is indexed by clip id.  Equal counts per rank are required (pad the batch).
    ``as_int16``: convert to the 16-bit PCM the reference writes BEFORE the exchange -- half the bytes on the xGMI links (226 MB
    instead of 451 MB for 256 clips); the result is int16.
    ``root``: None = ``all_gather`` (every rank ends up with the whole job's PCM); an int = north_star's "final gather" -- ``dist.gather`` to that
    rank only (one eighth of the bytes on the links: each peer sends its 56 MB share once, over its own xGMI link to the root); the root gets
    the [n_clips, L] tensor, every other rank None."""
    import torch.distributed as dist
    if as_int16:
        local_wav = pcm16(local_wav)
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    if world == 1:
        return local_wav
    dev = local_wav.device
    if dist.get_backend(group) == "gloo" and local_wav.is_cuda:     # CPU test rigs: gloo gathers host tensors
        local_wav = local_wav.cpu()
    dtype = local_wav.dtype
    wire = local_wav.contiguous()
    if dtype == torch.int16:                                        # neither RCCL nor gloo has a 16-bit integer type: the gather only moves bytes
        wire = wire.view(torch.uint8)
    if root is not None:
        me = dist.get_rank(group)
        out = [torch.empty_like(wire) for _ in range(world)] if me == root else None
        dist.gather(wire, out, dst=root, group=group)
        if me != root:
            return None
    else:
        out = [torch.empty_like(wire) for _ in range(world)]
        dist.all_gather(out, wire, group=group)
    full = torch.empty(n_clips, local_wav.shape[1], dtype=dtype, device=dev)
    for r in range(world):
        ids = shard_clips(n_clips, r, world)
        full[ids] = out[r].view(dtype)[:len(ids)].to(dev)
    return full
