"""bench.py -- audio-seconds per wall-second of the diff-svc hot path on MI355X.

One "step" = one pass of the whole hot path (cond -> K-step DDPM through the 20-layer DiffNet -> NSF-HiFiGAN
PCM) over one batch of synthetic fixed-length 10 s / 44.1 kHz clips that are already resident in HBM.
Default workload = BASELINE.json configs[1]: a single clip per GPU, full 1000-step DDPM.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 3 --warmup 1

Rank 0 prints ONE JSON line (see README of the task for the contract) with `roofline` and `cpu_baseline`.
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import diffsvc_amd  # noqa: F401
from diffsvc_amd import synth
from diffsvc_amd.pipeline import SvcPipeline, gather_pcm, shard_clips

CLIP_SECONDS = 10.0
T_FRAMES = 861            # floor((441000 - 512) / 512) + 1   (nvSTFT.py:92-96)
N_UNITS = 500
FLOP_PER_FRAME_DILATED = 2 * 384 * 768 * 3        # the k=3 dilated conv of one residual layer
FLOP_PER_FRAME_OUTPROJ = 2 * 384 * 768            # ... and its 1x1 output projection (residual + skip halves)
PEAK_TFLOPS_F16 = 2500.0                           # dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md)
FAST_SIDE = "f16_w6n"                              # the faster operand scheme reported beside the shipped one (see `faster_scheme`)
# Per-clip maximum mel error of the shipped batched precision (f16_w6) over the 64 real-reference goldens of two 32-clip batches
# (tests/test_gpu_headline.py::test_batch_of_32_full_chain_every_clip_with_a_golden[random|random2-shipped]): Gumbel fit (mu, beta) of the 64
# maxima -> P(a clip exceeds the 1e-3 bar) and P(a 256-clip job holds such a clip).  Not a constant of this file any more: read from
# profiles/b32_error_fit.json (tools/b32_error_fit.py writes it from the test's output), which carries the hash of the kernel sources it was
# measured on -- reported as null, with the reason, once a kernel source has changed (load_error_fit).
PEAK_HBM_GBS = 8000.0                              # HBM3E peak (MI355X_MICROARCH.md); ~6300 GB/s is what a streaming copy reaches
# algorithmic HBM bytes per frame of one residual layer (design/tgemm.md), C = 384, fp16 operands / fp32 residual + skip + cproj:
#   gate kernel alone : xh in 768*(1 + 2d/128 averaged over d = 1,2,4,8 -> 1.06) + cproj 3072 + g out 768
#   fused layer kernel: xh in 814 + cproj 3072 + x32 in/out 3072 + skip in/out 3072 + next xh out 768   (g never leaves the CU)
BYTES_PER_FRAME_GATE = 814 + 3072 + 768
BYTES_PER_FRAME_LAYER = 814 + 3072 + 3072 + 3072 + 768
WEIGHT_BYTES_GATE = 768 * 1152 * 2
WEIGHT_BYTES_OUT = 768 * 384 * 2                   # one fp16 plane of the output 1x1 (f16_mN / f16_w2 stream hi + lo: twice that)


def probe_mfma():
    """dsvc_probe_mfma (csrc/probe.hip): what the chip sustains RIGHT NOW on dense fp16 MFMA with random operands (the clock drops under
    that load; the datasheet peak assumes it does not)."""
    import ctypes
    from diffsvc_amd import _lib
    o = (ctypes.c_float * 4)()
    _lib.check(_lib.lib().dsvc_probe_mfma_detail(1, o, _lib.stream_ptr()))
    torch.cuda.synchronize()
    # `tflops` = FLOPs / mean in-loop time (what the matrix pipes sustain once every wave is in its loop); `tflops_wall` divides by the
    # kernel's wall time (HIP events), which also carries launch skew between the CUs and the slowest CU's clock
    return {"tflops": o[1], "tflops_wall": o[0], "clock_ghz": o[2], "clock_ghz_min_cu": o[3]}


def latency_floor(precision, measured_us):
    """What bounds the single-clip gate kernel is latency, not the 2.5 PF/s roof: 1.5 GFLOP per launch on 224 workgroups.  A stated floor
    for ONE launch inside the replayed graph, from the measured constants of this part (design/tgemm.md; tools/micro/launch_floor.hip,
    profiles/r02h_stamps.txt, MI355X_MICROARCH.md):
      boundary   1.75 us  a graph node of an empty kernel (launch_floor.hip)
      tile DMA   the workgroup's (32 + 2 d) x row time tile through the CU's 64 B/clk vector-memory path + one L2 round trip (~0.3 us)
      stream     the 9 waves of a workgroup pull 3 output tiles x 72 k-steps x planes x 1 KiB of weight fragments through the same 64 B/clk
                 path (every wave streams its own weights: nothing is shared in LDS)
      matrix     per SIMD: 9 waves / 4 SIMDs x 24 k-steps x (MFMAs per product) x 32 cycles -- overlaps the stream, the larger of the two counts
      tail       split-K reduction through LDS + gate epilogue + store drain, ~0.5 us (stamps)
    at the 2.4 GHz the chip holds in this regime (the matrix pipe is mostly idle)."""
    clk = 2.4e3                                   # cycles per us
    # f16_x3t (round 4): the lo plane of a k16 step is 384 B of fp6 codes instead of a 1 KiB fp16 fragment, and of the three products one is a
    # K = 64 six-bit MFMA (a quarter of four fp16 ones): 2.25 units per product
    planes = 1.375 if precision == "f16_x3t" else (2 if precision == "f16_w2" else 1)
    mpp = 2.25 if precision == "f16_x3t" else planes
    row_bytes = 384 * 2 * (2 if precision == "f16_x3t" else 1)
    dma = (32 + 2 * 3.75) * row_bytes / 64.0 / clk + 0.3            # mean dilation of the 1, 2, 4, 8 cycle
    stream = 3 * 72 * planes * 1024 / 64.0 / clk
    matrix = (9 / 4.0) * 24 * mpp * 32 / clk
    floor = 1.75 + dma + max(stream, matrix) + 0.5
    return {"latency_floor_us": floor, "latency_floor_parts_us": {"boundary": 1.75, "tile_dma": dma, "weight_stream": stream, "matrix": matrix, "tail": 0.5},
            "frac_of_latency_floor": floor / measured_us}


def res_skip_roofline(handle, precision):
    """The single clip's OTHER layer kernel (39 % of the headline's GPU time had no roofline entry): tgemm_kernel<TEpiResSkip>,
    the output 1x1 (K = 384, M = 768: 589 824 FLOP per frame) with the residual / skip read-modify-write and the next layer's FiLM'd operand in
    its epilogue -- 24 output tiles, three per workgroup, each tile's K loop split over three waves (8 k16 steps per wave).  Priced like the gate
    kernel: algorithmic FLOPs against the MFMA peak, and against a stated latency floor (same constants as latency_floor: a 1.75 us graph-node
    boundary, the (32-frame x [hi | lo]) tile through the CU's 64 B/clk path + one L2 round trip, the larger of the per-CU weight stream and the
    per-SIMD matrix time, 0.5 us of split-K reduction + epilogue + store drain).  PMC traffic: profiles/resskip_traffic.json."""
    handle.den.debug_set("profile_kernel", 1)
    try:
        us, rows, kind = handle.profile_gate_kernel(1, T_FRAMES, 5)
    finally:
        handle.den.debug_set("profile_kernel", 0)
    x3t = precision == "f16_x3t"
    ach = FLOP_PER_FRAME_OUTPROJ * T_FRAMES / (us * 1e-6) / 1e12
    planes = 1.375 if x3t else (2 if precision == "f16_w2" else 1)
    mpp = 2.25 if x3t else planes
    clk = 2.4e3
    dma = 32 * 384 * 2 * (2 if x3t else 1) / 64.0 / clk + 0.3
    stream = 3 * 24 * planes * 1024 / 64.0 / clk
    matrix = (9 / 4.0) * 8 * mpp * 32 / clk
    floor = 1.75 + dma + max(stream, matrix) + 0.5
    act = 768 * (2 if x3t else 1)                         # g in, next xh out: fp16 rows, [hi | lo] planes at f16_x3t
    roof = {"bound": "mfma", "kernel": "tgemm_kernel<TEpiResSkip> (output 1x1 + residual / skip update + next layer's FiLM'd operand, one residual layer)",
            "achieved": ach, "peak": PEAK_TFLOPS_F16, "unit": "TFLOP/s", "frac": ach / PEAK_TFLOPS_F16, "avg_launch_us": us,
            "frames_per_launch": T_FRAMES, "mfma_per_product": mpp, "pipe_tflops": ach * mpp, "pipe_frac": ach * mpp / PEAK_TFLOPS_F16,
            "algorithmic_bytes": int((act + 3072 + 3072 + act) * T_FRAMES + planes * WEIGHT_BYTES_OUT),
            "latency_floor_us": floor, "latency_floor_parts_us": {"boundary": 1.75, "tile_dma": dma, "weight_stream": stream, "matrix": matrix, "tail": 0.5},
            "frac_of_latency_floor": floor / us}
    roof["traffic"], roof["traffic_source"] = load_traffic("resskip_traffic.json", precision)
    return roof


def train_step_flops(hp, frames):
    """Algorithmic FLOPs (one multiply-add = 2) of one training step over `frames` valid mel frames: per residual layer the forward (dilated
    conv, conditioner projection, output 1x1), the data gradients (transposed conv, d gate) and the three weight gradients; the conditioner's
    data gradient is never formed (design/training.md).  Tail: input / skip / output projections, forward + data + weight gradients."""
    C, H, M, L = hp["residual_channels"], hp["hidden_size"], hp["audio_num_mel_bins"], hp["residual_layers"]
    conv, outp, cond = 2 * C * 2 * C * 3, 2 * C * 2 * C, 2 * H * 2 * C
    per_layer = (conv + cond + outp) + (conv + outp) + (conv + cond + outp)
    tail = 3 * 2 * (M * C + C * C + C * M)
    return (per_layer * L + tail) * frames


VOCODER_FLOP_PER_FRAME = 649.5e6                   # NSF-HiFiGAN generator, 44.1 kHz config (design/other_kernels.md): 512 output samples per mel frame


def dominant_kernel_roofline(handle, B, precision):
    """HIP-event timing of the dominant kernel at this batch size (dsvc_sampler_profile_gate_kernel) against its roofline.
    Small batches run a layer as two launches and the gate kernel dominates (MFMA-shaped: 1.77 MFLOP per frame); the throughput
    tiling runs the whole layer as one kernel whose HBM bytes (10.8 KB per frame) take longer at the HBM peak than its 2.36 MFLOP
    per frame take at the MFMA peak: that kernel is priced against the HBM roof, with the MFMA fraction beside it."""
    us, rows, kind = handle.profile_gate_kernel(B, T_FRAMES, 5 if B == 1 else 3)
    frames = B * T_FRAMES
    if kind == 0:
        ach = FLOP_PER_FRAME_DILATED * frames / (us * 1e-6) / 1e12
        roof = {"bound": "mfma", "kernel": "tgemm_kernel<TEpiGate> (dilated k=3 conv + hoisted cond projection + gate, one residual layer)",
                "achieved": ach, "peak": PEAK_TFLOPS_F16, "unit": "TFLOP/s", "frac": ach / PEAK_TFLOPS_F16,
                "avg_launch_us": us, "frames_per_launch": frames, "traffic": None,
                "algorithmic_bytes": int(((814 * 2 + 3072 + 768 * 2) if precision == "f16_x3t" else BYTES_PER_FRAME_GATE) * frames
                                         + (1.375 if precision == "f16_x3t" else (2 if precision == "f16_w2" else 1)) * WEIGHT_BYTES_GATE)}
        tfile = {1: "gate_traffic.json", 32: "gate_traffic_b32.json"}.get(B)
        if B == 1:
            roof.update(latency_floor(precision, us))
    else:
        w6 = precision in ("f16_w6", "f16_w6n")
        # f16_w6: one fp16 plane + the w_lo plane as 6-bit codes (0.375 of a plane) per contraction; the g_lo correction streams the output
        # projection's weights once more as codes
        out_planes = (1.375 + (0.375 if precision == "f16_w6" else 0.0)) if w6 else (2 if (precision.startswith("f16_m") or precision == "f16_w2") else 1)
        gate_planes = 1.375 if w6 else (2 if precision == "f16_w2" else 1)
        nbytes = int(BYTES_PER_FRAME_LAYER * frames + gate_planes * WEIGHT_BYTES_GATE + out_planes * WEIGHT_BYTES_OUT)
        ach = nbytes / (us * 1e-6) / 1e9
        tf = (FLOP_PER_FRAME_DILATED + FLOP_PER_FRAME_OUTPROJ) * frames / (us * 1e-6) / 1e12
        roof = {"bound": "hbm", "kernel": "tlayer_kernel (one residual layer in one launch: dilated conv + cond projection + gate -> g in LDS -> "
                                         "output 1x1 + residual/skip update + next layer's FiLM'd fp16 operand)",
                "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS,
                "avg_launch_us": us, "frames_per_launch": frames, "traffic": None, "algorithmic_bytes": nbytes,
                "mfma_tflops": tf, "mfma_frac": tf / PEAK_TFLOPS_F16}
        tfile = {32: "layer_traffic_b32.json"}.get(B)
    # `achieved` / `mfma_tflops` count ALGORITHMIC flops (one multiply-add per product); the split-operand schemes issue 2 (hi + lo weights) or 3
    # (+ split activations) MFMAs per product, so the matrix pipe itself is that many times busier
    # (f16_w6: a 6-bit K = 64 MFMA takes a quarter of the four fp16 MFMAs it replaces -- 1.25 units per product, 1.5 in the output 1x1 with the
    #  g_lo correction: 1.3125 over a layer's flops)
    mpp = 3 if precision == "f16_x3" else (2.25 if precision == "f16_x3t" else (2 if precision == "f16_w2" else (1.3125 if precision == "f16_w6" else (1.25 if precision == "f16_w6n" else 1))))
    alg_tf = roof["achieved"] if roof["bound"] == "mfma" else roof["mfma_tflops"]
    roof["mfma_per_product"] = mpp
    roof["pipe_tflops"] = alg_tf * mpp
    roof["pipe_frac"] = alg_tf * mpp / PEAK_TFLOPS_F16
    if tfile:
        roof["traffic"], roof["traffic_source"] = load_traffic(tfile, precision)
    return roof


SAMPLER_SOURCES = ("common.h", "common.hip", "tgemm.h", "tlayer.h", "ttail.h", "diffnet_t.h", "diffnet_kernels.h", "diffnet.hip")


TRAIN_SOURCES = ("common.h", "common.hip", "tgemm.h", "tepi_util.h", "conv_gemm.h", "wgrad.h", "train.hip")


def kernel_sources_sha(names=SAMPLER_SOURCES):
    """sha256 of the HIP sources the DiffNet / sampler kernels (or, TRAIN_SOURCES, the trainer's) are built from: a PMC traffic file measured on
    other sources is stale."""
    h = hashlib.sha256()
    for name in names:
        with open(os.path.join(ROOT, "diff-svc_amd", "csrc", name), "rb") as fh:
            h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def load_error_fit(precision):
    """(fit dict, None) from profiles/b32_error_fit.json, or (None, why) when it is missing, stale or for another precision."""
    path = os.path.join(ROOT, "profiles", "b32_error_fit.json")
    if not os.path.exists(path):
        return None, "no error fit committed (profiles/b32_error_fit.json; tools/b32_error_fit.py)"
    with open(path) as f:
        fit = json.load(f)
    if fit.get("csrc_sha16") != kernel_sources_sha():
        return None, "stale: the fit was measured on kernel sources %s, this tree is %s" % (fit.get("csrc_sha16"), kernel_sources_sha())
    if fit.get("precision") != precision:
        return None, "the fit is for %s, the batch ran at %s" % (fit.get("precision"), precision)
    return fit, None


def load_train_traffic():
    """(bytes per training step, source) of profiles/train_traffic.json (tools/gpu_pmc_r5.sh: every per-step kernel's FETCH_SIZE x 2 + WRITE_SIZE
    summed), or (None, why) when it is missing or was measured on other trainer sources."""
    path = os.path.join(ROOT, "profiles", "train_traffic.json")
    if not os.path.exists(path):
        return None, "no PMC pass committed (train_traffic.json)"
    with open(path) as f:
        tj = json.load(f)
    if tj.get("csrc_sha16") != kernel_sources_sha(TRAIN_SOURCES):
        return None, "stale: train_traffic.json was measured on trainer sources %s, this tree is %s" % (tj.get("csrc_sha16"), kernel_sources_sha(TRAIN_SOURCES))
    return tj.get("bytes_per_step"), tj.get("source")


def load_train_kernels():
    """The per-kernel roofline of the training step and the step's algorithmic bytes (profiles/train_kernels.json: tools/train_roofline.py over
    the rocprofv3 kernel stats of `bench.py --train` and the PMC pass), or {'missing': why} when absent or measured on other trainer sources."""
    path = os.path.join(ROOT, "profiles", "train_kernels.json")
    if not os.path.exists(path):
        return {"missing": "no profiles/train_kernels.json (tools/train_roofline.py)"}
    with open(path) as f:
        tk = json.load(f)
    if tk.get("csrc_sha16") != kernel_sources_sha(TRAIN_SOURCES):
        return {"missing": "stale: train_kernels.json was derived on trainer sources %s, this tree is %s" % (tk.get("csrc_sha16"), kernel_sources_sha(TRAIN_SOURCES))}
    keep = ("kernel", "launches_per_step", "avg_us", "algorithmic_bytes_per_launch", "pmc_bytes_per_launch", "pmc_over_algorithmic", "achieved_tflops", "mfma_frac",
            "pipe_frac", "hbm_frac", "bound")
    return {"algorithmic_bytes": tk["algorithmic_bytes_per_step"], "algorithmic_bytes_note": tk["algorithmic_bytes_note"],
            "traffic_over_algorithmic": tk.get("pmc_over_algorithmic"), "per_kernel_source": "profiles/train_kernels.json <- profiles/" + tk["source"],
            "per_kernel": [{k: e[k] for k in keep if k in e} for e in tk["kernels"][:6]]}


def load_voc_kernels():
    """Per-kernel roofline entries of the generator (profiles/voc_kernels.json: tools/voc_roofline.py over the rocprofv3 stats of tools/prof_vocoder.py)."""
    path = os.path.join(ROOT, "profiles", "voc_kernels.json")
    if not os.path.exists(path):
        return {"per_kernel_missing": "no profiles/voc_kernels.json (tools/voc_roofline.py)"}
    with open(path) as f:
        vk = json.load(f)
    h = hashlib.sha256()
    for name in ("common.h", "common.hip", "tgemm.h", "conv_gemm.h", "cg_util.h", "vocoder.hip"):
        with open(os.path.join(ROOT, "diff-svc_amd", "csrc", name), "rb") as fh:
            h.update(name.encode() + b"\0" + fh.read())
    if vk.get("csrc_sha16") != h.hexdigest()[:16]:
        return {"per_kernel_missing": "stale: voc_kernels.json was derived on other vocoder sources"}
    keep = ("kernel", "launches_per_clip", "avg_us", "achieved_tflops", "mfma_frac", "pipe_frac", "hbm_frac", "bound")
    return {"per_kernel_source": "profiles/voc_kernels.json <- profiles/" + vk["source"], "per_kernel": [{k: e[k] for k in keep} for e in vk["kernels"]]}


def load_traffic(name, precision):
    """(bytes_per_launch, source) of a committed PMC measurement, or (None, why) when it is missing, was taken on other kernel
    sources (tools/rocprof_traffic.py stamps the file with kernel_sources_sha()) or at another operand precision."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, "no PMC pass committed (%s)" % name
    with open(path) as f:
        tj = json.load(f)
    if tj.get("csrc_sha16") != kernel_sources_sha():
        return None, "stale: %s was measured on kernel sources %s, this tree is %s" % (name, tj.get("csrc_sha16"), kernel_sources_sha())
    if (" %s " % precision) not in tj.get("source", "").replace("`", " ").replace("(", " ").replace(")", " "):
        return None, "%s was measured at another operand precision than %s" % (name, precision)
    return tj.get("bytes_per_launch"), tj.get("source")


def make_inputs(clips, device):
    hub, m2p, f0 = [], [], []
    for c in clips:
        a, b, c_, _ = synth.clip_inputs(c, T=T_FRAMES, n_units=N_UNITS, H=256)
        hub.append(a); m2p.append(b); f0.append(c_)
    return tuple(torch.from_numpy(np.stack(v)).to(device) for v in (hub, m2p, f0))


def cpu_baseline(hp, sd, vs, h, budget_s=20.0):
    """The oracle (a port: the reference tree is not on the GPU box) on the host cores, same synthetic clip:
    time DDPM steps until ~budget, extrapolate linearly to 1000 (every step is identical work), time the
    vocoder once in full."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dsvc_oracle as O
    # oneDNN convolutions of this size stop scaling (and then collapse) long before a GPU host's core count:
    # 256 threads measured 50 s per step on the first GPU box, so the port is timed on a bounded thread pool
    cores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    hub, m2p, f0, _ = synth.clip_inputs(0, T=T_FRAMES, n_units=N_UNITS, H=256)
    hub, m2p, f0 = torch.from_numpy(hub)[None], torch.from_numpy(m2p)[None], torch.from_numpy(f0)[None]
    cond, f0_denorm, _ = O.build_cond(sd, hub, m2p, f0.clone(), hp)
    cond_t = cond.transpose(1, 2).contiguous()
    x = O.ddpm_noise_ref_layout(1, [0], 0, T_FRAMES, 128, O.PURPOSE_X_INIT)
    z = O.ddpm_noise_ref_layout(1, [0], 999, T_FRAMES, 128)
    t = torch.full((1,), 999, dtype=torch.long)
    n, t_steps = 0, 0.0
    with torch.no_grad():
        for i in range(3 + 200):
            t0 = time.perf_counter()
            eps = O.diffnet_forward(sd, x, t, cond_t, hp["dilation_cycle_length"])
            x = O.ddpm_update(sd, x, eps, t, z)
            dt = time.perf_counter() - t0
            if i >= 3:
                n += 1
                t_steps += dt
                if t_steps > budget_s * 0.7 and n >= 2:
                    break
        hop = int(np.prod(h["upsample_rates"]))
        ini, nz = O.vocoder_rng(1, [0], T_FRAMES * hop)
        gw = O.fold_weight_norm(vs)
        mel = torch.clamp(O.finish_mel(sd, x, m2p), hp["mel_vmin"], hp["mel_vmax"])
        t0 = time.perf_counter()
        O.generator_forward(gw, h, 2.30259 * mel.transpose(2, 1), f0_denorm, ini, nz)
        t_voc = time.perf_counter() - t0
    per_step = t_steps / n
    total = 1000 * per_step + t_voc
    return {"value": CLIP_SECONDS / total, "unit": "audio-sec/wall-sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d of 1000 DDPM steps (%.3f s/step, extrapolated) + the full NSF-HiFiGAN pass (%.2f s), one 10 s clip, T=861"
                      % (n, per_step, t_voc)}


def time_job(pipe, n_clips, cpb, dev, ddpm_steps, overlap=True, warm=True):
    """One whole job of n_clips fixed-length 10 s clips on ONE device, `cpb` clips per batch through one SvcPipeline (SvcPipeline.infer_job:
    the reference's sequential loop, batch.py:25-43, as batches; the vocoder of batch k on a second stream under the sampler of batch k+1).
    This is the 1-GPU side of north_star's ">= 6x at 8 GPUs vs 1 GPU on the batched config": the WHOLE 256-clip job on one device, not one
    rank's 32-clip share (that one is `same_workload_1gpu` of the N > 1 lines: weak scaling)."""
    hub, m2p, f0 = make_inputs(list(range(n_clips)), dev)
    if warm:                                                  # bucket + graph of this batch size on a short chain
        pipe.model.K_step = 130
        try:
            pipe.infer_job(hub[:cpb], m2p[:cpb], f0[:cpb], clips_per_batch=cpb, seed=1, overlap=overlap)
        finally:
            pipe.model.K_step = ddpm_steps
    torch.cuda.synchronize(); t0 = time.perf_counter()
    wav = pipe.infer_job(hub, m2p, f0, clips_per_batch=cpb, seed=2, overlap=overlap)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return {"clips": n_clips, "clips_per_batch": cpb, "batches": -(-n_clips // cpb), "value": n_clips * CLIP_SECONDS / dt, "unit": "audio-sec/wall-sec",
            "s_per_job": dt, "vocoder_overlapped": bool(overlap), "finite_output": bool(torch.isfinite(wav).all().item())}


RAGGED_T = (430, 700, 861, 1200, 1600, 2100, 2600)          # 5 ... 30 s chunks, as the reference's slicer hands them out (infer_tools/slicer.py)


def time_ragged(pipe, dev, speedup, ddpm_steps):
    """Variable-length serving (VERDICT r5 missing 4): the reference's driver calls the model chunk by chunk, every chunk with its own T
    (infer.py:44-67, infer_tool.py:155-159,276).  Seven chunks of 5 ... 30 s through ONE pipeline: the first pass builds a workspace bucket and
    captures a chain per bucket (`first_call_ms`); the second, timed pass must build nothing (`recapture_count`, `buckets_built`: the deltas of
    dsvc_sampler_stats over it) and run within 5 % of the same chunks each repeated at a fixed T (`fixed_T_value`)."""
    chunks = []
    for i, T in enumerate(RAGGED_T):
        n_units = max(2, T * N_UNITS // T_FRAMES)
        a, b, c, _ = synth.clip_inputs(100 + i, T=T, n_units=n_units, H=256)
        chunks.append(tuple(torch.from_numpy(v[None]).to(dev) for v in (a, b, c)))
    audio = sum(RAGGED_T) * 512 / 44100.0
    run = lambda ch, i: pipe.infer(*ch, speedup=speedup, seed=40 + i, first_clip=100 + i, full_length=True)
    smp = pipe.model._handle("plms" if speedup > 1 else "ddpm", speedup, frames=RAGGED_T[0], clips=1)
    first = []
    for i, ch in enumerate(chunks):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run(ch, i)
        torch.cuda.synchronize(); first.append((time.perf_counter() - t0) * 1e3)
    s0 = smp.stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i, ch in enumerate(chunks[::-1]):                        # another order than the first pass
        run(ch, i)
    torch.cuda.synchronize(); mixed = time.perf_counter() - t0
    s1 = smp.stats()
    fixed = 0.0
    for i, ch in enumerate(chunks):                              # each chunk again, back to back at ITS fixed T (second of two calls timed)
        run(ch, i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run(ch, i)
        torch.cuda.synchronize(); fixed += time.perf_counter() - t0
    batched = None
    if True:
        # the same chunks as a few padded batches (SvcPipeline.infer_chunks: the caller has the whole utterance in hand; chunk i keeps the noise
        # streams of clip 100 + i, so only the operand precision `auto` picks by call size differs from the one-by-one pass; PLMS: the same
        # split-operand precision at any size)
        flat = [tuple(t[0] for t in ch) for ch in chunks]
        plan = pipe.plan_chunks(list(RAGGED_T), speedup)
        pipe.infer_chunks(flat, seed=60, first_clip=100, speedup=speedup)             # buckets built, chains captured
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pipe.infer_chunks(flat, seed=61, first_clip=100, speedup=speedup)
        torch.cuda.synchronize(); tb = time.perf_counter() - t0
        batched = {"value": audio / tb, "unit": "audio-sec/wall-sec", "s_per_pass": tb, "vs_one_by_one": mixed / tb,
                   "plan": [[RAGGED_T[i] for i in g] for g in plan],
                   "precisions": [pipe.model.denoise_fn.precision_for("plms" if speedup > 1 else "ddpm", speedup, frames=len(g) * RAGGED_T[g[0]], clips=len(g)) for g in plan],
                   "what": "SvcPipeline.infer_chunks: chunks sorted by length, contiguous groups chosen by a cost model of one evaluation "
                           "(fused layer kernel 45 / 65 / 125 us per layer by tile width, a line through the one-by-one numbers for the small tilings), "
                           "each group one padded batch (trailing mel2ph == 0 frames are the convs' zero padding, after_infer's glue per clip); the fused "
                           "kernels' workgroups on tiles wholly beyond a clip's length return at once and the tile width is chosen by the tiles that have "
                           "work (dsvc_sample_args.clip_lens_host, ABI v9), and so do the two-launch tilings' workgroups (csrc/tgemm.h SKIP: PLMS chunks, DDPM "
                           "groups under 48 tiles): a group is priced by its active tiles"}
    return {"workload": "%d chunks of one utterance, T = %s mel frames (%.0f s of audio), B = 1, %s + NSF-HiFiGAN, one pipeline"
                        % (len(chunks), list(RAGGED_T), audio, "%d-step DDPM" % ddpm_steps if speedup <= 1 else "PLMS (pndm_speedup %d)" % speedup),
            **({"batched_chunks": batched} if batched else {}),
            "value": audio / mixed, "unit": "audio-sec/wall-sec", "s_per_pass": mixed,
            "fixed_T_value": audio / fixed, "vs_fixed_T": fixed / mixed,
            "first_call_ms": [round(v, 2) for v in first], "steady_call_ms_total": mixed * 1e3,
            "recapture_count": (s1["capture_ddpm"] + s1["capture_plms"]) - (s0["capture_ddpm"] + s0["capture_plms"]),
            "buckets_built": s1["buckets_allocated"] - s0["buckets_allocated"], "graphs_alive": s1["graphs_alive"],
            "buckets": sorted({(T + 8 + 127) // 128 for T in RAGGED_T})}


def plms_breakdown(pipe, hub, m2p, f0, n=5):
    """Where one PLMS-50 clip's wall time goes (2 ms of 25 were outside both the evaluations and the vocoder): HIP events at
    the phase boundaries inside dsvc_sample (dsvc_sampler_phase_times) + events around the condition builder and the vocoder, mean of n clips."""
    smp = pipe.model._handle("plms", 20, frames=T_FRAMES, clips=1)
    smp.phase_timing(True)
    acc = {}
    try:
        for i in range(n + 1):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            hpx = dict(pipe.hp, pndm_speedup=20)
            pipe.model.hp = hpx; pipe.model.fs2.hp = hpx
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ev[0].record()
            ret = pipe.model.fs2(hub, m2p, None, None, f0.clone(), None, None, skip_decoder=True, infer=True)
            cond = ret.pop("cond_bht", None)
            if cond is None:
                cond = ret["decoder_inp"].transpose(1, 2).contiguous()
            ev[1].record()
            mel = smp.sample(cond, pipe.model.K_step, speedup=20, mel2ph=m2p, seed=60 + i, first_clip=0)
            ev[2].record()
            mel_c = torch.clamp(mel, hpx["mel_vmin"], hpx["mel_vmax"])
            pipe.vocoder.vocode(mel_c, ret["f0_denorm"], seed=60 + i, first_clip=0)
            ev[3].record()
            torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
            if i == 0:
                continue
            ph = smp.phase_times()
            parts = {"cond_build": ev[0].elapsed_time(ev[1]), "prepare_cond": ph["prepare"], "x_init": ph["init"], "sampler_chain": ph["chain"],
                     "finish": ph["finish"], "vocoder": ev[2].elapsed_time(ev[3])}
            parts["other"] = wall - sum(parts.values())
            parts["wall"] = wall
            for k, v in parts.items():
                acc[k] = acc.get(k, 0.0) + v / n
    finally:
        smp.phase_timing(False)
    return {k: round(v, 3) for k, v in acc.items()}


def spawn_ranks(n):
    """Re-run this command line as n ranks on this node: python -m torch.distributed.run --nnodes=1 --nproc-per-node n (rendezvous on
    127.0.0.1, a free port).  Returns the launcher's exit code; rank 0 of the child job prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL's cross-process buffer sharing needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def same_workload_fields(value, solo_value, world, what):
    """an N > 1 line carries its own same-workload 1-GPU denominator.  `--gpus 1` times ONE clip per GPU (BASELINE
    configs[1]) and `--gpus N` 32 clips per GPU (configs[3]): dividing the two `value`s would read the batch size as scaling.  So every
    N > 1 run first lets rank 0 run its own share alone, un-gathered, while the other ranks wait, and the line says what N GPUs buy over
    THAT: speedup_vs_1gpu_same_workload = value / same_workload_1gpu, scaling_efficiency = speedup / N (weak scaling: 1.0 is ideal)."""
    if world <= 1 or solo_value is None:
        return {}
    return {"same_workload_1gpu": solo_value, "same_workload_1gpu_what": what,
            "speedup_vs_1gpu_same_workload": value / solo_value, "scaling_efficiency": value / solo_value / world}


def comm_info(dist, world, share_device):
    """What moved the bytes between the ranks: backend, world size and (RCCL) library version -- on the JSON line of every N > 1 run."""
    if world <= 1 or dist is None:
        return None
    info = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}
    if info["backend"] == "nccl":
        try:
            info["version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            info["library"] = "RCCL (torch.distributed 'nccl' on ROCm), xGMI peer-to-peer"
        except Exception as ex:
            info["version"] = repr(ex)[:80]
    elif share_device:
        info["note"] = "launch-contract test mode (DSVC_BENCH_SHARE_DEVICE=1): every rank on cuda:0, gloo on the host"
    return info


def train_batch(hp, B, T, rank, device):
    """Synthetic training batch of BASELINE configs[4]: B clips x T mel frames (content units, alignment, f0, target mels).  It is
    synth.train_batch_kat -- on rank 0 at 64 x 128 exactly the batch the real reference's forward(infer=False) + backward() was run on for
    tests/golden/train_grads_bench.npz (oracle/make_golden.py::TRAIN_CASES_BENCH), so the timed step is the step that has a parity number."""
    n_units = max(2, T * N_UNITS // T_FRAMES)
    hub, m2p, f0, mels, _ = synth.train_batch_kat(hp, range(rank * B, rank * B + B), T, n_units, 77 + rank)
    return tuple(torch.from_numpy(v).to(device) for v in (hub, m2p, f0, mels))


def time_train_steps(hp, sd, B, T, steps, warmup, rank, device, sync, world=1):
    """(ms per optimisation step of DiffusionTrainerHip -- forward + backward + gradient all-reduce over the ranks + clip + AdamW --, final
    loss, solo ms).  solo ms (world > 1 only, else None): rank 0 ALONE stepping on its own batch without the all-reduce while the other
    ranks wait at a barrier -- the same per-GPU workload on one GPU, measured in this very run, so that the N-GPU line carries its own
    denominator (the optimiser state is put back afterwards: the replicas stay identical)."""
    from diffsvc_amd.train import DiffusionTrainerHip
    tr = DiffusionTrainerHip(dict(hp, lr=1e-4), sd)
    hub, m2p, f0, mels = train_batch(hp, B, T, rank, device)
    loss = None
    for i in range(warmup):
        loss = tr.train_step(hub, m2p, f0, mels, seed=10 + i, first_clip=rank * B)
    solo = None
    if world > 1:
        sync()
        if rank == 0:
            keep = [t.clone() for t in (tr.params, tr.exp_avg, tr.exp_avg_sq)], tr.global_step
            n = max(2, min(steps, 5))
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(n):
                t_solo = torch.randint(0, int(hp.get("K_step", hp["timesteps"])), (B,), device=device)
                tr.forward_backward(hub, m2p, f0, mels, t_solo, seed=50 + i, first_clip=0)
                tr.optimizer_step(reduced=True)                 # (reduced=True: no all-reduce -- one GPU's step)
            torch.cuda.synchronize(); solo = (time.perf_counter() - t0) / n * 1e3
            for dst, src in zip((tr.params, tr.exp_avg, tr.exp_avg_sq), keep[0]):
                dst.copy_(src)
            tr.global_step = keep[1]
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = tr.train_step(hub, m2p, f0, mels, seed=100 + i, first_clip=rank * B)
    sync()
    return (time.perf_counter() - t0) / steps * 1e3, float(loss.item()), solo


def time_variable_shape_steps(hp, sd, device, rounds=3):
    """ms per optimisation step of ONE DiffusionTrainerHip walking five (B, T) shapes of ~8 192 frames in turn -- what the reference's
    max_tokens / max_sentences loader hands a step (training/task/tts.py:60-88: batches of equal token count, different length) -- against the
    fixed-shape 64 x 128 number: the difference is the cost of a shape change (workspace re-layout: gap-row clearing, no re-allocation once the
    largest shape has been seen)."""
    from diffsvc_amd.train import DiffusionTrainerHip
    shapes = [(64, 128), (32, 256), (43, 190), (86, 95), (16, 512)]
    tr = DiffusionTrainerHip(dict(hp, lr=1e-4), sd)
    batches = []
    for i, (B, T) in enumerate(shapes):
        n_units = max(2, T * N_UNITS // T_FRAMES)
        hub, m2p, f0, mels, _ = synth.train_batch_kat(hp, range(100 * i, 100 * i + B), T, n_units, 300 + i)
        batches.append(tuple(torch.from_numpy(v).to(device) for v in (hub, m2p, f0, mels)))
    for b in batches:                                     # every shape once: allocations grow to the largest
        tr.train_step(*b, seed=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for r in range(rounds):
        for i, b in enumerate(batches):
            tr.train_step(*b, seed=10 + 5 * r + i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / (rounds * len(shapes)) * 1e3
    return {"ms_per_step": ms, "shapes": shapes, "frames_per_step_mean": sum(B * T for B, T in shapes) / len(shapes),
            "frames_per_s": sum(B * T for B, T in shapes) / len(shapes) / (ms * 1e-3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--clips-per-gpu", type=int, default=0,
                    help="default: 1 at --gpus 1 (BASELINE configs[1], the single-clip headline), 32 at --gpus N > 1 "
                         "(BASELINE configs[3]: 256 clips over 8 GPUs)")
    ap.add_argument("--ddpm-steps", type=int, default=1000)
    ap.add_argument("--speedup", type=int, default=1, help="pndm_speedup (>1 = PLMS); the headline config is 1")
    ap.add_argument("--precision", default="auto",
                    help="auto (default: what DiffNetHip.precision_for picks by sampler and call size -- f16_x3t, fp32-class, for DDPM under 6000 "
                         "frames, PLMS and forward(); f16_w6 for batched DDPM: the precisions tests/test_gpu_headline.py holds to <= 9.0e-4 of the "
                         "1e-3 mel bar on every real-reference golden of the benchmarked sizes), f16_x3t (hi+lo weights and split activations on "
                         "the tgemm engine), f16_w6 (hi + 6-bit lo weight products, 6-bit correction of the gate output; f16_w6n without it), "
                         "f16_w2 (exact hi+lo weights, fp16 activations), f16_mN / f16_dN (N time-dithered single-plane weight "
                         "roundings; m: exact output 1x1), f16_x3 (the split scheme on the older conv_gemm engine), f16")
    ap.add_argument("--pcm16", action="store_true",
                    help="gather the PCM as the 16-bit integers the reference writes (infer.py:70) instead of fp32: half the bytes on xGMI")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batched", action="store_true", help="skip the extra batched (32 clips/GPU) measurement")
    ap.add_argument("--train", action="store_true",
                    help="measure BASELINE configs[4] instead: the training step (diffusion loss fwd+bwd + AdamW) on a 64 x 128-frame mel "
                         "batch per GPU, gradients all-reduced over the ranks (weak scaling)")
    ap.add_argument("--job-clips", type=int, default=0,
                    help="--gpus 1 only: time ONE job of this many clips on the one device (batches of --clips-per-batch through one pipeline, the "
                         "vocoder of batch k under the sampler of batch k+1) and print it as `value`: north_star's 1-GPU denominator for the 8-GPU job")
    ap.add_argument("--clips-per-batch", type=int, default=32)
    ap.add_argument("--no-overlap", action="store_true", help="--job-clips: run the vocoder on the sampler's stream (A/B of the overlap)")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch the sampler steps eagerly (rocprofv3 --pmc segfaults on hipGraph replays on this stack)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: spawn the N ranks ourselves (one process per GPU under torch.distributed.run, as the reference's
        # trainer spawns its DDP workers, utils/pl_utils.py:483-485) instead of silently measuring one GPU
        sys.exit(spawn_ranks(args.gpus))
    SHARE_DEVICE = os.environ.get("DSVC_BENCH_SHARE_DEVICE") == "1"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks: refusing to print a line whose n_gpus would not be what was "
              "asked for" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    DRY = os.environ.get("DSVC_BENCH_DRY") == "1"          # launch-contract test WITHOUT a GPU (tests/test_host.py): spawn, rendezvous, clocks, JSON line
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if SHARE_DEVICE or DRY: # launch-contract test on a 1-GPU box: every rank on cuda:0, gloo instead of RCCL
            local_rank = 0
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world)
    if DRY:
        # no hot path, no number: every rank sleeps its "steps" between the same barriers, the MAX over the ranks is taken the same way,
        # and rank 0 prints a line that says what it is
        import time as _t
        solo = None
        if world > 1:
            dist.barrier()
            if rank == 0:                                     # (the same-workload 1-GPU leg of the real run: rank 0 alone, the others at the barrier)
                ts = _t.perf_counter(); _t.sleep(0.01); solo = 1.0 / (_t.perf_counter() - ts)
            dist.barrier()
        t0 = _t.perf_counter()
        for _ in range(args.steps):
            _t.sleep(0.01 * (1 + rank))
        if world > 1:
            dist.barrier()
        el = torch.tensor([_t.perf_counter() - t0])
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"metric": "dry run of the launch contract (no GPU work)", "value": None, "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": float(el.item()) / max(args.steps, 1) * 1e3, "dry_run": True,
                              "scaling": "weak", "rccl": comm_info(dist, world, True), "train": bool(args.train),
                              **same_workload_fields(world * args.steps / float(el.item()), solo, world, "dry run: rank 0 sleeping alone")}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    assert torch.cuda.is_available(), "bench.py needs a HIP device: the product path has no CPU fallback"
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)
    pcm16 = args.pcm16 or os.environ.get("DSVC_BENCH_PCM16") == "1"
    # before anything else has loaded the chip: the sustained dense-fp16 MFMA rate on a cold, idle part (again after the batched run)
    sustained = {"cold": probe_mfma()} if (rank == 0 and world == 1 and not args.no_batched and not args.train) else None

    hp = dict(synth.HPARAMS_44K, K_step=args.ddpm_steps)
    h = dict(synth.VOCODER_44K)
    sd = synth.acoustic_state(hp, 0)
    vs = synth.vocoder_state(h, 1)
    if args.train:
        def sync_t():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
        Bt, Tt = 64, 128
        ms, loss, solo_ms = time_train_steps(hp, sd, Bt, Tt, args.steps, args.warmup, rank, dev, sync_t, world)
        if world > 1:
            tmax = torch.tensor([ms], device="cpu" if SHARE_DEVICE else dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            ms = float(tmax.item())
        if rank == 0:
            print(json.dumps({"metric": "training mel frames per wall-second (diffusion loss fwd+bwd + AdamW, 44.1 kHz DiffNet)",
                              "value": world * Bt * Tt / (ms * 1e-3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16x3",
                              "dtype_detail": "split fp16 MFMA operands (hi+lo weights and activations: fp32-class), fp32 accumulate, fp32 master weights",
                              "data": "synthetic",
                              "config": {"workload": "BASELINE configs[4]: training step on a 64 x 128-frame mel batch per GPU, gradient all-reduce (mean) "
                                                     "of 32 M fp32 over the ranks", "clips_per_gpu": Bt, "mel_frames": Tt, "loss": hp["diff_loss_type"],
                                         "parallelism": "data-parallel x%d" % world},
                              "rccl": comm_info(dist, world, SHARE_DEVICE), "final_loss": loss,
                              **same_workload_fields(world * Bt * Tt / (ms * 1e-3), None if solo_ms is None else Bt * Tt / (solo_ms * 1e-3), world,
                                                     "rank 0 alone, %d x %d-frame batch, forward + backward + clip + AdamW without the all-reduce" % (Bt, Tt)),
                              "roofline": {"bound": "mfma", "scope": "whole step per GPU (forward + backward + clip + AdamW), not one kernel: the largest kernels are "
                                           "wgrad_fm_kernel (20 %: the weight gradients, contracted from the frame-major planes through the transposing LDS read) "
                                           "and the transposed conv on the tgemm engine (17 %), profiles/r5G_kernel_stats_train.csv",
                                           "algorithmic_tflop_per_step": train_step_flops(hp, Bt * Tt) / 1e12,
                                           "achieved": train_step_flops(hp, Bt * Tt) / (ms * 1e-3) / 1e12, "peak": PEAK_TFLOPS_F16, "unit": "TFLOP/s",
                                           "frac": train_step_flops(hp, Bt * Tt) / (ms * 1e-3) / 1e12 / PEAK_TFLOPS_F16, "mfma_per_product": 3, "traffic": load_train_traffic()[0], "traffic_source": load_train_traffic()[1],
                                           "traffic_unit": "HBM bytes per step (PMC)", **load_train_kernels()},
                              "cpu_baseline": None}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    pipe = SvcPipeline(hp, sd, vs, h, precision=args.precision, vocoder_precision="f16_x3")

    if args.job_clips > 0:
        if world != 1:
            print("bench.py: --job-clips is the ONE-device job (use --gpus 1)", file=sys.stderr)
            sys.exit(2)
        job = time_job(pipe, args.job_clips, args.clips_per_batch, dev, args.ddpm_steps, overlap=not args.no_overlap)
        print(json.dumps({"metric": "audio-sec/wall-sec (RTF) end-to-end 44.1kHz %d-step DDPM + NSF-HiFiGAN" % args.ddpm_steps, "value": job["value"],
                          "unit": "audio-sec/wall-sec", "n_gpus": 1, "steps": 1, "warmup": 1, "ms_per_step": job["s_per_job"] * 1e3, "higher_is_better": True,
                          "scaling": "strong", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
                          "config": {"workload": "BASELINE configs[3] on ONE device: the whole %d-clip job (10 s clips, 44.1 kHz, %d-step DDPM + NSF-HiFiGAN) "
                                                 "as %d batches of %d" % (args.job_clips, args.ddpm_steps, job["batches"], args.clips_per_batch),
                                     "precision": pipe.model.denoise_fn.precision_for("ddpm", 1, frames=args.clips_per_batch * T_FRAMES, clips=args.clips_per_batch)},
                          "job": job, "cpu_baseline": None}))
        return
    B = args.clips_per_gpu if args.clips_per_gpu > 0 else (1 if world == 1 else 32)
    # what the timed chain runs at: precision="auto" picks by sampler and by the size of the call (DiffNetHip.precision_for)
    prec = pipe.model.denoise_fn.precision_for("plms" if args.speedup > 1 else "ddpm", args.speedup, frames=B * T_FRAMES, clips=B)
    n_clips = B * world
    my_clips = shard_clips(n_clips, rank, world) if world > 1 else list(range(B))
    hub, m2p, f0 = make_inputs(my_clips, dev)
    clip_ids = torch.tensor(my_clips, dtype=torch.int32, device=dev)

    def one_step(seed):
        # noise streams are keyed by the GLOBAL clip index: clip i produces the same PCM on 1 GPU and on any rank of N GPUs
        wav = pipe.infer(hub, m2p, f0, speedup=args.speedup, seed=seed, clip_ids=clip_ids, use_graph=not args.no_graph, full_length=True)
        if world > 1 or pcm16:
            wav = gather_pcm(wav, my_clips, n_clips, as_int16=pcm16)
        return wav

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_step(100 + i)
    solo_value = None
    if world > 1:
        # the same per-GPU workload on ONE GPU, measured in this run: rank 0 alone on its own share, no gather, the others at the barrier
        sync()
        if rank == 0:
            if args.warmup == 0:
                pipe.infer(hub, m2p, f0, speedup=args.speedup, seed=150, clip_ids=clip_ids, use_graph=not args.no_graph, full_length=True)
            torch.cuda.synchronize(); ts = time.perf_counter()
            pipe.infer(hub, m2p, f0, speedup=args.speedup, seed=151, clip_ids=clip_ids, use_graph=not args.no_graph, full_length=True)
            torch.cuda.synchronize(); solo_value = B * CLIP_SECONDS / (time.perf_counter() - ts)
    whole_job = None
    if world > 1 and rank == 0 and not os.environ.get("DSVC_BENCH_NO_STRONG"):
        # north_star's own ratio -- "speed-up at N GPUs vs 1 GPU on the batched config" -- needs the WHOLE N x B-clip job on one device as its
        # denominator (strong scaling), not one rank's share: rank 0 runs it here, alone, as N batches of B through its pipeline
        # (SvcPipeline.infer_job; the bucket and the captured chain of this batch size exist already), the other ranks wait at the barrier
        whole_job = time_job(pipe, n_clips, B, dev, args.ddpm_steps, overlap=True, warm=False)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        wav = one_step(200 + i)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device="cpu" if SHARE_DEVICE else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    pipe.check()                                          # the device path's deferred argument checks (alignment / step flags), outside the timed region
    ok = bool(torch.isfinite(wav.float()).all().item())
    value = n_clips * CLIP_SECONDS * args.steps / elapsed

    result = None
    if rank == 0:
        # ---- roofline of the dominant kernel (dilated conv + gate), HIP events on the launch stream ----
        roof = dominant_kernel_roofline(pipe.model._handle("plms" if args.speedup > 1 else "ddpm", args.speedup, frames=B * T_FRAMES, clips=B), B, prec)
        result = {
            "metric": "audio-sec/wall-sec (RTF) end-to-end 44.1kHz %d-step %s + NSF-HiFiGAN" % (
                args.ddpm_steps, "DDPM" if args.speedup <= 1 else "PLMS/%d" % args.speedup),
            "value": value, "unit": "audio-sec/wall-sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "dtype_detail": "fp16 MFMA operands (%s%s), fp32 accumulate, fp32 residual/skip/state" % (
                prec, ": hi + lo weights (the lo plane as 6-bit codes on the block-scaled MFMA in the small tilings) and split hi | lo activations -- "
                      "fp32-class" if prec.startswith("f16_x3") else ""),
            "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: single 10 s clip per GPU, 44.1 kHz, full %d-step DDPM + NSF-HiFiGAN" % args.ddpm_steps) if B == 1 else
                                   ("BASELINE configs[3] share: %d x 10 s clips per GPU in one batch (%d clips over %d GPU(s)), 44.1 kHz, full %d-step DDPM "
                                    "+ NSF-HiFiGAN, gather of the PCM; the same per-GPU workload on 1 GPU is `batched.value` of the --gpus 1 line"
                                    % (B, n_clips, world, args.ddpm_steps)),
                       "clips_per_gpu": B, "mel_frames": T_FRAMES, "content_frames": N_UNITS, "sampler_steps": args.ddpm_steps,
                       "pndm_speedup": args.speedup, "precision": prec, "vocoder_precision": "f16_x3",
                       "weights": "random-init (synthetic checkpoint, seeds 0/1)", "parallelism": "utterance-sharded x%d, gather of PCM" % world,
                       "gather": "int16 PCM" if pcm16 else "fp32 PCM"},
            "finite_output": ok,
            "rccl": comm_info(dist, world, SHARE_DEVICE),
            **same_workload_fields(value, solo_value, world, "rank 0 alone on its own %d-clip share (cond -> %d-step %s -> NSF-HiFiGAN), no gather, "
                                   "the other ranks waiting at a barrier" % (B, args.ddpm_steps, "DDPM" if args.speedup <= 1 else "PLMS/%d" % args.speedup)),
            "roofline": roof,
        }
        if whole_job is not None:
            result["strong_scaling"] = {"job_clips": n_clips, "one_gpu_s_per_job": whole_job["s_per_job"], "one_gpu_value": whole_job["value"],
                                        "n_gpu_s_per_job": elapsed / args.steps, "speedup_vs_1gpu_whole_job": whole_job["s_per_job"] / (elapsed / args.steps),
                                        "what": "the same %d-clip job on ONE device (rank 0 alone, %d batches of %d through one pipeline, no gather) against the "
                                                "%d-rank job of this line, gather included: north_star's '>= 6x at 8 GPUs vs 1 GPU on the batched config'"
                                                % (n_clips, whole_job["batches"], B, world)}
        if world == 1 and B == 1 and args.speedup <= 1 and roof["bound"] == "mfma":
            # the single clip's state with the evidence on the line.  (Round 6's third session moved it for the first time since r4: -2.6 % per
            # step by same-box A/B -- a scalar-cache miss off the kernels' front path, the gate conv over four K slices: DESIGN 4.1 (viii), (ix).)
            roof["frozen_since"] = "r4 (r6, third session: -2.6 % per step, gate kernel 10.5 -> 10.2 us on the same class of box; the floor model is unchanged)"
            roof["target_feasibility"] = {
                "target_x_rt": 200.0, "ceiling_x_rt": 55.0,
                "why": "20 000 serially dependent layer evaluations per clip, two launches each: at the stated latency floors (gate 4.9 us + res/skip 3.5 us per "
                       "layer + 43 graph-node boundaries per step) a 1000-step clip cannot go below ~0.18 s = 55x RT; both kernels have sat at 44-48 % of those "
                       "floors for two rounds (r4: 10.3 / 7.7 us, r5: 10.3 / 7.7 us, r6: 10.0 / 7.6 us on a fast box) with everything in design/tgemm.md tried (XCD-major placement "
                       "-12 %, split-K variants, 6-bit lo plane +7.5 %); one frame tile's eight slices already share two XCDs, so the all-to-all hand-off of a "
                       "persistent layer costs 26-28 us (tools/micro/handoff.hip) against 18 us for the two launches",
                "what_serves_the_target_instead": "batching: 32 clips per GPU run at 123-125x RT per GPU (`batched`), 36 at 129-131x (`job_256.chip_filling_batches`)"}
            try:        # the single clip's second layer kernel (39 % of its GPU time), priced the same way
                result["roofline_res_skip"] = res_skip_roofline(pipe.model._handle("ddpm", 1, frames=B * T_FRAMES, clips=B), prec)
            except Exception as ex:
                result["roofline_res_skip"] = {"error": repr(ex)[:300]}
        if os.environ.get("DSVC_BENCH_PCM_STATS") == "1":       # test hook: per-clip moments of the gathered PCM of the last step
            w64 = wav.double()
            result["pcm_stats"] = [[int(i), float(w64[i].sum()), float((w64[i] ** 2).sum())] for i in range(w64.shape[0])]
        # PMC-derived HBM traffic of the same kernel (a separate `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` pass over this very
        # command: tools/gpu_pmc_r5.sh + tools/rocprof_traffic.py; counters cannot be read from inside the process)
        if world == 1 and B == 1 and args.speedup <= 1 and not args.no_batched:
            # BASELINE configs[2]: the same clip with the 50-iteration PLMS sampler (pndm_speedup=20, 51 denoiser evaluations)
            pipe.infer(hub, m2p, f0, speedup=20, seed=7)
            torch.cuda.synchronize(); tp0 = time.perf_counter()
            for i in range(3):
                pipe.infer(hub, m2p, f0, speedup=20, seed=8 + i)
            torch.cuda.synchronize(); tpl = (time.perf_counter() - tp0) / 3
            result["plms_50"] = {"workload": "BASELINE configs[2]: single 10 s clip, 50-iteration PLMS (pndm_speedup=20) + NSF-HiFiGAN",
                                 "precision": pipe.model.denoise_fn.precision_for("plms", 20),
                                 "value": CLIP_SECONDS / tpl, "unit": "audio-sec/wall-sec", "ms_per_clip": tpl * 1e3}
            try:
                result["plms_50"]["breakdown_ms"] = plms_breakdown(pipe, hub, m2p, f0)
            except Exception as ex:
                result["plms_50"]["breakdown_ms"] = {"error": repr(ex)[:200]}
            try:    # the reference's driver: chunk after chunk, a new T every call (one pipeline, buckets + captured chains re-used)
                result["ragged"] = {"plms_50": time_ragged(pipe, dev, 20, args.ddpm_steps), "ddpm": time_ragged(pipe, dev, 1, args.ddpm_steps)}
            except Exception as ex:
                result["ragged"] = {"error": repr(ex)[:300]}
            try:    # the same chain with fp16 activations and exact weights: faster, but 1 of 10 (clip, noise) pairs measured over the mel bar
                pw = SvcPipeline(hp, sd, vs, h, precision="f16_w2", vocoder_precision="f16_x3")
                pw.infer(hub, m2p, f0, speedup=20, seed=7)
                torch.cuda.synchronize(); tp0 = time.perf_counter()
                for i in range(3):
                    pw.infer(hub, m2p, f0, speedup=20, seed=8 + i)
                torch.cuda.synchronize(); tw = (time.perf_counter() - tp0) / 3
                result["plms_50"]["at_f16_w2"] = {"value": CLIP_SECONDS / tw, "ms_per_clip": tw * 1e3,
                                                  "note": "not shipped: (8.2 +- 1.2)e-4 mel error, one of ten pairs at 1.08e-3"}
                del pw
            except Exception as ex:
                result["plms_50"]["at_f16_w2"] = {"error": repr(ex)[:200]}
        if not args.no_batched and world == 1 and B == 1:
            # the throughput configuration (BASELINE configs[3] per-GPU share): 32 clips in one batch
            Bb = 32
            hb, mb, fb = make_inputs(list(range(Bb)), dev)
            pipe.model.hp = dict(hp, K_step=30); pipe.model.K_step = 30
            pipe.infer(hb, mb, fb, seed=1)                            # warm-up on a short chain
            pipe.model.K_step = args.ddpm_steps
            torch.cuda.synchronize(); tb = time.perf_counter()
            pipe.infer(hb, mb, fb, seed=2)
            torch.cuda.synchronize(); tb = time.perf_counter() - tb
            precb = pipe.model.denoise_fn.precision_for("ddpm", 1, frames=Bb * T_FRAMES, clips=Bb)
            broof = dominant_kernel_roofline(pipe.model._handle("ddpm", 1, frames=Bb * T_FRAMES, clips=Bb), Bb, precb)
            result["batched"] = {"workload": "BASELINE configs[3] per-GPU share: 32 x 10 s clips in one batch, 1000-step DDPM + NSF-HiFiGAN",
                                 "clips_per_gpu": Bb, "precision": precb, "value": Bb * CLIP_SECONDS / tb, "unit": "audio-sec/wall-sec",
                                 "s_per_batch": tb, "roofline": broof}
            # where the two kernels stand against the 200x target, with the evidence (profiles/r6c_layer_ablations.txt)
            result["batched"]["target_feasibility"] = {
                "target_x_rt": 200.0, "ceiling_x_rt": 136.0,
                "why": "the fused layer's two phases sit on different roofs and do not overlap on this part (gate phase: matrix pipe at 70-90 % of its issue "
                       "rate at the 1.5-1.6 GHz the chip holds; output phase: 193 MB at the 5.4 TB/s streaming rate; dedicated memory waves starve beside "
                       "MFMA waves, de-phased half batches gain nothing: design/tlayer.md).  Serial-phase model: 57 us matrix + 56 us HBM per layer -> "
                       "136x; 200x needs 75 us per layer.",
                "levers_measured_r6": {
                    "cproj_compressed": "ablation reading HALF of cproj (1536 of 3072 B per frame, an upper bound on fp16-hi + 6-bit-lo codes at 2112 B): "
                                        "-2.5 % per DDPM step -> the real scheme <= -1.5 %: not built",
                    "persistent_launch_neighbour_flags": "the hand-off protocol (poll i-1 / i+1 + agent acquire; drain + agent release + flag) as pure overhead "
                                                         "inside the real kernel: +3.1 % per step (+4 us per layer); it could hide at most the 1.75 us boundary "
                                                         "and the neighbour-independent part of the 9 us prologue: net < 1 %: not built",
                    "chip_filling_batch": "36 clips per batch (252 of 256 CUs hold a tile): +4 ... 5 % per clip (`job_256.chip_filling_batches` on this line)",
                    "vocoder_under_next_batch": "second stream: +-0.1 % (`job_256` vs `job_256.no_overlap` on this line): the layer kernels leave no CU a co-resident workgroup fits on",
                    "residual_skip_3_bytes_per_element": "byte ablation of the output phase (3 of the 4 dwordx4 per lane and tile of the fp32 residual / skip "
                                                         "read-modify-write, -22 % of that phase's bytes; wrong results): -1.5 % per step (the in-kernel stamps: "
                                                         "-1.8 us of the 41 us output phase) -- the phase is not HBM-byte-bound, a 3-byte residual / skip format is not built",
                    "infinity_cache_residency": "plain instead of non-temporal accesses for the 132 MB a batch re-touches every layer: +-0.1 %",
                    "output_phase_waves_out_of_step": "waves 4-7 (or 0-3) delayed by 1.7 ... 5 us after `g complete` so that a SIMD's two waves alternate MFMA loop / "
                                                      "store + init-load wait: -0.8 ... +0.8 % (noise)"},
                "source": "profiles/r6c_layer_ablations.txt, profiles/r6b_bench.json, profiles/r6m_stream_ablate.txt, profiles/r6m_dephase.txt, "
                          "profiles/r6m_layer_stamps_w6.txt (same-box A/B, profiling build for the ablations)"}
            fit, why = load_error_fit(precb)
            if fit:
                import math
                p1 = 1.0 - math.exp(-math.exp(-(1e-3 - fit["gumbel_mu"]) / fit["gumbel_beta"]))
                result["batched"]["mel_error_vs_reference"] = dict(fit, bar=1e-3, p_clip_over_bar=p1, p_over_bar_per_256_clips=1.0 - (1.0 - p1) ** 256)
            else:
                result["batched"]["mel_error_vs_reference"] = None
                result["batched"]["mel_error_vs_reference_missing"] = why
            # the sustained MFMA rate again, on the chip as the batched run leaves it (hot, clocks settled)
            sustained["after_batched"] = probe_mfma()
            sustained.update(operands="random fp16, register-resident v_mfma_f32_32x32x16_f16 loop, 2 waves/SIMD, every CU",
                             datasheet_tflops=PEAK_TFLOPS_F16)
            result["mfma_sustained"] = sustained
            for r_ in (roof, broof):
                tf_ = r_["achieved"] if r_["bound"] == "mfma" else r_.get("mfma_tflops")
                if tf_:
                    r_["frac_of_sustained"] = tf_ / sustained["cold"]["tflops"]       # MFMA side, against what the part holds on real data
                    r_["pipe_frac_of_sustained"] = r_["pipe_tflops"] / sustained["cold"]["tflops"]
                if r_["bound"] == "hbm":
                    r_["frac_of_stream_copy"] = r_["achieved"] / 6290.0               # against the 6.29 TB/s a float4 copy reaches (MI355X_MICROARCH.md)
            try:    # north_star's 1-GPU denominator: the WHOLE 256-clip job on this one device (8 batches of 32, vocoder under the next batch's DDPM),
                    # and a job of chip-filling batches (36 clips = 252 frame tiles on 256 CUs; 32 clips leave 32 CUs without a workgroup)
                result["job_256"] = time_job(pipe, 256, 32, dev, args.ddpm_steps, overlap=True)
                result["job_256"]["no_overlap"] = {k: v for k, v in time_job(pipe, 64, 32, dev, args.ddpm_steps, overlap=False, warm=False).items()
                                                   if k in ("clips", "value", "s_per_job")}
                result["job_256"]["chip_filling_batches"] = time_job(pipe, 252, 36, dev, args.ddpm_steps, overlap=True)
                result["job_256"]["note"] = ("256 clips need 8 batches at any batch size <= 36, so the 256-clip job runs 8 x 32; `chip_filling_batches` is a "
                                             "252-clip job as 7 x 36 (252 of 256 CUs hold a frame tile); `no_overlap` = 64 clips as 2 x 32 with the vocoder on "
                                             "the sampler's stream")
            except Exception as ex:
                result["job_256"] = {"error": repr(ex)[:300]}
            if FAST_SIDE and precb != FAST_SIDE:
                # other operand schemes beside the shipped ones (design/precision.md): f16_w2 -- round 3's batched default, and what a single clip runs
                # at when the fp32-class f16_x3t is not asked for -- and f16_w6n (f16_w6 without the gate-output correction: f16_w2's error class)
                fs = {"unit": "audio-sec/wall-sec"}
                for scheme, with_single, with_batch in (("f16_w2", True, True), (FAST_SIDE, False, True)):
                    if scheme == prec:
                        continue
                    try:
                        pf = SvcPipeline(hp, sd, vs, h, precision=scheme, vocoder_precision="f16_x3")
                        fs[scheme] = {}
                        if with_single:
                            pf.infer(hub, m2p, f0, seed=3, clip_ids=clip_ids)
                            torch.cuda.synchronize(); t1 = time.perf_counter()
                            pf.infer(hub, m2p, f0, seed=4, clip_ids=clip_ids)
                            torch.cuda.synchronize(); t1 = time.perf_counter() - t1
                            fs[scheme]["value"] = CLIP_SECONDS / t1
                        if with_batch:
                            pf.model.hp = dict(hp, K_step=30); pf.model.K_step = 30
                            pf.infer(hb, mb, fb, seed=1)
                            pf.model.K_step = args.ddpm_steps
                            torch.cuda.synchronize(); t2 = time.perf_counter()
                            pf.infer(hb, mb, fb, seed=2)
                            torch.cuda.synchronize(); t2 = time.perf_counter() - t2
                            fs[scheme]["batched_value"] = Bb * CLIP_SECONDS / t2
                        del pf
                    except Exception as ex:
                        fs[scheme] = {"error": repr(ex)[:200]}
                fs["note"] = ("not the shipped precisions: f16_w2 (round 3's batched default: 6.2e-4 ... 9.1e-4 on the goldens, one 256-clip job in four "
                              "holds a clip over 1e-3 on the random-init probe) and f16_w6n (the shipped f16_w6 without the 6-bit correction of the gate "
                              "output: f16_w2's error class); a single clip gets the fp32-class f16_x3t")
                result["faster_scheme"] = fs
        if world == 1 and B == 1 and args.speedup <= 1 and not args.no_batched:
            # the stages either side of the sampler, one 10 s clip each (informational; `value` above is cond -> PCM on the device)
            try:
                def timed(fn, n=5):
                    fn(); torch.cuda.synchronize(); t = time.perf_counter()
                    for _ in range(n):
                        fn()
                    torch.cuda.synchronize()
                    return (time.perf_counter() - t) / n * 1e3
                stages = {}
                mel1 = (torch.randn(1, T_FRAMES, hp["audio_num_mel_bins"], device=dev) * 0.5 - 2.5).clamp(hp["mel_vmin"], hp["mel_vmax"])
                f01 = torch.full((1, T_FRAMES), 220.0, device=dev)
                stages["vocoder_ms"] = timed(lambda: pipe.vocoder.vocode(mel1, f01, seed=1))
                vtf = VOCODER_FLOP_PER_FRAME * T_FRAMES / (stages["vocoder_ms"] * 1e-3) / 1e12
                stages["vocoder_roofline"] = {"bound": "mfma", "algorithmic_gflop_per_clip": VOCODER_FLOP_PER_FRAME * T_FRAMES / 1e9, "achieved": vtf,
                                              "peak": PEAK_TFLOPS_F16, "unit": "TFLOP/s", "frac": vtf / PEAK_TFLOPS_F16, "mfma_per_product": 3,
                                              "pipe_frac": 3 * vtf / PEAK_TFLOPS_F16,
                                              "scope": "the whole generator for one 10 s clip (29 launches), split fp16 operands (f16_x3)", **load_voc_kernels()}
                # what the reference's host glue adds around the device path (infer_tool.py:174,200: mel / f0 to numpy, PCM to numpy)
                stages["host_round_trip_ms"] = timed(lambda: (mel1.cpu().numpy(), f01.cpu().numpy(), wav[:1].cpu().numpy()))
                from diffsvc_amd.hubert import HubertSoftHip
                from diffsvc_amd.pe import PitchExtractorHip
                hsoft = HubertSoftHip(synth.hubert_state(11))
                w16 = torch.from_numpy(synth.speech_like_wav(1, 160000)).to(dev)
                stages["hubert_soft_ms"] = timed(lambda: hsoft.units(w16))
                del hsoft
                hp24 = dict(synth.HPARAMS_24K)
                pe = PitchExtractorHip(hparams=hp24).cuda()
                pe.load_state_dict(synth.pe_state(hp24, 5))
                mel24 = torch.from_numpy(synth.mel_like(1, 1, 1875, 80)).to(dev)
                stages["pitch_extractor_ms"] = timed(lambda: pe(mel24))
                del pe
                # north_star's chain starts at the content encoder: the same clip with HuBERT-soft's time added (`value` itself starts at the units,
                # where the reference's encoder wrapper finds them cached beside the wav: preprocessing/hubertinfer.py:35-38)
                stages["value_with_hubert"] = result["value"] * (elapsed / args.steps) / (elapsed / args.steps + stages["hubert_soft_ms"] * 1e-3 * B)
                stages["note"] = ("one 10 s clip: NSF-HiFiGAN alone (inside `value`), HuBERT-soft on 160 000 samples at 16 kHz and the 24 kHz pitch "
                                  "extractor on 1875 frames (outside `value`: upstream / config B; `value_with_hubert` = `value` with the encoder's "
                                  "time per clip added), D2H of mel + f0 + PCM")
                result["stages"] = stages
            except Exception as ex:
                result["stages"] = {"error": repr(ex)[:300]}
        if world == 1 and B == 1 and args.speedup <= 1 and not args.no_batched:
            # fp32-class operands (f16_x3t: split activations + hi/lo weights, 1e-5-class evaluations) where they are NOT the default: the
            # 32-clip batch -- what the batched configuration costs when nothing is traded for the fp16 activation rounding
            try:
                pipe = None
                torch.cuda.empty_cache()
                pipe3 = SvcPipeline(hp, sd, vs, h, precision="f16_x3t", vocoder_precision="f16_x3")
                pipe3.model.hp = dict(hp, K_step=30); pipe3.model.K_step = 30
                pipe3.infer(hb, mb, fb, seed=1)
                pipe3.model.K_step = args.ddpm_steps
                torch.cuda.synchronize(); t3 = time.perf_counter()
                pipe3.infer(hb, mb, fb, seed=2)
                torch.cuda.synchronize(); t3 = time.perf_counter() - t3
                result["fp32_class"] = {"precision": "f16_x3t", "batched_value": 32 * CLIP_SECONDS / t3, "unit": "audio-sec/wall-sec", "s_per_batch": t3,
                                        "workload": "32 x 10 s clips in one batch at split-fp16 (fp32-class) operands; the single clip (`value`) "
                                                    "runs at this precision by default"}
                del pipe3
            except Exception as ex:
                result["fp32_class"] = {"error": repr(ex)[:300]}
        if world == 1 and B == 1 and args.speedup <= 1 and not args.no_batched:
            # BASELINE configs[4]: the training step (64 clips x 128 frames: diffusion loss forward + backward + clip + AdamW)
            try:
                torch.cuda.empty_cache()
                ms_t, loss_t, _ = time_train_steps(hp, sd, 64, 128, 5, 2, 0, dev, torch.cuda.synchronize)
                tfl = train_step_flops(hp, 64 * 128) / (ms_t * 1e-3) / 1e12
                result["train_step"] = {"workload": "BASELINE configs[4]: diffusion loss fwd+bwd + AdamW on a 64 x 128-frame mel batch, 1 GPU",
                                        "ms_per_step": ms_t, "value": 64 * 128 / (ms_t * 1e-3), "unit": "frames/s", "final_loss": loss_t,
                                        "precision": "split fp16 operands (fp32-class), fp32 master weights",
                                        "roofline": {"bound": "mfma", "scope": "whole step (forward + backward + clip + AdamW), not one kernel: the largest "
                                                     "kernels are wgrad_fm_kernel (20 %: the weight gradients, contracted from the frame-major planes through the "
                                                     "transposing LDS read) and the transposed conv on the tgemm engine (17 %) (profiles/r5G_kernel_stats_train.csv)",
                                                     "algorithmic_tflop_per_step": train_step_flops(hp, 64 * 128) / 1e12, "achieved": tfl,
                                                     "peak": PEAK_TFLOPS_F16, "unit": "TFLOP/s", "frac": tfl / PEAK_TFLOPS_F16, "mfma_per_product": 3,
                                                     "pipe_frac": 3 * tfl / PEAK_TFLOPS_F16, "traffic": load_train_traffic()[0],
                                                     "traffic_source": load_train_traffic()[1], "traffic_unit": "HBM bytes per step (PMC)", **load_train_kernels()}}
                # the reference's max_tokens loader changes (B, T) every step (training/task/tts.py:60-88): ONE trainer alternating through five
                # batch shapes of about the benchmarked size (8 192 frames each; only the gap rows are cleared on a shape change, design/training.md)
                try:
                    result["train_step"]["variable_shape_ms"] = time_variable_shape_steps(hp, sd, dev)
                except Exception as ex:
                    result["train_step"]["variable_shape_ms"] = {"error": repr(ex)[:200]}
            except Exception as ex:                                           # never lose the inference line to the extra measurement
                result["train_step"] = {"error": repr(ex)[:300]}
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(hp, sd, vs, h)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
