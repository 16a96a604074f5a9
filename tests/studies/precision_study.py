"""CPU study (test infrastructure): which operand-precision scheme for the two big per-layer contractions keeps a
full 1000-step DDPM chain within the 1e-3 mel bar?  Emulates the MFMA numerics (fp16 operands, fp32 accumulate) in
PyTorch-CPU on top of the oracle.  Schemes:
  f16        w -> fp16 (nearest), x -> fp16                       1 MFMA / product
  w2         w exact (hi+lo), x -> fp16                           2 MFMAs
  ditherK    w -> one of K fp16 roundings, variant = step % K; the K roundings average to w    1 MFMA
Usage: python tests/studies/precision_study.py [T] [steps] [schemes...]
"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch, torch.nn.functional as F
import diffsvc_amd
from diffsvc_amd import synth
import dsvc_oracle as O
from util import clip_batch

def fp16_floor_ceil(w):
    """(largest fp16 <= w, smallest fp16 >= w) as float32 arrays."""
    h = w.astype(np.float16)
    hf = h.astype(np.float32)
    up = np.nextafter(h, np.float16(np.inf)).astype(np.float32)
    dn = np.nextafter(h, np.float16(-np.inf)).astype(np.float32)
    lo = np.where(hf <= w, hf, dn)
    hi = np.where(hf >= w, hf, up)
    return lo, hi

def dither_variants(w, K, seed=0):
    """K fp16 roundings of w whose mean is w +- ulp/(2K): variant k rounds up iff frac(w) > u_k, with the thresholds
    u_k = ((k + 0.5)/K + phase(element)) mod 1 -- a per-element random phase decorrelates elements."""
    w = w.numpy().astype(np.float32)
    lo, hi = fp16_floor_ceil(w)
    span = hi - lo
    frac = np.where(span > 0, (w - lo) / np.where(span > 0, span, 1), 0.0)
    rng = np.random.Generator(np.random.PCG64(seed))
    phase = rng.random(w.shape).astype(np.float32) if K > 1 else np.zeros_like(w)
    out = []
    for k in range(K):
        u = ((k + 0.5) / K + phase) % 1.0
        out.append(torch.from_numpy(np.where(frac > u, hi, lo).astype(np.float32)))
    return out

def r16(x):
    return x.half().float()

# ---- round 4: the w_lo * x correction on the block-scaled 6-bit matrix instruction (v_mfma_scale_f32_32x32x64_f8f6f4) ----
E2M3 = np.array([m / 8 for m in range(8)] + [(1 + m / 8) * 2 ** e for e in range(3) for m in range(8)], dtype=np.float64)     # fp6: 0 .. 7.5
E3M2 = np.array([m / 16 for m in range(4)] + [(1 + m / 4) * 2.0 ** (e - 3) for e in range(1, 8) for m in range(4)], dtype=np.float64)  # bf6: 0 .. 28

def grid_floor_ceil(u, grid):
    """(largest grid value <= |u|, smallest >= |u|) with the sign restored; saturating at the grid's end."""
    a = np.minimum(np.abs(u), grid[-1])
    i = np.searchsorted(grid, a, side="right") - 1
    lo = grid[i]; hi = grid[np.minimum(i + 1, len(grid) - 1)]
    hi = np.where(lo == a, lo, hi)
    return np.sign(u) * lo, np.sign(u) * hi      # (towards zero, away from zero)

def q_nearest(u, grid):
    lo, hi = grid_floor_ceil(u, grid)
    return np.where(np.abs(u - lo) <= np.abs(hi - u), lo, hi)

def lo6_variants(w, K, grid, block=32, seed=0, blockscale=True):
    """K 6-bit roundings of w_lo = w - fp16(w) ([O, I, taps]): per (output row, tap, 32 input channels) block a power-of-two scale that
    maps the block's largest magnitude just inside the grid, element values rounded towards / away from zero by stratified thresholds
    (the K copies average to w_lo to 1/K of a grid step)."""
    w = w.numpy().astype(np.float64)
    wl = w - w.astype(np.float16).astype(np.float64)
    O_, I, taps = wl.shape
    b = wl.transpose(0, 2, 1).reshape(O_, taps, I // block, block)
    amax = np.abs(b).max(-1, keepdims=True) if blockscale else np.full((O_, taps, I // block, 1), np.abs(b).max())
    e = np.ceil(np.log2(np.maximum(amax, 1e-30) / grid[-1]))
    sc = 2.0 ** e
    u = b / sc
    lo, hi = grid_floor_ceil(u, grid)
    span = hi - lo
    frac = np.where(span != 0, (u - lo) / np.where(span != 0, span, 1), 0.0)
    rng = np.random.Generator(np.random.PCG64(seed + 1000))
    phase = rng.random(u.shape) if K > 1 else np.zeros_like(u)
    out = []
    for k in range(K):
        th = ((k + 0.5) / K + phase) % 1.0 if K > 1 else 0.5
        q = np.where(frac > th, hi, lo) * sc
        out.append(torch.from_numpy(q.reshape(O_, taps, I).transpose(0, 2, 1).astype(np.float32).copy()))
    return out

def x6(x, grid, scale):
    """activations for the 6-bit product: nearest on the grid after division by `scale` (saturating), as v_cvt_scalef32_pk32_*_f16 does"""
    return torch.from_numpy((q_nearest(x.numpy().astype(np.float64) / scale, grid) * scale).astype(np.float32))

class Net:
    def __init__(self, sd, hp, scheme):
        self.sd, self.hp, self.scheme = sd, hp, scheme
        self.L = hp["residual_layers"]; self.C = hp["residual_channels"]; self.cyc = hp["dilation_cycle_length"]
        self.K = 1
        p = "denoise_fn.residual_layers.%d.%s"
        self.hold = 1
        if scheme.startswith("dither"):
            body = scheme[6:]
            if "h" in body:                     # ditherNhR: each variant is kept for R consecutive steps
                body, hold = body.split("h")
                self.hold = int(hold)
            self.K = int(body)
        self.wd, self.wo = [], []
        for l in range(self.L):
            wd, wo = sd[p % (l, "dilated_conv.weight")], sd[p % (l, "output_projection.weight")]
            if scheme.startswith("w6"):            # w6[b|f]<K>[xb|xf][s<scale>]: gate = fp16(w) x + lo6_k x6; output projection exact (w2)
                self.wd.append([r16(wd)]); self.wo.append([wo])
            elif scheme == "f32" or scheme == "w2":
                self.wd.append([wd]); self.wo.append([wo])
            elif scheme == "f16":
                self.wd.append([r16(wd)]); self.wo.append([r16(wo)])
            else:
                self.wd.append(dither_variants(wd, self.K, 2 * l)); self.wo.append(dither_variants(wo, self.K, 2 * l + 1))
        self.lo6 = None
        if scheme.startswith("w6"):
            import re
            m = re.match(r"w6([bf])(\d+)(x[bf])?(s\d+)?(g)?$", scheme)
            wgrid = E3M2 if m.group(1) == "b" else E2M3
            self.K = int(m.group(2))
            self.xgrid = E2M3 if m.group(3) == "xf" else E3M2
            self.xscale = float(m.group(4)[1:]) if m.group(4) else 4.0
            self.lo6 = [lo6_variants(sd[p % (l, "dilated_conv.weight")], self.K, wgrid, seed=l, blockscale=not m.group(5)) for l in range(self.L)]
        self.order = list(range(self.K))
        if self.K > 1:      # bit-reversed visiting order: consecutive steps use far-apart thresholds
            bits = int(math.log2(self.K))
            self.order = [int(format(i, "0%db" % bits)[::-1], 2) for i in range(self.K)] if 2 ** bits == self.K else self.order

    def forward(self, spec, t, cond, cproj):
        sd = self.sd; P = "denoise_fn."
        p = lambda k: sd[P + k]
        C = self.C
        act = (lambda v: v) if self.scheme == "f32" else r16
        flags = os.environ.get("STUDY_ROUND", "state,skip,s2").split(",")      # which of the small projections see fp16 inputs
        a_state = act if "state" in flags else (lambda v: v)
        a_skip = act if "skip" in flags else (lambda v: v)
        a_s2 = act if "s2" in flags else (lambda v: v)
        x = F.relu(F.conv1d(a_state(spec[:, 0]), p("input_projection.weight"), p("input_projection.bias")))   # tgemm path: fp16 state copy
        emb = O.step_embedding(sd, t)
        skip = torch.zeros_like(x)
        k = self.order[(int(t[0]) // self.hold) % self.K]
        for l in range(self.L):
            q = lambda s: p("residual_layers.%d.%s" % (l, s))
            d = 2 ** (l % self.cyc)
            film = F.linear(emb, q("diffusion_projection.weight"), q("diffusion_projection.bias"))[:, :, None]
            which = os.environ.get("STUDY_ACT", "xg")          # which of the two big contractions see fp16-rounded activations
            act_x = act if "x" in which else (lambda v: v)
            act_g = act if "g" in which else (lambda v: v)
            if "G6" in which:                                   # g = fp16(g) + bf6((g - fp16(g)) * 2^11) * 2^-11: the output 1x1 with a 6-bit g_lo correction
                act_g = lambda v: r16(v) + x6((v - r16(v)) * 2048.0, E3M2, 1.0 / 16) / 2048.0
            if "X6" in which:
                act_x = lambda v: r16(v) + x6((v - r16(v)) * 2048.0, E3M2, 1.0) / 2048.0
            xin = act_x(x + film)
            y = F.conv1d(xin, self.wd[l][k % len(self.wd[l])], q("dilated_conv.bias"), padding=d, dilation=d) + cproj[l]
            if self.lo6 is not None:
                y = y + F.conv1d(x6(xin, self.xgrid, self.xscale), self.lo6[l][k], None, padding=d, dilation=d)
            z = torch.sigmoid(y[:, :C]) * torch.tanh(y[:, C:])
            o = F.conv1d(act_g(z), self.wo[l][k % len(self.wo[l])], q("output_projection.bias"))
            x = (x + o[:, :C]) / math.sqrt(2.0)
            skip = skip + o[:, C:]
        s = F.relu(F.conv1d(a_skip(skip), p("skip_projection.weight") / math.sqrt(self.L), p("skip_projection.bias")))   # fp16(skip), 1/sqrt(L) in the weights
        return F.conv1d(a_s2(s), p("output_projection.weight"), p("output_projection.bias"))[:, None]

def chain(net, sd, hp, cond_t, x, seed, clips, T, steps, K_total):
    M = hp["audio_num_mel_bins"]
    cproj = [F.conv1d(cond_t, sd["denoise_fn.residual_layers.%d.conditioner_projection.weight" % l],
                      sd["denoise_fn.residual_layers.%d.conditioner_projection.bias" % l]) for l in range(net.L)]
    snaps = {}
    with torch.no_grad():
        for i in reversed(range(K_total - steps, K_total)):
            t = torch.full((x.shape[0],), i, dtype=torch.long)
            eps = net.forward(x, t, cond_t, cproj)
            x = O.ddpm_update(sd, x, eps, t, O.ddpm_noise_ref_layout(seed, clips, i, T, M))
    return x

def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    schemes = sys.argv[3:] or ["f32", "f16", "w2", "dither2", "dither4", "dither8", "dither16"]
    torch.set_num_threads(int(os.environ.get("STUDY_THREADS", "8")))
    hp = dict(synth.HPARAMS_44K)
    sd = synth.acoustic_state(hp, 0)
    clips, seed = [0], 2024
    hub, m2p, f0 = clip_batch(hp, clips, T, max(2, int(T * 500 / 861)))
    cond, _, _ = O.build_cond(sd, hub, m2p, f0.clone(), hp)
    cond_t = cond.transpose(1, 2).contiguous()
    M = hp["audio_num_mel_bins"]
    x0 = O.ddpm_noise_ref_layout(seed, clips, 0, T, M, O.PURPOSE_X_INIT)
    ref = None
    for s in schemes:
        t0 = time.time()
        x = chain(Net(sd, hp, s), sd, hp, cond_t, x0.clone(), seed, clips, T, steps, 1000)
        mel = O.finish_mel(sd, x, m2p)
        if ref is None:
            ref = mel
        err = (mel - ref).abs()
        print("%-9s mel max-abs err %.3e  mean %.3e  p99.9 %.3e   (%.1fs)" % (s, err.max().item(), err.mean().item(),
              err.flatten().kthvalue(int(err.numel() * 0.999)).values.item(), time.time() - t0), flush=True)

if __name__ == "__main__":
    main()
