"""f16_w6 on 128- vs 64- / 32-frame tiles: where do the tilings first differ?  One evaluation at 8 x 861, stopped after n layers; residual stream,
skip sum and the next layer's operand of every tiling against the 128-frame tiling's and against the oracle's fp32 taps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import diffsvc_amd
from diffsvc_amd import _lib as _dsvc_lib
_dsvc_lib.hooks_build().__enter__()      # this tool sets dsvc_*_debug_set keys: they exist in the test-hooks build only (round 6)
from diffsvc_amd import synth
from diffsvc_amd.engine import DenoiserHandle
import dsvc_oracle as O
hp = dict(synth.HPARAMS_44K)
sd = synth.acoustic_state(hp, 0)
B, T = 8, 861
g = np.random.Generator(np.random.PCG64(23))
spec = torch.from_numpy(g.standard_normal((B, 1, 128, T)).astype(np.float32))
cond = torch.from_numpy((g.standard_normal((B, 256, T)) * 0.5).astype(np.float32))
t = torch.full((B,), 417, dtype=torch.long)
Tp = (T + 8 + 127) // 128 * 128          # csrc/diffnet.hip: bucket_rows
taps = {}
with torch.no_grad():
    O.diffnet_forward(sd, spec, t, cond, 4, taps=taps)
for prec, knobs in (("f16_w6", {}), ("f16_w6", {"g6_off": 1}), ("f16_w6n", {})):
    den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=prec, prefix="denoise_fn.")
    for k, v in knobs.items():
        den.debug_set(k, v)
    for n in (1, 2, 20):
        den.debug_set("stop_after_layers", n)
        bufs = {}
        for nt in (4, 2, 1):
            den.debug_set("fused_nt", nt)
            den.forward(spec.cuda(), t.cuda(), cond.cuda())
            bufs[nt] = {k: den.debug_buffer(k).cpu() for k in ("xres", "skip", "xh")}
        ref_x = torch.cat([taps["x%d" % (n - 1)][b].T for b in range(B)])
        ref_s = torch.cat([taps["s%d" % (n - 1)][b].T for b in range(B)])
        rows = torch.cat([torch.arange(b * Tp, b * Tp + T) for b in range(B)])
        line = "%s %s after %2d layers:" % (prec, knobs, n)
        for nt in (4, 2, 1):
            ex = (bufs[nt]["xres"][rows] - ref_x).abs().max().item()
            es = (bufs[nt]["skip"][rows] - ref_s).abs().max().item()
            dx = (bufs[nt]["xres"] - bufs[4]["xres"]).abs().max().item()
            ds = (bufs[nt]["skip"] - bufs[4]["skip"]).abs().max().item()
            dh = (bufs[nt]["xh"] - bufs[4]["xh"]).abs().max().item()
            line += "  | nt=%d: x err %.2e skip err %.2e; vs nt=4: x %.2e skip %.2e xh %.2e" % (nt, ex, es, dx, ds, dh)
        print(line, flush=True)
    den.debug_set("stop_after_layers", -1)
    del den

# where the 128- and 64-frame tilings of f16_w6 differ after ONE layer: rows modulo 128 and channels
den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision="f16_w6", prefix="denoise_fn.")
den.debug_set("stop_after_layers", 1)
out = {}
for nt in (4, 2):
    den.debug_set("fused_nt", nt)
    den.forward(spec.cuda(), t.cuda(), cond.cuda())
    out[nt] = {k: den.debug_buffer(k).cpu() for k in ("xres", "skip")}
for k in ("xres", "skip"):
    d = (out[2][k] - out[4][k]).abs()
    nz = d > 0
    print(k, "elements that differ: %d of %d; rows %d of %d" % (nz.sum().item(), nz.numel(), nz.any(1).sum().item(), nz.shape[0]))
    rows = nz.any(1).nonzero().flatten()
    print("  differing rows modulo 128, histogram by 32-row N-tile:", torch.bincount((rows % 128) // 32, minlength=4).tolist(), " modulo 32:", torch.bincount(rows % 32, minlength=32).tolist())
    cols = nz.any(0).nonzero().flatten()
    print("  differing channels: %d; by 32-channel tile:" % cols.numel(), torch.bincount(cols // 32, minlength=12).tolist())
    print("  per differing row, number of differing channels (first 10):", nz.sum(1)[rows[:10]].tolist(), "rows", rows[:10].tolist())
    print("  max |diff| %.2e, mean over differing %.2e" % (d.max().item(), d[nz].mean().item()))
