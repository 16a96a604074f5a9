"""Round 4 diagnostic: the residual stream and the skip sum after ONE fused layer (and after all 20) against the fp32 oracle, for f16_w2,
f16_w6 without and with the 6-bit g_lo correction of the output projection.  With the correction the layer-1 error must DROP (only the
fp16 rounding of the conv input is left); a broken code path shows as an error far above f16_w2's.
    python tools/gpu_g6_diag.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import diffsvc_amd
from diffsvc_amd import _lib as _dsvc_lib
_dsvc_lib.hooks_build().__enter__()      # this tool sets dsvc_*_debug_set keys: they exist in the test-hooks build only (round 6)
from diffsvc_amd import synth
from diffsvc_amd.engine import DenoiserHandle
import dsvc_oracle as O

hp = dict(synth.HPARAMS_44K, K_step=20)
sd = synth.acoustic_state(hp, 0)
B, T = 8, 861
g = np.random.Generator(np.random.PCG64(23))
spec = torch.from_numpy(g.standard_normal((B, 1, 128, T)).astype(np.float32))
cond = torch.from_numpy((g.standard_normal((B, 256, T)) * 0.5).astype(np.float32))
t = torch.from_numpy(g.integers(0, 1000, size=(B,)))
Tp = (T + 8 + 127) // 128 * 128          # csrc/diffnet.hip: bucket_rows
taps = {}
with torch.no_grad():
    ref_out = O.diffnet_forward(sd, spec, t, cond, 4, taps=taps)
for name, prec, g6off in (("f16_w2", "f16_w2", 0), ("f16_w6 no g_lo", "f16_w6d1", 1), ("f16_w6 + g_lo", "f16_w6d1", 0)):
    den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=prec, prefix="denoise_fn.")
    den.debug_set("two_launch_layer", -1)
    if g6off:
        den.debug_set("g6_off", 1)
    for n in (1, 2, 4):
        den.debug_set("stop_after_layers", n)
        den.forward(spec.cuda(), t.cuda(), cond.cuda())
        bx = den.debug_buffer("xres").cpu()
        ex = torch.cat([(bx[b * Tp:b * Tp + T] - taps["x%d" % (n - 1)][b].T) for b in range(B)])
        print("%-16s after %d layer(s): residual stream err rms %.3e max %.3e" % (name, n, ex.pow(2).mean().sqrt().item(), ex.abs().max().item()), flush=True)
    den.debug_set("stop_after_layers", -1)
    out = den.forward(spec.cuda(), t.cuda(), cond.cuda()).cpu()
    e = out - ref_out
    print("%-16s eps err rms %.3e max %.3e" % (name, e.pow(2).mean().sqrt().item(), e.abs().max().item()), flush=True)
