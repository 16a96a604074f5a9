"""Which skip tail is accurate?  One DiffNet evaluation at 8 x 861 (f16_w2): the per-layer gate outputs g_l are read back from the device
(they are bit-identical in both forms), the tail  eps = W_fin relu(W_sp (sum_l (W_s,l g_l + b_s,l)) / sqrt(L) + b_sp) + b_fin  is evaluated
in fp64 on the CPU from THOSE g_l, and compared with the device's eps in the deferred (tskip.h) and the in-layer form."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np, torch
import diffsvc_amd
from diffsvc_amd import _lib as _dsvc_lib
_dsvc_lib.hooks_build().__enter__()      # this tool sets dsvc_*_debug_set keys: they exist in the test-hooks build only (round 6)
from diffsvc_amd import synth
from diffsvc_amd.engine import DenoiserHandle
prec = sys.argv[1] if len(sys.argv) > 1 else "f16_w2"
hp = dict(synth.HPARAMS_44K)
sd = synth.acoustic_state(hp, 0)
den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=prec, prefix="denoise_fn.")
B, T, C, L, M = 8, 861, 384, 20, 128
g = np.random.Generator(np.random.PCG64(23))
spec = torch.from_numpy(g.standard_normal((B, 1, M, T)).astype(np.float32)).cuda()
cond = torch.from_numpy((g.standard_normal((B, 256, T)) * 0.5).astype(np.float32)).cuda()
t = torch.from_numpy(g.integers(0, 1000, size=(B,))).cuda()
Tp = (T + 8 + 127) // 128 * 128          # csrc/diffnet.hip: bucket_rows
den.debug_set("two_launch_layer", -1)
den.debug_set("defer_skip", 1)
out_d = den.forward(spec, t, cond).cpu()
gall = den.debug_buffer("gall").cpu().double()              # [L * rows, C]
s2_d = den.debug_buffer("s2").cpu()
rows = gall.shape[0] // L
den.debug_set("defer_skip", 0)
out_i = den.forward(spec, t, cond).cpu()
s2_i = den.debug_buffer("s2").cpu()
skip_i = den.debug_buffer("skip").cpu().double()
p = lambda k: sd["denoise_fn." + k].double()
skip = torch.zeros(rows, C, dtype=torch.float64)
for l in range(L):
    wo, bo = p("residual_layers.%d.output_projection.weight" % l)[C:, :, 0], p("residual_layers.%d.output_projection.bias" % l)[C:]
    skip += gall[l * rows:(l + 1) * rows] @ wo.T + bo
s = torch.relu((skip / L ** 0.5) @ p("skip_projection.weight")[:, :, 0].T + p("skip_projection.bias"))
eps = s @ p("output_projection.weight")[:, :, 0].T + p("output_projection.bias")          # [rows, M]
sel = torch.cat([torch.arange(b * Tp, b * Tp + T) for b in range(B)])
ref = eps[sel].reshape(B, T, M).permute(0, 2, 1)[:, None]
for name, o in (("deferred (tskip.h)", out_d), ("in-layer skip sum ", out_i)):
    d = (o.double() - ref).abs()
    print("%s %s: eps vs the fp64 tail of the SAME g_l: max %.3e rms %.3e   (|eps| rms %.3f)" % (prec, name, d.max().item(), d.pow(2).mean().sqrt().item(), ref.pow(2).mean().sqrt().item()))
print("in-layer running skip sum vs fp64 sum: max %.3e (|skip| rms %.2f)" % ((skip_i[sel] - skip[sel]).abs().max().item(), skip[sel].pow(2).mean().sqrt().item()))
for name, s2 in (("deferred", s2_d), ("in-layer", s2_i)):
    print("relu(skip proj) hi plane, %s: max |err| %.3e (|s| max %.2f)" % (name, (s2[sel].double() - s[sel]).abs().max().item(), s[sel].abs().max().item()))
print("deferred vs in-layer eps: max |diff| %.3e" % (out_d - out_i).abs().max().item())
