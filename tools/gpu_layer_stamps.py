"""Timeline of the fused residual-layer kernel at 32 clips from in-kernel s_memrealtime stamps (100 MHz), PROFILING build only
(python -m diffsvc_amd.build --profiling).  Per wave: 0 entry, 1 prologue loads issued, 2 they landed, 3 barrier passed, 4-6 end of the gate
passes' main loops, 7 g complete (phase boundary), 8/10/12 end of the output passes' main loops, 9/11/13 their stores issued, 14 stores
acknowledged.      DSVC_TL_STAMPS=1 python tools/gpu_layer_stamps.py <precision>"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DSVC_TL_STAMPS"] = "1"
import numpy as np, torch
import diffsvc_amd
from diffsvc_amd import _lib, synth
_lib.use_profiling_build()
from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
prec = sys.argv[1] if len(sys.argv) > 1 else "f16_w2"
prio = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hp = dict(synth.HPARAMS_44K)
sd = synth.acoustic_state(hp, 0)
den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=prec, prefix="denoise_fn.")
smp = SamplerHandle(den, sd)
den.debug_set("layer_prio", prio)
cond = torch.randn(32, 256, 861, device="cuda") * 0.5
smp.sample(cond, 24, seed=1, use_graph=False)
torch.cuda.synchronize()
lib = _lib.lib()
buf = np.zeros(4096 * 8 * 16, dtype=np.uint64)
groups = ctypes.c_int32(0)
rc = lib.dsvc_profile_layer_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.byref(groups))
assert rc == 0, rc
g = groups.value
st = buf[:g * 8 * 16].reshape(g, 8, 16).astype(np.int64)
t0 = st[:, :, 0].min()
rel = (st - t0) * 0.01                                     # us since the first wave of the launch entered
names = ["entry", "prologue issued", "prologue landed", "barrier", "gate loop 0", "gate loop 1", "gate loop 2", "g complete",
         "out loop 0", "out stores 0", "out loop 1", "out stores 1", "out loop 2", "out stores 2", "stores acked"]
print("layer_prio %d; " % prio, end="")
print("%s, last layer of the chain (dilation 8), %d workgroups x 8 waves; us since the first wave's entry: mean [min .. max] | mean step" % (prec, g))
prev = None
for i, n in enumerate(names):
    v = rel[:, :, i]
    if (st[:, :, i] == 0).all():
        continue
    m = v.mean()
    print("  %2d %-16s %7.2f  [%7.2f .. %7.2f]   %+7.2f" % (i, n, m, v.min(), v.max(), (m - prev) if prev is not None else 0.0))
    prev = m
for half, sl in (("waves 0-3 (priority 0)", slice(0, 4)), ("waves 4-7 (priority 1)", slice(4, 8))):
    print("  " + half + ": " + "  ".join("%s %.1f" % (names[i].split()[0] + str(i), rel[:, sl, i].mean()) for i in (3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14)))
