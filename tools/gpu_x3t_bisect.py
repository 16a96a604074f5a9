import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np, torch
import diffsvc_amd
from diffsvc_amd import _lib as _dsvc_lib
_dsvc_lib.hooks_build().__enter__()      # this tool sets dsvc_*_debug_set keys: they exist in the test-hooks build only (round 6)
from diffsvc_amd import synth
from diffsvc_amd.engine import DenoiserHandle
import dsvc_oracle as O
arch = sys.argv[1] if len(sys.argv) > 1 else "tiny"
hp = synth.tiny_hparams() if arch == "tiny" else dict(synth.HPARAMS_44K)
sd = synth.acoustic_state(hp, 3)
M, H, C, L = hp["audio_num_mel_bins"], hp["hidden_size"], hp["residual_channels"], hp["residual_layers"]
den = DenoiserHandle(sd, M, H, C, L, hp["dilation_cycle_length"], hp["timesteps"], precision="f16_x3t", prefix="denoise_fn.")
print("handle ok", flush=True)
B, T = 2, 40
g = np.random.Generator(np.random.PCG64(1))
spec = torch.from_numpy(g.standard_normal((B, 1, M, T)).astype(np.float32))
cond = torch.from_numpy((g.standard_normal((B, H, T)) * 0.5).astype(np.float32))
t = torch.from_numpy(g.integers(0, hp["timesteps"], size=(B,)))
for n in list(range(0, L + 1)) + [-1]:
    den.debug_set("stop_after_layers", n)
    out = den.forward(spec.cuda(), t.cuda(), cond.cuda())
    torch.cuda.synchronize()
    print("stop_after_layers %d ok" % n, flush=True)
with torch.no_grad():
    ref = O.diffnet_forward(sd, spec, t, cond, hp["dilation_cycle_length"])
print("forward err vs oracle %.3e" % (out.cpu() - ref).abs().max().item(), flush=True)
