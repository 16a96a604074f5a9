"""Round 4 diagnostic: where does the 6.7e-5 of d skip_projection.weight at the benchmarked 64 x 128 batch sit?  (tests/golden/train_grads_bench*.npz)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import diffsvc_amd
from diffsvc_amd import synth
from diffsvc_amd.train import DiffusionTrainerHip
from util import load_golden
g32, g64 = load_golden("train_grads_bench"), load_golden("train_grads_bench_f64")
case = "bench64x128_l2"
hp = dict(synth.HPARAMS_44K, diff_loss_type="l2")
sd = synth.acoustic_state(hp, 3)
clips = tuple(range(64))
hub, m2p, f0, mels, t = (torch.from_numpy(v).cuda() for v in synth.train_batch_kat(hp, clips, 128, 74, 77))
tr = DiffusionTrainerHip(hp, sd)
loss = tr.forward_backward(hub, m2p, f0, mels, t, seed=77, clip_ids=torch.tensor(list(clips), dtype=torch.int32, device="cuda"))
for k in ("denoise_fn.skip_projection.weight", "denoise_fn.input_projection.weight", "denoise_fn.output_projection.weight"):
    got = tr.view(tr.grads, k).cpu().double()
    part = got[synth.train_grad_slices(tuple(got.shape))].squeeze(-1)
    r64 = torch.from_numpy(g64[case + "/grad/" + k]).squeeze(-1)
    r32 = torch.from_numpy(g32[case + "/grad/" + k]).double().squeeze(-1)
    d = part - r64
    print(k, "shape", tuple(part.shape), "rel L2 vs f64 %.2e (fp32 ref %.2e)" % (d.norm() / r64.norm(), (r32 - r64).norm() / r64.norm()))
    # is the error a scale factor?  least-squares alpha with part ~ alpha * r64
    alpha = (part * r64).sum() / (r64 * r64).sum()
    print("   best scale alpha - 1 = %.3e, residual after rescaling %.2e" % (alpha - 1, (part - alpha * r64).norm() / r64.norm()))
    rows = d.norm(dim=1) / r64.norm(dim=1)
    cols = d.norm(dim=0) / r64.norm(dim=0)
    print("   per-row rel err: median %.2e max %.2e (row %d); per-col: median %.2e max %.2e (col %d)" % (rows.median(), rows.max(), rows.argmax(), cols.median(), cols.max(), cols.argmax()))
