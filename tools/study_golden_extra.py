"""HIP path vs the REAL reference on the extra headline goldens (tests/golden/e2e_44k_T861_k1000_c4/_c6.npz: the (clip, seed) pairs with
the largest 1000-step errors found by tools/study_headline_spread.py) and on the original two-clip golden.  No oracle run: seconds.
    [DSVC_OUT_W2=1] python tools/study_golden_extra.py [precision ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import diffsvc_amd
from diffsvc_amd import synth
from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
from util import clip_batch, load_golden
import dsvc_oracle as O

precs = sys.argv[1:] or ["f16_d64"]
hp = dict(synth.HPARAMS_44K)
sd = synth.acoustic_state(hp, 0)
for prec in precs:
    den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=prec, prefix="denoise_fn.")
    smp = SamplerHandle(den, sd)
    line = "%s (DSVC_OUT_W2=%s):" % (prec, os.environ.get("DSVC_OUT_W2", "0"))
    for name in ("e2e_44k_T861_k1000", "e2e_44k_T861_k1000_c4", "e2e_44k_T861_k1000_c6"):
        g = load_golden(name)
        clips = [int(c) for c in g["clips"]]
        hub, m2p, f0 = clip_batch(hp, clips, 861, 500)
        cond, _, _ = O.build_cond(sd, hub, m2p, f0.clone(), hp)
        ids = torch.tensor(clips, dtype=torch.int32, device="cuda")
        mel = smp.sample(cond.transpose(1, 2).contiguous().cuda(), 1000, seed=int(g["seed"]), clip_ids=ids, mel2ph=m2p.cuda())
        d = mel.cpu() - torch.from_numpy(g["mel_out"])
        err = d.abs().amax(dim=(1, 2))
        rms = d.pow(2).mean(dim=(1, 2)).sqrt()
        line += "  clips %s: max %s rms %s" % (clips, ["%.2e" % e for e in err.tolist()], ["%.2e" % e for e in rms.tolist()])
    print(line, flush=True)
