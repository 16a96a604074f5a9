"""Shared helpers of the parity tests (oracle side + HIP side)."""
import os

import numpy as np
import torch

import diffsvc_amd  # noqa: F401
from diffsvc_amd import synth
import dsvc_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


def hp_for(name):
    if "tiny" in name:
        return synth.tiny_hparams()
    if "24k" in name:
        return dict(synth.HPARAMS_24K)
    return dict(synth.HPARAMS_44K)


def golden_state(g, hp):
    """The synthetic checkpoint a golden was minted on: plain random-init, or the conditioned variant (PLMS probes)."""
    c = g.get("conditioned")
    if c is not None and np.size(c):
        return synth.acoustic_state_conditioned(hp, int(g["wseed"]), float(c[0]), float(c[1]))
    return synth.acoustic_state(hp, int(g["wseed"]))


def clip_batch(hp, clips, T, n_units):
    hub, m2p, f0 = [], [], []
    for c in clips:
        h, m, f, _ = synth.clip_inputs(int(c), T=T, n_units=n_units, H=hp["hidden_size"])
        hub.append(h); m2p.append(m); f0.append(f)
    return torch.from_numpy(np.stack(hub)), torch.from_numpy(np.stack(m2p)), torch.from_numpy(np.stack(f0))


def oracle_sample(hp, sd, clips, T, n_units, speedup, seed, K_step, t_stop=0):
    """The oracle's restatement of GaussianDiffusion.forward(infer=True) on synthetic clips."""
    hub, m2p, f0 = clip_batch(hp, clips, T, n_units)
    cond, f0_denorm, pitch = O.build_cond(sd, hub, m2p, f0.clone(), hp)
    cond_t = cond.transpose(1, 2).contiguous()
    M = hp["audio_num_mel_bins"]
    x = O.ddpm_noise_ref_layout(seed, clips, 0, T, M, O.PURPOSE_X_INIT)
    cyc = hp["dilation_cycle_length"]
    if speedup > 1:
        x = O.sample_plms(sd, cond_t, x, speedup, cyc, t_start=K_step)
    else:
        x = O.sample_ddpm(sd, cond_t, x, lambda i: O.ddpm_noise_ref_layout(seed, clips, i, T, M), cyc,
                          t_start=K_step, t_end=t_stop)
    mel = O.finish_mel(sd, x, m2p)
    return dict(mel_out=mel, x=x, cond=cond, cond_t=cond_t, f0_denorm=f0_denorm, pitch=pitch, mel2ph=m2p)
