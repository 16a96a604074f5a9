"""CPU: pin the oracle (oracle/dsvc_oracle.py) against vectors produced by the REAL reference
(tests/golden/*.npz, minted by oracle/make_golden.py from /root/reference)."""
import numpy as np
import pytest
import torch

from diffsvc_amd import synth
import dsvc_oracle as O
from util import hp_for, load_golden, oracle_sample


@pytest.mark.parametrize("name", ["diffnet_tiny", "diffnet_44k"])
def test_diffnet_forward_matches_reference(name):
    g = load_golden(name)
    hp = hp_for(name)
    sd = synth.acoustic_state(hp, int(g["wseed"]))
    with torch.no_grad():
        out = O.diffnet_forward(sd, torch.from_numpy(g["spec"]), torch.from_numpy(g["t"]), torch.from_numpy(g["cond"]),
                                hp["dilation_cycle_length"])
    err = (out - torch.from_numpy(g["out"])).abs().max().item()
    assert err < 2e-5, err          # fp32 summation-order noise only


@pytest.mark.parametrize("name", ["ddpm_tiny", "plms_tiny_s10", "plms_tiny_s5", "ddpm_44k_k20", "plms_44k_k100_s20"])
def test_sampler_matches_reference(name):
    g = load_golden(name)
    hp = dict(hp_for(name), K_step=int(g["K_step"]))
    sd = synth.acoustic_state(hp, int(g["wseed"]))
    r = oracle_sample(hp, sd, [int(c) for c in g["clips"]], int(g["T"]), int(g["n_units"]), int(g["speedup"]),
                      int(g["seed"]), int(g["K_step"]))
    assert np.array_equal(r["pitch"].numpy()[..., None], g["pitch"])                      # index work: bit-exact
    assert np.array_equal(r["f0_denorm"].numpy(), g["f0_denorm"])
    assert np.abs(r["cond"].numpy() - g["decoder_inp"]).max() == 0.0
    err = np.abs(r["mel_out"].numpy() - g["mel_out"]).max()
    assert err < 5e-4, err


def test_schedule_tables_match_checkpoint_buffers():
    hp = dict(synth.HPARAMS_44K)
    bufs = synth.schedule_buffers(hp)
    tabs = O.schedule_tables(O.linear_betas(hp["timesteps"], hp["max_beta"]))
    for k in O.SCHEDULE_KEYS:
        assert torch.equal(bufs[k], tabs[k]), k
    cos = O.schedule_tables(O.cosine_betas(1000))
    assert cos["betas"].shape == (1000,) and float(cos["betas"].max()) <= 0.9990001


def test_get_align_kat():
    # integer recurrence of infer_tool.py:231-242; T=861, N_h=500 is the 10 s clip of the benchmark
    m = O.get_align(861, 500)
    assert m.min() == 1 and m.max() == 500 and (np.diff(m) >= 0).all()
    assert np.array_equal(m, synth.align_units(861, 500))
    assert np.array_equal(m[:8], [1, 1, 2, 2, 3, 3, 4, 4]) or m[0] == 1


def test_philox_known_answer():
    # Random123 known-answer test for Philox4x32-10: counter = key = 0  /  all ones
    r = O.philox4x32(np.uint64(0), np.uint64(0), np.uint64(0), np.uint64(0), 0)
    assert [int(x) for x in r] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    r = O.philox4x32(np.uint64(0xFFFFFFFF), np.uint64(0xFFFFFFFF), np.uint64(0xFFFFFFFF), np.uint64(0xFFFFFFFF),
                     0xFFFFFFFFFFFFFFFF)
    assert [int(x) for x in r] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    z = O.frame_major_noise(1, 0, 3, 64, 16)
    assert abs(float(z.mean())) < 0.15 and abs(float(z.std()) - 1.0) < 0.1
