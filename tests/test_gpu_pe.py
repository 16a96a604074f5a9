"""GPU parity of the pitch extractor: PitchExtractor.forward on the HIP kernels (dsvc_pe_*) against the REAL
reference module's outputs (tests/golden/pe_24k.npz) and against the oracle at other shapes."""
import numpy as np
import pytest
import torch

from diffsvc_amd import synth
import dsvc_oracle as O
from util import load_golden

pytestmark = pytest.mark.gpu

PE_CASES = ((2, 50, (0, 7), False), (1, 300, (0,), False), (2, 20, (3, 0), False), (3, 33, (0, 5, 33), True))   # as oracle/make_golden.py
PRED_TOL = 1e-3           # pitch_pred is log2(Hz) (about 7..8) and a logit: the mel's max-abs bar
F0_REL_TOL = 1e-3         # f0 = 2**pred: 1e-3 in log2 is 7e-4 relative (1.2 cents)


def _extractor(hp, wseed, conv_layers=2, n_mel=80):
    from diffsvc_amd.pe import PitchExtractorHip
    sd = synth.pe_state(hp, wseed, n_mel=n_mel, conv_layers=conv_layers)
    pe = PitchExtractorHip(n_mel_bins=n_mel, conv_layers=conv_layers, hparams=hp).cuda()
    pe.load_state_dict({"" + k: v for k, v in sd.items()}, strict=True)
    return pe.eval(), sd


def _check(out, pred_ref, f0_ref, uv_ref=None, what=""):
    pred, f0 = out["pitch_pred"].cpu(), out["f0_denorm_pred"].cpu()
    assert pred.shape == pred_ref.shape and f0.shape == f0_ref.shape
    e_pred = (pred - pred_ref).abs().max().item()
    # the voiced/unvoiced decision is a sign test on a logit: frames whose reference logit is within the tolerance of 0 may flip
    firm = torch.ones_like(f0_ref, dtype=torch.bool) if uv_ref is None else (uv_ref.abs() > PRED_TOL)
    assert torch.equal((f0 == 0) & firm, (f0_ref == 0) & firm)
    both = (f0 != 0) & (f0_ref != 0)
    e_f0 = ((f0 - f0_ref).abs() / f0_ref.clamp(min=1))[both].max().item() if both.any() else 0.0
    print("pe %s: pitch_pred max-abs err %.2e, f0 max rel err %.2e" % (what, e_pred, e_f0))
    assert e_pred < PRED_TOL, e_pred
    assert e_f0 < F0_REL_TOL, e_f0


def test_pe_vs_reference_golden():
    g = load_golden("pe_24k")
    hp = dict(synth.HPARAMS_24K)
    for i, (B, T, tails, use_uv) in enumerate(PE_CASES):
        hp_i = dict(hp, use_uv=use_uv)
        pe, _ = _extractor(hp_i, int(g["wseed"]))
        mel = torch.from_numpy(synth.mel_like(40 + i, B, T, 80, tails)).cuda()
        pred_ref = torch.from_numpy(g["pitch_pred%d" % i])
        _check(pe(mel), pred_ref, torch.from_numpy(g["f0_%d" % i]), pred_ref[..., 1] if use_uv else None, str((B, T, tails, use_uv)))


@pytest.mark.parametrize("B,T,tails", [(1, 1, (0,)), (1, 31, (4,)), (4, 128, (0, 1, 64, 127)), (1, 862, (0,)), (2, 4200, (100, 0))])
def test_pe_shapes_vs_oracle(B, T, tails):
    """Edge shapes: one frame, under one 32-row slot, a batch with every padding pattern, the headline clip length, and a clip longer than
    the 4096-row position table the reference starts with (it regrows the table: common_layers.py:127-134)."""
    hp = dict(synth.HPARAMS_24K)
    pe, sd = _extractor(hp, 9)
    mel = torch.from_numpy(synth.mel_like(7, B, T, 80, tails))
    with torch.no_grad():
        pred_ref, f0_ref = O.pitch_extractor(sd, mel, hp)
    out = pe(mel.cuda())
    _check(out, pred_ref, f0_ref, None, str((B, T, tails)))
    again = pe(mel.cuda())
    assert torch.equal(again["f0_denorm_pred"], out["f0_denorm_pred"])


def test_pe_variants_vs_oracle():
    """conv_layers=0 (no mel_encoder, pe.py:128), a wider predictor (predictor_hidden > 0), 'standard' pitch normalisation, and a
    shorter call after a longer one on the same handle (no stale workspace rows)."""
    hp = dict(synth.HPARAMS_24K, predictor_hidden=384, pitch_norm="standard", f0_mean=180.0, f0_std=25.0)
    for conv_layers in (0, 1):
        pe, sd = _extractor(hp, 3, conv_layers=conv_layers)
        for T in (140, 60):
            mel = torch.from_numpy(synth.mel_like(T, 2, T, 80, (0, 9)))
            with torch.no_grad():
                pred_ref, f0_ref = O.pitch_extractor(sd, mel, hp, conv_layers=conv_layers)
            out = pe(mel.cuda())
            pred, f0 = out["pitch_pred"].cpu(), out["f0_denorm_pred"].cpu()
            assert (pred - pred_ref).abs().max().item() < PRED_TOL
            assert torch.equal(f0 == 0, f0_ref == 0)
            assert (f0 - f0_ref).abs().max().item() < 25.0 * PRED_TOL * 2


def test_pe_module_contract():
    """The nn.Module surface the reference touches (infer_tool.py:134-136): strict load rejects a missing / unexpected key and a wrong
    shape, hparams are read at call time (use_uv flips without reloading), a CPU tensor is refused (no CPU path)."""
    from diffsvc_amd.pe import PitchExtractorHip
    hp = dict(synth.HPARAMS_24K)
    sd = synth.pe_state(hp, 1)
    pe = PitchExtractorHip(hparams=hp).cuda()
    bad = dict(sd); bad.pop("mel_encoder.conv.1.norm.bias")
    with pytest.raises(RuntimeError, match="missing"):
        pe.load_state_dict(bad, strict=True)
    bad = dict(sd); bad["extra.weight"] = torch.zeros(1)
    with pytest.raises(RuntimeError, match="unexpected"):
        pe.load_state_dict(bad, strict=True)
    bad = dict(sd); bad["pitch_predictor.linear.weight"] = torch.zeros(3, 256)
    with pytest.raises(RuntimeError, match="size mismatch"):
        pe.load_state_dict(bad, strict=True)
    with pytest.raises(RuntimeError, match="no weights"):
        pe(torch.zeros(1, 8, 80).cuda())
    pe.load_state_dict(sd, strict=True).eval()
    assert list(pe.state_dict().keys()) == list(sd.keys())
    mel = torch.from_numpy(synth.mel_like(2, 1, 64, 80, (0,)))
    with pytest.raises(RuntimeError, match="HIP device"):
        pe(mel)
    a = pe(mel.cuda())["f0_denorm_pred"]
    hp["use_uv"] = True
    b = pe(mel.cuda())
    uv = b["pitch_pred"][..., 1] > 0
    assert uv.any() and not uv.all()
    assert torch.equal(b["f0_denorm_pred"], torch.where(uv, torch.zeros_like(a), a))
