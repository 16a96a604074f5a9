"""Run INSIDE a subprocess by tests/test_reference_seams.py (the reference tree on sys.path and its process-global hparams dict must
not leak into the other tests): drive the drop-ins through the REAL reference's own seam code and print one JSON line.
TEST INFRASTRUCTURE (container only: /root/reference does not exist on the GPU box)."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch  # noqa: E402
import refshim  # noqa: E402

out = {}
refshim.install()
import diffsvc_amd  # noqa: E402,F401
from diffsvc_amd import synth  # noqa: E402

# ---- 1. vocoder plugin seam: network/vocoders/base_vocoder.py:5-19 resolves the dotted class path with importlib and the class
#         registers itself through the reference's own register_vocoder (infer_tool.py:244-247 looks it up by bare class name) ----
from network.vocoders import base_vocoder as BV  # noqa: E402
cls = BV.get_vocoder_cls({"vocoder": "diffsvc_amd.vocoder.NsfHifiGANHip"})
out["vocoder_cls"] = cls.__name__
out["vocoder_module"] = cls.__module__
out["registered_bare"] = BV.VOCODERS.get("NsfHifiGANHip") is cls
out["registered_lower"] = BV.VOCODERS.get("nsfhifiganhip") is cls
out["is_base_vocoder"] = issubclass(cls, BV.BaseVocoder)
out["by_short_name"] = BV.get_vocoder_cls({"vocoder": "NsfHifiGANHip"}) is cls
out["has_contract"] = all(callable(getattr(cls, n, None)) for n in ("spec2wav", "wav2spec"))

# ---- 2. checkpoint seam: the reference's own utils.load_ckpt (utils/__init__.py:178-209: torch.load -> ['state_dict'] -> strip
#         'model.' -> load_state_dict(strict=True)) into the drop-in built the way Svc.__init__ builds the model
#         (infer_tool.py:122-129), with the process-global hparams dict of the reference ----
hp = synth.tiny_hparams()
refshim.set_hparams(hp)
import utils  # noqa: E402  (the reference's package)
from diffsvc_amd.denoiser import DiffNetHip  # noqa: E402
from diffsvc_amd.sampler import GaussianDiffusionHip  # noqa: E402
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "model_ckpt_steps_100.ckpt")
    sd = synth.save_acoustic_ckpt(path, hp, seed=3)
    model = GaussianDiffusionHip(phone_encoder=None, out_dims=hp["audio_num_mel_bins"], denoise_fn=DiffNetHip(hp["audio_num_mel_bins"]),
                                 timesteps=hp["timesteps"], K_step=hp["K_step"], loss_type=hp["diff_loss_type"],
                                 spec_min=hp["spec_min"], spec_max=hp["spec_max"])
    utils.load_ckpt(model, path, "model", force=True, strict=True)
    own = model.state_dict()
    out["ckpt_keys_equal"] = sorted(own.keys()) == sorted(sd.keys())
    out["ckpt_values_equal"] = all(torch.equal(own[k].cpu(), sd[k]) for k in sd)
    # the same file into the REAL reference model: both accept it strictly, with identical key sets
    from network.diff.diffusion import GaussianDiffusion  # noqa: E402
    from network.diff.net import DiffNet  # noqa: E402
    ref = GaussianDiffusion(None, hp["audio_num_mel_bins"], DiffNet(hp["audio_num_mel_bins"]), timesteps=hp["timesteps"], K_step=hp["K_step"],
                            loss_type=hp["diff_loss_type"], spec_min=hp["spec_min"], spec_max=hp["spec_max"])
    utils.load_ckpt(ref, path, "model", force=True, strict=True)
    out["same_keys_as_reference_model"] = sorted(ref.state_dict().keys()) == sorted(own.keys())
    # a directory instead of a file: load_ckpt picks the highest model_ckpt_steps_*.ckpt (utils/__init__.py:183-187)
    model2 = GaussianDiffusionHip(None, hp["audio_num_mel_bins"], DiffNetHip(hp["audio_num_mel_bins"]), timesteps=hp["timesteps"],
                                  K_step=hp["K_step"], loss_type=hp["diff_loss_type"], spec_min=hp["spec_min"], spec_max=hp["spec_max"])
    utils.load_ckpt(model2, td, "model", force=True, strict=True)
    out["dir_load_ok"] = all(torch.equal(model2.state_dict()[k].cpu(), sd[k]) for k in sd)
    # a checkpoint that lacks a key must fail the strict load exactly like the reference model does
    bad = {"state_dict": {"model." + k: v for k, v in sd.items() if k != "denoise_fn.skip_projection.bias"}}
    torch.save(bad, os.path.join(td, "bad.ckpt"))
    try:
        utils.load_ckpt(model2, os.path.join(td, "bad.ckpt"), "model", force=True, strict=True)
        out["strict_rejects_missing"] = False
    except RuntimeError:
        out["strict_rejects_missing"] = True
# the drop-ins read the SAME process-global dict as the reference (utils/hparams.py:6)
from utils.hparams import hparams as ref_hparams  # noqa: E402
from diffsvc_amd.hparams import get_hparams  # noqa: E402
out["shares_global_hparams"] = get_hparams() is ref_hparams
print("RESULT " + json.dumps(out))
