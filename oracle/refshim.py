"""Container-only import shim for the upstream reference at /root/reference.

TEST INFRASTRUCTURE -- never imported by the product path (diff-svc_amd/).

The reference is pure Python but imports a number of third-party packages that are
absent from this image (librosa, soundfile, torchaudio, parselmouth, ...).  None of them
is touched by the hot path (sampler / DiffNet / NSF-HiFiGAN generator), so empty module
stubs are enough to import ``network.diff.diffusion``, ``network.diff.net`` and
``modules.nsf_hifigan.models`` unmodified and run them on CPU.  This file is used only by
``oracle/make_golden.py`` (which mints tests/golden/*.npz) and by the oracle pinning tests
when /root/reference is present; /root/reference does not exist on the GPU box.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("DSVC_REFERENCE_ROOT", "/root/reference")

_STUBS = [
    "librosa", "librosa.util", "librosa.filters", "librosa.core", "soundfile", "torchaudio",
    "parselmouth", "pyloudnorm", "resampy", "torchcrepe", "webrtcvad",
    "skimage", "skimage.transform", "pycwt", "pycwt.wavelet", "h5py",
]


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "network", "diff"))


def install():
    """Put /root/reference on sys.path behind stubs for the missing third-party packages."""
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    for name in _STUBS:
        if name in sys.modules:
            continue
        try:
            __import__(name)
            continue
        except Exception:
            pass
        mod = types.ModuleType(name)
        mod.__dict__["__path__"] = []
        sys.modules[name] = mod
        if "." in name:
            parent, child = name.rsplit(".", 1)
            setattr(sys.modules[parent], child, mod)
    # names the reference does ``from X import Y`` on
    sys.modules["librosa.util"].normalize = lambda x, *a, **k: x
    sys.modules["librosa.filters"].mel = _unavailable("librosa.filters.mel")
    sys.modules["skimage.transform"].resize = _unavailable("skimage.transform.resize")
    import scipy.signal
    import scipy.signal.windows
    if not hasattr(scipy.signal, "kaiser"):
        scipy.signal.kaiser = scipy.signal.windows.kaiser
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)


def _unavailable(what):
    def f(*a, **k):
        raise RuntimeError("%s is stubbed (package not installed in this image)" % what)
    return f


def set_hparams(hp: dict):
    """Fill the reference's process-global hparams dict (utils/hparams.py:6)."""
    install()
    from utils.hparams import hparams
    hparams.clear()
    hparams.update(hp)
    return hparams
