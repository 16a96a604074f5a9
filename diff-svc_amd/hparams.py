"""Access to the process-global ``hparams`` dict.  Inside the reference tree this IS the reference's own
dict (utils/hparams.py:6) so ``set_hparams`` / ``Svc`` keep working unchanged; standalone (benchmarks, the
GPU box) it is a private dict with the same keys."""
_own = {}


def get_hparams():
    try:
        import utils.hparams as ref            # the reference's module, when the reference tree is on sys.path
        if isinstance(getattr(ref, "hparams", None), dict) and hasattr(ref, "set_hparams"):
            return ref.hparams
    except Exception:
        pass
    return _own


def set_hparams(new, clear=True):
    hp = get_hparams()
    if clear:
        hp.clear()
    hp.update(new)
    return hp
