import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


import torch

# the oracle's small CPU convolutions collapse when oneDNN fans them out over a GPU host's 256 hardware threads
# (measured: 50 s per DiffNet evaluation at T=861 instead of 0.1 s); a modest pool is fastest everywhere
torch.set_num_threads(min(16, os.cpu_count() or 1))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def hooks():
    """Handles created inside a test that takes this fixture live in libdsvc_hip_hooks.so -- the product library plus the dsvc_*_debug_set keys
    that change which kernel computes a result (per-layer taps, A/B partners; csrc/diffnet.hip and csrc/train.hip compiled with -DDSVC_TEST_HOOKS,
    every other object shared with the product).  The product library refuses those keys (tests/test_abi.py), and every test WITHOUT this
    fixture -- the parity tests proper -- runs on libdsvc_hip.so."""
    from diffsvc_amd import _lib
    with _lib.hooks_build() as lib:
        yield lib


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_collection_modifyitems(config, items):
    """A GPU test that hangs (a kernel that never returns) must fail the run, not sit until the visit's time limit takes the box with it: every
    ``gpu`` test gets a 300 s limit (the longest takes 19 s, profiles/r5H_pytest_gpu.txt) through pytest-timeout's THREAD method, which ends the
    process even while the main thread is blocked inside a HIP call.  Without the plugin nothing changes."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for it in items:
        if it.get_closest_marker("gpu") and not it.get_closest_marker("timeout"):
            it.add_marker(pytest.mark.timeout(300, method="thread"))
