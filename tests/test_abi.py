"""CPU: the C-ABI shared library builds, loads and exports every symbol include/dsvc.h declares.
No compute call is made here (there is no GPU in the build container)."""
import os
import re

import pytest

import diffsvc_amd
from diffsvc_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    from diffsvc_amd import build
    build.build(verbose=False)              # incremental: a no-op when the .so is newer than its sources
    lib = _lib.lib()
    assert lib.dsvc_abi_version() == 9
    header = open(os.path.join(ROOT, "include", "dsvc.h")).read()
    declared = set(re.findall(r"\b(dsvc_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    # round 4: measurement / test-support entry points are declared apart from the product surface (include/dsvc_debug.h)
    debug = set(re.findall(r"\b(dsvc_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", "dsvc_debug.h")).read()))
    assert debug == {"dsvc_probe_mfma", "dsvc_probe_mfma_detail", "dsvc_denoiser_debug_buffer", "dsvc_denoiser_debug_set",
                     "dsvc_sampler_profile_gate_kernel", "dsvc_sampler_phase_times", "dsvc_trainer_debug_set"} and not (debug & declared)
    assert not re.search(r"debug|probe|profile|phase_times", " ".join(sorted(declared)))
    declared |= debug
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert declared == bound, (declared - bound, bound - declared)
    for name in declared:
        assert hasattr(lib, name), name


def test_the_libraries_export_the_c_abi_and_nothing_else():
    """`nm -D` of the product library listed 57 mangled C++ internals (dsvc_denoiser::eval, dsvc::fail, device stubs, the
    kernel handle objects) beside the entry points -- in a host process that also maps torch's HIP libraries any of them can interpose.  Built
    with -fvisibility=hidden + DSVC_API on the declarations + a linker version script: the dynamic symbol table holds dsvc_* functions only, in
    the product library and in the test-hooks build; and the keys of dsvc_*_debug_set that change which kernel computes a result do not exist in
    the product library at all (their string literals are absent from the binary; the hooks build has them)."""
    import subprocess
    from diffsvc_amd import build
    build.build(verbose=False)
    declared = {name for name, _, _ in _lib.SYMBOLS}
    for so in (build.OUT, build.OUT_HOOKS):
        nm = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
        syms = [line.split() for line in nm.splitlines() if line.strip()]
        foreign = [sy[-1] for sy in syms if not sy[-1].startswith("dsvc_")]
        assert not foreign, (so, foreign[:10])
        assert {sy[-1] for sy in syms} == declared, (so, {sy[-1] for sy in syms} ^ declared)
        assert all(sy[-2] == "T" for sy in syms), [sy for sy in syms if sy[-2] != "T"]
    blob = open(build.OUT, "rb").read()
    hooks_blob = open(build.OUT_HOOKS, "rb").read()
    for key in (b"two_launch_layer", b"stop_after_layers", b"w6_off", b"g6_off", b"fused_nt", b"fused_tail", b"defer_skip", b"wgrad_fm"):
        assert key + b"\0" not in blob, key
        assert key + b"\0" in hooks_blob, key
    assert b"profile_kernel\0" in blob           # the one measurement key the product library keeps (which kernel the bench's timer times)


def test_hooks_build_is_a_second_library_bound_per_handle():
    """`_lib.hooks_build()` binds to libdsvc_hip_hooks.so for its duration and restores the product library afterwards; both carry the whole ABI
    (same version, every symbol); the product library's debug_set refuses a hook key with the message that says where it lives, the hooks
    library gets past the key check (it then fails on the null handle -- no device call is made here)."""
    from diffsvc_amd import build
    build.build(verbose=False)
    prod = _lib.lib()
    with _lib.hooks_build() as hooks:
        assert hooks is not prod and _lib.lib() is hooks and hooks.dsvc_abi_version() == prod.dsvc_abi_version() == _lib.ABI_VERSION
        for name, _, _ in _lib.SYMBOLS:
            assert hasattr(hooks, name), name
    assert _lib.lib() is prod
    assert prod.dsvc_denoiser_debug_set(None, b"two_launch_layer", 1) != 0 and b"null" in prod.dsvc_last_error()


def test_error_convention_without_gpu():
    import ctypes
    lib = _lib.lib()
    h = ctypes.c_void_p(0)
    rc = lib.dsvc_denoiser_create(None, ctypes.byref(h))
    assert rc != 0 and b"null" in lib.dsvc_last_error()
    with pytest.raises(RuntimeError):
        _lib.check(rc)


def test_hot_kernels_do_not_spill():
    """Register budget guard (CPU: read from the compiler's resource report of the last build).  The two per-layer kernels of
    the 128-frame tiling sit at the 256-VGPR limit of two waves per SIMD: an innocent edit (a profiling hook, an index
    computation) tips them into scratch spills and costs 12 % of the batched throughput without failing any parity test."""
    from diffsvc_amd import build
    build.build(verbose=False)
    res = build.kernel_resources("diffnet.hip")
    hot = {k: v for k, v in res.items() if "tgemm_kernelILi4ELi8ELi2ELi8ELi1E" in k and k.endswith("ELi1ELi1ELi0ELi0ELi1ELi0EEEvNS_9TGemmArgsENT4_4ArgsE")}       # (..., FS = 1, SKIP = 0: the instantiation every non-ragged call runs)
    gate = [v for k, v in hot.items() if "TEpiGate" in k]
    out = [v for k, v in hot.items() if "TEpiResSkip" in k]
    assert len(gate) == 1 and len(out) == 1, sorted(hot)
    assert gate[0]["spill"] == 0 and gate[0]["scratch"] == 0, gate
    assert out[0]["spill"] <= 2, out
    small = [v for k, v in res.items() if "tgemm_kernelILi1ELi3ELi3E" in k]
    assert small and all(v["spill"] == 0 for v in small), small
    # the fused layer kernels of the batched path (tlayer.h; ...W6 = 2 / 1 / 0 = f16_w6 / f16_w6n / f16_w2): a few spilled registers at pass
    # boundaries are what they ship with; a second accumulator set prefetched beside f16_w6's code operands made the allocator spill whole
    # accumulator tiles INSIDE the loops (scratch 460 ... 500 bytes per lane, 166 instead of 128 us per layer)
    # the trainer's layer kernels (round 4: 64-frame tiles, split activations; the two data-gradient GEMMs -- and the gate conv beyond dilation 16 -- stream K in phases): no scratch at all
    tr = build.kernel_resources("train.hip")
    tk = {k: v for k, v in tr.items() if "tgemm_kernelILi2ELi8ELi2ELi4ELi2E" in k}
    assert len(tk) == 6 and all(v["spill"] == 0 and v["scratch"] == 0 for v in tk.values()), tk
    # round 5: the weight-gradient kernel on the frame-major planes (wgrad.h) -- both instantiations (with / without the bias column sums) keep their
    # two fragment sets, four accumulator tiles and the dynamically indexed problem descriptor in registers / SGPRs (no scratch), two waves per SIMD
    wk = {k: v for k, v in tr.items() if "wgrad_fm_kernel" in k}
    assert len(wk) == 2 and all(v["spill"] == 0 and v["scratch"] == 0 and v["vgprs"] <= 256 for v in wk.values()), wk
    # the vocoder's wide-stage kernels (round 4: tgemm with split activations; 128 channels as two frame sub-tiles per workgroup)
    vk = {k: v for k, v in build.kernel_resources("vocoder.hip").items() if "tgemm_kernel" in k}
    assert len(vk) == 6 and all(v["spill"] == 0 and v["scratch"] == 0 for v in vk.values()), vk
    for tag, limit in (("ELi0ELi2ELi4EEEv", 128), ("ELi0ELi1ELi4EEEv", 96), ("ELi0ELi0ELi4EEEv", 64)):      # (128-frame tiles: NT = 4)
        k = [v for name, v in res.items() if "tlayer_kernelILi3ELi4ELi2ELi0ELi2ELi0" in name and tag in name]
        assert len(k) == 1 and k[0]["scratch"] <= limit, (tag, k)


def test_fresh_checkout_builds_from_tracked_sources_only(tmp_path):
    """What a fresh clone has is enough: `git archive HEAD` (tracked files only: no prebuilt .so, no object cache) -> __graft_entry__.build()
    compiles every HIP unit for gfx950 and the library exports every symbol of include/dsvc.h.  (The in-tree build is incremental, so
    without this a file missing from git would only be noticed on someone else's machine.)"""
    import re
    import subprocess
    import sys
    if not os.path.isdir(os.path.join(ROOT, ".git")):
        pytest.skip("not a git checkout")
    dst = tmp_path / "clone"
    dst.mkdir()
    ar = subprocess.run(["git", "-C", ROOT, "archive", "HEAD"], capture_output=True)
    assert ar.returncode == 0, ar.stderr[-500:]
    subprocess.run(["tar", "-x", "-C", str(dst)], input=ar.stdout, check=True)
    assert not (dst / "diff-svc_amd" / "libdsvc_hip.so").exists()
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=str(dst), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    so = dst / "diff-svc_amd" / "libdsvc_hip.so"
    assert so.exists()
    nm = subprocess.run(["nm", "-D", "--defined-only", str(so)], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (dsvc_\w+)", nm))
    with open(os.path.join(ROOT, "include", "dsvc.h")) as f:
        declared = set(re.findall(r"^DSVC_API (?:int|void|const char\*)\s+(dsvc_\w+)\(", f.read(), re.M))
    assert declared and declared <= exported, sorted(declared - exported)
