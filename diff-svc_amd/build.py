"""Build libdsvc_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

``python -m diffsvc_amd.build`` or ``diffsvc_amd.build.build()``.  hipcc cross-compiles gfx950 without a
GPU; the resulting .so is git-ignored but travels with the tree to the GPU box.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libdsvc_hip.so")
OUT_PROF = os.path.join(HERE, "libdsvc_hip_prof.so")
SOURCES = ["common.hip", "diffnet.hip", "vocoder.hip", "melspec.hip", "train.hip", "hubert.hip", "pe.hip", "cond.hip", "probe.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-fvisibility=hidden", "-fvisibility-inlines-hidden",      # the entry points (DSVC_API in include/dsvc.h) are the ONLY dynamic symbols
         "-Rpass-analysis=kernel-resource-usage"]       # the per-kernel register / scratch report is kept next to the object
OUT_HOOKS = os.path.join(HERE, "libdsvc_hip_hooks.so")
HOOK_SOURCES = ["diffnet.hip", "train.hip"]             # the two units that hold dsvc_*_debug_set: compiled a second time with -DDSVC_TEST_HOOKS


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP extension cannot be built (no CPU fallback exists)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, profiling=False):
    """Builds libdsvc_hip.so (the product) and libdsvc_hip_hooks.so (the TEST-HOOKS build: the same objects except csrc/diffnet.hip and
    csrc/train.hip, compiled with -DDSVC_TEST_HOOKS, which lets dsvc_*_debug_set accept the keys that change which kernel computes a result --
    per-layer taps and A/B partners for the parity tests; the product library refuses them).
    profiling=True compiles a SEPARATE library, libdsvc_hip_prof.so, with -DDSVC_PROFILING: the ablation / A-B knobs of the kernels
    (environment variables DSVC_TG_DEBUG, DSVC_TG_STAMPS, DSVC_PROFILE_KERNEL, ... -- several of them give WRONG results by design) exist
    only there.  The product library reads no environment variable; tools load the other one with _lib.use_profiling_build()."""
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build_prof" if profiling else "build")
    hookdir = os.path.join(HERE, "build_hooks")
    out = OUT_PROF if profiling else OUT
    flags = FLAGS + (["-DDSVC_PROFILING"] if profiling else [])
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "dsvc.h"))
    headers.append(os.path.join(os.path.dirname(HERE), "include", "dsvc_debug.h"))
    missing = [s for s in SOURCES if not os.path.exists(os.path.join(CSRC, s))]
    if missing:
        raise RuntimeError("HIP sources missing from %s: %s" % (CSRC, ", ".join(missing)))
    srcs = list(SOURCES)
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in srcs]
    jobs = [(s, o, flags) for s, o in zip(srcs, objs)]
    hook_objs = []
    if not profiling:
        os.makedirs(hookdir, exist_ok=True)
        hook_objs = [os.path.join(hookdir, s.replace(".hip", ".o")) for s in HOOK_SOURCES]
        jobs += [(s, o, flags + ["-DDSVC_TEST_HOOKS"]) for s, o in zip(HOOK_SOURCES, hook_objs)]

    def compile_one(job):
        src, obj, fl = job
        if not force and not _stale(obj, [os.path.join(CSRC, src)] + headers):
            return None
        cmd = [hipcc] + fl + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr[-8000:]))
        with open(obj.replace(".o", ".resources.txt"), "w") as f:
            f.write(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
        built = [b for b in ex.map(compile_one, jobs) if b]

    def link(target, objects):
        if not (built or force or _stale(target, objects + [os.path.join(CSRC, "exports.map")])):
            if verbose:
                print("up to date: %s" % target)
            return
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), "-o", target] + objects
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-8000:])
        if verbose:
            print("built %s (%s)" % (target, ", ".join(os.path.relpath(b, HERE) for b in built) if built else "relink"))

    link(out, objs)
    if hook_objs:
        swap = {os.path.basename(o): o for o in hook_objs}
        link(OUT_HOOKS, [swap.get(os.path.basename(o), o) for o in objs])
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv, profiling="--profiling" in sys.argv)


def kernel_resources(source="diffnet.hip"):
    """{mangled kernel name: {"vgprs", "agprs", "spill", "scratch", "occupancy"}} from the last build of ``source``."""
    import re
    path = os.path.join(HERE, "build", source.replace(".hip", ".resources.txt"))
    out, cur = {}, None
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "VGPRs Spill": "spill", "ScratchSize [bytes/lane]": "scratch",
            "Occupancy [waves/SIMD]": "occupancy"}
    with open(path) as f:
        for line in f:
            m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\S+)", line)
            if not m:
                continue
            if m.group(1) == "Function Name":
                cur = out.setdefault(m.group(2), {})
            elif cur is not None:
                cur[keys[m.group(1)]] = int(m.group(2))
    return out
