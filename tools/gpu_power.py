"""Board power and shader clock while the per-layer kernels loop (32 clips): is the main loop power-limited?
python tools/gpu_power.py [precision]   -> prints mean power / sclk per ablation variant (sysfs hwmon, rocm-smi fallback)."""
import glob, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import diffsvc_amd
from diffsvc_amd import synth
from diffsvc_amd.engine import DenoiserHandle, SamplerHandle


def sensors():
    out = {}
    for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name, key in (("power1_average", "power_uW"), ("power1_input", "power_uW"), ("freq1_input", "sclk_Hz")):
            p = os.path.join(h, name)
            if os.path.exists(p) and key not in out:
                out[key] = p
    return out


SENS = sensors()


def read_once():
    r = {}
    for k, p in SENS.items():
        try:
            r[k] = float(open(p).read().strip())
        except Exception:
            pass
    if not r:
        try:
            t = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
            for line in t.splitlines():
                if "Power" in line and "(W)" in line:
                    r["power_uW"] = float(line.split(":")[-1]) * 1e6
                if "sclk" in line and "Mhz" in line:
                    r["sclk_Hz"] = float(line.split("(")[-1].split("Mhz")[0]) * 1e6
        except Exception:
            pass
    return r


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.rows, self.stop = [], False

    def run(self):
        while not self.stop:
            self.rows.append(read_once())
            time.sleep(0.05)


prec = sys.argv[1] if len(sys.argv) > 1 else "f16_d64"
print("sensors:", SENS or "rocm-smi", flush=True)
hp = dict(synth.HPARAMS_44K)
sd = synth.acoustic_state(hp, 0)
den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=prec, prefix="denoise_fn.")
smp = SamplerHandle(den, sd)
B = 32
cond = torch.randn(B, 256, 861, device="cuda") * 0.5
smp.sample(cond, 3, seed=1, use_graph=False)
time.sleep(2.0)
print("idle", read_once(), flush=True)
names = {0: "full", 7: "mainloop only", 8: "memory phases only (no MFMA)", 15: "empty"}
for which in ("gate", "out"):
    os.environ["DSVC_PROFILE_KERNEL"] = which
    for dbg in (0, 7, 8, 15):
        os.environ["DSVC_TG_DEBUG"] = str(dbg)
        s = Sampler(); s.start()
        t0 = time.time(); us = []
        while time.time() - t0 < 4.0:
            us.append(smp.profile_gate_kernel(B, 861, 50)[0])
        s.stop = True; s.join()
        rows = s.rows[len(s.rows) // 3:]
        pw = [r["power_uW"] / 1e6 for r in rows if "power_uW" in r]
        ck = [r["sclk_Hz"] / 1e6 for r in rows if "sclk_Hz" in r]
        print("%-4s dbg=%-2d %-30s %7.1f us  power %6.0f W (max %6.0f)  sclk %6.0f MHz (min %6.0f)  n=%d" % (
            which, dbg, names[dbg], sum(us) / len(us), sum(pw) / max(len(pw), 1), max(pw or [0]),
            sum(ck) / max(len(ck), 1), min(ck or [0]), len(rows)), flush=True)
os.environ.pop("DSVC_TG_DEBUG"); os.environ.pop("DSVC_PROFILE_KERNEL")
