"""Phase timeline of the tgemm kernels from in-kernel s_memrealtime stamps (DSVC_TG_STAMPS=<prefix>):
python tools/stamps_report.py <prefix> [first_launch]
Stamps per wave: 0 entry, 1 loads issued, 2 vmcnt(0), 3 barrier passed, 4..8 after group pairs, 9 main loop issued,
10 epilogue issued, 11 stores drained (first pass only is meaningful for multi-pass waves: later passes overwrite)."""
import sys
import numpy as np
prefix = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
meta = [l.rstrip("\n").split("|") for l in open(prefix + ".meta")]
raw = np.fromfile(prefix + ".bin", dtype=np.uint64).reshape(len(meta), -1, 16)
TICK = 0.01  # us per s_memrealtime tick (100 MHz)
prev_end = None
for i, (name, gx, gy, waves) in enumerate(meta):
    if i < first:
        continue
    n = int(gx) * int(gy) * int(waves)
    s = raw[i, :n].astype(np.int64)
    ok = s[:, 0] > 0
    s = s[ok]
    if not len(s):
        continue
    t0 = s[:, 0].min()
    end = s[:, 11].max()
    short = name.split("Epi = ")[-1].split(";")[0].split("]")[0] if "Epi = " in name else name[:40]
    cfg = name.split("[with ")[-1][:60] if "[with " in name else ""
    rel = (s - t0) * TICK
    gap = (t0 - prev_end) * TICK if prev_end is not None else float("nan")
    prev_end = end
    def med(c): return float(np.median(rel[:, c]))
    def mx(c): return float(rel[:, c].max())
    grp = " ".join("%.2f" % med(c) for c in range(4, 9) if (s[:, c] > 0).all())
    print("%3d %-14s grid %sx%s w%s gap %5.2f | start med %.2f max %.2f | dma-issued %.2f step-read %.2f ring-issued %.2f | issued %.2f | landed med %.2f max %.2f | barrier %.2f | groups %s | loop %.2f | epi %.2f | drained med %.2f max %.2f"
          % (i, short[:14], gx, gy, waves, gap, med(0), mx(0), med(12), med(13), med(14), med(1), med(2), mx(2), med(3), grp, med(9), med(10), med(11), mx(11)))
