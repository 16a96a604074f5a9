"""HBM traffic per launch of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE):
   python tools/rocprof_traffic.py <fetch_dir> <write_dir> <kernel substring> <out.json>
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) streaming reads
(MI355X_MICROARCH.md, HBM section), so the read side is doubled."""
import csv, glob, hashlib, json, os, sys
SAMPLER_SOURCES = ("common.h", "common.hip", "tgemm.h", "tlayer.h", "diffnet_t.h", "diffnet_kernels.h", "diffnet.hip")   # as bench.py
def kernel_sources_sha():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for name in SAMPLER_SOURCES:
        with open(os.path.join(root, "diff-svc_amd", "csrc", name), "rb") as fh:
            h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]
def mean_counter(d, name, sub):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and sub in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)
fd, wd, sub, out = sys.argv[1:5]
cmd = sys.argv[5] if len(sys.argv) > 5 else "bench.py --steps 1 --warmup 1 --no-batched --no-cpu-baseline --no-graph"
f, nf = mean_counter(fd, "FETCH_SIZE", sub)
w, nw = mean_counter(wd, "WRITE_SIZE", sub)
res = {"kernel": sub, "csrc_sha16": kernel_sources_sha(), "launches_sampled": [nf, nw], "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
       "bytes_per_launch": (2.0 * f * 1024 + w * 1024) if f is not None and w is not None else None,
       "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `" + cmd + "`; "
                 "read side doubled per the gfx950 FETCH_SIZE correction"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
