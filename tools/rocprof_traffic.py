"""HBM traffic per launch of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE):
   python tools/rocprof_traffic.py <fetch_dir> <write_dir> <kernel substring> <out.json>
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) streaming reads
(MI355X_MICROARCH.md, HBM section), so the read side is doubled."""
import csv, glob, hashlib, json, os, sys
SAMPLER_SOURCES = ("common.h", "common.hip", "tgemm.h", "tlayer.h", "ttail.h", "diffnet_t.h", "diffnet_kernels.h", "diffnet.hip")   # as bench.py
TRAIN_SOURCES = ("common.h", "common.hip", "tgemm.h", "tepi_util.h", "conv_gemm.h", "wgrad.h", "train.hip")                # as bench.py
def kernel_sources_sha(names=SAMPLER_SOURCES):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for name in names:
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
def per_kernel(d, name):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                a = acc.setdefault(r["Kernel_Name"], [0.0, 0])
                a[0] += float(r["Counter_Value"]); a[1] += 1
    return acc
fd, wd, sub, out = sys.argv[1:5]
if sub == "@step":
    # whole training step: every kernel launched a multiple of <steps> times (the per-step kernels; the trainer's one-off set-up kernels are not),
    # summed and divided by the number of steps.  The read side is doubled like everywhere (the layer kernels, which are most of the bytes, read
    # with 16 B per lane); argv: ... @step out.json "<command>" <steps>
    cmd, nsteps = sys.argv[5], int(sys.argv[6])
    fk, wk = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
    rows = []
    for k in fk:
        if fk[k][1] % nsteps or k not in wk or wk[k][1] != fk[k][1]:
            continue
        rows.append((k, fk[k][1] // nsteps, (2.0 * fk[k][0] + wk[k][0]) * 1024 / nsteps))
    rows.sort(key=lambda r: -r[2])
    res = {"kernel": "every per-step kernel of dsvc_trainer_step (+ clip + AdamW)", "csrc_sha16": kernel_sources_sha(TRAIN_SOURCES), "steps_sampled": nsteps,
           "bytes_per_step": sum(r[2] for r in rows), "kernels": len(rows),
           "largest": [{"kernel": r[0][:90], "launches_per_step": r[1], "bytes_per_step": r[2]} for r in rows[:8]],
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `" + cmd + "`; read side doubled per the gfx950 FETCH_SIZE correction"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res)[:600])
    sys.exit(0)
cmd = sys.argv[5] if len(sys.argv) > 5 else "bench.py --steps 1 --warmup 1 --no-batched --no-cpu-baseline --no-graph"
f, nf = mean_counter(fd, "FETCH_SIZE", sub)
w, nw = mean_counter(wd, "WRITE_SIZE", sub)
res = {"kernel": sub, "csrc_sha16": kernel_sources_sha(), "launches_sampled": [nf, nw], "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
       "bytes_per_launch": (2.0 * f * 1024 + w * 1024) if f is not None and w is not None else None,
       "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `" + cmd + "`; "
                 "read side doubled per the gfx950 FETCH_SIZE correction"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
