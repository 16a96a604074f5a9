"""Kernel stats from a rocprofv3 kernel-trace CSV: python tools/rocprof_stats.py <dir> [pmc]"""
import csv, glob, os, re, sys, collections
d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
pmcf = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
def short(n):
    n = re.sub(r"dsvc::|\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*", "", n)
    return n[:90]
if files:
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in agg.values())
    print("%-92s %8s %12s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "%"))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:25]:
        print("%-92s %8d %12.1f %10.2f %7.2f" % (k, len(v), sum(v) / 1e3, sum(v) / len(v) / 1e3, 100.0 * sum(v) / tot))
if files and len(sys.argv) > 2 and sys.argv[2] == "gaps":
    ev = []
    for f in files:
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    ev.sort()
    gaps = collections.defaultdict(list)
    for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
        if "tgemm" in n0 and "tgemm" in n1:
            gaps[n0[-40:] + " -> " + n1[-40:]].append(s1 - e0)
    print("gap after kernel -> next kernel (ns): count mean p50 p90")
    for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:8]:
        v = sorted(v)
        print("  %-84s %6d %8.0f %8d %8d" % (k, len(v), sum(v) / len(v), v[len(v) // 2], v[int(len(v) * 0.9)]))
if pmcf:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in pmcf:
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in sorted(agg.items(), key=lambda kv: -len(next(iter(kv[1].values()))))[:12]:
        print(k)
        for c, v in sorted(cs.items()):
            print("    %-36s n=%-6d mean=%.4g" % (c, len(v), sum(v) / len(v)))
