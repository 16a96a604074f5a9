"""Print VGPR/AGPR/scratch/LDS/occupancy of every kernel in a .hip file (hipcc -Rpass-analysis)."""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c", src, "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur = {}
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|TotalSGPRs|VGPRs Spill): (.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        if cur: rows.append(cur)
        cur = {"name": v}
    else:
        cur[k] = v
if cur: rows.append(cur)
for c in rows:
    name = subprocess.run(["c++filt", c["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"dsvc::|\(anonymous namespace\)::|ConvGemmArgs.*", "", name)
    if flt and flt not in name: continue
    print("%-70s V%-4s A%-4s S%-3s spill %-3s scratch %-4s occ %s" % (name[:70], c.get("VGPRs"), c.get("AGPRs"), c.get("TotalSGPRs"), c.get("VGPRs Spill"), c.get("ScratchSize [bytes/lane]"), c.get("Occupancy [waves/SIMD]")))
