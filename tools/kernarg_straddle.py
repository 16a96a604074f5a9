"""Scalar loads of kernel arguments that straddle a 64-byte line (round 6, third session: one such s_load_dwordx4 in front of the single clip's gate
kernel cost the headline 1.5 %).  Reads device assembly (hipcc --cuda-device-only -S <unit>.hip) and lists, per kernel, every
s_load_dwordxN from the kernarg pointer s[0:1] whose bytes cross a 64-byte boundary.
    python tools/kernarg_straddle.py /tmp/isa/*.s"""
import re, sys
pat = re.compile(r"\ts_load_dword(x(\d+))?\s+s\[?[\d:]+\]?, s\[0:1\], (0x[0-9a-f]+|\d+)")
lab = re.compile(r"^(_Z\w+):")
for path in sys.argv[1:]:
    kern, n_k, n_bad = None, 0, 0
    for line in open(path):
        m = lab.match(line)
        if m:
            kern = m.group(1); n_k += 1
            continue
        m = pat.match(line)
        if m and kern:
            size = 4 * int(m.group(2) or 1)
            off = int(m.group(3), 0)
            if off % 64 + size > 64:
                n_bad += 1
                print("%s: %s: s_load %d B at kernarg + 0x%x straddles a line" % (path.split("/")[-1], kern[:170], size, off))
    print("%s: %d kernels, %d straddling kernarg loads" % (path.split("/")[-1], n_k, n_bad))
