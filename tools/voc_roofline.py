"""profiles/voc_kernels.json: per-kernel roofline entries of the NSF-HiFiGAN generator for one 10 s clip (the k_pair_mfma
family is 40 % of the generator and had no entry of its own).  python tools/voc_roofline.py <kernel_stats_voc.csv> <clips in that profile>

44.1 kHz config: stage i has C_i = 512 >> (i + 1) channels at L_i = T * prod(rates[:i + 1]) samples (T = 861; 256 / 128 / 64 / 32 / 16 channels at
8x / 64x / 128x / 256x / 512x the frame rate); a stage's MRF = 3 kernel sizes (3, 7, 11) x 3 conv pairs (dilated conv + plain conv, C -> C).
One multiply-add = 2 FLOP; every product is three fp16 MFMAs (hi/lo weights x hi|lo activations: fp32-class)."""
import csv, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T, RATES, CH0, KS = 861, (8, 8, 2, 2, 2), 512, (3, 7, 11)
VOC_SOURCES = ("common.h", "common.hip", "tgemm.h", "conv_gemm.h", "cg_util.h", "vocoder.hip")


def sha():
    h = hashlib.sha256()
    for n in VOC_SOURCES:
        with open(os.path.join(ROOT, "diff-svc_amd", "csrc", n), "rb") as f:
            h.update(n.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def stage(i):
    L = T
    for r in RATES[:i + 1]:
        L *= r
    return CH0 >> (i + 1), L


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    clips = int(sys.argv[2]) if len(sys.argv) > 2 else 21
    kavg = sum(KS) / 3.0
    classes = []
    for i, pat in ((2, "k_pair_mfmaILi64E"), (3, "k_pair_mfmaILi32E"), (4, "k_pair_mfmaILi16E")):
        C, L = stage(i)
        classes.append((pat, "k_pair_mfma<%d>: a ResBlock conv pair (lrelu -> dilated conv -> lrelu -> conv -> + x) fused through LDS, %d channels x %d samples" % (C, C, L),
                        2 * 2 * C * C * kavg * L, 2 * C * L * 4,
                        "neither roof: the pair's intermediate never leaves LDS, so a launch moves 2 x C x L x 4 B (<= 0.7 TB/s) -- the time is the fp32 -> [hi | lo] fp16 "
                        "split of every operand in registers (VALU), the k-tap operand gather from LDS (k = 11: 11 shifted reads per output) and three MFMAs per product "
                        "on %dx%d-wide tiles that leave the matrix pipe %s" % (C, C, "idle most of the time (a 16-channel tile fills 1/4 of an MFMA's K)" if C == 16 else "waiting for the gathers")))
    for i, tag in ((0, "1, 8, 2, 4, 2, TEpiVoc"), (1, "2, 8, 2, 4, 2, TEpiVoc")):
        C, L = stage(i)
        for half, name in (("Mid", "first (dilated)"), ("Out", "second")):
            classes.append((tag + half, "tgemm_kernel<TEpiVoc%s>: the %s conv of a pair at the %d-channel stage (%d samples) on the DiffNet engine" % (half, name, C, L),
                            2 * C * C * kavg * L, 2 * C * L * 4 * (2 if half == "Mid" else 1.5),
                            "MFMA issue + weight stream: K = C x k <= 2816 per output, 3 MFMAs per product; launch-bound between the 9 pairs of a stage"))
    out = []
    for pat, what, flop, nbytes, bound in classes:
        hit = [r for r in rows if pat in r["Name"]]
        if not hit:
            continue
        calls = sum(int(r["Calls"]) for r in hit)
        us = sum(float(r["TotalDurationNs"]) for r in hit) / calls / 1e3
        tf = flop / us / 1e6
        out.append({"kernel": pat, "what": what, "launches_per_clip": round(calls / clips, 1), "avg_us": round(us, 1), "ms_per_clip": round(us * calls / clips / 1e3, 3),
                    "algorithmic_gflop_per_launch": round(flop / 1e9, 2), "achieved_tflops": round(tf, 1), "mfma_frac": round(tf / 2500.0, 4), "mfma_per_product": 3,
                    "pipe_frac": round(3 * tf / 2500.0, 4), "algorithmic_bytes_per_launch": int(nbytes), "hbm_frac": round(nbytes / us / 1e3 / 8000.0, 4), "bound": bound})
    res = {"csrc_sha16": sha(), "source": os.path.basename(sys.argv[1]), "clips_in_profile": clips, "kernels": out}
    json.dump(res, open(os.path.join(ROOT, "profiles", "voc_kernels.json"), "w"), indent=1)
    for e in out:
        print("%-28s %5.1f us x %4.1f  %6.1f TFLOP/s (pipe %4.1f %%)  hbm %4.1f %%" % (e["kernel"], e["avg_us"], e["launches_per_clip"], e["achieved_tflops"], 100 * e["pipe_frac"], 100 * e["hbm_frac"]))


if __name__ == "__main__":
    main()
