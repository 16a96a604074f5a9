"""profiles/train_kernels.json: a roofline one can READ for the training step -- per kernel class the rocprofv3 average
duration, the algorithmic FLOPs and HBM bytes of one launch (formulas below, one multiply-add = 2 FLOP; fp16 [hi | lo] operand planes = 4 B
per channel and row, fp32 rows 4 B), the PMC bytes of the same launch (profiles/train_traffic.json) and what bounds it; and the step's
algorithmic bytes, so that the PMC total becomes a ratio.

    python tools/train_roofline.py <kernel_stats_train.csv> [steps in that profile]

Batch of BASELINE configs[4]: 64 clips x 128 frames, clips 8 rows apart -> R = 8704 rows of which F = 8192 are frames."""
import csv, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import TRAIN_SOURCES, kernel_sources_sha, PEAK_TFLOPS_F16

C, H, M, L, R, F, S = 384, 256, 128, 20, 8704, 8192, 8           # S: frame slices of a weight-gradient launch (partial tiles, reduced in fixed order)
PL = 4                                                             # bytes per channel and row of an fp16 [hi | lo] plane pair
NPAR = 33708675


def classes():
    wconv, wout, wcond = 2 * C * 3 * C, 2 * C * C, 2 * C * H      # elements
    return [
        # (name pattern, what, launches per step, FLOP per launch, algorithmic bytes per launch, bound)
        ("wgrad_fm_kernel", "weight gradients from the frame-major planes (conv taps; output + conditioner projection sharing a launch)", 2 * L,
         (2 * F * 2 * C * 3 * C + 2 * F * 2 * C * (C + H)) / 2,
         ((R * 2 * C * PL + R * C * PL + S * wconv * 4) + (R * C * PL + 2 * R * 2 * C * PL + R * H * PL + S * (wout + wcond) * 4)) / 2,
         "LDS + stage barriers: every product of three MFMAs needs two transposing ds_read_b64_tr_b16 per fragment (LDS >= 67 % busy at the pipe's rate), "
         "three 48 KB stages with one bare barrier each; 840 TFLOP/s MFMA-equivalent = 55 % of the sustained matrix rate"),
        ("TEpiDxT", "transposed dilated conv dx = conv^T(dy) + residual-gradient update + next dO planes", L, 2 * R * 2 * C * 3 * C,
         R * 2 * C * PL * 1.06 + 2 * R * C * 4 + R * C * PL + 2 * C * 3 * C * PL,
         "K = 2304 per output does not fit LDS as a resident tile: streamed in phases (one barrier + vmcnt(0) per 256-channel phase) -- the phases keep the eight waves in "
         "lock step; L2 -> LDS operand delivery at 1.75x the algorithmic bytes (each of the 3 output passes re-reads the dy tile)"),
        ("TEpiGateT", "dilated conv + conditioner projection + gate (forward), sigma / tau rows kept for the backward pass", L, 2 * R * 3 * C * 2 * C,
         R * C * PL * 1.06 + R * 2 * C * 4 + 2 * R * C * 4 + R * C * PL + 2 * C * 3 * C * PL,
         "operand delivery + epilogue stores (fp32 sigma / tau rows): 272 workgroups run as 256 + 16, one per CU, nothing to overlap a workgroup's DMA / MFMA / store phases with"),
        ("TEpiGateBwdT", "dg = W_o^T dO through the gate's derivative -> dy planes", L, 2 * R * 2 * C * C,
         R * 2 * C * PL + 2 * R * C * 4 + R * 2 * C * PL + 2 * C * C * PL, "HBM / L2 delivery: 81 MB per 5.1 GFLOP launch (63 FLOP/B, far below the ridge)"),
        ("TEpiResSkipT", "output 1x1 + residual / skip update + next layer's operand planes (forward)", L, 2 * R * C * 2 * C,
         R * C * PL + 2 * R * C * 4 + 2 * R * C * 4 + R * C * PL + 2 * C * C * PL, "HBM: 82 MB per 5.1 GFLOP launch"),
        ("k_wgrad_nt_reduce", "fixed-order sum of the S partial tiles of every weight gradient (deterministic)", 63, 0, (S + 1) * NPAR * 4 / 63.0,
         "HBM: pure streaming of the partial slabs"),
        ("TEpiCprojT", "all layers' conditioner projections as one stacked product (forward)", 1, 2 * R * H * 2 * C * L, R * H * PL + L * R * 2 * C * 4, "HBM: 20 fp32 accumulator-tiled slabs written"),
        ("k_adamw", "AdamW over the flat 33.7 M-float buffers", 1, 0, NPAR * 28, "HBM: p, g, m, v read; p, m, v written"),
        ("k_tpack_batch", "re-pack of the updated weights into fragment order (100 tensors, one launch)", 1, 0, NPAR * (4 + PL), "HBM"),
    ]


def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    rows = list(csv.DictReader(open(path)))
    tpath = os.path.join(ROOT, "profiles", "train_traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    pmc = {e["kernel"]: e for e in traffic.get("largest", [])} if traffic.get("csrc_sha16") == kernel_sources_sha(TRAIN_SOURCES) else {}
    out, alg_total, t_total = [], 0.0, 0.0
    for pat, what, n, flop, nbytes, bound in classes():
        hit = [r for r in rows if pat in r["Name"]]
        if not hit:
            continue
        calls = sum(int(r["Calls"]) for r in hit)
        us = sum(float(r["TotalDurationNs"]) for r in hit) / calls / 1e3
        e = {"kernel": pat, "what": what, "launches_per_step": round(calls / steps, 1), "avg_us": round(us, 2), "ms_per_step": round(us * calls / steps / 1e3, 3),
             "algorithmic_bytes_per_launch": int(nbytes), "achieved_gbs": round(nbytes / us / 1e3, 1), "hbm_frac": round(nbytes / us / 1e3 / 8000.0, 4), "bound": bound}
        if flop:
            tf = flop / us / 1e6
            e.update(algorithmic_gflop_per_launch=round(flop / 1e9, 3), achieved_tflops=round(tf, 1), mfma_frac=round(tf / PEAK_TFLOPS_F16, 4),
                     mfma_per_product=3, pipe_frac=round(3 * tf / PEAK_TFLOPS_F16, 4))
        p = next((v for k, v in pmc.items() if pat in k), None)
        if p:
            e["pmc_bytes_per_launch"] = int(p["bytes_per_step"] / p["launches_per_step"])
            e["pmc_over_algorithmic"] = round(e["pmc_bytes_per_launch"] / nbytes, 2)
        alg_total += nbytes * calls / steps
        t_total += us * calls / steps
        out.append(e)
    res = {"csrc_sha16": kernel_sources_sha(TRAIN_SOURCES), "source": os.path.basename(path), "steps_in_profile": steps,
           "batch": "64 clips x 128 frames (R = %d rows, F = %d frames), 44.1 kHz architecture" % (R, F),
           "algorithmic_bytes_per_step": int(alg_total), "algorithmic_bytes_covers_ms": round(t_total / 1e3, 3),
           "algorithmic_bytes_note": "sum over the kernel classes below (operand planes, fp32 read-modify-writes, weights, partial gradient tiles, Adam state); the "
                                     "remaining small kernels (pitch-bin sums, column sums, step-embedding GEMMs, loss) add < 3 %",
           "kernels": out}
    if traffic.get("bytes_per_step") and pmc:
        res["pmc_bytes_per_step"] = int(traffic["bytes_per_step"])
        res["pmc_over_algorithmic"] = round(traffic["bytes_per_step"] / alg_total, 2)
    with open(os.path.join(ROOT, "profiles", "train_kernels.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1)[:3000])


if __name__ == "__main__":
    main()
