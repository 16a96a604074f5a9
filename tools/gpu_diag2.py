"""GPU diagnostics of the tgemm path (prints, never asserts): per-stage errors against the oracle's taps with the
layer loop cut short (DSVC_DEBUG_STOP_AFTER_LAYERS), then timings.  Usage on the box:
    python tools/gpu_diag2.py > gpurun_out/diag2.txt"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import diffsvc_amd
from diffsvc_amd import synth
from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
import dsvc_oracle as O

print("torch", torch.__version__, torch.cuda.get_device_name(0), flush=True)

def stage(name, fn):
    t0 = time.time()
    try:
        fn(); print("[ok] %s (%.2fs)" % (name, time.time() - t0), flush=True)
    except Exception:
        print("[FAIL] %s" % name); traceback.print_exc(file=sys.stdout); sys.stdout.flush()

def err(a, b):
    return (a - b).abs().max().item()

def fwd(hp, wseed, prec, B, T, seed, tag, stops=(0, 1, 2, None)):
    sd = synth.acoustic_state(hp, wseed)
    den = DenoiserHandle(sd, hp["audio_num_mel_bins"], hp["hidden_size"], hp["residual_channels"], hp["residual_layers"],
                         hp["dilation_cycle_length"], hp["timesteps"], precision=prec, prefix="denoise_fn.")
    g = np.random.Generator(np.random.PCG64(seed))
    M, H, C, L = hp["audio_num_mel_bins"], hp["hidden_size"], hp["residual_channels"], hp["residual_layers"]
    spec = torch.from_numpy(g.standard_normal((B, 1, M, T)).astype(np.float32))
    cond = torch.from_numpy((g.standard_normal((B, H, T)) * 0.5).astype(np.float32))
    t = torch.full((B,), int(g.integers(0, hp["timesteps"])), dtype=torch.long)
    taps = {}
    bsel = B - 1                                       # check the LAST clip of the batch (row offsets, gaps)
    with torch.no_grad():
        ref = O.diffnet_forward(sd, spec[bsel:bsel + 1], t[:1], cond[bsel:bsel + 1], hp["dilation_cycle_length"], taps=taps)
    emb = O.step_embedding(sd, t[:1])
    def film(l):
        return torch.nn.functional.linear(emb, sd["denoise_fn.residual_layers.%d.diffusion_projection.weight" % l],
                                          sd["denoise_fn.residual_layers.%d.diffusion_projection.bias" % l])[0]
    for stop in stops:
        if stop is None:
            os.environ.pop("DSVC_DEBUG_STOP_AFTER_LAYERS", None)
        else:
            os.environ["DSVC_DEBUG_STOP_AFTER_LAYERS"] = str(stop)
        out = den.forward(spec.cuda(), t.cuda(), cond.cuda()).cpu()
        rows = den.debug_buffer("xres").shape[0]
        Tp = rows // B if B > 1 else rows
        # clip b occupies rows [b*Tp', b*Tp'+T): Tp' = round_up(T + 8, 32)
        Tpp = ((T + 8 + 31) // 32) * 32
        r0 = bsel * Tpp
        xl = den.debug_buffer("xres")[r0:r0 + T].cpu()
        n = L if stop is None else stop
        if n == 0:
            xh = den.debug_buffer("xh")[r0:r0 + T].cpu()
            print(tag, "stop=0  x_in err %.2e   xh err %.2e (vs x_in+film0)   gap rows zero: %s" % (
                err(xl, taps["x_in"][0].T), err(xh, taps["x_in"][0].T + film(0)[None]),
                bool((den.debug_buffer("xh")[r0 + T:r0 + Tpp] == 0).all().item())), flush=True)
            continue
        gl = den.debug_buffer("g")[r0:r0 + T].cpu(); sk = den.debug_buffer("skip")[r0:r0 + T].cpu()
        skref = sum(taps.get("s%d" % l, 0) for l in range(n)) if "s0" in taps else None
        msg = "%s stop=%s  g%d err %.2e  x%d err %.2e" % (tag, stop, n - 1, err(gl, taps["g%d" % (n - 1)][0].T), n - 1, err(xl, taps["x%d" % (n - 1)][0].T))
        if n < L:
            xh = den.debug_buffer("xh")[r0:r0 + T].cpu()
            msg += "  xh err %.2e" % err(xh, taps["x%d" % (n - 1)][0].T + film(n)[None])
        else:
            msg += "  skip err %.2e  out err %.2e (ref std %.2f)" % (err(sk / L ** 0.5, taps["skip"][0].T), err(out[bsel:bsel + 1], ref), ref.std().item())
        print(msg, flush=True)
    os.environ.pop("DSVC_DEBUG_STOP_AFTER_LAYERS", None)

tiny = synth.tiny_hparams(); full = dict(synth.HPARAMS_44K)
for prec in ("f16_w2", "f16", "f16_d16"):
    stage("fwd tiny %s" % prec, lambda: fwd(tiny, 3, prec, 2, 40, 1, "tiny/" + prec))
for prec in ("f16_w2", "f16", "f16_d16"):
    stage("fwd 44k %s B=1 T=45" % prec, lambda: fwd(full, 0, prec, 1, 45, 2, "44k/" + prec))
stage("fwd 44k w2 B=2 T=861 (latency tiling)", lambda: fwd(full, 0, "f16_w2", 2, 861, 3, "44k-B2/w2", stops=(1, None)))
stage("fwd 44k w2 B=8 T=861 (128-frame tiling)", lambda: fwd(full, 0, "f16_w2", 8, 861, 3, "44k-B8/w2", stops=(0, 1, None)))
stage("fwd 44k f16 B=8 T=861 (128-frame tiling)", lambda: fwd(full, 0, "f16", 8, 861, 3, "44k-B8/f16", stops=(1, None)))

def timing():
    hp = full
    sd = synth.acoustic_state(hp, 0)
    for prec in ("f16", "f16_w2", "f16_d16"):
        den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=prec, prefix="denoise_fn.")
        smp = SamplerHandle(den, sd)
        for B in (1, 8, 32):
            T = 861
            cond = torch.randn(B, 256, T, device="cuda") * 0.5
            smp.sample(cond, 40, seed=1, use_graph=True)   # warm
            torch.cuda.synchronize(); t0 = time.time()
            n = 200 if B == 1 else 40
            smp.sample(cond, n, seed=1, use_graph=True)
            torch.cuda.synchronize(); dt = time.time() - t0
            flop = 55541760 * (1 - 393216 * 20 / 55541760) * T * B * n
            print("timing %s B=%d: %.3f ms/step  -> %.1fx RT @1000 steps, %.1f TFLOP/s(non-hoisted)" % (
                prec, B, dt / n * 1e3, 10.0 * B / (dt / n * 1000), flop / dt / 1e12), flush=True)
            us, rows = smp.profile_gate_kernel(B, T, 3)
            print("  gate kernel %s B=%d: %.1f us/launch, rows %d -> %.1f TFLOP/s (valid frames)" % (prec, B, us, rows, 2 * 768 * 1152 * B * T / us / 1e6), flush=True)
stage("timing", timing)
