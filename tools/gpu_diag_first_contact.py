"""First-contact diagnostics for the GPU box: prints per-stage errors instead of asserting.
Usage on the box: python tools/gpu_diag_first_contact.py > gpurun_out/diag.txt"""
import sys, os, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import diffsvc_amd
from diffsvc_amd import synth
from diffsvc_amd.engine import DenoiserHandle, SamplerHandle
import dsvc_oracle as O

print("torch", torch.__version__, "cuda", torch.cuda.is_available(), torch.cuda.get_device_name(0))
print(set(l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'libdsvc' in l))

def stage(name, fn):
    t0 = time.time()
    try:
        fn()
        print("[ok] %s (%.2fs)" % (name, time.time() - t0), flush=True)
    except Exception:
        print("[FAIL] %s" % name); traceback.print_exc(file=sys.stdout); sys.stdout.flush()

def fwd(hp, wseed, prec, B, T, seed, tag):
    sd = synth.acoustic_state(hp, wseed)
    den = DenoiserHandle(sd, hp["audio_num_mel_bins"], hp["hidden_size"], hp["residual_channels"], hp["residual_layers"],
                         hp["dilation_cycle_length"], hp["timesteps"], precision=prec, prefix="denoise_fn.")
    g = np.random.Generator(np.random.PCG64(seed))
    M, H, C = hp["audio_num_mel_bins"], hp["hidden_size"], hp["residual_channels"]
    spec = torch.from_numpy(g.standard_normal((B, 1, M, T)).astype(np.float32))
    cond = torch.from_numpy((g.standard_normal((B, H, T)) * 0.5).astype(np.float32))
    t = torch.from_numpy(g.integers(0, hp["timesteps"], size=(B,)))
    taps = {}
    with torch.no_grad():
        ref = O.diffnet_forward(sd, spec[:1], t[:1], cond[:1], hp["dilation_cycle_length"], taps=taps)
    out = den.forward(spec.cuda(), t.cuda(), cond.cuda()).cpu()
    L = hp["residual_layers"]
    # film table check
    film = den.debug_buffer("film").cpu().reshape(hp["timesteps"], L, C)
    emb = O.step_embedding(sd, t[:1])
    f0 = torch.nn.functional.linear(emb, sd["denoise_fn.residual_layers.0.diffusion_projection.weight"], sd["denoise_fn.residual_layers.0.diffusion_projection.bias"])
    print(tag, "film err %.2e" % (film[int(t[0]), 0] - f0[0]).abs().max().item())
    cp = den.debug_buffer("cproj").cpu()
    print(tag, "cproj shape", tuple(cp.shape), "finite", bool(torch.isfinite(cp[:T]).all()))
    xl = den.debug_buffer("xres")[:T].cpu(); gl = den.debug_buffer("g")[:T].cpu(); sk = den.debug_buffer("skip")[:T].cpu()
    print(tag, "x_last err %.2e  g_last err %.2e  skip err %.2e  out err %.2e (ref std %.2f)" % (
        (xl - taps["x%d" % (L - 1)][0].T).abs().max().item(), (gl - taps["g%d" % (L - 1)][0].T).abs().max().item(),
        (sk / L ** 0.5 - taps["skip"][0].T).abs().max().item(), (out[:1] - ref).abs().max().item(), ref.std().item()))

tiny = synth.tiny_hparams(); full = dict(synth.HPARAMS_44K)
for prec in ("f16_x3", "f16_w2", "f16"):
    stage("fwd tiny %s" % prec, lambda: fwd(tiny, 3, prec, 2, 40, 1, "tiny/" + prec))
for prec in ("f16_x3", "f16_w2", "f16"):
    stage("fwd 44k %s T=45" % prec, lambda: fwd(full, 0, prec, 1, 45, 2, "44k/" + prec))
stage("fwd 44k x3 B=8 T=861 (L tiling)", lambda: fwd(full, 0, "f16_x3", 8, 861, 3, "44kL/x3"))
stage("fwd 44k w2 B=8 T=861 (L tiling)", lambda: fwd(full, 0, "f16_w2", 8, 861, 3, "44kL/w2"))

def timing():
    hp = full
    sd = synth.acoustic_state(hp, 0)
    for prec in ("f16", "f16_w2", "f16_x3"):
        den = DenoiserHandle(sd, 128, 256, 384, 20, 4, 1000, precision=prec, prefix="denoise_fn.")
        smp = SamplerHandle(den, sd)
        for B in (1, 8, 32):
            T = 861
            cond = torch.randn(B, 256, T, device="cuda") * 0.5
            for use_graph in (False, True):
                smp.sample(cond, 40, seed=1, use_graph=use_graph)   # warm
                torch.cuda.synchronize(); t0 = time.time()
                n = 100 if B == 1 else 40
                smp.sample(cond, n, seed=1, use_graph=use_graph)
                torch.cuda.synchronize(); dt = time.time() - t0
                flop = 55541760 * (1 - 393216 * 20 / 55541760) * T * B * n
                print("timing %s B=%d graph=%d: %.3f ms/step  -> %.1fx RT @1000 steps, %.1f TFLOP/s(non-hoisted)" % (
                    prec, B, use_graph, dt / n * 1e3, 10.0 * B / (dt / n * 1000), flop / dt / 1e12), flush=True)
            us, rows = smp.profile_gate_kernel(B, T, 3)
            print("  gate kernel %s B=%d: %.1f us/launch, rows %d -> %.1f TFLOP/s" % (prec, B, us, rows, 2 * 768 * 1152 * rows / us / 1e6))
stage("timing", timing)
