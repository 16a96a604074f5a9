/* dsvc_debug.h -- measurement and test-support entry points of libdsvc_hip.so.
 *
 * NOT part of the product surface (include/dsvc.h is): nothing here replaces an interface of the reference.  bench.py's roofline fields and
 * the parity tests' per-layer taps go through these; a host that only wants the drop-in never includes this file.  The functions live in
 * the same shared library (the tests need them on the very kernels that ship), behind the same error convention (int codes,
 * dsvc_last_error). */
#ifndef DSVC_DEBUG_H
#define DSVC_DEBUG_H

#include "dsvc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* measurement aid for bench.py: the dense fp16 MFMA rate (TFLOP/s) and shader clock (GHz) this chip sustains right now on a
 * register-resident loop with random (1) or zero (0) operands -- the rate a roofline fraction can actually approach. */
DSVC_API int dsvc_probe_mfma(int32_t random_data, float* tflops, float* clock_ghz, void* stream);
/* the same probe in detail: out4 = { TFLOP/s over the kernel's wall time (HIP events), TFLOP/s over the 4 ms in-kernel window every wave
 * issues MFMAs for, mean shader clock over all waves [GHz], lowest clock any wave saw [GHz] } */
DSVC_API int dsvc_probe_mfma_detail(int32_t random_data, float* out4, void* stream);

/* debugging aid for the parity tests: copy an internal frame-major buffer ("xres", "g", "skip", "s2", "eps",
 * "condT", "cproj", "film", "xin", "xh") to a device pointer as fp32; rows/ld receive its logical shape. */
DSVC_API int dsvc_denoiser_debug_buffer(dsvc_denoiser* d, const char* name, float* dst, int64_t numel, int32_t* rows, int32_t* ld);

/* test support (explicit handle state; the library reads no environment variable): "stop_after_layers" = n >= 0 makes an evaluation
 * return after n residual layers so that dsvc_denoiser_debug_buffer taps layer n-1 (-1 = off); "two_launch_layer" = 1 runs a residual
 * layer as its two tgemm launches even where the fused layer kernel is the automatic choice, -1 runs the fused kernel wherever it is
 * supported (>= 48 frame tiles) and not only where it is faster (>= 120 tiles), 0 = automatic (bit-equality test of the two forms); "w6_off" /
 * "g6_off" = 1 make a DSVC_PREC_F16_W6 handle run its fused layers with the fp16 lo planes (= F16_W2) / without the gate-output correction
 * (= F16_W6N): the A/B partners of the 6-bit products on one set of packed weights; "defer_skip" != 0 makes
 * the fused layer kernels write the gate output to HBM and leave the skip halves of all layers to ONE contraction per evaluation with
 * pre-composed skip-projection weights (csrc/tskip.h: 14 % fewer HBM bytes per layer, measured time-neutral; default off);
 * "fused_nt" = 1 / 2 / 4 forces the fused layer kernel onto 32- / 64- / 128-frame tiles (round 5: the tile widths give bit-identical results and
 * the library picks the one with the fewest rounds of workgroups; 0 = automatic); "x3t_w6_off" = 1 keeps the fp16 lo plane in an F16_X3T handle's
 * DDPM chain; "fused_tail" = 0 runs the tail of a batched f16_w6 DDPM step (skip projection, output projection + posterior step, the next
 * evaluation's input projection) as the three tgemm launches instead of the one fused kernel (csrc/ttail.h; 1 = automatic, 2 / 3 = its 64- /
 * 32-frame tiling); "profile_kernel" = 1 makes dsvc_sampler_profile_gate_kernel time the output kernel of the two-launch layer instead of the gate kernel.
 * NONE of these is part of the supported surface: they exist for the parity tests and the bench's roofline entries, they change which kernel
 * computes a result (never to an unsupported precision), and a deployment should not call this function.  Pure tuning knobs of variants that
 * were measured and not kept ("layer_prio", "tail_tiling") exist in the -DDSVC_PROFILING build only. */
DSVC_API int dsvc_denoiser_debug_set(dsvc_denoiser* d, const char* key, int32_t value);

/* per-kernel timing for bench.py's roofline: average duration in microseconds of the dominant kernel at this batch size,
 * measured with HIP events on the launch stream over back-to-back launches of all layers (a different dither variant per round:
 * weights as cold as in the real chain), and the number of frames (rows) one launch processed.
 * kind (may be NULL) receives which kernel that is: 0 = the gate kernel (dilated conv + conditioner projection + gate: small
 * batches run a layer as two launches), > 0 = the fused residual-layer kernel (gate GEMM + output projection) and the number of
 * 32-frame N-tiles a workgroup covers: 4 = the throughput tiling, 2 / 1 = the mid-size tilings of f16_w6 / f16_w6n (round 5). */
DSVC_API int dsvc_sampler_profile_gate_kernel(dsvc_sampler* s, int32_t B, int32_t T, int32_t iters,
                                     float* avg_us, int64_t* rows, int32_t* kind, void* stream);

/* measurement aid for bench.py's `plms_50.breakdown_ms` / `ragged`: enable != 0 makes every later dsvc_sample call record HIP events at its
 * phase boundaries on the caller's stream (four extra event records per call).  out_ms (may be NULL; else float[4]) receives the phases of the
 * LAST timed call, in ms, after waiting for it: [0] workspace bucket + clip metadata + the L hoisted conditioner projections, [1] the initial
 * state (x_T noise / q_sample of the reference mel) and its fp16 planes, [2] the sampling chain (eager evaluations, graph capture if the
 * bucket had none, graph replays), [3] denormalisation + mask. */
DSVC_API int dsvc_sampler_phase_times(dsvc_sampler* s, int32_t enable, float* out_ms);

/* test support for the trainer, as dsvc_denoiser_debug_set: "wgrad_fm" = 0 makes the residual layers' weight gradients take the k_split_t +
 * wgrad_nt_kernel path (channel-major copies of every operand) where the architecture would let them be contracted straight from the
 * frame-major operand planes (csrc/wgrad.h: wgrad_fm_kernel, the default since round 5); 1 = automatic.  Both paths compute the same products;
 * tests/test_gpu_train.py holds them to each other.  Not for a deployment. */
DSVC_API int dsvc_trainer_debug_set(dsvc_trainer* t, const char* key, int32_t value);

#ifdef __cplusplus
}
#endif
#endif
