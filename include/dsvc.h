/* dsvc.h -- C ABI of the MI355X-native diff-svc hot path (libdsvc_hip.so).
 *
 * The reference (prophesier/diff-svc) is pure Python; its "FFI" for this path is the set of Python
 * plugin seams listed in SURVEY.md 8(b).  Each entry point below names the reference interface it
 * replaces (file:line under /root/reference).  A binding needs nothing but ctypes: plain pointers,
 * sizes and an opaque stream handle -- no torch types cross this boundary.
 *
 * Conventions
 *   - every function returns 0 on success, a non-zero DSVC_E* code otherwise; nothing throws across
 *     the ABI; dsvc_last_error() returns the calling thread's last message.
 *   - "device" pointers are HIP device pointers owned by the caller (e.g. tensor.data_ptr()); they are
 *     never retained past the call.  "host" pointers are read during the call only.
 *   - work is enqueued on the caller's hipStream_t (void*; NULL = default stream).  Handles are
 *     single-caller (the reference is not re-entrant either: diffusion.py:96,270).
 *   - all floating-point tensors are fp32, C-contiguous, in the reference's own layouts.
 */
#ifndef DSVC_H
#define DSVC_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSVC_ABI_VERSION 9

/* The library is built with -fvisibility=hidden: the entry points below (and those of dsvc_debug.h) are its ONLY dynamic symbols -- no C++
 * internal, kernel stub or runtime template can interpose with (or be interposed by) another HIP library of the host process. */
#if defined(__GNUC__) || defined(__clang__)
#define DSVC_API __attribute__((visibility("default")))
#else
#define DSVC_API
#endif

#define DSVC_VARIANTS_DEFAULT (-1)
enum { DSVC_OK = 0, DSVC_EINVAL = 1, DSVC_EHIP = 2, DSVC_ESTATE = 3, DSVC_ENOMEM = 4 };

/* Operand precision of the two big per-layer MFMA contractions (fp32 accumulate everywhere):
 *   F16    : w, x rounded to fp16                      1 MFMA per product.  With round-to-nearest weights the rounding
 *            error is the SAME at every diffusion step and accumulates to ~6e-3 of mel over a 1000-step chain; with
 *            cfg.weight_variants = N > 1 the library keeps N differently-rounded fp16 copies of the weights that
 *            average to w (time-dithered rounding, step t uses copy t % N): N = 64 measures ~7e-4, inside the bar.
 *            (measured over eight (clip, noise) pairs against the real reference: 7.8e-4 ... 1.27e-3, two of fourteen over -- NOT robustly inside the bar)
 *   F16_MIX: F16 with N dithered copies for the dilated conv, w = w_hi + w_lo for the output 1x1 (whose error goes straight
 *            into the residual stream and the skip sum): 6.7e-4 ... 9.7e-4 on single clips, 1.14e-3 on one clip of a batch of 32
 *            (over the bar): round 2's DDPM default, no longer shipped.
 *   F16_W2 : w = w_hi + w_lo, x fp16                   2 MFMAs              (6.2e-4 ... 8.8e-4 mel after 1000 steps over 27 goldens;
 *            2.3e-4 on conditioned checkpoints).  What the Python drop-in's "auto" runs batched DDPM at (>= 6000 frames per call).
 *   F16_X3 : w = w_hi + w_lo, x = x_hi + x_lo          3 MFMAs              (fp32-class, ~1e-5) on the conv_gemm engine
 *   F16_X3T: the same operand scheme on the tgemm engine (activation rows hold [x_hi | x_lo] planes; the weight stream is F16_W2's):
 *            the fp32-class scheme at the speed class of the small-batch tgemm kernels (3.3e-5 ... 4.9e-5 mel after 1000 steps on 21
 *            goldens, +14 ... 20 % over F16_W2 up to six 10 s clips, 1.5 ... 1.9x beyond).  "auto": DDPM calls under 6000 frames, PLMS, forward().
 *            Round 4: in the 32-frame tilings (up to ~6 clips) and when the caller knows the step (the sampler's DDPM loop), the w_lo * x_hi
 *            product runs as one K = 64 six-bit MFMA per 64 input channels on time-dithered fp6 codes of w_lo (weight_variants roundings):
 *            the lo plane streams 384 B instead of 1 KiB per k16 step -- 0.391 instead of 0.417 ms per step for one clip, 7.6e-5 ... 8.4e-5
 *            instead of 3.2e-5 ... 3.3e-5 mel after 1000 steps.
 *   F16_W6 : (round 4) F16_W2 with every correction term on the block-scaled 6-bit matrix instruction (v_mfma_scale_f32_32x32x64_f8f6f4, 4x the
 *            fp16 MFMA rate) inside the fused layer kernel.  Per 64 input channels: 16 fp16 MFMAs (w_hi x) + 4 six-bit ones (w_lo as
 *            time-dithered fp6 E2M3 codes -- weight_variants roundings, one power-of-two scale per conv -- against x converted to bf6 E3M2 in
 *            registers) instead of 32 fp16 MFMAs, in the dilated conv AND the output 1x1; the output 1x1 additionally adds W6 * g_lo6: the gate
 *            epilogue keeps bf6((g - fp16(g)) 2^16) beside fp16(g) in LDS, which removes the fp16 rounding of the gate output (57 % of the
 *            chain's error variance).  A correction term is 2^-12 of its product, so 4 significant bits carry it.  Measured at 32 clips:
 *            133 instead of 145 us per layer, 4.0e-4 ... 5.7e-4 mel after 1000 steps on 31 real-reference goldens (F16_W2: 6.2e-4 ... 9.1e-4).
 *            Batches too small for the fused layer kernel run F16_W2 itself (the handle keeps the fp16 lo planes as well).
 *            "auto": DDPM calls from 6000 frames.
 *   F16_W6N: F16_W6 without the g_lo correction: F16_W2's error class at 125 us per layer. */
enum { DSVC_PREC_F16 = 0, DSVC_PREC_F16_W2 = 1, DSVC_PREC_F16_X3 = 2, DSVC_PREC_F16_MIX = 3, DSVC_PREC_F16_X3T = 4, DSVC_PREC_F16_W6 = 5,
       DSVC_PREC_F16_W6N = 6 };

DSVC_API int dsvc_abi_version(void);
DSVC_API const char* dsvc_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Denoiser -- replaces network/diff/net.py:86-135 (class DiffNet), selected through the DIFF_DECODERS
 * registry (infer_tools/infer_tool.py:107-111,124; training/task/SVC_task.py:19-23).
 * ---------------------------------------------------------------------------------------------- */
typedef struct dsvc_denoiser dsvc_denoiser;

typedef struct {
    int32_t mel_bins;        /* in_dims = hparams['audio_num_mel_bins']          (net.py:87)  */
    int32_t hidden;          /* encoder_hidden = hparams['hidden_size']          (net.py:91)  */
    int32_t channels;        /* residual_channels                                 (net.py:93)  */
    int32_t layers;          /* residual_layers                                   (net.py:92)  */
    int32_t dilation_cycle;  /* dilation_cycle_length                             (net.py:94)  */
    int32_t max_steps;       /* number of integer diffusion steps to tabulate (timesteps)      */
    int32_t precision;       /* DSVC_PREC_* for the two big per-layer contractions              */
    int32_t weight_variants; /* DSVC_PREC_F16 / F16_MIX / F16_W6 / F16_W6N / F16_X3T: number of dithered weight roundings; 0 or 1 = one nearest rounding
                              * (what a zero-initialised cfg has always meant), DSVC_VARIANTS_DEFAULT (-1) = the scheme's default: 64 for the
                              * three 6-bit schemes (64x the fp6 code-plane memory: 0.4 GB for F16_W6, 1.1 GB for F16_X3T at the 44.1 kHz
                              * architecture), 1 otherwise */
} dsvc_denoiser_cfg;

DSVC_API int dsvc_denoiser_create(const dsvc_denoiser_cfg* cfg, dsvc_denoiser** out);
/* name = state_dict key of DiffNet (e.g. "residual_layers.3.dilated_conv.weight"); host fp32 data in the
 * checkpoint's own layout (Conv1d [out,in,k], Linear [out,in]).  Replaces load_state_dict for this module
 * (utils/__init__.py:178-209). */
DSVC_API int dsvc_denoiser_load_tensor(dsvc_denoiser* d, const char* name, const float* host, int64_t numel);
/* packs weights into MFMA fragment order on the device and builds the step-embedding / FiLM tables */
DSVC_API int dsvc_denoiser_finalize(dsvc_denoiser* d);
DSVC_API void dsvc_denoiser_destroy(dsvc_denoiser* d);

/* DiffNet.forward(spec, diffusion_step, cond)  (net.py:112-135)
 *   spec [B,1,M,T] device, t [B] int32 device (one step per clip), cond [B,H,T] device -> out [B,1,M,T] device.
 * cond_changed != 0 recomputes the hoisted conditioner projections (they depend on cond only). */
DSVC_API int dsvc_denoiser_forward(dsvc_denoiser* d, const float* spec, const int32_t* t, const float* cond,
                          float* out, int32_t B, int32_t T, int32_t cond_changed, void* stream);

/* dsvc_denoiser_forward clamps diffusion steps outside [0, max_steps) on the device (the step embedding is tabulated for the integer
 * steps of the schedule, net.py:32-44,99-103) and raises a sticky flag instead of synchronising on every call of the 1000-calls-per-clip
 * denoiser seam; the flag is reported by this function ONLY (it first waits for `stream`, then clears the flag): later, valid forward calls are
 * executed, not rejected for a predecessor's argument.  The host wrappers call it where they synchronise anyway (DiffNetHip.check(), the
 * trainer every 100 steps, SvcPipeline.check()). */
DSVC_API int dsvc_denoiser_check(dsvc_denoiser* d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Sampler -- replaces GaussianDiffusion.forward(infer=True) from the initial x to mel_out
 * (network/diff/diffusion.py:255-283) with p_sample (:156-163) and p_sample_plms (:165-198).
 * ---------------------------------------------------------------------------------------------- */
typedef struct dsvc_sampler dsvc_sampler;

DSVC_API int dsvc_sampler_create(dsvc_denoiser* d, dsvc_sampler** out);
/* the registered buffers of GaussianDiffusion (diffusion.py:100-123): "alphas_cumprod",
 * "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1",
 * "posterior_mean_coef2", "posterior_log_variance_clipped", "sqrt_alphas_cumprod",
 * "sqrt_one_minus_alphas_cumprod", "spec_min", "spec_max".  Taken from the checkpoint, never recomputed
 * (SURVEY.md 0.8). */
DSVC_API int dsvc_sampler_load_tensor(dsvc_sampler* s, const char* name, const float* host, int64_t numel);
DSVC_API int dsvc_sampler_finalize(dsvc_sampler* s);
DSVC_API void dsvc_sampler_destroy(dsvc_sampler* s);

typedef struct {
    int32_t B, T;              /* clips, mel frames per clip                                               */
    const float* cond;         /* [B,H,T] device: ret['decoder_inp'].transpose(1,2)  (diffusion.py:232-234) */
    const float* x_init;       /* [B,1,M,T] device, or NULL: x_T ~ N(0,1) from Philox(seed, clip)          */
    const float* ref_mel;      /* [B,T,M] device (log10 mel) or NULL: use_gt_mel start (diffusion.py:255-261):
                                * x = q_sample(norm_spec(ref_mel), t_start - 1, noise) (diffusion.py:200-205,286-287) with the
                                * noise drawn from the x_T Philox stream; takes precedence over x_init                  */
    const int32_t* mel2ph;     /* [B,T] device or NULL: output mask (mel2ph > 0)      (diffusion.py:280-281) */
    uint64_t seed;             /* Philox key for x_T and the per-step noise z                                */
    int32_t first_clip;        /* Philox clip id of batch element 0 (clip b uses first_clip + b)             */
    const int32_t* clip_ids;   /* [B] device or NULL: explicit Philox clip id per batch element (overrides first_clip) -- a
                                * clip keeps its noise stream wherever a sharded job places it                */
    const int32_t* clip_lens;  /* [B] device or NULL: valid frames per clip (1..T).  Frames >= clip_lens[b] are the convs'
                                * ZERO PADDING, exactly as if clip b had been run alone at its own length (the reference is
                                * B=1, infer_tool.py:277); their mel_out rows are 0.  NULL = every clip has T frames.       */
    const int32_t* clip_lens_host; /* [B] HOST copy of clip_lens or NULL (ABI v9; read during the call only).  Scheduling only, never results: the
                                * fused layer kernel's workgroups on tiles that lie wholly beyond their clip's length return at once, and
                                * with the lengths known on the host the tile width is chosen by the tiles that have work (a ragged batch --
                                * the chunks of one utterance, infer.py:44-67 -- then costs its own frames, not the padded rectangle)       */
    int32_t t_start;           /* K_step, or add_noise_step with ref_mel: runs t = t_start-1 ... 0           */
    int32_t t_stop;            /* normally 0; tests may stop the chain early (runs down to t_stop)           */
    int32_t speedup;           /* hparams['pndm_speedup']: <= 1 -> DDPM, > 1 -> PLMS with that interval      */
    int32_t use_graph;         /* 1: replay the steps through captured hipGraphs where that pays (the latency-bound small tilings: ~10 us
                                * kernels); large calls on the fused layer kernel are launched eagerly either way -- same speed, no capture.
                                * 0: always launch eagerly                                                                              */
    float* mel_out;            /* [B,T,M] device: denorm_spec(x) * (mel2ph > 0)                              */
    float* x_out;              /* [B,1,M,T] device or NULL: final normalised state (for tests)              */
} dsvc_sample_args;

DSVC_API int dsvc_sample(dsvc_sampler* s, const dsvc_sample_args* a, void* stream);

/* Serving telemetry.  The reference's driver calls the model chunk by chunk, every chunk with its own T (infer.py:44-67,
 * infer_tools/infer_tool.py:155-159,276).  dsvc_sample lays a call out in a BUCKET -- a clip occupies round_up(T + largest dilation, 128) rows --
 * and keeps the buckets it has seen (workspace zeroed once, captured DDPM / PLMS chains) in an LRU, so a chunk whose bucket exists costs no
 * allocation, no clearing and no graph capture.  out[0..n) (n <= 6) receives: DDPM periods captured, PLMS chains captured, graph launches,
 * buckets allocated, calls served by an existing bucket, captured chains alive -- all since the handle was created. */
DSVC_API int dsvc_sampler_stats(dsvc_sampler* s, int64_t* out, int32_t n);

/* ------------------------------------------------------------------------------------------------
 * Vocoder -- replaces modules/nsf_hifigan/models.py:325-387 (Generator.forward) + :14-30 (load_model),
 * called through network/vocoders/nsf_hifigan.py:47-73 (NsfHifiGAN.spec2wav); the same kernels serve the 24 kHz
 * generator modules/hifigan/hifigan.py:104-178 (HifiGanGenerator: identical structure, identical source module,
 * mel_scale 1, optional source) behind network/vocoders/hifigan.py:46-76 (HifiGAN.spec2wav).
 * ---------------------------------------------------------------------------------------------- */
typedef struct dsvc_vocoder dsvc_vocoder;

typedef struct {
    int32_t num_mels, upsample_initial_channel, sampling_rate;
    int32_t n_ups;                 /* len(upsample_rates) <= 8   */
    int32_t upsample_rates[8];
    int32_t upsample_kernel_sizes[8];
    int32_t n_kernels;             /* len(resblock_kernel_sizes) <= 4 */
    int32_t resblock_kernel_sizes[4];
    int32_t resblock_dilations[4][3];
    int32_t harmonics;             /* harmonic_num = 8 (models.py:334) */
    int32_t precision;             /* DSVC_PREC_* (F16_X3 keeps the waveform within 1e-4 RMS) */
    float mel_scale;               /* the generator input is mel_scale * mel: 2.30259 (log10 -> ln) for NsfHifiGAN.spec2wav
                                    * (nsf_hifigan.py:63-65), 1 for the 24 kHz HifiGAN wrapper (network/vocoders/hifigan.py:64) */
    int32_t use_source;            /* 1: harmonic source + noise convs (NSF: models.py:363-375; HifiGanGenerator with
                                    * use_pitch_embed and an f0, modules/hifigan/hifigan.py:110-116,150-162); 0: plain HiFi-GAN */
    int32_t resblock;              /* h.resblock: 1 = ResBlock1 (conv pairs, three dilations: models.py:33-64), 2 = ResBlock2 (one conv per
                                    * residual step, models.py:73-91; n_dilations of them, two in the published configs); 0 means 1 */
    int32_t n_dilations;           /* entries used of resblock_dilations[j] (ResBlock1: 3); 0 means 3 */
} dsvc_vocoder_cfg;

DSVC_API int dsvc_vocoder_create(const dsvc_vocoder_cfg* cfg, dsvc_vocoder** out);
/* name = key of the checkpoint's 'generator' dict, weight-norm pairs included
 * ("ups.0.weight_g", "ups.0.weight_v", ...); folded here like remove_weight_norm (models.py:28,389-396). */
DSVC_API int dsvc_vocoder_load_tensor(dsvc_vocoder* v, const char* name, const float* host, int64_t numel);
DSVC_API int dsvc_vocoder_finalize(dsvc_vocoder* v);
DSVC_API void dsvc_vocoder_destroy(dsvc_vocoder* v);

/* Generator.forward(mel_scale * mel^T, f0)   mel [B,T,M] device, f0 [B,T] Hz device (NULL iff use_source == 0) -> wav [B, T*hop] device.
 * The source module's random draws (models.py:192,271) come from Philox(seed, clip id), clip id = clip_ids[b] (device [B]) or,
 * when clip_ids is NULL, first_clip + b. */
DSVC_API int dsvc_vocode(dsvc_vocoder* v, const float* mel, const float* f0, float* wav, int32_t B, int32_t T,
                uint64_t seed, int32_t first_clip, const int32_t* clip_ids, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Mel front-end -- replaces modules/nsf_hifigan/nvSTFT.py:72-104 (STFT.get_mel) + the log10 scale of
 * network/vocoders/nsf_hifigan.py:86-91 (wav2spec).
 * ---------------------------------------------------------------------------------------------- */
typedef struct dsvc_melspec dsvc_melspec;

typedef struct {
    int32_t n_fft, win_size, hop, n_mels;
    float clip_val;                /* mode 0: 1e-5 (nvSTFT's clamp); mode 1: hparams['wav2spec_eps'] */
    int32_t mode;                  /* 0: nvSTFT.py:72-104 (reflect pad (n_fft-hop)/2, sqrt(|X|^2 + 1e-9), ln -> log10);
                                      1: process_utterance, preprocessing/data_gen_utils.py:124-136, the 24 kHz PWG / HifiGAN front-end
                                         (librosa.stft centred with zero padding, |X|, log10(max(eps, mel)), T = 1 + N // hop) */
} dsvc_melspec_cfg;

/* mel_basis: host [n_mels][n_fft/2+1] fp32 (librosa.filters.mel, built by the Python host) */
DSVC_API int dsvc_melspec_create(const dsvc_melspec_cfg* cfg, const float* mel_basis, dsvc_melspec** out);
DSVC_API void dsvc_melspec_destroy(dsvc_melspec* m);
/* wav [B,N] device -> mel [B,T,n_mels] log10 device, T = (N - hop) / hop + 1 ... see dsvc_melspec_frames */
DSVC_API int dsvc_melspec_frames(const dsvc_melspec* m, int64_t n_samples, int32_t* frames);
DSVC_API int dsvc_melspec_run(dsvc_melspec* m, const float* wav, float* mel, int32_t B, int64_t n_samples, void* stream);
/* process_utterance(..., return_linear=True) (preprocessing/data_gen_utils.py:144-149; PWG.wav2spec(return_linear=True), network/vocoders/pwg.py:
 * 106-122): the mel as above AND the normalised linear spectrogram audio.normalize(audio.amp_to_db(|X|)) = (20 log10(max(1e-5, |X|)) - min_level_db)
 * / -min_level_db (utils/audio.py:51-56; hparams['min_level_db'], -120 in training/config.yaml:90), linear [B, frames, n_fft / 2 + 1] device. */
DSVC_API int dsvc_melspec_run_linear(dsvc_melspec* m, const float* wav, float* mel, float* linear, float min_level_db, int32_t B, int64_t n_samples,
                                     void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pitch index work of the condition builder -- replaces the host arithmetic of add_pitch (modules/fastspeech/fs2.py:229-237):
 *   f0_denorm = denorm_f0(f0, uv, pitch_padding = mel2ph == 0)   (utils/pitch_utils.py:63-76, pitch_norm 'log': 2**f0, 0 where off)
 *   coarse    = f0_to_coarse(f0_denorm)                          (utils/pitch_utils.py:17-31) = 1 + #{k: f0 >= thresholds[k]}
 * thresholds: the ascending fp32 values of the NORMALISED pitch at which the reference's own expression steps to the next bin
 * (found by bisection with that expression, diff-svc_amd/cond.py), n = B*T elements, all pointers device; uv may be NULL.
 * ---------------------------------------------------------------------------------------------- */
DSVC_API int dsvc_pitch_coarse(const float* f0_log2, const int64_t* mel2ph, const float* uv, const float* thresholds, int32_t n_thresholds,
                      int64_t n, float* f0_denorm, int64_t* coarse, void* stream);

/* The whole condition builder of the no_fs2 configuration in one launch -- replaces FastSpeech2.forward's no_fs2 branch
 * (modules/fastspeech/fs2.py:133-148: decoder_inp = gather(pad(hubert), mel2ph)), add_pitch (:229-237, as dsvc_pitch_coarse above) and
 * insert4's (decoder_inp + pitch_embed[coarse]) * (mel2ph > 0) (training/train_pipeline.py:213-218), and the transpose of
 * GaussianDiffusion.forward (network/diff/diffusion.py:236):
 *   hubert [B,N,H], mel2ph [B,T] int64 (0 = padding, else 1-based unit index), f0_log2 [B,T] (IN/OUT: zeroed where mel2ph == 0, as the
 *   reference mutates its argument, fs2.py:231), uv [B,T] or NULL, pitch_embed [vocab,H]  ->  decoder_inp [B,T,H], cond_bht [B,H,T],
 *   f0_denorm [B,T], coarse [B,T] int64.  All pointers device.  Values are bit-identical to the reference's fp32 expression. */
DSVC_API int dsvc_cond_build(const float* hubert, const int64_t* mel2ph, float* f0_log2, const float* uv, const float* thresholds,
                    int32_t n_thresholds, const float* pitch_embed, int32_t B, int32_t N, int32_t T, int32_t H,
                    float* decoder_inp, float* cond_bht, float* f0_denorm, int64_t* coarse, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Content encoder -- replaces network/hubert/hubert_model.py:67-77 (HubertSoft.units: pad 40|40 -> FeatureExtractor ->
 * FeatureProjection -> + PositionalConvEmbedding -> LayerNorm -> 12 post-LN transformer layers -> proj), loaded by
 * hubert_soft() (:218-231) and called through preprocessing/hubertinfer.py:30-42 (Hubertencoder.encode -> get_units).
 * The architecture is fixed (HuBERT-base, 768/12/3072, 256-dim soft units at 50 Hz from 16 kHz audio).
 * ---------------------------------------------------------------------------------------------- */
typedef struct dsvc_hubert dsvc_hubert;

DSVC_API int dsvc_hubert_create(dsvc_hubert** out);
/* name = key of HubertSoft.state_dict() ("feature_extractor.conv0.weight", "positional_embedding.conv.weight_g",
 * "encoder.layers.3.self_attn.in_proj_weight", ...); host fp32 in the checkpoint's layout.  Keys the inference path does not use
 * (masked_spec_embed, label_embedding.weight) are accepted and ignored. */
DSVC_API int dsvc_hubert_load_tensor(dsvc_hubert* h, const char* name, const float* host, int64_t numel);
DSVC_API int dsvc_hubert_finalize(dsvc_hubert* h);
DSVC_API void dsvc_hubert_destroy(dsvc_hubert* h);
/* frames produced for n_samples of 16 kHz audio: seven valid convolutions of total stride 320 over n_samples + 80 */
DSVC_API int dsvc_hubert_frames(int64_t n_samples, int32_t* frames);
/* wav [n_samples] device fp32 in [-1, 1] at 16 kHz (one utterance) -> units [frames][256] device */
DSVC_API int dsvc_hubert_units(dsvc_hubert* h, const float* wav, int64_t n_samples, float* units, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pitch extractor (mel -> f0) -- replaces modules/fastspeech/pe.py:120-148 (PitchExtractor.forward: Prenet :7-42 -> ConvStacks
 * :81-117 -> PitchPredictor modules/fastspeech/tts_modules.py:192-235 -> denorm_f0 utils/pitch_utils.py:63-76), which the 24 kHz
 * path runs on the sampled mel to drive the vocoder (infer_tools/infer_tool.py:134-136,165-166).  Eval mode (BatchNorm running
 * statistics, no dropout), 'SAME' padding.
 * ---------------------------------------------------------------------------------------------- */
typedef struct dsvc_pe dsvc_pe;

typedef struct {
    int32_t n_mel;              /* PitchExtractor(n_mel_bins), 80 */
    int32_t hidden;             /* hparams['hidden_size'], multiple of 16 */
    int32_t predictor_hidden;   /* hparams['predictor_hidden'] if > 0 else hidden */
    int32_t prenet_layers;      /* Prenet n_layers, 3 */
    int32_t conv_layers;        /* PitchExtractor(conv_layers), 2; 0 = no mel_encoder */
    int32_t predictor_layers;   /* 5 */
    int32_t kernel;             /* Prenet / ConvStacks kernel, 5 */
    int32_t predictor_kernel;   /* hparams['predictor_kernel'], 5 */
    int32_t pitch_norm;         /* hparams['pitch_norm']: 0 = 'log' (f0 = 2**x), 1 = 'standard' (x*f0_std + f0_mean), 2 = neither */
    int32_t use_uv;             /* pitch_type == 'frame' and hparams['use_uv']: f0 = 0 where pitch_pred[..., 1] > 0 */
    float f0_mean, f0_std;
} dsvc_pe_cfg;

DSVC_API int dsvc_pe_create(const dsvc_pe_cfg* cfg, dsvc_pe** out);
/* name = key of PitchExtractor.state_dict() ("mel_prenet.layers.0.0.weight", "mel_prenet.layers.0.2.running_var",
 * "mel_encoder.conv.1.norm.weight", "pitch_predictor.conv.4.3.bias", "pitch_predictor.pos_embed_alpha", ...); host fp32 in the
 * checkpoint's layout.  Keys the forward does not read (num_batches_tracked, embed_positions._float_tensor) are accepted and ignored. */
DSVC_API int dsvc_pe_load_tensor(dsvc_pe* p, const char* name, const float* host, int64_t numel);
DSVC_API int dsvc_pe_finalize(dsvc_pe* p);
/* the sinusoidal table SinusoidalPositionalEmbedding.weights [n_rows][hidden] (common_layers.py:105-122; a constructor constant, not
 * in the state dict); a run over T frames needs n_rows >= T + 1 */
DSVC_API int dsvc_pe_set_positions(dsvc_pe* p, const float* host_table, int32_t n_rows);
DSVC_API void dsvc_pe_destroy(dsvc_pe* p);
/* mel [B][T][n_mel] device fp32 (all-zero rows are padding) -> pitch_pred [B][T][2] (may be NULL), f0 [B][T] (Hz) device */
DSVC_API int dsvc_pe_run(dsvc_pe* p, const float* mel, int32_t B, int32_t T, float* pitch_pred, float* f0, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training step -- replaces GaussianDiffusion.forward(infer=False) -> p_losses (network/diff/diffusion.py:200-225,237-241;
 * training/train_pipeline.py:222-238) with autograd through DiffNet (network/diff/net.py:112-135), and the optimizer step of
 * training/task/SVC_task.py:60-66,116-125 (AdamW) with utils/pl_utils.py:1081-1084 (clip_grad_norm_).
 * Parameters and gradients live in two caller-owned flat fp32 device buffers (a torch tensor each: DDP all-reduces the gradient
 * buffer with RCCL between dsvc_trainer_step and dsvc_adamw_step); dsvc_trainer_param_info gives each state-dict tensor's slice.
 * ---------------------------------------------------------------------------------------------- */
typedef struct dsvc_trainer dsvc_trainer;

typedef struct {
    int32_t mel_bins, hidden, channels, layers, dilation_cycle;   /* as dsvc_denoiser_cfg */
    int32_t timesteps;        /* schedule length K (t ~ U{0..K-1}, train_pipeline.py:233) */
    int32_t loss_l1;          /* hparams['diff_loss_type']: 1 = 'l1', 0 = 'l2' (diffusion.py:213-223) */
    int32_t pitch_vocab;      /* rows of fs2.pitch_embed.weight (300, fs2.py:73) */
} dsvc_trainer_cfg;

typedef struct {
    int32_t B, T;             /* clips, mel frames per clip */
    const float* mel;         /* [B,T,M] device: target log-mel (ref_mels); normalised inside (norm_spec, diffusion.py:286-287) */
    const float* cond;        /* [B,H,T] device: decoder_inp^T of this batch (fs2.py:94-154) */
    const int32_t* t;         /* [B] device: the diffusion step of every clip (the reference draws torch.randint) */
    const int32_t* pitch;     /* [B,T] device or NULL: coarse pitch bins -> gradient of fs2.pitch_embed.weight through cond */
    const int32_t* mel2ph;    /* [B,T] device or NULL: frames with mel2ph == 0 carry no pitch-embedding gradient */
    uint64_t seed;            /* Philox key of the noise eps ~ N(0,1) (stream 5: (element/4, 0, clip id)) */
    int32_t first_clip;       /* Philox clip id of batch element 0 ... */
    const int32_t* clip_ids;  /* ... or explicit ids [B] device */
} dsvc_train_args;

DSVC_API int dsvc_trainer_create(const dsvc_trainer_cfg* cfg, dsvc_trainer** out);
DSVC_API void dsvc_trainer_destroy(dsvc_trainer* t);
/* the flat layout: number of tensors / of floats; tensor i's state-dict name ("denoise_fn.*" in DiffNet.state_dict() order, then
 * "fs2.pitch_embed.weight"), offset and size in floats.  Tensors keep the checkpoint's own layouts (Conv1d [out,in,k], Linear [out,in]). */
DSVC_API int dsvc_trainer_param_count(const dsvc_trainer* t, int64_t* n_tensors, int64_t* n_floats);
DSVC_API int dsvc_trainer_param_info(const dsvc_trainer* t, int64_t i, const char** name, int64_t* offset, int64_t* numel);
DSVC_API int dsvc_trainer_bind(dsvc_trainer* t, float* params, float* grads);          /* device, n_floats each; retained until destroy */
/* registered buffers of GaussianDiffusion the loss needs (diffusion.py:107-108,122-123), host pointers */
DSVC_API int dsvc_trainer_set_schedule(dsvc_trainer* t, const float* sqrt_alphas_cumprod, const float* sqrt_one_minus_alphas_cumprod, int32_t K,
                              const float* spec_min, const float* spec_max, int32_t n_spec);
/* forward + backward of one batch: grads <- d loss / d params (overwritten), *loss_out (device float, may be NULL) <- the loss */
DSVC_API int dsvc_trainer_step(dsvc_trainer* t, const dsvc_train_args* a, float* loss_out, void* stream);
/* The same step in three phases, for a data-parallel host that all-reduces finished gradient slices while the backward pass of the
 * earlier layers still runs (what the reference's DDP reducer does bucket by bucket, utils/pl_utils.py:187-221):
 *   begin              inputs, forward, loss, backward of the tail.  Final afterwards: the contiguous slice
 *                      [denoise_fn.skip_projection.weight .. denoise_fn.output_projection.bias]
 *   layers(l_hi, l_lo) backward of residual layers l_hi-1 ... l_lo (top down; the first call has l_hi = residual_layers, each later call
 *                      continues where the previous one stopped).  Final afterwards: the contiguous slice
 *                      [denoise_fn.residual_layers.<l_lo>.dilated_conv.weight .. denoise_fn.residual_layers.<l_hi-1>.output_projection.bias]
 *   end                input projection, step-embedding MLP, pitch embedding; *loss_out.  Final afterwards: everything
 *                      ([denoise_fn.input_projection.weight .. denoise_fn.mlp.2.bias] and fs2.pitch_embed.weight were the missing slices).
 * dsvc_trainer_step == begin; layers(residual_layers, 0); end.  All three enqueue on `stream` and return. */
DSVC_API int dsvc_trainer_step_begin(dsvc_trainer* t, const dsvc_train_args* a, void* stream);
/* Diffusion steps outside [0, timesteps) are clamped on the device (the reference's extract() would raise an IndexError, diffusion.py:22-25,
 * but a device-to-host check per step is not what a training loop wants): a sticky flag records it, and this call -- which waits for
 * `stream` -- returns DSVC_EINVAL once and clears it.  Same contract as dsvc_denoiser_check. */
DSVC_API int dsvc_trainer_check(dsvc_trainer* t, void* stream);
DSVC_API int dsvc_trainer_step_layers(dsvc_trainer* t, int32_t l_hi, int32_t l_lo, void* stream);
DSVC_API int dsvc_trainer_step_end(dsvc_trainer* t, float* loss_out, void* stream);
/* torch.optim.AdamW update of a flat buffer.  The gradient is multiplied by *grad_scale_dev (device, e.g. the clip coefficient)
 * when that pointer is not NULL, else by grad_scale. */
DSVC_API int dsvc_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int64_t step, const float* grad_scale_dev, float grad_scale, void* stream);
/* clip_grad_norm_: *sqnorm_dev <- ||g||^2, *coef_dev <- min(1, max_norm / (||g|| + 1e-6)) */
DSVC_API int dsvc_grad_clip_coef(const float* grads, int64_t n, float max_norm, float* sqnorm_dev, float* coef_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DSVC_H */
