// STFT -> mel front-end behind the C ABI (include/dsvc.h).
// Reference: modules/nsf_hifigan/nvSTFT.py:72-104 (STFT.get_mel: reflect pad (n_fft-hop)/2, hann window,
// torch.stft(center=False), sqrt(re^2+im^2+1e-9), mel_basis @ |X|, log(clamp(.,1e-5))) and the log -> log10
// scale of network/vocoders/nsf_hifigan.py:86-91.
//
// One workgroup per frame: the windowed frame goes to LDS, an in-LDS radix-2 FFT (fp32, twiddles tabulated
// once per workgroup with sincospif) produces the spectrum, and the (sparse, triangular) mel filters are
// applied from the magnitudes still in LDS.  HBM traffic is the algorithmic minimum: 4 B per sample in,
// n_mels*4 B per frame out; the filterbank (0.5 MB) stays in L2.
#include <math.h>

#include <vector>

#include "common.h"

using namespace dsvc;

struct dsvc_melspec {
    dsvc_melspec_cfg cfg;
    int log2n = 0, n_bins = 0;
    void* basis = nullptr;      // [n_mels][n_bins] fp32
    void* range = nullptr;      // [n_mels][2] int: first / one-past-last non-zero bin
};

namespace {

// linear (may be null): the normalised linear spectrogram process_utterance(return_linear=True) returns beside the mel
// (preprocessing/data_gen_utils.py:144-149: audio.normalize(audio.amp_to_db(|X|)) = (20 log10(max(1e-5, |X|)) - min_level_db) / -min_level_db,
// utils/audio.py:51-56), [B][n_frames][n_fft / 2 + 1]
__global__ void k_melspec(const float* __restrict__ wav, float* __restrict__ mel, const float* __restrict__ basis,
                          const int* __restrict__ range, int n_samples, int n_frames, int n_fft, int log2n, int win,
                          int hop, int n_mels, float clip_val, int mode, float* __restrict__ linear, float min_level_db) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* buf = reinterpret_cast<float2*>(smem);                 // [n_fft] complex
    float2* tw = buf + n_fft;                                      // [n_fft/2] twiddles e^{-2 pi i k / n_fft}
    const int t = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, nt = blockDim.x;
    // mode 0: nvSTFT (reflect pad (n_fft - hop)/2, sqrt(. + 1e-9), natural log of the clamped mel scaled to log10)
    // mode 1: process_utterance (preprocessing/data_gen_utils.py:124-136: librosa.stft centred with ZERO padding n_fft/2, |X|, log10(max(eps, mel)))
    const int pad = mode == 1 ? n_fft / 2 : (n_fft - hop) / 2;
    const int woff = (n_fft - win) / 2;                            // torch.stft / librosa centre a short window in the frame
    const float* x = wav + (size_t)b * n_samples;
    for (int k = tid; k < n_fft / 2; k += nt) {
        float s, c;
        sincospif(-2.0f * (float)k / (float)n_fft, &s, &c);
        tw[k] = make_float2(c, s);
    }
    for (int n = tid; n < n_fft; n += nt) {
        int i = t * hop + n - pad;
        bool inside = true;
        if (mode == 1) {
            inside = i >= 0 && i < n_samples;                      // pad_mode='constant'
            if (!inside) i = 0;
        } else {
            if (i < 0) i = -i;                                     // F.pad(mode='reflect')
            if (i >= n_samples) i = 2 * (n_samples - 1) - i;
        }
        float w = 0.f;
        const int wn = n - woff;
        if (wn >= 0 && wn < win) w = 0.5f - 0.5f * cospif(2.0f * (float)wn / (float)win);   // periodic hann
        const int r = (int)(__brev((unsigned)n) >> (32 - log2n));  // bit-reversed order for the DIT butterflies
        buf[r] = make_float2(inside ? x[i] * w : 0.f, 0.f);
    }
    __syncthreads();
    for (int s = 1; s <= log2n; ++s) {
        const int half = 1 << (s - 1);
        const int tstride = n_fft >> s;
        for (int q = tid; q < n_fft / 2; q += nt) {
            const int grp = q / half, k = q - grp * half;
            const int i0 = grp * 2 * half + k, i1 = i0 + half;
            const float2 w = tw[k * tstride];
            const float2 a = buf[i0], c = buf[i1];
            const float2 m = make_float2(c.x * w.x - c.y * w.y, c.x * w.y + c.y * w.x);
            buf[i0] = make_float2(a.x + m.x, a.y + m.y);
            buf[i1] = make_float2(a.x - m.x, a.y - m.y);
        }
        __syncthreads();
    }
    const int n_bins = n_fft / 2 + 1;
    float* mag = reinterpret_cast<float*>(tw);                     // twiddles are dead now; n_bins <= n_fft floats
    for (int k = tid; k < n_bins; k += nt) {
        const float2 v = buf[k];
        mag[k] = sqrtf(v.x * v.x + v.y * v.y + (mode == 1 ? 0.f : 1e-9f));
        if (linear) linear[((size_t)b * n_frames + t) * n_bins + k] = (20.0f * log10f(fmaxf(1e-5f, mag[k])) - min_level_db) / -min_level_db;
    }
    __syncthreads();
    for (int m = tid; m < n_mels; m += nt) {
        const int lo = range[2 * m], hi = range[2 * m + 1];
        const float* bw = basis + (size_t)m * n_bins;
        float acc = 0.f;
        for (int k = lo; k < hi; ++k) acc = fmaf(bw[k], mag[k], acc);
        mel[((size_t)b * n_frames + t) * n_mels + m] = mode == 1 ? log10f(fmaxf(acc, clip_val)) : 0.434294f * logf(fmaxf(acc, clip_val));
    }
}

}  // namespace

extern "C" {

int dsvc_melspec_create(const dsvc_melspec_cfg* cfg, const float* mel_basis, dsvc_melspec** out) {
    if (!cfg || !mel_basis || !out) return fail(DSVC_EINVAL, "null argument");
    int l2 = 0;
    while ((1 << l2) < cfg->n_fft) ++l2;
    if ((1 << l2) != cfg->n_fft || cfg->n_fft < 64 || cfg->n_fft > 4096) return fail(DSVC_EINVAL, "melspec: n_fft must be a power of two in [64, 4096], got %d", cfg->n_fft);
    if (cfg->win_size > cfg->n_fft || cfg->win_size < 2 || cfg->hop < 1 || cfg->hop > cfg->n_fft || cfg->n_mels < 1)
        return fail(DSVC_EINVAL, "melspec: bad win/hop/n_mels");
    if (cfg->mode != 0 && cfg->mode != 1) return fail(DSVC_EINVAL, "melspec: mode must be 0 (nvSTFT) or 1 (centred, zero-padded)");
    int ndev = 0;
    DSVC_HIP(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(DSVC_EHIP, "no HIP device visible");
    dsvc_melspec* m = new dsvc_melspec();
    m->cfg = *cfg; m->log2n = l2; m->n_bins = cfg->n_fft / 2 + 1;
    std::vector<int> rg(2 * cfg->n_mels);
    for (int i = 0; i < cfg->n_mels; ++i) {
        int lo = m->n_bins, hi = 0;
        for (int k = 0; k < m->n_bins; ++k)
            if (mel_basis[(size_t)i * m->n_bins + k] != 0.f) { if (k < lo) lo = k; hi = k + 1; }
        if (hi == 0) lo = 0;
        rg[2 * i] = lo; rg[2 * i + 1] = hi;
    }
    const size_t bb = (size_t)cfg->n_mels * m->n_bins * 4;
    if (hipMalloc(&m->basis, bb) != hipSuccess || hipMalloc(&m->range, rg.size() * 4) != hipSuccess) {
        delete m;
        return fail(DSVC_ENOMEM, "melspec: hipMalloc failed");
    }
    DSVC_HIP(hipMemcpy(m->basis, mel_basis, bb, hipMemcpyHostToDevice));
    DSVC_HIP(hipMemcpy(m->range, rg.data(), rg.size() * 4, hipMemcpyHostToDevice));
    *out = m;
    return DSVC_OK;
}

void dsvc_melspec_destroy(dsvc_melspec* m) {
    if (!m) return;
    if (m->basis) (void)hipFree(m->basis);
    if (m->range) (void)hipFree(m->range);
    delete m;
}

int dsvc_melspec_frames(const dsvc_melspec* m, int64_t n_samples, int32_t* frames) {
    if (!m || !frames) return fail(DSVC_EINVAL, "null argument");
    if (m->cfg.mode == 1) {                                                  // librosa.stft(center=True): 1 + N // hop
        if (n_samples < 1) return fail(DSVC_EINVAL, "melspec: empty input");
        *frames = (int32_t)(1 + n_samples / m->cfg.hop);
        return DSVC_OK;
    }
    const int64_t pad = (m->cfg.n_fft - m->cfg.hop) / 2;
    if (n_samples <= pad) return fail(DSVC_EINVAL, "melspec: %lld samples is not enough for a reflect pad of %lld", (long long)n_samples, (long long)pad);
    const int64_t padded = n_samples + 2 * pad;
    *frames = (int32_t)((padded - m->cfg.n_fft) / m->cfg.hop + 1);       // torch.stft(center=False)
    return DSVC_OK;
}

static int melspec_launch(dsvc_melspec* m, const float* wav, float* mel, float* linear, float min_level_db, int32_t B, int64_t n_samples, void* stream);

int dsvc_melspec_run(dsvc_melspec* m, const float* wav, float* mel, int32_t B, int64_t n_samples, void* stream) {
    return melspec_launch(m, wav, mel, nullptr, -100.0f, B, n_samples, stream);
}

int dsvc_melspec_run_linear(dsvc_melspec* m, const float* wav, float* mel, float* linear, float min_level_db, int32_t B, int64_t n_samples, void* stream) {
    if (!linear) return fail(DSVC_EINVAL, "melspec: null output for the linear spectrogram");
    if (!(min_level_db < 0.f)) return fail(DSVC_EINVAL, "melspec: min_level_db must be negative (audio.normalize divides by it)");
    return melspec_launch(m, wav, mel, linear, min_level_db, B, n_samples, stream);
}

}  // extern "C"

static int melspec_launch(dsvc_melspec* m, const float* wav, float* mel, float* linear, float min_level_db, int32_t B, int64_t n_samples, void* stream) {
    if (!m || !wav || !mel || B < 1) return fail(DSVC_EINVAL, "bad argument");
    int32_t T = 0;
    DSVC_TRY(dsvc_melspec_frames(m, n_samples, &T));
    if (n_samples > 0x7fffffff) return fail(DSVC_EINVAL, "melspec: clip too long");
    const size_t smem = (size_t)m->cfg.n_fft * 8 + (size_t)m->cfg.n_fft * 4;
    hipLaunchKernelGGL(k_melspec, dim3(T, B), dim3(256), smem, (hipStream_t)stream, wav, mel, (const float*)m->basis,
                       (const int*)m->range, (int)n_samples, T, m->cfg.n_fft, m->log2n, m->cfg.win_size, m->cfg.hop,
                       m->cfg.n_mels, m->cfg.clip_val, m->cfg.mode, linear, min_level_db);
    DSVC_HIP(hipGetLastError());
    return DSVC_OK;
}
