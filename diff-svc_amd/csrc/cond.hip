// Pitch index work of the condition builder on the device (include/dsvc.h: dsvc_pitch_coarse).
// Reference: modules/fastspeech/fs2.py:229-237 (add_pitch: f0_denorm = denorm_f0(f0, uv, pitch_padding), pitch = f0_to_coarse(f0_denorm))
// with utils/pitch_utils.py:17-31 (f0_to_coarse) and :63-76 (denorm_f0, pitch_norm 'log').
//
// f0_to_coarse is a non-decreasing step function of the normalised pitch x = log2(f0): coarse(x) = 1 + #{k : x >= thr[k]}.  The
// thresholds are found ON THE HOST by bisection over the fp32 line with the reference's own torch-CPU expression (cond.py), so the
// device needs no log / pow of its own for the bin and agrees with that expression at every fp32 input -- index work stays exact, and
// the [B, T] device -> host -> device round trip of the host path is gone.
#include <math.h>

#include "../../include/dsvc.h"
#include "common.h"

using namespace dsvc;

namespace {

__global__ void k_pitch_coarse(const float* __restrict__ f0, const long long* __restrict__ mel2ph, const float* __restrict__ uv,
                               const float* __restrict__ thr, int n_thr, long long n, float* __restrict__ f0_denorm, long long* __restrict__ coarse) {
    extern __shared__ float sthr[];
    for (int i = threadIdx.x; i < n_thr; i += blockDim.x) sthr[i] = thr[i];
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float x = f0[i];
        const bool off = mel2ph[i] == 0 || (uv && uv[i] > 0.f);
        int lo = 0, hi = n_thr;                         // number of thresholds <= x
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (x >= sthr[mid]) lo = mid + 1; else hi = mid;
        }
        f0_denorm[i] = off ? 0.f : exp2f(x);
        coarse[i] = off ? 1 : 1 + lo;                   // f0_to_coarse(0) == 1
    }
}

// K13 in ONE launch (fs2.py:133-148 the no_fs2 branch: decoder_inp = gather(pad(hubert, one zero row in front), mel2ph);
// :229-237 add_pitch; insert4: (decoder_inp + pitch_embed[coarse]) * (mel2ph > 0)): one workgroup = 32 frames of one clip.
//   phase 1: per frame f0_denorm / coarse bin (threshold search, as k_pitch_coarse) and the reference's in-place f0[mel2ph == 0] = 0
//   phase 2: row t of decoder_inp [B, T, H] = (hubert[b, mel2ph - 1, :] + emb[coarse, :]) * nonpad  -- coalesced along H, the same
//            two fp32 operations in the same order as the reference, so the result is bit-identical (a padded frame gives
//            (0 + emb[1]) * 0, signed zeros included)
//   phase 3: the [B, H, T] transpose the denoiser seam takes (diffusion.py:236) from an LDS tile, coalesced along T
__global__ void __launch_bounds__(256) k_cond_build(const float* __restrict__ hubert, const long long* __restrict__ mel2ph, float* __restrict__ f0,
                                                    const float* __restrict__ uv, const float* __restrict__ thr, int n_thr,
                                                    const float* __restrict__ emb, int N, int T, int H, float* __restrict__ dec,
                                                    float* __restrict__ cond_bht, float* __restrict__ f0_denorm, long long* __restrict__ coarse) {
    extern __shared__ float sm[];
    float* sthr = sm;                                   // [n_thr]
    int* srow = reinterpret_cast<int*>(sm + n_thr);     // [32] source unit row (-1 = the zero pad row), [32] embedding row, [32] nonpad
    float* tile = sm + n_thr + 96;                      // [32][H + 1]
    const int b = blockIdx.y, t0 = blockIdx.x * 32;
    for (int i = threadIdx.x; i < n_thr; i += blockDim.x) sthr[i] = thr[i];
    __syncthreads();
    if (threadIdx.x < 32) {
        const int t = t0 + threadIdx.x;
        if (t < T) {
            const size_t i = (size_t)b * T + t;
            const long long m = mel2ph[i];
            const float x = f0[i];
            const bool off = m == 0 || (uv && uv[i] > 0.f);
            int lo = 0, hi = n_thr;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (x >= sthr[mid]) lo = mid + 1; else hi = mid;
            }
            const int c = off ? 1 : 1 + lo;
            f0_denorm[i] = off ? 0.f : exp2f(x);
            coarse[i] = c;
            if (m == 0) f0[i] = 0.f;
            srow[threadIdx.x] = (m >= 1 && m <= N) ? (int)(m - 1) : -1;
            srow[32 + threadIdx.x] = c;
            srow[64 + threadIdx.x] = m > 0 ? 1 : 0;
        }
    }
    __syncthreads();
    const int nt = T - t0 < 32 ? T - t0 : 32;
    for (int e = threadIdx.x; e < nt * H; e += blockDim.x) {
        const int r = e / H, h = e - r * H;
        const int u = srow[r];
        const float g = u >= 0 ? hubert[((size_t)b * N + u) * H + h] : 0.f;
        const float v = (g + emb[(size_t)srow[32 + r] * H + h]) * (srow[64 + r] ? 1.0f : 0.0f);
        dec[((size_t)b * T + t0 + r) * H + h] = v;
        tile[r * (H + 1) + h] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < H * 32; e += blockDim.x) {
        const int h = e >> 5, r = e & 31;
        if (r < nt) cond_bht[((size_t)b * H + h) * T + t0 + r] = tile[r * (H + 1) + h];
    }
}

}  // namespace

extern "C" int dsvc_cond_build(const float* hubert, const int64_t* mel2ph, float* f0_log2, const float* uv, const float* thresholds,
                               int32_t n_thresholds, const float* pitch_embed, int32_t B, int32_t N, int32_t T, int32_t H,
                               float* decoder_inp, float* cond_bht, float* f0_denorm, int64_t* coarse, void* stream) {
    if (!hubert || !mel2ph || !f0_log2 || !thresholds || !pitch_embed || !decoder_inp || !cond_bht || !f0_denorm || !coarse)
        return fail(DSVC_EINVAL, "null argument");
    if (n_thresholds < 1 || n_thresholds > 4096) return fail(DSVC_EINVAL, "cond_build: %d thresholds", n_thresholds);
    if (B < 1 || N < 1 || T < 1 || H < 1 || H > 4096) return fail(DSVC_EINVAL, "cond_build: bad shape B=%d N=%d T=%d H=%d", B, N, T, H);
    const size_t smem = ((size_t)n_thresholds + 96 + (size_t)32 * (H + 1)) * 4;
    if (smem > 64 * 1024) return fail(DSVC_EINVAL, "cond_build: hidden size %d needs %zu B of LDS", H, smem);
    hipLaunchKernelGGL(k_cond_build, dim3((unsigned)ceil_div(T, 32), (unsigned)B), dim3(256), smem, (hipStream_t)stream, hubert,
                       (const long long*)mel2ph, f0_log2, uv, thresholds, n_thresholds, pitch_embed, N, T, H, decoder_inp, cond_bht, f0_denorm,
                       (long long*)coarse);
    DSVC_HIP(hipGetLastError());
    return DSVC_OK;
}

extern "C" int dsvc_pitch_coarse(const float* f0_log2, const int64_t* mel2ph, const float* uv, const float* thresholds, int32_t n_thresholds,
                                 int64_t n, float* f0_denorm, int64_t* coarse, void* stream) {
    if (!f0_log2 || !mel2ph || !thresholds || !f0_denorm || !coarse) return fail(DSVC_EINVAL, "null argument");
    if (n_thresholds < 1 || n_thresholds > 4096) return fail(DSVC_EINVAL, "pitch_coarse: %d thresholds", n_thresholds);
    if (n < 0) return fail(DSVC_EINVAL, "pitch_coarse: n %lld", (long long)n);
    if (n == 0) return DSVC_OK;
    const long long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_pitch_coarse, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), (size_t)n_thresholds * 4, (hipStream_t)stream, f0_log2,
                       (const long long*)mel2ph, uv, thresholds, n_thresholds, (long long)n, f0_denorm, (long long*)coarse);
    DSVC_HIP(hipGetLastError());
    return DSVC_OK;
}
