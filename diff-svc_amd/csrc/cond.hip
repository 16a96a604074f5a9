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

}  // namespace

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
