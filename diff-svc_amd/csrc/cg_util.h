// cg_util.h -- what the translation units built on the conv_gemm engine share (train.hip, hubert.hip): a device buffer, the
// tiling dispatcher for split-fp16 (fp32-class) contractions, and the on-device packer of MFMA weight fragments.
#pragma once
#include "conv_gemm.h"

namespace dsvc {
namespace {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int alloc(size_t n) {
        if (n <= bytes && p) return DSVC_OK;
        release();
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) { p = nullptr; return fail(DSVC_ENOMEM, "hipMalloc(%zu) failed: %s", n, hipGetErrorString(e)); }
        bytes = n;
        return DSVC_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// split fp16 operands on both sides (NW = NA = 2: three MFMAs per product, fp32-class); tiling by problem shape
template <class Epi>
int launch(const ConvGemmArgs& a, const typename Epi::Args& e, hipStream_t st) {
    //                                                             WM WN WK KCB PF SPT NW NA
    // many rows: 2 x 2 accumulator tiles (two waves per SIMD under 256 VGPRs) and two row slabs per staged 128-row tile -- 13 % faster
    // over the training step than one wave per SIMD with 4 x 2 tiles (25.1 -> 21.8 ms, A/B in profiles/r2r_train_ab.txt)
    if (a.n_rows >= 6144 && a.cin % 64 == 0) return conv_gemm_launch<2, 4, 1, 64, 4, 3, 2, 2, Epi, 2>(a, e, st);
    if (a.cin % 128 == 0) return conv_gemm_launch<1, 1, 4, 128, 2, 3, 2, 2, Epi>(a, e, st);
    if (a.cin % 64 == 0) return conv_gemm_launch<1, 2, 1, 64, 4, 2, 2, 2, Epi>(a, e, st);
    return conv_gemm_launch<1, 2, 1, 16, 1, 2, 2, 2, Epi>(a, e, st);
}

// weights -> conv_gemm fragment layout [ctile][tap][k16][plane 2][lane][8] on the device.
//   W(col, tap, ci) = src[colmap(col) * s_col + ci * s_ci + tap_of(tap) * s_tap] * scale,  0 outside (cout, cin)
//   flip: tap_of(tap) = taps-1-tap (transposed conv).  colmap: optional packed-column -> source-column permutation (-1 = zero).
__global__ void k_pack_w(const float* __restrict__ src, const int* __restrict__ colmap, _Float16* __restrict__ dst, int n_ctiles,
                         int taps, int cin_pad, int cout, int cin, long long s_col, long long s_ci, long long s_tap, int flip, float scale) {
    const int nk16 = cin_pad >> 4;
    const long long total = (long long)n_ctiles * taps * nk16 * 512;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        long long r = idx;
        const int e = (int)(r & 7), l = (int)((r >> 3) & 63);
        r >>= 9;
        const int k = (int)(r % nk16); r /= nk16;
        const int tap = (int)(r % taps);
        const int ct = (int)(r / taps);
        int col = ct * 32 + (l & 31);
        const int ci = k * 16 + 8 * (l >> 5) + e;
        if (colmap) col = colmap[col];
        float w = 0.f;
        if (col >= 0 && col < cout && ci < cin) w = src[(long long)col * s_col + (long long)ci * s_ci + (long long)(flip ? taps - 1 - tap : tap) * s_tap] * scale;
        const _Float16 hi = (_Float16)w;
        _Float16* f = dst + ((((size_t)ct * taps + tap) * nk16 + k) * 2) * 512 + l * 8 + e;
        f[0] = hi;
        f[512] = (_Float16)(w - (float)hi);
    }
}

}  // namespace
}  // namespace dsvc
