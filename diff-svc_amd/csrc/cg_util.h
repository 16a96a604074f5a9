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
struct PackDesc {
    const float* src; const int* colmap; _Float16* dst;
    int n_ctiles, taps, cin_pad, cout, cin;
    long long s_col, s_ci, s_tap;
    int flip; float scale;
};
__device__ __forceinline__ void pack_w_body(const PackDesc& d, long long first, long long step) {
    const int nk16 = d.cin_pad >> 4;
    const long long total = (long long)d.n_ctiles * d.taps * nk16 * 512;
    for (long long idx = first; idx < total; idx += step) {
        long long r = idx;
        const int e = (int)(r & 7), l = (int)((r >> 3) & 63);
        r >>= 9;
        const int k = (int)(r % nk16); r /= nk16;
        const int tap = (int)(r % d.taps);
        const int ct = (int)(r / d.taps);
        int col = ct * 32 + (l & 31);
        const int ci = k * 16 + 8 * (l >> 5) + e;
        if (d.colmap) col = d.colmap[col];
        float w = 0.f;
        if (col >= 0 && col < d.cout && ci < d.cin)
            w = d.src[(long long)col * d.s_col + (long long)ci * d.s_ci + (long long)(d.flip ? d.taps - 1 - tap : tap) * d.s_tap] * d.scale;
        const _Float16 hi = (_Float16)w;
        _Float16* f = d.dst + ((((size_t)ct * d.taps + tap) * nk16 + k) * 2) * 512 + l * 8 + e;
        f[0] = hi;
        f[512] = (_Float16)(w - (float)hi);
    }
}
__global__ void k_pack_w(const float* __restrict__ src, const int* __restrict__ colmap, _Float16* __restrict__ dst, int n_ctiles,
                         int taps, int cin_pad, int cout, int cin, long long s_col, long long s_ci, long long s_tap, int flip, float scale) {
    const PackDesc d{src, colmap, dst, n_ctiles, taps, cin_pad, cout, cin, s_col, s_ci, s_tap, flip, scale};
    pack_w_body(d, (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x);
}
// many weight tensors in one launch (a training step re-packs ~65 of them: 5 us each as separate launches): blockIdx.y = descriptor
__global__ void k_pack_w_batch(const PackDesc* __restrict__ descs) {
    pack_w_body(descs[blockIdx.y], (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x);
}

}  // namespace
}  // namespace dsvc
