// NSF-HiFiGAN generator behind the C ABI (include/dsvc.h).
// Reference: modules/nsf_hifigan/models.py:14-30 (load_model), :33-64 (ResBlock1), :148-323 (SineGen,
// SourceModuleHnNSF), :325-387 (Generator); called through network/vocoders/nsf_hifigan.py:47-73.
//
// Layout: every activation is frame-major fp32 [rows][channels]; row = clip*clip_stride_s + n at the stage's
// sample rate.  All dense convs go through conv_gemm (MFMA, split-fp16 operands):
//   * Conv1d(k, dilation d)          -> taps = k, dil = d
//   * ConvTranspose1d(k, stride u)   -> polyphase: a 3-tap conv over the INPUT rows producing u*cout columns;
//                                       column phi*cout+co of input row q is output sample q*u+phi, which is
//                                       exactly the frame-major address of the upsampled signal.
//   * leaky_relu in front of a conv  -> applied while the tile is staged into LDS (ConvGemmArgs::in_slope)
//   * "+ x", MRF mean, "+ x_source"  -> epilogue (EpiAffine: out = alpha*(acc + b + res) + beta*out)
#include <math.h>

#include <map>
#include <string>
#include <type_traits>
#include <vector>

#include "conv_gemm.h"
#include "tepi_util.h"

using namespace dsvc;

typedef float f32x4v __attribute__((ext_vector_type(4)));

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

int upload(DevBuf& b, const void* host, size_t bytes) {
    DSVC_TRY(b.alloc(bytes));
    DSVC_HIP(hipMemcpy(b.p, host, bytes, hipMemcpyHostToDevice));
    return DSVC_OK;
}

// ---- epilogues ----
struct EpiAffine {
    static constexpr bool PAIRED = false;
    struct Args {
        float* out; int ld;
        const float* bias; int bias_mod;     // bias index = col % bias_mod
        int cout;                            // valid columns
        const float* res; int ldres;         // optional residual
        float alpha;
        int accumulate;                      // out = alpha*(..) + out
    };
    __device__ __forceinline__ void one(const Args& e, int row, int col, float v) const {
        if (col >= e.cout) return;
        v += e.bias[col % e.bias_mod];
        if (e.res) v += e.res[(size_t)row * e.ldres + col];
        v *= e.alpha;
        float* p = e.out + (size_t)row * e.ld + col;
        if (e.accumulate) v += *p;
        *p = v;
    }
};

// conv_post + tanh (models.py:383-385): column 0 only, written to the caller's [B][clip_len] waveform
struct EpiTanhWav {
    static constexpr bool PAIRED = false;
    struct Args { float* wav; const float* bias; int clip_stride, clip_len; };
    __device__ __forceinline__ void one(const Args& e, int row, int col, float v) const {
        if (col != 0) return;
        const int clip = row / e.clip_stride;
        const int n = row - clip * e.clip_stride;
        if (n < e.clip_len) e.wav[(size_t)clip * e.clip_len + n] = tanhf(v + e.bias[0]);
    }
};

template <class Epi, int NW, int NA>
int voc_tiling(const ConvGemmArgs& a, const typename Epi::Args& e, hipStream_t st) {
    if (a.cin % 64 == 0) {
        if constexpr (NW == 2 && NA == 2 && std::is_same<Epi, EpiAffine>::value) {
            // The shipped (fp32-class) MRF convs: two waves per SIMD (<= 256 VGPRs: 2 x 2 accumulator tiles per wave) and several
            // row slabs per staged tile beat the one-wave-per-SIMD 128-row tiles above by 10-40 % per stage (profiles/r2k_ab.txt,
            // r2l_ab.txt: 221 -> 175 ms per 32 clips, 9.1 -> 7.7 ms per clip).            WM WN WK KCB PF SPT        row slabs
            if (a.n_ctiles >= 8) {
                if (a.n_rows < 65536) return conv_gemm_launch<1, 4, 1, 64, 4, 2, NW, NA, Epi, 1>(a, e, st);   // few rows: 32-row tiles fill the chip
                return conv_gemm_launch<2, 4, 1, 32, 2, 3, NW, NA, Epi, 2>(a, e, st);
            }
            if (a.n_ctiles >= 4) return conv_gemm_launch<2, 2, 1, 32, 2, 3, NW, NA, Epi, 4>(a, e, st);
            return conv_gemm_launch<2, 1, 1, 32, 2, 3, NW, NA, Epi, 8>(a, e, st);
        }
        //                                                 WM WN WK KCB PF SPT
        if (a.n_ctiles >= 8) return conv_gemm_launch<4, 4, 1, 64, 4, 6, NW, NA, Epi>(a, e, st);
        if (a.n_ctiles >= 4) return conv_gemm_launch<4, 2, 1, 64, 4, 6, NW, NA, Epi>(a, e, st);
        return conv_gemm_launch<4, 1, 1, 64, 4, 8, NW, NA, Epi>(a, e, st);
    }
    if (a.cin % 32 == 0) return conv_gemm_launch<4, 1, 1, 32, 2, 8, NW, NA, Epi>(a, e, st);
    return conv_gemm_launch<4, 1, 1, 16, 1, 8, NW, NA, Epi>(a, e, st);
}

template <class Epi>
int voc_dispatch(const ConvGemmArgs& a, const typename Epi::Args& e, int prec, hipStream_t st) {
    switch (prec) {
        case DSVC_PREC_F16: return voc_tiling<Epi, 1, 1>(a, e, st);
        case DSVC_PREC_F16_W2: return voc_tiling<Epi, 2, 1>(a, e, st);
        default: return voc_tiling<Epi, 2, 2>(a, e, st);
    }
}

struct PackedConv {
    DevBuf w, bias;
    int n_ctiles = 0, taps = 1, cin = 0, cout = 0, dil = 1;
};

template <class FW>
int pack_conv(PackedConv& pc, int cout, int taps, int cin, int dil, FW&& src, const float* bias, int nbias) {
    pc.n_ctiles = round_up(ceil_div(cout, 32), 2);
    pc.taps = taps; pc.cin = cin; pc.cout = cout; pc.dil = dil;
    std::vector<_Float16> h(packed_halfs(pc.n_ctiles, taps, cin, 2));
    pack_fragments(h.data(), pc.n_ctiles, taps, cin, 2, [&](int col, int tap, int ci) { return col < cout ? src(col, tap, ci) : 0.f; });
    DSVC_TRY(upload(pc.w, h.data(), h.size() * sizeof(_Float16)));
    return upload(pc.bias, bias, (size_t)nbias * sizeof(float));
}

// ---- fused ResBlock1 pair for the narrow stages (C = 16 / 32 channels at 256x / 512x the frame rate) ----
//   out = alpha * ( x + b2 + conv_k,1( lrelu( b1 + conv_k,d( lrelu(x) ) ) ) )  [+ out]        (models.py:57-64)
// These stages are HBM-bound by nature (64-128 B per frame row, a few hundred FMAs per output), but as two passes of the
// generic MFMA engine each conv was a read + write of the whole fp32 stage through 64-column tiles of which 16 or 32 are
// real: measured 0.7 TB/s, 3.9 ms per conv at 32 clips.  Here one workgroup stages a (TN + 2H) x C tile of lrelu(x) in LDS,
// computes the intermediate activation for TN + 2*(k/2) rows into LDS, and the second conv + residual from there: one read
// and one write of the stage per PAIR.  Arithmetic is plain fp32 FMA (exact products, like the reference); the weights are
// wave-uniform and come through the scalar cache ([tap][ci][co] layout: 16 / 32 consecutive co per scalar load).
template <int C, int TN>
__global__ void __launch_bounds__(TN) k_resblock_pair(const float* __restrict__ x, float* __restrict__ out, const float* __restrict__ w1,
                                                      const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
                                                      int k, int d, int n_rows, int stride, int len, float alpha, int accumulate) {
    constexpr int XS = C + 4;                              // row stride in floats: 16-B aligned, conflict-free ds_read_b128 across rows
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int h2 = k >> 1, h1 = h2 * d, H = h1 + h2;
    float* xs = reinterpret_cast<float*>(smem);            // [TN + 2H][XS]   lrelu(x), zero outside the clip
    float* ts = xs + (size_t)(TN + 2 * H) * XS;            // [TN + 2h2][XS]  lrelu(conv1), zero outside the clip
    const int tid = threadIdx.x;
    const long long tile0 = (long long)blockIdx.x * TN;
    auto valid = [&](long long g) { return g >= 0 && g < n_rows && (int)(g % stride) < len; };
    // ---- stage lrelu(x) ----
    {
        constexpr int Q = C / 4;
        const int total = (TN + 2 * H) * Q;
        for (int i = tid; i < total; i += TN) {
            const int r = i / Q, q = i - r * Q;
            const long long g = tile0 - H + r;
            f32x4v v = {0.f, 0.f, 0.f, 0.f};
            if (valid(g)) {
                v = *reinterpret_cast<const f32x4v*>(x + (size_t)g * C + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.1f * v[e];
            }
            *reinterpret_cast<f32x4v*>(xs + (size_t)r * XS + 4 * q) = v;
        }
    }
    __syncthreads();
    // ---- conv1 (dilation d) -> ts ----
    for (int rr = tid; rr < TN + 2 * h2; rr += TN) {
        const long long g = tile0 - h2 + rr;
        float acc[C];
#pragma unroll
        for (int co = 0; co < C; ++co) acc[co] = b1[co];
        if (valid(g)) {
            for (int tap = 0; tap < k; ++tap) {
                const float* xr = xs + (size_t)(rr + tap * d) * XS;            // x row g + (tap - k/2)*d
                const float* wt = w1 + (size_t)tap * C * C;
#pragma unroll
                for (int q = 0; q < C / 4; ++q) {
                    const f32x4v xv = *reinterpret_cast<const f32x4v*>(xr + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float* wr = wt + (size_t)(4 * q + e) * C;
#pragma unroll
                        for (int co = 0; co < C; ++co) acc[co] = fmaf(wr[co], xv[e], acc[co]);
                    }
                }
            }
#pragma unroll
            for (int co = 0; co < C; ++co) acc[co] = acc[co] > 0.f ? acc[co] : 0.1f * acc[co];
        } else {
#pragma unroll
            for (int co = 0; co < C; ++co) acc[co] = 0.f;
        }
#pragma unroll
        for (int q = 0; q < C / 4; ++q)
            *reinterpret_cast<f32x4v*>(ts + (size_t)rr * XS + 4 * q) = f32x4v{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
    }
    __syncthreads();
    // ---- conv2 (dilation 1) + residual ----
    {
        const long long g = tile0 + tid;
        if (g >= n_rows) return;
        float acc[C];
        const bool ok = valid(g);
        if (ok) {
#pragma unroll
            for (int co = 0; co < C; ++co) acc[co] = b2[co];
            for (int tap = 0; tap < k; ++tap) {
                const float* tr = ts + (size_t)(tid + tap) * XS;                // t row g + tap - k/2
                const float* wt = w2 + (size_t)tap * C * C;
#pragma unroll
                for (int q = 0; q < C / 4; ++q) {
                    const f32x4v tv = *reinterpret_cast<const f32x4v*>(tr + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float* wr = wt + (size_t)(4 * q + e) * C;
#pragma unroll
                        for (int co = 0; co < C; ++co) acc[co] = fmaf(wr[co], tv[e], acc[co]);
                    }
                }
            }
        }
        float* po = out + (size_t)g * C;
        const float* px = x + (size_t)g * C;
#pragma unroll
        for (int q = 0; q < C / 4; ++q) {
            f32x4v o = {0.f, 0.f, 0.f, 0.f};
            if (ok) {
                const f32x4v xv = *reinterpret_cast<const f32x4v*>(px + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = alpha * (acc[4 * q + e] + xv[e]);
                if (accumulate) {
                    const f32x4v pv = *reinterpret_cast<const f32x4v*>(po + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += pv[e];
                }
            }
            *reinterpret_cast<f32x4v*>(po + 4 * q) = o;
        }
    }
}

template <int C, int TN>
int resblock_pair_launch(const float* x, float* out, const float* w1, const float* b1, const float* w2, const float* b2, int k, int d,
                         int n_rows, int stride, int len, float alpha, int accumulate, hipStream_t st) {
    const int h2 = k / 2, H = h2 * d + h2;
    const size_t smem = ((size_t)(TN + 2 * H) + (size_t)(TN + 2 * h2)) * (C + 4) * 4;
    if (smem > 64 * 1024) return fail(DSVC_EINVAL, "resblock pair: %zu B of LDS", smem);
    hipLaunchKernelGGL((k_resblock_pair<C, TN>), dim3(ceil_div(n_rows, TN)), dim3(TN), smem, st, x, out, w1, b1, w2, b2, k, d, n_rows, stride, len,
                       alpha, accumulate);
    DSVC_HIP(hipGetLastError());
    return DSVC_OK;
}

// ---- the same fused pair on the matrix cores (C = 16 / 32 / 64) ----
// One workgroup of W waves owns 32*(2W-1) output rows.  lrelu(x) for those rows plus both convs' halos is staged ONCE as split fp16
// (hi + lo planes, row stride C+8 halfs: conflict-free ds_read_b128).  Both convs run as the TRANSPOSED product (weights as the A
// operand), so a lane ends up holding 4 consecutive channels of ONE row: conv1 parks lrelu(b1 + .) back into LDS with 8-byte writes
// -- over the x tile, which is dead by then -- and conv2's epilogue (bias + residual + MRF scale / accumulate) moves 16 bytes per
// access.  Products are the three-MFMA split (x_hi w_hi + x_hi w_lo + x_lo w_hi: fp32-class, as the rest of the shipped vocoder);
// each wave keeps 2 row tiles, so a 1 KiB weight fragment (streamed from L2 in conv_gemm's fragment order through a 4-deep register
// ring) feeds 6 MFMAs.
// Measured at 32 clips (profiles/r2o_pair_ablate.txt, DSVC_PAIR_DBG): C = 64 / 32 / 16: 1.99 / 1.12 / 1.01 ms per pair, of which the
// two MFMA loops are 1.1 / 0.44 / 0.42 ms = 78-90 % of the matrix rate the chip sustains on real data, staging + LDS park
// 0.49 / 0.30 / 0.32 and the epilogue 0.3-0.4: the phases of a workgroup do not overlap, the rest is occupancy (LDS: 3-5 per CU).
template <int C, int W>
__global__ void __launch_bounds__(64 * W, 2)
k_pair_mfma(const float* __restrict__ x, float* __restrict__ out, const _Float16* __restrict__ w1, const float* __restrict__ b1,
            const _Float16* __restrict__ w2, const float* __restrict__ b2, int k, int d, int n_rows, int stride, int len, float alpha, int accumulate, int dbg) {
    constexpr int MT = 2;                      // row tiles per wave
    constexpr int MM = W * MT;                 // intermediate row tiles (32 rows each)
    constexpr int MO = MM - 1;                 // output row tiles
    constexpr int NT = (C + 31) / 32;          // 32-column tiles
    constexpr int KS = C / 16;                 // k16 steps per tap
    constexpr int XS = C + 8;                  // LDS row stride in halfs
    constexpr int RD = 4;                      // weight ring depth (steps)
    constexpr int NTH = 64 * W;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* lds = reinterpret_cast<_Float16*>(smem);
    const int h2 = k >> 1, h1 = h2 * d;
    const int XR = 32 * MM + 2 * h1;           // staged x rows: row r <-> global row tile0 - h2 - h1 + r
    const int xplane = XR * XS;                // halfs per x plane (hi, then lo)
    constexpr int mplane = 32 * MM * XS;       // halfs per plane of the intermediate (aliases the x planes)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile0 = blockIdx.x * (32 * MO);            // (n_rows < 2^31 - a tile: checked by the launcher)
    // rows of a tile are consecutive: one division per tile, then offsets with a wrap
    auto valid = [&](int g) {
        if (g < 0 || g >= n_rows) return false;
        return (g - (g / stride) * stride) < len;
    };
    const int nk16 = KS;
    const size_t tile_halfs = (size_t)k * nk16 * 2 * 512;      // one 32-column tile of a packed conv: [tap][k16][plane 2][lane][8]
    // step s = tap * KS + ks; fragment (n, plane) of step s
    auto wload = [&](const _Float16* w, half8 (&dst)[NT][2], int s) {
        const _Float16* p = w + (size_t)s * (2 * 512) + lane * 8;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            dst[n][0] = *reinterpret_cast<const half8*>(p + n * tile_halfs);
            dst[n][1] = *reinterpret_cast<const half8*>(p + n * tile_halfs + 512);
        }
    };
    const int S = k * KS;
    half8 ring[RD][NT][2];
#pragma unroll
    for (int u = 0; u < RD; ++u)
        if (u < S) wload(w1, ring[u], u);
    // ---- stage lrelu(x) as fp16 hi | lo ----
    {
        constexpr int IPR = C / 8;
        const int items = XR * IPR;
        for (int i = tid; i < items; i += NTH) {
            const int r = i / IPR, c8 = (i - r * IPR) * 8;
            const int g = tile0 - h2 - h1 + r;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (valid(g)) {
                const f32x4v a = *reinterpret_cast<const f32x4v*>(x + (size_t)g * C + c8);
                const f32x4v b = *reinterpret_cast<const f32x4v*>(x + (size_t)g * C + c8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = a[e] > 0.f ? a[e] : 0.1f * a[e]; v[4 + e] = b[e] > 0.f ? b[e] : 0.1f * b[e]; }
            }
            half8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) { hi[e] = (_Float16)v[e]; lo[e] = (_Float16)(v[e] - (float)hi[e]); }
            *reinterpret_cast<half8*>(lds + r * XS + c8) = hi;
            *reinterpret_cast<half8*>(lds + xplane + r * XS + c8) = lo;
        }
    }
    __syncthreads();
    const int arow = lane & 31, acol = 8 * (lane >> 5);
    // ---- conv1 (dilation d), transposed: acc1[m][n][r] = mid[row 32*(wave*MT+m) + (lane&31)][channel 32n + (r&3) + 8(r>>2) + 4(lane>>5)] ----
    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    for (int s0 = 0; s0 < ((dbg & 1) ? 0 : S); s0 += RD) {
#pragma unroll
        for (int u = 0; u < RD; ++u) {
            const int s = s0 + u;
            if (s < S) {
                const int tap = s / KS, ks = s - tap * KS;
                half8 wf[NT][2];
#pragma unroll
                for (int n = 0; n < NT; ++n) { wf[n][0] = ring[u][n][0]; wf[n][1] = ring[u][n][1]; }
                if (s + RD < S) wload(w1, ring[u], s + RD);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const _Float16* xr = lds + (32 * (wave * MT + m) + arow + tap * d) * XS + ks * 16 + acol;
                    const half8 xh = *reinterpret_cast<const half8*>(xr);
                    const half8 xl = *reinterpret_cast<const half8*>(xr + xplane);
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[n][0], xh, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[n][1], xh, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[n][0], xl, acc[m][n], 0, 0, 0);
                    }
                }
            }
        }
    }
    // conv2's first weight fragments fly while the intermediate is parked
#pragma unroll
    for (int u = 0; u < RD; ++u)
        if (u < S) wload(w2, ring[u], u);
    __syncthreads();                                   // every wave is done reading the x tile
    // ---- lrelu(b1 + conv1) -> LDS (hi | lo), zero on rows outside the clip (conv2's zero padding) ----
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int j = 32 * (wave * MT + m) + arow;
        const bool ok = valid(tile0 - h2 + j);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ch = 32 * n + 8 * q + 4 * (lane >> 5);
                if (ch < C) {
                    const f32x4v bb = *reinterpret_cast<const f32x4v*>(b1 + ch);
                    half4 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[m][n][4 * q + e] + bb[e];
                        v = v > 0.f ? v : 0.1f * v;
                        if (!ok) v = 0.f;
                        hi[e] = (_Float16)v;
                        lo[e] = (_Float16)(v - (float)hi[e]);
                    }
                    *reinterpret_cast<half4*>(lds + j * XS + ch) = hi;
                    *reinterpret_cast<half4*>(lds + mplane + j * XS + ch) = lo;
                }
            }
    }
    __syncthreads();
    // ---- conv2 (dilation 1), transposed as well: acc[m][n][r] = out[row 32*(wave*MT+m) + (lane&31)][channel 32n + (r&3) + 8(r>>2) + 4(lane>>5)] ----
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    for (int s0 = 0; s0 < ((dbg & 1) ? 0 : S); s0 += RD) {
#pragma unroll
        for (int u = 0; u < RD; ++u) {
            const int s = s0 + u;
            if (s < S) {
                const int tap = s / KS, ks = s - tap * KS;
                half8 wf[NT][2];
#pragma unroll
                for (int n = 0; n < NT; ++n) { wf[n][0] = ring[u][n][0]; wf[n][1] = ring[u][n][1]; }
                if (s + RD < S) wload(w2, ring[u], s + RD);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    if (wave * MT + m < MO) {
                        const _Float16* tr = lds + (32 * (wave * MT + m) + arow + tap) * XS + ks * 16 + acol;
                        const half8 th = *reinterpret_cast<const half8*>(tr);
                        const half8 tl = *reinterpret_cast<const half8*>(tr + mplane);
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[n][0], th, acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[n][1], th, acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[n][0], tl, acc[m][n], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    // ---- out = alpha * (x + b2 + conv2) [+ out], zero on gap rows: a lane owns 4 consecutive channels of its row (16-byte accesses) ----
    if (dbg & 2) return;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (wave * MT + m >= MO) continue;
        const int g = tile0 + 32 * (wave * MT + m) + arow;
        if (g >= n_rows) continue;
        const bool ok = (g - (g / stride) * stride) < len;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ch = 32 * n + 8 * q + 4 * (lane >> 5);
                if (ch < C) {
                    f32x4v o = {0.f, 0.f, 0.f, 0.f};
                    float* po = out + (size_t)g * C + ch;
                    if (ok) {
                        const f32x4v bb = *reinterpret_cast<const f32x4v*>(b2 + ch);
                        const f32x4v xv = *reinterpret_cast<const f32x4v*>(x + (size_t)g * C + ch);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = alpha * (acc[m][n][4 * q + e] + bb[e] + xv[e]);
                        if (accumulate) {
                            const f32x4v pv = *reinterpret_cast<const f32x4v*>(po);
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] += pv[e];
                        }
                    }
                    *reinterpret_cast<f32x4v*>(po) = o;
                }
            }
    }
}

template <int C, int W>
int pair_mfma_launch(const float* x, float* out, const _Float16* w1, const float* b1, const _Float16* w2, const float* b2, int k, int d,
                     int n_rows, int stride, int len, float alpha, int accumulate, hipStream_t st) {
    const int h1 = (k / 2) * d;
    const size_t smem = (size_t)2 * (32 * 2 * W + 2 * h1) * (C + 8) * sizeof(_Float16);
    if (smem > 64 * 1024) return fail(DSVC_EINVAL, "resblock pair (mfma): %zu B of LDS", smem);
    if (n_rows > 0x7fffff00 - 64 * W) return fail(DSVC_EINVAL, "resblock pair (mfma): %d rows", n_rows);
#ifdef DSVC_PROFILING
    static const int dbg = getenv("DSVC_PAIR_DBG") ? atoi(getenv("DSVC_PAIR_DBG")) : 0;       // phase ablation (timing only): 1 = no MFMA loops, 2 = no epilogue
#else
    constexpr int dbg = 0;
#endif
    hipLaunchKernelGGL((k_pair_mfma<C, W>), dim3(ceil_div(n_rows, 32 * (2 * W - 1))), dim3(64 * W), smem, st, x, out, w1, b1, w2, b2, k, d, n_rows,
                       stride, len, alpha, accumulate, dbg);
    DSVC_HIP(hipGetLastError());
    return DSVC_OK;
}

// ---- small kernels ----

// mel [B][T][M] (log10) -> frame-major [B*stride][M] natural log (nsf_hifigan.py:63-65: c = 2.30259 * mel)
__global__ void k_prep_mel(const float* __restrict__ mel, float* __restrict__ dst, int B, int T, int M, int stride, float scale) {
    const size_t n = (size_t)B * T * M;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i % M);
        const size_t bt = i / M;
        const int t = (int)(bt % T), b = (int)(bt / T);
        dst[((size_t)b * stride + t) * M + m] = scale * mel[i];
    }
}

// ---- harmonic source (SineGen._f02sine, models.py:183-213) in closed form ----
// torch's CPU cumsum accumulates fp32 inputs in double, and f0 is piecewise constant (nearest upsampling), so
// the running sum inside frame f is  base_f + (j+1)*r_f.  The "-1 at every wrap" shift (models.py:205-209)
// fires exactly when floor(fp32(cumsum)) increases, so the number of shifts up to a sample is a difference
// of floors, and the fp32 rounding of (rad - 1) adds delta_f per shift.  A tiny sequential pass per
// (clip, harmonic) over the T frames produces the per-frame constants; the per-sample pass is then parallel.
struct SrcFrame { double base; double flprev; double delta_acc; };

__global__ void k_source_frames(const float* __restrict__ f0, SrcFrame* __restrict__ fr, float* __restrict__ fl00,
                                int T, int hop, int dim, float sr, unsigned long long seed, int clip0, const int* __restrict__ clip_ids) {
    const int b = blockIdx.x, h = threadIdx.x;
    if (h >= dim) return;
    const unsigned cid = (unsigned)(clip_ids ? clip_ids[b] : clip0 + b);
    float ini = 0.f;
    if (h > 0) {
        const u32x4 r = philox4x32((unsigned)(h >> 2), 0u, cid, PURPOSE_SINE_PHASE, seed);
        const unsigned w = (h & 3) == 0 ? r.x : (h & 3) == 1 ? r.y : (h & 3) == 2 ? r.z : r.w;
        ini = u01_open_high(w);
    }
    const float mult = (float)(h + 1);
    double base = 0.0, delta_acc = 0.0, flprev = 0.0;
    // The chain over the frames is sequential (about ten dependent double operations per frame); the f0 loads are not part of it: sixteen frames
    // are fetched while the previous sixteen run the chain (with the load inside the loop every frame paid a memory round trip: 89 us per clip)
    constexpr int CH = 16;
    const float* f0b = f0 + (size_t)b * T;
    float nxt[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) nxt[k] = k < T ? f0b[k] : 0.f;
    for (int c0 = 0; c0 < T; c0 += CH) {
        float cur[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) cur[k] = nxt[k];
#pragma unroll
        for (int k = 0; k < CH; ++k) nxt[k] = c0 + CH + k < T ? f0b[c0 + CH + k] : 0.f;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int f = c0 + k;
            if (f >= T) break;
            const float fh = cur[k] * mult;
            const float q = __fdiv_rn(fh, sr);
            const float r = q - floorf(q);                          // (f / sr) % 1, f >= 0
            if (f == 0) {
                const float rad0 = r + ini;                         // rad_values[:, 0, :] += rand_ini  (fp32)
                base = (double)rad0 - (double)r;
                flprev = (double)floorf(rad0);                      // floor(fp32(cumsum[0])): no shift at sample 0
                fl00[b * dim + h] = (float)flprev;
            }
            SrcFrame o; o.base = base; o.flprev = flprev; o.delta_acc = delta_acc;
            fr[((size_t)b * T + f) * dim + h] = o;
            const double d_end = base + (double)hop * (double)r;
            const double flend = (double)floorf((float)d_end);
            const float sr32 = r + (-1.0f);                         // fp32(rad + shift)
            const double delta = (double)sr32 - ((double)r - 1.0);
            delta_acc += delta * (flend - flprev);
            flprev = flend;
            base = d_end;
        }
    }
}

__global__ void k_source_samples(const float* __restrict__ f0, const SrcFrame* __restrict__ fr, const float* __restrict__ fl00,
                                 const float* __restrict__ lin_w, const float* __restrict__ lin_b, float* __restrict__ har,
                                 int T, int hop, int dim, float sr, int stride_samples, unsigned long long seed, int clip0,
                                 const int* __restrict__ clip_ids, float sine_amp, float noise_std) {
    const int b = blockIdx.y;
    const unsigned cid = (unsigned)(clip_ids ? clip_ids[b] : clip0 + b);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * hop) return;
    const int f = i / hop, j = i - f * hop;
    const float f0v = f0[(size_t)b * T + f];
    const float uv = f0v > 0.f ? 1.f : 0.f;
    const float noise_amp = uv * noise_std + ((1.f - uv) * sine_amp) / 3.f;
    float zs[12];
    for (int q = 0; q < (dim + 3) / 4; ++q)
        philox_normal4((unsigned)i, (unsigned)q, cid, PURPOSE_SINE_NOISE, seed, zs + 4 * q);
    float acc = 0.f;
    for (int h = 0; h < dim; ++h) {
        const float fh = f0v * (float)(h + 1);
        const float q = __fdiv_rn(fh, sr);
        const float r = q - floorf(q);
        const SrcFrame s = fr[((size_t)b * T + f) * dim + h];
        const double D = s.base + (double)(j + 1) * (double)r;
        const double fl = (double)floorf((float)D);
        const float sr32 = r + (-1.0f);
        const double delta = (double)sr32 - ((double)r - 1.0);
        const double P = D - (fl - (double)fl00[b * dim + h]) + s.delta_acc + delta * (fl - s.flprev);
        const float ph = (float)P;
        const float sine = sinf((ph * 2.0f) * 3.14159265358979323846f) * sine_amp;
        const float v = sine * uv + noise_amp * zs[h];
        acc += lin_w[h] * v;
    }
    har[(size_t)b * stride_samples + i] = tanhf(acc + lin_b[0]);
}

// noise_convs[i] (models.py:346-350): Conv1d(1, cout, kernel K, stride s, padding pad) on the excitation,
// written (not accumulated) into the stage's frame-major buffer before the transposed conv adds onto it.
// w is stored TRANSPOSED, [K][cout]: the lanes of a wave hold consecutive output channels, so a tap's weights are one coalesced load (with the
// checkpoint's [cout][K] every lane read its own cache line: 68 us per stage for 28 MB of output, round 4 profile)
// conv_post + tanh (models.py:357,383-385): Conv1d(c_last, 1, 7, padding 3) on leaky_relu(x, 0.01) -- ONE output channel, so this is a 7 x c_last
// dot product per sample, not a GEMM (as a 32-column MFMA tile it took 174 us per clip for 49 M multiply-adds).  fp32 FMAs, a block of 256
// samples stages its 262 rows (already through the leaky ReLU, zero outside the clip) in LDS with an odd row pitch.
__global__ void __launch_bounds__(256) k_conv_post(const float* __restrict__ x, const float* __restrict__ w /* [c][7] */, const float* __restrict__ bias,
                                                   float* __restrict__ wav, int C, int clip_stride, int clip_len) {
    extern __shared__ float sm[];                  // [262][C + 1] rows, then [7][C] weights (tap-major)
    const int clip = blockIdx.y, n0 = blockIdx.x * 256;
    const int pitch = C + 1;
    float* wl = sm + 262 * pitch;
    for (int i = threadIdx.x; i < 7 * C; i += 256) { const int tap = i / C, ci = i - tap * C; wl[i] = w[ci * 7 + tap]; }
    const int c4n = C >> 2;
    for (int i = threadIdx.x; i < 262 * c4n; i += 256) {
        const int r = i / c4n, c4 = (i - r * c4n) * 4;
        const int n = n0 - 3 + r;
        f32x4 v{0.f, 0.f, 0.f, 0.f};
        if (n >= 0 && n < clip_len) v = ld4(x + ((size_t)clip * clip_stride + n) * C + c4);
#pragma unroll
        for (int j = 0; j < 4; ++j) sm[r * pitch + c4 + j] = v[j] > 0.f ? v[j] : 0.01f * v[j];
    }
    __syncthreads();
    const int n = n0 + threadIdx.x;
    if (n >= clip_len) return;
    float acc = 0.f;
    for (int tap = 0; tap < 7; ++tap) {
        const float* xr = sm + (threadIdx.x + tap) * pitch;
        const float* wr = wl + tap * C;
        for (int ci = 0; ci < C; ++ci) acc = fmaf(wr[ci], xr[ci], acc);
    }
    wav[(size_t)clip * clip_len + n] = tanhf(acc + bias[0]);
}

// any channel count (one output per thread and pass)
__global__ void k_noise_conv_any(const float* __restrict__ har, const float* __restrict__ w, const float* __restrict__ bias,
                                 float* __restrict__ out, int cout, int K, int s, int pad, int len_out, int len_in, int stride_out, int stride_in) {
    extern __shared__ float sm[];
    const int b = blockIdx.y;
    const int n0 = blockIdx.x * 64;
    const int span = 63 * s + K;
    const int base = n0 * s - pad;
    for (int i = threadIdx.x; i < span; i += blockDim.x) {
        const int p = base + i;
        sm[i] = (p >= 0 && p < len_in) ? har[(size_t)b * stride_in + p] : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * cout; e += blockDim.x) {
        const int nl = e / cout, co = e - nl * cout;
        const int n = n0 + nl;
        if (n >= len_out) continue;
        float acc = 0.f;
        for (int j = 0; j < K; ++j) acc = fmaf(w[(size_t)j * cout + co], sm[nl * s + j], acc);
        out[((size_t)b * stride_out + n) * cout + co] = acc + bias[co];
    }
}

// The power-of-two widths: a thread owns ONE output channel and NPT of the block's FR = NPT * 256 / cout output samples.  FR = 64 everywhere but
// at the first stage (256 channels x 6 888 samples per 10 s clip: 108 blocks of 64 samples left 58 % of the CUs idle and ran 180 us; 16 samples per
// block = 431 blocks).  Four taps per LDS read (ds_read_b128; s and K are multiples of 4 there): the fmaf chain of an output still runs over its
// taps in ascending order -- bit-identical to the scalar form.
template <int NPT, int FR = 64>
__global__ void __launch_bounds__(256) k_noise_conv(const float* __restrict__ har, const float* __restrict__ w, const float* __restrict__ bias,
                                                    float* __restrict__ out, int cout, int K, int s, int pad, int len_out, int len_in,
                                                    int stride_out, int stride_in) {
    extern __shared__ __attribute__((aligned(16))) float sm[];          // [(FR - 1) * s + K] excitation samples
    const int b = blockIdx.y;
    const int n0 = blockIdx.x * FR;
    const int span = (FR - 1) * s + K;
    const int base = n0 * s - pad;
    for (int i = threadIdx.x; i < span; i += blockDim.x) {
        const int p = base + i;
        sm[i] = (p >= 0 && p < len_in) ? har[(size_t)b * stride_in + p] : 0.f;
    }
    __syncthreads();
    // tap-major: one (coalesced) weight load per tap feeds NPT outputs; the excitation sample is an LDS broadcast
    const int co = threadIdx.x % cout, grp = threadIdx.x / cout, G = 256 / cout;
    float acc[NPT];
#pragma unroll
    for (int q = 0; q < NPT; ++q) acc[q] = 0.f;
    if (((s | K) & 3) == 0) {
        for (int j = 0; j < K; j += 4) {
            float wv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) wv[t] = w[(size_t)(j + t) * cout + co];
#pragma unroll
            for (int q = 0; q < NPT; ++q) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(sm + (grp + q * G) * s + j);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[q] = fmaf(wv[t], x[t], acc[q]);
            }
        }
    } else {
        for (int j = 0; j < K; ++j) {
            const float wv = w[(size_t)j * cout + co];
#pragma unroll
            for (int q = 0; q < NPT; ++q) acc[q] = fmaf(wv, sm[(grp + q * G) * s + j], acc[q]);
        }
    }
    const float bv = bias[co];
#pragma unroll
    for (int q = 0; q < NPT; ++q) {
        const int n = n0 + grp + q * G;
        if (n < len_out) out[((size_t)b * stride_out + n) * cout + co] = acc[q] + bv;
    }
}

}  // namespace

// =================================================================================================
// ------------------------------------------------------------------------------------------------
// Round 4: the MRF convs of the WIDE stages (128 / 256 channels: 60 % of the generator's time) on the sampler's tgemm engine (tgemm.h, split
// activations: the same three-MFMA products as conv_gemm's staging path, 1.5-1.8x its rate -- the operand arrives as fp16 [hi | lo] row
// planes of lrelu(x), written by the producing epilogue and DMA'd into LDS, instead of being split from fp32 rows on the way in).
//   step of a ResBlock1 (models.py:57-64):  xt = c1(lrelu(x))  -> only lrelu(xt) is ever used: planes, no fp32 store  (TEpiVocMid)
//                                           x  = c2(lrelu(xt)) + x  -> fp32 rows (the residual stream / the MRF mean) + planes of lrelu(x)
//                                                                      for the next step; the residual is the accumulator init  (TEpiVocOut)
//   step of a ResBlock2 (models.py:86-91):  x  = c(lrelu(x)) + x    (TEpiVocOut)
// Planes are zero on gap rows (the convs' zero padding); the fp32 rows are not masked (every reader masks, as on the conv_gemm path).
// ------------------------------------------------------------------------------------------------
struct VRows {
    int stride, len, n_rows;
    __device__ __forceinline__ bool valid(int row) const { return row < n_rows && (row - (row / stride) * stride) < len; }
};

struct TEpiVocMid {
    struct Args { _Float16* ph; const float* bias; int C; VRows vr; };
    template <int NT_N>
    __device__ __forceinline__ void init(const Args&, int, int, int, f32x16 (&acc)[NT_N]) const {
#pragma unroll
        for (int nt = 0; nt < NT_N; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
    }
    template <int NT_N>
    __device__ __forceinline__ void finish(const Args& e, int mt, int row0, int lane, f32x16 (&acc)[NT_N]) const {
        const int cb = mt * 32 + 16 * (lane >> 5);
        if (cb >= e.C) return;
        float b[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const f32x4 v = ld4(e.bias + cb + 4 * q); b[4 * q] = v[0]; b[4 * q + 1] = v[1]; b[4 * q + 2] = v[2]; b[4 * q + 3] = v[3]; }
#pragma unroll
        for (int nt = 0; nt < NT_N; ++nt) {
            const int frame = row0 + 32 * nt + (lane & 31);
            const bool ok = e.vr.valid(frame);
            float hv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { const float v = acc[nt][i] + b[i]; hv[i] = ok ? (v > 0.f ? v : 0.1f * v) : 0.f; }
            store_hi_lo16(e.ph + (size_t)frame * (2 * e.C) + cb, e.C, hv);
        }
    }
};

struct TEpiVocOut {
    struct Args { const float* res; float* out; _Float16* ph; const float* bias; int C; float alpha; int accumulate; VRows vr; };
    template <int NT_N>
    __device__ __forceinline__ void init(const Args& e, int mt, int row0, int lane, f32x16 (&acc)[NT_N]) const {
        const int cb = mt * 32 + 16 * (lane >> 5);
#pragma unroll
        for (int nt = 0; nt < NT_N; ++nt) {
            const float* p = e.res + (size_t)(row0 + 32 * nt + (lane & 31)) * e.C + cb;
            const f32x4 v0 = ld4(p), v1 = ld4(p + 4), v2 = ld4(p + 8), v3 = ld4(p + 12);
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc[nt][i] = v0[i]; acc[nt][4 + i] = v1[i]; acc[nt][8 + i] = v2[i]; acc[nt][12 + i] = v3[i]; }
        }
    }
    template <int NT_N>
    __device__ __forceinline__ void finish(const Args& e, int mt, int row0, int lane, f32x16 (&acc)[NT_N]) const {
        const int cb = mt * 32 + 16 * (lane >> 5);
        if (cb >= e.C) return;
        float b[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const f32x4 v = ld4(e.bias + cb + 4 * q); b[4 * q] = v[0]; b[4 * q + 1] = v[1]; b[4 * q + 2] = v[2]; b[4 * q + 3] = v[3]; }
#pragma unroll
        for (int nt = 0; nt < NT_N; ++nt) {
            const int frame = row0 + 32 * nt + (lane & 31);
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = acc[nt][i] + b[i];
            float* po = e.out + (size_t)frame * e.C + cb;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 o{v[4 * q] * e.alpha, v[4 * q + 1] * e.alpha, v[4 * q + 2] * e.alpha, v[4 * q + 3] * e.alpha};
                if (e.accumulate) { const f32x4 old = ld4(po + 4 * q); o[0] += old[0]; o[1] += old[1]; o[2] += old[2]; o[3] += old[3]; }
                st4(po + 4 * q, o);
            }
            if (e.ph) {
                const bool ok = e.vr.valid(frame);
                float hv[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) hv[i] = ok ? (v[i] > 0.f ? v[i] : 0.1f * v[i]) : 0.f;
                store_hi_lo16(e.ph + (size_t)frame * (2 * e.C) + cb, e.C, hv);
            }
        }
    }
};

// planes of lrelu(x) for the first step of every resblock of a stage (x = the upsampled signal + source, fp32 rows)
__global__ void k_lrelu_planes(const float* __restrict__ x, _Float16* __restrict__ ph, int C, VRows vr, int rows) {
    const int per_row = C >> 2;
    const long long n = (long long)rows * per_row;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / per_row), c4 = (int)(i - (long long)row * per_row) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (vr.valid(row)) {
            const f32x4 a = ld4(x + (size_t)row * C + c4);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = a[j] > 0.f ? a[j] : 0.1f * a[j];
        }
        _Float16* q = ph + (size_t)row * (2 * C) + c4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const _Float16 h = (_Float16)v[j];
            q[j] = h;
            q[C + j] = (_Float16)(v[j] - (float)h);
        }
    }
}

constexpr int VT_GUARD = 32;          // zero rows in front of / behind a plane buffer (the largest resblock halo is (11 / 2) * 5 = 25)

// tiling: 256 channels = 8 output tiles = one per wave; 128 channels = 4 tiles, the workgroup's eight waves then cover TWO 64-frame sub-tiles
template <class Epi>
int vt_launch(const _Float16* x, int C, int taps, int dil, const _Float16* w, const typename Epi::Args& e, int rows, hipStream_t st) {
    TGemmArgs a{};
    a.x = x; a.cin = C; a.taps = taps; a.dil = dil; a.w = w; a.m_tiles = C / 32; a.w_planes = 2; a.variant_halfs = 0; a.n_variants = 1;
    a.step_ptr = nullptr; a.step_off = 0;
    // clip_rows = the frame tile: every tile walks its K loop from group 0, so a clip's samples are bit-identical alone and at any batch position
    // (the staggered starts of tgemm's small tilings are a latency measure for grids that do not fill the chip)
    a.clip_rows = 64;
    if (C == 128) return tgemm_launch<2, 8, 2, 4, 2, Epi, 1, 1, 2, 0, 0, 2>(a, e, rows, 1, st);
    if (rows / 64 >= 200) return tgemm_launch<2, 8, 2, 4, 2, Epi, 1, 1, 2>(a, e, rows, 1, st);
    a.clip_rows = 32;
    return tgemm_launch<1, 8, 2, 4, 2, Epi, 1, 1, 2>(a, e, rows, 1, st);          // few rows (one clip at the 256-channel stage): 32-frame tiles fill the chip
}

struct dsvc_vocoder {
    dsvc_vocoder_cfg cfg;
    std::map<std::string, std::vector<float>> host;
    bool finalized = false;
    int hop = 1, dim = 9;

    PackedConv conv_pre, conv_post;
    DevBuf conv_post_w;                          // conv_post's weights as they come, fp32 [c_last][7] (k_conv_post)
    std::vector<PackedConv> ups;                 // polyphase transposed convs
    std::vector<DevBuf> nc_w, nc_b;              // noise convs (plain fp32)
    std::vector<int> nc_k, nc_s, nc_pad;
    std::vector<PackedConv> rb1, rb2;            // [stage*nk*3 + j*3 + m]
    std::vector<DevBuf> rb1_f32, rb2_f32;        // same index: [tap][ci][co] fp32 for the narrow stages' fused pair kernel (else empty)
    std::vector<DevBuf> rb1_t, rb2_t;            // same index: tgemm fragment order (hi | lo planes) for the wide stages' convs (else empty)
    DevBuf pl[8][4];                             // per wide stage: [VT_GUARD + rows + VT_GUARD][2 C] fp16 operand planes -- lrelu(x_stage), xt, two ping-pong.
                                                 // (Not shared between stages: the guard rows are zero because nothing ever writes them, and another
                                                 //  stage's rows would land on them -- row pitch and row count differ)
    bool t_stage[8] = {};                        // stage i runs its resblock convs on the tgemm engine: split precision, 128 / 256 channels, halos within VT_GUARD
    DevBuf lin_w, lin_b;
    int gap_frames = 8;

    // workspace
    int wsB = 0, wsT = 0, Tp = 0;
    DevBuf mel_in, har, frames, fl00, buf[5];

    ~dsvc_vocoder() {
        auto rel = [](PackedConv& p) { p.w.release(); p.bias.release(); };
        rel(conv_pre); rel(conv_post); conv_post_w.release();
        for (auto& p : ups) rel(p);
        for (auto& p : rb1) rel(p);
        for (auto& p : rb2) rel(p);
        for (auto& b : rb1_f32) b.release();
        for (auto& b : rb2_f32) b.release();
        for (auto& b : rb1_t) b.release();
        for (auto& b : rb2_t) b.release();
        for (auto& st : pl) for (auto& b : st) b.release();
        for (auto& b : nc_w) b.release();
        for (auto& b : nc_b) b.release();
        for (DevBuf* b : {&lin_w, &lin_b, &mel_in, &har, &frames, &fl00, &buf[0], &buf[1], &buf[2], &buf[3], &buf[4]}) b->release();
    }

    // folded weight of a weight-normed layer, or the plain weight (remove_weight_norm, models.py:28,389-396)
    int folded(const std::string& base, size_t numel, int dim0, std::vector<float>& out) {
        auto wi = host.find(base + ".weight");
        if (wi != host.end()) {
            if (wi->second.size() != numel) return fail(DSVC_EINVAL, "vocoder: '%s.weight' has %zu elements, expected %zu", base.c_str(), wi->second.size(), numel);
            out = wi->second;
            return DSVC_OK;
        }
        auto gi = host.find(base + ".weight_g"), vi = host.find(base + ".weight_v");
        if (gi == host.end() || vi == host.end()) return fail(DSVC_ESTATE, "vocoder: neither '%s.weight' nor its weight_g/weight_v pair was loaded", base.c_str());
        if (vi->second.size() != numel || (int)gi->second.size() != dim0)
            return fail(DSVC_EINVAL, "vocoder: '%s' weight_v/weight_g have %zu/%zu elements, expected %zu/%d", base.c_str(), vi->second.size(), gi->second.size(), numel, dim0);
        out.resize(numel);
        const size_t inner = numel / dim0;
        for (int o = 0; o < dim0; ++o) {
            const float* v = vi->second.data() + (size_t)o * inner;
            float ss = 0.f;
            for (size_t i = 0; i < inner; ++i) ss += v[i] * v[i];
            const float sc = gi->second[o] / sqrtf(ss);
            for (size_t i = 0; i < inner; ++i) out[(size_t)o * inner + i] = v[i] * sc;
        }
        return DSVC_OK;
    }
    const std::vector<float>* plain(const std::string& k, size_t numel) {
        auto it = host.find(k);
        if (it == host.end()) { fail(DSVC_ESTATE, "vocoder: tensor '%s' was never loaded", k.c_str()); return nullptr; }
        if (it->second.size() != numel) { fail(DSVC_EINVAL, "vocoder: tensor '%s' has %zu elements, expected %zu", k.c_str(), it->second.size(), numel); return nullptr; }
        return &it->second;
    }

    int finalize();
    int ensure_ws(int B, int T, hipStream_t st);
    int run(const float* mel, const float* f0, float* wav, int B, int T, unsigned long long seed, int clip0, const int* clip_ids, hipStream_t st);
};

int dsvc_vocoder::finalize() {
    const int nu = cfg.n_ups, nk = cfg.n_kernels, ch0 = cfg.upsample_initial_channel, M = cfg.num_mels;
    if (nu < 1 || nu > 8 || nk < 1 || nk > 4) return fail(DSVC_EINVAL, "vocoder: bad stage / kernel counts");
    const bool rb2_type = cfg.resblock == 2;
    const int ndil = cfg.n_dilations > 0 ? cfg.n_dilations : 3;
    if (cfg.resblock < 0 || cfg.resblock > 2 || ndil > 3 || (!rb2_type && ndil != 3)) return fail(DSVC_EINVAL, "vocoder: resblock %d with %d dilations", cfg.resblock, ndil);
    if (M % 16) return fail(DSVC_EINVAL, "vocoder: num_mels must be a multiple of 16");
    dim = cfg.harmonics + 1;
    if (dim > 12) return fail(DSVC_EINVAL, "vocoder: at most 11 harmonics");
    hop = 1;
    for (int i = 0; i < nu; ++i) hop *= cfg.upsample_rates[i];
    if ((ch0 >> nu) < 16 || ((ch0 >> nu) % 16)) return fail(DSVC_EINVAL, "vocoder: channel count after the last stage must be a multiple of 16");
    std::vector<float> w;
    {   // conv_pre: Conv1d(num_mels, ch0, 7, padding 3)  (models.py:336)
        DSVC_TRY(folded("conv_pre", (size_t)ch0 * M * 7, ch0, w));
        const std::vector<float>* b = plain("conv_pre.bias", ch0);
        if (!b) return DSVC_ESTATE;
        DSVC_TRY(pack_conv(conv_pre, ch0, 7, M, 1, [&](int co, int tap, int ci) { return w[((size_t)co * M + ci) * 7 + tap]; }, b->data(), ch0));
    }
    ups.resize(nu); nc_w.resize(nu); nc_b.resize(nu); nc_k.resize(nu); nc_s.resize(nu); nc_pad.resize(nu);
    rb1.resize((size_t)nu * nk * 3); rb2.resize((size_t)nu * nk * 3);
    rb1_f32.resize(rb1.size()); rb2_f32.resize(rb2.size());
    rb1_t.resize(rb1.size()); rb2_t.resize(rb2.size());
    auto pack_t = [&](DevBuf& dst, const std::vector<float>& wsrc, int c, int rk) -> int {     // [O][I][K] fp32 -> tgemm fragments (tgemm.h: k_tpack)
        const int mt = c / 32;
        std::vector<int> rm(mt * 32);
        for (int r = 0; r < mt * 32; ++r) rm[r] = (r >> 5) * 32 + trow_to_ch16(r & 31);
        DevBuf dsrc, drm;
        DSVC_TRY(upload(dsrc, wsrc.data(), wsrc.size() * 4));
        DSVC_TRY(upload(drm, rm.data(), rm.size() * 4));
        const size_t halfs = tpacked_halfs(mt, rk, c, 2, 1);
        DSVC_TRY(dst.alloc(halfs * 2));
        const long long total = (long long)halfs / 2;
        hipLaunchKernelGGL(k_tpack, dim3((unsigned)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535)), dim3(256), 0, 0, dsrc.as<float>(), drm.as<int>(),
                           (const float*)nullptr, dst.as<_Float16>(), c, rk, c, 0, mt, 2, 1, 1.0f, 0u);
        DSVC_HIP(hipGetLastError());
        DSVC_HIP(hipDeviceSynchronize());
        dsrc.release(); drm.release();
        return DSVC_OK;
    };
    int need_gap = 3;
    int rate = 1;
    for (int i = 0; i < nu; ++i) {
        const int u = cfg.upsample_rates[i], k = cfg.upsample_kernel_sizes[i];
        const int cin = ch0 >> i, cout = ch0 >> (i + 1);
        const int p = (k - u) / 2;
        if ((k - u) % 2) return fail(DSVC_EINVAL, "vocoder: upsample kernel %d / rate %d: odd padding is not supported", k, u);
        // ConvTranspose1d(cin, cout, k, u, padding p) (models.py:342-345), weight [cin][cout][k], as a polyphase conv:
        //   out[q*u + phi] = sum_o sum_ci W[ci][co][phi + p - o*u] * x[q + o],   o in [omin, omax]
        DSVC_TRY(folded("ups." + std::to_string(i), (size_t)cin * cout * k, cin, w));
        const std::vector<float>* b = plain("ups." + std::to_string(i) + ".bias", cout);
        if (!b) return DSVC_ESTATE;
        const int omax = (u - 1 + p) / u;             // largest o with phi + p - o*u >= 0   (phi = u-1)
        const int omin = -((k - 1 - p) / u);          // smallest o with phi + p - o*u <= k-1 (phi = 0)
        const int reach = omax > -omin ? omax : -omin;
        const int taps = 2 * reach + 1;
        DSVC_TRY(pack_conv(ups[i], u * cout, taps, cin, 1,
                           [&](int col, int tap, int ci) {
                               const int phi = col / cout, co = col % cout, o = tap - reach;
                               const int j = phi + p - o * u;
                               return (j >= 0 && j < k) ? w[((size_t)ci * cout + co) * k + j] : 0.f;
                           },
                           b->data(), cout));
        rate *= u;
        // noise conv (models.py:346-350)
        int s = 1;
        for (int q = i + 1; q < nu; ++q) s *= cfg.upsample_rates[q];
        const bool last = (i + 1 == nu);
        nc_s[i] = last ? 1 : s; nc_k[i] = last ? 1 : 2 * s; nc_pad[i] = last ? 0 : s / 2;
        if (cfg.use_source) {
            const std::vector<float>* nw = plain("noise_convs." + std::to_string(i) + ".weight", (size_t)cout * nc_k[i]);
            const std::vector<float>* nb = plain("noise_convs." + std::to_string(i) + ".bias", cout);
            if (!nw || !nb) return DSVC_ESTATE;
            std::vector<float> wt(nw->size());                      // [cout][K] -> [K][cout] (k_noise_conv)
            for (int co = 0; co < cout; ++co)
                for (int kk = 0; kk < nc_k[i]; ++kk) wt[(size_t)kk * cout + co] = (*nw)[(size_t)co * nc_k[i] + kk];
            DSVC_TRY(upload(nc_w[i], wt.data(), wt.size() * 4)); DSVC_TRY(upload(nc_b[i], nb->data(), nb->size() * 4));
        }
        // resblocks (ResBlock1, models.py:33-64: conv pairs; ResBlock2, models.py:73-91: one conv per residual step)
        {
            int mh = 0;
            for (int j = 0; j < nk; ++j)
                for (int m = 0; m < ndil; ++m) { const int hh = (cfg.resblock_kernel_sizes[j] / 2) * cfg.resblock_dilations[j][m]; if (hh > mh) mh = hh; }
            t_stage[i] = cfg.precision == DSVC_PREC_F16_X3 && (cout == 128 || cout == 256) && mh <= VT_GUARD;
        }
        for (int j = 0; j < nk; ++j) {
            const int rk = cfg.resblock_kernel_sizes[j];
            if (!(rk & 1)) return fail(DSVC_EINVAL, "vocoder: even resblock kernel size");
            for (int m = 0; m < ndil; ++m) {
                const int d = cfg.resblock_dilations[j][m];
                const std::string base = "resblocks." + std::to_string(i * nk + j) + ".";
                const size_t idx = ((size_t)i * nk + j) * 3 + m;
                const int halo = (rk / 2) * d;
                if (ceil_div(halo, rate) > need_gap) need_gap = ceil_div(halo, rate);
                if (rb2_type) {
                    DSVC_TRY(folded(base + "convs." + std::to_string(m), (size_t)cout * cout * rk, cout, w));
                    const std::vector<float>* b1 = plain(base + "convs." + std::to_string(m) + ".bias", cout);
                    if (!b1) return DSVC_ESTATE;
                    DSVC_TRY(pack_conv(rb1[idx], cout, rk, cout, d, [&](int co, int tap, int ci) { return w[((size_t)co * cout + ci) * rk + tap]; }, b1->data(), cout));
                    if (t_stage[i]) DSVC_TRY(pack_t(rb1_t[idx], w, cout, rk));
                    continue;
                }
                DSVC_TRY(folded(base + "convs1." + std::to_string(m), (size_t)cout * cout * rk, cout, w));
                const std::vector<float>* b1 = plain(base + "convs1." + std::to_string(m) + ".bias", cout);
                if (!b1) return DSVC_ESTATE;
                DSVC_TRY(pack_conv(rb1[idx], cout, rk, cout, d, [&](int co, int tap, int ci) { return w[((size_t)co * cout + ci) * rk + tap]; }, b1->data(), cout));
                if (t_stage[i]) DSVC_TRY(pack_t(rb1_t[idx], w, cout, rk));
                auto pack_f32 = [&](DevBuf& dst) {      // [tap][ci][co]
                    std::vector<float> t((size_t)rk * cout * cout);
                    for (int tap = 0; tap < rk; ++tap)
                        for (int ci = 0; ci < cout; ++ci)
                            for (int co = 0; co < cout; ++co) t[((size_t)tap * cout + ci) * cout + co] = w[((size_t)co * cout + ci) * rk + tap];
                    return upload(dst, t.data(), t.size() * 4);
                };
                const bool narrow = (cout == 16 || cout == 32) && (rk / 2) * d + rk / 2 <= 40;
                if (narrow) DSVC_TRY(pack_f32(rb1_f32[idx]));
                DSVC_TRY(folded(base + "convs2." + std::to_string(m), (size_t)cout * cout * rk, cout, w));
                const std::vector<float>* b2 = plain(base + "convs2." + std::to_string(m) + ".bias", cout);
                if (!b2) return DSVC_ESTATE;
                DSVC_TRY(pack_conv(rb2[idx], cout, rk, cout, 1, [&](int co, int tap, int ci) { return w[((size_t)co * cout + ci) * rk + tap]; }, b2->data(), cout));
                if (t_stage[i]) DSVC_TRY(pack_t(rb2_t[idx], w, cout, rk));
                if (narrow) DSVC_TRY(pack_f32(rb2_f32[idx]));
            }
        }
        if (ceil_div(reach, rate / u) > need_gap) need_gap = ceil_div(reach, rate / u);
    }
    {   // conv_post: Conv1d(c_last, 1, 7, padding 3)  (models.py:357)
        const int cl = ch0 >> nu;
        DSVC_TRY(folded("conv_post", (size_t)cl * 7, 1, w));
        const std::vector<float>* b = plain("conv_post.bias", 1);
        if (!b) return DSVC_ESTATE;
        DSVC_TRY(pack_conv(conv_post, 1, 7, cl, 1, [&](int, int tap, int ci) { return w[(size_t)ci * 7 + tap]; }, b->data(), 1));
        DSVC_TRY(upload(conv_post_w, w.data(), (size_t)cl * 7 * 4));
    }
    if (cfg.use_source) {
        const std::vector<float>* lw = plain("m_source.l_linear.weight", dim);
        const std::vector<float>* lb = plain("m_source.l_linear.bias", 1);
        if (!lw || !lb) return DSVC_ESTATE;
        DSVC_TRY(upload(lin_w, lw->data(), dim * 4)); DSVC_TRY(upload(lin_b, lb->data(), 4));
    }
    gap_frames = need_gap < 8 ? 8 : need_gap;
    host.clear();
    finalized = true;
    return DSVC_OK;
}

int dsvc_vocoder::ensure_ws(int B, int T, hipStream_t st) {
    if (B == wsB && T == wsT) return DSVC_OK;
    if (B < 1 || T < 1) return fail(DSVC_EINVAL, "bad batch/frames %d/%d", B, T);
    Tp = round_up(T + gap_frames, 32);
    const size_t frames_total = (size_t)B * Tp;
    if (frames_total * hop > 0x7fffffffull) return fail(DSVC_EINVAL, "vocoder: batch too large for 32-bit row indices; split the batch");
    DSVC_TRY(mel_in.alloc(frames_total * cfg.num_mels * 4));
    DSVC_TRY(har.alloc(frames_total * hop * 4));
    DSVC_TRY(frames.alloc((size_t)B * T * dim * sizeof(SrcFrame)));
    DSVC_TRY(fl00.alloc((size_t)B * dim * 4));
    // largest stage buffer: rows * channels is maximal where rate/2^(i+1) peaks
    size_t mx = frames_total * cfg.upsample_initial_channel;
    int rate = 1;
    for (int i = 0; i < cfg.n_ups; ++i) {
        rate *= cfg.upsample_rates[i];
        const size_t e = frames_total * rate * (cfg.upsample_initial_channel >> (i + 1));
        if (e > mx) mx = e;
    }
    for (int i = 0; i < 5; ++i) DSVC_TRY(buf[i].alloc(mx * 4));
    {   // operand planes of the wide stages (tgemm path): [VT_GUARD | rows | VT_GUARD] x 2 C halfs.  Only the guard rows have to be cleared: every row
        // of [0, rows) is rewritten by k_lrelu_planes or a producing epilogue (zeros on gap rows) before a conv reads it.  Round 4 cleared all
        // four whole buffers of every wide stage with hipMemset on the NULL stream at every (B, T) change -- 117 MB per 10 s clip at the
        // 128-channel stage, on every chunk of a different length, and not ordered against work in flight on a non-blocking stream.
        // Now: the leading and the trailing guard of the NEW layout, asynchronously on the caller's stream.
        int r2 = 1;
        for (int i = 0; i < cfg.n_ups; ++i) {
            r2 *= cfg.upsample_rates[i];
            const int c = cfg.upsample_initial_channel >> (i + 1);
            if (!t_stage[i]) continue;
            const size_t row_bytes = (size_t)2 * c * 2, rows = frames_total * r2;
            for (int q = 0; q < 4; ++q) {
                DSVC_TRY(pl[i][q].alloc((rows + 2 * VT_GUARD) * row_bytes));
                DSVC_HIP(hipMemsetAsync(pl[i][q].p, 0, VT_GUARD * row_bytes, st));
                DSVC_HIP(hipMemsetAsync(pl[i][q].as<char>() + (VT_GUARD + rows) * row_bytes, 0, VT_GUARD * row_bytes, st));
            }
        }
    }
    wsB = B; wsT = T;
    return DSVC_OK;
}

int dsvc_vocoder::run(const float* mel, const float* f0, float* wav, int B, int T, unsigned long long seed, int clip0, const int* clip_ids, hipStream_t st) {
    DSVC_TRY(ensure_ws(B, T, st));
    const int nu = cfg.n_ups, nk = cfg.n_kernels, ch0 = cfg.upsample_initial_channel, M = cfg.num_mels;
    const bool rb2_type = cfg.resblock == 2;
    const int ndil = cfg.n_dilations > 0 ? cfg.n_dilations : 3;
    const int prec = cfg.precision;
    {
        const size_t n = (size_t)B * T * M;
        hipLaunchKernelGGL(k_prep_mel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, st, mel, mel_in.as<float>(), B, T, M, Tp, cfg.mel_scale);
    }
    const bool src = cfg.use_source != 0;
    // excitation (models.py:363-366)
    if (src) hipLaunchKernelGGL(k_source_frames, dim3(B), dim3(32), 0, st, f0, frames.as<SrcFrame>(), fl00.as<float>(), T, hop, dim,
                       (float)cfg.sampling_rate, seed, clip0, clip_ids);
    if (src) hipLaunchKernelGGL(k_source_samples, dim3(ceil_div(T * hop, 256), B), dim3(256), 0, st, f0, frames.as<SrcFrame>(), fl00.as<float>(),
                       lin_w.as<float>(), lin_b.as<float>(), har.as<float>(), T, hop, dim, (float)cfg.sampling_rate, Tp * hop, seed, clip0, clip_ids,
                       0.1f, 0.003f);
    auto conv = [&](const PackedConv& pc, const float* x, int rows, int stride, int len, float slope) {
        ConvGemmArgs a{};
        a.x = x; a.ldx = pc.cin; a.n_rows = rows; a.clip_stride = stride; a.clip_len = len;
        a.cin = pc.cin; a.taps = pc.taps; a.dil = pc.dil; a.w = pc.w.as<_Float16>(); a.n_ctiles = pc.n_ctiles; a.w_planes = 2;
        a.in_slope = slope;
        return a;
    };
    float* U = buf[0].as<float>();
    float* A = buf[1].as<float>();
    float* Tm = buf[2].as<float>();
    float* prev = buf[4].as<float>();
    int rate = 1;
    {   // conv_pre (models.py:367)
        ConvGemmArgs a = conv(conv_pre, mel_in.as<float>(), B * Tp, Tp, T, 1.0f);
        EpiAffine::Args e{prev, ch0, conv_pre.bias.as<float>(), ch0, ch0, nullptr, 0, 1.0f, 0};
        DSVC_TRY(voc_dispatch<EpiAffine>(a, e, prec, st));
    }
    for (int i = 0; i < nu; ++i) {
        const int u = cfg.upsample_rates[i];
        const int cin = ch0 >> i, cout = ch0 >> (i + 1);
        const int rows_in = B * Tp * rate, stride_in = Tp * rate, len_in = T * rate;
        rate *= u;
        const int rows = B * Tp * rate, stride = Tp * rate, len = T * rate;
        float* S = buf[(i & 1) ? 4 : 3].as<float>();
        // x_source = noise_convs[i](har)  (models.py:373) written into U ...
        if (src) {
            auto nc = [&](auto kern, int fr) {      // fr output samples per block
                const size_t sm = (size_t)(fr * nc_s[i] + nc_k[i]) * 4;
                hipLaunchKernelGGL(kern, dim3(ceil_div(len, fr), B), dim3(256), sm, st, har.as<float>(), nc_w[i].as<float>(),
                                   nc_b[i].as<float>(), U, cout, nc_k[i], nc_s[i], nc_pad[i], len, T * hop, stride, Tp * hop);
            };
            switch (cout) {
                case 256: nc(k_noise_conv<16, 16>, 16); break;
                case 128: nc(k_noise_conv<32>, 64); break;
                case 64: nc(k_noise_conv<16>, 64); break;
                case 32: nc(k_noise_conv<8>, 64); break;
                case 16: nc(k_noise_conv<4>, 64); break;
                default: nc(k_noise_conv_any, 64); break;
            }
        }
        // ... then x = ups[i](leaky_relu(x, 0.1)) + x_source  (models.py:369-375)
        {
            ConvGemmArgs a = conv(ups[i], prev, rows_in, stride_in, len_in, 0.1f);
            (void)cin;
            EpiAffine::Args e{U, u * cout, ups[i].bias.as<float>(), cout, u * cout, nullptr, 0, 1.0f, src ? 1 : 0};
            DSVC_TRY(voc_dispatch<EpiAffine>(a, e, prec, st));
        }
        // MRF: mean over the nk resblocks (models.py:376-382)
        const bool tw = t_stage[i] && rows % 128 == 0;
        const VRows vr{stride, len, rows};
        auto plane = [&](int q) { return pl[i][q].as<_Float16>() + (size_t)VT_GUARD * 2 * cout; };
        if (tw) hipLaunchKernelGGL(k_lrelu_planes, dim3(2048), dim3(256), 0, st, U, plane(0), cout, vr, rows);
        for (int j = 0; j < nk; ++j) {
            for (int m = 0; m < ndil; ++m) {
                const size_t idx = ((size_t)i * nk + j) * 3 + m;
                if (tw) {      // the wide stages on the tgemm engine (epilogues above); planes: 0 = lrelu(U), 1 = xt, 2 / 3 = ping-pong
                    const bool lastm = (m + 1 == ndil);
                    const float* xin = (m == 0) ? U : A;
                    float* xout = lastm ? S : A;
                    const float al = lastm ? 1.0f / (float)nk : 1.0f;
                    const int accu = (lastm && j > 0) ? 1 : 0;
                    const _Float16* pin = (m == 0) ? plane(0) : plane(2 + ((m - 1) & 1));
                    _Float16* pout = lastm ? nullptr : plane(2 + (m & 1));
                    if (rb2_type) {
                        TEpiVocOut::Args e{xin, xout, pout, rb1[idx].bias.as<float>(), cout, al, accu, vr};
                        DSVC_TRY(vt_launch<TEpiVocOut>(pin, cout, rb1[idx].taps, rb1[idx].dil, rb1_t[idx].as<_Float16>(), e, rows, st));
                    } else {
                        TEpiVocMid::Args e1{plane(1), rb1[idx].bias.as<float>(), cout, vr};
                        DSVC_TRY(vt_launch<TEpiVocMid>(pin, cout, rb1[idx].taps, rb1[idx].dil, rb1_t[idx].as<_Float16>(), e1, rows, st));
                        TEpiVocOut::Args e2{xin, xout, pout, rb2[idx].bias.as<float>(), cout, al, accu, vr};
                        DSVC_TRY(vt_launch<TEpiVocOut>(plane(1), cout, rb2[idx].taps, rb2[idx].dil, rb2_t[idx].as<_Float16>(), e2, rows, st));
                    }
                    continue;
                }
                if (rb2_type) {   // ResBlock2 (models.py:86-91): x = c(leaky_relu(x)) + x, ping-pong U -> A -> Tm -> ...; the last step lands in the MRF mean
                    const float* xin = (m == 0) ? U : ((m & 1) ? A : Tm);
                    const bool lastm = (m + 1 == ndil);
                    float* xout = lastm ? S : ((m & 1) ? Tm : A);
                    ConvGemmArgs a = conv(rb1[idx], xin, rows, stride, len, 0.1f);
                    EpiAffine::Args e{xout, cout, rb1[idx].bias.as<float>(), cout, cout, xin, cout, lastm ? 1.0f / (float)nk : 1.0f, (lastm && j > 0) ? 1 : 0};
                    DSVC_TRY(voc_dispatch<EpiAffine>(a, e, prec, st));
                    continue;
                }
#ifdef DSVC_PROFILING
                static const bool no_fused = getenv("DSVC_VOC_NO_FUSED") && atoi(getenv("DSVC_VOC_NO_FUSED"));   // A/B knobs
                static const bool pair_f32 = getenv("DSVC_VOC_PAIR_F32") && atoi(getenv("DSVC_VOC_PAIR_F32"));
#else
                constexpr bool no_fused = false, pair_f32 = false;
#endif
                const int pk = rb1[idx].taps, pd = rb1[idx].dil;
                const bool mfma_pair = prec == DSVC_PREC_F16_X3 && !no_fused && !pair_f32 && (cout == 16 || cout == 32 || cout == 64) &&
                                       rb2[idx].taps == pk && rb2[idx].dil == 1 &&
                                       (size_t)2 * (32 * 2 * (cout == 64 ? 2 : 4) + 2 * (pk / 2) * pd) * (cout + 8) * 2 <= 64 * 1024;
                if (mfma_pair) {
                    // one LDS-fused MFMA kernel per conv pair; ping-pong U -> A -> Tm -> S (the kernel reads its input with halos, so it
                    // cannot run in place)
                    const float* fin = (m == 0) ? U : (m == 1 ? A : Tm);
                    float* fout = (m == 0) ? A : (m == 1 ? Tm : S);
                    const float al = (m == 2) ? 1.0f / (float)nk : 1.0f;
                    const int accu = (m == 2 && j > 0) ? 1 : 0;
                    const _Float16* pw1 = rb1[idx].w.as<_Float16>();
                    const _Float16* pw2 = rb2[idx].w.as<_Float16>();
                    const float* pb1 = rb1[idx].bias.as<float>();
                    const float* pb2 = rb2[idx].bias.as<float>();
                    if (cout == 16) DSVC_TRY((pair_mfma_launch<16, 4>(fin, fout, pw1, pb1, pw2, pb2, pk, pd, rows, stride, len, al, accu, st)));
                    else if (cout == 32) DSVC_TRY((pair_mfma_launch<32, 4>(fin, fout, pw1, pb1, pw2, pb2, pk, pd, rows, stride, len, al, accu, st)));
                    else DSVC_TRY((pair_mfma_launch<64, 2>(fin, fout, pw1, pb1, pw2, pb2, pk, pd, rows, stride, len, al, accu, st)));
                    continue;
                }
                if (rb1_f32[idx].p && !no_fused) {
                    // narrow stage: one LDS-fused kernel per conv pair; ping-pong U -> A -> Tm -> S (the pair kernel reads its
                    // input with halos, so it cannot run in place)
                    const float* fin = (m == 0) ? U : (m == 1 ? A : Tm);
                    float* fout = (m == 0) ? A : (m == 1 ? Tm : S);
                    const float al = (m == 2) ? 1.0f / (float)nk : 1.0f;
                    const int accu = (m == 2 && j > 0) ? 1 : 0;
                    const int rk = rb1[idx].taps, dd = rb1[idx].dil;
                    if (cout == 16)
                        DSVC_TRY((resblock_pair_launch<16, 256>(fin, fout, rb1_f32[idx].as<float>(), rb1[idx].bias.as<float>(), rb2_f32[idx].as<float>(),
                                                                rb2[idx].bias.as<float>(), rk, dd, rows, stride, len, al, accu, st)));
                    else
                        DSVC_TRY((resblock_pair_launch<32, 128>(fin, fout, rb1_f32[idx].as<float>(), rb1[idx].bias.as<float>(), rb2_f32[idx].as<float>(),
                                                                rb2[idx].bias.as<float>(), rk, dd, rows, stride, len, al, accu, st)));
                    continue;
                }
                const float* xin = (m == 0) ? U : A;
                {   // xt = c1(leaky_relu(x))
                    ConvGemmArgs a = conv(rb1[idx], xin, rows, stride, len, 0.1f);
                    EpiAffine::Args e{Tm, cout, rb1[idx].bias.as<float>(), cout, cout, nullptr, 0, 1.0f, 0};
                    DSVC_TRY(voc_dispatch<EpiAffine>(a, e, prec, st));
                }
                {   // x = c2(leaky_relu(xt)) + x ; the last one lands in the running MRF mean
                    ConvGemmArgs a = conv(rb2[idx], Tm, rows, stride, len, 0.1f);
                    const bool lastm = (m == 2);
                    EpiAffine::Args e{lastm ? S : A, cout, rb2[idx].bias.as<float>(), cout, cout, xin, cout,
                                      lastm ? 1.0f / (float)nk : 1.0f, (lastm && j > 0) ? 1 : 0};
                    DSVC_TRY(voc_dispatch<EpiAffine>(a, e, prec, st));
                }
            }
        }
        prev = S;
    }
    {   // x = tanh(conv_post(leaky_relu(x)))  -- F.leaky_relu default slope 0.01 (models.py:383-385)
        const int cl = ch0 >> nu;
        const size_t psm = ((size_t)262 * (cl + 1) + 7 * cl) * 4;
        if (prec == DSVC_PREC_F16_X3 && cl % 4 == 0 && psm <= 64 * 1024) {      // one output channel: fp32 dot products (k_conv_post)
            hipLaunchKernelGGL(k_conv_post, dim3(ceil_div(T * rate, 256), B), dim3(256), psm, st, prev, conv_post_w.as<float>(), conv_post.bias.as<float>(),
                               wav, cl, Tp * rate, T * rate);
        } else {
            ConvGemmArgs a = conv(conv_post, prev, B * Tp * rate, Tp * rate, T * rate, 0.01f);
            EpiTanhWav::Args e{wav, conv_post.bias.as<float>(), Tp * rate, T * rate};
            DSVC_TRY(voc_dispatch<EpiTanhWav>(a, e, prec, st));
        }
    }
    DSVC_HIP(hipGetLastError());
    return DSVC_OK;
}

extern "C" {

int dsvc_vocoder_create(const dsvc_vocoder_cfg* cfg, dsvc_vocoder** out) {
    if (!cfg || !out) return fail(DSVC_EINVAL, "null argument");
    if (cfg->precision < DSVC_PREC_F16 || cfg->precision > DSVC_PREC_F16_X3) return fail(DSVC_EINVAL, "unknown precision %d", cfg->precision);
    int ndev = 0;
    DSVC_HIP(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(DSVC_EHIP, "no HIP device visible");
    if (!(cfg->mel_scale > 0.f)) return fail(DSVC_EINVAL, "vocoder: mel_scale must be positive (2.30259 for log10 mels, 1 for natural-log mels)");
    dsvc_vocoder* v = new dsvc_vocoder();
    v->cfg = *cfg;
    *out = v;
    return DSVC_OK;
}

int dsvc_vocoder_load_tensor(dsvc_vocoder* v, const char* name, const float* host, int64_t numel) {
    if (!v || !name || !host || numel < 0) return fail(DSVC_EINVAL, "null argument");
    if (v->finalized) return fail(DSVC_ESTATE, "vocoder already finalized");
    v->host[name].assign(host, host + numel);
    return DSVC_OK;
}

int dsvc_vocoder_finalize(dsvc_vocoder* v) {
    if (!v) return fail(DSVC_EINVAL, "null handle");
    if (v->finalized) return DSVC_OK;
    return v->finalize();
}

void dsvc_vocoder_destroy(dsvc_vocoder* v) { delete v; }

int dsvc_vocode(dsvc_vocoder* v, const float* mel, const float* f0, float* wav, int32_t B, int32_t T, uint64_t seed,
                int32_t first_clip, const int32_t* clip_ids, void* stream) {
    if (!v || !mel || !wav) return fail(DSVC_EINVAL, "null argument");
    if (!v->finalized) return fail(DSVC_ESTATE, "vocoder not finalized");
    if (v->cfg.use_source && !f0) return fail(DSVC_EINVAL, "this generator has a harmonic source: f0 is required");
    return v->run(mel, f0, wav, B, T, seed, first_clip, clip_ids, (hipStream_t)stream);
}

}  // extern "C"
