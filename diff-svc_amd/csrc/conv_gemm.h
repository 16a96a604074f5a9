// conv_gemm.h -- the one MFMA engine behind every dense contraction on the diff-svc hot path.
//
//   out[r][co] = EPI( sum_{tap} sum_{ci}  W[co][tap][ci] * PRE(x)[r + (tap - taps/2)*dil][ci] )
//
// covers the DiffNet 1x1 projections and the dilated k=3 conv (network/diff/net.py:58-135), the
// hoisted conditioner projection, and every Conv1d / polyphase ConvTranspose1d of the NSF-HiFiGAN
// generator (modules/nsf_hifigan/models.py:33-64,325-387).
//
// Design (gfx950 / CDNA4, wave64):
//   * activations are FRAME-MAJOR fp32 in HBM: row = one time frame, channels contiguous, so a frame is
//     one coalesced burst and a conv tap is a row offset.  Clips of a batch are laid along the row axis
//     in slots of `clip_stride` rows of which `clip_len` are valid; gap rows read as zero, which is
//     exactly the convs' zero padding, so one launch serves any batch with no per-clip logic.
//   * a workgroup stages a (TM + 2*halo) x KCB time tile ONCE in LDS (fp32 -> fp16 hi [+ lo] planes,
//     row stride KCB+8 halfs = odd multiple of 16 B -> conflict-free ds_read_b128) and every tap
//     re-reads it at a shifted row: the dilated conv costs one HBM/L2 read of x, not three.
//   * weights never touch LDS: they are packed on the host in MFMA *fragment order*
//     [col-tile 32][tap][k16][plane][lane 64][8 halfs], so a wave's B operand is one fully coalesced
//     1 KiB global_load_dwordx4 straight into VGPRs, prefetched PF steps ahead through a register ring.
//   * math: v_mfma_f32_32x32x16_f16, fp32 accumulate.  fp16 operands alone miss the reference by ~1e-2
//     after a 1000-step chain (weight rounding is systematic), so operands can be split
//     w = w_hi + w_lo (NW=2) and x = x_hi + x_lo (NA=2):  acc += xh*wh [+ xh*wl] [+ xl*wh].
//   * WAVES_K > 1 splits the reduction across the waves (SIMDs) of a workgroup and combines through
//     LDS: this is what keeps a B=1 clip (only ~900 frames) from leaving 3/4 of every CU idle.
//   * the launch is a 1-D grid remapped so that all column groups of one time tile run on the same
//     XCD (block b runs on XCD b%8): the x tile is fetched into one L2 instead of eight.
#pragma once
#include "common.h"

namespace dsvc {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvGemmArgs {
    const float* x;        // [n_rows][ldx] fp32 frame-major
    int ldx;
    int n_rows;            // batch * clip_stride
    int clip_stride;       // rows per clip slot (>= 32)
    int clip_len;          // valid rows per slot
    const int* clip_lens;  // optional device [n_rows/clip_stride]: per-clip valid rows (overrides clip_len)
    int cin;               // input channels, multiple of KCB
    int taps, dil;         // tap offset = (tap - taps/2) * dil rows
    const _Float16* w;     // fragment-packed weights
    int n_ctiles;          // number of 32-wide output column tiles (even)
    int w_planes;          // planes stored per fragment in `w` (1 or 2); the kernel uses the first NW
    const float* film;     // optional per-channel add on valid rows: film[step*film_step_stride + c]
    const int* step_ptr;   // optional device int(s) holding the current diffusion step
    int step_off;          // step = step_ptr[clip*step_per_clip] - step_off
    int step_per_clip;     // 0: one shared scalar, 1: one entry per clip (DiffNet.forward's t[B])
    int film_step_stride;
    float in_slope;        // leaky-relu slope applied to the staged input (1 = identity)
    int k_slices;          // > 1: the reduction (cin) is cut into k_slices ranges of k_slice_len channels over blockIdx.y; every slice
    int k_slice_len;       //      runs its own epilogue (which tells slices apart by blockIdx.y): weight-gradient GEMMs (train.hip)
    int nz;                // > 1: blockIdx.z walks nz independent problems of one shape -- x += z * x_z floats, w += z * w_z halfs; the epilogue tells
    long long x_z, w_z;    //      them apart by blockIdx.z (hubert.hip: the heads of an attention layer, the groups of the positional conv, in ONE launch)
};

// number of halfs of one packed [lane 64][8] fragment
constexpr int FRAG_HALFS = 512;

__device__ __forceinline__ int clip_local(int r, int stride) { return r - (r / stride) * stride; }

// latency tilings want two waves per SIMD resident (two 4-wave workgroups, or one 8-wave one)
constexpr int min_waves_per_simd(int wm_tiles, int waves) { return wm_tiles <= 2 ? (waves > 8 ? (waves + 3) / 4 : 2) : 1; }

// latency tilings (one 32-frame tile) cap their registers at 256 so two workgroups share a CU; the
// 128-frame throughput tiling needs 128 accumulators + staging and runs one wave per SIMD.
// WAVES_M > 1: several waves take consecutive 32*WM_TILES-row slabs of ONE staged time tile (more waves per CU for the same halo and LDS
// planes: the vocoder's narrow stages, where a one-wave workgroup leaves the CU with under one wave per SIMD).
template <int WM_TILES, int WAVES_N, int WAVES_K, int KCB, int PF, int SPT, int NW, int NA, class Epi, int WAVES_M = 1>
__global__ void __launch_bounds__(64 * WAVES_N * WAVES_K * WAVES_M, min_waves_per_simd(WM_TILES, WAVES_N * WAVES_K * WAVES_M))
conv_gemm_kernel(const ConvGemmArgs a, const typename Epi::Args ea) {
    constexpr int NT = 64 * WAVES_N * WAVES_K * WAVES_M;
    constexpr int TM = 32 * WM_TILES * WAVES_M;
    static_assert(WAVES_M == 1 || WAVES_K == 1, "row-split and k-split waves are not combined");
    constexpr int TN = 64 * WAVES_N;
    constexpr int KCW = KCB / WAVES_K;   // channels of a staged chunk owned by one k-slice wave
    constexpr int KS = KCW / 16;         // k16 steps per tap per chunk per wave
    constexpr int XS = KCB + 8;          // LDS row stride in halfs
    static_assert(KCW % 16 == 0, "k-slice must be a multiple of 16 channels");
    static_assert(KS % PF == 0, "prefetch depth must divide the k16 steps per tap");
    static_assert(SPT >= 1 && SPT <= 32, "staging batch");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* xs = reinterpret_cast<_Float16*>(smem);

    // ---- block -> (time tile, column group), XCD-aware: block b is dispatched to XCD b % 8 ----
    const int ncg = (a.n_ctiles + 2 * WAVES_N - 1) / (2 * WAVES_N);
    const int nrt = (a.n_rows + TM - 1) / TM;
    // (k-sliced launches rotate the XCD of a row tile with the slice: a GEMM with fewer than 8 row tiles would otherwise keep every
    //  slice of a row tile -- all of its workgroups -- on the same few XCDs; gridDim.x is a multiple of 8, so block (x, y) runs on XCD x % 8)
    const int xcd = ((int)(blockIdx.x & 7) + (a.k_slices > 1 ? (int)blockIdx.y : 0)) & 7, slot = blockIdx.x >> 3;
    const int rt = xcd + 8 * (slot / ncg);
    const int cg = slot % ncg;
    if (rt >= nrt) return;
    const int row0 = rt * TM;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WAVES_N;
    const int kz = (wave / WAVES_N) % WAVES_K;
    const int wm = wave / (WAVES_N * WAVES_K);          // row slab of this wave inside the staged tile
    const int ct0 = (cg * WAVES_N + wn) * 2;          // first of this wave's two column tiles
    const bool active = ct0 < a.n_ctiles;

    const int c_begin = a.k_slices > 1 ? (int)blockIdx.y * a.k_slice_len : 0;
    const int c_end = a.k_slices > 1 ? (c_begin + a.k_slice_len < a.cin ? c_begin + a.k_slice_len : a.cin) : a.cin;
    const int halo = (a.taps / 2) * a.dil;
    const int rows_lds = TM + 2 * halo;
    const int plane_halfs = rows_lds * XS;
    const int nk16 = a.cin >> 4;

    const float* film = nullptr;             // shared-step case: hoisted once
    if (a.film && !a.step_per_clip) {
        const int step = a.step_ptr ? (*a.step_ptr - a.step_off) : 0;
        film = a.film + (size_t)step * a.film_step_stride;
    }

    f32x16 acc[WM_TILES][2];
#pragma unroll
    for (int m = 0; m < WM_TILES; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // ---- weight fragment ring ----
    half8 bring[PF][2][NW];
    const int wpl = a.w_planes;
    const size_t tile_halfs = (size_t)a.taps * nk16 * wpl * FRAG_HALFS;
    const _Float16* wbase = a.w + (size_t)blockIdx.z * a.w_z + (size_t)ct0 * tile_halfs + lane * 8;
    auto wload = [&](half8 (&dst)[2][NW], int c0, int tap, int ks) {
        const int k16 = ((c0 + kz * KCW) >> 4) + ks;
        const _Float16* p = wbase + ((size_t)tap * nk16 + k16) * (wpl * FRAG_HALFS);
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int q = 0; q < NW; ++q)
                dst[n][q] = *reinterpret_cast<const half8*>(p + n * tile_halfs + q * FRAG_HALFS);
    };
    if (active) {
#pragma unroll
        for (int u = 0; u < PF; ++u) wload(bring[u], c_begin, 0, u);     // PF <= KS, so these are tap 0 of the first chunk
    }

    const int arow = (lane & 31);
    const int acol = 8 * (lane >> 5);

    // ---- staging: global fp32 -> registers (all loads of a batch in flight together) -> fp16 planes in LDS ----
    constexpr int IPR = KCB / 8;                 // 8-channel items per row
    const int items = rows_lds * IPR;
    float* film_lds = reinterpret_cast<float*>(smem + (((size_t)NA * plane_halfs * 2 + 15) & ~(size_t)15));
    if (film) {                                  // shared diffusion step: park the FiLM vector in LDS once
        for (int c = tid * 4; c < a.cin; c += NT * 4)
            *reinterpret_cast<float4*>(film_lds + c) = *reinterpret_cast<const float4*>(film + c);
    }
    auto stage_load = [&](int c0, int b0, float4 (&sr)[SPT][2], unsigned& vmask) {
        vmask = 0;
#pragma unroll
        for (int u = 0; u < SPT; ++u) {
            const int it = b0 + tid + u * NT;
            sr[u][0] = make_float4(0.f, 0.f, 0.f, 0.f);
            sr[u][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (it < items) {
                const int rr = it / IPR;
                const int c8 = (it - rr * IPR) * 8;
                const int r = row0 - halo + rr;
                bool valid = (r >= 0) && (r < a.n_rows);
                if (valid) {
                    const int clip = r / a.clip_stride;
                    valid = (r - clip * a.clip_stride) < (a.clip_lens ? a.clip_lens[clip] : a.clip_len);
                }
                if (valid) {
                    const float4* src = reinterpret_cast<const float4*>(a.x + (size_t)blockIdx.z * a.x_z + (size_t)r * a.ldx + c0 + c8);
                    sr[u][0] = src[0];
                    sr[u][1] = src[1];
                    vmask |= 1u << u;
                }
            }
        }
    };
    auto stage_store = [&](int c0, int b0, const float4 (&sr)[SPT][2], unsigned vmask) {
#pragma unroll
        for (int u = 0; u < SPT; ++u) {
            const int it = b0 + tid + u * NT;
            if (it < items) {
                const int rr = it / IPR;
                const int c8 = (it - rr * IPR) * 8;
                float v[8] = {sr[u][0].x, sr[u][0].y, sr[u][0].z, sr[u][0].w, sr[u][1].x, sr[u][1].y, sr[u][1].z, sr[u][1].w};
                if ((vmask >> u) & 1u) {
                    if (a.film) {
                        float4 f0, f1;
                        if (film) {
                            f0 = *reinterpret_cast<const float4*>(film_lds + c0 + c8);
                            f1 = *reinterpret_cast<const float4*>(film_lds + c0 + c8 + 4);
                        } else {                 // one step per clip (DiffNet.forward's t[B]): rare path, from global
                            const int clip = (row0 - halo + rr) / a.clip_stride;
                            const float* fr = a.film + (size_t)(a.step_ptr[clip] - a.step_off) * a.film_step_stride + c0 + c8;
                            f0 = *reinterpret_cast<const float4*>(fr);
                            f1 = *reinterpret_cast<const float4*>(fr + 4);
                        }
                        v[0] += f0.x; v[1] += f0.y; v[2] += f0.z; v[3] += f0.w;
                        v[4] += f1.x; v[5] += f1.y; v[6] += f1.z; v[7] += f1.w;
                    }
                    if (a.in_slope != 1.0f) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * a.in_slope;
                    }
                }
                half8 hi;
#pragma unroll
                for (int j = 0; j < 8; ++j) hi[j] = (_Float16)v[j];
                *reinterpret_cast<half8*>(xs + rr * XS + c8) = hi;
                if constexpr (NA == 2) {
                    half8 lo;
#pragma unroll
                    for (int j = 0; j < 8; ++j) lo[j] = (_Float16)(v[j] - (float)hi[j]);
                    *reinterpret_cast<half8*>(xs + plane_halfs + rr * XS + c8) = lo;
                }
            }
        }
    };

    float4 sreg[SPT][2];
    unsigned vmask = 0;
    stage_load(c_begin, 0, sreg, vmask);             // first batch of the first chunk in flight beside the weight ring
    if (film) __syncthreads();                       // film_lds visible before the first store phase

    for (int c0 = c_begin; c0 < c_end; c0 += KCB) {
        if (c0 > c_begin) __syncthreads();           // previous chunk fully consumed
        stage_store(c0, 0, sreg, vmask);
        for (int b0 = NT * SPT; b0 < items; b0 += NT * SPT) {     // tile larger than the register batch (big halo)
            stage_load(c0, b0, sreg, vmask);
            stage_store(c0, b0, sreg, vmask);
        }
        __syncthreads();
        if (c0 + KCB < c_end) stage_load(c0 + KCB, 0, sreg, vmask);   // next chunk's loads fly under this chunk's MFMAs

        if (active) {
            for (int tap = 0; tap < a.taps; ++tap) {
                const int roff = halo + (tap - a.taps / 2) * a.dil;
                const _Float16* xrow = xs + (roff + wm * (32 * WM_TILES) + arow) * XS + kz * KCW + acol;
#pragma unroll
                for (int ksg = 0; ksg < KS; ksg += PF) {
#pragma unroll
                    for (int u = 0; u < PF; ++u) {
                        const int ks = ksg + u;
                        // A fragments (activations) for this k16 step
                        half8 ah[WM_TILES], al[WM_TILES];
#pragma unroll
                        for (int m = 0; m < WM_TILES; ++m) {
                            ah[m] = *reinterpret_cast<const half8*>(xrow + m * 32 * XS + ks * 16);
                            if constexpr (NA == 2)
                                al[m] = *reinterpret_cast<const half8*>(xrow + plane_halfs + m * 32 * XS + ks * 16);
                        }
                        half8 bcur[2][NW];
#pragma unroll
                        for (int n = 0; n < 2; ++n)
#pragma unroll
                            for (int q = 0; q < NW; ++q) bcur[n][q] = bring[u][n][q];
                        // refill this ring slot with the fragment PF steps ahead
                        {
                            int nks = ks + PF, ntap = tap, nc0 = c0;
                            if (nks >= KS) { nks -= KS; ntap += 1; }
                            if (ntap >= a.taps) { ntap = 0; nc0 += KCB; }
                            if (nc0 < c_end) wload(bring[u], nc0, ntap, nks);
                        }
#pragma unroll
                        for (int m = 0; m < WM_TILES; ++m)
#pragma unroll
                            for (int n = 0; n < 2; ++n) {
                                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bcur[n][0], acc[m][n], 0, 0, 0);
                                if constexpr (NW == 2)
                                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bcur[n][1], acc[m][n], 0, 0, 0);
                                if constexpr (NA == 2)
                                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bcur[n][0], acc[m][n], 0, 0, 0);
                            }
                    }
                }
            }
        }
    }

    // ---- epilogue ----
    Epi epi;
    if constexpr (WAVES_K == 1) {
        if (!active) return;
#pragma unroll
        for (int m = 0; m < WM_TILES; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + (wm * WM_TILES + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < a.n_rows) {
                    if constexpr (Epi::PAIRED) {
                        epi.pair(ea, row, ct0, lane & 31, acc[m][0][r], acc[m][1][r]);
                    } else {
                        epi.one(ea, row, ct0 * 32 + (lane & 31), acc[m][0][r]);
                        epi.one(ea, row, ct0 * 32 + 32 + (lane & 31), acc[m][1][r]);
                    }
                }
            }
        }
    } else {
        // split-K: every k-slice wave parks its partial tile in LDS, then all threads combine and
        // run the epilogue with a row-contiguous (coalesced) thread->element mapping.
        __syncthreads();                              // everyone done reading the x tile
        float* part = reinterpret_cast<float*>(smem); // [WAVES_K][TM][TN]
        if (active) {
#pragma unroll
            for (int m = 0; m < WM_TILES; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int i = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        part[(kz * TM + i) * TN + wn * 64 + n * 32 + (lane & 31)] = acc[m][n][r];
                    }
        }
        __syncthreads();
        constexpr int EPT = Epi::PAIRED ? (TM * TN / 2) : (TM * TN);
        constexpr int CW = Epi::PAIRED ? (TN / 2) : TN;     // epilogue elements per row
        for (int e = tid; e < EPT; e += NT) {
            const int i = e / CW;
            const int jc = e - i * CW;
            const int row = row0 + i;
            if (row >= a.n_rows) continue;
            if constexpr (Epi::PAIRED) {
                const int w2 = jc >> 5, j = jc & 31;          // wave-column, lane column
                const int ct = (cg * WAVES_N + w2) * 2;
                if (ct >= a.n_ctiles) continue;
                float v0 = 0.f, v1 = 0.f;
#pragma unroll
                for (int z = 0; z < WAVES_K; ++z) {
                    v0 += part[(z * TM + i) * TN + w2 * 64 + j];
                    v1 += part[(z * TM + i) * TN + w2 * 64 + 32 + j];
                }
                epi.pair(ea, row, ct, j, v0, v1);
            } else {
                const int ctile = cg * WAVES_N * 2 + (jc >> 5);
                if (ctile >= a.n_ctiles) continue;
                float v = 0.f;
#pragma unroll
                for (int z = 0; z < WAVES_K; ++z) v += part[(z * TM + i) * TN + jc];
                epi.one(ea, row, ctile * 32 + (jc & 31), v);
            }
        }
    }
}

// LDS bytes a launch needs (x tile planes, or the split-K partial tiles, whichever is larger)
template <int WM_TILES, int WAVES_N, int WAVES_K, int KCB, int NA, int WAVES_M = 1>
inline size_t conv_gemm_smem(int taps, int dil, int cin) {
    const int halo = (taps / 2) * dil;
    size_t x = (size_t)NA * (32 * WM_TILES * WAVES_M + 2 * halo) * (KCB + 8) * sizeof(_Float16);
    x = ((x + 15) & ~(size_t)15) + (size_t)cin * sizeof(float);          // + the FiLM vector
    size_t p = WAVES_K > 1 ? (size_t)WAVES_K * 32 * WM_TILES * 64 * WAVES_N * sizeof(float) : 0;
    return x > p ? x : p;
}

template <int WM_TILES, int WAVES_N, int WAVES_K, int KCB, int PF, int SPT, int NW, int NA, class Epi, int WAVES_M = 1>
inline int conv_gemm_launch(const ConvGemmArgs& a, const typename Epi::Args& ea, hipStream_t stream) {
    if (a.cin % KCB != 0) return fail(DSVC_EINVAL, "conv_gemm: cin %d not a multiple of the staged chunk %d", a.cin, KCB);
    if (a.k_slices > 1 && (a.k_slice_len % KCB != 0 || (long long)a.k_slices * a.k_slice_len < a.cin || a.film))
        return fail(DSVC_EINVAL, "conv_gemm: %d k-slices of %d channels do not tile cin %d in chunks of %d", a.k_slices, a.k_slice_len, a.cin, KCB);
    if (a.w_planes < NW) return fail(DSVC_EINVAL, "conv_gemm: weights packed with %d plane(s), kernel needs %d", a.w_planes, NW);
    if (a.ldx % 4 != 0) return fail(DSVC_EINVAL, "conv_gemm: ldx %d not a multiple of 4", a.ldx);
    if (a.n_ctiles & 1) return fail(DSVC_EINVAL, "conv_gemm: odd column-tile count %d", a.n_ctiles);
    if (a.clip_stride < 32) return fail(DSVC_EINVAL, "conv_gemm: clip_stride %d < 32", a.clip_stride);
    auto kern = conv_gemm_kernel<WM_TILES, WAVES_N, WAVES_K, KCB, PF, SPT, NW, NA, Epi, WAVES_M>;
    const size_t smem = conv_gemm_smem<WM_TILES, WAVES_N, WAVES_K, KCB, NA, WAVES_M>(a.taps, a.dil, a.cin);
    if (smem > 160 * 1024) return fail(DSVC_EINVAL, "conv_gemm: %zu B of LDS requested", smem);
    static thread_local size_t smem_set = 0;
    if (smem > 64 * 1024 && smem > smem_set) {
        DSVC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        smem_set = smem;
    }
    const int ncg = ceil_div(a.n_ctiles, 2 * WAVES_N);
    const int nrt = ceil_div(a.n_rows, 32 * WM_TILES * WAVES_M);
    const int grid = round_up(nrt, 8) * ncg;
    if (a.nz > 1 && (a.x_z % 4 != 0 || a.w_z % 8 != 0)) return fail(DSVC_EINVAL, "conv_gemm: batch strides %lld / %lld break the 16-byte accesses", a.x_z, a.w_z);
    hipLaunchKernelGGL(kern, dim3(grid, a.k_slices > 1 ? a.k_slices : 1, a.nz > 1 ? a.nz : 1), dim3(64 * WAVES_N * WAVES_K * WAVES_M), smem, stream, a, ea);
    DSVC_HIP(hipGetLastError());
    return DSVC_OK;
}

// ---------------------------------------------------------------------------------------------
// Host-side packing of a conv weight into MFMA fragment order.
//   src(co, tap, ci) -> float; output layout [ctile][tap][k16][plane][lane][8]:
//   lane l of fragment (ctile, tap, k16) holds W[co = ctile*32 + (l&31)][tap][ci = k16*16 + 8*(l>>5) + e]
//   plane 0 = fp16(w), plane 1 = fp16(w - plane0).  Rows >= cout / channels >= cin are zero.
// ---------------------------------------------------------------------------------------------
template <class F>
inline void pack_fragments(_Float16* dst, int n_ctiles, int taps, int cin_pad, int planes, F&& src) {
    const int nk16 = cin_pad / 16;
    for (int ct = 0; ct < n_ctiles; ++ct)
        for (int tap = 0; tap < taps; ++tap)
            for (int k = 0; k < nk16; ++k) {
                _Float16* f = dst + (((size_t)ct * taps + tap) * nk16 + k) * planes * FRAG_HALFS;
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 8; ++e) {
                        const float w = src(ct * 32 + (l & 31), tap, k * 16 + 8 * (l >> 5) + e);
                        const _Float16 hi = (_Float16)w;
                        f[l * 8 + e] = hi;
                        if (planes == 2) f[FRAG_HALFS + l * 8 + e] = (_Float16)(w - (float)hi);
                    }
            }
}

inline size_t packed_halfs(int n_ctiles, int taps, int cin_pad, int planes) {
    return (size_t)n_ctiles * taps * (cin_pad / 16) * planes * FRAG_HALFS;
}

}  // namespace dsvc
