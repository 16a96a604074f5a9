// ttail.h -- the tail of a DDPM step in the throughput tiling as ONE kernel (gfx950 / CDNA4, wave64; round 5).
//
//   phase 1  skip projection   s2[C][64]  = relu(W_sp * (sum of skips / sqrt(L)) + b_sp)                        net.py:131-133   (K9a)
//   phase 2  output projection eps[M][64] = W_out * s2 + b_out, then the posterior step x <- p_sample(x, eps)    net.py:134, diffusion.py:131-163 (K9b + K10)
//   phase 3  input projection of the NEXT evaluation: x32 = relu(W_in * x + b_in), xh = fp16(x32 + film_0(t-1))  net.py:120-123 (K1 of step t-1)
//
// All three are 1x1 convolutions: a 64-frame tile never needs another tile's rows, so what the three tgemm launches of round 4 exchanged
// through HBM -- relu(skip projection) as fp16 hi | lo planes (3072 B per frame, written and read back) and the fp16 planes of the new
// state (1024 B) -- stays in LDS, and two kernel boundaries + two prologues per step go away.  258 -> 139 MB of HBM traffic per step at 32 clips.
// Operand scheme: hi + lo weight planes against [hi | lo] activation rows, three products (W_hi x_hi + W_lo x_hi + W_hi x_lo: fp32-class; the three
// launches also compute W_lo x_lo, 2^-22 of a product); the K loops run in plain order (the tgemm launches rotate their start group per tile).
// LDS (128 KB): [0, 96 KB) the skip-sum tile (DMA'd, source-side swizzle as tgemm.h), overwritten by the s2 tile after phase 1;
// [96 KB, 128 KB) the new state's [hi | lo] rows.
#pragma once
#include <type_traits>
#include "tlayer.h"

namespace dsvc {

struct TTailArgs {
    const _Float16* skiph;      // [rows][2 Cp] fp16 hi | lo planes of the skip sum (the last layer's epilogue wrote them)
    const _Float16 *wsp, *wout, *win;      // packed A fragments [m_tile][k16][plane 2][lane][8], K folded ([hi | lo] rows see the same weights)
    const float *bsp, *bout, *bin;
    int C, Cp, M, Mp;
    float* x;                   // [rows][M] sampler state (in/out)
    DdpmTables tab;
    StepRef step;               // the step of THIS evaluation (shared by all clips)
    RowMap rm;
    const unsigned long long* seedp;
    const int* clipid;
    float* x32;                 // residual stream of the next evaluation, accumulator-tiled
    _Float16* xh;               // layer 0's operand of the next evaluation, row 0 (guard rows precede)
    int ldh, xh_lo;             // halfs per xh row; > 0: rows are [hi | lo] planes, lo plane xh_lo halfs in
    const float* film;          // FiLM table of layer 0: film[step * film_step_stride + c]
    int film_step_stride;
#ifdef DSVC_PROFILING
    float* stamps;              // [workgroup][wave][8] shader-clock deltas at the phase boundaries (null: none)
#endif
};

// SMALL = false: 64-frame tiles x 8 waves (128 KB of LDS: one workgroup per CU).  SMALL = true: 32-frame tiles x 4 waves (64 KB: two workgroups
// per CU, whose serial phases -- tile DMA, three GEMMs with barriers between them, epilogues -- overlap each other's).
template <bool SMALL, int C, int S>
__global__ void __launch_bounds__(SMALL ? 256 : 512, 2) __attribute__((amdgpu_waves_per_eu(2, 2)))
ttail_kernel(const TTailArgs a) {
    constexpr int TT_TN = SMALL ? 32 : 64, WAVES = SMALL ? 4 : 8, NT = SMALL ? 1 : 2, NTH = 64 * WAVES;
    constexpr int KG = 4, NW = 2, GROUP_HALFS = KG * NW * TFRAG_HALFS;
    constexpr int MP = 128;                                 // padded mel bins (ttail_supported)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * TT_TN;
#ifdef DSVC_PROFILING
    const unsigned long long clk0 = __builtin_readcyclecounter();
    int stamp_i = 0;
#define TT_STAMP() do { if (a.stamps && lane == 0) a.stamps[((size_t)blockIdx.x * WAVES + wave) * 8 + stamp_i] = (float)(__builtin_readcyclecounter() - clk0); ++stamp_i; } while (0)
#else
#define TT_STAMP() do { } while (0)
#endif
    // a tile beyond its clip's length (ragged batch): nothing to compute -- its rows of the next evaluation's layer-0 operand are zero padding
    // (rewritten here: the bucket may hold an earlier, longer clip), everything else it would write is read by its own rows only (tlayer.h)
    if (a.rm.rowclip[row0] < 0) {
        const size_t n16 = (size_t)TT_TN * (size_t)a.ldh / 8;
        half8* dst = reinterpret_cast<half8*>(a.xh + (size_t)row0 * a.ldh);
        const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t i = tid; i < n16; i += NTH) dst[i] = z;
        return;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    constexpr int K1 = 2 * C, K3 = 2 * MP;                  // folded K of the skip / output projections and of the input projection
    constexpr unsigned rb1 = (unsigned)K1 * 2u, rb3 = (unsigned)K3 * 2u;  // LDS row bytes of the two tiles
    constexpr unsigned xs_off = (unsigned)TT_TN * rb1;      // the state tile sits behind the skip / s2 tile
    // ---- the skip-sum tile: HBM / L2 -> LDS by DMA, swizzled on the source side (tgemm.h) ----
    {
        const int chunks = K1 >> 3, total = TT_TN * chunks;
        const int dq = NTH / chunks, dr = NTH - dq * chunks;
        int slot = wave * 64 + lane;
        int r = slot / chunks, c = slot - r * chunks;
        const _Float16* xrow0 = a.skiph + (long long)row0 * K1;
        for (int it = wave; it * 64 < total; it += WAVES) {
            const int rc = r < TT_TN ? r : TT_TN - 1;
            const _Float16* src = xrow0 + (long long)rc * K1 + ((c ^ (rc & 15)) << 3);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(smem + it * 1024), 16, 0, 0);
            c += dr; r += dq;
            if (c >= chunks) { c -= chunks; r += 1; }
        }
    }
    const int lane8 = lane * 8;
    const int fr = lane & 31, h = lane >> 5;
    // The weight stream.  A "unit" is eight 1 KiB fragments = one ring stage.  The rows are [hi | lo] planes and the packed weights repeat themselves
    // over the lo half of K (tgemm's "fold"): the first G / 2 units are one group each, both weight planes -- (W_hi + W_lo) x_hi --, the last G / 4
    // units are the hi planes of TWO groups each -- W_hi x_lo (W_lo x_lo is 2^-22 of the product: not computed, not streamed; the three tgemm
    // launches this kernel replaces do compute it).  A wave has only 16 (or 8) MFMAs to issue per unit, far less than an L2 round trip, so the ring
    // is S stages deep and the first S units of the NEXT projection are requested before the barrier / epilogue between two phases; every loop
    // bound is a compile-time constant so that the compiler's s_waitcnt vmcnt(n) are exact (a conditional load would make them conservative).
    half8 ring[S][KG * NW];
    auto load_unit = [&](half8 (&r)[KG * NW], const _Float16* wt, int u, int Gh) {
        const long long base = u < Gh ? (long long)u * GROUP_HALFS : (long long)(2 * u - Gh) * GROUP_HALFS;
        const int fs = u < Gh ? TFRAG_HALFS : 2 * TFRAG_HALFS;
#pragma unroll
        for (int j = 0; j < KG * NW; ++j) r[j] = *reinterpret_cast<const half8*>(wt + base + j * fs + lane8);
    };
    auto prefetch = [&](const _Float16* wt, auto Gc) {
        constexpr int G = decltype(Gc)::value, Gh = G / 2, U = Gh + Gh / 2;
#pragma unroll
        for (int u = 0; u < (S < U ? S : U); ++u) load_unit(ring[u], wt, u, Gh);
    };
    // one output tile of a 1x1 projection over NT N-tiles starting at N-tile n0: K = 16 * KG * G, B operand rows of `rb` bytes at `tile0`
    auto gemm = [&](auto& acc, const _Float16* wt, auto Gc, unsigned tile0, unsigned rb, int n0, auto pre) {
        constexpr int NT = sizeof(acc) / sizeof(acc[0]);
        constexpr int G = decltype(Gc)::value, Gh = G / 2, U = Gh + Gh / 2;
        static_assert(Gh % 2 == 0, "the hi-only units pair two groups");
        const unsigned base0 = tile0 + (unsigned)(32 * n0 + fr) * rb;
        const unsigned xsw = (unsigned)(((fr & 15) ^ h) << 4);
        if constexpr (!decltype(pre)::value) prefetch(wt, Gc);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            half8 (&r)[KG * NW] = ring[u % S];
            if (u < Gh) tl_compute_group<KG, NW, NT>(r, acc, base0, 32u * rb, xsw ^ ((unsigned)u << 7));
            else tl_compute_group<KG * NW, 1, NT>(r, acc, base0, 32u * rb, xsw ^ ((unsigned)(2 * u - Gh) << 7));
            if (u + S < U) load_unit(r, wt, u + S, Gh);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    constexpr std::integral_constant<int, (K1 >> 4) / KG> G1c{};
    constexpr std::integral_constant<int, (K3 >> 4) / KG> G3c{};
    constexpr std::true_type PRE{};
    constexpr std::false_type COLD{};
    auto zero = [&](auto& acc) {
        constexpr int NT = sizeof(acc) / sizeof(acc[0]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
    };
    // 16 channels (cb ..) of frame row `f` of an [hi | lo] tile: chunk c of row f sits at slot c ^ (f & 15)
    auto put_hi_lo = [&](unsigned tile0, unsigned rb, int plane_chunks, int f, int cb, const float (&v)[16]) {
        half8 h0, h1, l0, l1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            h0[i] = (_Float16)v[i]; h1[i] = (_Float16)v[8 + i];
            l0[i] = (_Float16)(v[i] - (float)h0[i]); l1[i] = (_Float16)(v[8 + i] - (float)h1[i]);
        }
        const unsigned rowb = tile0 + (unsigned)f * rb, sw = (unsigned)(f & 15);
        const unsigned c0 = (unsigned)(cb >> 3);
        typedef half8 __attribute__((address_space(3))) * lp;
        *(lp)(size_t)(rowb + (((c0) ^ sw) << 4)) = h0;
        *(lp)(size_t)(rowb + (((c0 + 1) ^ sw) << 4)) = h1;
        *(lp)(size_t)(rowb + (((c0 + (unsigned)plane_chunks) ^ sw) << 4)) = l0;
        *(lp)(size_t)(rowb + (((c0 + 1 + (unsigned)plane_chunks) ^ sw) << 4)) = l1;
    };

    constexpr int n_sp = C >> 5;                            // output tiles of the skip projection (12 at C = 384, 8 at C = 256)
    constexpr int G1 = (K1 >> 4) / KG, G3 = (K3 >> 4) / KG;
    constexpr long long tile1 = (long long)G1 * GROUP_HALFS, tile3 = (long long)G3 * GROUP_HALFS;
    constexpr int split = SMALL ? 0 : n_sp - 8;             // 64-frame tiles at C = 384: a second pass of four tiles, one N-tile per wave
    constexpr int passes1 = SMALL ? n_sp / WAVES : 1;       // 32-frame tiles: n_sp / 4 passes of one tile per wave (2 or 3)
    // first weights in flight before the barrier
    prefetch(a.wsp + (long long)wave * tile1, G1c);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    TT_STAMP();                                            // 0: tile + first weights landed

    // =========================== phase 1: skip projection + ReLU, results held in registers ===========================
    // (each epilogue loads its biases, THEN requests the next weights, then does its arithmetic: the bias wait does not cover the prefetch)
    const int mt2 = wave & 3, nt2 = SMALL ? 0 : wave >> 2;  // phase 2: M <= 128 = 4 output tiles x NT N-tiles = one (tile, N-tile) per wave
    const bool act2 = mt2 * 32 < a.M;
    const _Float16* wout_w = a.wout + (long long)(act2 ? mt2 : 0) * tile1;       // (idle waves prefetch tile 0: the packed buffer has ceil(M / 32) tiles)
    float keep0[SMALL ? 3 : NT][16], keep1[16];
    if constexpr (SMALL) {
#pragma unroll
        for (int p = 0; p < passes1; ++p) {
            f32x16 acc[1];
            zero(acc);
            const int mt = p * WAVES + wave;
            gemm(acc, a.wsp + (long long)mt * tile1, G1c, lds0, rb1, 0, PRE);
            const int cb = mt * 32 + 16 * h;
            f32x4 b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) b[q] = ld4(a.bsp + cb + 4 * q);
            if (p + 1 < passes1) prefetch(a.wsp + (long long)(mt + WAVES) * tile1, G1c);
            else prefetch(wout_w, G1c);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) keep0[p][4 * q + i] = fmaxf(acc[0][4 * q + i] + b[q][i], 0.f);
        }
    } else {
        f32x16 acc[2];
        zero(acc);
        gemm(acc, a.wsp + (long long)wave * tile1, G1c, lds0, rb1, 0, PRE);
        const int cb = wave * 32 + 16 * h;
        f32x4 b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] = ld4(a.bsp + cb + 4 * q);
        if constexpr (split > 0) prefetch(a.wsp + (long long)(8 + (wave & 3)) * tile1, G1c);
        else prefetch(wout_w, G1c);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) keep0[nt][4 * q + i] = fmaxf(acc[nt][4 * q + i] + b[q][i], 0.f);
    }
    if constexpr (split > 0) {
        f32x16 acc[1];
        zero(acc);
        const int mt = 8 + (wave & 3);
        gemm(acc, a.wsp + (long long)mt * tile1, G1c, lds0, rb1, wave >> 2, PRE);
        const int cb = mt * 32 + 16 * h;
        f32x4 b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] = ld4(a.bsp + cb + 4 * q);
        prefetch(wout_w, G1c);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) keep1[4 * q + i] = fmaxf(acc[0][4 * q + i] + b[q][i], 0.f);
    }
    TT_STAMP();                                            // 1: this wave's skip projection done
    __syncthreads();                                       // every wave is done reading the skip-sum tile
    {
        constexpr int cp_chunks = C >> 3;
        if constexpr (SMALL) {
#pragma unroll
            for (int p = 0; p < passes1; ++p) put_hi_lo(lds0, rb1, cp_chunks, fr, (p * WAVES + wave) * 32 + 16 * h, keep0[p]);
        } else {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) put_hi_lo(lds0, rb1, cp_chunks, 32 * nt + fr, wave * 32 + 16 * h, keep0[nt]);
            if constexpr (split > 0) put_hi_lo(lds0, rb1, cp_chunks, 32 * (wave >> 2) + fr, (8 + (wave & 3)) * 32 + 16 * h, keep1);
        }
    }
    __syncthreads();
    TT_STAMP();                                            // 2: s2 parked

    // =========================== phase 2: output projection + posterior step; the new state -> HBM (fp32) and LDS ([hi | lo]) ===========================
    const int t = a.step.get(0);
    {
        f32x16 acc[1];
        zero(acc);
        if (act2) gemm(acc, wout_w, G1c, lds0, rb1, nt2, PRE);
        TT_STAMP();                                        // 3: output projection done
        prefetch(a.win + (long long)wave * tile3, G3c);     // (also on the chain's last step, where nobody uses them: an unconditional load keeps the wait counts exact)
        const int frame = row0 + 32 * nt2 + fr;
        const int cb = mt2 * 32 + 16 * h;
        const int clip = a.rm.rowclip[frame];
        float hv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) hv[i] = 0.f;
        if (act2 && cb < a.M && clip >= 0) {
            const int tl = frame - clip * a.rm.clip_stride;
            const float ra = a.tab.sqrt_recip_ac[t], rbb = a.tab.sqrt_recipm1_ac[t], c1 = a.tab.coef1[t], c2 = a.tab.coef2[t], sg = a.tab.sigma[t];
            float* px = a.x + (size_t)frame * a.M + cb;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 xt = ld4(px + 4 * q);
                const f32x4 b = ld4(a.bout + cb + 4 * q);
                float z[4] = {0.f, 0.f, 0.f, 0.f};
                if (t > 0) {
                    const unsigned el = (unsigned)tl * (unsigned)a.M + (unsigned)(cb + 4 * q);
                    philox_normal4(el >> 2, (unsigned)t, (unsigned)a.clipid[clip], PURPOSE_DDPM_NOISE, *a.seedp, z);
                }
                f32x4 out;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float eps = acc[0][4 * q + i] + b[i];
                    float x0 = ra * xt[i] - rbb * eps;
                    x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
                    float o = c1 * x0 + c2 * xt[i];
                    if (t > 0) o += sg * z[i];
                    out[i] = o;
                    hv[4 * q + i] = o;
                }
                st4(px + 4 * q, out);
            }
        }
        if (cb < MP) put_hi_lo(lds0 + xs_off, rb3, MP >> 3, 32 * nt2 + fr, cb, hv);      // zeros on gap rows and pad columns (every column of the tile is written)
    }
    TT_STAMP();                                            // 4: posterior step done
    if (t <= 0) return;                                    // the chain's last step: there is no next evaluation
    __syncthreads();
    TT_STAMP();                                            // 5: state tile visible

    // =========================== phase 3: the next evaluation's input projection + ReLU, x32 and layer 0's operand ===========================
    const int tn = t - 1;
    auto in_epi = [&](const f32x16& acc_nt, int mt, int nt) {
        const int cb = mt * 32 + 16 * h;
        const int frame = row0 + 32 * nt + fr;
        float v[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = ld4(a.bin + cb + 4 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[4 * q + i] = fmaxf(acc_nt[4 * q + i] + b[i], 0.f);
        }
        float* p = a.x32 + tiled_lane_base(row0 + 32 * nt, C >> 5, mt, lane);
#pragma unroll
        for (int q = 0; q < 4; ++q) st4(p + 256 * q, f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]});
        const bool ok = a.rm.rowclip[frame] >= 0;
        float hv[16];
        if (ok) {
            const float* fp = a.film + (size_t)tn * a.film_step_stride + cb;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 f = ld4(fp + 4 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i) hv[4 * q + i] = v[4 * q + i] + f[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) hv[i] = 0.f;
        }
        _Float16* q = a.xh + (size_t)frame * a.ldh + cb;
        if (a.xh_lo > 0) {
            store_hi_lo16(q, a.xh_lo, hv);
        } else {
            half8 o0, o1;
#pragma unroll
            for (int i = 0; i < 8; ++i) { o0[i] = (_Float16)hv[i]; o1[i] = (_Float16)hv[8 + i]; }
            *reinterpret_cast<half8*>(q) = o0;
            *reinterpret_cast<half8*>(q + 8) = o1;
        }
    };
    if constexpr (SMALL) {
#pragma unroll
        for (int p = 0; p < passes1; ++p) {
            f32x16 acc[1];
            zero(acc);
            const int mt = p * WAVES + wave;
            gemm(acc, a.win + (long long)mt * tile3, G3c, lds0 + xs_off, rb3, 0, PRE);
            if (p + 1 < passes1) prefetch(a.win + (long long)(mt + WAVES) * tile3, G3c);
            in_epi(acc[0], mt, 0);
        }
    } else {
        {
            f32x16 acc[2];
            zero(acc);
            gemm(acc, a.win + (long long)wave * tile3, G3c, lds0 + xs_off, rb3, 0, PRE);
            if constexpr (split > 0) prefetch(a.win + (long long)(8 + (wave & 3)) * tile3, G3c);
            in_epi(acc[0], wave, 0);
            in_epi(acc[1], wave, 1);
        }
        if constexpr (split > 0) {
            f32x16 acc[1];
            zero(acc);
            const int mt = 8 + (wave & 3);
            gemm(acc, a.win + (long long)mt * tile3, G3c, lds0 + xs_off, rb3, wave >> 2, PRE);
            in_epi(acc[0], mt, wave >> 2);
        }
    }
    TT_STAMP();                                            // 6: input projection done
}
#undef TT_STAMP

inline bool ttail_supported(int C, int Cp, int M, int Mp, int n_rows) {
    return (C == 384 || C == 256) && C == Cp && M <= 128 && Mp == 128 && M % 16 == 0 && n_rows % 64 == 0 && n_rows / 128 >= 48;
}

constexpr int TTAIL_STAGES = 3;       // ring depth (4 measured the same -- 84.5 vs 84.3 us at 32 clips -- and spills)

template <bool SMALL, int C>
inline int ttail_launch_t(const TTailArgs& a, int n_rows, hipStream_t stream) {
    constexpr int TN = SMALL ? 32 : 64;
    const size_t smem = (size_t)TN * (2 * C) * 2 + (size_t)TN * (2 * 128) * 2;
    static thread_local size_t smem_set = 0;
    if (smem > 64 * 1024 && smem > smem_set) {
        DSVC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ttail_kernel<SMALL, C, TTAIL_STAGES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        smem_set = smem;
    }
    hipLaunchKernelGGL((ttail_kernel<SMALL, C, TTAIL_STAGES>), dim3(n_rows / TN), dim3(SMALL ? 256 : 512), smem, stream, a);
    DSVC_HIP(hipGetLastError());
    return DSVC_OK;
}

// small_tiles: 32-frame tiles x 4 waves instead of 64-frame tiles x 8 waves (bit-identical results; the caller picks by workgroups per CU)
inline int ttail_launch(const TTailArgs& a, int n_rows, hipStream_t stream, bool small_tiles) {
    if (!ttail_supported(a.C, a.Cp, a.M, a.Mp, n_rows)) return fail(DSVC_EINVAL, "ttail: shape not supported by the fused step tail");
    if (a.C == 384) return small_tiles ? ttail_launch_t<true, 384>(a, n_rows, stream) : ttail_launch_t<false, 384>(a, n_rows, stream);
    return small_tiles ? ttail_launch_t<true, 256>(a, n_rows, stream) : ttail_launch_t<false, 256>(a, n_rows, stream);
}

}  // namespace dsvc
