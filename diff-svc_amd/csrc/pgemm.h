// pgemm.h -- the many-row contractions of the DiffNet training step on pre-split fp16 planes (gfx950 / CDNA4, wave64):
//     C[m][n] = EPI( sum_tap sum_k A[m + (tap - taps/2) * dil][k] * B[n][tap * K + k] )
// Reference: the Conv1d / ConvTranspose-shaped products of network/diff/net.py:58-84,120-135 and of their autograd (forward, data gradients).
//
// Round 2 ran these on the conv engine (conv_gemm.h): fp32 activations converted to fp16 hi|lo planes while they are staged, weights streamed as
// per-wave register fragments, one LDS buffer.  With ~8 700 rows and 256 ... 2 304 reduction channels those kernels sat at 150 ... 580 TFLOP/s
// MFMA-equivalent, and neither more workgroups nor larger chunks moved them (profiles/r3r / r3s_kernel_stats_train.csv).  Here the operands
// are fp16 hi|lo planes ALREADY (activation planes are written by the epilogue that produces the activation, weight planes once per step):
//   A  [2 planes][rows][lda]   row = frame (clip * Tp + t, zero gap rows = the convs' padding, 64 zero guard rows before and after)
//   B  [2 planes][N_pad][ldb]  row = output channel, k = tap * K + input channel
//   both in the fragment-tiled order of wgrad.h (pl_off: 32-row x 16-k pieces of 1 KiB)
// and the kernel is wgrad.h's: a 256 (rows) x 128 (columns) tile per workgroup, 8 waves of 64 x 64, three MFMAs per product, 32-deep stages
// DMA'd to LDS in fragment order (no swizzle, no conversion, no VGPR round trip), three stages in flight, one bare barrier per stage.  A
// conv tap is a row offset of the A pieces.  The accumulator layout equals conv_gemm's (lane & 31 = column inside a 32-column tile, the
// 16 registers = rows), so the epilogue functors of train.hip serve both engines.
#pragma once
#include "wgrad.h"

namespace dsvc {

struct PGemmArgs {
    const _Float16* a;          // A planes, hi plane, base of the tiled buffer: data row r is tiled row r + 64 (the guard)
    const _Float16* b;          // B planes, hi plane
    long long a_plane, b_plane; // halfs between the hi and the lo plane
    int lda, ldb;               // halfs per row (% 8 == 0)
    int n_rows;                 // rows with data (the epilogue skips the rest)
    int row_blocks, col_blocks; // (64 WR)-row / 128-column tiles
    int K, taps, dil;           // reduction channels per tap (% 32 == 0), taps (1 or 3), dilation
};

// WR = waves along the rows: 4 -> a 256 x 128 tile, 8 waves, three 48 KB stages (one workgroup per CU); 2 -> a 128 x 128 tile, 4 waves, three
// 32 KB stages (narrow outputs over ~8 700 rows: 256-row tiles would leave 60 % of the CUs without a workgroup)
template <class Epi, int WR>
__global__ void __launch_bounds__(128 * WR, WR == 4 ? 2 : 4) __attribute__((amdgpu_waves_per_eu(2, 2)))
pgemm_kernel(const PGemmArgs a, const typename Epi::Args ea) {
    constexpr int NWAVES = 2 * WR, A_PIECES = 8 * WR, PIECES = A_PIECES + 16, PER_WAVE = PIECES / NWAVES;     // 48 / 8 = 6, 32 / 4 = 8
    constexpr int STAGE_BYTES = PIECES * 1024;
    constexpr int STAGES = WR == 4 ? 3 : 2;                  // WR = 2: 2 x 32 KB, so that two workgroups (8 waves) share a CU
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef const half8 __attribute__((address_space(3))) * lds_frag_ptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo = wave >> 1, wk = wave & 1;                 // this wave's 64 x 64 corner of the tile
    // workgroup i runs on XCD i % 8: the column blocks of one row block share that row block's A pieces through ONE L2
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int rb = xcd + 8 * (j / a.col_blocks), cb = j % a.col_blocks;
    if (rb >= a.row_blocks) return;
    const int m0 = rb * (64 * WR), n0 = cb * 128;
    const int kst = a.K >> 5;                                // stages per tap
    const int stages = a.taps * kst;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    const int bkp = a.ldb >> 4;                                // B pieces per 32-row block
    auto dma = [&](int s) {
        char* dst = smem + (s % STAGES) * STAGE_BYTES;
        const int tap = s / kst, kk = (s - tap * kst) << 5;
        // a conv tap shifts the A rows: lane l's row of the piece (two contiguous runs in the tiled plane unless the shift is 0)
        const int arow = m0 + 64 + (tap - (a.taps >> 1)) * a.dil + (lane & 31);
        const int bk0 = tap * a.K + kk;
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int pc = wave + NWAVES * i;                  // piece 0 .. A_PIECES-1: A (tile r, plane p, k-step q), then 16 of B
            const _Float16* src;
            if (i < A_PIECES / NWAVES) {
                const int r = pc >> 2, p = (pc >> 1) & 1, q = pc & 1;
                src = a.a + (long long)p * a.a_plane + pl_off(arow + r * 32, kk + q * 16 + 8 * (lane >> 5), a.lda);
            } else {
                const int t = pc - A_PIECES, c = t >> 2, p = (t >> 1) & 1, q = t & 1;
                src = a.b + (long long)p * a.b_plane + ((long long)((n0 >> 5) + c) * bkp + ((bk0 >> 4) + q)) * 512 + lane * 8;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][q][r] = 0.f;

    dma(0);
    if (STAGES == 3 && stages > 1) dma(1);
    for (int s = 0; s < stages; ++s) {
        // vmcnt retires in order and a wave's only vector-memory traffic in the loop is its PER_WAVE DMA pieces per stage (wgrad.h): with
        // three buffers stage s + 1 may stay in flight, with two nothing else has been issued yet
        if (STAGES == 3 && s + 1 < stages) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + STAGES - 1 < stages) dma(s + STAGES - 1);
        const unsigned buf = lds0 + (unsigned)(s % STAGES) * STAGE_BYTES + (unsigned)lane * 16u;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            half8 fa[2][2], fb[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    fa[i][p] = *(lds_frag_ptr)(size_t)(buf + (unsigned)((((2 * wo + i) * 2 + p) * 2 + q) * 1024));
                    fb[i][p] = *(lds_frag_ptr)(size_t)(buf + (unsigned)((A_PIECES + ((2 * wk + i) * 2 + p) * 2 + q) * 1024));
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fb[c][0], acc[i][c], 0, 0, 0);
                    acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][1], fb[c][0], acc[i][c], 0, 0, 0);
                    acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fb[c][1], acc[i][c], 0, 0, 0);
                }
        }
    }
    // epilogue, conv_gemm's convention: register r of lane l = row 8 (r >> 2) + 4 (l >> 5) + (r & 3) of the 32-row tile, column l & 31 of the
    // 32-column tile; a wave's two column tiles are adjacent (PAIRED epilogues: gate | filter halves of the same 32 channels)
    Epi epi;
    const int ct0 = (n0 + wk * 64) >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wo * 64 + i * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
            if (row < a.n_rows) {
                if constexpr (Epi::PAIRED) {
                    epi.pair(ea, row, ct0, lane & 31, acc[i][0][r], acc[i][1][r]);
                } else {
                    epi.one(ea, row, ct0 * 32 + (lane & 31), acc[i][0][r]);
                    epi.one(ea, row, ct0 * 32 + 32 + (lane & 31), acc[i][1][r]);
                }
            }
        }
}

template <class Epi, int WR>
int pgemm_launch_wr(const PGemmArgs& a, const typename Epi::Args& e, hipStream_t st) {
    constexpr int smem = (WR == 4 ? 3 : 2) * (8 * WR + 16) * 1024;
    static bool attr_set = false;
    if (!attr_set) {
        DSVC_HIP(hipFuncSetAttribute((const void*)pgemm_kernel<Epi, WR>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    const int grid = 8 * ((a.row_blocks + 7) / 8) * a.col_blocks;
    hipLaunchKernelGGL((pgemm_kernel<Epi, WR>), dim3(grid), dim3(128 * WR), smem, st, a, e);
    DSVC_HIP(hipGetLastError());
    return DSVC_OK;
}

// rows_pad: the A planes' data rows rounded up to 256; the tile height is chosen so that the launch has >= ~200 workgroups when it can
template <class Epi>
int pgemm_launch(PGemmArgs a, int rows_pad, int n_cols, const typename Epi::Args& e, hipStream_t st) {
    if (a.K % 32 || a.lda % 16 || a.ldb % 16 || (a.taps != 1 && a.taps != 3) || rows_pad % 256 || n_cols < 1)
        return fail(DSVC_EINVAL, "pgemm: K %d, lda %d, ldb %d, taps %d, rows %d", a.K, a.lda, a.ldb, a.taps, rows_pad);
    a.col_blocks = (n_cols + 127) / 128;
    if ((rows_pad / 256) * a.col_blocks < 160) {
        a.row_blocks = rows_pad / 128;
        return pgemm_launch_wr<Epi, 2>(a, e, st);
    }
    a.row_blocks = rows_pad / 256;
    return pgemm_launch_wr<Epi, 4>(a, e, st);
}

// ---- operand planes -------------------------------------------------------------------------------------------------------------
// weights:  dst[plane][row][tap * K_pad + ci] = src[rowmap(row) * s_row + ci * s_ci + tap_of(tap) * s_tap] * scale   (0 outside n_rows / cin);
// rowmap: optional packed-row -> source-row permutation; flip: tap_of(tap) = taps - 1 - tap (transposed conv).  One thread per 8 k.
struct WPlaneDesc {
    const float* src; const int* rowmap; _Float16* dst; long long plane_halfs;
    int rows_pad, n_rows, taps, K_pad, cin;
    long long s_row, s_ci, s_tap;
    int flip; float scale;
};
__device__ __forceinline__ void wplanes_body(const WPlaneDesc& d, long long first, long long step) {
    const int ldb = d.taps * d.K_pad;
    const long long total = (long long)d.rows_pad * (ldb >> 3);
    for (long long idx = first; idx < total; idx += step) {
        const int row = (int)(idx / (ldb >> 3));
        const int k8 = (int)(idx - (long long)row * (ldb >> 3)) << 3;
        const int tap = k8 / d.K_pad, ci0 = k8 - tap * d.K_pad;
        const int srow = row < d.n_rows ? (d.rowmap ? d.rowmap[row] : row) : -1;
        half8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = ci0 + e;
            float w = 0.f;
            if (srow >= 0 && ci < d.cin)
                w = d.src[(long long)srow * d.s_row + (long long)ci * d.s_ci + (long long)(d.flip ? d.taps - 1 - tap : tap) * d.s_tap] * d.scale;
            hi[e] = (_Float16)w;
            lo[e] = (_Float16)(w - (float)hi[e]);
        }
        _Float16* p = d.dst + pl_off(row, k8, ldb);
        *reinterpret_cast<half8*>(p) = hi;
        *reinterpret_cast<half8*>(p + d.plane_halfs) = lo;
    }
}
// many weight tensors in one launch: blockIdx.y = descriptor
__global__ __launch_bounds__(256) void k_wplanes_batch(const WPlaneDesc* __restrict__ descs) {
    wplanes_body(descs[blockIdx.y], (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x);
}

// activations (dst = base of the tiled planes, data row r = tiled row r + 64):  dst[plane][row][c] = valid(row) ? src[row][c] + add[clip][c] : 0   for c < C (% 8 == 0); one thread per 8 channels
__global__ __launch_bounds__(256) void k_split_rows(const float* __restrict__ src, int ld_src, _Float16* __restrict__ dst, long long plane_halfs, int ldd,
                                                    int rows, int C, const float* __restrict__ add, int add_stride, int clip_stride, int clip_len, int n_valid) {
    const int c8n = C >> 3;
    const long long total = (long long)rows * c8n;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(idx / c8n), c = (int)(idx - (long long)row * c8n) << 3;
        const int clip = row / clip_stride;
        const bool ok = row < n_valid && row - clip * clip_stride < clip_len;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (ok) {
            const float4 x0 = *reinterpret_cast<const float4*>(src + (size_t)row * ld_src + c), x1 = *reinterpret_cast<const float4*>(src + (size_t)row * ld_src + c + 4);
            v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
            if (add) {
                const float* f = add + (size_t)clip * add_stride + c;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += f[e];
            }
        }
        half8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) { hi[e] = (_Float16)v[e]; lo[e] = (_Float16)(v[e] - (float)hi[e]); }
        _Float16* p = dst + pl_off(row + 64, c, ldd);
        *reinterpret_cast<half8*>(p) = hi;
        *reinterpret_cast<half8*>(p + plane_halfs) = lo;
    }
}

}  // namespace dsvc
