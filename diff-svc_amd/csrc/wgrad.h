// wgrad.h -- weight gradients of the DiffNet training step (gfx950 / CDNA4, wave64):  dW[o][k] = sum_n dY[n][o] * X[n][k]
// Reference: the autograd of network/diff/net.py:58-135 under network/diff/diffusion.py:207-225 (every Conv1d / Linear weight of
// the denoiser); the contraction index n is the frame index (~8 000 per step at the 64 x 128-frame batch).
//
// Both operands are produced frame-major ([n][channel] fp32) by the forward / data-gradient kernels, but an MFMA fragment wants the
// contraction index contiguous per lane.  Round 2 ran this contraction on the conv engine: dY^T repacked into weight fragments
// (k_pack_w), X^T as an fp32 transposed copy, fp32 -> hi|lo conversion at staging, 32 x 64 output tiles: 41 % of the training step.
// Round 3 (still the path of the small projections at both ends of the network and of channel counts that are not multiples of 128):
//   * k_split_t  writes an operand ONCE as channel-major fp16 planes  T[plane hi|lo][channel][n]  (v = hi + lo, fp32-class), with the
//                validity mask, the FiLM shift of the dilated conv's input and the conv tap's frame shift folded in.
//   * wgrad_nt_kernel  is then a plain "NT" GEMM on those planes: a 256 (o) x 128 (k) output tile per workgroup, 8 waves of 64 x 64,
//                three MFMAs per product (hi*hi + lo*hi + hi*lo).  The frame range is cut into slices (a handful of output tiles would
//                not fill 256 CUs) whose partial tiles a second kernel adds in a fixed order (deterministic); every XCD gets whole slices.
//                Operands are staged 32 frames at a time by LDS-DMA (global_load_lds_dwordx4) in FRAGMENT order: lane l of a 1 KiB
//                piece loads row (l & 31), frames 8 (l >> 5) .. +7 and the DMA drops it at piece + 16 l -- exactly what ds_read_b128 at
//                lane*16 hands to the MFMA, so there is no swizzle and no bank conflict.  Three stages (144 KB) in flight, one bare
//                s_barrier per stage, counted vmcnt.
// Round 5 (the residual layers, 97 % of the weight-gradient FLOPs):
//   * wgrad_fm_kernel  contracts the frame-major fp16 [hi | lo] planes the layer kernels write anyway -- no k_split_t pass, no copies: the
//                stage is DMA'd as the planes lie and ds_read_b64_tr_b16 transposes on the way out of LDS (further down).
#pragma once
#include "conv_gemm.h"

namespace dsvc {

// ---- fragment-tiled planes ------------------------------------------------------------------------------------------------------------
// Every fp16 operand plane of wgrad.h is stored in the order an MFMA fragment wants it: a [R rows][K] plane is cut into pieces of
// 32 rows x 16 k, piece (row / 32, k / 16) is 1 KiB and holds, for lane l = (row % 32) + 32 * ((k / 8) % 2), the 8 halves k % 8 = 0..7.
// A piece is then ONE contiguous 1 KiB global_load_lds per wave (a row-major plane made that instruction touch 32 cache lines for 32 bytes
// each: 17 GB/s per CU, profiles/r3t_kernel_stats_train.csv), and a row-shifted piece (a conv tap) is two contiguous runs.
__host__ __device__ __forceinline__ size_t pl_off(int row, int k, int K) {
    return ((size_t)(row >> 5) * (size_t)(K >> 4) + (size_t)(k >> 4)) * 512 + (size_t)((((row & 31) + 32 * ((k >> 3) & 1)) << 3) + (k & 7));
}

// ---- operand planes -------------------------------------------------------------------------------------------------------------
// The planes hold the REAL frames only (the gap rows between clips would be 20 % zeros in the contraction): plane column m = clip * clip_len + t.
// n_taps = 1:  dst[plane][c][m] = (src[clip * clip_stride + t][c] + add[clip][c]) * scale                                  for m in [0, n_out)
// n_taps = 3:  the three taps of a dilated conv's input in one pass over src: tap j goes to plane rows + j * tap_rows and holds the frame
//              t + (j - 1) * dil of the SAME clip, or the conv's zero padding when that leaves the clip.
// colsum (n_taps = 1): colsum[c] += sum_m of the values written (a bias gradient: dY is read here anyway).
// Columns past the last real frame are written as zeros.  256 threads; n_out % 64 == 0, dil <= 64; channels >= C of a padded plane are left as
// they are (callers ignore them).
struct SplitRows { int clip_stride, clip_len, n_clips; };

// dst = the planes' base + first_row * ldT (first_row % 32 == 0); tap_halfs = tap row offset * ldT.  A block handles 64 frames x 64 channels:
// 16-byte loads along the channels (256 B per row), 16-byte stores along the frames.  grid (n_out / 64, ceil(C / 64)); C % 4 == 0.
__global__ __launch_bounds__(256) void k_split_t(const float* __restrict__ src, int ld_src, _Float16* __restrict__ dst, long long plane_halfs,
                                                 int ldT, int C, const float* __restrict__ add, int add_stride, SplitRows ri, int n_taps, int dil,
                                                 long long tap_halfs, float scale, float* __restrict__ colsum) {
    __shared__ float tile[64 + 128][68];
    const int n0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int halo = n_taps == 3 ? dil : 0;
    const int n_real = ri.n_clips * ri.clip_len;
    {
        const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4, c = c0 + 4 * tx;
        for (int i = ty; i < 64 + 2 * halo; i += 16) {
            const int m = n0 - halo + i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m >= 0 && m < n_real && c < C) {
                const int clip = m / ri.clip_len, t = m - clip * ri.clip_len;
                v = *reinterpret_cast<const float4*>(src + ((size_t)clip * ri.clip_stride + t) * ld_src + c);
                if (add) {
                    const float4 f = *reinterpret_cast<const float4*>(add + (size_t)clip * add_stride + c);
                    v.x += f.x; v.y += f.y; v.z += f.z; v.w += f.w;
                }
                v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
            }
            *reinterpret_cast<float4*>(&tile[i][4 * tx]) = v;
        }
    }
    __syncthreads();
    if (colsum && threadIdx.x < 64 && c0 + (int)threadIdx.x < C) {          // the bias gradient of the layer dY belongs to: column sums of this tile (n_taps == 1)
        float sum = 0.f;
#pragma unroll 8
        for (int i = 0; i < 64; ++i) sum += tile[i][threadIdx.x];
        atomicAdd(colsum + c0 + threadIdx.x, sum);
    }
    // write: thread = (channel c0 + cw, frames n0 + 8 g .. + 7) -> one 16-byte store per plane into the fragment-tiled layout
    const int cw = threadIdx.x & 63;
    if (c0 + cw < C) {
        for (int g8 = threadIdx.x >> 6; g8 < 8; g8 += 4)
            for (int j = 0; j < n_taps; ++j) {
                half8 hi, lo;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int tn = g8 * 8 + e, m = n0 + tn, t = m % ri.clip_len;
                    const int ts = t + (j - (n_taps >> 1)) * halo;           // the tap's source frame inside the clip
                    const bool ok = m < n_real && ts >= 0 && ts < ri.clip_len;
                    const float v = ok ? tile[tn + j * halo][cw] : 0.f;
                    hi[e] = (_Float16)v;
                    lo[e] = (_Float16)(v - (float)hi[e]);
                }
                _Float16* p = dst + (size_t)j * tap_halfs + pl_off(c0 + cw, n0 + g8 * 8, ldT);
                *reinterpret_cast<half8*>(p) = hi;
                *reinterpret_cast<half8*>(p + plane_halfs) = lo;
            }
    }
}

// ---- the contraction ------------------------------------------------------------------------------------------------------------
struct WgradNtArgs {
    const _Float16* at;         // dY^T planes [2][>= O_pad rows][ldT]
    const _Float16* bt;         // X^T planes  [2][>= K_pad rows][ldT]
    long long a_plane, b_plane; // halfs between the hi and the lo plane
    int ldT;                    // halfs per row (% 8 == 0)
    int n_total;                // frames to contract (% 32 == 0)
    int slice_len;              // frames per blockIdx.z slice (% 32 == 0)
    float* part;                // [slices][O_pad][K_pad] partial tiles
    int O_pad, K_pad;           // % 256 == 0, % 128 == 0
    int tiles;                  // output tiles per slice = (O_pad / 256) * (K_pad / 128); grid = tiles * slices workgroups (1-D)
    int xcd_map;                // slices % 8 == 0: workgroup i runs on XCD i % 8 (round-robin dispatch) -- give every XCD whole slices, so that
                                // the tiles that share operand rows share an L2 (otherwise each of the 8 L2s pulls every operand byte)
};

constexpr int WG_STAGE_BYTES = 48 * 1024;       // 32 frames: A 8 tiles x 2 planes x 2 k-steps + B 4 x 2 x 2 pieces of 1 KiB
constexpr int WG_STAGES = 3;

__global__ void __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2)))
wgrad_nt_kernel(const WgradNtArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef const half8 __attribute__((address_space(3))) * lds_frag_ptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo = wave >> 1, wk = wave & 1;                 // this wave's 64 x 64 corner of the 256 x 128 tile
    int slice, tile;
    if (a.xcd_map) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        slice = xcd + 8 * (j / a.tiles); tile = j % a.tiles;
    } else {
        slice = blockIdx.x / a.tiles; tile = blockIdx.x % a.tiles;
    }
    const int kt = a.K_pad >> 7;
    const int o0 = (tile / kt) * 256, k0 = (tile % kt) * 128;
    const int n_begin = slice * a.slice_len;
    int n_end = n_begin + a.slice_len;
    if (n_end > a.n_total) n_end = a.n_total;
    const int stages = n_end > n_begin ? (n_end - n_begin) >> 5 : 0;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // fragment-tiled planes: a 32-row x 16-frame piece is 1 KiB contiguous, lane l reads its 16 bytes at + 16 l
    const int kp = a.ldT >> 4;                                 // pieces per 32-row block
    auto dma = [&](int s) {
        char* dst = smem + (s % WG_STAGES) * WG_STAGE_BYTES;
        const int n = n_begin + s * 32;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int pc = wave + 8 * i;                       // piece 0..31: A (tile r, plane p, k-step j), 32..47: B
            const _Float16* src;
            if (i < 4) {
                const int r = pc >> 2, p = (pc >> 1) & 1, j = pc & 1;
                src = a.at + (long long)p * a.a_plane + ((long long)((o0 >> 5) + r) * kp + (n >> 4) + j) * 512 + lane * 8;
            } else {
                const int q = pc - 32, c = q >> 2, p = (q >> 1) & 1, j = q & 1;
                src = a.bt + (long long)p * a.b_plane + ((long long)((k0 >> 5) + c) * kp + (n >> 4) + j) * 512 + lane * 8;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (stages > 0) dma(0);
    if (stages > 1) dma(1);
    for (int s = 0; s < stages; ++s) {
        // vmcnt retires in order and this wave's only vector-memory traffic is its 6 DMA pieces per stage: allowing stage s+1's six to be
        // outstanding means stage s has landed; the barrier publishes it and retires buffer (s + 2) % 3 (read during stage s - 1)
        if (s + 1 < stages) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + 2 < stages) dma(s + 2);
        const unsigned buf = lds0 + (unsigned)(s % WG_STAGES) * WG_STAGE_BYTES + (unsigned)lane * 16u;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            half8 fa[2][2], fb[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    fa[i][p] = *(lds_frag_ptr)(size_t)(buf + (unsigned)((((2 * wo + i) * 2 + p) * 2 + j) * 1024));
                    fb[i][p] = *(lds_frag_ptr)(size_t)(buf + (unsigned)((32 + ((2 * wk + i) * 2 + p) * 2 + j) * 1024));
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fb[q][0], acc[i][q], 0, 0, 0);
                    acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][1], fb[q][0], acc[i][q], 0, 0, 0);
                    acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fb[q][1], acc[i][q], 0, 0, 0);
                }
        }
    }
    // partial tile: accumulator register r of lane l = row 8 (r >> 2) + 4 (l >> 5) + (r & 3), column l & 31
    float* out = a.part + ((size_t)slice * a.O_pad + o0 + wo * 64) * a.K_pad + k0 + wk * 64 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                out[(size_t)row * a.K_pad + q * 32] = acc[i][q][r];
            }
}

// ---- round 5: the same contraction straight from the FRAME-MAJOR planes the layer kernels already write ------------------------------------
// k_split_t is a pure layout pass: it re-reads the fp32 rows a producing epilogue wrote and writes them again, channel-major, once per operand and
// conv tap (87 launches, 3.7 GB of HBM traffic and 10.6 % of the training step at the 64 x 128-frame batch, profiles/r5F_kernel_stats_train.csv).
// Every operand of the residual layers' weight gradients ALSO exists as fp16 [hi | lo] row planes -- the tgemm layer kernels write them for their
// own contractions (train.hip: TEpiGateT -> g, TEpiResSkipT -> x + film, TEpiGateBwdT -> dy, TEpiDxT -> dO), frame-major, zero on every gap row.
// wgrad_fm_kernel contracts THOSE: a 32-frame stage of the A planes [32][256 channels] and of the B planes [32][128] is DMA'd into LDS as it
// lies in memory (512 B / 256 B contiguous per frame), and gfx950's transposing LDS read (ds_read_b64_tr_b16) hands each lane the four
// consecutive FRAMES of its channel that an MFMA fragment wants: within a 16-lane group, lane s supplies the address of 4 consecutive channels of
// frame (s >> 2) and lane i receives channel i of the four frames.  Two such reads (frames f .. f+3, f+4 .. f+7) make one half8 fragment, at the
// LDS cost of the one ds_read_b128 of the fragment-tiled planes.  A conv tap is a row offset of the B source (the planes carry guard rows), so the
// three taps of a dilated conv read ONE copy of x.  The contraction runs over all rows, gap rows included: their products are zero (both operands
// are), 6 % more MFMAs at 8 gap rows per 128-frame clip.
// LDS image of a stage (48 KB): A hi [32 frames][512 B] | A lo | B hi [32][256 B] | B lo.  A frame's 64-byte blocks (32 channels) are XOR-swizzled
// by (frame & 3) on the SOURCE side of the DMA, so the 4 frames x 64 B that 32 lanes of a transposing read touch cover all 64 banks once.
// BIAS: the column sums of A (a bias gradient) ride along as MFMAs against a fragment of ones; the stages are dealt round-robin to the k tiles of
// an output row block, each wave sums one 32-row tile: 2 more MFMAs per k16 step in one stage out of (k tiles).
struct WgradFmSeg {
    const _Float16* b;      // B planes, row 0 (guard rows precede where shift < 0): [n][ld] halfs, hi at column 0, lo `lo` halfs further
    int ld, lo;
    int shift;              // rows added to the frame index (a conv tap: (tap - 1) * dil)
    int k_tiles;            // 128-column tiles of this segment on the k axis
};
// one contraction of a launch
struct WgradFmProb {
    const _Float16* a;      // A planes, row 0: [n][a_ld] halfs, hi at column 0, lo a_lo halfs further
    int a_ld, a_lo;
    WgradFmSeg seg[4];
    int n_seg;
    int slices;             // frame slices (blockIdx -> slice, tile as wgrad_nt_kernel; xcd_map: a multiple of 8, every XCD gets whole slices)
    int slice_stages;       // 32-row stages per slice
    float* part;            // [slices][O_pad][K_pad] partial tiles
    int O_pad, K_pad, tiles;
    float* bias_part;       // BIAS: [slices][K_pad / 128][O_pad] partial column sums of A, or null
};
// A launch runs ONE or TWO contractions (`n_prob`): two that would each leave CUs idle or need many thin slices to fill the chip -- a layer's output
// projection (9 output tiles) and its conditioner projection (6) -- share a grid: workgroups [0, wgs0) belong to p[0], the rest to p[1] (with
// xcd_map per XCD: the first wgs0 / 8 workgroups of an XCD).
// Rows: stage t covers rows 32 t .. 32 t + 31 of the workspace -- or, with spc > 0 (every clip holds spc whole stages: T % 32 == 0), the rows
// clip_rows * (t / spc) + 32 * (t % spc) ..: the gap rows between clips (zeros in every plane: 6 % of the rows at 128 + 8) are not contracted.
struct WgradFmArgs {
    WgradFmProb p[2];
    int n_prob, wgs0;
    int n_stages;           // stages in all
    int clip_rows, spc;
    int xcd_map;
};

typedef short wg_short4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef short wg_short8 __attribute__((__vector_size__(8 * sizeof(short))));

// The transposing read as inline asm: through the builtin (__builtin_amdgcn_ds_read_tr16_b64_v4i16) hipcc puts an s_waitcnt vmcnt(0) in front of
// the first read of every stage -- it cannot tell the read from the LDS-DMA pieces in flight for the stages after it, and the three-stage
// pipeline collapses into load / wait / compute.  An asm read is invisible to that pass; its lgkmcnt wait is ours to place (wg_frags_wait ties the
// registers to the wait so that no MFMA moves above it).
template <int OFF>
__device__ __forceinline__ void wg_tr_read(wg_short4& d, unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
// the 8 fragments of one k16 step: A [i = 32-row tile][p = hi | lo], B [q][p]; each two 64-bit reads (frames f .. f+3 | f+4 .. f+7)
struct WgFrags { wg_short4 a[2][2][2], b[2][2][2]; };
template <int J>
__device__ __forceinline__ void wg_frags_read(WgFrags& f, const unsigned (&adA)[2], const unsigned (&adB)[2]) {
#define WG_RA(i, p, h) wg_tr_read<16384 * (p) + 8192 * J + 2048 * (h)>(f.a[i][p][h], adA[i])
#define WG_RB(i, p, h) wg_tr_read<8192 * (p) + 4096 * J + 1024 * (h)>(f.b[i][p][h], adB[i])
    WG_RA(0, 0, 0); WG_RA(0, 0, 1); WG_RB(0, 0, 0); WG_RB(0, 0, 1); WG_RB(1, 0, 0); WG_RB(1, 0, 1); WG_RA(1, 0, 0); WG_RA(1, 0, 1);
    WG_RA(0, 1, 0); WG_RA(0, 1, 1); WG_RA(1, 1, 0); WG_RA(1, 1, 1); WG_RB(0, 1, 0); WG_RB(0, 1, 1); WG_RB(1, 1, 0); WG_RB(1, 1, 1);
#undef WG_RA
#undef WG_RB
}
__device__ __forceinline__ void wg_frags_wait(WgFrags& f) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f.a[0][0][0]), "+v"(f.a[0][0][1]), "+v"(f.a[0][1][0]), "+v"(f.a[0][1][1]), "+v"(f.a[1][0][0]), "+v"(f.a[1][0][1]), "+v"(f.a[1][1][0]),
                   "+v"(f.a[1][1][1])
                 :: "memory");
    asm volatile("" : "+v"(f.b[0][0][0]), "+v"(f.b[0][0][1]), "+v"(f.b[0][1][0]), "+v"(f.b[0][1][1]), "+v"(f.b[1][0][0]), "+v"(f.b[1][0][1]), "+v"(f.b[1][1][0]),
                 "+v"(f.b[1][1][1]));
}
__device__ __forceinline__ half8 wg_frag(const wg_short4 (&v)[2]) {
    const wg_short8 w = __builtin_shufflevector(v[0], v[1], 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(half8, w);
}

template <bool BIAS>
__global__ void __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2)))
wgrad_fm_kernel(const WgradFmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo = wave >> 1, wk = wave & 1;                 // this wave's 64 x 64 corner of the 256 x 128 tile
    // which contraction, which slice, which output tile
    int prob = 0, slice, tile;
    if (a.xcd_map) {
        const int xcd = blockIdx.x & 7;
        int j = blockIdx.x >> 3;
        const int per0 = a.wgs0 >> 3;
        if (a.n_prob > 1 && j >= per0) { prob = 1; j -= per0; }
        const int tl = a.p[prob].tiles;
        slice = xcd + 8 * (j / tl); tile = j % tl;
    } else {
        int j = blockIdx.x;
        if (a.n_prob > 1 && j >= a.wgs0) { prob = 1; j -= a.wgs0; }
        const int tl = a.p[prob].tiles;
        slice = j / tl; tile = j % tl;
    }
    const WgradFmProb& P = a.p[prob];
    const int kt = P.K_pad >> 7;
    const int ot = tile / kt, kti = tile - ot * kt;
    const int o0 = ot * 256;
    // the B source of this k tile (wave-uniform)
    const _Float16* bsrc = P.seg[0].b;
    int b_ld = P.seg[0].ld, b_lo = P.seg[0].lo, shift = P.seg[0].shift, kc0 = 0;
    {
        int ks = kti;
        bool found = false;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (!found && s < P.n_seg) {
                if (ks < P.seg[s].k_tiles) { bsrc = P.seg[s].b; b_ld = P.seg[s].ld; b_lo = P.seg[s].lo; shift = P.seg[s].shift; kc0 = ks * 128; found = true; }
                else ks -= P.seg[s].k_tiles;
            }
        }
    }
    const _Float16* asrc = P.a;
    const int a_ld = P.a_ld, a_lo = P.a_lo;
    const int st_begin = slice * P.slice_stages;
    int st_end = st_begin + P.slice_stages;
    if (st_end > a.n_stages) st_end = a.n_stages;
    const int stages = st_end > st_begin ? st_end - st_begin : 0;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // row of the next stage to DMA (stages are DMA'd in order): one division here, then increments
    int dma_c = 0;
    long long dma_row = (long long)st_begin * 32;
    if (a.spc > 0) { const int cl = st_begin / a.spc; dma_c = st_begin - cl * a.spc; dma_row = (long long)cl * a.clip_rows + (long long)dma_c * 32; }

    auto dma = [&](int s) {
        char* dst = smem + (s % WG_STAGES) * WG_STAGE_BYTES;
        const long long n = dma_row;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int pc = wave + 8 * i;                       // KiB pieces 0..15: A hi (2 frames each), 16..31: A lo, 32..39: B hi (4 frames each), 40..47: B lo
            const _Float16* src;
            if (i < 4) {
                const int p = i >> 1, fr = 2 * (pc & 15) + (lane >> 5);
                const int c = (lane & 31) ^ ((fr & 3) << 2);   // 16-byte chunk of the source row that lands in LDS chunk (lane & 31)
                src = asrc + (n + fr) * (long long)a_ld + (long long)p * a_lo + o0 + c * 8;
            } else {
                const int p = i - 4, fr = 4 * (pc & 7) + (lane >> 4);
                const int c = (lane & 15) ^ ((fr & 3) << 2);
                src = bsrc + (n + fr + shift) * (long long)b_ld + (long long)p * b_lo + kc0 + c * 8;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, 0, 0);
        }
        dma_row += 32;
        if (a.spc > 0 && ++dma_c == a.spc) { dma_c = 0; dma_row += a.clip_rows - 32 * a.spc; }
    };

    // fragment addresses: lane (i = lane & 15, g = lane >> 4) supplies frame 8 (g >> 1) + (i >> 2) [+ 4 for the second read], channels
    // 16 (g & 1) + 4 (i & 3) .. + 3 of a 32-channel block; block b of frame f sits at b ^ (f & 3)
    const int li = lane & 15, lg = lane >> 4;
    const unsigned fr_l = 8u * (unsigned)(lg >> 1) + (unsigned)(li >> 2);
    const unsigned inblk = 32u * (unsigned)(lg & 1) + 8u * (unsigned)(li & 3);
    unsigned baseA[2], baseB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        baseA[i] = lds0 + fr_l * 512u + ((((unsigned)(2 * wo + i)) ^ (unsigned)(li >> 2)) << 6) + inblk;
        baseB[i] = lds0 + 32768u + fr_l * 256u + ((((unsigned)(2 * wk + i)) ^ (unsigned)(li >> 2)) << 6) + inblk;
    }

    f32x16 acc[2][2], accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    half8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (_Float16)1.0f;

    auto products = [&](const WgFrags& f, bool bias_stage) {
        half8 fa[2][2], fb[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p) { fa[i][p] = wg_frag(f.a[i][p]); fb[i][p] = wg_frag(f.b[i][p]); }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fb[q][0], acc[i][q], 0, 0, 0);
                acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][1], fb[q][0], acc[i][q], 0, 0, 0);
                acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fb[q][1], acc[i][q], 0, 0, 0);
            }
        if constexpr (BIAS) {
            if (bias_stage) {
                const half8 ah = wk ? fa[1][0] : fa[0][0], al = wk ? fa[1][1] : fa[0][1];
                accb = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ones, accb, 0, 0, 0);
                accb = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, ones, accb, 0, 0, 0);
            }
        }
    };

    if (stages > 0) dma(0);
    if (stages > 1) dma(1);
    for (int s = 0; s < stages; ++s) {
        // (as wgrad_nt_kernel: this wave's only vector-memory traffic is its 6 DMA pieces per stage)
        if (s + 1 < stages) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + 2 < stages) dma(s + 2);
        const unsigned sb = (unsigned)(s % WG_STAGES) * WG_STAGE_BYTES;
        const bool bias_stage = BIAS && P.bias_part && ((st_begin + s) % kt) == kti;
        const unsigned adA[2] = {baseA[0] + sb, baseA[1] + sb}, adB[2] = {baseB[0] + sb, baseB[1] + sb};
        WgFrags f0, f1;
        wg_frags_read<0>(f0, adA, adB);
        wg_frags_wait(f0);
        wg_frags_read<1>(f1, adA, adB);                        // the second k16 step's reads fly under the first one's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        products(f0, bias_stage);
        __builtin_amdgcn_sched_barrier(0);
        wg_frags_wait(f1);
        products(f1, bias_stage);
    }
    // partial tile: accumulator register r of lane l = row 8 (r >> 2) + 4 (l >> 5) + (r & 3), column l & 31
    float* out = P.part + ((size_t)slice * P.O_pad + o0 + wo * 64) * P.K_pad + kti * 128 + wk * 64 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                out[(size_t)row * P.K_pad + q * 32] = acc[i][q][r];
            }
    if constexpr (BIAS) {
        if (P.bias_part && (lane & 31) == 0) {                                // every column of accb holds the row sums: take column 0
            float* bp = P.bias_part + ((size_t)slice * kt + kti) * P.O_pad + o0 + wo * 64 + wk * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) bp[8 * (r >> 2) + (r & 3)] = accb[r];
        }
    }
}

// ---- slice reduction + scatter into the parameter gradients -----------------------------------------------------------------------
// The k axis of a launch may concatenate several operands (the three taps of a dilated conv and its conditioner projection share dY):
// segment s covers columns [k_begin, k_begin + k_len) and lands at dst[o * stride_o + (k - k_begin) * stride_k + off].
struct WgradSeg { float* dst; int k_begin, k_len; long long stride_o, stride_k, off; };
struct WgradSegs { WgradSeg s[4]; int n; };

__global__ __launch_bounds__(256) void k_wgrad_nt_reduce(const float* __restrict__ part, int n_slices, int O_pad, int K_pad, int n_o, WgradSegs segs,
                                                         float scale, const float* __restrict__ bias_part = nullptr, int n_bias = 0,
                                                         float* __restrict__ bias_dst = nullptr, float* __restrict__ bias_dst2 = nullptr) {
    if (bias_part && blockIdx.x == gridDim.x - 1) {
        // wgrad_fm_kernel<BIAS>'s partial column sums [n_bias][O_pad] ride on this launch (grid.x has one more column): block y sums 64 outputs,
        // four threads per output walk the partials, thread group 0 adds the four in a fixed order
        __shared__ float red[4][64];
        const int o = blockIdx.y * 64 + (threadIdx.x & 63), zl = threadIdx.x >> 6;
        if (blockIdx.y * 64 >= n_o) return;
        float v = 0.f;
        if (o < n_o)
            for (int z = zl; z < n_bias; z += 4) v += bias_part[(size_t)z * O_pad + o];
        red[zl][threadIdx.x & 63] = v;
        __syncthreads();
        if (zl == 0 && o < n_o) {
            const float r = (((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x]) * scale;
            bias_dst[o] = r;
            if (bias_dst2) bias_dst2[o] = r;                   // (a second tensor with the same gradient: the conditioner projection's bias)
        }
        return;
    }
    const int k = (blockIdx.x * 256 + threadIdx.x) * 4;           // four adjacent columns per thread (segments begin at multiples of 128, lengths % 4 == 0)
    const int o = blockIdx.y;
    if (k >= K_pad || o >= n_o) return;
    int si = -1;
#pragma unroll
    for (int s = 0; s < 4; ++s)
        if (s < segs.n && k >= segs.s[s].k_begin && k < segs.s[s].k_begin + segs.s[s].k_len) si = s;
    if (si < 0) return;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t slice = (size_t)O_pad * K_pad;
    const float* p = part + (size_t)o * K_pad + k;
#pragma unroll 8
    for (int z = 0; z < n_slices; ++z) {
        const float4 x = *reinterpret_cast<const float4*>(p + z * slice);
        v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
    }
    float* dptr = nullptr; int kb = 0; long long so = 0, sk = 0, off = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s)
        if (s == si) { dptr = segs.s[s].dst; kb = segs.s[s].k_begin; so = segs.s[s].stride_o; sk = segs.s[s].stride_k; off = segs.s[s].off; }
    float* d = dptr + (long long)o * so + (long long)(k - kb) * sk + off;
    d[0] = v.x * scale; d[sk] = v.y * scale; d[2 * sk] = v.z * scale; d[3 * sk] = v.w * scale;
}

}  // namespace dsvc
