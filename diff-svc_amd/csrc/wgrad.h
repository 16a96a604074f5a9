// wgrad.h -- weight gradients of the DiffNet training step (gfx950 / CDNA4, wave64):  dW[o][k] = sum_n dY[n][o] * X[n][k]
// Reference: the autograd of network/diff/net.py:58-135 under network/diff/diffusion.py:207-225 (every Conv1d / Linear weight of
// the denoiser); the contraction index n is the frame index (~8 000 per step at the 64 x 128-frame batch).
//
// Both operands are produced frame-major ([n][channel] fp32) by the forward / data-gradient kernels, but an MFMA fragment wants the
// contraction index contiguous per lane.  Round 2 ran this contraction on the conv engine: dY^T repacked into weight fragments
// (k_pack_w), X^T as an fp32 transposed copy, fp32 -> hi|lo conversion at staging, 32 x 64 output tiles: 41 % of the training step.
// Here:
//   * k_split_t  writes an operand ONCE as channel-major fp16 planes  T[plane hi|lo][channel][n]  (v = hi + lo, fp32-class), with the
//                validity mask, the FiLM shift of the dilated conv's input and the conv tap's frame shift folded in.
//   * wgrad_nt_kernel  is then a plain "NT" GEMM on those planes: a 256 (o) x 128 (k) output tile per workgroup, 8 waves of 64 x 64,
//                three MFMAs per product (hi*hi + lo*hi + hi*lo).  The frame range is cut into slices (a handful of output tiles would
//                not fill 256 CUs) whose partial tiles a second kernel adds in a fixed order (deterministic); every XCD gets whole slices.
//                Operands are staged 32 frames at a time by LDS-DMA (global_load_lds_dwordx4) in FRAGMENT order: lane l of a 1 KiB
//                piece loads row (l & 31), frames 8 (l >> 5) .. +7 and the DMA drops it at piece + 16 l -- exactly what ds_read_b128 at
//                lane*16 hands to the MFMA, so there is no swizzle and no bank conflict.  Three stages (144 KB) in flight, one bare
//                s_barrier per stage, counted vmcnt.
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

// ---- slice reduction + scatter into the parameter gradients -----------------------------------------------------------------------
// The k axis of a launch may concatenate several operands (the three taps of a dilated conv and its conditioner projection share dY):
// segment s covers columns [k_begin, k_begin + k_len) and lands at dst[o * stride_o + (k - k_begin) * stride_k + off].
struct WgradSeg { float* dst; int k_begin, k_len; long long stride_o, stride_k, off; };
struct WgradSegs { WgradSeg s[4]; int n; };

__global__ __launch_bounds__(256) void k_wgrad_nt_reduce(const float* __restrict__ part, int n_slices, int O_pad, int K_pad, int n_o, WgradSegs segs,
                                                         float scale) {
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
