// tskip.h -- the deferred skip path of DiffNet (net.py:80-84,131-133) as ONE contraction per evaluation (gfx950 / CDNA4, wave64).
//
// Reference:  skip_l = W_out,l[C:2C] g_l + b_out,l[C:2C]   for every residual layer l (second half of the layer's 1x1, net.py:80-84)
//             s      = relu(W_sp (sum_l skip_l / sqrt(L)) + b_sp)                                   (net.py:131-133)
// Both maps are linear up to the ReLU, so  s = relu( sum_l W'_l g_l + b' )  with  W'_l = W_sp W_out,l[C:2C] / sqrt(L)  (composed once, in
// fp64, when the weights are loaded) and  b' = b_sp + W_sp sum_l b_out,l[C:2C] / sqrt(L):  a [C x L*C] contraction over the gate outputs
// g_l that the layer kernels leave in HBM as fp16 -- the very operand values the per-layer skip halves would have consumed.
//
// Why: the per-layer form keeps a running fp32 skip sum in HBM and read-modify-writes it in every layer (3 KB per frame per layer of the
// layer kernel's 10.8); this form writes g once (768 B) and reads it once here.  The matrix work is the same (it moves from the layers
// to this kernel), and the separate skip projection disappears.  On a part whose matrix and HBM phases do not overlap
// (tools/micro/overlap.hip) the saved bytes are saved time.
//
// Structure (throughput tiling only: 128 frames per workgroup, 8 waves, two per SIMD):
//   * K is walked in HALF-slabs: [128 frames x C/2 channels] of one layer's g = 48 KB (C = 384), double-buffered in LDS (96 KB) by LDS-DMA
//     with the source-side XOR swizzle of tgemm.h; one bare s_barrier per half-slab (the DMA of half-slab h+1 and every wave's weight
//     prefetch stay in flight across it: counted vmcnt).
//   * output tiles: C/32 = 12 (C = 384).  Wave w owns tile w for all 128 frames (4 accumulator N-tiles) AND tile 8 + (w & 3) for the 64
//     frames of half (w >> 2) (2 N-tiles): every wave issues 12 MFMAs per k-step and plane, the activations are staged once.
//     (C = 256: 8 tiles, one per wave.)
//   * weights: A fragments [tile][layer][k16][plane hi|lo][lane][8] streamed through two register rings of KG = 2 k-steps (both tiles).
//   * epilogue: relu(acc + b') -> fp16 hi|lo planes of the final projection's operand (TEpiReluHalf, diffnet_t.h).
#pragma once
#include "diffnet_t.h"

namespace dsvc {

struct TSkipArgs {
    const _Float16* g;          // gate outputs of all layers: [L][rows_alloc][cin] fp16
    long long slab_halfs;       // rows_alloc * cin
    int cin, L;                 // channels per row (= C), residual layers
    const _Float16* w;          // composed weights, packed [C/32 tiles][L][C/16][2 planes][lane][8]
};

// NB = C / 128 (3 or 2)
template <int NB>
__global__ void __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2)))
tskip_kernel(const TSkipArgs a, const TEpiReluHalf::Args e) {
    constexpr int C = 128 * NB, HS = C / 2;               // half-slab width in channels
    constexpr int KSTEPS = HS / 16;                       // k16 steps per half-slab (12 / 8)
    constexpr int KG = 2, NW = 2, FR = KG * NW;           // ring group: 2 k-steps x (hi, lo) per tile
    constexpr int GROUPS = KSTEPS / KG;                   // groups per half-slab (6 / 4)
    constexpr bool HALF = NB == 3;                        // a second (half-width) tile per wave
    constexpr int CHUNKS = HS / 8, ROW_BYTES = HS * 2, BUF_BYTES = 128 * ROW_BYTES;
    constexpr unsigned SWZ = (CHUNKS % 16 == 0) ? 15u : 7u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef const half8 __attribute__((address_space(3))) * lds_frag_ptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * 128;
    const int mt1 = 8 + (wave & 3), nh = wave >> 2;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int n_hs = 2 * a.L;

    auto dma = [&](int hs) {                              // half-slab hs -> buffer hs & 1 (48 x 1 KiB pieces, 6 per wave)
        const _Float16* src0 = a.g + (long long)(hs >> 1) * a.slab_halfs + (long long)row0 * a.cin + (hs & 1) * HS;
        char* dst = smem + (hs & 1) * BUF_BYTES;
#pragma unroll
        for (int i = 0; i < (128 * CHUNKS) / 512; ++i) {
            const int it = wave + 8 * i;                  // 1 KiB piece
            const int slot = it * 64 + lane;
            const int r = slot / CHUNKS, c = slot - r * CHUNKS;
            const _Float16* src = src0 + (long long)r * a.cin + ((c ^ (r & (int)SWZ)) << 3);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + it * 1024), 16, 0, 0);
        }
    };
    const long long tile_halfs = (long long)a.L * (C / 16) * NW * TFRAG_HALFS;
    const _Float16* wa = a.w + (long long)wave * tile_halfs + lane * 8;
    const _Float16* wb = a.w + (long long)mt1 * tile_halfs + lane * 8;
    constexpr int GROUP_HALFS = FR * TFRAG_HALFS;         // consecutive groups of one tile are contiguous: [layer][k16][plane]
    auto load_ring = [&](half8 (&ra)[FR], half8 (&rb)[FR], int gi) {
#pragma unroll
        for (int u = 0; u < FR; ++u) ra[u] = *reinterpret_cast<const half8*>(wa + (long long)gi * GROUP_HALFS + u * TFRAG_HALFS);
        if constexpr (HALF) {
#pragma unroll
            for (int u = 0; u < FR; ++u) rb[u] = *reinterpret_cast<const half8*>(wb + (long long)gi * GROUP_HALFS + u * TFRAG_HALFS);
        }
    };

    f32x16 acc[4], acc2[2];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc2[nt][i] = 0.f;
    half8 rA0[FR], rB0[FR], rA1[FR], rB1[FR];

    // lane's LDS row inside a buffer for N-tile 0 of tile A / of tile B, and its swizzle term
    const unsigned rrow = (unsigned)(lane & 31);
    const unsigned xs0 = ((rrow & SWZ) ^ (unsigned)(lane >> 5)) << 4;          // rows 32*nt + rrow share (row & SWZ): 32 is a multiple of 16
    const unsigned rowA = rrow * ROW_BYTES, rowB = (64u * nh + rrow) * ROW_BYTES;

    // B fragments of k-step k of the current buffer: 4 N-tiles of tile A, the 2 N-tiles of this wave's half of tile B
    auto lds_b = [&](unsigned buf, int k, half8 (&ba)[4], half8 (&bb)[2]) {
        const unsigned off = ((unsigned)k << 5) ^ xs0;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) ba[nt] = *(lds_frag_ptr)(size_t)(buf + rowA + (unsigned)nt * 32u * ROW_BYTES + off);
        if constexpr (HALF) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bb[nt] = *(lds_frag_ptr)(size_t)(buf + rowB + (unsigned)nt * 32u * ROW_BYTES + off);
        }
    };
    auto mfma_step = [&](const half8 (&ra)[FR], const half8 (&rb)[FR], int kk, const half8 (&ba)[4], const half8 (&bb)[2]) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ra[kk * NW], ba[nt], acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ra[kk * NW + 1], ba[nt], acc[nt], 0, 0, 0);
        }
        if constexpr (HALF) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rb[kk * NW], bb[nt], acc2[nt], 0, 0, 0);
                acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rb[kk * NW + 1], bb[nt], acc2[nt], 0, 0, 0);
            }
        }
    };
    constexpr int MFMA_STEP = HALF ? 12 : 8, READS_STEP = HALF ? 6 : 4;

    dma(0);
    load_ring(rA0, rB0, 0);
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);         // the two waves of a SIMD drift apart (tgemm.h)
    constexpr int RING_LOADS = HALF ? 2 * FR : FR;        // loads of one ring refill: may stay in flight across the barrier
    const int n_groups = n_hs * GROUPS;
    for (int hs = 0; hs < n_hs; ++hs) {
        // every DMA piece of half-slab hs was issued BEFORE the ring refill that is in flight now: vmcnt retires in order, so allowing
        // that refill's loads to be outstanding still covers them; the barrier then publishes the buffer and retires buffer (hs + 1) & 1
        if constexpr (RING_LOADS == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (hs + 1 < n_hs) dma(hs + 1);
        const unsigned buf = lds0 + (unsigned)(hs & 1) * BUF_BYTES;
        const int g0 = hs * GROUPS;
        half8 bA[2][4], bB[2][2];
        lds_b(buf, 0, bA[0], bB[0]);
        // the half-slab's k-steps as ONE software pipeline: the B fragments of step k+1 are read under the MFMAs of step k, a ring is
        // refilled (for the group after next) at every group boundary; only the first read of a half-slab is exposed
#pragma unroll
        for (int k = 0; k < KSTEPS; ++k) {
            const int g = k / KG, kk = k % KG;
            if (kk == 0) {                                 // ring (g & 1) is being consumed: refill the other one with group g + 1
                const int gn = g0 + g + 1 < n_groups ? g0 + g + 1 : n_groups - 1;
                if ((g & 1) == 0) load_ring(rA1, rB1, gn); else load_ring(rA0, rB0, gn);
            }
            if (k + 1 < KSTEPS) lds_b(buf, k + 1, bA[(k + 1) & 1], bB[(k + 1) & 1]);
            if ((g & 1) == 0) mfma_step(rA0, rB0, kk, bA[k & 1], bB[k & 1]); else mfma_step(rA1, rB1, kk, bA[k & 1], bB[k & 1]);
            // pin: the refill first, then one B-fragment read of the next step behind every second MFMA
            if (kk == 0) __builtin_amdgcn_sched_group_barrier(0x020, RING_LOADS, 0);
#pragma unroll
            for (int i = 0; i < READS_STEP; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, MFMA_STEP / READS_STEP, 0);
                if (k + 1 < KSTEPS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
    }
    TEpiReluHalf epi;
    epi.finish(e, wave, row0, lane, acc);
    if constexpr (HALF) epi.template finish<2>(e, mt1, row0 + 64 * nh, lane, acc2);
}

inline bool tskip_supported(int C, int n_rows) { return (C == 256 || C == 384) && n_rows % 128 == 0; }

inline int tskip_launch(const TSkipArgs& a, const TEpiReluHalf::Args& e, int n_rows, hipStream_t stream) {
    if (!tskip_supported(a.cin, n_rows)) return fail(DSVC_EINVAL, "tskip: shape not supported (C %d, %d rows)", a.cin, n_rows);
    const size_t smem = (size_t)2 * 128 * a.cin;          // two half-slab buffers of 128 x C/2 fp16
    static thread_local size_t smem_set[2] = {0, 0};
    if (a.cin == 384) {
        if (smem > smem_set[0]) {
            DSVC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(tskip_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            smem_set[0] = smem;
        }
        hipLaunchKernelGGL(tskip_kernel<3>, dim3(n_rows / 128), dim3(512), smem, stream, a, e);
    } else {
        if (smem > 64 * 1024 && smem > smem_set[1]) {
            DSVC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(tskip_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            smem_set[1] = smem;
        }
        hipLaunchKernelGGL(tskip_kernel<2>, dim3(n_rows / 128), dim3(512), smem, stream, a, e);
    }
    DSVC_HIP(hipGetLastError());
    return DSVC_OK;
}

}  // namespace dsvc
