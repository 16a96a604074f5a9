// tlayer.h -- one residual layer of DiffNet (net.py:66-84) as ONE kernel in the throughput tiling (gfx950 / CDNA4, wave64).
//
//   phase 1  gate GEMM        y[2C][128] = cproj + W_dil[2C][3*C] * xh[frame + (tap-1)*d][C]      (dilated k=3 conv, K4+K5)
//            epilogue         g = sigmoid(y[:C]) * tanh(y[C:])  -> fp16, kept ON CHIP                 (K6)
//   phase 2  output 1x1       [r; s][2C][128] = W_out[2C][C] * g,  x <- (x + r)/sqrt(2),  skip += s   (K7+K8)
//            epilogue         + the NEXT layer's operand xh = fp16(x + film_next)                      (K3 of layer l+1)
//
// versus the two tgemm launches it replaces (tgemm.h: TEpiGate, TEpiResSkip) the gate output never travels: 768 B per frame per
// layer less written and 768 B less read (12 % of the layer's HBM bytes), one kernel boundary and one prologue (kernarg, tile DMA of
// g, first weight ring) less per layer.  Everything else is tgemm's machinery: the (128 + 2d) x C fp16 time tile DMA'd once into
// LDS with the source-side XOR swizzle, weights streamed as A fragments through two register rings, accumulator-init loads instead
// of epilogue reads, 8 waves of [32 output rows x 128 frames].
//
// LDS plan (C = 384, d = 8: 143 360 B of the CU's 160 KB):
//   X  [0, (128+2d)*2C)         the time tile, live through all gate passes
//   S  [round_up(X), +32 KB)    the g block of the FIRST gate pass (a pass = 8 waves x 16 g-channels = one 128-channel block of
//                               128 frames x 256 B); written as soon as that pass's epilogue has it
//   the middle pass's g block waits in 16 VGPRs per lane (there is no room for a second block beside X: 52 KB free, 64 needed)
//   after the last gate pass: barrier (X is dead) -> the remaining blocks are written over X -> barrier -> phase 2 reads g.
// A g block is [128 frames][128 channels] fp16, 16-B chunk c of row f at slot c ^ (f & 15): the same conflict-free ds_read_b128
// pattern as the time tile, and ds_write_b128 of 8 consecutive frames x one chunk is conflict-free too.
#pragma once
#include "diffnet_t.h"

namespace dsvc {

constexpr int TL_TN = 128;                 // frames per workgroup in the throughput tiling (NT = 4 N-tiles of 32 frames; round 5: the kernel is
                                           // templated on NT -- 64- and 32-frame tiles fill the chip for mid-size batches, where 128-frame tiles
                                           // leave most CUs without a workgroup)
constexpr int TL_BLOCK_BYTES = TL_TN * 256;   // one g block: 128 frames x 128 channels fp16 (NT = 4)

// what the fused kernel needs of the two contractions (a trimmed TGemmArgs pair: kernel arguments live in SGPRs, and this kernel
// sits at the 256-VGPR limit where spilled SGPRs cost vector registers)
struct TLayerArgs {
    const _Float16* x;          // fp16 layer operand xh, row 0 (guard rows precede)
    int cin, swz, dil;          // channels per row (= C), chunk swizzle mask, dilation of the k=3 conv
    const _Float16* gw;         // gate weights  [variant][C/16 m_tiles][3 taps][C/16][planes][lane][8]
    const _Float16* ow;         // output 1x1    [variant][2C/32 m_tiles][C/16][planes][lane][8]
    long long gvar, ovar;       // halfs between dither variants
    _Float16* gall;             // DEFER: this layer's slab of the gate-output buffer [rows][cin] fp16 (read by tskip.h); else null
    int n_variants;             // > 1: variant = (*step_ptr - step_off) mod n_variants is resolved in the kernel; the sampler passes
    const int* step_ptr;        //      the variant by value (gw / ow already offset, n_variants = 1)
    int step_off;
    // W6 (DSVC_PREC_F16_W6): the gate's w_lo plane as fp6 codes (tgemm.h: k_tpack6), one 1536-B fragment per (m_tile, tap, 64 input channels);
    // gw then holds hi | lo fp16 planes of which only the hi fragments are streamed
    const unsigned* gw6;
    long long g6var;            // dwords between dither variants of gw6
    int sc6;                    // E8M0 scale bytes of the 6-bit products: gate weights (2^e6) | x (TL_X6_E8M0) << 8 | output-1x1 weights << 16 | g_lo << 24
    // G6 (W6 == 2): the output projection gets a 6-bit g_lo correction too -- the gate epilogue keeps bf6((g - fp16(g)) * 2^16) beside
    // fp16(g) in LDS, and ow6 holds fp6 codes of the output-projection weights THEMSELVES (k_tpack6 with `whole`)
    const unsigned* ow6;
    const unsigned* ow6lo;      // W6: fp6 codes of the output projection's w_lo plane (dither variants o6var dwords apart), scale byte sc6b & 255
    long long o6var;
    int sc6b;
#ifdef DSVC_PROFILING
    unsigned long long* stamps; // profiling build: 16 s_memrealtime stamps per wave of the LAST launch (tools/gpu_layer_stamps.py)
    int abl;                    // round-6 ablations (env DSVC_TL_ABL, tools/gpu_r6_ablate.py): 1 = half of cproj's bytes (WRONG results; TEpiGate::Args::abl);
                                // 2 = the neighbour hand-off protocol a persistent per-evaluation launch would need, as PURE OVERHEAD inside the real
                                // kernel (results unchanged): before its tile DMA a workgroup polls the flags of tiles i-1 / i+1 (set by the previous
                                // launch) and takes one agent-scope acquire; after its last store it drains, releases at agent scope and sets its flag
    unsigned* flags;            // abl & 2: one word per frame tile
    int dephase;                // env DSVC_TL_DEPHASE (second session of round 6): after "g complete" waves 4-7 (> 0) or 0-3 (< 0) sleep |dephase| x 1024 clocks,
                                // so that the two waves of a SIMD enter the output phase out of step (one in its MFMA loop while the other stores / waits for its init loads)
#endif
};

#ifdef DSVC_PROFILING
#define TL_STAMP(i) do { if (ga.stamps && lane == 0) ga.stamps[((size_t)blockIdx.x * 8 + wave) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
inline unsigned long long*& tl_stamp_buffer() { static unsigned long long* p = nullptr; return p; }
inline int& tl_stamp_groups() { static int n = 0; return n; }
#else
#define TL_STAMP(i) do { } while (0)
#endif

// KG k16-steps of one output tile against 4 N-tiles of 32 frames: tgemm's compute_group with the LDS geometry passed in.
//   base0     LDS byte address of this lane's row of N-tile 0
//   nt_stride bytes between N-tiles (32 rows)
//   xs        (((row & swz) ^ (lane >> 5)) << 4) ^ (first k16 step of the group << 5)
// (rings are flat arrays of KG * NW fragments -- [k-step][plane] -- so that the two phases of a layer can use different (KG, NW) splits of
//  the same eight registers: F16_MIX streams one dithered plane for the gate and hi + lo planes for the output projection)
template <int KG, int NW, int NT = 4>
__device__ __forceinline__ void tl_compute_group(const half8 (&ring)[KG * NW], f32x16 (&acc)[NT], unsigned base0, unsigned nt_stride, unsigned xs) {
    typedef const half8 __attribute__((address_space(3))) * lds_frag_ptr;
    unsigned base[NT];
    base[0] = base0;
#pragma unroll
    for (int nt = 1; nt < NT; ++nt) base[nt] = base[nt - 1] + nt_stride;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+v"(base[nt]));      // keep the bases materialised (see tgemm.h)
    half8 bq[2][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bq[0][nt] = *(lds_frag_ptr)(size_t)(base[nt] + xs);
#pragma unroll
    for (int kk = 0; kk < KG; ++kk) {
        if (kk + 1 < KG) {
            const unsigned off = ((unsigned)(kk + 1) << 5) ^ xs;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bq[(kk + 1) & 1][nt] = *(lds_frag_ptr)(size_t)(base[nt] + off);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[kk * NW], bq[kk & 1][nt], acc[nt], 0, 0, 0);
            if constexpr (NW == 2) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[kk * NW + 1], bq[kk & 1][nt], acc[nt], 0, 0, 0);
        }
    }
    // pin the software pipeline: one B-fragment read of step kk+1 behind each MFMA (pair) of step kk
    __builtin_amdgcn_sched_group_barrier(0x100, NT, 0);
#pragma unroll
    for (int kk = 0; kk < KG; ++kk) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            __builtin_amdgcn_sched_group_barrier(0x008, NW, 0);
            if (kk + 1 < KG) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    }
}

template <int KG, int NW>
__device__ __forceinline__ void tl_load_group(half8 (&ring)[KG * NW], const _Float16* p) {
#pragma unroll
    for (int u = 0; u < KG * NW; ++u) ring[u] = *reinterpret_cast<const half8*>(p + u * TFRAG_HALFS);
}

typedef int tl_v3i __attribute__((ext_vector_type(3)));
typedef int v6i_t __attribute__((ext_vector_type(6)));
typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef _Float16 half16_t __attribute__((ext_vector_type(16)));
typedef _Float16 half32_t __attribute__((ext_vector_type(32)));
constexpr float TL_X6_SCALE = 4.0f;        // activations of the 6-bit product are converted as bf6(x / 4): saturation at |x| = 112, rel. rms error 5.7 %
constexpr int TL_X6_E8M0 = 129;            // 2^2
constexpr float TL_G6_UP = 65536.0f;       // g_lo = g - fp16(g), |g_lo| <= 2^-12 (|g| < 1), is converted as bf6(g_lo * 2^16): |.| <= 16 of the format's 28
constexpr int TL_G6_E8M0 = 127 - 16;
constexpr float TL_G_X6_SCALE = 0.0625f;   // the gate output g (|g| < 1) enters the output projection's 6-bit w_lo product as bf6(16 g)
constexpr int TL_G_X6_E8M0 = 127 - 4;
constexpr int TL_S6_BYTES = TL_TN * 128;   // the 6-bit g_lo codes of one g block: 128 frames x [4 slots (64-channel group, lane half) x 32 B]

// W6: one group = 64 input channels (of one tap) against 4 N-tiles, N-tile by N-tile: four fp16 MFMAs on the hi fragments, then the lane's four
// B fragments (k = 16 kk + 8 h + e: exactly the order k_tpack6 packs the weight codes in) are converted to bf6 by ONE instruction and one
// K = 64 block-scaled MFMA adds w_lo * x.  The next N-tile's four fragments are read while the current one computes.
//   xscale / sc_x   the activations are converted as bf6(x / xscale); sc_x = 127 + log2(xscale)
// G6 (the output projection of DSVC_PREC_F16_W6): a second 6-bit MFMA adds W6 * g_lo6 -- the fp6 codes `wg` of the weights THEMSELVES against the
// bf6 codes of g_lo = g - fp16(g) the gate epilogue left in LDS (c0: byte address of the lane's code row of N-tile 0, rows of N-tiles 4096 B
// apart; o0 / o1: offsets of its two 16-byte chunks, 12 bytes used of each), read one N-tile ahead like the fragments.
template <bool G6, int NT = 4>
__device__ __forceinline__ void tl_compute_group_w6(const half8 (&hi)[4], const v6i_t& lo, f32x16 (&acc)[NT], unsigned base0, unsigned nt_stride,
                                                    unsigned xs, int sc_w, float xscale, int sc_x, const v6i_t& wg, unsigned c0, unsigned o0,
                                                    unsigned o1, int sc_wg, int sc_g) {
    typedef const half8 __attribute__((address_space(3))) * lds_frag_ptr;
    typedef const tl_v3i __attribute__((address_space(3))) * lds_code_ptr;     // 12 of a chunk's 16 bytes: ds_read_b96.  (Integers: __builtin_bit_cast of a
                                                                                 //  vector ELEMENT expression silently reads element 0 with this clang.)
    // (NBUF = 1: single-buffered fragments -- tried for G6 beside a prefetched second accumulator set; spilled all the same)
    constexpr int NBUF = NT > 1 ? 2 : 1;
    unsigned base[NT];
    base[0] = base0;
#pragma unroll
    for (int nt = 1; nt < NT; ++nt) base[nt] = base[nt - 1] + nt_stride;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+v"(base[nt]));
    const v8i_t a6 = __builtin_shufflevector(lo, lo, 0, 1, 2, 3, 4, 5, -1, -1);
    const v8i_t ag = __builtin_shufflevector(wg, wg, 0, 1, 2, 3, 4, 5, -1, -1);
    half8 b[NBUF][4];
    tl_v3i cu[2];
    if constexpr (NBUF == 2) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) b[0][kk] = *(lds_frag_ptr)(size_t)(base[0] + (((unsigned)kk << 5) ^ xs));
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        if constexpr (NBUF == 1) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) b[0][kk] = *(lds_frag_ptr)(size_t)(base[nt] + (((unsigned)kk << 5) ^ xs));
        }
        if constexpr (G6) {
            cu[0] = *(lds_code_ptr)(size_t)(c0 + (unsigned)nt * 4096u + o0);
            cu[1] = *(lds_code_ptr)(size_t)(c0 + (unsigned)nt * 4096u + o1);
        }
        if (NBUF == 2 && nt + 1 < NT) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) b[(nt + 1) % NBUF][kk] = *(lds_frag_ptr)(size_t)(base[nt + 1] + (((unsigned)kk << 5) ^ xs));
        }
        const half8 (&bn)[4] = b[nt % NBUF];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hi[kk], bn[kk], acc[nt], 0, 0, 0);
        const half16_t v01 = __builtin_shufflevector(bn[0], bn[1], 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
        const half16_t v23 = __builtin_shufflevector(bn[2], bn[3], 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
        const half32_t v = __builtin_shufflevector(v01, v23, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25,
                                                   26, 27, 28, 29, 30, 31);
        const v6i_t q = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(v, xscale);
        const v8i_t b6 = __builtin_shufflevector(q, q, 0, 1, 2, 3, 4, 5, -1, -1);
        acc[nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a6, b6, acc[nt], 2 /* A: fp6 E2M3 */, 3 /* B: bf6 E3M2 */, 0, sc_w, 0, sc_x);
        if constexpr (G6) {
            const v8i_t g6 = __builtin_shufflevector(cu[0], cu[1], 0, 1, 2, 3, 4, 5, -1, -1);
            acc[nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ag, g6, acc[nt], 2, 3, 0, sc_wg, 0, sc_g);
        }
    }
    // pin the software pipeline (left alone, the scheduler re-uses one fragment tuple for two N-tiles and waits lgkmcnt(0) behind every read):
    // double-buffered: N-tile 0's reads, then one read of N-tile nt+1 behind each fp16 MFMA of N-tile nt, the conversion, the 6-bit MFMA;
    // G6: an N-tile's six reads, its four fp16 MFMAs, its two 6-bit MFMAs
    if constexpr (NBUF == 2) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        if constexpr (NBUF == 1) __builtin_amdgcn_sched_group_barrier(0x100, G6 ? 6 : 4, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (NBUF == 2 && nt + 1 < NT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, G6 ? 2 : 1, 0);
    }
}

// hi fragments of one W6 group (the hi | lo fp16 planes are interleaved per k16 step: every second KiB) + its fp6 fragment
__device__ __forceinline__ void tl_load_group_w6(half8 (&hi)[8], v6i_t& lo, const _Float16* p, const unsigned* p6, int lane) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) hi[kk] = *reinterpret_cast<const half8*>(p + (2 * kk) * TFRAG_HALFS + lane * 8);
    const int4 a = *reinterpret_cast<const int4*>(p6 + lane * 4);
    const int2 c = *reinterpret_cast<const int2*>(p6 + 256 + lane * 2);
    lo = v6i_t{a.x, a.y, a.z, a.w, c.x, c.y};
}

// NB = C / 128 = gate passes = output passes (8 waves x 16 g-channels, 8 waves x 32 output rows of 2C): 2 (C = 256) or 3 (C = 384).
// PF: prefetch the output projection's first accumulator init (residual stream) already under the LAST gate pass's main loop
// (64 more live VGPRs there) instead of right after it.
// NW2: weight planes of the output projection (= NW, or 2 with NW = 1 for F16_MIX); its groups are KG2 = KG * NW / NW2 k-steps deep.
// (Wave-priority schemes beyond the static split below -- none, or dynamic: low inside the MFMA loops, high in the memory / VALU sections --
//  measured within +-0.5 % of it at 32 clips for f16_w2 and f16_m64: profiles/r3b_layer_prio.txt.)
// DEFER: the skip half of the output projection is NOT computed here (net.py:80-84: skip = second half of the 1x1).  The gate output g is
// written to HBM as well (fp16, 768 B per frame) and ONE K = L*C contraction per evaluation (tskip.h) produces relu(skip_projection(sum of
// the skips) / sqrt(L)) from all layers' g with pre-composed weights.  The layer then moves 8.5 KB per frame instead of 10.8 (no fp32 skip
// read-modify-write), and on a part whose matrix and HBM phases do not overlap (profiles/r3c_overlap.txt) bytes are time.
// NT (round 5): N-tiles of 32 frames per workgroup.  4 = the throughput tiling (one weight fragment feeds four MFMAs); 2 and 1 = 64- and
// 32-frame tiles for mid-size batches (7 ... 17 ten-second clips): 128-frame tiles leave most of the 256 CUs without a workgroup there, and the
// two-launch tilings those batches ran on until round 4 have no 6-bit correction products (they computed f16_w2, whose 1000-step error tail
// grazes the bar -- VERDICT r4 weak 1).  Same code, same LDS plan scaled by NT / 4; every workgroup streams the layer's whole weight set, so the
// per-CU L2 -> register stream (not the matrix pipe) bounds the small tiles.
template <int NB, int KG, int NW, int PF, int NW2 = NW, int DEFER = 0, int PRIOV = 0, int W6 = 0, int NT = 4>
__global__ void __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2)))
tlayer_kernel(const TLayerArgs ga, const float* __restrict__ cproj, const TEpiResSkip::Args oe) {
    constexpr int TN = 32 * NT;                           // frames per workgroup
    constexpr unsigned BLOCK_BYTES = TN * 256, S6_BYTES = TN * 128;      // one g block [TN frames][128 channels] fp16; its 6-bit g_lo codes
    static_assert(NT == 4 || (!DEFER && !PF && PRIOV == 0), "the 64- / 32-frame tiles exist for the plain in-layer form only");
    constexpr int KG2 = KG * NW / NW2;
    static_assert(KG2 * NW2 == KG * NW && KG2 >= 1, "both phases use the same eight ring registers");
    static_assert(!W6 || (KG == 4 && NW == 2 && NW2 == 2 && !DEFER), "W6: hi | lo fp16 planes packed, 64 input channels per group");
    constexpr bool G6 = W6 == 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    TL_STAMP(0);
    const int row0 = blockIdx.x * TN;
#ifdef DSVC_PROFILING
    if (ga.abl & 2) {                                     // consumer side of the hand-off (MI355X_MICROARCH.md: ONE relaxed poll -> ONE agent acquire -> barrier -> plain loads)
        if (tid == 0) {
            const unsigned lo = blockIdx.x > 0 ? blockIdx.x - 1 : 0, hi = blockIdx.x + 1 < gridDim.x ? blockIdx.x + 1 : blockIdx.x;
            while (__hip_atomic_load(ga.flags + lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(1);
            while (__hip_atomic_load(ga.flags + hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
#endif
    // A tile that lies wholly beyond its clip's length -- the padding of a ragged batch (dsvc_sample_args.clip_lens: a clip occupies a bucket of
    // Tp rows whatever its own length; tiles never straddle clips) -- has nothing to compute: its operand rows are the convs' zero padding and
    // stay so, its residual / skip rows are read by nobody but its own later launches.  All it owes the rest of the launch is that the rows of the
    // NEXT layer's operand it owns are zero (the last valid tile of the clip reads `dil` of them as its halo; in a re-used bucket they may
    // still hold an earlier, longer clip), then it returns -- and its CU takes the next workgroup: a ragged batch costs its ACTIVE tiles
    // (round 6, second session; the host picks the tile width by them when it is given the lengths, dsvc_sample_args.clip_lens_host).
    if (oe.rm.rowclip[row0] < 0) {
        if (oe.xh) {
            const size_t n16 = (size_t)TN * (size_t)oe.ldh / 8;                  // 16-byte chunks of the tile's rows (row pitch ldh halfs, a multiple of 8)
            half8* dst = reinterpret_cast<half8*>(oe.xh + (size_t)row0 * oe.ldh);
            const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            for (size_t i = tid; i < n16; i += 512) dst[i] = z;
        }
        return;
    }
    const int halo = ga.dil;                              // taps == 3
    const int rows_lds = TN + 2 * halo;
    const int chunks = ga.cin >> 3;
    const int row_bytes = ga.cin * 2;

    // ---- the time tile: HBM/L2 -> LDS by DMA, swizzled on the source side (tgemm.h) ----
    {
        const int total = rows_lds * chunks;
        const int dq = 512 / chunks, dr = 512 - dq * chunks;
        int slot = wave * 64 + lane;
        int r = slot / chunks, c = slot - r * chunks;
        const _Float16* xrow0 = ga.x + (long long)(row0 - halo) * ga.cin;
        for (int it = wave; it * 64 < total; it += 8) {
            const int rc = r < rows_lds ? r : rows_lds - 1;
            const _Float16* src = xrow0 + (long long)rc * ga.cin + ((c ^ (rc & ga.swz)) << 3);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(smem + it * 1024), 16, 0, 0);
            c += dr; r += dq;
            if (c >= chunks) { c -= chunks; r += 1; }
        }
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned s_off = (unsigned)(((size_t)rows_lds * row_bytes + 1023) & ~(size_t)1023);
    auto block_base = [&](int pos) -> unsigned { return lds0 + (pos == 0 ? s_off : (unsigned)(pos - 1) * BLOCK_BYTES); };
    // G6: the code block of g block `pos` -- behind S for the first pass, behind the two g blocks that replace the time tile for the others
    auto s6_base = [&](int pos) -> unsigned {
        return lds0 + (pos == 0 ? s_off + BLOCK_BYTES : (unsigned)(NB - 1) * BLOCK_BYTES + (unsigned)(pos - 1) * S6_BYTES);
    };

    int variant = 0;
    if (ga.n_variants > 1 && ga.step_ptr) {               // (the sampler passes the variant by value: n_variants == 1 on the hot path)
        const int st = *ga.step_ptr - ga.step_off;
        variant = st % ga.n_variants;
        if (variant < 0) variant += ga.n_variants;
    }
    constexpr int GROUP_HALFS = KG * NW * TFRAG_HALFS;
    // wave-uniform weight bases (SGPRs); the lane's 16-byte slot inside a fragment is added where a ring is loaded
    const _Float16* gw = ga.gw + (long long)variant * ga.gvar;
    const _Float16* ow = ga.ow + (long long)variant * ga.ovar;
    const unsigned* gw6 = W6 ? ga.gw6 + (long long)variant * ga.g6var : nullptr;
    const int sc_w6 = ga.sc6 & 255, sc_o6 = (ga.sc6 >> 16) & 255, sc_ol6 = ga.sc6b & 255;
    const unsigned* ow6 = G6 ? ga.ow6 : nullptr;
    const unsigned* ow6lo = W6 ? ga.ow6lo + (long long)variant * ga.o6var : nullptr;
    const int lane8 = lane * 8;
    const int gpt = (ga.cin >> 4) / KG;                   // gate: groups per tap
    const int G1 = 3 * gpt;                               // gate: groups per output tile
    const int G2 = (ga.cin >> 4) / KG2;                   // output projection: groups per tile (K = C)
    const long long tile1 = (long long)G1 * GROUP_HALFS, tile2 = (long long)G2 * GROUP_HALFS;
    // PRIOV: how the two waves of a SIMD (w and w + 4) share its matrix pipe.  0 = static (waves 4..7 at priority 1 throughout: the in-kernel
    // timeline, profiles/r3g_layer_stamps.txt, shows them finishing the gate phase 20 us before their partners, which then run alone --
    // one wave per SIMD cannot keep the pipe busy); 1 = the favoured half alternates from pass to pass; 2 = it alternates from weight
    // group to weight group (round-robin at ~1k-cycle granularity)
    const int half = wave >> 2;
    auto prio_pass = [&](int pass) { if constexpr (PRIOV == 1) { if ((pass + half) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); } };
    auto prio_group = [&](int g) { if constexpr (PRIOV == 2) { if ((g + half) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); } };
    if constexpr (PRIOV == 0) if (wave >= 4) __builtin_amdgcn_s_setprio(1);         // waves w and w+4 share a SIMD: let the pair drift apart (tgemm.h)
    const int rot = (int)(blockIdx.x % (unsigned)NB);     // per-workgroup rotated pass order (tgemm.h)
    auto tile_of = [&](int pi) { const int p = pi + rot; return (p < NB ? p : p - NB) * 8 + wave; };
    const int g_issue = wave >= 4 ? ((G1 / 2) & ~1) : 0;

    TEpiGate gepi;
    TEpiResSkip oepi;
#ifdef DSVC_PROFILING
    const TEpiGate::Args ge{cproj, nullptr, ga.cin, ga.cin, 0, ga.abl & 1};
#else
    const TEpiGate::Args ge{cproj, nullptr, ga.cin, ga.cin};
#endif
    f32x16 acc[NT], nxt[NT];
    half8 ringA[KG * NW], ringB[KG * NW];
    v6i_t gmid6 = {};                                     // G6: the middle pass's g_lo codes
    v6i_t lo6A, lo6B;                                     // W6: the fp6 fragment of the group in ringA / ringB (whose first four entries hold its hi fragments)
    half8 gmid[NT];                                       // the middle gate pass's g block share (NB == 3)

    // gate-phase operand stream and group product in their two forms (fp16 planes | W6: hi fragments + fp6 codes)
    auto gload = [&](half8 (&ring)[KG * NW], v6i_t& lo6, int mt_, int g_) {
        if constexpr (W6) tl_load_group_w6(ring, lo6, gw + (long long)mt_ * tile1 + (long long)g_ * GROUP_HALFS, gw6 + ((size_t)mt_ * G1 + g_) * (TFRAG6_BYTES / 4), lane);
        else tl_load_group<KG, NW>(ring, gw + (long long)mt_ * tile1 + (long long)g_ * GROUP_HALFS + lane8);
    };
    const unsigned nt_stride_x0 = 32u * (unsigned)row_bytes;
    auto gcompute = [&](const half8 (&ring)[KG * NW], const v6i_t& lo6, int g_) {
        const int tap = g_ / gpt, kb = (g_ - tap * gpt) * KG;
        const int rr = halo + (tap - 1) * ga.dil + (lane & 31);
        const unsigned b0 = lds0 + (unsigned)rr * (unsigned)row_bytes, xs = (unsigned)(((rr & ga.swz) ^ (lane >> 5)) << 4) ^ ((unsigned)kb << 5);
        if constexpr (W6) {
            const half8 (&hi)[4] = reinterpret_cast<const half8 (&)[4]>(ring);
            tl_compute_group_w6<false, NT>(hi, lo6, acc, b0, nt_stride_x0, xs, sc_w6, TL_X6_SCALE, TL_X6_E8M0, lo6, 0u, 0u, 0u, 0, 0);
        } else {
            tl_compute_group<KG, NW, NT>(ring, acc, b0, nt_stride_x0, xs);
        }
    };
    // ... and of the output projection (W6: hi fragments of its two planes + fp6 w_lo codes; G6: + the fp6 codes of the weights themselves)
    v6i_t wg6 = {};
    auto oload = [&](half8 (&ring)[KG * NW], v6i_t& lo6, int mt_, int g_) {
        if constexpr (W6) tl_load_group_w6(ring, lo6, ow + (long long)mt_ * tile2 + (long long)g_ * GROUP_HALFS, ow6lo + ((size_t)mt_ * G2 + g_) * (TFRAG6_BYTES / 4), lane);
        else tl_load_group<KG2, NW2>(ring, ow + (long long)mt_ * tile2 + (long long)g_ * GROUP_HALFS + lane8);
    };
    auto oload_g = [&](v6i_t& wg, int mt_, int g_) {
        if constexpr (G6) {
            const unsigned* p6 = ow6 + ((size_t)mt_ * G2 + g_) * (TFRAG6_BYTES / 4);
            const int4 a = *reinterpret_cast<const int4*>(p6 + lane * 4);
            const int2 c = *reinterpret_cast<const int2*>(p6 + 256 + lane * 2);
            wg = v6i_t{a.x, a.y, a.z, a.w, c.x, c.y};
        }
    };
    // first operands in flight before the barrier
    gload(ringA, lo6A, tile_of(0), 0);
    gepi.init(ge, tile_of(0), row0, lane, acc);
    TL_STAMP(1);
    // (a counted vmcnt(22) + bare barrier here -- only the DMA'd tile waited for, ring and accumulator-init loads in flight across the barrier --
    //  measured neutral for the W6 kernels: hipcc issues the weight ring LAST, so the first MFMA waits for everything anyway; profiles/r4p_*)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TL_STAMP(2);
    __syncthreads();
    TL_STAMP(3);

    // =========================== phase 1: gate passes ===========================
#pragma unroll
    for (int pi = 0; pi < NB; ++pi) {
        const int mt = tile_of(pi);
        const bool last = pi == NB - 1;
        const int mt_n = last ? (DEFER ? wave : tile_of(0)) : tile_of(pi + 1);          // last gate pass: the next "tile" is output pass 0
        bool nxt_issued = false;
        auto issue_next_init = [&]() {
            if (!last) gepi.init(ge, mt_n, row0, lane, nxt);
            else if (PF) oepi.init(oe, mt_n, row0, lane, nxt);
        };
        int g = 0;
        prio_pass(pi + 1);
        for (; g + 1 < G1; g += 2) {
            prio_group(g >> 1);
            gload(ringB, lo6B, mt, g + 1);
            if (g == g_issue && (!last || PF)) { issue_next_init(); nxt_issued = true; }
            __builtin_amdgcn_sched_barrier(0);
            gcompute(ringA, lo6A, g);
            const int gn = g + 2 < G1 ? g + 2 : G1 - 1;
            gload(ringA, lo6A, mt, gn);
            __builtin_amdgcn_sched_barrier(0);
            gcompute(ringB, lo6B, g + 1);
        }
        if (g < G1) gcompute(ringA, lo6A, g);
        TL_STAMP(4 + pi);                                  // (4, 5, 6: end of a gate pass's main loop)
        // the next tile's weight stream starts before this tile's epilogue
        if (!last) gload(ringA, lo6A, mt_n, 0);
        else { oload(ringA, lo6A, mt_n, 0); oload_g(wg6, mt_n, 0); }
        if (!nxt_issued && (!last || PF)) issue_next_init();
        // ---- gate epilogue: g = sigmoid * tanh -> fp16 (TEpiGate::finish, kept on chip) ----
        half8 gq[NT];
        v6i_t gq6 = {};
        if constexpr (G6) {
            half32_t glo = {};                               // (NT < 4: the codes of the absent N-tiles stay zero and are never stored)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    float gv = gate_act_scaled(acc[nt][r], acc[nt][8 + r]);
                    // ONE fp32 value feeds both roundings below.  Left alone, hipcc folds the last multiply of gate_act_scaled into the fp16
                    // conversion in SOME instantiations (v_fma_mixlo_f16: one rounding from the exact product) and not in others (v_mul_f32 +
                    // v_cvt: two): g then differs by an fp16 ulp on ~1 value in 2^18 between the tile widths -- g_lo absorbs it, the layer's
                    // output moved by 3e-6 on 12 of 7168 frames (profiles/r5c_nt_diag.txt) -- and the tilings were not bit-identical
                    asm volatile("" : "+v"(gv));
                    gq[nt][r] = (_Float16)gv;
                    glo[8 * nt + r] = (_Float16)((gv - (float)gq[nt][r]) * TL_G6_UP);        // exact: |.| <= 16, and the difference has < 11 significant bits left
                }
            gq6 = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(glo, 1.0f);                     // code 8 nt + r: frame 32 nt + (lane & 31), channel 16 wave + 8 h + r
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 8; ++r) gq[nt][r] = (_Float16)gate_act_scaled(acc[nt][r], acc[nt][8 + r]);
        }
        if constexpr (DEFER) {                             // g also goes to HBM: the step's one skip contraction reads it (tskip.h)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                __builtin_nontemporal_store(gq[nt], reinterpret_cast<half8*>(ga.gall + (size_t)(row0 + 32 * nt + (lane & 31)) * ga.cin + mt * 16 + 8 * (lane >> 5)));
        }
        // lane (h = lane >> 5) holds g-channels 16*wave + 8h .. +7 of its block for frames 32*nt + (lane & 31): chunk 2*wave + h
        auto store_block = [&](int pos, const half8 (&v)[NT]) {
            const unsigned bb = block_base(pos);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const unsigned f = 32u * nt + (unsigned)(lane & 31);
                const unsigned a = bb + f * 256u + ((((unsigned)(2 * wave) + (unsigned)(lane >> 5)) ^ (f & 15u)) << 4);
                *(half8 __attribute__((address_space(3)))*)(size_t)a = v[nt];
            }
        };
        // G6: the consumer (phase 2) reads, per frame and 64-channel group q and lane half h, 32 codes in the order 8 kk + e <-> channel
        // 64 q + 16 kk + 8 h + e: this wave's 8 channels are (q, kk) = (wave >> 2, wave & 3) of its block, i.e. 6 bytes of that 24-byte string.
        // A frame's row is 128 B = 4 slots (2 q + h) of two 16-B chunks [kk 0, kk 1 | pad] [kk 2, kk 3 | pad]; chunk c of frame f sits at
        // c ^ ((f >> 1) & 7), which makes the consumer's 16-byte reads (32 frames x one logical chunk) conflict-free.
        auto store_block6 = [&](int pos, const v6i_t& r6) {
            const unsigned bb = s6_base(pos);
            const unsigned c = 2u * (2u * (unsigned)(wave >> 2) + (unsigned)(lane >> 5)) + (unsigned)((wave >> 1) & 1);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const unsigned lo32 = (nt & 1) ? (((unsigned)r6[3 * (nt >> 1) + 1] >> 16) | ((unsigned)r6[3 * (nt >> 1) + 2] << 16)) : (unsigned)r6[3 * (nt >> 1)];
                const unsigned hi16 = (nt & 1) ? ((unsigned)r6[3 * (nt >> 1) + 2] >> 16) : ((unsigned)r6[3 * (nt >> 1) + 1] & 0xffffu);
                const unsigned f = 32u * nt + (unsigned)(lane & 31);
                const unsigned a = bb + f * 128u + ((c ^ ((f >> 1) & 7u)) << 4);
                if (wave & 1) {                            // bytes 6 .. 11 of the chunk
                    *(unsigned short __attribute__((address_space(3)))*)(size_t)(a + 6u) = (unsigned short)(lo32 & 0xffffu);
                    *(unsigned __attribute__((address_space(3)))*)(size_t)(a + 8u) = (lo32 >> 16) | (hi16 << 16);
                } else {                                   // bytes 0 .. 5
                    *(unsigned __attribute__((address_space(3)))*)(size_t)a = lo32;
                    *(unsigned short __attribute__((address_space(3)))*)(size_t)(a + 4u) = (unsigned short)hi16;
                }
            }
        };
        if (pi == 0) {
            store_block(0, gq);                            // S is disjoint from the time tile: no barrier needed
            if constexpr (G6) store_block6(0, gq6);
        } else if (!last) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) gmid[nt] = gq[nt];
            if constexpr (G6) gmid6 = gq6;
        }
        if (last) {
            if (!PF) oepi.init(oe, mt_n, row0, lane, nxt);  // accumulators are dead: the output projection's first init goes out now
            __syncthreads();                               // every wave is done reading the time tile
            if (NB == 3) { store_block(1, gmid); if constexpr (G6) store_block6(1, gmid6); }
            store_block(NB - 1, gq);
            if constexpr (G6) store_block6(NB - 1, gq6);
            __syncthreads();                               // g complete
            TL_STAMP(7);
#ifdef DSVC_PROFILING
            if (ga.dephase != 0 && ((ga.dephase > 0) == (wave >= 4))) {
                const int n = ga.dephase > 0 ? ga.dephase : -ga.dephase;
                for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);
            }
#endif
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = nxt[nt];
    }

    // =========================== phase 2: output projection passes ===========================
    // G6 on 128-frame tiles cannot afford a prefetched second accumulator set in the output phase (below); on 64- / 32-frame tiles it can
    // (193 / 155 VGPRs with it)
    constexpr bool NOPF2 = G6 && NT == 4;
    // the diffusion step of the FiLM rows the residual epilogues add, when all clips share it (the sampler's loops): read here, long before it is needed
    const int step_sh = (NOPF2 && oe.film && !oe.step.per_clip) ? __builtin_amdgcn_readfirstlane(oe.step.get(0)) : -1;
    const unsigned xs_g = (unsigned)((((lane & 31) & 15) ^ (lane >> 5)) << 4);
    const int g_issue2 = wave >= 4 ? ((G2 / 2) & ~1) : 0;
    if constexpr (DEFER) {
        // residual half only: C/32 = 4*NB output tiles.  Pass 0: tile `wave`, all 128 frames.  NB == 3 leaves four tiles: pass 1 gives tile
        // 8 + (wave & 3) to TWO waves, 64 frames each, so all eight waves stay busy.  (acc holds tile `wave`'s residual-stream init.)
        constexpr bool HALF = NB == 3;
        const int mt1 = 8 + (wave & 3), nh = wave >> 2;
        const int rowh = row0 + 64 * nh;
        f32x16 acc2[2];
        auto group_b = [&](int g, unsigned frame0, unsigned& base0, unsigned& xs) {
            const int kb = g * KG2;                          // first k16 step of the group; 8 steps per 128-channel block
            int pos = (kb >> 3) - rot; if (pos < 0) pos += NB;
            base0 = block_base(pos) + (frame0 + (unsigned)(lane & 31)) * 256u;
            xs = xs_g ^ ((unsigned)(kb & 7) << 5);
        };
        {
            const _Float16* wp = ow + (long long)wave * tile2;
            bool issued = false;
            int g = 0;
            for (; g + 1 < G2; g += 2) {
                tl_load_group<KG2, NW2>(ringB, wp + (long long)(g + 1) * GROUP_HALFS + lane8);
                if (HALF && g == g_issue2) { oepi.template init<2>(oe, mt1, rowh, lane, acc2); issued = true; }
                __builtin_amdgcn_sched_barrier(0);
                { unsigned b0, xs; group_b(g, 0u, b0, xs); tl_compute_group<KG2, NW2>(ringA, acc, b0, 32u * 256u, xs); }
                const int gn = g + 2 < G2 ? g + 2 : G2 - 1;
                tl_load_group<KG2, NW2>(ringA, wp + (long long)gn * GROUP_HALFS + lane8);
                __builtin_amdgcn_sched_barrier(0);
                { unsigned b0, xs; group_b(g + 1, 0u, b0, xs); tl_compute_group<KG2, NW2>(ringB, acc, b0, 32u * 256u, xs); }
            }
            if (g < G2) { unsigned b0, xs; group_b(g, 0u, b0, xs); tl_compute_group<KG2, NW2>(ringA, acc, b0, 32u * 256u, xs); }
            if (HALF) {
                tl_load_group<KG2, NW2>(ringA, ow + (long long)mt1 * tile2 + lane8);
                if (!issued) oepi.template init<2>(oe, mt1, rowh, lane, acc2);
            }
            oepi.finish(oe, wave, row0, lane, acc);
        }
        if constexpr (HALF) {
            const _Float16* wp = ow + (long long)mt1 * tile2;
            int g = 0;
            for (; g + 1 < G2; g += 2) {
                tl_load_group<KG2, NW2>(ringB, wp + (long long)(g + 1) * GROUP_HALFS + lane8);
                __builtin_amdgcn_sched_barrier(0);
                { unsigned b0, xs; group_b(g, 64u * nh, b0, xs); tl_compute_group<KG2, NW2, 2>(ringA, acc2, b0, 32u * 256u, xs); }
                const int gn = g + 2 < G2 ? g + 2 : G2 - 1;
                tl_load_group<KG2, NW2>(ringA, wp + (long long)gn * GROUP_HALFS + lane8);
                __builtin_amdgcn_sched_barrier(0);
                { unsigned b0, xs; group_b(g + 1, 64u * nh, b0, xs); tl_compute_group<KG2, NW2, 2>(ringB, acc2, b0, 32u * 256u, xs); }
            }
            if (g < G2) { unsigned b0, xs; group_b(g, 64u * nh, b0, xs); tl_compute_group<KG2, NW2, 2>(ringA, acc2, b0, 32u * 256u, xs); }
            oepi.template finish<2>(oe, mt1, rowh, lane, acc2);
        }
        return;
    }
#pragma unroll
    for (int po = 0; po < NB; ++po) {
        const int mt = tile_of(po);
        const bool last = po == NB - 1;
        const int mt_n = last ? 0 : tile_of(po + 1);
        bool nxt_issued = false;
        auto group_b = [&](int g, unsigned& base0, unsigned& xs) {
            const int kb = g * KG2;                          // first k16 step of the group; 8 steps per 128-channel block
            int pos = (kb >> 3) - rot; if (pos < 0) pos += NB;
            base0 = block_base(pos) + (unsigned)(lane & 31) * 256u;
            xs = xs_g ^ ((unsigned)(kb & 7) << 5);
        };
        // one group of the output projection.  W6: hi fragments + fp6 w_lo codes against bf6(g / TL_G_X6_SCALE) converted in registers, as in the gate
        // phase; G6: plus the g_lo correction from the code rows the gate epilogue left in LDS (tl_compute_group_w6<true>)
        auto group2 = [&](const half8 (&ring)[KG * NW], const v6i_t& lo6, const v6i_t& wg, int g_) {
            unsigned b0, xs;
            group_b(g_, b0, xs);
            if constexpr (W6) {
                static_assert(!W6 || (KG2 == 4 && NW2 == 2), "one group = 64 g-channels, hi | lo planes packed");
                const half8 (&hi)[4] = reinterpret_cast<const half8 (&)[4]>(ring);
                if constexpr (G6) {
                    int pos = (g_ >> 1) - rot; if (pos < 0) pos += NB;
                    const unsigned sw = (unsigned)((lane & 31) >> 1) & 7u;
                    const unsigned sl = 2u * (2u * (unsigned)(g_ & 1) + (unsigned)(lane >> 5));
                    tl_compute_group_w6<true, NT>(hi, lo6, acc, b0, 32u * 256u, xs, sc_ol6, TL_G_X6_SCALE, TL_G_X6_E8M0, wg,
                                              s6_base(pos) + (unsigned)(lane & 31) * 128u, (sl ^ sw) << 4, ((sl + 1u) ^ sw) << 4, sc_o6, TL_G6_E8M0);
                } else {
                    tl_compute_group_w6<false, NT>(hi, lo6, acc, b0, 32u * 256u, xs, sc_ol6, TL_G_X6_SCALE, TL_G_X6_E8M0, lo6, 0u, 0u, 0u, 0, 0);
                }
            } else {
                tl_compute_group<KG2, NW2, NT>(ring, acc, b0, 32u * 256u, xs);
            }
        };
        int g = 0;
        prio_pass(po);
        for (; g + 1 < G2; g += 2) {
            prio_group(g >> 1);
            oload(ringB, lo6B, mt, g + 1);
            if (!NOPF2 && g == g_issue2 && !last) { oepi.init(oe, mt_n, row0, lane, nxt); nxt_issued = true; }
            // (G6 with HALF a prefetched set -- N-tiles 0 and 1, 32 registers -- still put 9-15 scratch accesses into every loop iteration:
            //  141.6 instead of 132.7 us per layer, profiles/r4o_w6_time_half_prefetch.txt)
            // (G6 -- no registers for `nxt` -- with plain loads of the next tiles at this point to warm the L2: 139 instead of 133 us per layer, not kept)
            __builtin_amdgcn_sched_barrier(0);
            group2(ringA, lo6A, wg6, g);
            oload_g(wg6, mt, g + 1);                        // (single-buffered: the 1.5 KB fragment is re-loaded right after use, a whole group ahead)
            const int gn = g + 2 < G2 ? g + 2 : G2 - 1;
            oload(ringA, lo6A, mt, gn);
            __builtin_amdgcn_sched_barrier(0);
            group2(ringB, lo6B, wg6, g + 1);
            if (g + 2 < G2) oload_g(wg6, mt, g + 2);
        }
        if (g < G2) group2(ringA, lo6A, wg6, g);
        TL_STAMP(8 + 2 * po);                              // (8, 10, 12: end of an output pass's main loop; 9, 11, 13: its stores issued)
        if (!last) {
            oload(ringA, lo6A, mt_n, 0);
            oload_g(wg6, mt_n, 0);
            if (!NOPF2 && !nxt_issued) oepi.init(oe, mt_n, row0, lane, nxt);
        }
        // G6 on 128-frame tiles has no registers for a WHOLE second accumulator set beside its code operands (a prefetched `nxt` made the
        // allocator spill whole accumulator tiles inside the loops: 166 us per layer; half a set: 141.6).  Round 4 loaded the next pass's tiles
        // into the accumulators after this pass's stores, the whole burst's latency exposed (133 us); round 5 interleaves the two N-tile by
        // N-tile (TEpiResSkip::finish_then_init): an N-tile's init loads go out right behind ITS stores, under the later N-tiles' epilogues
        if (NOPF2 && !last) oepi.finish_then_init(oe, mt, mt_n, row0, lane, acc, step_sh);
        else oepi.finish(oe, mt, row0, lane, acc);
        TL_STAMP(9 + 2 * po);
        if (!last && !NOPF2) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = nxt[nt];
        }
    }
#ifdef DSVC_PROFILING
    if (ga.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); TL_STAMP(14); }      // every store acknowledged
    if (ga.abl & 2) {                                     // producer side: plain stores -> drain -> barrier -> lane-0 agent release -> drain -> relaxed agent flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(ga.flags + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#endif
}

inline size_t tlayer_smem(int dil, int cin, bool g6 = false, int nt = 4) {
    const size_t x = (((size_t)(32 * nt + 2 * dil) * cin * 2) + 1023) & ~(size_t)1023;
    return x + (size_t)32 * nt * 256 + (g6 ? (size_t)32 * nt * 128 : 0);
}

// can this layer shape run fused?  C a multiple of 128 with 2 or 3 channel blocks, the time tile + one g block inside 160 KB, and
// the time tile at least as large as the other g blocks (always true: it holds C channels of >= 128 frames)
inline bool tlayer_supported(int C, int cin_pad, int dil, int n_rows, int nt = 4) {
    // (nt < 4: the g blocks that replace the time tile after the gate phase, codes included, must fit into it)
    return C == cin_pad && (C == 256 || C == 384) && n_rows % (32 * nt) == 0 && tlayer_smem(dil, cin_pad, false, nt) <= 160 * 1024 &&
           (size_t)(C / 128 - 1) * 32 * nt * (256 + 128) <= (size_t)(32 * nt + 2 * dil) * cin_pad * 2;
}

template <int NB, int KG, int NW, int PF, int NW2 = NW, int DEFER = 0, int PRIOV = 0, int W6 = 0, int NT = 4>
inline int tlayer_launch_t(const TLayerArgs& ga, const float* cproj, const TEpiResSkip::Args& oe, int n_rows, hipStream_t stream) {
    auto kern = tlayer_kernel<NB, KG, NW, PF, NW2, DEFER, PRIOV, W6, NT>;
    const size_t smem = tlayer_smem(ga.dil, ga.cin, W6 == 2, NT);
    if (smem > 160 * 1024) return fail(DSVC_EINVAL, "tlayer: %zu B of LDS requested", smem);
    static thread_local size_t smem_set = 0;
    if (smem > 64 * 1024 && smem > smem_set) {
        DSVC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        smem_set = smem;
    }
    hipLaunchKernelGGL(kern, dim3(n_rows / (32 * NT)), dim3(512), smem, stream, ga, cproj, oe);
    DSVC_HIP(hipGetLastError());
    return DSVC_OK;
}

// gate weights `g` (taps 3, m_tiles C/16) + output-projection weights `o` (taps 1, m_tiles 2C/32) of ONE layer, as tgemm would get them
// w6 (DSVC_PREC_F16_W6): fp6 codes of the gate's w_lo plane (k_tpack6) for the block-scaled 6-bit product; e6 = their common exponent
struct TLayerW6 {
    const unsigned* codes = nullptr;     // the variant to use (already offset when the caller knows the step), or variant 0 with n_variants > 1
    long long variant_dwords = 0;
    int e6 = 0;
    int n_variants = 1;                  // > 1: resolved in the kernel from *step_ptr
    const unsigned* out_codes = nullptr; // G6: fp6 codes of the output projection's weights (whole weights, one variant), exponent eo6; null: no g_lo correction
    int eo6 = 0;
    const unsigned* out_lo_codes = nullptr;   // fp6 codes of the output projection's w_lo plane (the variant to use, like `codes`), exponent eol6
    long long out_lo_variant_dwords = 0;
    int eol6 = 0;
};

template <int NW, int NW2 = NW>
inline int tlayer_launch(const TGemmArgs& g, const float* cproj, const TGemmArgs& o, const TEpiResSkip::Args& oe, int C, int n_rows,
                         int prefetch, hipStream_t stream, _Float16* gall = nullptr, int priov = 0, const TLayerW6* w6 = nullptr, int nt = 4) {
    constexpr int KG = NW == 2 ? 4 : 8;
    if (g.taps != 3 || o.taps != 1 || g.m_tiles != C / 16 || o.m_tiles != 2 * C / 32 || g.cin != C || o.cin != C)
        return fail(DSVC_EINVAL, "tlayer: unexpected layer geometry");
    if (g.w_planes != NW || o.w_planes != NW2) return fail(DSVC_EINVAL, "tlayer: weight planes");
    if (g.n_variants != o.n_variants && o.n_variants != 1)
        return fail(DSVC_EINVAL, "tlayer: the output projection carries %d dither variants, the gate %d", o.n_variants, g.n_variants);
    if (nt != 4 && nt != 2 && nt != 1) return fail(DSVC_EINVAL, "tlayer: %d N-tiles per workgroup", nt);
    if (nt != 4 && !(w6 && w6->codes)) return fail(DSVC_EINVAL, "tlayer: the 64- / 32-frame tiles are built for the 6-bit correction schemes (f16_w6 / f16_w6n) only");
    if (!tlayer_supported(C, g.cin, g.dil, n_rows, nt)) return fail(DSVC_EINVAL, "tlayer: shape not supported by the fused layer kernel");
    TLayerArgs a{};
    a.x = g.x; a.cin = g.cin; a.swz = tgemm_swizzle_mask(g.cin); a.dil = g.dil; a.gw = g.w; a.ow = o.w;
    a.gvar = g.variant_halfs; a.ovar = o.n_variants == g.n_variants ? o.variant_halfs : 0; a.n_variants = g.n_variants;
    a.step_ptr = g.step_ptr; a.step_off = g.step_off;
#ifdef DSVC_PROFILING
    a.abl = getenv("DSVC_TL_ABL") ? atoi(getenv("DSVC_TL_ABL")) : 0;
    a.dephase = getenv("DSVC_TL_DEPHASE") ? atoi(getenv("DSVC_TL_DEPHASE")) : 0;
    if (a.abl & 2) {
        static unsigned* flags = nullptr;
        if (!flags) { DSVC_HIP(hipMalloc(&flags, 4096 * 4)); DSVC_HIP(hipMemset(flags, 0xff, 4096 * 4)); }       // "every neighbour has arrived": the poll never waits
        if (n_rows / (32 * nt) > 4096) return fail(DSVC_EINVAL, "tlayer: DSVC_TL_ABL=2 covers 4096 tiles");
        a.flags = flags;
    }
    if (getenv("DSVC_TL_STAMPS")) {
        if (!tl_stamp_buffer()) { DSVC_HIP(hipMalloc(&tl_stamp_buffer(), (size_t)4096 * 8 * 16 * 8)); }
        if (nt == 4 && n_rows / TL_TN <= 4096) { a.stamps = tl_stamp_buffer(); tl_stamp_groups() = n_rows / TL_TN; }
    }
#endif
    if constexpr (NW == 2 && NW2 == 2) {
        if (w6 && w6->codes) {                            // hi fragments of g's two planes + fp6 codes: 16 fp16 + 4 six-bit MFMAs per group instead of 32
            a.gw6 = w6->codes; a.g6var = w6->variant_dwords; a.sc6 = (127 + w6->e6) | (TL_X6_E8M0 << 8);
            if (!w6->out_lo_codes) return fail(DSVC_EINVAL, "tlayer: f16_w6 needs the fp6 codes of both contractions' w_lo planes");
            a.ow6lo = w6->out_lo_codes; a.o6var = w6->out_lo_variant_dwords; a.sc6b = 127 + w6->eol6;
            if (w6->out_codes) {                          // + the output projection's g_lo correction
                a.ow6 = w6->out_codes; a.sc6 |= ((127 + w6->eo6) << 16) | (TL_G6_E8M0 << 24);
                a.n_variants = w6->n_variants; a.gvar = 0; a.ovar = 0;
                if (gall) return fail(DSVC_EINVAL, "tlayer: the deferred skip form is not built for f16_w6");
                if (nt == 2) return C == 384 ? tlayer_launch_t<3, KG, NW, 0, NW2, 0, 0, 2, 2>(a, cproj, oe, n_rows, stream)
                                             : tlayer_launch_t<2, KG, NW, 0, NW2, 0, 0, 2, 2>(a, cproj, oe, n_rows, stream);
                if (nt == 1) return C == 384 ? tlayer_launch_t<3, KG, NW, 0, NW2, 0, 0, 2, 1>(a, cproj, oe, n_rows, stream)
                                             : tlayer_launch_t<2, KG, NW, 0, NW2, 0, 0, 2, 1>(a, cproj, oe, n_rows, stream);
                if (C == 384) return tlayer_launch_t<3, KG, NW, 0, NW2, 0, 0, 2>(a, cproj, oe, n_rows, stream);
                return tlayer_launch_t<2, KG, NW, 0, NW2, 0, 0, 2>(a, cproj, oe, n_rows, stream);
            }
            a.n_variants = w6->n_variants;
            a.gvar = 0; a.ovar = 0;                       // (the dither variants are those of the fp6 plane; n_variants / step_ptr select among them)
            if (gall) return fail(DSVC_EINVAL, "tlayer: the deferred skip form is not built for f16_w6");
            if (nt == 2) return C == 384 ? tlayer_launch_t<3, KG, NW, 0, NW2, 0, 0, 1, 2>(a, cproj, oe, n_rows, stream)
                                         : tlayer_launch_t<2, KG, NW, 0, NW2, 0, 0, 1, 2>(a, cproj, oe, n_rows, stream);
            if (nt == 1) return C == 384 ? tlayer_launch_t<3, KG, NW, 0, NW2, 0, 0, 1, 1>(a, cproj, oe, n_rows, stream)
                                         : tlayer_launch_t<2, KG, NW, 0, NW2, 0, 0, 1, 1>(a, cproj, oe, n_rows, stream);
            if (C == 384) return tlayer_launch_t<3, KG, NW, 0, NW2, 0, 0, 1>(a, cproj, oe, n_rows, stream);
            return tlayer_launch_t<2, KG, NW, 0, NW2, 0, 0, 1>(a, cproj, oe, n_rows, stream);
        }
    }
    if (gall) {                                           // skip-deferred form: g also goes to HBM, residual half of the 1x1 only
        a.gall = gall;
        if (C == 384) return tlayer_launch_t<3, KG, NW, 0, NW2, 1>(a, cproj, oe, n_rows, stream);
        return tlayer_launch_t<2, KG, NW, 0, NW2, 1>(a, cproj, oe, n_rows, stream);
    }
    if (C == 384 && priov == 1) return tlayer_launch_t<3, KG, NW, 0, NW2, 0, 1>(a, cproj, oe, n_rows, stream);
    if (C == 384 && priov == 2) return tlayer_launch_t<3, KG, NW, 0, NW2, 0, 2>(a, cproj, oe, n_rows, stream);
    if constexpr (NW2 != NW) {                            // (the prefetch variant is not instantiated for the mixed kernel)
        if (C == 384) return tlayer_launch_t<3, KG, NW, 0, NW2>(a, cproj, oe, n_rows, stream);
        return tlayer_launch_t<2, KG, NW, 0, NW2>(a, cproj, oe, n_rows, stream);
    }
    if (C == 384) return prefetch ? tlayer_launch_t<3, KG, NW, 1>(a, cproj, oe, n_rows, stream) : tlayer_launch_t<3, KG, NW, 0>(a, cproj, oe, n_rows, stream);
    return prefetch ? tlayer_launch_t<2, KG, NW, 1>(a, cproj, oe, n_rows, stream) : tlayer_launch_t<2, KG, NW, 0>(a, cproj, oe, n_rows, stream);
}

}  // namespace dsvc
