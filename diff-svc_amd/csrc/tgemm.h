// tgemm.h -- the throughput MFMA engine of the DiffNet hot loop (gfx950 / CDNA4, wave64).
//
//   out[ch][frame] = EPI( INIT[ch][frame] + sum_{tap} sum_{ci}  W[ch][tap][ci] * xh[frame + (tap - taps/2)*dil][ci] )
//
// i.e. the same conv-as-GEMM contraction as conv_gemm.h, but TRANSPOSED and fed differently:
//
//   * activations live in HBM/L2 as fp16 frame-major rows ([frame][channel], already FiLM-shifted and rounded by the
//     producing kernel's epilogue), so a workgroup's (TN + 2*halo) x K time tile is copied ONCE, straight into LDS,
//     by the LDS-DMA path (global_load_lds_dwordx4: no VGPR round trip, no conversion VALU).  The DMA writes LDS
//     lane-linearly, so the bank swizzle is applied on the SOURCE side: LDS slot (row r, 16-B chunk c') holds global
//     chunk c' ^ (r & swz); readers XOR the same value, which makes every ds_read_b128 lane group hit 16 distinct
//     16-B slots (conflict-free) although a row is a multiple of 256 B.
//   * the tile stays resident for the whole workgroup: every tap and every output-channel pass re-reads it, and the
//     main loop contains NO barrier -- the 8 waves (two per SIMD) drift apart, so one wave's epilogue VALU and
//     memory latency hide under the other wave's MFMAs.
//   * weights are the MFMA *A* operand (rows = output channels) and never touch LDS: they are packed on the host in
//     fragment order [m_tile][step][plane][lane 64][8 halfs]; a wave streams its own 32 output channels as fully
//     coalesced 1 KiB loads through a KG-deep register ring.  One weight fragment feeds NT_N (=4) MFMAs, so at full
//     MFMA rate a CU pulls 32 B/clk of weights from L2 -- half of what the L1 can deliver.
//   * D = W * X^T puts 16 CONSECUTIVE channels of one frame in a lane's accumulator (with the row permutation the
//     packer applies), so every epilogue load/store is a 16-byte access in the frame-major layouts, and the
//     epilogue's read-modify-write inputs (conditioner projection, residual stream, skip sum) are loaded straight
//     into the accumulators before the main loop: "acc init" replaces an epilogue read.
//   * math: v_mfma_f32_32x32x16_f16, fp32 accumulate.  NW = 2 adds the w_lo plane (w = w_hi + w_lo).  With one plane
//     the systematic weight-rounding error would cost ~6e-3 of mel after 1000 DDPM steps, so the packer can emit
//     n_variants differently-rounded copies (time-dithered rounding: step t uses copy t % n_variants; the copies
//     average to w), which keeps 1 MFMA per product inside the 1e-3 bar.
//   * small batches: the output-channel passes are spread over blockIdx.y, frame tiles start their K loop at staggered
//     groups, and for a single clip (KS = 3) a tile's K loop is split over three waves that reduce through LDS -- the
//     latency regime, analysed with the in-kernel stamps (DSVC_TG_STAMPS) in DESIGN.md 4.1.
#pragma once
#include <stdio.h>
#include <stdlib.h>

#include <string>
#include <type_traits>
#include <vector>

#include "common.h"

namespace dsvc {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TFRAG_HALFS = 512;          // one packed [lane 64][8 halfs] fragment

struct TGemmArgs {
    const _Float16* x;      // fp16 activations, row 0 of [guard | rows | guard][cin]; guard/gap rows are zero
    int cin;                // channels per row (K per tap), multiple of 16
    int swz;                // chunk swizzle mask (15, 7, 3, 1 or 0): largest 2^b-1 <= 15 with (cin/8) % 2^b == 0
    int taps, dil;          // tap offset = (tap - taps/2) * dil rows
    const _Float16* w;      // packed weights [variant][m_tile][taps*cin/16][planes][lane][8]
    int m_tiles;            // 32-row output-channel tiles
    int w_planes;           // planes stored per fragment (>= NW)
    long long variant_halfs;// stride between dither variants
    int n_variants;         // >= 1
    unsigned skip_lo;       // (ragged batches: low word of the rowclip pointer, see skip_hi)
    const int* step_ptr;    // device int: current diffusion step (variant = (step - step_off) % n_variants); may be null
    int step_off;           // (a launcher that knows the step passes w already offset to the variant and n_variants = 1: that takes the
                            //  dependent scalar load, ~0.4 us per kernel in the single-clip regime, out of the front of the weight stream)
    int clip_rows;          // rows per clip (clip stride) or 0: the K-loop stagger is keyed on a tile's position inside its clip, so a
                            // clip computes bit-identically alone and inside a batch
    // W6 kernels (round 4; the small-batch split-activation tilings): the w_lo * x_hi term on the block-scaled 6-bit MFMA.  `w6` = fp6 codes of
    // the w_lo plane (k_tpack6: one 1536-B fragment per (m_tile, tap, 64 input channels); the variant to use), sc6 = the E8M0 bytes of the
    // weight scale | the activation scale << 8, x6_scale = what the activations are divided by before the bf6 conversion
    const unsigned* w6;
    int sc6;
    float x6_scale;
    // KP kernels (round 4; the trainer's data-gradient GEMMs, whose 2C-channel split rows do not fit LDS whole): input channels per plane that
    // one K phase brings into LDS (% 128 == 0, divides cin); the phases are double-buffered, see the KP path of the kernel
    int kp_cin;
    // ragged batches (round 6; the sampler's plain tilings only -- KS == 1, no KP, FS == 1): rowclip table of the launch or null.  A workgroup
    // whose frame tile lies wholly beyond its clip's length (rowclip[first row] < 0: tiles never straddle clips, a clip's valid rows are a
    // prefix of its bucket) returns at once.  Everything such a tile would write is read by its own rows' later launches only -- every
    // contraction is over channels -- except the operand rows a conv's halo reads, which the sampler clears once per ragged call.
    // The pointer travels as two words in the struct's two alignment holes (skip_lo above, skip_hi here): the kernel arguments of every
    // launch keep the size and the offsets they had (a ninth 8-byte field behind kp_cin moved the epilogue's arguments by 8 bytes: the single clip's
    // gate kernel then fetches them with an s_load_dwordx4 at kernarg + 0x78, which straddles into a third 64-byte line -- a serialized scalar-cache
    // miss in front of the weight stream, measured +1.5 % on the headline, profiles/r6ab_skip_field_lib_ab.txt; the ISA differs in nothing else).
    unsigned skip_hi;
    __host__ __device__ void set_skip_rowclip(const int* p) {
        const unsigned long long v = (unsigned long long)p;
        skip_lo = (unsigned)v; skip_hi = (unsigned)(v >> 32);
    }
    __device__ __forceinline__ const int* skip_rowclip() const { return (const int*)(((unsigned long long)skip_hi << 32) | skip_lo); }
#ifdef DSVC_PROFILING       // profiling builds only (python -m diffsvc_amd.build --profiling): the product library has neither field nor branch
    unsigned long long* stamps;   // per-wave phase time stamps (s_memrealtime, 100 MHz), 16 per wave (env DSVC_TG_STAMPS)
    int dbg;                // ablation knobs (env DSVC_TG_DEBUG; results are WRONG when set): 1 = no acc-init loads, 2 = no epilogue,
                            // 4 = no tile DMA, 8 = no MFMA main loop, 16 = no wave priority split, 32 = all tiles stream
                            // tile 0's weights (L2-hot), 64 = no pass rotation, 128 = next-tile init loads issued at the
                            // end of the pass (both waves of a SIMD together) instead of staggered inside it, 1024 = no K-loop stagger
#endif
};

#ifndef DSVC_PROFILING
static_assert(sizeof(TGemmArgs) == 96, "TGemmArgs: the kernel-argument layout is part of the single clip's launch cost (see skip_hi)");
#endif

#ifdef DSVC_PROFILING
#define TG_DBG(a, bit) ((a).dbg & (bit))
#define TG_STAMPS(a) ((a).stamps)
#else
#define TG_DBG(a, bit) (0)
#define TG_STAMPS(a) ((unsigned long long*)nullptr)
#endif

// the row <-> channel permutation inside a 32-row output tile that makes a lane's 16 accumulator registers hold 16
// consecutive channels:  tile row i = 4h + 8j + e  (h = lane>>5, j = reg>>2, e = reg&3)  <->  channel 16h + 4j + e
__host__ __device__ inline int trow_to_ch16(int i) { return 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3); }
// paired variant (gate | filter): rows 0..15 and 16..31 each hold 16 channels, 8 consecutive ones per lane half:
// row i (< 16) = 4h + 8j + e  <->  channel 8h + 4j + e
__host__ __device__ inline int trow_to_ch8(int i) { return 8 * ((i >> 2) & 1) + 4 * ((i >> 3) & 1) + (i & 3); }

// KS > 1 (small batches only, one output tile per wave): the K loop of a tile is split over KS waves, which reduce through
// LDS before the epilogue -- a lone wave per SIMD issues its loads, LDS reads and MFMAs strictly one after the other
// (measured ~85 cycles per k-step), so at B = 1 the way to shorten a kernel is more waves per tile, not a better loop.
// NA = 2 (round 3, the fp32-class scheme on this engine: DSVC_PREC_F16_X3T): the activation rows hold TWO fp16 planes [x_hi | x_lo] (row =
// 2 * cin halfs, x_lo = fp16(x - x_hi)); a k-step issues W_hi x_hi + W_lo x_hi + W_hi x_lo (the 2^-22 W_lo x_lo term is dropped, as conv_gemm's
// split scheme does): 3 MFMAs per product from the SAME weight fragments -- the weight stream, which is what bounds the small-batch
// kernels, is that of f16_w2.
typedef int tg_v6i __attribute__((ext_vector_type(6)));
typedef int tg_v8i __attribute__((ext_vector_type(8)));
typedef _Float16 tg_half16 __attribute__((ext_vector_type(16)));
typedef _Float16 tg_half32 __attribute__((ext_vector_type(32)));
constexpr int TFRAG6_DWORDS = 384;        // one k_tpack6 fragment: [lane 64][16 B] + [lane 64][8 B]

// W6 = 1 (round 4, NA = 2 and NT_N = 1 only: the single-clip / small-batch tilings of DSVC_PREC_F16_X3T): of the three products
// W_hi x_hi + W_hi x_lo + W_lo x_hi the last one runs as ONE K = 64 block-scaled 6-bit MFMA per group -- w_lo as time-dithered fp6 codes,
// x_hi converted to bf6 in registers from the four fragments the fp16 MFMAs read (tlayer.h's W6 scheme).  These kernels are bound by the
// weight stream and by in-order issue, not by the matrix pipe: the lo plane shrinks from 1 KiB to 384 B per k16 step and 12 MFMAs per
// group become 8 + 1.
// KP = 1 (round 4, NA = 2 and KS = 1: the trainer's data-gradient GEMMs): the K axis is STREAMED through LDS in phases of a.kp_cin input channels
// per plane instead of being resident -- a 64-frame tile of [hi | lo] rows of 2C = 768 channels is 192 KB.  Two buffers: the DMA of phase p + 1
// is issued right behind the barrier that publishes phase p, so it runs under phase p's MFMAs; one barrier per phase.
// FS = 2 (round 4; the vocoder's 128-channel stage: four 32-channel output tiles do not fill eight waves): the workgroup's time tile holds FS
// frame sub-tiles of 32 NT_N frames, waves [0, WAVES/FS) work on the first, the rest on the second -- same LDS tile, same weight stream.
template <int NT_N, int WAVES, int MINW, int KG, int NW, class Epi, int SCHED = 1, int KS = 1, int NA = 1, int W6 = 0, int KP = 0, int FS = 1, int SKIP = 0>
__global__ void __launch_bounds__(64 * WAVES * KS, MINW) __attribute__((amdgpu_waves_per_eu(MINW, MINW)))
tgemm_kernel(const TGemmArgs a, const typename Epi::Args ea) {
    static_assert(FS == 1 || (KS == 1 && !KP && WAVES % FS == 0), "frame sub-tiles: the resident-tile flow only");
    constexpr int WM = WAVES / FS;                       // waves (= output tiles of a pass) per frame sub-tile
    static_assert(NA == 1 || NW == 2, "split activations are combined with hi + lo weight planes");
    static_assert(!W6 || (NA == 2 && NT_N == 1 && KG == 4), "W6: the small split-activation tilings, 64 input channels per group");
    static_assert(!KP || (KS == 1 && !W6), "KP: streamed K phases on the plain tilings");
    constexpr int TN = 32 * NT_N;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = KS > 1 ? wave_all % WAVES : wave_all;      // which output tile of the pass
    const int ks = KS > 1 ? wave_all / WAVES : 0;               // which slice of the K loop
    const int fb = FS > 1 ? (wave_all / WM) * TN : 0;         // this wave's frame sub-tile inside the workgroup's time tile
    const int row0 = blockIdx.x * TN * FS + fb;
    if constexpr (SKIP) {                                 // (a ragged call's own instantiation: TGemmArgs::skip_hi)
        if (a.skip_rowclip()[row0] < 0) return;
    }
    constexpr bool STAMPS = NT_N == 1;                    // the phase stamps exist only in the small-batch kernels: in the 128-frame
                                                          // tiling their few SGPRs/VGPRs tip the register allocation into spills
    auto stamp = [&](int i) {
#ifdef DSVC_PROFILING
        if constexpr (STAMPS) if (a.stamps && lane == 0)
            a.stamps[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (WAVES * KS) + wave_all) * 16 + i] = __builtin_amdgcn_s_memrealtime();
#else
        (void)i;
#endif
    };
    stamp(0);
    const int halo = (a.taps >> 1) * a.dil;
    if constexpr (NT_N == 1 && sizeof(TGemmArgs) + sizeof(typename Epi::Args) > 128) {
        // the 32-frame tilings (the single clip and small batches; round 6, third session): the scalar cache fetches a 64-byte line of the kernel arguments when it is first
        // read, and the compiler reads a by-value struct lazily -- the first read of line 2 or 3 (the epilogue's later fields) sits in the middle
        // of the front path, a serialized miss of ~0.14 us in front of the weight stream.  One word of each further line is read here, in the
        // same batch as the first argument loads: -0.8 % on the headline, on three and on six clips (same-box A/Bs, profiles/r6ag_kernarg_touch_lib_ab.txt).  Reading ALL
        // arguments in one batch at entry (values laundered through an empty asm) was measured too: no gain over not touching at all.
        typedef const int __attribute__((address_space(4))) kint;
        kint* ka = (kint*)__builtin_amdgcn_kernarg_segment_ptr();
        int touch = ka[32];
        if constexpr (sizeof(TGemmArgs) + sizeof(typename Epi::Args) > 192) touch |= ka[48];
        asm volatile("; kernel-argument lines 2+ are in the scalar cache (%0)" :: "s"(touch));
    }
    const int rows_lds = TN * FS + 2 * halo;
    const int row_halfs = a.cin * NA;                    // NA = 2: [hi plane | lo plane]
    const int chunks = row_halfs >> 3;                   // 16-B chunks per row
    const int row_bytes = row_halfs * 2;

    // ---- stage the time tile: HBM/L2 -> LDS by DMA, swizzled on the source side ----
    if constexpr (!KP) {
        const int total = rows_lds * chunks;             // 16-B slots
        const int dq = (WAVES * KS * 64) / chunks, dr = (WAVES * KS * 64) - dq * chunks;
        int slot = wave_all * 64 + lane;
        int r = slot / chunks, c = slot - r * chunks;
        const _Float16* xrow0 = a.x + (long long)(row0 - fb - halo) * row_halfs;
        for (int it = wave_all; it * 64 < total && !TG_DBG(a, 4); it += WAVES * KS) {
            const int rc = r < rows_lds ? r : rows_lds - 1;            // lanes past the tile re-read its last row
            const _Float16* src = xrow0 + (long long)rc * row_halfs + ((c ^ (rc & a.swz)) << 3);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(smem + it * 1024), 16, 0, 0);
            c += dr; r += dq;
            if (c >= chunks) { c -= chunks; r += 1; }
        }
    }

    stamp(12);
    int variant = 0;
    if (a.n_variants > 1 && a.step_ptr) {
        const int st = *a.step_ptr - a.step_off;
        variant = st % a.n_variants;
        if (variant < 0) variant += a.n_variants;
    }
    if constexpr (STAMPS) if (TG_STAMPS(a)) { asm volatile("" :: "s"(variant)); stamp(13); }
    constexpr int GROUP_HALFS = KG * NW * TFRAG_HALFS;   // one ring refill = KG k16-steps of one output tile
    const _Float16* wbase = a.w + (long long)variant * a.variant_halfs + lane * 8;
    const int gpt = (a.cin >> 4) / KG;                   // groups per tap
    const int G = a.taps * gpt;                          // groups per output tile
    const int passes = (a.m_tiles + WM - 1) / WM;
    const long long tile_halfs = (long long)G * GROUP_HALFS;

    // two waves share a SIMD (waves w and w + 4): give one of them priority so the pair drifts apart and one wave's
    // epilogue / memory waits sit under the other's MFMAs instead of both stalling together
    if (WAVES > 4 && wave >= 4 && !TG_DBG(a, 16)) __builtin_amdgcn_s_setprio(1);     // waves w and w+4 share SIMD (w & 3)

    // (W6: ring[u][1] holds nothing -- the lo plane of the group is the 24-byte code string `c6` -- and p6 is its fragment)
    auto load_group = [&](half8 (&ring)[KG][NW], const _Float16* p) {
#pragma unroll
        for (int u = 0; u < KG; ++u)
#pragma unroll
            for (int q = 0; q < (W6 ? 1 : NW); ++q)
                ring[u][q] = *reinterpret_cast<const half8*>(p + (u * NW + q) * TFRAG_HALFS);
    };
    auto load_codes = [&](tg_v6i& c6, const unsigned* p6) {
        if constexpr (W6) {
            const int4 lo = *reinterpret_cast<const int4*>(p6 + lane * 4);
            const int2 hi = *reinterpret_cast<const int2*>(p6 + 256 + lane * 2);
            c6 = tg_v6i{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y};
        }
    };
    typedef const half8 __attribute__((address_space(3))) * lds_frag_ptr;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // what compute_group reads (KP: the current phase buffer, its narrower rows and its groups per tap; otherwise the resident tile)
    unsigned cg_lds = lds0, cg_row_bytes = (unsigned)row_bytes, cg_lo_off = (unsigned)a.cin * 2u;
    int cg_gpt = gpt;
    if constexpr (KP) { cg_row_bytes = (unsigned)(a.kp_cin * NA * 2); cg_lo_off = (unsigned)a.kp_cin * 2u; cg_gpt = (a.kp_cin >> 4) / KG; }
    const unsigned nt_stride = 32u * cg_row_bytes;
    auto compute_group = [&](const half8 (&ring)[KG][NW], const tg_v6i& c6, f32x16 (&acc)[NT_N], int g) {
        const int tap = g / cg_gpt, kb = (g - tap * cg_gpt) * KG;
        if constexpr (W6) {
            // one N-tile, four k-steps: all eight fragments (x_hi and x_lo of the four steps) are read up front -- the conversion needs the
            // four x_hi fragments together -- then 4 x (W_hi x_hi, W_hi x_lo), the conversion, and W_lo6 x_hi6
            const int rr6 = halo + (tap - (a.taps >> 1)) * a.dil + (lane & 31);
            const unsigned xs6 = (unsigned)(((rr6 & a.swz) ^ (lane >> 5)) << 4) ^ ((unsigned)kb << 5);
            const unsigned b0 = lds0 + (unsigned)rr6 * (unsigned)row_bytes, lo_off6 = (unsigned)a.cin * 2u;
            half8 xh[4], xl[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                xh[kk] = *(lds_frag_ptr)(size_t)(b0 + (((unsigned)kk << 5) ^ xs6));
                xl[kk] = *(lds_frag_ptr)(size_t)(b0 + (((unsigned)kk << 5) ^ xs6) + lo_off6);
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[kk][0], xh[kk], acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[kk][0], xl[kk], acc[0], 0, 0, 0);
            }
            const tg_half16 v01 = __builtin_shufflevector(xh[0], xh[1], 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
            const tg_half16 v23 = __builtin_shufflevector(xh[2], xh[3], 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
            const tg_half32 v = __builtin_shufflevector(v01, v23, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24,
                                                        25, 26, 27, 28, 29, 30, 31);
            const tg_v6i q = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(v, a.x6_scale);
            const tg_v8i a6 = __builtin_shufflevector(c6, c6, 0, 1, 2, 3, 4, 5, -1, -1), b6 = __builtin_shufflevector(q, q, 0, 1, 2, 3, 4, 5, -1, -1);
            acc[0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a6, b6, acc[0], 2 /* A: fp6 E2M3 */, 3 /* B: bf6 E3M2 */, 0, a.sc6 & 255, 0,
                                                                     (a.sc6 >> 8) & 255);
            return;
        }
        const int rr = halo + (tap - (a.taps >> 1)) * a.dil + fb + (lane & 31);      // LDS row of this lane's frame, N-tile 0
        // chunk of k16-step k, half h, row r lives at slot (2k + h) ^ (r & swz), i.e. at byte offset
        // (k << 5) ^ xs with xs = ((r & swz) ^ h) << 4.  The group's first step kb is a multiple of KG and kk < KG, so
        // ((kb + kk) << 5) ^ xs = ((kb << 5) ^ xs) ^ (kk << 5): one per-group VALU, then an immediate XOR per step.
        const unsigned xs = (unsigned)(((rr & a.swz) ^ (lane >> 5)) << 4) ^ ((unsigned)kb << 5);
        unsigned base[NT_N];
        base[0] = cg_lds + (unsigned)rr * cg_row_bytes;
#pragma unroll
        for (int nt = 1; nt < NT_N; ++nt) base[nt] = base[nt - 1] + nt_stride;
        // keep the per-group bases materialised: without this LLVM re-derives every address from scratch (4-5 VALU per ds_read)
#pragma unroll
        for (int nt = 0; nt < NT_N; ++nt) asm volatile("" : "+v"(base[nt]));
        // B fragments: BQ-deep software pipeline -- step kk computes from bq[kk % BQ] while steps kk+1 .. kk+BQ-1 are in flight
        constexpr int BQ = 2;                                          // (3- and 4-deep pipelines measured 3-4 % slower)
        half8 bq[BQ][NT_N], bl[BQ][NA == 2 ? NT_N : 1];
        const unsigned lo_off = cg_lo_off;                             // the lo plane of a row starts cin halfs in: the swizzle only touches the low
                                                                       // four chunk bits and cin / 8 is a multiple of 16, so lo = hi address + cin * 2
#pragma unroll
        for (int d = 0; d < BQ - 1; ++d) {
            if (d < KG) {
                const unsigned off = ((unsigned)d << 5) ^ xs;
#pragma unroll
                for (int nt = 0; nt < NT_N; ++nt) {
                    bq[d][nt] = *(lds_frag_ptr)(size_t)(base[nt] + off);
                    if constexpr (NA == 2) bl[d][nt] = *(lds_frag_ptr)(size_t)(base[nt] + off + lo_off);
                }
            }
        }
#pragma unroll
        for (int kk = 0; kk < KG; ++kk) {
            if (kk + BQ - 1 < KG) {
                const unsigned off = ((unsigned)(kk + BQ - 1) << 5) ^ xs;
#pragma unroll
                for (int nt = 0; nt < NT_N; ++nt) {
                    bq[(kk + BQ - 1) % BQ][nt] = *(lds_frag_ptr)(size_t)(base[nt] + off);
                    if constexpr (NA == 2) bl[(kk + BQ - 1) % BQ][nt] = *(lds_frag_ptr)(size_t)(base[nt] + off + lo_off);
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT_N; ++nt) {
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[kk][0], bq[kk % BQ][nt], acc[nt], 0, 0, 0);
                if constexpr (NW == 2)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[kk][1], bq[kk % BQ][nt], acc[nt], 0, 0, 0);
                if constexpr (NA == 2)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[kk][0], bl[kk % BQ][nt], acc[nt], 0, 0, 0);
            }
        }
        // pin the software pipeline: left alone, the scheduler sinks every ds_read to just above its MFMA (register
        // pressure heuristic) and the wave then eats the full LDS latency once per MFMA
#pragma unroll
        for (int d = 0; d < BQ - 1; ++d)
            if (d < KG) __builtin_amdgcn_sched_group_barrier(0x100, NT_N * NA, 0);    // B fragments of the first BQ-1 steps
#pragma unroll
        for (int kk = 0; kk < KG; ++kk) {
            if constexpr (SCHED == 0) {                   // block form: all reads of step kk+1, then all MFMAs of step kk
                if (kk + BQ - 1 < KG) __builtin_amdgcn_sched_group_barrier(0x100, NT_N * NA, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NT_N * (NW + NA - 1), 0);
            } else {                                       // 1:1 interleave (default, measured 1-3 % faster): one read of step kk+1
                                                           // behind each MFMA of step kk
#pragma unroll
                for (int nt = 0; nt < NT_N; ++nt) {
                    __builtin_amdgcn_sched_group_barrier(0x008, NW + NA - 1, 0);
                    if (kk + BQ - 1 < KG) __builtin_amdgcn_sched_group_barrier(0x100, NA, 0);
                }
            }
        }
    };

    Epi epi;
    f32x16 acc[NT_N];
    half8 ringA[KG][NW], ringB[KG][NW];
    tg_v6i codeA = {}, codeB = {};
    auto load_wg = [&](half8 (&ring)[KG][NW], tg_v6i& c6, int tile, int grp) {      // weight group `grp` of output tile `tile`: fragments (+ W6: codes)
        load_group(ring, wbase + (long long)tile * tile_halfs + (long long)grp * GROUP_HALFS);
        if constexpr (W6) load_codes(c6, a.w6 + ((size_t)tile * (size_t)G + (size_t)grp) * TFRAG6_DWORDS);
    };
    // output-channel passes are visited in a per-workgroup rotated order: all workgroups stream the SAME weights, and
    // without the rotation they all miss L2 on the same fragment at the same moment (the whole chip then advances at
    // first-touch latency); rotated, a tile's first toucher warms it for the other two thirds
    const int rot = gridDim.y == 1 && !TG_DBG(a, 64) ? (int)(blockIdx.x % (unsigned)passes) : 0;
    auto tile_of = [&](int pi) { const int p = pi + rot; return (p < passes ? p : p - passes) * WM + (FS > 1 ? wave % WM : wave); };
    auto next_active = [&](int pi) {                      // next position of this workgroup's sequence where this wave has a tile
        for (; pi < passes; pi += gridDim.y)
            if (tile_of(pi) < a.m_tiles) return pi;
        return -1;
    };
    // ... and the K loop of a tile starts at a per-workgroup group offset (and wraps): the workgroups of different frame
    // tiles stream the SAME fragments, and started together they all wait on the same few KB at any moment -- at B = 1
    // the unique bytes in flight are then 24 tiles x 16 KB and the 1.8 MB weight variant (HBM-cold under time-dithering)
    // arrives at latency x 0.4 MB instead of at bandwidth.  Staggered starts multiply the unique bytes in flight by G.
    // (small tilings only: in the 128-frame tiling the stream is bandwidth-, not latency-bound -- measured neutral -- and the
    //  extra index arithmetic tips its 256-register allocation into spills)
    constexpr bool STAGGER = NT_N < 4;
    const unsigned tiles_pc = (a.clip_rows > 0 && a.clip_rows % TN == 0) ? (unsigned)(a.clip_rows / TN) : 0u;
    const int rk = (!STAGGER || TG_DBG(a, 1024)) ? 0 : (int)((tiles_pc ? blockIdx.x % tiles_pc : blockIdx.x) % (unsigned)G);
    auto gmap = [&](int g) { if constexpr (!STAGGER) return g; const int x = g + rk; return x >= G ? x - G : x; };
    auto wgrp = [&](int g) { return TG_DBG(a, 4096) ? 0 : gmap(g); };    // dbg 4096: the ring re-reads group 0 (L1-hot weight stream)
    if constexpr (KP) {
        // ---- streamed-K flow ----
        const int kpc = a.kp_cin, n_ph = a.cin / kpc;
        const int chunks_l = (kpc * NA) >> 3, cpp = kpc >> 3;                 // 16-B chunks per LDS row / per plane of it
        const unsigned buf_bytes = (((unsigned)rows_lds * cg_row_bytes) + 1023u) & ~1023u;
        const int Gp = a.taps * cg_gpt;                                      // groups of one phase
        const int total = rows_lds * chunks_l;
        const _Float16* xrow0 = a.x + (long long)(row0 - halo) * row_halfs;
        auto dma = [&](int p, int b) {      // phase p -> buffer b: LDS slot (row r, chunk c') holds chunk c = c' ^ (r & swz) of [hi kpc | lo kpc]
            const int dq = (WAVES * 64) / chunks_l, dr = (WAVES * 64) - dq * chunks_l;
            int slot = wave_all * 64 + lane;
            int r = slot / chunks_l, c = slot - r * chunks_l;
            for (int it = wave_all; it * 64 < total; it += WAVES) {
                const int rc = r < rows_lds ? r : rows_lds - 1;
                const int cs = c ^ (rc & a.swz);
                const int plane = cs >= cpp ? 1 : 0;
                const _Float16* src = xrow0 + (long long)rc * row_halfs + plane * a.cin + p * kpc + ((cs - plane * cpp) << 3);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(smem + (size_t)b * buf_bytes + it * 1024), 16, 0, 0);
                c += dr; r += dq;
                if (c >= chunks_l) { c -= chunks_l; r += 1; }
            }
        };
        auto gidx = [&](int p, int gl) { const int tap = gl / cg_gpt; return tap * gpt + p * cg_gpt + (gl - tap * cg_gpt); };   // group of the weight stream
        Epi epi;
        f32x16 acc[NT_N];
        half8 ringA[KG][NW], ringB[KG][NW];
        tg_v6i c6 = {};
        const int passes_ = (a.m_tiles + WAVES - 1) / WAVES;
        int ph = 0;                                                          // running phase count: buffer = ph & 1
        dma(0, 0);
        for (int pi = blockIdx.y; pi < passes_; pi += gridDim.y) {
            const int mt = pi * WAVES + wave;
            const bool active = mt < a.m_tiles;
            const long long tbase = (long long)(active ? mt : 0) * G * GROUP_HALFS;
            if (active) {
                load_group(ringA, wbase + tbase + (long long)gidx(0, 0) * GROUP_HALFS);
                epi.init(ea, mt, row0, lane, acc);
            }
            for (int p = 0; p < n_ph; ++p, ++ph) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's pieces of phase p have landed ...
                __builtin_amdgcn_s_barrier();                               // ... everyone's have, and nobody still reads the other buffer
                asm volatile("" ::: "memory");
                const bool more = p + 1 < n_ph || pi + (int)gridDim.y < passes_;
                if (more) dma(p + 1 < n_ph ? p + 1 : 0, (ph + 1) & 1);
                cg_lds = lds0 + (unsigned)(ph & 1) * buf_bytes;
                if (active) {
                    int gl = 0;
                    for (; gl + 1 < Gp; gl += 2) {
                        load_group(ringB, wbase + tbase + (long long)gidx(p, gl + 1) * GROUP_HALFS);
                        __builtin_amdgcn_sched_barrier(0);
                        compute_group(ringA, c6, acc, gl);
                        if (gl + 2 < Gp) load_group(ringA, wbase + tbase + (long long)gidx(p, gl + 2) * GROUP_HALFS);
                        else if (p + 1 < n_ph) load_group(ringA, wbase + tbase + (long long)gidx(p + 1, 0) * GROUP_HALFS);
                        __builtin_amdgcn_sched_barrier(0);
                        compute_group(ringB, c6, acc, gl + 1);
                    }
                    if (gl < Gp) {
                        compute_group(ringA, c6, acc, gl);
                        if (p + 1 < n_ph) load_group(ringA, wbase + tbase + (long long)gidx(p + 1, 0) * GROUP_HALFS);
                    }
                }
            }
            if (active) epi.finish(ea, mt, row0, lane, acc);
        }
        return;
    }
    if constexpr (KS > 1) {
        // ---- split-K flow: one tile per wave triple, gridDim.y == passes (host-checked), reduction through LDS ----
        const int mt = blockIdx.y * WAVES + wave;
        const bool active = mt < a.m_tiles;
        const int g0 = ks * G / KS, n = (ks + 1) * G / KS - g0;
        if (active && n > 0) load_wg(ringA, codeA, mt, gmap(g0));
        // the second group is put in flight before the barrier too: a ring refill issued inside the loop is waited for at full
        // L2/HBM latency, there is no other work in a 3-group slice to hide it behind
        if (active && n > 1) load_wg(ringB, codeB, mt, gmap(g0 + 1));
        stamp(14);
        if (active && ks == 0 && !TG_DBG(a, 1)) {
            epi.init(ea, mt, row0, lane, acc);
        } else {
#pragma unroll
            for (int nt = 0; nt < NT_N; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
        }
        stamp(1);
        // Only the DMA'd tile has to have landed before the barrier; the weight rings issued behind it may stay in flight across
        // it (vmcnt retires in order: allowing the 8 / 16 most recent loads to be outstanding still covers every DMA piece; the
        // accumulator-init loads, issued last, only make the allowance more conservative).  __syncthreads() would drain
        // everything (its fence waits vmcnt(0)), so the bare barrier is used: the tile was written by DMA, not by ds_write, and
        // each wave's own vmcnt wait + the barrier is exactly what makes it visible (cdna_hip_programming.md 5.7 item 1).
        {
            const int later = (active && n > 0 ? 1 : 0) + (active && n > 1 ? 1 : 0);      // rings of KG*NW = 8 loads each ...
            static_assert(KG * NW == 8, "the counted vmcnt immediates below assume 8 loads per ring");
            if constexpr (W6) {       // ... W6: KG hi fragments + the code string's two loads = 6 (an allowance of 8 per ring would let the barrier
                                      // pass with up to four of this wave's DMA pieces still in flight -- round 4 shipped that for a while; found by
                                      // reading, never by a test: the pieces are the oldest loads and had always landed)
                if (later == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else if (later == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                if (later == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else if (later == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        stamp(2);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stamp(3);
        if (active && !TG_DBG(a, 8)) {
            int i = 0;
            for (; i + 1 < n; i += 2) {
                if (i > 0) load_wg(ringB, codeB, mt, gmap(g0 + i + 1));
                __builtin_amdgcn_sched_barrier(0);
                compute_group(ringA, codeA, acc, gmap(g0 + i));
                if (i + 2 < n) load_wg(ringA, codeA, mt, gmap(g0 + i + 2));
                __builtin_amdgcn_sched_barrier(0);
                compute_group(ringB, codeB, acc, gmap(g0 + i + 1));
            }
            if (i < n) compute_group(ringA, codeA, acc, gmap(g0 + i));
        }
        stamp(9);
        // partial sums of slices 1 .. KS-1 -> LDS [slice][tile wave][N-tile][quad][lane] (16 B per lane: conflict-free)
        f32x4* red = reinterpret_cast<f32x4*>(smem + (((size_t)rows_lds * row_bytes + 1023) & ~(size_t)1023));
        if (ks > 0) {
#pragma unroll
            for (int nt = 0; nt < NT_N; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    red[((((ks - 1) * WAVES + wave) * NT_N + nt) * 4 + q) * 64 + lane] =
                        f32x4{acc[nt][4 * q], acc[nt][4 * q + 1], acc[nt][4 * q + 2], acc[nt][4 * q + 3]};
        }
        __syncthreads();
        if (ks == 0 && active) {
#pragma unroll
            for (int s2 = 1; s2 < KS; ++s2)
#pragma unroll
                for (int nt = 0; nt < NT_N; ++nt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = red[((((s2 - 1) * WAVES + wave) * NT_N + nt) * 4 + q) * 64 + lane];
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[nt][4 * q + i] += v[i];
                    }
            if (!TG_DBG(a, 2)) epi.finish(ea, mt, row0, lane, acc);
            if constexpr (STAMPS) if (TG_STAMPS(a)) { stamp(10); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(11); }
        } else if constexpr (STAMPS) { if (TG_STAMPS(a)) { stamp(10); stamp(11); } }
        return;
    }
    int pi = next_active(blockIdx.y);
    int mt = pi >= 0 ? tile_of(pi) : 0;
    if (pi >= 0) {
        load_wg(ringA, codeA, TG_DBG(a, 32) ? 0 : mt, gmap(0));
        stamp(14);
        if TG_DBG(a, 1) {
#pragma unroll
            for (int nt = 0; nt < NT_N; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
        } else {
            epi.init(ea, mt, row0, lane, acc);
        }
    }
    stamp(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the DMA'd tile (and the first operands) have landed
    stamp(2);
    __syncthreads();
    stamp(3);

    // The accumulator-init loads of a wave's NEXT tile are issued in the middle of its current tile's main loop: they are a
    // burst from HBM (the conditioner projection / the residual stream), vmcnt is in-order, so whenever they are issued the
    // wave's next weight-ring wait stalls until they have landed.  The two waves of a SIMD (w and w+4) issue theirs at
    // DIFFERENT points of the pass (first vs middle group pair), so one of them always has MFMAs to issue while the other
    // sits behind its burst, and the chip-wide HBM demand is spread instead of arriving from every workgroup at once.
    const int g_issue = (WAVES > 4 && wave >= WAVES / 2) ? ((G / 2) & ~1) : 0;
    while (pi >= 0) {
        const int pn = next_active(pi + gridDim.y);
        const int mt_n = pn >= 0 ? tile_of(pn) : 0;
        f32x16 nxt[NT_N];
        bool nxt_issued = false;
        auto issue_next_init = [&]() {
            if TG_DBG(a, 1) {
#pragma unroll
                for (int nt = 0; nt < NT_N; ++nt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) nxt[nt][i] = 0.f;
            } else {
                epi.init(ea, mt_n, row0, lane, nxt);
            }
        };
        int g = (TG_DBG(a, 8) || (TG_DBG(a, 256) && wave >= WAVES / 2)) ? G : 0;      // dbg 256: half the waves skip their MFMAs
        for (; g + 1 < G; g += 2) {     // straight-line body (no branch around the prefetches): IR-level sinking cannot move one below its group
            if constexpr (STAMPS) if TG_DBG(a, 2048) {                         // profiling: no weight stream inside the loop
                compute_group(ringA, codeA, acc, gmap(g));
                compute_group(ringA, codeA, acc, gmap(g + 1));
                if constexpr (STAMPS) if (TG_STAMPS(a) && g < 8) stamp(4 + (g >> 1));
                continue;
            }
            load_wg(ringB, codeB, TG_DBG(a, 32) ? 0 : mt, wgrp(g + 1));        // ringB <- group g+1, under group g's MFMAs
            if (g == g_issue && pn >= 0 && !TG_DBG(a, 128)) { issue_next_init(); nxt_issued = true; }
            __builtin_amdgcn_sched_barrier(0);
            compute_group(ringA, codeA, acc, gmap(g));
            const int gn = g + 2 < G ? g + 2 : G - 1;
            load_wg(ringA, codeA, TG_DBG(a, 32) ? 0 : mt, wgrp(gn));           // ringA <- group g+2, under group g+1's MFMAs
            __builtin_amdgcn_sched_barrier(0);
            compute_group(ringB, codeB, acc, gmap(g + 1));
            if constexpr (STAMPS) if (TG_STAMPS(a) && g < 8) stamp(4 + (g >> 1));
        }
        if (g < G) compute_group(ringA, codeA, acc, gmap(g));                  // odd group count: the tail group
        stamp(9);
        // the next tile's weight stream starts before this tile's epilogue, so its latency sits under the epilogue
        if (pn >= 0) {
            load_wg(ringA, codeA, TG_DBG(a, 32) ? 0 : mt_n, gmap(0));
            if (!nxt_issued) issue_next_init();                                // short K loops (or dbg 128): issue here instead
        }
        if (!TG_DBG(a, 2)) epi.finish(ea, mt, row0, lane, acc);
        else asm volatile("" :: "v"(acc[0][0]), "v"(acc[NT_N - 1][15]));
        if constexpr (STAMPS) if (TG_STAMPS(a)) {
            stamp(10);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamp(11);
        }
        if (pn < 0) break;
#pragma unroll
        for (int nt = 0; nt < NT_N; ++nt) acc[nt] = nxt[nt];
        pi = pn; mt = mt_n;
    }
}

template <int NT_N>
inline size_t tgemm_smem(int taps, int dil, int cin) {                 // cin = halfs per activation row (both planes when they are split)
    const size_t bytes = (size_t)(32 * NT_N + 2 * (taps / 2) * dil) * cin * 2;
    return (bytes + 1023) & ~(size_t)1023;               // whole 1 KiB DMA pieces
}

inline int tgemm_swizzle_mask(int cin) {
    const int chunks = cin / 8;
    int m = 15;
    while (m > 0 && (chunks % (m + 1)) != 0) m >>= 1;
    return m;
}

// profiling aid (env DSVC_TG_STAMPS=<path prefix>): the first `slots` tgemm launches of the process record 16 s_memrealtime
// stamps per wave (see stamp() in the kernel); when the last slot is used the log is written to <prefix>.bin (u64) and
// <prefix>.meta (one line per launch: kernel, grid x, grid y, waves).  tools/stamps_report.py reads it.
struct TStampLog {
    unsigned long long* dev = nullptr;
    int slots = 172, used = 0;
    size_t per_slot = (size_t)2048 * 16;              // waves per launch (upper bound) x stamps
    std::string meta;
    bool done = false;
};
inline TStampLog& tstamp_log() { static TStampLog l; return l; }
inline void tstamp_dump(const char* prefix) {
    TStampLog& L = tstamp_log();
    if (L.done || !L.dev) return;
    L.done = true;
    if (hipDeviceSynchronize() != hipSuccess) return;
    std::vector<unsigned long long> host((size_t)L.used * L.per_slot);
    if (hipMemcpy(host.data(), L.dev, host.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
    std::string p(prefix);
    if (FILE* f = fopen((p + ".bin").c_str(), "wb")) { fwrite(host.data(), 8, host.size(), f); fclose(f); }
    if (FILE* f = fopen((p + ".meta").c_str(), "w")) { fputs(L.meta.c_str(), f); fclose(f); }
}

// n_rows must be a multiple of 32*NT_N; m_split = number of blockIdx.y slices the output-channel passes are dealt over
// epilogues of the sampler's launches opt in to the padded-tile skip of ragged batches (`static constexpr bool RAGGED_SKIP = true`): only they get
// the second instantiation
template <class E, class = void> struct epi_ragged_skip { static constexpr bool value = false; };
template <class E> struct epi_ragged_skip<E, std::void_t<decltype(E::RAGGED_SKIP)>> { static constexpr bool value = E::RAGGED_SKIP; };

template <int NT_N, int WAVES, int MINW, int KG, int NW, class Epi, int SCHED = 1, int KS = 1, int NA = 1, int W6 = 0, int KP = 0, int FS = 1, int SKIP = 0>
inline int tgemm_launch(TGemmArgs a, const typename Epi::Args& ea, int n_rows, int m_split, hipStream_t stream) {
    if constexpr (!SKIP && KS == 1 && !KP && FS == 1 && epi_ragged_skip<Epi>::value) {
        if ((a.skip_lo | a.skip_hi) != 0) return tgemm_launch<NT_N, WAVES, MINW, KG, NW, Epi, SCHED, KS, NA, W6, KP, FS, 1>(a, ea, n_rows, m_split, stream);
    }
    if (W6 && (!a.w6 || a.n_variants != 1 || a.cin % 64 != 0)) return fail(DSVC_EINVAL, "tgemm: the W6 kernels need the code plane of the variant to use");
    if (a.cin % (16 * KG) != 0) return fail(DSVC_EINVAL, "tgemm: cin %d not a multiple of %d", a.cin, 16 * KG);
    if (n_rows % (32 * NT_N * FS) != 0) return fail(DSVC_EINVAL, "tgemm: %d rows not a multiple of the %d-frame tile", n_rows, 32 * NT_N * FS);
    if (a.w_planes != NW) return fail(DSVC_EINVAL, "tgemm: weights packed with %d plane(s), kernel streams %d", a.w_planes, NW);
    a.swz = tgemm_swizzle_mask((KP ? a.kp_cin : a.cin) * NA);
    if (NA == 2 && (a.cin / 8) % 16 != 0) return fail(DSVC_EINVAL, "tgemm: split activations need cin %% 128 == 0 (got %d)", a.cin);
    if (KP && (NA != 2 || a.kp_cin <= 0 || a.kp_cin % 128 != 0 || a.cin % a.kp_cin != 0))
        return fail(DSVC_EINVAL, "tgemm: %d-channel K phases of %d split channels", a.kp_cin, a.cin);
#ifdef DSVC_PROFILING
    const char* dbg_s = getenv("DSVC_TG_DEBUG");           // profiling ablations only; results are WRONG when set
    a.dbg = dbg_s ? atoi(dbg_s) : 0;
#endif
    auto kern = tgemm_kernel<NT_N, WAVES, MINW, KG, NW, Epi, SCHED, KS, NA, W6, KP, FS, SKIP>;
    const size_t smem = KP ? 2 * tgemm_smem<NT_N>(a.taps, a.dil, a.kp_cin * NA)
                           : tgemm_smem<NT_N * FS>(a.taps, a.dil, a.cin * NA) + (size_t)(KS - 1) * WAVES * NT_N * 4096;
    if (smem > 160 * 1024) return fail(DSVC_EINVAL, "tgemm: %zu B of LDS requested", smem);
    static thread_local size_t smem_set = 0;
    if (smem > 64 * 1024 && smem > smem_set) {
        DSVC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        smem_set = smem;
    }
    const int passes = ceil_div(a.m_tiles, WAVES / FS);
    if (m_split < 1) m_split = 1;
    if (m_split > passes) m_split = passes;
    if (KS > 1 && m_split != passes) return fail(DSVC_EINVAL, "tgemm: the split-K tiling needs one output tile per wave (m_split %d, passes %d)", m_split, passes);
#ifdef DSVC_PROFILING
    static const char* stamp_path = getenv("DSVC_TG_STAMPS");
    if (stamp_path) {
        TStampLog& L = tstamp_log();
        const size_t waves = (size_t)(n_rows / (32 * NT_N)) * m_split * WAVES * KS;
        if (!L.dev && !L.done) {
            DSVC_HIP(hipMalloc(&L.dev, (size_t)L.slots * L.per_slot * 8));
            DSVC_HIP(hipMemset(L.dev, 0, (size_t)L.slots * L.per_slot * 8));
        }
        if (!L.done && L.used < L.slots && waves * 16 <= L.per_slot) {
            a.stamps = L.dev + (size_t)L.used * L.per_slot;
            char line[512];
            snprintf(line, sizeof line, "%s|%d|%d|%d\n", __PRETTY_FUNCTION__, n_rows / (32 * NT_N), m_split, WAVES * KS);
            L.meta += line;
            L.used++;
        }
    }
#endif
    hipLaunchKernelGGL(kern, dim3(n_rows / (32 * NT_N * FS), m_split), dim3(64 * WAVES * KS), smem, stream, a, ea);
    DSVC_HIP(hipGetLastError());
#ifdef DSVC_PROFILING
    if (stamp_path && tstamp_log().used == tstamp_log().slots) tstamp_dump(stamp_path);
#endif
    return DSVC_OK;
}

// ---------------------------------------------------------------------------------------------
// Packing (on the device: a 64-variant dithered DiffNet is 3 GB of fragments).
//   src [O][I][taps] fp32 in the checkpoint's Conv1d layout; rowmap[m_tiles*32] gives the source output channel of
//   every packed row (or -1 = zero row) -- the caller folds the tile-row permutation (trow_to_ch16 / trow_to_ch8)
//   into it; rowscale (optional) multiplies a packed row (the gate kernel folds -log2(e) / -2 log2(e) into its rows).  Layout [variant][m_tile][tap][k16][plane][lane][8]; lane l of a fragment holds packed row (l & 31),
//   k = k16*16 + 8*(l >> 5) + e.  plane 0 = fp16(w), plane 1 = fp16(w - plane0).  fold > 0 packs K = 2*fold input
//   channels whose upper half repeats the lower one: the matching activation buffer holds [x_hi | x_lo] planes.
//   Variant v of n rounds w to fp16 DOWN or UP so that the mean over the variants is w +- ulp/(2n): round up iff
//   frac(w) > ((v' + 0.5)/n + phase(element)) mod 1, v' = bit-reversed v (consecutive diffusion steps use far-apart
//   thresholds), phase = a per-element hash.  n == 1 is plain round-to-nearest.
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline uint32_t tg_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__device__ inline uint16_t tg_half_bits(_Float16 h) { return __builtin_bit_cast(uint16_t, h); }
__device__ inline _Float16 tg_bits_half(uint16_t b) { return __builtin_bit_cast(_Float16, b); }

// next representable fp16 towards +inf (up) or -inf of a finite value
__device__ inline uint16_t tg_half_step(uint16_t b, bool up) {
    const bool neg = (b & 0x8000) != 0;
    const uint16_t mag = b & 0x7fff;
    if (mag == 0) return up ? (uint16_t)0x0001 : (uint16_t)0x8001;
    if (neg == up) return (uint16_t)((neg ? 0x8000 : 0) | (mag - 1));
    return (uint16_t)((neg ? 0x8000 : 0) | (mag + 1));
}

__device__ inline _Float16 tg_round_dither(float w, float thresh) {
    const _Float16 n = (_Float16)w;                       // nearest
    const float nf = (float)n;
    if (nf == w) return n;
    _Float16 lo, hi;
    if (nf < w) { lo = n; hi = tg_bits_half(tg_half_step(tg_half_bits(n), true)); }
    else        { hi = n; lo = tg_bits_half(tg_half_step(tg_half_bits(n), false)); }
    const float lf = (float)lo, hf = (float)hi;
    const float frac = (w - lf) / (hf - lf);
    return frac > thresh ? hi : lo;
}

static __global__ void k_tpack(const float* __restrict__ src, const int* __restrict__ rowmap, const float* __restrict__ rowscale,
                        _Float16* __restrict__ dst, int I, int taps, int cin_pad, int fold, int m_tiles, int planes, int n_variants,
                        float scale, unsigned salt) {
    const int nk16 = cin_pad >> 4;
    const long long per_variant = (long long)m_tiles * taps * nk16 * 512;      // elements of ONE plane
    const long long total = per_variant * n_variants;
    int bits = 0;
    while ((1 << bits) < n_variants) ++bits;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(idx / per_variant);
        long long r = idx - (long long)v * per_variant;
        const int e = (int)(r & 7), l = (int)((r >> 3) & 63);
        r >>= 9;
        const int k = (int)(r % nk16); r /= nk16;
        const int tap = (int)(r % taps);
        const int mt = (int)(r / taps);
        const int row = mt * 32 + (l & 31), ci = k * 16 + 8 * (l >> 5) + e;
        const int o = rowmap[row];
        const int cs = fold > 0 ? ci % fold : ci;          // fold: input channels [fold, 2*fold) repeat [0, fold) -- the lo plane of a
        const float w = (o >= 0 && cs < I) ? src[((size_t)o * I + cs) * taps + tap] * scale * (rowscale ? rowscale[row] : 1.0f) : 0.f;   // split activation sees the same weights
        _Float16 hi;
        if (n_variants > 1) {
            int vr = 0;
            for (int b = 0; b < bits; ++b) vr |= ((v >> b) & 1) << (bits - 1 - b);
            if (vr >= n_variants) vr = v;
            const uint32_t h = tg_hash32(((uint32_t)row * 0x9E3779B1u) ^ ((uint32_t)(tap * cin_pad + ci) * 0x85EBCA77u) ^ salt);
            float th = ((float)vr + 0.5f) / (float)n_variants + (float)(h >> 8) * (1.0f / 16777216.0f);
            if (th >= 1.0f) th -= 1.0f;
            hi = tg_round_dither(w, th);
        } else {
            hi = (_Float16)w;
        }
        _Float16* f = dst + (size_t)v * per_variant * planes + ((((size_t)mt * taps + tap) * nk16 + k) * planes) * 512 + l * 8 + e;
        f[0] = hi;
        if (planes == 2) f[512] = (_Float16)(w - (float)hi);
    }
}

// ---------------------------------------------------------------------------------------------
// Round 4 (DSVC_PREC_F16_W6): the w_lo plane as 6-bit floats for v_mfma_scale_f32_32x32x64_f8f6f4.
//   The correction term w_lo * x of the exact-weight scheme (w = fp16(w) + w_lo, |w_lo| <= ulp/2) needs ~4 significant bits, not 11: it is
//   2^-12 of the product.  One K = 64 block-scaled MFMA on fp6 (E2M3) weights x bf6 (E3M2) activations replaces four fp16 MFMAs at a quarter
//   of their issue time (tools/micro/mx_probe.hip: 20 ns against 4 x 21 ns per SIMD).  Layout [variant][m_tile][tap][k64 group][1536 B]:
//   a fragment is [lane 64][16 B] followed by [lane 64][8 B] -- the 24 bytes (32 codes, little-endian bit stream) of lane l = (row l & 31,
//   half h = l >> 5) split so that both loads of a wave are contiguous.  Code m = 8 kk + e of the lane is input channel
//   64 q + 16 kk + 8 h + e: exactly the element order of the lane's four fp16 activation fragments kk = 0..3, which the kernel converts with
//   ONE v_cvt_scalef32_pk32_bf6_f16.  All codes of a layer share the power-of-two scale 2^e6 (w_lo is uniform in +-ulp/2, not heavy-tailed:
//   per-block scales measured 2.9 % against 4.6 % relative rms error of w_lo, i.e. 6e-6 against 9.5e-6 of w -- plain fp16 rounding is 2.1e-4),
//   and the rounding to the fp6 grid is time-dithered like k_tpack's (towards / away from zero by a stratified per-element threshold; the
//   n variants average to w_lo to 1/n of a grid step), so what is left of the systematic weight error is below the fp16 activation rounding
//   by two orders of magnitude (tests/studies/precision_study.py: w6f64).
// ---------------------------------------------------------------------------------------------
constexpr int TFRAG6_BYTES = 1536;

// magnitude grid of E2M3: index i (= the code without its sign bit) -> value
__host__ __device__ inline float tg_e2m3_value(int i) {
    const int e = i >> 3, m = i & 7;
    return e == 0 ? (float)m * 0.125f : (float)(8 + m) * 0.125f * (float)(1 << (e - 1));
}

// largest grid index whose value is <= a (0 <= a; saturates at 31 = 7.5)
__host__ __device__ inline int tg_e2m3_floor(float a) {
    if (a >= 7.5f) return 31;
    if (a < 2.0f) return (int)(a * 8.0f);                 // subnormals and [1, 2): step 1/8, indices 0 .. 15
    if (a < 4.0f) return 16 + (int)((a - 2.0f) * 4.0f);
    return 24 + (int)((a - 4.0f) * 2.0f);
}

static __global__ void k_tpack6(const float* __restrict__ src, const int* __restrict__ rowmap, const float* __restrict__ rowscale,
                         unsigned* __restrict__ dst, int I, int taps, int cin_pad, int m_tiles, int n_variants, float scale, float inv6,
                         unsigned salt, int whole) {         // whole: the codes are those of w itself (the g_lo correction's weight operand), not of w_lo
    const int nq = cin_pad >> 6;
    const long long per_variant = (long long)m_tiles * taps * nq * 64;         // lanes of one variant
    const long long total = per_variant * n_variants;
    int bits = 0;
    while ((1 << bits) < n_variants) ++bits;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(idx / per_variant);
        long long r = idx - (long long)v * per_variant;
        const int l = (int)(r & 63);
        r >>= 6;
        const int q = (int)(r % nq); r /= nq;
        const int tap = (int)(r % taps);
        const int mt = (int)(r / taps);
        const int row = mt * 32 + (l & 31), h = l >> 5;
        const int o = rowmap[row];
        int vr = 0;
        for (int b = 0; b < bits; ++b) vr |= ((v >> b) & 1) << (bits - 1 - b);
        if (vr >= n_variants) vr = v;
        unsigned long long acc = 0;
        int nbits = 0, word = 0;
        unsigned out[6];
        for (int m = 0; m < 32; ++m) {
            const int ci = 64 * q + 16 * (m >> 3) + 8 * h + (m & 7);
            const float w = (o >= 0 && ci < I) ? src[((size_t)o * I + ci) * taps + tap] * scale * (rowscale ? rowscale[row] : 1.0f) : 0.f;
            const float wl = whole ? w : w - (float)(_Float16)w;
            const float u = wl * inv6, a = fabsf(u);
            int i0 = tg_e2m3_floor(a);
            if (i0 < 31) {
                const float lo = tg_e2m3_value(i0), hi = tg_e2m3_value(i0 + 1);
                const float frac = (a - lo) / (hi - lo);
                float th = 0.5f;
                if (n_variants > 1) {
                    const uint32_t hh = tg_hash32(((uint32_t)row * 0x9E3779B1u) ^ ((uint32_t)(tap * cin_pad + ci) * 0x85EBCA77u) ^ salt);
                    th = ((float)vr + 0.5f) / (float)n_variants + (float)(hh >> 8) * (1.0f / 16777216.0f);
                    if (th >= 1.0f) th -= 1.0f;
                }
                if (frac > th) ++i0;
            }
            const unsigned code = (unsigned)i0 | (u < 0.f ? 32u : 0u);
            acc |= (unsigned long long)code << nbits;
            nbits += 6;
            if (nbits >= 32) { out[word++] = (unsigned)acc; acc >>= 32; nbits -= 32; }
        }
        unsigned* f = dst + ((size_t)v * per_variant / 64 + ((size_t)mt * taps + tap) * nq + q) * (TFRAG6_BYTES / 4);
        f[l * 4 + 0] = out[0]; f[l * 4 + 1] = out[1]; f[l * 4 + 2] = out[2]; f[l * 4 + 3] = out[3];
        f[256 + l * 2 + 0] = out[4]; f[256 + l * 2 + 1] = out[5];
    }
}

inline size_t tpacked_halfs(int m_tiles, int taps, int cin_pad, int planes, int n_variants) {
    return (size_t)n_variants * m_tiles * taps * (cin_pad / 16) * planes * TFRAG_HALFS;
}

}  // namespace dsvc
