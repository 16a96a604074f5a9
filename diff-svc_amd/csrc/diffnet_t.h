// Epilogues of the DiffNet layer kernels on the tgemm engine (tgemm.h) and the small kernels around them.
// Reference math: network/diff/net.py:58-135 (DiffNet / ResidualBlock), network/diff/diffusion.py:131-163 (p_sample).
//
// Conventions shared by every epilogue below (see tgemm.h): the accumulator of N-tile `nt` belongs to frame
// row0 + 32*nt + (lane & 31); with h = lane >> 5 its 16 registers are 16 consecutive output channels
// 32*m_tile + 16*h + r (plain tiles, packed with trow_to_ch16) or, for the gate kernel, registers 0..7 = gate and
// 8..15 = filter pre-activations of g-channels 16*m_tile + 8*h + (r & 7) (packed with trow_to_ch8).
#pragma once
#include "diffnet_kernels.h"
#include "tgemm.h"
#include "tepi_util.h"

namespace dsvc {

// which rows of the frame-major buffers are real frames: rowclip[row] = the clip a row belongs to, or -1 for a gap / padded row
// (row = clip*clip_stride + t with t < that clip's own length).  The table is rebuilt per call from the per-clip lengths
// (k_build_rowclip): the epilogues pay one 4-byte load per row instead of an integer division, and a clip shorter than the
// batch's T gets true zero padding (dsvc_sample_args.clip_lens).
struct RowMap {
    int clip_stride;
    const int* rowclip;         // device [rows_alloc]
    __device__ __forceinline__ bool valid(int row, int& clip, int& tl) const {
        clip = rowclip[row];
        tl = row - clip * clip_stride;
        return clip >= 0;
    }
};

__global__ void k_build_rowclip(int* __restrict__ rowclip, const int* __restrict__ lens, int clip_stride, int rows, int rows_alloc) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows_alloc) return;
    const int clip = row / clip_stride, tl = row - clip * clip_stride;
    rowclip[row] = (row < rows && tl < lens[clip]) ? clip : -1;
}

// ---- K4+K5+K6: dilated conv + hoisted conditioner projection + gate -> g (fp16) ----
struct TEpiGate {
    static constexpr bool RAGGED_SKIP = true;     // (tgemm.h: epi_ragged_skip)
    struct Args {
        const float* cproj;     // accumulator-tiled (C/16 m_tiles per frame tile): registers 0..7 gate, 8..15 filter; both biases
                                // folded in, pre-scaled like the weights (see gate_act_scaled)
        _Float16* g;            // [rows][ldg] fp16
        int C, ldg;
        int lo_off;             // > 0: also store the lo plane fp16(g - fp16(g)) lo_off halfs into the row (split activations, tgemm NA = 2)
#ifdef DSVC_PROFILING
        int abl;                // profiling build, round 6 (tools/gpu_r6_ablate.py; WRONG results): 1 = the accumulator init reads HALF of cproj's bytes
                                // (1536 instead of 3072 B per frame): an upper bound on what ANY compressed cproj (fp16 hi + 6-bit lo: 2112 B) can buy
#endif
    };
    template <int NT_N>
    __device__ __forceinline__ void init(const Args& e, int mt, int row0, int lane, f32x16 (&acc)[NT_N]) const {
        const int n_mt = e.C >> 4;
#pragma unroll
        for (int nt = 0; nt < NT_N; ++nt) {
            const float* p = e.cproj + tiled_lane_base(row0 + 32 * nt, n_mt, mt, lane);
#ifdef DSVC_PROFILING
            if (e.abl & 1) {
                const f32x4 h0 = ld4_nt(p), h1 = ld4_nt(p + 256);
#pragma unroll
                for (int i = 0; i < 4; ++i) { acc[nt][i] = h0[i]; acc[nt][4 + i] = h1[i]; acc[nt][8 + i] = h0[i]; acc[nt][12 + i] = h1[i]; }
                continue;
            }
#endif
            const f32x4 v0 = ld4_nt(p), v1 = ld4_nt(p + 256), v2 = ld4_nt(p + 512), v3 = ld4_nt(p + 768);
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc[nt][i] = v0[i]; acc[nt][4 + i] = v1[i]; acc[nt][8 + i] = v2[i]; acc[nt][12 + i] = v3[i]; }
        }
    }
    template <int NT_N>
    __device__ __forceinline__ void finish(const Args& e, int mt, int row0, int lane, f32x16 (&acc)[NT_N]) const {
#pragma unroll
        for (int nt = 0; nt < NT_N; ++nt) {
            const int frame = row0 + 32 * nt + (lane & 31);
            half8 o, ol;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float v = gate_act_scaled(acc[nt][r], acc[nt][8 + r]);
                o[r] = (_Float16)v;
                ol[r] = (_Float16)(v - (float)o[r]);
            }
            *reinterpret_cast<half8*>(e.g + (size_t)frame * e.ldg + mt * 16 + 8 * (lane >> 5)) = o;
            if (e.lo_off > 0) *reinterpret_cast<half8*>(e.g + (size_t)frame * e.ldg + e.lo_off + mt * 16 + 8 * (lane >> 5)) = ol;
        }
    }
};

// ---- K7+K8 (+K3 of the NEXT layer): output 1x1; residual half updates x and emits the next layer's fp16 operand
//      xh = fp16(x + film_next), skip half accumulates ----
struct TEpiResSkip {
    static constexpr bool RAGGED_SKIP = true;     // (tgemm.h: epi_ragged_skip)
    struct Args {
        float* x32;             // residual stream (in/out), accumulator-tiled with C/32 m_tiles
        _Float16* xh;           // [rows][ldh] next layer's MFMA operand, row 0 (guard rows precede); null on the last layer
        float* skip;            // running skip sum, accumulator-tiled with C/32 m_tiles
        _Float16* skiph;        // [rows][2*ldh] fp16 hi|lo planes of the skip sum for the skip projection; last layer only
        const float* bias;      // [2C]
        const float* film;      // next layer's FiLM table slice: film[step*film_step_stride + c]; null on the last layer
        int film_step_stride;
        StepRef step;
        int C, ldh;
        int first;              // layer 0: skip = s (no read)
        RowMap rm;
        int stream;             // large batches: the fp32 residual / skip tiles are touched once per layer and do not fit any
                                // cache -> non-temporal loads and stores (no dirty-line build-up to flush at the kernel boundary)
        int xh_lo;              // > 0: xh rows are [hi | lo] planes (ldh halfs per row, lo plane xh_lo halfs in): split activations
    };
    // one N-tile's accumulator init (the residual-stream / skip-sum tile of frames row0 + 32 nt ..)
    __device__ __forceinline__ void init_one(const Args& e, int mt, int row0, int lane, int nt, f32x16& a) const {
        const int rt = e.C >> 5;                            // residual tiles
        const bool res = mt < rt;
        const float* base = res ? e.x32 : e.skip;
        const int tl_mt = res ? mt : mt - rt;
        if (!res && e.first) {
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = 0.f;
        } else {
            const float* p = base + tiled_lane_base(row0 + 32 * nt, rt, tl_mt, lane);
            f32x4 v0, v1, v2, v3;
#ifdef DSVC_PROFILING
            // round 6 byte ablation (DSVC_TL_STREAM=3, WRONG results): 3 of the 4 dwordx4 per lane -- what 3-byte residual / skip elements could buy
            if (e.stream & 2) { v0 = ld4_nt(p); v1 = ld4_nt(p + 256); v2 = ld4_nt(p + 512); v3 = v2; } else
#endif
            if (e.stream) { v0 = ld4_nt(p); v1 = ld4_nt(p + 256); v2 = ld4_nt(p + 512); v3 = ld4_nt(p + 768); }
            else { v0 = ld4(p); v1 = ld4(p + 256); v2 = ld4(p + 512); v3 = ld4(p + 768); }
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = v0[i]; a[4 + i] = v1[i]; a[8 + i] = v2[i]; a[12 + i] = v3[i]; }
        }
    }
    template <int NT_N>
    __device__ __forceinline__ void init(const Args& e, int mt, int row0, int lane, f32x16 (&acc)[NT_N]) const {
#pragma unroll
        for (int nt = 0; nt < NT_N; ++nt) init_one(e, mt, row0, lane, nt, acc[nt]);
    }
    // what an epilogue needs beside the accumulators: the tile's bias, the clips of the lane's NT_N frames (ONE batch of loads at the top, not a
    // dependent load inside every N-tile's epilogue) and -- when every clip is at the same diffusion step (the sampler's loops) -- the next
    // layer's FiLM values of the lane's 16 channels once instead of once per N-tile
    template <int NT_N>
    struct Ctx { float b[16]; int clip[NT_N]; float film[16]; bool film_shared; };
    // HOIST = false: only the bias (the two-launch kernels: their register allocation has no room for more live values across the N-tiles)
    // step_shared: the diffusion step when every clip is at the same one, read ONCE at kernel entry by the caller (a scalar load there instead
    // of a dependent load in front of every epilogue's FiLM loads), or -1 (per-clip steps / not read yet: e.step.get())
    template <bool HOIST, int NT_N>
    __device__ __forceinline__ void prepare(const Args& e, int mt, int row0, int lane, Ctx<NT_N>& c, int step_shared = -1) const {
        const int rt = e.C >> 5;
        const bool res = mt < rt;
        const int cb = (res ? mt : mt - rt) * 32 + 16 * (lane >> 5);
        const bool need_clip = HOIST && ((res && e.xh) || (!res && e.skiph));
#pragma unroll
        for (int nt = 0; nt < NT_N; ++nt) c.clip[nt] = need_clip ? e.rm.rowclip[row0 + 32 * nt + (lane & 31)] : -1;
        {
            const float* bp = e.bias + (res ? 0 : e.C) + cb;
#pragma unroll
            for (int q = 0; q < 4; ++q) { const f32x4 v = ld4(bp + 4 * q); c.b[4 * q] = v[0]; c.b[4 * q + 1] = v[1]; c.b[4 * q + 2] = v[2]; c.b[4 * q + 3] = v[3]; }
        }
        c.film_shared = HOIST && res && e.xh && !e.step.per_clip;
        if (c.film_shared) {
            const float* fp = e.film + (size_t)(step_shared >= 0 ? step_shared : e.step.get(0)) * e.film_step_stride + cb;
#pragma unroll
            for (int q = 0; q < 4; ++q) { const f32x4 f = ld4(fp + 4 * q); c.film[4 * q] = f[0]; c.film[4 * q + 1] = f[1]; c.film[4 * q + 2] = f[2]; c.film[4 * q + 3] = f[3]; }
        }
    }
    template <bool HOIST, int NT_N>
    __device__ __forceinline__ void finish_one(const Args& e, int mt, int row0, int lane, int nt, const f32x16& a, const Ctx<NT_N>& c) const {
        const int rt = e.C >> 5;
        const bool res = mt < rt;
        const int cb = (res ? mt : mt - rt) * 32 + 16 * (lane >> 5);
        const int frame = row0 + 32 * nt + (lane & 31);
        float v[16];
        if (res) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = (a[i] + c.b[i]) * 0.70710678118654752440f;
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = a[i] + c.b[i];
        }
        float* p = (res ? e.x32 : e.skip) + tiled_lane_base(row0 + 32 * nt, rt, res ? mt : mt - rt, lane);
#ifdef DSVC_PROFILING
        if (e.stream & 2) {
#pragma unroll
            for (int q = 0; q < 3; ++q) st4_nt(p + 256 * q, f32x4{v[4 * q] + v[12 + q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]});
        } else
#endif
        if (e.stream) {
#pragma unroll
            for (int q = 0; q < 4; ++q) st4_nt(p + 256 * q, f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]});
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) st4(p + 256 * q, f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]});
        }
        if (res && e.xh) {
            const int clip = HOIST ? c.clip[nt] : e.rm.rowclip[frame];
            const bool ok = clip >= 0;
            float hv[16];
            if (ok && c.film_shared) {
#pragma unroll
                for (int i = 0; i < 16; ++i) hv[i] = v[i] + c.film[i];
            } else if (ok) {
                const float* fp = e.film + (size_t)e.step.get(clip) * e.film_step_stride + cb;
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
            _Float16* q = e.xh + (size_t)frame * e.ldh + cb;
            if (e.xh_lo > 0) {
                store_hi_lo16(q, e.xh_lo, hv);
            } else {
                half8 o0, o1;
#pragma unroll
                for (int i = 0; i < 8; ++i) { o0[i] = (_Float16)hv[i]; o1[i] = (_Float16)hv[8 + i]; }
                *reinterpret_cast<half8*>(q) = o0;
                *reinterpret_cast<half8*>(q + 8) = o1;
            }
        }
        if (!res && e.skiph) {
            const bool ok = (HOIST ? c.clip[nt] : e.rm.rowclip[frame]) >= 0;
            float hv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) hv[i] = ok ? v[i] : 0.f;
            const int cp = e.xh_lo > 0 ? e.xh_lo : e.ldh;          // padded channel count (ldh is twice that when the xh rows are split)
            store_hi_lo16(e.skiph + (size_t)frame * (2 * cp) + cb, cp, hv);
        }
    }
    template <int NT_N>
    __device__ __forceinline__ void finish(const Args& e, int mt, int row0, int lane, f32x16 (&acc)[NT_N]) const {
        Ctx<NT_N> c;
        prepare<false>(e, mt, row0, lane, c);
#pragma unroll
        for (int nt = 0; nt < NT_N; ++nt) finish_one<false>(e, mt, row0, lane, nt, acc[nt], c);
    }
    // finish of tile mt and the accumulator init of the wave's NEXT tile mt_n, N-tile by N-tile: an N-tile's init loads go out right behind ITS
    // stores, while the later N-tiles' epilogues still run.  For the fused layer kernel's G6 output phase (tlayer.h), which has no registers for
    // a prefetched second accumulator set and used to expose the whole init burst's latency after the last store.
    template <int NT_N>
    __device__ __forceinline__ void finish_then_init(const Args& e, int mt, int mt_n, int row0, int lane, f32x16 (&acc)[NT_N], int step_shared = -1) const {
        Ctx<NT_N> c;
        prepare<true>(e, mt, row0, lane, c, step_shared);
#pragma unroll
        for (int nt = 0; nt < NT_N; ++nt) {
            finish_one<true>(e, mt, row0, lane, nt, acc[nt], c);
            init_one(e, mt_n, row0, lane, nt, acc[nt]);
        }
    }
};

// ---- K1: input projection + ReLU -> x (fp32) and layer 0's operand xh = fp16(x + film_0) ----
struct TEpiInProj {
    static constexpr bool RAGGED_SKIP = true;     // (tgemm.h: epi_ragged_skip)
    struct Args {
        float* x32; _Float16* xh;
        const float* bias; const float* film; int film_step_stride; StepRef step;
        int C, ldh; RowMap rm;
        int xh_lo;              // > 0: xh rows are [hi | lo] planes, lo plane xh_lo halfs in (split activations)
    };
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
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = fmaxf(acc[nt][i] + b[i], 0.f);
            float* p = e.x32 + tiled_lane_base(row0 + 32 * nt, e.C >> 5, mt, lane);
#pragma unroll
            for (int q = 0; q < 4; ++q) st4(p + 256 * q, f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]});
            int clip, tl;
            const bool ok = e.rm.valid(frame, clip, tl);
            float hv[16];
            if (ok) {
                const float* fp = e.film + (size_t)e.step.get(clip) * e.film_step_stride + cb;
#pragma unroll
                for (int i = 0; i < 16; ++i) hv[i] = v[i] + fp[i];
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) hv[i] = 0.f;
            }
            _Float16* q = e.xh + (size_t)frame * e.ldh + cb;
            if (e.xh_lo > 0) {
                store_hi_lo16(q, e.xh_lo, hv);
            } else {
                half8 o0, o1;
#pragma unroll
                for (int i = 0; i < 8; ++i) { o0[i] = (_Float16)hv[i]; o1[i] = (_Float16)hv[8 + i]; }
                *reinterpret_cast<half8*>(q) = o0;
                *reinterpret_cast<half8*>(q + 8) = o1;
            }
        }
    }
};

// ---- K9a: skip projection + ReLU -> fp16 hi|lo operand planes of the final projection ----
struct TEpiReluHalf {
    static constexpr bool RAGGED_SKIP = true;     // (tgemm.h: epi_ragged_skip)
    struct Args { _Float16* out; int ld; const float* bias; int cout; };   // out [rows][2*ld]: hi plane, lo plane at +ld
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
        if (cb >= e.cout) return;
        float b[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) b[i] = e.bias[cb + i];
#pragma unroll
        for (int nt = 0; nt < NT_N; ++nt) {
            const int frame = row0 + 32 * nt + (lane & 31);
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = fmaxf(acc[nt][i] + b[i], 0.f);
            store_hi_lo16(e.out + (size_t)frame * (2 * e.ld) + cb, e.ld, v);
        }
    }
};

// ---- K9b: final projection -> eps (fp32), for PLMS and DiffNet.forward ----
struct TEpiEps {
    static constexpr bool RAGGED_SKIP = true;     // (tgemm.h: epi_ragged_skip)
    struct Args { float* out; int M; const float* bias; };
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
        if (cb >= e.M) return;
#pragma unroll
        for (int nt = 0; nt < NT_N; ++nt) {
            const int frame = row0 + 32 * nt + (lane & 31);
            float* p = e.out + (size_t)frame * e.M + cb;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = ld4(e.bias + cb + 4 * q);
                st4(p + 4 * q, f32x4{acc[nt][4 * q] + bv[0], acc[nt][4 * q + 1] + bv[1], acc[nt][4 * q + 2] + bv[2], acc[nt][4 * q + 3] + bv[3]});
            }
        }
    }
};

// ---- K9b+K10: final projection fused with the DDPM posterior step (diffusion.py:131-163); also refreshes the
//      fp16 copy of the state that the next step's input projection reads ----
struct TEpiDdpm {
    static constexpr bool RAGGED_SKIP = true;     // (tgemm.h: epi_ragged_skip)
    struct Args {
        float* x;                   // [rows][M] sampler state (in/out)
        _Float16* xsh;              // [rows][2*ldh] fp16 hi|lo planes of the state (zero on gap rows)
        const float* bias;          // [M]
        int M, ldh;
        DdpmTables tab;
        StepRef step;
        RowMap rm;
        const unsigned long long* seedp;   // device (see EpiDdpm)
        const int* clipid;
    };
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
        if (cb >= e.M) return;
        float b[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) b[i] = e.bias[cb + i];
#pragma unroll
        for (int nt = 0; nt < NT_N; ++nt) {
            const int frame = row0 + 32 * nt + (lane & 31);
            int clip, tl;
            if (!e.rm.valid(frame, clip, tl)) continue;
            const int t = e.step.get(clip);
            const float ra = e.tab.sqrt_recip_ac[t], rb = e.tab.sqrt_recipm1_ac[t], c1 = e.tab.coef1[t], c2 = e.tab.coef2[t];
            const float sg = e.tab.sigma[t];
            float* px = e.x + (size_t)frame * e.M + cb;
            float hv[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 xt = ld4(px + 4 * q);
                float z[4] = {0.f, 0.f, 0.f, 0.f};
                if (t > 0) {
                    const unsigned el = (unsigned)tl * (unsigned)e.M + (unsigned)(cb + 4 * q);
                    philox_normal4(el >> 2, (unsigned)t, (unsigned)e.clipid[clip], PURPOSE_DDPM_NOISE, *e.seedp, z);
                }
                f32x4 out;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float eps = acc[nt][4 * q + i] + b[4 * q + i];
                    float x0 = ra * xt[i] - rb * eps;
                    x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
                    float o = c1 * x0 + c2 * xt[i];
                    if (t > 0) o += sg * z[i];
                    out[i] = o;
                    hv[4 * q + i] = o;
                }
                st4(px + 4 * q, out);
            }
            store_hi_lo16(e.xsh + (size_t)frame * (2 * e.ldh) + cb, e.ldh, hv);
        }
    }
};

// ---- small kernels of the tgemm path ----

// fp32 frame-major [rows][C] -> fp16 hi|lo planes [rows][2*ld] (valid rows only; gap rows and pad columns stay zero)
__global__ void k_rows_to_half(const float* __restrict__ src, _Float16* __restrict__ dst, int C, int ld, RowMap rm, int rows) {
    // (rows beyond a clip's own length are skipped: they keep whatever an earlier call left there, which no valid frame ever reads --
    //  the projections that consume this buffer are 1x1)
    const int per_row = C >> 2;
    const long long n = (long long)rows * per_row;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / per_row), c4 = (int)(i - (long long)row * per_row) * 4;
        int clip, tl;
        if (!rm.valid(row, clip, tl)) continue;
        const f32x4 v = ld4(src + (size_t)row * C + c4);
        _Float16* q = dst + (size_t)row * (2 * ld) + c4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const _Float16 h = (_Float16)v[j];
            q[j] = h;
            q[ld + j] = (_Float16)(v[j] - (float)h);
        }
    }
}

// accumulator-tiled fp32 -> frame-major [rows][width]  (debug taps).  paired = the gate kernel's register meaning
// (m_tile = 16 channels: regs 0..7 gate, 8..15 filter; output column = the conv_gemm-free order [block 16][gate|filter])
__global__ void k_untile(const float* __restrict__ src, float* __restrict__ dst, int width, int rows, int paired) {
    const long long n = (long long)rows * width;
    const int n_mt = width >> 5;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / width), c = (int)(i - (long long)row * width);
        const int mt = c >> 5, w = c & 31;
        int h, r;
        if (paired) { h = (w >> 3) & 1; r = 8 * (w >> 4) + (w & 7); }
        else { h = w >> 4; r = w & 15; }
        dst[i] = src[tiled_off(row, n_mt, mt, r, h)];
    }
}

// fp16 [rows][ld] -> fp32 [rows][C]  (debug taps)
__global__ void k_half_to_rows(const _Float16* __restrict__ src, float* __restrict__ dst, int C, int ld, int rows) {
    const long long n = (long long)rows * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / C), c = (int)(i - (long long)row * C);
        dst[i] = (float)src[(size_t)row * ld + c];
    }
}

}  // namespace dsvc
