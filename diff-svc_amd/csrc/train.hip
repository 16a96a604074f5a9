// Training step of the DiffNet denoiser behind the C ABI (include/dsvc.h: dsvc_trainer_*): diffusion loss forward + backward,
// gradients of every denoise_fn.* parameter and of fs2.pitch_embed.weight, plus the AdamW update.
// Reference: network/diff/diffusion.py:200-225 (q_sample, p_losses), network/diff/net.py:58-135 (the network that is
// differentiated), training/task/SVC_task.py:60-66,116-125 (AdamW, optimizer_step), utils/pl_utils.py:1081-1084 (grad-norm clip).
//
// Every contraction runs on MFMA with split fp16 operands (w = w_hi + w_lo, x = x_hi + x_lo: three products, fp32 accumulate -- fp32-class), so
// the gradients match the reference's fp32 autograd to ~1e-5 and nothing about the optimisation trajectory changes.  The residual layers --
// 93 % of the FLOPs -- run on the sampler's tgemm engine (tgemm.h; round 4): activations as fp16 [hi | lo] row planes written by the producing
// epilogue and DMA'd straight into LDS, weights re-packed per step into fragment order (k_tpack_batch); the data gradients' 2C-channel
// operands are streamed through LDS in K phases (tgemm.h KP).  The small projections at both ends stay on conv_gemm.h, the weight gradients
// on wgrad.h.  Layout: fp32 frame-major rows (row = clip*Tp + t, gap rows zero = the convs' zero padding) for what the backward pass
// re-reads as fp32 (x, sigma, tau), fp16 [hi | lo] row planes -- one set per layer for x + film and g -- for what it contracts.
//   forward   y = conv_dil(x + film) + W_c cond + b ;  g = sigmoid(y_a) tanh(y_b) ;  [r; s] = W_o g + b ;  x' = (x + r)/sqrt2 ; skip += s
//             (sigma, tau, g and every layer's x are kept: ~1.3 GB for the 64 x 128-frame batch)
//   backward  dO = [dx/sqrt2 ; dskip] ;  dg = W_o^T dO ;  dy = dg (tau sigma(1-sigma) ; sigma(1-tau^2)) ;  dx = dx/sqrt2 + convT(dy) ;
//             (the conditioner's data gradient is never formed: k_bin_sums) ;  weight gradients dW[o][k] = sum_n A[n][o] B[n][k], the contraction index is
//             the frame index: round 5 contracts the residual layers' straight from those frame-major planes (wgrad.h: wgrad_fm_kernel, transposing LDS
//             reads; conv taps = row offsets, bias sums = MFMAs against ones); the three projections at both ends, and architectures whose channel
//             counts are not multiples of 128, go through channel-major copies (k_split_t + wgrad_nt_kernel).
//   loss scaling: d loss / d eps is ~1/(B M T) ~ 1e-6, inside fp16's SUBNORMAL range where a hi + lo split keeps 4 bits; the backward
//             pass is linear in it, so it runs on deps * 2^k (k chosen from 1/(B M T): operands in fp16's normal range) and the flat
//             gradient buffer is multiplied by 2^-k once at the end -- exact, powers of two.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/dsvc.h"
#include "../../include/dsvc_debug.h"
#include "cg_util.h"
#include "wgrad.h"
#include "tepi_util.h"

using namespace dsvc;

namespace {

constexpr uint32_t PURPOSE_TRAIN_NOISE = 5;
constexpr float RSQRT2 = 0.70710678118654752440f;

// ------------------------------------------------------------------------------------------------
// conv_gemm epilogues of the training graph.  "valid" = row < n_rows with t < clip_len (gap rows carry no data).
// ------------------------------------------------------------------------------------------------
struct RowInfo {
    int clip_stride, clip_len, n_valid;
    __device__ __forceinline__ bool valid(int row) const { return row < n_valid && (row - (row / clip_stride) * clip_stride) < clip_len; }
};

// out = act(acc + bias); relu optional; invalid rows -> 0 (keeps gap rows zero for the next conv)
struct EpStore {
    static constexpr bool PAIRED = false;
    struct Args { float* out; int ld; const float* bias; int cout; int relu; RowInfo ri; };
    __device__ __forceinline__ void one(const Args& e, int row, int col, float v) const {
        if (col >= e.cout) return;
        v += e.bias ? e.bias[col] : 0.f;
        if (e.relu) v = fmaxf(v, 0.f);
        e.out[(size_t)row * e.ld + col] = e.ri.valid(row) ? v : 0.f;
    }
};

// out (+)= acc * scale on valid rows, optionally gated by mask[row][col] > 0 (ReLU backward)
struct EpBwd {
    static constexpr bool PAIRED = false;
    struct Args { float* out; int ld; int cout; const float* mask; int ldm; float scale; int accumulate; RowInfo ri; };
    __device__ __forceinline__ void one(const Args& e, int row, int col, float v) const {
        if (col >= e.cout) return;
        v *= e.scale;
        if (e.mask && !(e.mask[(size_t)row * e.ldm + col] > 0.f)) v = 0.f;
        if (!e.ri.valid(row)) v = 0.f;
        float* p = e.out + (size_t)row * e.ld + col;
        *p = e.accumulate ? *p + v : v;
    }
};

// ------------------------------------------------------------------------------------------------
// Round 4: the FORWARD pass of the residual layers on the sampler's tgemm engine (tgemm.h, NA = 2: the activations as fp16 [hi | lo] row planes
// written by the producing epilogue and DMA'd straight into LDS, weights streamed in fragment order -- the same three-MFMA split products as
// conv_gemm, 2.2x its rate at this shape).  The backward pass keeps reading the fp32 frame-major stores (x^l, sigma, tau, g, skip), which these
// epilogues write as 16-byte pieces of a lane's 8 / 16 consecutive channels.
//   accumulator of N-tile nt = frame row0 + 32 nt + (lane & 31); h = lane >> 5
//   gate kernel:  registers 0..7 = gate, 8..15 = filter pre-activations of g-channels 16 m_tile + 8 h + (r & 7), PRE-SCALED by -log2(e) /
//                 -2 log2(e) through the packed weight rows and the hoisted conditioner projection (diffnet_t.h: gate_act_scaled)
//   out kernel:   registers = channels 32 m_tile + 16 h + r; tiles [0, C/32) are the residual half, [C/32, 2C/32) the skip half
// ------------------------------------------------------------------------------------------------
// W_c,l cond + b_c,l + b_d,l for ALL layers, accumulator-tiled per layer (what the gate kernels load as their accumulator init)
struct TEpiCprojT {
    struct Args { float* out; long long slab; int mpl; const float* bd; const float* bc; long long layer_stride; int C; };
    template <int NT_N>
    __device__ __forceinline__ void init(const Args&, int, int, int, f32x16 (&acc)[NT_N]) const {
#pragma unroll
        for (int nt = 0; nt < NT_N; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
    }
    template <int NT_N>
    __device__ __forceinline__ void finish(const Args& e, int mt, int row0, int lane, f32x16 (&acc)[NT_N]) const {
        const int l = mt / e.mpl, ml = mt - l * e.mpl;
        const int cg = 16 * ml + 8 * (lane >> 5);
        const float* bd = e.bd + (long long)l * e.layer_stride + cg;
        const float* bc = e.bc + (long long)l * e.layer_stride + cg;
        float b[16];
#pragma unroll
        for (int r = 0; r < 8; ++r) { b[r] = GATE_SCALE * (bd[r] + bc[r]); b[8 + r] = FILT_SCALE * (bd[e.C + r] + bc[e.C + r]); }
#pragma unroll
        for (int nt = 0; nt < NT_N; ++nt) {
            float* p = e.out + (long long)l * e.slab + tiled_lane_base(row0 + 32 * nt, e.mpl, ml, lane);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                st4(p + 256 * q, f32x4{acc[nt][4 * q] + b[4 * q], acc[nt][4 * q + 1] + b[4 * q + 1], acc[nt][4 * q + 2] + b[4 * q + 2], acc[nt][4 * q + 3] + b[4 * q + 3]});
        }
    }
};

// gate forward (net.py:71-77): sigma, tau (kept for the backward pass), g = sigma tau as fp32 rows and as the output projection's fp16 planes
struct TEpiGateT {
    struct Args { const float* cproj; float* sig; float* tau; float* g; _Float16* gh; int C, Cp; RowInfo ri; };
    template <int NT_N>
    __device__ __forceinline__ void init(const Args& e, int mt, int row0, int lane, f32x16 (&acc)[NT_N]) const {
        const int n_mt = e.C >> 4;
#pragma unroll
        for (int nt = 0; nt < NT_N; ++nt) {
            const float* p = e.cproj + tiled_lane_base(row0 + 32 * nt, n_mt, mt, lane);
            const f32x4 v0 = ld4_nt(p), v1 = ld4_nt(p + 256), v2 = ld4_nt(p + 512), v3 = ld4_nt(p + 768);
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc[nt][i] = v0[i]; acc[nt][4 + i] = v1[i]; acc[nt][8 + i] = v2[i]; acc[nt][12 + i] = v3[i]; }
        }
    }
    template <int NT_N>
    __device__ __forceinline__ void finish(const Args& e, int mt, int row0, int lane, f32x16 (&acc)[NT_N]) const {
        const int cg = 16 * mt + 8 * (lane >> 5);
#pragma unroll
        for (int nt = 0; nt < NT_N; ++nt) {
            const int frame = row0 + 32 * nt + (lane & 31);
            const bool ok = e.ri.valid(frame);
            float s[8], t[8], gv[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float bf = __builtin_amdgcn_fmed3f(acc[nt][8 + r], -43.28f, 43.28f);      // |b| <= 15 (gate_act_scaled)
                const float e1 = __builtin_amdgcn_exp2f(acc[nt][r]), e2 = __builtin_amdgcn_exp2f(bf);
                s[r] = __builtin_amdgcn_rcpf(1.0f + e1);
                t[r] = (1.0f - e2) * __builtin_amdgcn_rcpf(1.0f + e2);
                gv[r] = ok ? s[r] * t[r] : 0.f;
            }
            const size_t o = (size_t)frame * e.C + cg;
            st4(e.sig + o, f32x4{s[0], s[1], s[2], s[3]}); st4(e.sig + o + 4, f32x4{s[4], s[5], s[6], s[7]});
            st4(e.tau + o, f32x4{t[0], t[1], t[2], t[3]}); st4(e.tau + o + 4, f32x4{t[4], t[5], t[6], t[7]});
            if (e.g) { st4(e.g + o, f32x4{gv[0], gv[1], gv[2], gv[3]}); st4(e.g + o + 4, f32x4{gv[4], gv[5], gv[6], gv[7]}); }      // (null: nothing reads g as fp32 rows, wgrad_fm takes the planes)
            half8 hi, lo;
#pragma unroll
            for (int r = 0; r < 8; ++r) { hi[r] = (_Float16)gv[r]; lo[r] = (_Float16)(gv[r] - (float)hi[r]); }
            _Float16* q = e.gh + (size_t)frame * (2 * e.Cp) + cg;
            *reinterpret_cast<half8*>(q) = hi;
            *reinterpret_cast<half8*>(q + e.Cp) = lo;
        }
    }
};

// output projection forward (net.py:79-84): x^{l+1} = (x^l + r) / sqrt 2 into the next slab (+ the next gate's operand planes fp16(x^{l+1} + film_{l+1})),
// skip += s; invalid rows -> 0 everywhere (the convs' zero padding)
struct TEpiResSkipT {
    struct Args { const float* x; float* xnext; float* skip; _Float16* xh; const float* bias; const float* film; int film_stride; int C, Cp; int first; RowInfo ri; };
    template <int NT_N>
    __device__ __forceinline__ void init(const Args& e, int mt, int row0, int lane, f32x16 (&acc)[NT_N]) const {
        const int rt = e.C >> 5;
        const bool res = mt < rt;
        const int cb = (res ? mt : mt - rt) * 32 + 16 * (lane >> 5);
#pragma unroll
        for (int nt = 0; nt < NT_N; ++nt) {
            if (!res && e.first) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
            } else {
                const float* p = (res ? e.x : e.skip) + (size_t)(row0 + 32 * nt + (lane & 31)) * e.C + cb;
                const f32x4 v0 = ld4(p), v1 = ld4(p + 4), v2 = ld4(p + 8), v3 = ld4(p + 12);
#pragma unroll
                for (int i = 0; i < 4; ++i) { acc[nt][i] = v0[i]; acc[nt][4 + i] = v1[i]; acc[nt][8 + i] = v2[i]; acc[nt][12 + i] = v3[i]; }
            }
        }
    }
    template <int NT_N>
    __device__ __forceinline__ void finish(const Args& e, int mt, int row0, int lane, f32x16 (&acc)[NT_N]) const {
        const int rt = e.C >> 5;
        const bool res = mt < rt;
        const int cb = (res ? mt : mt - rt) * 32 + 16 * (lane >> 5);
        float b[16];
        {
            const float* bp = e.bias + (res ? 0 : e.C) + cb;
#pragma unroll
            for (int q = 0; q < 4; ++q) { const f32x4 v = ld4(bp + 4 * q); b[4 * q] = v[0]; b[4 * q + 1] = v[1]; b[4 * q + 2] = v[2]; b[4 * q + 3] = v[3]; }
        }
#pragma unroll
        for (int nt = 0; nt < NT_N; ++nt) {
            const int frame = row0 + 32 * nt + (lane & 31);
            const bool ok = e.ri.valid(frame);
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = ok ? (res ? (acc[nt][i] + b[i]) * RSQRT2 : acc[nt][i] + b[i]) : 0.f;
            float* p = (res ? e.xnext : e.skip) + (size_t)frame * e.C + cb;
#pragma unroll
            for (int q = 0; q < 4; ++q) st4(p + 4 * q, f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]});
            if (res && e.xh) {
                float hv[16];
                if (ok) {
                    const float* fp = e.film + (size_t)(frame / e.ri.clip_stride) * e.film_stride + cb;
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
                store_hi_lo16(e.xh + (size_t)frame * (2 * e.Cp) + cb, e.Cp, hv);
            }
        }
    }
};

// ---- backward, data gradients (streamed-K kernels: tgemm.h KP) ----
// dg = W_o^T dO through the gate:  dy_a = dg tau sigma (1 - sigma),  dy_b = dg sigma (1 - tau^2); dy as fp32 rows (weight gradients, pitch-bin
// sums) and as the transposed conv's operand planes [hi 2C | lo 2C]
struct TEpiGateBwdT {
    struct Args { const float* sig; const float* tau; float* dy; _Float16* dyh; int C, C2p; RowInfo ri; };
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
#pragma unroll
        for (int nt = 0; nt < NT_N; ++nt) {
            const int frame = row0 + 32 * nt + (lane & 31);
            const bool ok = e.ri.valid(frame);
            const size_t o = (size_t)frame * e.C + cb;
            float da[16], db[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 s = ld4(e.sig + o + 4 * q), t = ld4(e.tau + o + 4 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = acc[nt][4 * q + i];
                    da[4 * q + i] = ok ? v * t[i] * s[i] * (1.0f - s[i]) : 0.f;
                    db[4 * q + i] = ok ? v * s[i] * (1.0f - t[i] * t[i]) : 0.f;
                }
            }
            if (e.dy) {      // (null: every reader takes the planes -- wgrad_fm_kernel, k_bin_sums)
                float* pa = e.dy + (size_t)frame * (2 * e.C) + cb;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    st4(pa + 4 * q, f32x4{da[4 * q], da[4 * q + 1], da[4 * q + 2], da[4 * q + 3]});
                    st4(pa + e.C + 4 * q, f32x4{db[4 * q], db[4 * q + 1], db[4 * q + 2], db[4 * q + 3]});
                }
            }
            _Float16* ph = e.dyh + (size_t)frame * (2 * e.C2p) + cb;
            store_hi_lo16(ph, e.C2p, da);
            store_hi_lo16(ph + e.C, e.C2p, db);
        }
    }
};

// dxin = convT(dy) (kept: its per-clip column sums are the FiLM gradient), dx <- dx / sqrt 2 + dxin, and the residual half of the next layer's
// dO = dx / sqrt 2 as fp32 rows (weight gradients) and operand planes
struct TEpiDxT {
    struct Args { float* dx; float* dxin; float* dO; _Float16* dOh; int C, C2p; RowInfo ri; };
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
#pragma unroll
        for (int nt = 0; nt < NT_N; ++nt) {
            const int frame = row0 + 32 * nt + (lane & 31);
            const bool ok = e.ri.valid(frame);
            const size_t o = (size_t)frame * e.C + cb;
            float ov[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 x = ld4(e.dx + o + 4 * q);
                f32x4 vi, vx, vo;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    vi[i] = ok ? acc[nt][4 * q + i] : 0.f;
                    vx[i] = x[i] * RSQRT2 + vi[i];
                    vo[i] = vx[i] * RSQRT2;
                    ov[4 * q + i] = vo[i];
                }
                st4(e.dxin + o + 4 * q, vi);
                st4(e.dx + o + 4 * q, vx);
                if (e.dO) st4(e.dO + (size_t)frame * (2 * e.C) + cb + 4 * q, vo);
            }
            store_hi_lo16(e.dOh + (size_t)frame * (2 * e.C2p) + cb, e.C2p, ov);
        }
    }
};

// fp32 rows [rows][ld_src] (+ add[clip]) -> fp16 [hi | lo] row planes [rows][2 Cp]: the tgemm operand of layer 0 and of the conditioner
// projection; invalid rows and the channel padding are written as zeros
__global__ void k_rows_to_planes(const float* __restrict__ src, int ld_src, int C, const float* __restrict__ add, int add_stride,
                                 _Float16* __restrict__ dst, int Cp, RowInfo ri, int rows) {
    const int per_row = Cp >> 2;
    const long long n = (long long)rows * per_row;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / per_row), c4 = (int)(i - (long long)row * per_row) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (c4 < C && ri.valid(row)) {
            const f32x4 x = ld4(src + (size_t)row * ld_src + c4);
            const float* ap = add ? add + (size_t)(row / ri.clip_stride) * add_stride + c4 : nullptr;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = x[j] + (ap ? ap[j] : 0.f);
        }
        _Float16* q = dst + (size_t)row * (2 * Cp) + c4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const _Float16 h = (_Float16)v[j];
            q[j] = h;
            q[Cp + j] = (_Float16)(v[j] - (float)h);
        }
    }
}

// the step's weight re-packs into tgemm fragment order, one launch (tgemm.h: k_tpack with two planes, one variant): blockIdx.y = descriptor
// element (packed row -> o, input channel ci, tap) = src[o * s_o + ci * s_i + (flip ? taps - 1 - tap : tap) * s_tap]: a Conv1d weight [O][I][taps] as it
// stands (s_o = I * taps, s_i = taps, s_tap = 1) or transposed for the data gradients
struct TPackDesc { const float* src; const int* rowmap; const float* rowscale; _Float16* dst; int I, taps, cin_pad, m_tiles; long long s_o, s_i, s_tap; int flip; };
__global__ void k_tpack_batch(const TPackDesc* __restrict__ descs) {
    const TPackDesc d = descs[blockIdx.y];
    const int nk16 = d.cin_pad >> 4;
    const long long total = (long long)d.m_tiles * d.taps * nk16 * 512;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        long long r = idx;
        const int e = (int)(r & 7), l = (int)((r >> 3) & 63);
        r >>= 9;
        const int k = (int)(r % nk16); r /= nk16;
        const int tap = (int)(r % d.taps);
        const int mt = (int)(r / d.taps);
        const int row = mt * 32 + (l & 31), ci = k * 16 + 8 * (l >> 5) + e;
        const int o = d.rowmap[row];
        const float w = (o >= 0 && ci < d.I) ? d.src[o * d.s_o + ci * d.s_i + (d.flip ? d.taps - 1 - tap : tap) * d.s_tap] * (d.rowscale ? d.rowscale[row] : 1.0f) : 0.f;
        const _Float16 hi = (_Float16)w;
        _Float16* f = d.dst + ((((size_t)mt * d.taps + tap) * nk16 + k) * 2) * 512 + l * 8 + e;
        f[0] = hi;
        f[512] = (_Float16)(w - (float)hi);
    }
}

constexpr int WGRAD_MAX_TILES = 288;     // frame slices x output tiles of one weight-gradient GEMM (workgroups per launch; 128 KB of scratch each)

// ------------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------------
// dW[o][k] = sum over the k-slices of a weight-gradient GEMM (deterministic: fixed order)
__global__ void k_wgrad_reduce(const float* __restrict__ part, float* __restrict__ dst, int n_slices, int n_k, int ld, int n_o, long long stride_o,
                               long long stride_k, long long off) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_k * n_o) return;
    const int row = i / n_o, col = i - row * n_o;
    float s = 0.f;
    for (int z = 0; z < n_slices; ++z) s += part[((size_t)z * n_k + row) * ld + col];
    dst[(long long)col * stride_o + (long long)row * stride_k + off] = s;
}

// [B, C, T] (reference layout) -> frame-major [B*stride][C] on valid rows (gap rows stay zero)
__global__ void k_bct_to_rows(const float* __restrict__ src, float* __restrict__ dst, int B, int C, int T, int stride) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, t = t0 + tx;
        tile[i][tx] = (c < C && t < T) ? src[((size_t)b * C + c) * T + t] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, c = c0 + tx;
        if (t < T && c < C) dst[((size_t)b * stride + t) * C + c] = tile[tx][i];
    }
}

// Zero the rows of a [slabs][rows][ld] fp32 buffer that the step never writes: the gap rows t in [T, Tp) of every clip (the convs' zero padding and
// the separation between clips) and the tail rows [B * Tp, rows).  After a change of (B, T) they hold the previous layout's activations; every
// other row is rewritten by the step itself, so this replaces a memset of the whole workspace (1.9 GB at the 64 x 128 batch: a max_tokens loader,
// training/task/tts.py:60-88, changes the shape every step).  One thread per float4; ld % 4 == 0.
__global__ void k_zero_gap_rows(float* __restrict__ buf, long long slab_floats, int ld, int B, int Tp, int T, int rows) {
    const int gap = Tp - T, n_gap = B * gap, n_rows = n_gap + (rows - B * Tp), q_per_row = ld >> 2;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n_rows * q_per_row) return;
    const int ri = (int)(i / q_per_row), q = (int)(i - (long long)ri * q_per_row);
    const int row = ri < n_gap ? (ri / gap) * Tp + T + ri % gap : B * Tp + (ri - n_gap);
    reinterpret_cast<float4*>(buf + (size_t)blockIdx.y * slab_floats + (size_t)row * ld)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ void k_clamp_copy(int* __restrict__ dst, const int* __restrict__ src, int lo, int hi, int n, int* __restrict__ flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int v = src[i];
        if ((v < lo || v > hi) && flag) *flag = 1;        // sticky, host-mapped: dsvc_trainer_check reports it (as dsvc_denoiser_check does)
        dst[i] = v < lo ? lo : (v > hi ? hi : v);
    }
}

// x_t = sa[t_b] * norm_spec(mel) + sb[t_b] * noise   (diffusion.py:200-205,286-287); mel [B][T][M] -> frame-major [B*stride][M]
__global__ void k_make_xt(const float* __restrict__ mel, float* __restrict__ xt, const int* __restrict__ tstep, const float* __restrict__ sa,
                          const float* __restrict__ sb, const float* __restrict__ spec_min, const float* __restrict__ spec_max, int n_spec,
                          int B, int T, int M, int stride, unsigned long long seed, const int* __restrict__ clipid) {
    const int quads = T * M / 4;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (q >= quads) return;
    float z[4];
    philox_normal4((unsigned)q, 0u, (unsigned)clipid[b], PURPOSE_TRAIN_NOISE, seed, z);
    const int ts = tstep[b];
    const float a = sa[ts], s = sb[ts];
    const int el = q * 4, t = el / M, m0 = el - t * M;
    const float* mp = mel + ((size_t)b * T + t) * M + m0;
    float* xp = xt + ((size_t)b * stride + t) * M + m0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float lo = spec_min[n_spec == 1 ? 0 : m0 + i], hi = spec_max[n_spec == 1 ? 0 : m0 + i];
        const float x0 = (mp[i] - lo) / (hi - lo) * 2.0f - 1.0f;
        xp[i] = a * x0 + s * z[i];
    }
}

// loss (diffusion.py:213-223) and its gradient w.r.t. eps on valid rows: l1 mean |noise - eps|, l2 mean (noise - eps)^2
__global__ void k_loss(const float* __restrict__ eps, float* __restrict__ deps, float* __restrict__ loss, int B, int T, int M, int stride,
                       unsigned long long seed, const int* __restrict__ clipid, int l1, float inv_n, float gscale) {
    const int quads = T * M / 4;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    float part = 0.f;
    if (q < quads) {
        float z[4];
        philox_normal4((unsigned)q, 0u, (unsigned)clipid[b], PURPOSE_TRAIN_NOISE, seed, z);
        const int el = q * 4, t = el / M, m0 = el - t * M;
        const size_t o = ((size_t)b * stride + t) * M + m0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float d = eps[o + i] - z[i];
            if (l1) { part += fabsf(d); deps[o + i] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * (inv_n * gscale); }
            else { part += d * d; deps[o + i] = 2.0f * d * (inv_n * gscale); }
        }
    }
    // block reduction -> one atomic per block
    __shared__ float red[256];
    red[threadIdx.x] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(loss, red[0] * inv_n);
}

// dst[c] += sum over rows of src[row][c]   (bias gradients); grid (ceil(C/256), row chunks): a thread owns four adjacent columns (one
// 16-byte load per row) and every fourth row of the chunk, four loads in flight; C, ld % 4 == 0
// blockIdx.z walks a batch of independent sums src + z * src_z -> dst + z * dst_z (the layers' diffusion-projection biases in one launch)
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ src, float* __restrict__ dst, int rows, int C, int ld, int rows_per_block,
                                                long long src_z = 0, long long dst_z = 0) {
    src += (long long)blockIdx.z * src_z; dst += (long long)blockIdx.z * dst_z;
    const int c = blockIdx.x * 256 + (threadIdx.x & 63) * 4;
    const int sub = threadIdx.x >> 6;                        // 4 row phases
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (c < C) {
        int r = r0 + sub;
        for (; r + 4 < r1; r += 8) {
            const float4 a = *reinterpret_cast<const float4*>(src + (size_t)r * ld + c);
            const float4 b = *reinterpret_cast<const float4*>(src + (size_t)(r + 4) * ld + c);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
        }
        if (r < r1) {
            const float4 a = *reinterpret_cast<const float4*>(src + (size_t)r * ld + c);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        }
    }
    __shared__ float4 red[4][64];
    red[sub][threadIdx.x & 63] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
    __syncthreads();
    if (sub == 0 && c < C) {
        const int i = threadIdx.x;
        atomicAdd(dst + c + 0, red[0][i].x + red[1][i].x + red[2][i].x + red[3][i].x);
        atomicAdd(dst + c + 1, red[0][i].y + red[1][i].y + red[2][i].y + red[3][i].y);
        atomicAdd(dst + c + 2, red[0][i].z + red[1][i].z + red[2][i].z + red[3][i].z);
        atomicAdd(dst + c + 3, red[0][i].w + red[1][i].w + red[2][i].w + red[3][i].w);
    }
}

// dst[clip][c] = sum_t src[clip*stride + t][c]   (FiLM gradient); grid (ceil(C/128), B): a thread owns 4 columns and every eighth frame, 16-byte
// loads (round 5; one column and every fourth frame, 4-byte loads before: 10 us for 13 MB)
__global__ __launch_bounds__(256) void k_clip_colsum(const float* __restrict__ src, float* __restrict__ dst, int T, int C, int stride, int dst_ld) {
    const int c = blockIdx.x * 128 + (threadIdx.x & 31) * 4;
    const int sub = threadIdx.x >> 5;
    const int b = blockIdx.y;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (c < C)
        for (int t = sub; t < T; t += 8) { const f32x4 v = ld4(src + ((size_t)b * stride + t) * C + c); s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3]; }
    __shared__ f32x4 red[8][32];
    red[sub][threadIdx.x & 31] = s;
    __syncthreads();
    if (sub == 0 && c < C) {
        f32x4 r = red[0][threadIdx.x];
#pragma unroll
        for (int q = 1; q < 8; ++q) { const f32x4 v = red[q][threadIdx.x]; r[0] += v[0]; r[1] += v[1]; r[2] += v[2]; r[3] += v[3]; }
        st4(dst + (size_t)b * dst_ld + c, r);
    }
}

// dst[row][off + c] = src[row][c] * scale
__global__ void k_copy_cols(const float* __restrict__ src, float* __restrict__ dst, int rows, int C, int ld_dst, int off, float scale) {
    const size_t n = (size_t)rows * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / C), c = (int)(i - (size_t)row * C);
        dst[(size_t)row * ld_dst + off + c] = src[i] * scale;
    }
}

// p[i] *= scale (the loss scale leaves the gradients: a power of two, exact); n % 4 == 0 is not required
__global__ void k_scale_inplace(float* __restrict__ p, size_t n, float scale) {
    const size_t n4 = n / 4;
    float4* p4 = reinterpret_cast<float4*>(p);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = p4[i];
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        p4[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[n4 * 4 + threadIdx.x] *= scale;
}

// out[i] = in[i] * (mask[i] > 0)
__global__ void k_relu_bwd(const float* __restrict__ in, const float* __restrict__ mask, float* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = mask[i] > 0.f ? in[i] : 0.f;
}

// small dense fp32 GEMMs of the [B x C]-sized step-embedding path:  C[m][n] (+)= sum_k A(m,k) * B(k,n) (+ bias[n]),
// A(m,k) = ta ? A[k*lda + m] : A[m*lda + k],  B(k,n) = tb ? B[n*ldb + k] : B[k*ldb + n].  64 x 64 output tiles, 16-deep k chunks staged
// in LDS with the global loads running along whichever index is contiguous in memory (one thread per output with a stride-ldb operand
// measured 357 us for the 20 diffusion projections); blockIdx.z walks a batch of independent problems (one per residual layer).
struct SmallBatch { const float* A[32]; const float* B[32]; float* C[32]; const float* bias[32]; int n; };
// (round 5: a thread's 4 x 4 outputs are ADJACENT rows / columns -- one ds_read_b128 per operand and k instead of four ds_read_b32, 16-byte
//  stores; the k order of every output's fp32 chain is unchanged)
__global__ __launch_bounds__(256) void k_gemm_small_b(const SmallBatch t, int M, int N, int K, int lda, int ldb, int ldc, int ta, int tb, int accumulate) {
    // 64-deep k chunks (round 5; 16 before): a chunk is one round trip to memory and two barriers, and the K = 384 ... 768 problems of the
    // step-embedding / pitch-embedding path -- a few hundred workgroups -- spent their time waiting for 24 ... 48 of them
    constexpr int KC = 64;
    __shared__ __attribute__((aligned(16))) float As[KC][68], Bs[KC][68];
    const float* __restrict__ A = t.A[blockIdx.z];
    const float* __restrict__ Bm = t.B[blockIdx.z];
    float* __restrict__ Cm = t.C[blockIdx.z];
    const float* __restrict__ bias = t.bias[blockIdx.z];
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;          // thread -> outputs (m0 + 4 ty + i, n0 + 4 tx + j)
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += KC) {
        // a chunk's 2 x 16 loads per thread go out together (addresses clamped, values selected: no branch between them) and are parked in LDS
        // afterwards -- one memory round trip per chunk.  The index that is contiguous in memory runs fastest across the lanes; where that is k,
        // a wave takes 16 k x 4 rows, so that its LDS column stores spread over all banks
        float ra[16], rb[16];
        const int tq = threadIdx.x;
        const int kq = (tq & 15) + 16 * ((tq >> 6) & 3), rq0 = (tq >> 4) & 3;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int kk = ta ? (tq >> 6) + 4 * it : kq, mm = ta ? (tq & 63) : rq0 + 4 * it;
            const int m = m0 + mm, k = k0 + kk;
            const bool oka = m < M && k < K;
            const size_t ia = oka ? (ta ? (size_t)k * lda + m : (size_t)m * lda + k) : 0;
            const float va = A[ia];
            ra[it] = oka ? va : 0.f;
            const int kb = tb ? kq : (tq >> 6) + 4 * it, nn = tb ? rq0 + 4 * it : (tq & 63);
            const int n = n0 + nn, k2 = k0 + kb;
            const bool okb = n < N && k2 < K;
            const size_t ib = okb ? (tb ? (size_t)n * ldb + k2 : (size_t)k2 * ldb + n) : 0;
            const float vb = Bm[ib];
            rb[it] = okb ? vb : 0.f;
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int kk = ta ? (tq >> 6) + 4 * it : kq, mm = ta ? (tq & 63) : rq0 + 4 * it;
            As[kk][mm] = ra[it];
            const int kb = tb ? kq : (tq >> 6) + 4 * it, nn = tb ? rq0 + 4 * it : (tq & 63);
            Bs[kb][nn] = rb[it];
        }
        __syncthreads();
#pragma unroll 16
        for (int kk = 0; kk < KC; ++kk) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(&As[kk][4 * ty]), bv = *reinterpret_cast<const f32x4*>(&Bs[kk][4 * tx]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
    const int n = n0 + 4 * tx;
    const bool vec = n + 3 < N && (ldc & 3) == 0 && (reinterpret_cast<size_t>(Cm) & 15) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + 4 * ty + i;
        if (m >= M) continue;
        float* p = Cm + (size_t)m * ldc + n;
        if (vec) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[i][j] + (bias ? bias[n + j] : 0.f);
            if (accumulate) { const f32x4 o = ld4(p); v = f32x4{o[0] + v[0], o[1] + v[1], o[2] + v[2], o[3] + v[3]}; }
            st4(p, v);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (n + j < N) {
                    const float v = acc[i][j] + (bias ? bias[n + j] : 0.f);
                    p[j] = accumulate ? p[j] + v : v;
                }
        }
    }
}

// the Linear-forward shape of the same path:  C[m][n] = sum_k A[m*lda + k] * B[n*ldb + k] (+ bias[n]) with BOTH operands k-contiguous and
// only B x C outputs: a wave owns a 4 x 4 output block and its lanes split k (16-byte loads, coalesced), then reduce across the wave --
// hundreds of waves instead of the 6 ... 24 workgroups a 64 x 64 tiling gives these shapes.  K, lda, ldb % 4 == 0.
__global__ __launch_bounds__(256) void k_gemm_nt_dot(const SmallBatch t, int M, int N, int K, int lda, int ldb, int ldc) {
    const float* __restrict__ A = t.A[blockIdx.z];
    const float* __restrict__ Bm = t.B[blockIdx.z];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.y * 4, n0 = (blockIdx.x * 4 + wave) * 4;
    if (n0 >= N) return;
    float acc[4][4] = {};
    for (int k = lane * 4; k < K; k += 256) {
        float4 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = m0 + i < M ? *reinterpret_cast<const float4*>(A + (size_t)(m0 + i) * lda + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            b[i] = n0 + i < N ? *reinterpret_cast<const float4*>(Bm + (size_t)(n0 + i) * ldb + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] += a[i].x * b[j].x + a[i].y * b[j].y + a[i].z * b[j].z + a[i].w * b[j].w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = acc[i][j];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            acc[i][j] = v;
        }
    if (lane < 16) {
        const int i = lane >> 2, j = lane & 3;
        float v = 0.f;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                if (ii == i && jj == j) v = acc[ii][jj];
        if (m0 + i < M && n0 + j < N) t.C[blockIdx.z][(size_t)(m0 + i) * ldc + n0 + j] = v + (t.bias[blockIdx.z] ? t.bias[blockIdx.z][n0 + j] : 0.f);
    }
}

// SinusoidalPosEmb (net.py:32-44) for the batch's steps
__global__ void k_sin_emb_b(float* __restrict__ emb, const int* __restrict__ tstep, int B, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i % C;
    const int half = C / 2;
    const float scale = logf(10000.0f) / (float)(half - 1);
    const int k = c < half ? c : c - half;
    const float ang = (float)tstep[b] * expf((float)k * -scale);
    emb[i] = c < half ? sinf(ang) : cosf(ang);
}

// Mish (common_layers.py:485-487) forward and backward on [n]: y = x tanh(softplus(x))
__global__ void k_mish(const float* __restrict__ x, float* __restrict__ y, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    const float sp = v > 20.f ? v : log1pf(expf(v));
    y[i] = v * tanhf(sp);
}
__global__ void k_mish_bwd(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    const float sp = v > 20.f ? v : log1pf(expf(v));
    const float th = tanhf(sp);
    const float sg = 1.0f / (1.0f + expf(-v));              // d softplus / dx
    dx[i] = dy[i] * (th + v * (1.0f - th * th) * sg);
}

// ---- pitch-embedding gradient without the conditioner data gradient --------------------------------------------------------------------
// cond = (gather(hubert) + pitch_embed[pitch]) * (mel2ph > 0) (fs2.py:229-237) reaches the loss only through the layers' conditioner
// projections, and pitch_embed is the only trained tensor behind it:
//     d pitch_embed[p][h] = sum_frames(pitch = p, mel2ph > 0) sum_l sum_o W_c,l[o][h] dy_l[frame][o]
//                         = sum_l sum_o W_c,l[o][h] S_l[p][o],      S_l[p][o] = sum_frames(pitch = p, mel2ph > 0) dy_l[frame][o]
// so instead of dcond += W_c,l^T dy_l over ~8 700 rows per layer (20 GEMMs, 8 % of the step) the frames are sorted by pitch bin once per step,
// every layer adds its dy rows up per bin (one read of dy), and ONE [vocab x 2C] x [2C x H] product per layer closes it.  Bin 0 is the
// embedding's padding_idx and receives no gradient (nn.Embedding).
__device__ __forceinline__ bool bin_ok(const int* __restrict__ pitch, const int* __restrict__ mel2ph, int i, int vocab, int& p) {
    p = pitch[i];
    return p > 0 && p < vocab && !(mel2ph && mel2ph[i] <= 0);
}
__global__ void k_bin_count(const int* __restrict__ pitch, const int* __restrict__ mel2ph, int n, int vocab, int* __restrict__ count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int p;
    if (i < n && bin_ok(pitch, mel2ph, i, vocab, p)) atomicAdd(count + p, 1);
}
// one thread: cursor[p] = first slot of bin p in `order`; segments of at most seg_len frames: segs[s] = (bin, first slot, length)
__global__ void k_bin_scan(const int* __restrict__ count, int vocab, int* __restrict__ cursor, int* __restrict__ segs, int* __restrict__ n_segs, int seg_len) {
    if (blockIdx.x || threadIdx.x) return;
    int pos = 0, ns = 0;
    for (int p = 0; p < vocab; ++p) {
        cursor[p] = pos;
        for (int o = 0; o < count[p]; o += seg_len) {
            const int len = count[p] - o < seg_len ? count[p] - o : seg_len;
            segs[3 * ns] = p; segs[3 * ns + 1] = pos + o; segs[3 * ns + 2] = len;
            ++ns;
        }
        pos += count[p];
    }
    *n_segs = ns;
}
__global__ void k_bin_scatter(const int* __restrict__ pitch, const int* __restrict__ mel2ph, int n, int vocab, int* __restrict__ cursor, int* __restrict__ order) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int p;
    if (i < n && bin_ok(pitch, mel2ph, i, vocab, p)) order[atomicAdd(cursor + p, 1)] = i;
}
// S[bin][c] += sum over the segment's frames of src[row(frame)][c]; grid (max segments, ceil(C / 256)), 64 threads x 4 columns.  src = the fp16
// [hi | lo] row planes of dy (TEpiGateBwdT writes them for the transposed conv; round 5: the fp32 rows are no longer written): ld halfs per row,
// lo plane `lo` halfs in; hi + lo is dy to 2^-22
__global__ __launch_bounds__(256) void k_bin_sums(const _Float16* __restrict__ src, int ld, int lo, const int* __restrict__ order, const int* __restrict__ segs,
                                                  const int* __restrict__ n_segs, int T, int stride, float* __restrict__ S, int C) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    if ((int)blockIdx.x >= *n_segs) return;
    const int bin = segs[3 * blockIdx.x], first = segs[3 * blockIdx.x + 1], len = segs[3 * blockIdx.x + 2];
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;          // four waves walk every fourth frame of the segment (round 5: one wave, <= 32 dependent row loads)
    const int c = blockIdx.y * 256 + lane * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C)
        for (int i = sub; i < len; i += 4) {
            const int bt = order[first + i], b = bt / T, t = bt - b * T;
            const _Float16* p = src + ((size_t)b * stride + t) * ld + c;
            const h4 vh = *reinterpret_cast<const h4*>(p), vl = *reinterpret_cast<const h4*>(p + lo);
            s.x += (float)vh[0] + (float)vl[0]; s.y += (float)vh[1] + (float)vl[1]; s.z += (float)vh[2] + (float)vl[2]; s.w += (float)vh[3] + (float)vl[3];
        }
    __shared__ float4 red[4][64];
    red[sub][lane] = s;
    __syncthreads();
    if (sub || c >= C) return;
    s = red[0][lane];
#pragma unroll
    for (int q = 1; q < 4; ++q) { s.x += red[q][lane].x; s.y += red[q][lane].y; s.z += red[q][lane].z; s.w += red[q][lane].w; }
    float* d = S + (size_t)bin * C + c;
    atomicAdd(d, s.x); atomicAdd(d + 1, s.y); atomicAdd(d + 2, s.z); atomicAdd(d + 3, s.w);
}

__global__ void k_iota(int* p, int base, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = base + i;
}

// AdamW (torch.optim.AdamW semantics: decoupled weight decay, bias-corrected moments), gradients pre-scaled by gscale
__global__ void k_adamw(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                        float lr, float b1, float b2, float eps, float wd, float bc1, float bc2, const float* __restrict__ gscale_dev, float gscale) {
    const float gs = gscale_dev ? *gscale_dev : gscale;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gr = g[i] * gs;
        float w = p[i];
        w *= 1.0f - lr * wd;
        const float mi = b1 * m[i] + (1.0f - b1) * gr;
        const float vi = b2 * v[i] + (1.0f - b2) * gr * gr;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        w -= (lr / bc1) * mi / denom;
        p[i] = w;
    }
}

__global__ void k_sqsum(const float* __restrict__ g, size_t n, float* __restrict__ out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += g[i] * g[i];
    __shared__ float red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];            // per-block partials, summed in a fixed order by k_clip_coef: every rank of a
}                                                               // data-parallel job must derive the SAME coefficient from the same gradients

// clip coefficient of torch.nn.utils.clip_grad_norm_: min(1, max_norm / (norm + 1e-6)); one block of 256 threads
__global__ void k_clip_coef(const float* __restrict__ part, int n_part, float* __restrict__ sq, float max_norm, float* __restrict__ coef) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n_part; i += 256) s += part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x) return;
    *sq = red[0];
    const float nrm = sqrtf(*sq);
    const float c = max_norm / (nrm + 1e-6f);
    *coef = c < 1.0f ? c : 1.0f;
}

struct Packed {
    DevBuf w;
    int n_ctiles = 0, taps = 1, cin_pad = 0;
};

}  // namespace

// =================================================================================================
struct dsvc_trainer {
    dsvc_trainer_cfg cfg;
    std::vector<std::string> names;
    std::map<std::string, std::pair<int64_t, int64_t>> index;       // name -> (offset, numel) in the flat layout
    int64_t total = 0;
    float* params = nullptr;
    float* grads = nullptr;
    int* step_err = nullptr;                                        // host-mapped sticky flag: a step saw a diffusion step outside [0, timesteps)
    std::vector<float> h_sa, h_sb;
    DevBuf sa, sb, spec_min, spec_max;
    int n_spec = 0;

    // workspace for (B, T)
    int wsB = 0, wsT = 0, Tp = 0, rows = 0, nr = 0;                 // nr = B*Tp data rows; rows = nr rounded up to 128
    int ldT = 0, a_rows = 0, b_rows = 0, cp128 = 0, hp128 = 0;      // operand planes of the weight-gradient GEMMs (wgrad.h)
    float loss_scale = 1.0f;                                        // 2^k the backward pass is scaled by (see the header)
    DevBuf xt, xs, sig, tau, g, skip, ypre, s2pre, eps, deps, condT, tstep, clipid, iotaB;
    DevBuf e0, e1pre, e1, e2, filmB, dfilm, de2, de1, de1pre;
    DevBuf dx, dxin, dO, dy, ds2pre, dh0, loss;
    DevBuf bin_count, bin_cursor, bin_segs, bin_nsegs, bin_order, bin_S;     // frames sorted by pitch bin; S_l[vocab][2C] per layer
    DevBuf AT, BT;                                 // dY^T and X^T as channel-major fp16 hi|lo planes: [2][a_rows | b_rows][ldT]
    DevBuf wpart;                                  // frame-slice partial tiles of one weight-gradient GEMM
    // per-step repacked weights
    Packed w_in, w_skip, w_fin, w_finT, w_skipT;

    // round 4: the residual layers on the tgemm engine
    int Cp = 0, Hp = 0;                            // channels of a plane: C / H rounded up to 128
    static constexpr int TGUARD = 64;              // zero rows in front of and behind xhP (the dilated conv's halo at the first / last tile)
    DevBuf xhP, ghP, condHP;                       // fp16 [hi | lo] row planes: x^l + film_l [TGUARD + rows + TGUARD][2 Cp], g_l [rows][2 Cp], cond [rows][2 Hp]
    DevBuf gate_t, out_t, cproj_t;                 // fragment-ordered hi|lo weights of every layer (k_tpack_batch, per step)
    size_t gate_halfs = 0, out_halfs = 0, cproj_halfs = 0;     // per layer
    // ... and the two data-gradient GEMMs of the backward pass (streamed-K kernels: their 2C-channel split rows do not fit LDS whole)
    int C2p = 0;                                   // 2C rounded up to 128
    DevBuf dOh, dyh;                               // planes of dO [rows][2 C2p] and of dy [TGUARD + rows + TGUARD][2 C2p]
    DevBuf oT_t, dT_t;                             // W_o^T and the flipped W_d^T, rows = input channels of the layer
    size_t oT_halfs = 0, dT_halfs = 0;
    // round 5: the residual layers' weight gradients are contracted straight from these frame-major planes (wgrad.h: wgrad_fm_kernel) instead of
    // channel-major copies a k_split_t pass writes -- which needs layer l's x / g planes alive in the backward pass: every layer gets its own
    // (2 x 270 MB at the 64 x 128 batch).  fm = the architecture fits the kernel's tiles (C % 128, H % 128: the shipped configs; others keep k_split_t)
    bool fm = false;
    bool fm_off = false;                           // test support (dsvc_trainer_debug_set "wgrad_fm" 0): keep the k_split_t path where fm would apply
    size_t xh_layer = 0, gh_layer = 0;             // halfs between the layers' planes in xhP / ghP (0: one buffer all layers share)
    DevBuf wbias;                                  // partial column sums (bias gradients) of one wgrad_fm launch: [slices][k tiles][O_pad]
    struct FmProb {                                // one contraction: dW[o][k] = sum_n A[n][o] * B_seg[n + shift][k]; bias_dst (and bias_dst2) <- column sums of A
        const _Float16* a; int a_ld, a_lo, O;
        const WgradFmSeg* segs; int n_seg;
        WgradSegs rsegs; float scale;
        float* bias_dst; float* bias_dst2;
    };
    int wgrad_fm(const FmProb* probs, int n_prob, hipStream_t st);
    DevBuf t_gate_rm, t_gate_rs, t_out_rm;         // packed row -> source channel (+ the gate rows' pre-scale)
    std::vector<TPackDesc> tpack_q;
    DevBuf tpack_dev;
    std::vector<char> tpack_cached;
    template <class Epi, int KP = 0>
    int tg(const _Float16* x, int cin, int taps, int dil, const _Float16* w, int m_tiles, const typename Epi::Args& e, hipStream_t st, int kp_cin = 0);

    ~dsvc_trainer() {
        for (DevBuf* b : {&sa, &sb, &spec_min, &spec_max, &xt, &xs, &sig, &tau, &g, &skip, &ypre, &s2pre, &eps, &deps, &condT, &tstep,
                          &clipid, &iotaB, &e0, &e1pre, &e1, &e2, &filmB, &dfilm, &de2, &de1, &de1pre, &dx, &dxin, &dO, &dy, &ds2pre,
                          &dh0, &loss, &AT, &BT, &wpart, &pack_dev, &bin_count, &bin_cursor, &bin_segs, &bin_nsegs, &bin_order, &bin_S,
                          &xhP, &ghP, &condHP, &gate_t, &out_t, &cproj_t, &t_gate_rm, &t_gate_rs, &t_out_rm, &tpack_dev, &dOh, &dyh, &oT_t, &dT_t, &wbias})
            b->release();
        if (step_err) (void)hipHostFree(step_err);
        auto rel = [](Packed& p) { p.w.release(); };
        rel(w_in); rel(w_skip); rel(w_fin); rel(w_finT); rel(w_skipT);
    }

    void add(const std::string& n, int64_t numel) { names.push_back(n); index[n] = {total, numel}; total += numel; }
    float* P(const std::string& n) const { return params + index.at(n).first; }
    float* G(const std::string& n) const { return grads + index.at(n).first; }

    void layout();
    int ensure_ws(int B, int T, hipStream_t st);
    int pack(Packed& pk, const float* src, const int* colmap, int cout_pad, int taps, int cin, int cout, long long s_col, long long s_ci,
             long long s_tap, int flip, float scale, hipStream_t st);
    int repack(hipStream_t st);
    std::vector<PackDesc> pack_q;                  // the weight re-packs of a step, queued by pack() and launched together
    DevBuf pack_dev;
    std::vector<char> pack_cached;
    int flush_packs(hipStream_t st);
    // operand planes: rows [row0, row0 + C) of AT (a_side) or BT <- src[frame][0..C) (+ add[clip]) over the real frames; dil > 0: the three
    // taps of a dilated conv's input (frames t - dil, t, t + dil) into rows row0 + {0, 1, 2} * cp128
    // colsum != nullptr (dil == 0 only): colsum[c] += sum over the frames of src[.][c] -- the bias gradient rides on the pass that reads dY anyway
    int split_t(bool a_side, int row0, const float* src, int ld_src, int C, const float* add, int add_stride, int dil, hipStream_t st,
                float* colsum = nullptr);
    // dW[o][k] = sum_n AT[o][n] * BT[b_row0 + k][n] for o < O, k < K_pad; the k axis is cut into the segments of `segs`
    int wgrad_nt(int O, int K_pad, int b_row0, const WgradSegs& segs, float scale, hipStream_t st);
    // one training step in phases (so that a data-parallel host can all-reduce the gradients of finished layers while the backward pass of
    // the earlier ones still runs): PH_BEGIN = inputs, forward, loss, backward of the tail; PH_LAYERS = backward of residual layers
    // [l_lo, l_hi) from the top down, their gradients final (diffusion_projection included, loss scale removed); PH_END = input projection,
    // step-embedding MLP, pitch embedding.  Gradient slices that are final after each phase: dsvc.h.
    enum { PH_BEGIN = 1, PH_LAYERS = 2, PH_END = 4, PH_ALL = 7 };
    dsvc_train_args cur{};                                          // the arguments of the step in flight (phases after PH_BEGIN)
    int next_layer = -1;                                            // the layer PH_LAYERS continues from (-1: no step in flight)
    int run(int phases, int l_hi, int l_lo, const dsvc_train_args* a, float* loss_out, hipStream_t st);
    int step(const dsvc_train_args* a, float* loss_out, hipStream_t st) { return run(PH_ALL, cfg.layers, 0, a, loss_out, st); }
};

void dsvc_trainer::layout() {
    const int M = cfg.mel_bins, H = cfg.hidden, C = cfg.channels, L = cfg.layers;
    // the order of DiffNet.state_dict() (net.py:86-110), then the one fs2 parameter the path trains (fs2.py:73-79)
    add("denoise_fn.input_projection.weight", (int64_t)C * M); add("denoise_fn.input_projection.bias", C);
    add("denoise_fn.mlp.0.weight", (int64_t)4 * C * C); add("denoise_fn.mlp.0.bias", 4 * C);
    add("denoise_fn.mlp.2.weight", (int64_t)4 * C * C); add("denoise_fn.mlp.2.bias", C);
    for (int l = 0; l < L; ++l) {
        const std::string q = "denoise_fn.residual_layers." + std::to_string(l) + ".";
        add(q + "dilated_conv.weight", (int64_t)2 * C * C * 3); add(q + "dilated_conv.bias", 2 * C);
        add(q + "diffusion_projection.weight", (int64_t)C * C); add(q + "diffusion_projection.bias", C);
        add(q + "conditioner_projection.weight", (int64_t)2 * C * H); add(q + "conditioner_projection.bias", 2 * C);
        add(q + "output_projection.weight", (int64_t)2 * C * C); add(q + "output_projection.bias", 2 * C);
    }
    add("denoise_fn.skip_projection.weight", (int64_t)C * C); add("denoise_fn.skip_projection.bias", C);
    add("denoise_fn.output_projection.weight", (int64_t)M * C); add("denoise_fn.output_projection.bias", M);
    add("fs2.pitch_embed.weight", (int64_t)cfg.pitch_vocab * H);
}

int dsvc_trainer::pack(Packed& pk, const float* src, const int* colmap, int cout_pad, int taps, int cin, int cout, long long s_col,
                       long long s_ci, long long s_tap, int flip, float scale, hipStream_t st) {
    pk.n_ctiles = round_up(ceil_div(cout_pad, 32), 2);
    pk.taps = taps;
    pk.cin_pad = round_up(cin, 16);
    const size_t halfs = packed_halfs(pk.n_ctiles, taps, pk.cin_pad, 2);
    DSVC_TRY(pk.w.alloc(halfs * 2));
    (void)st;                                                    // queued: repack() launches every descriptor of the step at once
    pack_q.push_back(PackDesc{src, colmap, pk.w.as<_Float16>(), pk.n_ctiles, taps, pk.cin_pad, cout, cin, s_col, s_ci, s_tap, flip, scale});
    return DSVC_OK;
}

// the queued weight re-packs of a step as two launches; the descriptor tables only change when a buffer was (re)allocated
int dsvc_trainer::flush_packs(hipStream_t st) {
    auto upload = [&](DevBuf& dev, std::vector<char>& cached, const void* host, size_t bytes) -> int {
        if (cached.size() != bytes || memcmp(cached.data(), host, bytes) != 0) {
            DSVC_HIP(hipStreamSynchronize(st));                  // a launch reading the old table may still be in flight
            DSVC_TRY(dev.alloc(bytes));
            DSVC_HIP(hipMemcpy(dev.p, host, bytes, hipMemcpyHostToDevice));
            cached.assign((const char*)host, (const char*)host + bytes);
        }
        return DSVC_OK;
    };
    if (!pack_q.empty()) {
        DSVC_TRY(upload(pack_dev, pack_cached, pack_q.data(), pack_q.size() * sizeof(PackDesc)));
        hipLaunchKernelGGL(k_pack_w_batch, dim3(96, (unsigned)pack_q.size()), dim3(256), 0, st, pack_dev.as<PackDesc>());
    }
    if (!tpack_q.empty()) {
        DSVC_TRY(upload(tpack_dev, tpack_cached, tpack_q.data(), tpack_q.size() * sizeof(TPackDesc)));
        hipLaunchKernelGGL(k_tpack_batch, dim3(96, (unsigned)tpack_q.size()), dim3(256), 0, st, tpack_dev.as<TPackDesc>());
    }
    DSVC_HIP(hipGetLastError());
    return DSVC_OK;
}

// one tgemm launch of the training forward: 64-frame tiles x 8 waves, split activations (the batched tiling of DSVC_PREC_F16_X3T)
template <class Epi, int KP>
int dsvc_trainer::tg(const _Float16* x, int cin, int taps, int dil, const _Float16* w, int m_tiles, const typename Epi::Args& e, hipStream_t st, int kp_cin) {
    TGemmArgs a{};
    a.kp_cin = kp_cin;
    a.x = x; a.cin = cin; a.taps = taps; a.dil = dil; a.w = w; a.m_tiles = m_tiles; a.w_planes = 2; a.variant_halfs = 0; a.n_variants = 1;
    a.step_ptr = nullptr; a.step_off = 0;
    // the K loops start at staggered groups (clip_rows = 0: keyed on the tile index).  Measured on the 64 x 128 golden: with every tile summing in
    // the SAME order (clip_rows = 64) the most cancellation-prone gradients sit 7.8e-4 from the fp64 evaluation instead of 2.1e-4 -- the
    // accumulation's rounding is then the same function of the data in all 8704 rows and does not average out in the weight-gradient
    // contraction over the frames; the price is that a row's last bits depend on where its tile sits in the batch (profiles/r4_tfwd_tests.txt)
    a.clip_rows = 0;
    const int tiles = rows / 64, passes = ceil_div(m_tiles, 8);
    int ms = 512 / tiles; ms = ms < 1 ? 1 : (ms > passes ? passes : ms);
    // (9..12 output tiles -- C = 384 has 12 in the data-gradient GEMMs -- are a full pass of 8 waves plus a half-empty one, 2 x tiles workgroups;
    //  twelve waves doing all tiles in one pass, 136 workgroups instead of 272, measured slower: 10.70 against 10.45 ms per step; so did one
    //  workgroup per tile running both passes: 11.17)
    return tgemm_launch<2, 8, 2, 4, 2, Epi, 1, 1, 2, 0, KP>(a, e, rows, ms, st);
}

int dsvc_trainer::repack(hipStream_t st) {
    const int M = cfg.mel_bins, H = cfg.hidden, C = cfg.channels, L = cfg.layers;
    pack_q.clear(); tpack_q.clear();
    const float isl = 1.0f / sqrtf((float)L);
    // forward (natural [O][I][taps] sources)
    DSVC_TRY(pack(w_in, P("denoise_fn.input_projection.weight"), nullptr, C, 1, M, C, M, 1, 0, 0, 1.0f, st));
    DSVC_TRY(pack(w_skip, P("denoise_fn.skip_projection.weight"), nullptr, C, 1, C, C, C, 1, 0, 0, isl, st));        // sum(skip)/sqrt(L) folded (net.py:131)
    DSVC_TRY(pack(w_fin, P("denoise_fn.output_projection.weight"), nullptr, M, 1, C, M, C, 1, 0, 0, 1.0f, st));
    // transposed (data gradients): W^T(col = input channel, ci = output channel) = W[ci][col]
    DSVC_TRY(pack(w_finT, P("denoise_fn.output_projection.weight"), nullptr, C, 1, M, C, 1, C, 0, 0, 1.0f, st));
    DSVC_TRY(pack(w_skipT, P("denoise_fn.skip_projection.weight"), nullptr, C, 1, C, C, 1, C, 0, 0, isl, st));
    // The residual layers (round 4): every layer GEMM -- gate conv, output projection, the stacked conditioner projections, dg = W_o^T dO and the
    // transposed conv -- runs on the tgemm engine (tgemm.h) with fragment-ordered fp16 hi|lo weights: gate rows paired and pre-scaled (tepi_util.h),
    // the data gradients' operands transposed (packed row = input channel c of the layer, K = its 2C output channels).  Rounds 2-3 ran them on
    // conv_gemm.h / a plane-based engine: 96 / 62 / 27 / 63 / 77 us per layer against 64 / 39 / 12 / 48 / 77 now (profiles/r4z, r4u_kernel_stats_train.csv).
    const int mpl = C / 16;
    for (int l = 0; l < L; ++l) {
        const std::string q = "denoise_fn.residual_layers." + std::to_string(l) + ".";
        tpack_q.push_back(TPackDesc{P(q + "dilated_conv.weight"), t_gate_rm.as<int>(), t_gate_rs.as<float>(), gate_t.as<_Float16>() + (size_t)l * gate_halfs, C, 3, Cp, mpl,
                                    (long long)C * 3, 3, 1, 0});
        tpack_q.push_back(TPackDesc{P(q + "output_projection.weight"), t_out_rm.as<int>(), nullptr, out_t.as<_Float16>() + (size_t)l * out_halfs, C, 1, Cp, mpl, C, 1, 1, 0});
        tpack_q.push_back(TPackDesc{P(q + "conditioner_projection.weight"), t_gate_rm.as<int>(), t_gate_rs.as<float>(), cproj_t.as<_Float16>() + (size_t)l * cproj_halfs, H, 1, Hp, mpl,
                                    H, 1, 1, 0});
        // W_o^T(c, k) = W_o[k][c]
        tpack_q.push_back(TPackDesc{P(q + "output_projection.weight"), t_out_rm.as<int>(), nullptr, oT_t.as<_Float16>() + (size_t)l * oT_halfs, 2 * C, 1, C2p, C / 32, 1, C, 1, 0});
        // transposed conv: dxin[c] = sum_tap sum_o W_d[o][c][2 - tap] * dy[row + (tap-1)*d][o]
        tpack_q.push_back(TPackDesc{P(q + "dilated_conv.weight"), t_out_rm.as<int>(), nullptr, dT_t.as<_Float16>() + (size_t)l * dT_halfs, 2 * C, 3, C2p, C / 32, 3, (long long)C * 3, 1, 1});
    }
    return flush_packs(st);
}

// The workspace layout depends on (B, T): clips sit Tp rows apart with ZERO gap rows (the convs' padding, and rows the weight-gradient
// contractions sum over), so a new shape re-zeroes it -- ~1.3 GB of memset (~0.4 ms) at the 64 x 128-frame batch; allocations are kept
// and only grow.  A loader that changes B and T every step (the reference's max_tokens batching) pays that per step: bucket T
// (e.g. multiples of 32 frames) to keep the layout stable.
int dsvc_trainer::ensure_ws(int B, int T, hipStream_t st) {
    if (B == wsB && T == wsT) return DSVC_OK;
    const int M = cfg.mel_bins, H = cfg.hidden, C = cfg.channels, L = cfg.layers;
    int max_dil = 1;
    for (int l = 0; l < L; ++l) { const int d = 1 << (l % cfg.dilation_cycle); if (d > max_dil) max_dil = d; }
    // (the dilated conv and its transpose keep a 64-frame tile + both halos in LDS: whole up to dilation 16, as two 128-channel K phases up to 32)
    if (max_dil > 32) return fail(DSVC_EINVAL, "trainer: dilation %d too large (dilation_cycle_length <= 6 is supported)", max_dil);
    Tp = round_up((T + max_dil < 32 ? 32 : T + max_dil), 8);                       // gap >= the largest dilation (the convs' zero padding); every gap row is work for the conv kernels
    nr = B * Tp;
    rows = round_up(nr, 128);                   // the contraction length of the weight-gradient GEMMs: a multiple of the staged chunk
    const size_t r = (size_t)rows;
    auto z = [&](DevBuf& b, size_t bytes) -> int { DSVC_TRY(b.alloc(bytes)); DSVC_HIP(hipMemsetAsync(b.p, 0, bytes, st)); return DSVC_OK; };
    // the five per-layer activation stores are 95 % of the workspace: when their allocation is re-used under a new (B, T), only the rows the
    // step itself never writes are zeroed (k_zero_gap_rows); a fresh allocation is cleared whole
    auto zg = [&](DevBuf& b, int ld, int slabs) -> int {
        const size_t bytes = r * (size_t)ld * 4 * slabs;
        const bool reused = b.p && bytes <= b.bytes;
        DSVC_TRY(b.alloc(bytes));
        if (!reused) { DSVC_HIP(hipMemsetAsync(b.p, 0, bytes, st)); return DSVC_OK; }
        const long long quads = (long long)(B * (Tp - T) + (rows - nr)) * (ld / 4);
        if (quads > 0) hipLaunchKernelGGL(k_zero_gap_rows, dim3((unsigned)ceil_div((int)quads, 256), slabs), dim3(256), 0, st, b.as<float>(), (long long)r * ld, ld, B, Tp, T, rows);
        DSVC_HIP(hipGetLastError());
        return DSVC_OK;
    };
    // (round 5) fm: the residual layers' weight gradients are contracted from the layers' fp16 operand planes (wgrad.h: wgrad_fm_kernel) -- then
    // g and dy are never stored as fp32 rows and their buffers (0.3 GB at the 64 x 128 batch) are not allocated
    fm = !fm_off && C % 128 == 0 && (2 * C) % 256 == 0 && H % 128 == 0 && max_dil <= TGUARD;
    if (fm) {   //  the per-layer operand planes fm keeps for the backward pass (L x rows x 2 Cp halfs, x and g: 2 x 270 MB at 64 x 128 frames,
        // growing with the frames per step) must fit beside everything else: if the device cannot hold them, take the k_split_t path instead of failing
        const size_t cp = (size_t)round_up(C, 128), want = (size_t)L * ((r + 2 * TGUARD) + r) * 2 * cp * 2;
        const size_t have_planes = xhP.bytes + ghP.bytes;
        size_t free_b = 0, total_b = 0;
        if (want > have_planes && hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b < (want - have_planes) + ((size_t)2 << 30)) fm = false;
    }
    DSVC_TRY(z(xt, r * M * 4)); DSVC_TRY(zg(xs, C, L + 1)); DSVC_TRY(zg(sig, C, L)); DSVC_TRY(zg(tau, C, L));
    if (!fm) DSVC_TRY(zg(g, C, L));
    DSVC_TRY(z(skip, r * C * 4)); DSVC_TRY(zg(ypre, round_up(2 * C, 128), L)); DSVC_TRY(z(s2pre, r * C * 4));
    DSVC_TRY(z(eps, r * M * 4)); DSVC_TRY(z(deps, r * M * 4)); DSVC_TRY(z(condT, r * H * 4));
    DSVC_TRY(z(tstep, (size_t)B * 4)); DSVC_TRY(z(clipid, (size_t)B * 4)); DSVC_TRY(z(iotaB, (size_t)B * 4));
    DSVC_TRY(z(e0, (size_t)B * C * 4)); DSVC_TRY(z(e1pre, (size_t)B * 4 * C * 4)); DSVC_TRY(z(e1, (size_t)B * 4 * C * 4)); DSVC_TRY(z(e2, (size_t)B * C * 4));
    DSVC_TRY(z(filmB, (size_t)B * L * C * 4)); DSVC_TRY(z(dfilm, (size_t)B * L * C * 4)); DSVC_TRY(z(de2, (size_t)B * C * 4));
    DSVC_TRY(z(de1, (size_t)B * 4 * C * 4)); DSVC_TRY(z(de1pre, (size_t)B * 4 * C * 4));
    DSVC_TRY(z(dx, r * C * 4)); DSVC_TRY(z(dxin, r * C * 4)); DSVC_TRY(z(dO, r * 2 * C * 4));
    if (!fm) DSVC_TRY(z(dy, r * 2 * C * 4));
    DSVC_TRY(z(ds2pre, r * C * 4)); DSVC_TRY(z(dh0, r * C * 4)); DSVC_TRY(z(loss, 16));
    {   // pitch-bin bookkeeping (k_bin_*): at most ceil(B T / 32) + vocab segments of <= 32 frames
        const int V = cfg.pitch_vocab, max_segs = ceil_div(B * T, 32) + V;
        DSVC_TRY(z(bin_count, (size_t)V * 4)); DSVC_TRY(z(bin_cursor, (size_t)V * 4)); DSVC_TRY(z(bin_segs, (size_t)max_segs * 12));
        DSVC_TRY(z(bin_nsegs, 16)); DSVC_TRY(z(bin_order, (size_t)B * T * 4)); DSVC_TRY(z(bin_S, (size_t)L * V * 2 * C * 4));
    }
    {   // operands of the residual layers' GEMMs on the tgemm engine (channels % 64 == 0, hidden % 16 == 0, dilations <= TGUARD: checked at create)
        Cp = round_up(C, 128); Hp = round_up(H, 128);
        const int mpl = C / 16;
        // (every row of [0, rows) of these planes is rewritten by the producing epilogue each step, zeros on gap rows included; the guard rows
        //  are what has to be zero, and a layout change moves them)
        xh_layer = fm ? (r + 2 * TGUARD) * 2 * Cp : 0; gh_layer = fm ? r * 2 * Cp : 0;
        const size_t nl = fm ? (size_t)L : 1;
        {   // a re-used allocation only needs its guard rows cleared (they move with the layout); a fresh one is cleared whole
            const size_t row_b = (size_t)2 * Cp * 2, layer_b = (r + 2 * TGUARD) * row_b, guard_b = (size_t)TGUARD * row_b;
            const void* was = xhP.p;
            const bool reused = xhP.p && nl * layer_b <= xhP.bytes;
            DSVC_TRY(xhP.alloc(nl * layer_b));
            if (!reused || xhP.p != was) DSVC_HIP(hipMemsetAsync(xhP.p, 0, nl * layer_b, st));
            else {
                DSVC_HIP(hipMemset2DAsync(xhP.p, layer_b, 0, guard_b, nl, st));
                DSVC_HIP(hipMemset2DAsync((char*)xhP.p + guard_b + r * row_b, layer_b, 0, guard_b, nl, st));
            }
            const bool g_reused = ghP.p && nl * r * row_b <= ghP.bytes;
            DSVC_TRY(ghP.alloc(nl * r * row_b));
            if (!g_reused) DSVC_HIP(hipMemsetAsync(ghP.p, 0, nl * r * row_b, st));       // (no guard rows: every row of [0, rows) is rewritten each step)
        }
        DSVC_TRY(z(condHP, r * 2 * Hp * 2));
        if (fm) DSVC_TRY(wbias.alloc((size_t)WGRAD_MAX_TILES * 256 * 4));
        gate_halfs = tpacked_halfs(mpl, 3, Cp, 2, 1); out_halfs = tpacked_halfs(mpl, 1, Cp, 2, 1); cproj_halfs = tpacked_halfs(mpl, 1, Hp, 2, 1);
        DSVC_TRY(gate_t.alloc(gate_halfs * L * 2)); DSVC_TRY(out_t.alloc(out_halfs * L * 2)); DSVC_TRY(cproj_t.alloc(cproj_halfs * L * 2));
        C2p = round_up(2 * C, 128);
        DSVC_TRY(z(dOh, r * 2 * C2p * 2)); DSVC_TRY(z(dyh, (r + 2 * TGUARD) * 2 * C2p * 2));
        oT_halfs = tpacked_halfs(C / 32, 1, C2p, 2, 1); dT_halfs = tpacked_halfs(C / 32, 3, C2p, 2, 1);
        DSVC_TRY(oT_t.alloc(oT_halfs * L * 2)); DSVC_TRY(dT_t.alloc(dT_halfs * L * 2));
        if (!t_gate_rm.p) {
            std::vector<int> grm(mpl * 32), orm(mpl * 32);
            std::vector<float> grs(mpl * 32);
            for (int q = 0; q < mpl * 32; ++q) {
                const int mt = q >> 5, i = q & 31;
                grm[q] = (i >> 4) * C + mt * 16 + trow_to_ch8(i & 15);
                grs[q] = i < 16 ? GATE_SCALE : FILT_SCALE;
                orm[q] = mt * 32 + trow_to_ch16(i);
            }
            DSVC_TRY(t_gate_rm.alloc(grm.size() * 4)); DSVC_TRY(t_gate_rs.alloc(grs.size() * 4)); DSVC_TRY(t_out_rm.alloc(orm.size() * 4));
            DSVC_HIP(hipMemcpyAsync(t_gate_rm.p, grm.data(), grm.size() * 4, hipMemcpyHostToDevice, st));
            DSVC_HIP(hipMemcpyAsync(t_gate_rs.p, grs.data(), grs.size() * 4, hipMemcpyHostToDevice, st));
            DSVC_HIP(hipMemcpyAsync(t_out_rm.p, orm.data(), orm.size() * 4, hipMemcpyHostToDevice, st));
            DSVC_HIP(hipStreamSynchronize(st));
        }
    }
    // weight-gradient operands (wgrad.h): frames contiguous, zero beyond the data rows and in the channel padding
    ldT = round_up(B * T, 128);                                  // real frames only (no gap rows); whole 32-frame stages, 64-frame split tiles
    cp128 = round_up(C, 128); hp128 = round_up(H, 128);
    a_rows = round_up(2 * C > M ? 2 * C : M, 256);
    b_rows = 3 * cp128 + hp128;                                  // [tap 0 | tap 1 | tap 2 | cond] of a layer's dilated conv + conditioner projection
    if (b_rows < round_up(M, 128)) b_rows = round_up(M, 128);
    DSVC_TRY(z(AT, (size_t)2 * a_rows * ldT * 2));
    DSVC_TRY(z(BT, (size_t)2 * b_rows * ldT * 2));
    DSVC_TRY(wpart.alloc((size_t)WGRAD_MAX_TILES * 256 * 128 * 4));
    {   // loss scale: the largest power of two that keeps |d loss / d eps| * scale <= 2^-6 for the l1 loss (the l2 gradient is 2 d times that)
        const double inv_n = 1.0 / ((double)B * M * T);
        int k = 0;
        while (k < 24 && inv_n * ldexp(1.0, k + 1) <= 1.0 / 64.0) ++k;
        loss_scale = (float)ldexp(1.0, k);
    }
    hipLaunchKernelGGL(k_iota, dim3(ceil_div(B, 256)), dim3(256), 0, st, iotaB.as<int>(), 0, B);
    wsB = B; wsT = T;
    return DSVC_OK;
}

int dsvc_trainer::split_t(bool a_side, int row0, const float* src, int ld_src, int C, const float* add, int add_stride, int dil, hipStream_t st,
                          float* colsum) {
    DevBuf& buf = a_side ? AT : BT;
    const int nrows = a_side ? a_rows : b_rows;
    if (row0 < 0 || row0 + (dil > 0 ? 2 * cp128 : 0) + C > nrows || dil > 64) return fail(DSVC_EINVAL, "split_t: rows [%d, %d) outside the %d-row operand planes", row0, row0 + C, nrows);
    const long long plane = (long long)nrows * ldT;
    hipLaunchKernelGGL(k_split_t, dim3(ldT / 64, ceil_div(C, 64)), dim3(256), 0, st, src, ld_src, buf.as<_Float16>() + (size_t)row0 * ldT, plane, ldT, C,
                       add, add_stride, SplitRows{Tp, wsT, wsB}, dil > 0 ? 3 : 1, dil, (long long)cp128 * ldT, 1.0f, dil > 0 ? nullptr : colsum);
    DSVC_HIP(hipGetLastError());
    return DSVC_OK;
}

int dsvc_trainer::wgrad_nt(int O, int K_pad, int b_row0, const WgradSegs& segs, float scale, hipStream_t st) {
    const int O_pad = round_up(O, 256);
    if (O_pad > a_rows || b_row0 + K_pad > b_rows || K_pad % 128) return fail(DSVC_EINVAL, "wgrad_nt: %d x %d outside the operand planes", O_pad, K_pad);
    const int tiles = (O_pad / 256) * (K_pad / 128);
    // a handful of output tiles does not fill 256 CUs: the frame range is cut into slices (one workgroup per tile and slice).  Slices come in
    // multiples of 8 and every XCD gets whole slices, as many as fit its 32 CUs -- the tiles of a slice walk the same frames of the same
    // operand rows, so they are served by ONE L2 (with a plain grid every XCD's L2 pulled every operand byte: 4.7 TB/s through the fabric
    // at 440 TFLOP/s, profiles/r3n_kernel_stats_train.csv)
    int per_xcd = tiles <= 32 ? 32 / tiles : 1;
    int S = 8 * per_xcd, xcd_map = 1;
    if (S > ldT / 128) { S = ldT / 128 < 1 ? 1 : ldT / 128; xcd_map = S % 8 == 0; }      // at least four 32-frame stages per slice
    if ((long long)S * tiles > WGRAD_MAX_TILES) return fail(DSVC_EINVAL, "wgrad_nt: %d slices x %d output tiles exceed the partial-tile scratch", S, tiles);
    const int slice_len = round_up(ceil_div(ldT, S), 32);
    WgradNtArgs a{};
    a.at = AT.as<_Float16>(); a.bt = BT.as<_Float16>() + (size_t)b_row0 * ldT;
    a.a_plane = (long long)a_rows * ldT; a.b_plane = (long long)b_rows * ldT;
    a.ldT = ldT; a.n_total = ldT; a.slice_len = slice_len;
    a.part = wpart.as<float>(); a.O_pad = O_pad; a.K_pad = K_pad; a.tiles = tiles; a.xcd_map = xcd_map;
    static bool attr_set = false;
    if (!attr_set) {
        DSVC_HIP(hipFuncSetAttribute((const void*)wgrad_nt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WG_STAGES * WG_STAGE_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL(wgrad_nt_kernel, dim3(tiles * S), dim3(512), WG_STAGES * WG_STAGE_BYTES, st, a);
    DSVC_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_wgrad_nt_reduce, dim3(ceil_div(K_pad, 1024), O), dim3(256), 0, st, wpart.as<float>(), S, O_pad, K_pad, O, segs, scale);
    DSVC_HIP(hipGetLastError());
    return DSVC_OK;
}

// One or two contractions dW[o][k] = sum_n A[n][o] * B_seg[n + shift][k] in one launch (wgrad.h: wgrad_fm_kernel), the k axis = the segments'
// 128-column tiles in order; bias_dst != nullptr: + the column sums of A.  Slices, XCD placement and the fixed-order reduction as wgrad_nt.
// Two contractions share the launch when their output tiles fit an XCD together (the output projection's 9 and the conditioner projection's 6):
// 2 x 8 slices each and 240 workgroups instead of 24 / 40 thin slices in two launches that each leave the chip waiting for its last workgroups.
int dsvc_trainer::wgrad_fm(const FmProb* probs, int n_prob, hipStream_t st) {
    if (n_prob < 1 || n_prob > 2 || rows % 32) return fail(DSVC_EINVAL, "wgrad_fm: %d contractions over %d rows", n_prob, rows);
    int tiles[2] = {0, 0}, kts[2] = {0, 0};
    for (int i = 0; i < n_prob; ++i) {
        if (probs[i].O % 256 || probs[i].n_seg < 1 || probs[i].n_seg > 4) return fail(DSVC_EINVAL, "wgrad_fm: %d output rows, %d segments", probs[i].O, probs[i].n_seg);
        for (int s = 0; s < probs[i].n_seg; ++s) kts[i] += probs[i].segs[s].k_tiles;
        tiles[i] = (probs[i].O / 256) * kts[i];
    }
    if (n_prob == 2 && tiles[0] + tiles[1] > 32) {          // no room for a slice of each on an XCD: two launches
        DSVC_TRY(wgrad_fm(probs, 1, st));
        return wgrad_fm(probs + 1, 1, st);
    }
    // gap rows are zero in every plane: with whole 32-row stages per clip they are skipped
    const int spc = wsT % 32 == 0 ? wsT / 32 : 0;
    const int n_stages = spc ? wsB * spc : rows / 32;
    const int tsum = tiles[0] + tiles[1];
    int per_xcd = tsum <= 32 ? 32 / tsum : 1;
    int S = 8 * per_xcd, xcd_map = 1;
    if (S > n_stages / 4) {                                 // at least four stages per slice
        if (n_prob == 2) { DSVC_TRY(wgrad_fm(probs, 1, st)); return wgrad_fm(probs + 1, 1, st); }
        S = n_stages / 4 < 1 ? 1 : n_stages / 4; xcd_map = S % 8 == 0;
    }
    if ((long long)S * tsum > WGRAD_MAX_TILES) return fail(DSVC_EINVAL, "wgrad_fm: %d slices x %d output tiles exceed the partial-tile scratch", S, tsum);
    WgradFmArgs w{};
    w.n_prob = n_prob; w.wgs0 = S * tiles[0]; w.n_stages = n_stages; w.clip_rows = Tp; w.spc = spc; w.xcd_map = xcd_map;
    float* part = wpart.as<float>();
    float* bpart = wbias.as<float>();
    bool any_bias = false;
    for (int i = 0; i < n_prob; ++i) {
        WgradFmProb& q = w.p[i];
        q.a = probs[i].a; q.a_ld = probs[i].a_ld; q.a_lo = probs[i].a_lo; q.n_seg = probs[i].n_seg;
        for (int s = 0; s < probs[i].n_seg; ++s) q.seg[s] = probs[i].segs[s];
        q.slices = S; q.slice_stages = ceil_div(n_stages, S);
        q.part = part; q.O_pad = probs[i].O; q.K_pad = kts[i] * 128; q.tiles = tiles[i];
        q.bias_part = probs[i].bias_dst ? bpart : nullptr;
        any_bias = any_bias || probs[i].bias_dst;
        part += (size_t)S * tiles[i] * 256 * 128;
        bpart += (size_t)S * tiles[i] * 256;
    }
    static bool attr_set = false;
    if (!attr_set) {
        DSVC_HIP(hipFuncSetAttribute((const void*)wgrad_fm_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, WG_STAGES * WG_STAGE_BYTES));
        DSVC_HIP(hipFuncSetAttribute((const void*)wgrad_fm_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, WG_STAGES * WG_STAGE_BYTES));
        attr_set = true;
    }
    if (any_bias) hipLaunchKernelGGL(wgrad_fm_kernel<true>, dim3(S * tsum), dim3(512), WG_STAGES * WG_STAGE_BYTES, st, w);
    else hipLaunchKernelGGL(wgrad_fm_kernel<false>, dim3(S * tsum), dim3(512), WG_STAGES * WG_STAGE_BYTES, st, w);
    DSVC_HIP(hipGetLastError());
    for (int i = 0; i < n_prob; ++i) {      // (the bias column sums are reduced by one more column of blocks of the same launch)
        const WgradFmProb& q = w.p[i];
        hipLaunchKernelGGL(k_wgrad_nt_reduce, dim3(ceil_div(q.K_pad, 1024) + (q.bias_part ? 1 : 0), q.O_pad), dim3(256), 0, st, q.part, S, q.O_pad, q.K_pad,
                           q.O_pad, probs[i].rsegs, probs[i].scale, (const float*)q.bias_part, S * kts[i], probs[i].bias_dst, probs[i].bias_dst2);
    }
    DSVC_HIP(hipGetLastError());
    return DSVC_OK;
}

int dsvc_trainer::run(int phases, int l_hi, int l_lo, const dsvc_train_args* ta_in, float* loss_out, hipStream_t st) {
    if (phases & PH_BEGIN) {
        if (!ta_in) return fail(DSVC_EINVAL, "trainer: null step arguments");
        cur = *ta_in;
    } else if (next_layer < 0) return fail(DSVC_ESTATE, "trainer: no step in flight (call dsvc_trainer_step_begin first)");
    const dsvc_train_args* ta = &cur;
    const int M = cfg.mel_bins, H = cfg.hidden, C = cfg.channels, L = cfg.layers, B = ta->B, T = ta->T;
    if (phases & PH_LAYERS) {
        if (l_lo < 0 || l_hi > L || l_lo >= l_hi) return fail(DSVC_EINVAL, "trainer: layer range [%d, %d) of %d", l_lo, l_hi, L);
        if (!(phases & PH_BEGIN) && l_hi != next_layer) return fail(DSVC_ESTATE, "trainer: layers must be walked from the top down (expected %d, got %d)", next_layer, l_hi);
    }
    if ((phases & PH_END) && !(phases & PH_LAYERS) && next_layer != 0) return fail(DSVC_ESTATE, "trainer: %d residual layers still to go", next_layer);
    if (phases & PH_BEGIN) {
        DSVC_TRY(ensure_ws(B, T, st));
        DSVC_TRY(repack(st));
    }
    const RowInfo ri{Tp, T, nr};
    const size_t r = (size_t)rows, slab = r * C;
    const float unscale = 1.0f / loss_scale;
    auto unscale_range = [&](const std::string& first, const std::string& last) {       // the loss scale leaves the gradients of [first, last]
        const int64_t o0 = index.at(first).first, o1 = index.at(last).first + index.at(last).second;
        if (loss_scale != 1.0f) hipLaunchKernelGGL(k_scale_inplace, dim3(512), dim3(256), 0, st, grads + o0, (size_t)(o1 - o0), unscale);
    };
    auto base = [&](const float* x, int ldx, int cin, const Packed& pk, int dil) {
        ConvGemmArgs a{};
        a.x = x; a.ldx = ldx; a.n_rows = nr; a.clip_stride = Tp; a.clip_len = T;
        a.cin = cin; a.taps = pk.taps; a.dil = dil; a.w = pk.w.as<_Float16>(); a.n_ctiles = pk.n_ctiles; a.w_planes = 2; a.in_slope = 1.0f;
        return a;
    };
    auto small_b = [&](const SmallBatch& sb, int Mm, int N, int K, int lda, int ldb, int ldc, int tA, int tB, int acc) {
        if (!tA && tB && !acc && K % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0) {       // Linear forward: both operands k-contiguous
            hipLaunchKernelGGL(k_gemm_nt_dot, dim3(ceil_div(N, 16), ceil_div(Mm, 4), sb.n), dim3(256), 0, st, sb, Mm, N, K, lda, ldb, ldc);
            return;
        }
        hipLaunchKernelGGL(k_gemm_small_b, dim3(ceil_div(N, 64), ceil_div(Mm, 64), sb.n), dim3(256), 0, st, sb, Mm, N, K, lda, ldb, ldc, tA, tB, acc);
    };
    auto small = [&](const float* A, const float* Bm, float* Cm, int Mm, int N, int K, int lda, int ldb, int ldc, int tA, int tB, int acc,
                     const float* bias) {
        SmallBatch sb{};
        sb.n = 1; sb.A[0] = A; sb.B[0] = Bm; sb.C[0] = Cm; sb.bias[0] = bias;
        small_b(sb, Mm, N, K, lda, ldb, ldc, tA, tB, acc);
    };
    const int ew = 2048;
    const bool batched_small = L <= 32 && (size_t)L * B * C * 4 <= wpart.bytes;
    auto seg1 = [&](float* dst, int K, long long stride_o) {
        WgradSegs sg{};
        sg.n = 1; sg.s[0] = WgradSeg{dst, 0, K, stride_o, 1, 0};
        return sg;
    };
  if (phases & PH_BEGIN) {
    DSVC_HIP(hipMemsetAsync(grads, 0, (size_t)total * 4, st));
    DSVC_HIP(hipMemsetAsync(loss.p, 0, 16, st));
    // ---- inputs ----
    // the diffusion steps index the noise schedule (k_make_xt) and the step embedding: clamped into [0, timesteps) on the way in
    if (!step_err) {
        DSVC_HIP(hipHostMalloc(reinterpret_cast<void**>(&step_err), sizeof(int), hipHostMallocMapped));
        *step_err = 0;
    }
    hipLaunchKernelGGL(k_clamp_copy, dim3(ceil_div(B, 256)), dim3(256), 0, st, tstep.as<int>(), ta->t, 0, cfg.timesteps - 1, B, step_err);
    if (ta->clip_ids) DSVC_HIP(hipMemcpyAsync(clipid.p, ta->clip_ids, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
    else hipLaunchKernelGGL(k_iota, dim3(ceil_div(B, 256)), dim3(256), 0, st, clipid.as<int>(), ta->first_clip, B);
    hipLaunchKernelGGL(k_make_xt, dim3(ceil_div(T * M / 4, 256), B), dim3(256), 0, st, ta->mel, xt.as<float>(), tstep.as<int>(), sa.as<float>(),
                       sb.as<float>(), spec_min.as<float>(), spec_max.as<float>(), n_spec, B, T, M, Tp, ta->seed, clipid.as<int>());
    hipLaunchKernelGGL(k_bct_to_rows, dim3(ceil_div(T, 32), ceil_div(H, 32), B), dim3(256), 0, st, ta->cond, condT.as<float>(), B, H, T, Tp);
    // ---- step embedding: emb -> Linear -> Mish -> Linear -> per-layer diffusion_projection  (net.py:99-103,124-125,67) ----
    hipLaunchKernelGGL(k_sin_emb_b, dim3(ceil_div(B * C, 256)), dim3(256), 0, st, e0.as<float>(), tstep.as<int>(), B, C);
    small(e0.as<float>(), P("denoise_fn.mlp.0.weight"), e1pre.as<float>(), B, 4 * C, C, C, C, 4 * C, 0, 1, 0, P("denoise_fn.mlp.0.bias"));
    hipLaunchKernelGGL(k_mish, dim3(ceil_div(B * 4 * C, 256)), dim3(256), 0, st, e1pre.as<float>(), e1.as<float>(), (size_t)B * 4 * C);
    small(e1.as<float>(), P("denoise_fn.mlp.2.weight"), e2.as<float>(), B, C, 4 * C, 4 * C, 4 * C, C, 0, 1, 0, P("denoise_fn.mlp.2.bias"));
    if (batched_small) {   // film_l = e2 W_l^T + b_l for all layers in one launch
        SmallBatch sb{};
        sb.n = L;
        for (int l = 0; l < L; ++l) {
            const std::string q = "denoise_fn.residual_layers." + std::to_string(l) + ".diffusion_projection.";
            sb.A[l] = e2.as<float>(); sb.B[l] = P(q + "weight"); sb.C[l] = filmB.as<float>() + (size_t)l * C; sb.bias[l] = P(q + "bias");
        }
        small_b(sb, B, C, C, C, C, L * C, 0, 1, 0);
    } else for (int l = 0; l < L; ++l) {
        const std::string q = "denoise_fn.residual_layers." + std::to_string(l) + ".diffusion_projection.";
        small(e2.as<float>(), P(q + "weight"), filmB.as<float>() + (size_t)l * C, B, C, C, C, C, L * C, 0, 1, 0, P(q + "bias"));
    }
    // ---- forward ----
    {   // x^0 = relu(W_in x_t + b)  (net.py:120-123)
        ConvGemmArgs a = base(xt.as<float>(), M, M, w_in, 1);
        EpStore::Args e{xs.as<float>(), C, P("denoise_fn.input_projection.bias"), C, 1, ri};
        DSVC_TRY(launch<EpStore>(a, e, st));
    }
    {
        // the residual layers on the tgemm engine (epilogues above): operands as fp16 [hi | lo] row planes, every layer's stores as fp32 rows
        const int mpl = C / 16;
        _Float16* xh0 = xhP.as<_Float16>() + (size_t)TGUARD * 2 * Cp;      // layer l's planes: xh0 + l * xh_layer (fm: kept for the weight gradients)
        const size_t tslab = r * 2 * C;                                    // one layer of the accumulator-tiled conditioner projection (in ypre)
        const RowInfo ri_all{Tp, Tp, nr};
        hipLaunchKernelGGL(k_rows_to_planes, dim3(2048), dim3(256), 0, st, xs.as<float>(), C, C, filmB.as<float>(), L * C, xh0, Cp, ri, rows);
        hipLaunchKernelGGL(k_rows_to_planes, dim3(2048), dim3(256), 0, st, condT.as<float>(), H, H, (const float*)nullptr, 0, condHP.as<_Float16>(), Hp, ri_all, rows);
        {
            const std::string q0 = "denoise_fn.residual_layers.0.", q1 = "denoise_fn.residual_layers.1.";
            const long long lstride = L > 1 ? index.at(q1 + "dilated_conv.bias").first - index.at(q0 + "dilated_conv.bias").first : 0;
            TEpiCprojT::Args e{ypre.as<float>(), (long long)tslab, mpl, P(q0 + "dilated_conv.bias"), P(q0 + "conditioner_projection.bias"), lstride, C};
            DSVC_TRY(tg<TEpiCprojT>(condHP.as<_Float16>(), Hp, 1, 1, cproj_t.as<_Float16>(), L * mpl, e, st));
        }
        for (int l = 0; l < L; ++l) {
            const std::string q = "denoise_fn.residual_layers." + std::to_string(l) + ".";
            const int d = 1 << (l % cfg.dilation_cycle);
            float* xl = xs.as<float>() + (size_t)l * slab;
            _Float16* xh = xh0 + (size_t)l * xh_layer;
            _Float16* gh = ghP.as<_Float16>() + (size_t)l * gh_layer;
            {
                TEpiGateT::Args e{ypre.as<float>() + (size_t)l * tslab, sig.as<float>() + (size_t)l * slab, tau.as<float>() + (size_t)l * slab,
                                  fm ? nullptr : g.as<float>() + (size_t)l * slab, gh, C, Cp, ri};
                if (tgemm_smem<2>(3, d, 2 * Cp) <= 160 * 1024) DSVC_TRY(tg<TEpiGateT>(xh, Cp, 3, d, gate_t.as<_Float16>() + (size_t)l * gate_halfs, mpl, e, st));
                else DSVC_TRY((tg<TEpiGateT, 1>(xh, Cp, 3, d, gate_t.as<_Float16>() + (size_t)l * gate_halfs, mpl, e, st, 128)));      // wide halos: streamed K
            }
            {
                TEpiResSkipT::Args e{xl, xs.as<float>() + (size_t)(l + 1) * slab, skip.as<float>(), l + 1 < L ? xh + xh_layer : nullptr, P(q + "output_projection.bias"),
                                     filmB.as<float>() + (size_t)(l + 1 < L ? l + 1 : l) * C, L * C, C, Cp, l == 0 ? 1 : 0, ri};
                DSVC_TRY(tg<TEpiResSkipT>(gh, Cp, 1, 1, out_t.as<_Float16>() + (size_t)l * out_halfs, mpl, e, st));
            }
        }
    }
    {   // s2pre = W_s skip/sqrt(L) + b ;  eps = W_out relu(s2pre) + b   (net.py:131-134)
        ConvGemmArgs a = base(skip.as<float>(), C, C, w_skip, 1);
        EpStore::Args e{s2pre.as<float>(), C, P("denoise_fn.skip_projection.bias"), C, 0, ri};
        DSVC_TRY(launch<EpStore>(a, e, st));
        ConvGemmArgs a2 = base(s2pre.as<float>(), C, C, w_fin, 1);
        a2.in_slope = 0.0f;                                   // leaky_relu with slope 0 at staging = ReLU
        EpStore::Args e2s{eps.as<float>(), M, P("denoise_fn.output_projection.bias"), M, 0, ri};
        DSVC_TRY(launch<EpStore>(a2, e2s, st));
    }
    // ---- loss and d eps ----
    const float inv_n = 1.0f / ((float)B * (float)M * (float)T);
    hipLaunchKernelGGL(k_loss, dim3(ceil_div(T * M / 4, 256), B), dim3(256), 0, st, eps.as<float>(), deps.as<float>(), loss.as<float>(), B, T, M, Tp,
                       ta->seed, clipid.as<int>(), cfg.loss_l1, inv_n, loss_scale);
    // ---- backward: tail ----
    {   // dW_out[m][c] = sum_n deps[n][m] relu(s2pre)[n][c]
        hipLaunchKernelGGL(k_relu_bwd, dim3(ew), dim3(256), 0, st, s2pre.as<float>(), s2pre.as<float>(), dh0.as<float>(), r * C);   // dh0 = relu(s2pre) (scratch)
        DSVC_TRY(split_t(true, 0, deps.as<float>(), M, M, nullptr, 0, 0, st, G("denoise_fn.output_projection.bias")));
        DSVC_TRY(split_t(false, 0, dh0.as<float>(), C, C, nullptr, 0, 0, st));
        DSVC_TRY(wgrad_nt(M, cp128, 0, seg1(G("denoise_fn.output_projection.weight"), C, C), 1.0f, st));
        // d s2pre = (W_out^T deps) * [s2pre > 0]
        ConvGemmArgs a = base(deps.as<float>(), M, M, w_finT, 1);
        EpBwd::Args e{ds2pre.as<float>(), C, C, s2pre.as<float>(), C, 1.0f, 0, ri};
        DSVC_TRY(launch<EpBwd>(a, e, st));
        // dW_s[o][c] = sum_n ds2pre[n][o] skip[n][c] / sqrt(L)
        DSVC_TRY(split_t(true, 0, ds2pre.as<float>(), C, C, nullptr, 0, 0, st, G("denoise_fn.skip_projection.bias")));
        DSVC_TRY(split_t(false, 0, skip.as<float>(), C, C, nullptr, 0, 0, st));
        DSVC_TRY(wgrad_nt(C, cp128, 0, seg1(G("denoise_fn.skip_projection.weight"), C, C), 1.0f / sqrtf((float)L), st));
        // dskip = W_s^T ds2pre / sqrt(L)  -> the skip half of dO (the same for every layer)
        ConvGemmArgs a2 = base(ds2pre.as<float>(), C, C, w_skipT, 1);
        EpBwd::Args e2{dO.as<float>() + C, 2 * C, C, nullptr, 0, 1.0f, 0, ri};
        DSVC_TRY(launch<EpBwd>(a2, e2, st));
    }
    DSVC_HIP(hipMemsetAsync(dx.p, 0, r * C * 4, st));                                                                 // d x^L = 0: the loss sees x only through skip
    hipLaunchKernelGGL(k_copy_cols, dim3(ew), dim3(256), 0, st, dx.as<float>(), dO.as<float>(), nr, C, 2 * C, 0, 0.0f);   // ... so the residual half of dO starts at 0
    // dO as the operand planes of the top layer's dg = W_o^T dO (the layers' epilogues refresh the residual half; the skip half is the same for all)
        hipLaunchKernelGGL(k_rows_to_planes, dim3(2048), dim3(256), 0, st, dO.as<float>(), 2 * C, 2 * C, (const float*)nullptr, 0, dOh.as<_Float16>(), C2p, ri, rows);
    // cond^T planes once (weight gradients of every conditioner projection): the last segment of the layers' k axis, which nothing else writes
    if (!fm) DSVC_TRY(split_t(false, 3 * cp128, condT.as<float>(), H, H, nullptr, 0, 0, st));      // (fm reads the frame-major planes of cond the forward pass left)
    if (ta->pitch) {   // frames by pitch bin, once per step (the pitch-embedding gradient: k_bin_sums per layer, one product at the end)
        const int V = cfg.pitch_vocab, n = B * T;
        DSVC_HIP(hipMemsetAsync(bin_count.p, 0, (size_t)V * 4, st));
        DSVC_HIP(hipMemsetAsync(bin_S.p, 0, (size_t)L * V * 2 * C * 4, st));
        hipLaunchKernelGGL(k_bin_count, dim3(ceil_div(n, 256)), dim3(256), 0, st, ta->pitch, ta->mel2ph, n, V, bin_count.as<int>());
        hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(64), 0, st, bin_count.as<int>(), V, bin_cursor.as<int>(), bin_segs.as<int>(), bin_nsegs.as<int>(), 32);
        hipLaunchKernelGGL(k_bin_scatter, dim3(ceil_div(n, 256)), dim3(256), 0, st, ta->pitch, ta->mel2ph, n, V, bin_cursor.as<int>(), bin_order.as<int>());
    }
    unscale_range("denoise_fn.skip_projection.weight", "denoise_fn.output_projection.bias");      // final: the tail's four tensors
    next_layer = L;
  }
  if (phases & PH_LAYERS) {
    // ---- backward: layers ----
    for (int l = l_hi - 1; l >= l_lo; --l) {
        const std::string q = "denoise_fn.residual_layers." + std::to_string(l) + ".";
        const int d = 1 << (l % cfg.dilation_cycle);
        const float* xl = xs.as<float>() + (size_t)l * slab;
        const float* gl = fm ? nullptr : g.as<float>() + (size_t)l * slab;
        if (!fm) {
            DSVC_TRY(split_t(true, 0, dO.as<float>(), 2 * C, 2 * C, nullptr, 0, 0, st, G(q + "output_projection.bias")));
            DSVC_TRY(split_t(false, 0, gl, C, C, nullptr, 0, 0, st));
            DSVC_TRY(wgrad_nt(2 * C, cp128, 0, seg1(G(q + "output_projection.weight"), C, C), 1.0f, st));
        }   // (fm: dW_o shares a launch with the conditioner projection's gradient below -- dOh is intact until this layer's TEpiDxT)
        {   // dg = W_o^T dO -> dy, K = 2C streamed in phases of 256 (128 where 2C is not a multiple of 256) channels
            TEpiGateBwdT::Args e{sig.as<float>() + (size_t)l * slab, tau.as<float>() + (size_t)l * slab, fm ? nullptr : dy.as<float>(),
                                 dyh.as<_Float16>() + (size_t)TGUARD * 2 * C2p, C, C2p, ri};
            DSVC_TRY((tg<TEpiGateBwdT, 1>(dOh.as<_Float16>(), C2p, 1, 1, oT_t.as<_Float16>() + (size_t)l * oT_halfs, C / 32, e, st, C2p % 256 == 0 ? 256 : 128)));
        }
        // dW_d[o][c][tap] = sum_n dy[n][o] (x^l + film)[n + (tap-1) d][c]  and  dW_c[o][h] = sum_n dy[n][o] cond[n][h]  share dy^T: ONE
        // contraction over the k axis [tap 0 | tap 1 | tap 2 | cond]
        WgradSegs sg{};
        sg.n = 3;
        for (int tap = 0; tap < 3; ++tap) sg.s[tap] = WgradSeg{G(q + "dilated_conv.weight"), tap * cp128, C, (long long)C * 3, 3, tap};
        if (fm) {   // the three taps are row offsets of ONE copy of (x^l + film_l): the planes the forward gate conv read, guard rows included
            const _Float16* xh = xhP.as<_Float16>() + (size_t)TGUARD * 2 * Cp + (size_t)l * xh_layer;
            const _Float16* dyp = dyh.as<_Float16>() + (size_t)TGUARD * 2 * C2p;
            WgradFmSeg taps[3];
            for (int tap = 0; tap < 3; ++tap) taps[tap] = WgradFmSeg{xh, 2 * Cp, Cp, (tap - 1) * d, C / 128};
            const FmProb conv{dyp, 2 * C2p, C2p, 2 * C, taps, 3, sg, 1.0f, G(q + "dilated_conv.bias"), G(q + "conditioner_projection.bias")};
            DSVC_TRY(wgrad_fm(&conv, 1, st));
            // dW_o[o][c] = sum_n dO[n][o] g[n][c] (the planes dg = W_o^T dO read and the forward pass left) and dW_c[o][h] = sum_n dy[n][o] cond[n][h]: one launch
            const WgradFmSeg gs{ghP.as<_Float16>() + (size_t)l * gh_layer, 2 * Cp, Cp, 0, C / 128}, cs{condHP.as<_Float16>(), 2 * Hp, Hp, 0, H / 128};
            const FmProb two[2] = {{dOh.as<_Float16>(), 2 * C2p, C2p, 2 * C, &gs, 1, seg1(G(q + "output_projection.weight"), C, C), 1.0f, G(q + "output_projection.bias"), nullptr},
                                   {dyp, 2 * C2p, C2p, 2 * C, &cs, 1, seg1(G(q + "conditioner_projection.weight"), H, H), 1.0f, nullptr, nullptr}};
            DSVC_TRY(wgrad_fm(two, 2, st));
        } else {
            DSVC_TRY(split_t(true, 0, dy.as<float>(), 2 * C, 2 * C, nullptr, 0, 0, st, G(q + "dilated_conv.bias")));
            DSVC_HIP(hipMemcpyAsync(G(q + "conditioner_projection.bias"), G(q + "dilated_conv.bias"), (size_t)2 * C * 4, hipMemcpyDeviceToDevice, st));
            DSVC_TRY(split_t(false, 0, xl, C, C, filmB.as<float>() + (size_t)l * C, L * C, d, st));
            DSVC_TRY(wgrad_nt(2 * C, 3 * cp128, 0, sg, 1.0f, st));
            // (one launch over [taps | cond] is 33 tiles at C = 384: one more than an XCD has CUs)
            DSVC_TRY(wgrad_nt(2 * C, hp128, 3 * cp128, seg1(G(q + "conditioner_projection.weight"), H, H), 1.0f, st));
        }
        if (ta->pitch)     // S_l[bin] = sum of this layer's dy rows per pitch bin (instead of dcond += W_c^T dy: see k_bin_sums)
            hipLaunchKernelGGL(k_bin_sums, dim3(ceil_div(B * T, 32) + cfg.pitch_vocab, ceil_div(2 * C, 256)), dim3(256), 0, st,
                               (const _Float16*)(dyh.as<_Float16>() + (size_t)TGUARD * 2 * C2p), 2 * C2p, C2p, bin_order.as<int>(), bin_segs.as<int>(), bin_nsegs.as<int>(), T, Tp,
                               bin_S.as<float>() + (size_t)l * cfg.pitch_vocab * 2 * C, 2 * C);
        {   // dxin = convT(dy); dx <- dx / sqrt 2 + dxin; the residual half of dO (rows and planes) <- dx / sqrt 2
            TEpiDxT::Args e{dx.as<float>(), dxin.as<float>(), fm ? nullptr : dO.as<float>(), dOh.as<_Float16>(), C, C2p, ri};
            DSVC_TRY((tg<TEpiDxT, 1>(dyh.as<_Float16>() + (size_t)TGUARD * 2 * C2p, C2p, 3, d, dT_t.as<_Float16>() + (size_t)l * dT_halfs, C / 32, e, st, (C2p % 256 == 0 && (64 + 2 * d) * 2048 <= 160 * 1024) ? 256 : 128)));      // (two phase buffers of 64 + 2d rows in LDS)
        }
        hipLaunchKernelGGL(k_clip_colsum, dim3(ceil_div(C, 128), B), dim3(256), 0, st, dxin.as<float>(), dfilm.as<float>() + (size_t)l * C, T, C, Tp, L * C);
    }
    // the step-embedding side of these layers: d diffusion_projection from dfilm_l (the FiLM gradient) -- with them the layers' gradients are final
    if (batched_small) {
        SmallBatch dw{};
        dw.n = l_hi - l_lo;
        for (int l = l_lo; l < l_hi; ++l) {
            const std::string q = "denoise_fn.residual_layers." + std::to_string(l) + ".diffusion_projection.";
            const float* df = dfilm.as<float>() + (size_t)l * C;
            dw.A[l - l_lo] = df; dw.B[l - l_lo] = e2.as<float>(); dw.C[l - l_lo] = G(q + "weight");     // dWp[o][i] = sum_b dfilm[b][o] e2[b][i]
        }
        {   // dbp_l = sum_b dfilm_l[b], all layers of the range in one launch (the layers' parameters sit a constant stride apart)
            const std::string b0 = "denoise_fn.residual_layers." + std::to_string(l_lo) + ".diffusion_projection.bias";
            const long long lstride = l_hi - l_lo > 1 ? index.at("denoise_fn.residual_layers." + std::to_string(l_lo + 1) + ".diffusion_projection.bias").first - index.at(b0).first : 0;
            hipLaunchKernelGGL(k_colsum, dim3(ceil_div(C, 256), 1, l_hi - l_lo), dim3(256), 0, st, dfilm.as<float>() + (size_t)l_lo * C, G(b0), B, C, L * C, B,
                               (long long)C, lstride);
        }
        small_b(dw, C, C, B, L * C, C, C, 1, 0, 0);
    } else for (int l = l_lo; l < l_hi; ++l) {
        const std::string q = "denoise_fn.residual_layers." + std::to_string(l) + ".diffusion_projection.";
        const float* df = dfilm.as<float>() + (size_t)l * C;
        small(df, e2.as<float>(), G(q + "weight"), C, C, B, L * C, C, C, 1, 0, 0, nullptr);
        hipLaunchKernelGGL(k_colsum, dim3(ceil_div(C, 256), 1), dim3(256), 0, st, df, G(q + "bias"), B, C, L * C, B);
    }
    unscale_range("denoise_fn.residual_layers." + std::to_string(l_lo) + ".dilated_conv.weight",
                  "denoise_fn.residual_layers." + std::to_string(l_hi - 1) + ".output_projection.bias");
    next_layer = l_lo;
  }
  if (phases & PH_END) {
    {   // input projection: d h0pre = dx^0 [x^0 > 0]
        hipLaunchKernelGGL(k_relu_bwd, dim3(ew), dim3(256), 0, st, dx.as<float>(), xs.as<float>(), dh0.as<float>(), r * C);
        DSVC_TRY(split_t(true, 0, dh0.as<float>(), C, C, nullptr, 0, 0, st, G("denoise_fn.input_projection.bias")));
        DSVC_TRY(split_t(false, 0, xt.as<float>(), M, M, nullptr, 0, 0, st));
        DSVC_TRY(wgrad_nt(C, round_up(M, 128), 0, seg1(G("denoise_fn.input_projection.weight"), M, M), 1.0f, st));
    }
    // ---- backward: step embedding:  de2[b][i] = sum_l sum_o dfilm_l[b][o] Wp_l[o][i] ----
    DSVC_HIP(hipMemsetAsync(de2.p, 0, (size_t)B * C * 4, st));
    if (batched_small) {
        SmallBatch dx2{};
        dx2.n = L;
        for (int l = 0; l < L; ++l) {
            const std::string q = "denoise_fn.residual_layers." + std::to_string(l) + ".diffusion_projection.";
            dx2.A[l] = dfilm.as<float>() + (size_t)l * C; dx2.B[l] = P(q + "weight");
            dx2.C[l] = wpart.as<float>() + (size_t)l * B * C;       // per-layer partials into the (idle) weight-gradient scratch, then a fixed-order sum
        }
        small_b(dx2, B, C, C, L * C, C, C, 0, 0, 0);
        hipLaunchKernelGGL(k_wgrad_reduce, dim3(ceil_div(B * C, 256)), dim3(256), 0, st, wpart.as<float>(), de2.as<float>(), L, B, C, C, 1LL, (long long)C, 0LL);
    } else for (int l = 0; l < L; ++l) {
        const std::string q = "denoise_fn.residual_layers." + std::to_string(l) + ".diffusion_projection.";
        small(dfilm.as<float>() + (size_t)l * C, P(q + "weight"), de2.as<float>(), B, C, C, L * C, C, C, 0, 0, 1, nullptr);
    }
    small(de2.as<float>(), e1.as<float>(), G("denoise_fn.mlp.2.weight"), C, 4 * C, B, C, 4 * C, 4 * C, 1, 0, 0, nullptr);
    hipLaunchKernelGGL(k_colsum, dim3(ceil_div(C, 256), 1), dim3(256), 0, st, de2.as<float>(), G("denoise_fn.mlp.2.bias"), B, C, C, B);
    small(de2.as<float>(), P("denoise_fn.mlp.2.weight"), de1.as<float>(), B, 4 * C, C, C, 4 * C, 4 * C, 0, 0, 0, nullptr);
    hipLaunchKernelGGL(k_mish_bwd, dim3(ceil_div(B * 4 * C, 256)), dim3(256), 0, st, e1pre.as<float>(), de1.as<float>(), de1pre.as<float>(), (size_t)B * 4 * C);
    small(de1pre.as<float>(), e0.as<float>(), G("denoise_fn.mlp.0.weight"), 4 * C, C, B, 4 * C, C, C, 1, 0, 0, nullptr);
    hipLaunchKernelGGL(k_colsum, dim3(ceil_div(4 * C, 256), 1), dim3(256), 0, st, de1pre.as<float>(), G("denoise_fn.mlp.0.bias"), B, 4 * C, 4 * C, B);
    // ---- pitch embedding (through cond) ----
    if (ta->pitch) {   // d pitch_embed[p][h] = sum_l sum_o S_l[p][o] W_c,l[o][h]: per-layer partials into the idle scratch, then a fixed-order sum
        const int V = cfg.pitch_vocab;
        if (L > 32 || (size_t)L * V * H * 4 > wpart.bytes) return fail(DSVC_EINVAL, "trainer: %d layers x %d pitch bins exceed the scratch", L, V);
        SmallBatch pe{};
        pe.n = L;
        for (int l = 0; l < L; ++l) {
            pe.A[l] = bin_S.as<float>() + (size_t)l * V * 2 * C;
            pe.B[l] = P("denoise_fn.residual_layers." + std::to_string(l) + ".conditioner_projection.weight");
            pe.C[l] = wpart.as<float>() + (size_t)l * V * H;
        }
        small_b(pe, V, H, 2 * C, 2 * C, H, H, 0, 0, 0);
        hipLaunchKernelGGL(k_wgrad_reduce, dim3(ceil_div(V * H, 256)), dim3(256), 0, st, wpart.as<float>(), G("fs2.pitch_embed.weight"), L, V, H, H, 1LL,
                           (long long)H, 0LL);
    }
    unscale_range("denoise_fn.input_projection.weight", "denoise_fn.mlp.2.bias");
    unscale_range("fs2.pitch_embed.weight", "fs2.pitch_embed.weight");
    if (loss_out) DSVC_HIP(hipMemcpyAsync(loss_out, loss.p, 4, hipMemcpyDeviceToDevice, st));
    next_layer = -1;
  }
    DSVC_HIP(hipGetLastError());
    return DSVC_OK;
}

// =================================================================================================
extern "C" {

int dsvc_trainer_create(const dsvc_trainer_cfg* cfg, dsvc_trainer** out) {
    if (!cfg || !out) return fail(DSVC_EINVAL, "null argument");
    if (cfg->mel_bins % 16 || cfg->hidden % 16 || cfg->channels % 64) return fail(DSVC_EINVAL, "trainer: need mel_bins%%16==0, hidden%%16==0, channels%%64==0");
    if (cfg->layers < 1 || cfg->dilation_cycle < 1 || cfg->pitch_vocab < 2) return fail(DSVC_EINVAL, "trainer: bad configuration");
    int ndev = 0;
    DSVC_HIP(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(DSVC_EHIP, "no HIP device visible");
    dsvc_trainer* t = new dsvc_trainer();
    t->cfg = *cfg;
    t->layout();
    *out = t;
    return DSVC_OK;
}

void dsvc_trainer_destroy(dsvc_trainer* t) { delete t; }

int dsvc_trainer_param_count(const dsvc_trainer* t, int64_t* n_tensors, int64_t* n_floats) {
    if (!t) return fail(DSVC_EINVAL, "null handle");
    if (n_tensors) *n_tensors = (int64_t)t->names.size();
    if (n_floats) *n_floats = t->total;
    return DSVC_OK;
}

int dsvc_trainer_param_info(const dsvc_trainer* t, int64_t i, const char** name, int64_t* offset, int64_t* numel) {
    if (!t || i < 0 || i >= (int64_t)t->names.size()) return fail(DSVC_EINVAL, "bad parameter index");
    const auto& e = t->index.at(t->names[(size_t)i]);
    if (name) *name = t->names[(size_t)i].c_str();
    if (offset) *offset = e.first;
    if (numel) *numel = e.second;
    return DSVC_OK;
}

int dsvc_trainer_bind(dsvc_trainer* t, float* params, float* grads) {
    if (!t || !params || !grads) return fail(DSVC_EINVAL, "null argument");
    t->params = params; t->grads = grads;
    return DSVC_OK;
}

int dsvc_trainer_set_schedule(dsvc_trainer* t, const float* sqrt_ac, const float* sqrt_1mac, int32_t K, const float* spec_min,
                              const float* spec_max, int32_t n_spec) {
    if (!t || !sqrt_ac || !sqrt_1mac || !spec_min || !spec_max || K < 1) return fail(DSVC_EINVAL, "null argument");
    if (n_spec != 1 && n_spec != t->cfg.mel_bins) return fail(DSVC_EINVAL, "trainer: spec_min/spec_max must have 1 or mel_bins entries");
    DSVC_TRY(t->sa.alloc((size_t)K * 4)); DSVC_TRY(t->sb.alloc((size_t)K * 4));
    DSVC_TRY(t->spec_min.alloc((size_t)n_spec * 4)); DSVC_TRY(t->spec_max.alloc((size_t)n_spec * 4));
    DSVC_HIP(hipMemcpy(t->sa.p, sqrt_ac, (size_t)K * 4, hipMemcpyHostToDevice));
    DSVC_HIP(hipMemcpy(t->sb.p, sqrt_1mac, (size_t)K * 4, hipMemcpyHostToDevice));
    DSVC_HIP(hipMemcpy(t->spec_min.p, spec_min, (size_t)n_spec * 4, hipMemcpyHostToDevice));
    DSVC_HIP(hipMemcpy(t->spec_max.p, spec_max, (size_t)n_spec * 4, hipMemcpyHostToDevice));
    t->n_spec = n_spec;
    t->cfg.timesteps = K;
    return DSVC_OK;
}

int dsvc_trainer_check(dsvc_trainer* t, void* stream) {
    if (!t) return fail(DSVC_EINVAL, "null handle");
    DSVC_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (t->step_err && *t->step_err) {
        *t->step_err = 0;
        return fail(DSVC_EINVAL, "a training step was given a diffusion step outside [0, %d): it ran at the clamped step", t->cfg.timesteps);
    }
    return DSVC_OK;
}

int dsvc_trainer_debug_set(dsvc_trainer* t, const char* key, int32_t value) {
    if (!t || !key) return fail(DSVC_EINVAL, "null argument");
    if (t->next_layer >= 0) return fail(DSVC_ESTATE, "trainer: a step is in flight");
#if defined(DSVC_TEST_HOOKS) || defined(DSVC_PROFILING)       // (the test-hooks build only: see dsvc_denoiser_debug_set)
    if (std::string(key) == "wgrad_fm") { t->fm_off = value == 0; t->wsB = t->wsT = 0; return DSVC_OK; }      // (the next step lays the workspace out again)
#endif
    (void)value;
    return fail(DSVC_EINVAL, "trainer: unknown debug key '%s' (test hooks live in libdsvc_hip_hooks.so, not in the product library)", key);
}

int dsvc_trainer_step_begin(dsvc_trainer* t, const dsvc_train_args* a, void* stream) {
    if (!t || !a) return fail(DSVC_EINVAL, "null argument");
    if (!t->params || !t->grads || !t->sa.p) return fail(DSVC_ESTATE, "trainer: bind the parameter buffers and set the schedule first");
    if (a->B < 1 || a->T < 1 || !a->mel || !a->cond || !a->t || (a->T * t->cfg.mel_bins) % 4) return fail(DSVC_EINVAL, "trainer: bad step arguments");
    return t->run(dsvc_trainer::PH_BEGIN, 0, 0, a, nullptr, (hipStream_t)stream);
}

int dsvc_trainer_step_layers(dsvc_trainer* t, int32_t l_hi, int32_t l_lo, void* stream) {
    if (!t) return fail(DSVC_EINVAL, "null argument");
    return t->run(dsvc_trainer::PH_LAYERS, l_hi, l_lo, nullptr, nullptr, (hipStream_t)stream);
}

int dsvc_trainer_step_end(dsvc_trainer* t, float* loss_out, void* stream) {
    if (!t) return fail(DSVC_EINVAL, "null argument");
    return t->run(dsvc_trainer::PH_END, 0, 0, nullptr, loss_out, (hipStream_t)stream);
}

int dsvc_trainer_step(dsvc_trainer* t, const dsvc_train_args* a, float* loss_out, void* stream) {
    if (!t || !a || !a->mel || !a->cond || !a->t) return fail(DSVC_EINVAL, "null argument");
    if (!t->params || !t->grads) return fail(DSVC_ESTATE, "trainer: bind the flat parameter / gradient buffers first");
    if (!t->sa.p) return fail(DSVC_ESTATE, "trainer: set the noise schedule first");
    if (a->B < 1 || a->T < 1 || (a->T * t->cfg.mel_bins) % 4) return fail(DSVC_EINVAL, "trainer: bad batch geometry");
    return t->step(a, loss_out, (hipStream_t)stream);
}

int dsvc_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int64_t step, const float* grad_scale_dev, float grad_scale, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || n < 0 || step < 1) return fail(DSVC_EINVAL, "bad argument");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_adamw, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, (size_t)n, lr, beta1,
                       beta2, eps, weight_decay, bc1, bc2, grad_scale_dev, grad_scale);
    DSVC_HIP(hipGetLastError());
    return DSVC_OK;
}

int dsvc_grad_clip_coef(const float* grads, int64_t n, float max_norm, float* sqnorm_dev, float* coef_dev, void* stream) {
    if (!grads || !sqnorm_dev || !coef_dev || n < 0) return fail(DSVC_EINVAL, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (blocks < 1) blocks = 1;
    float* part = nullptr;                                       // stream-ordered scratch for the per-block partial sums
    DSVC_HIP(hipMallocAsync((void**)&part, (size_t)blocks * 4, st));
    hipLaunchKernelGGL(k_sqsum, dim3(blocks), dim3(256), 0, st, grads, (size_t)n, part);
    hipLaunchKernelGGL(k_clip_coef, dim3(1), dim3(256), 0, st, part, blocks, sqnorm_dev, max_norm, coef_dev);
    DSVC_HIP(hipGetLastError());
    DSVC_HIP(hipFreeAsync(part, st));
    return DSVC_OK;
}

}  // extern "C"
