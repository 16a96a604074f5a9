// DiffNet + sampler device code: conv_gemm epilogues and the small HBM-bound kernels around them.
// Reference math: network/diff/net.py:58-135 (DiffNet / ResidualBlock), network/diff/diffusion.py:131-198
// (p_sample / p_sample_plms).  Everything is frame-major: row = clip*clip_stride + t.
#pragma once
#include "conv_gemm.h"

namespace dsvc {

// which diffusion step a row belongs to: one shared scalar, or one int per clip (DiffNet.forward's t[B])
struct StepRef {
    const int* ptr;      // device
    int off;             // step = ptr[clip*per_clip] - off
    int per_clip;        // 0: shared scalar, 1: one entry per clip
    __device__ __forceinline__ int get(int clip) const { return ptr[per_clip ? clip : 0] - off; }
};

__device__ __forceinline__ float fast_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) {
    // tanh(x) = 2*sigmoid(2x) - 1 ; exact enough (<= 2 ulp of the fp32 result range used here)
    const float e = __expf(-2.0f * fabsf(x));
    const float t = (1.0f - e) / (1.0f + e);
    return copysignf(t, x);
}

// ---- K4+K5+K6: gate.  tile 2g = gate half, tile 2g+1 = filter half of the same 32 channels ----
struct EpiGate {
    static constexpr bool PAIRED = true;
    struct Args {
        const float* cproj;   // [rows][2C] in packed column order: conditioner(cond) + both biases
        float* g;             // [rows][C]
        int C;
    };
    __device__ __forceinline__ void pair(const Args& e, int row, int ct0, int j, float vg, float vf) const {
        const float* cp = e.cproj + (size_t)row * (2 * e.C) + ct0 * 32 + j;
        const float gate = vg + cp[0];
        const float filt = vf + cp[32];
        e.g[(size_t)row * e.C + (ct0 >> 1) * 32 + j] = fast_sigmoid(gate) * fast_tanh(filt);
    }
};

// ---- K7+K8: output 1x1: residual half updates x, skip half accumulates ----
struct EpiResSkip {
    static constexpr bool PAIRED = false;
    struct Args {
        float* x;             // [rows][C] residual stream (in/out)
        float* skip;          // [rows][C] running skip sum
        const float* bias;    // [2C]
        int C;
        int first;            // layer 0: skip = s  (no read)
    };
    __device__ __forceinline__ void one(const Args& e, int row, int col, float v) const {
        v += e.bias[col];
        if (col < e.C) {
            float* p = e.x + (size_t)row * e.C + col;
            *p = (*p + v) * 0.70710678118654752440f;
        } else {
            float* p = e.skip + (size_t)row * e.C + (col - e.C);
            *p = e.first ? v : (*p + v);
        }
    }
};

// ---- K1 / K9a: bias + ReLU ----
struct EpiBiasRelu {
    static constexpr bool PAIRED = false;
    struct Args { float* out; int ld; const float* bias; int cout; };
    __device__ __forceinline__ void one(const Args& e, int row, int col, float v) const {
        if (col < e.cout) e.out[(size_t)row * e.ld + col] = fmaxf(v + e.bias[col], 0.f);
    }
};

// ---- plain bias store (hoisted conditioner projection, eps for PLMS / denoiser_forward) ----
struct EpiBias {
    static constexpr bool PAIRED = false;
    struct Args { float* out; int ld; const float* bias; int cout; };
    __device__ __forceinline__ void one(const Args& e, int row, int col, float v) const {
        if (col < e.cout) e.out[(size_t)row * e.ld + col] = v + e.bias[col];
    }
};

// ---- hoisted conditioner projection for the tgemm path: same math as EpiBias, written in the accumulator-tiled layout
//      the gate kernel initialises its accumulators from (diffnet_t.h: tiled_off; column p = 32*block + 16*half + 8*h + r')
struct EpiBiasTiled {
    static constexpr bool PAIRED = false;
    struct Args { float* out; int n_mtiles; const float* bias; int cout; };
    __device__ __forceinline__ void one(const Args& e, int row, int col, float v) const {
        if (col >= e.cout) return;
        const int mt = col >> 5, w = col & 31;
        const int h = (w >> 3) & 1, r = 8 * (w >> 4) + (w & 7);
        e.out[(((size_t)(row >> 5) * e.n_mtiles + mt) * 4 + (r >> 2)) * 256 + (size_t)((row & 31) + 32 * h) * 4 + (r & 3)] = v + e.bias[col];
    }
};

// ---- K9b+K10: final 1x1 fused with the DDPM posterior step (diffusion.py:131-163) ----
struct DdpmTables {
    const float* sqrt_recip_ac;     // sqrt_recip_alphas_cumprod
    const float* sqrt_recipm1_ac;   // sqrt_recipm1_alphas_cumprod
    const float* coef1;             // posterior_mean_coef1
    const float* coef2;             // posterior_mean_coef2
    const float* sigma;             // exp(0.5 * posterior_log_variance_clipped)
};

struct EpiDdpm {
    static constexpr bool PAIRED = false;
    struct Args {
        float* x;                   // [rows][M] sampler state (in/out)
        const float* bias;          // [M]
        int M;
        DdpmTables tab;
        StepRef step;
        int clip_stride;
        const int* lens;                   // device [B]: valid frames per clip
        const unsigned long long* seedp;   // device: the Philox seed and the clip ids live in memory so that a captured
        const int* clipid;                 // graph serves every call (they change per call, the graph does not); [B]
    };
    __device__ __forceinline__ void one(const Args& e, int row, int col, float v) const {
        if (col >= e.M) return;
        const int clip = row / e.clip_stride;
        const int tl = row - clip * e.clip_stride;
        if (tl >= e.lens[clip]) return;
        const int t = e.step.get(clip);
        const float eps = v + e.bias[col];
        float* px = e.x + (size_t)row * e.M + col;
        const float xt = *px;
        float x0 = e.tab.sqrt_recip_ac[t] * xt - e.tab.sqrt_recipm1_ac[t] * eps;
        x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
        float out = e.tab.coef1[t] * x0 + e.tab.coef2[t] * xt;
        if (t > 0) {
            const unsigned el = (unsigned)tl * (unsigned)e.M + (unsigned)col;
            const float z = philox_normal_lane(el >> 2, (unsigned)t, (unsigned)e.clipid[clip], PURPOSE_DDPM_NOISE, *e.seedp, el & 3);
            out += e.tab.sigma[t] * z;
        }
        *px = out;
    }
};

// ---------------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------------

// [B, C, T] (reference layout, T contiguous)  ->  frame-major [B*stride][C]; gap rows untouched
__global__ void k_to_frame_major(const float* __restrict__ src, float* __restrict__ dst, int B, int C, int T, int stride, float scale) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, t = t0 + tx;
        tile[i][tx] = (c < C && t < T) ? src[((size_t)b * C + c) * T + t] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, c = c0 + tx;
        if (t < T && c < C) dst[((size_t)b * stride + t) * C + c] = tile[tx][i] * scale;
    }
}

// frame-major [B*stride][C] -> [B, C, T]
__global__ void k_from_frame_major(const float* __restrict__ src, float* __restrict__ dst, int B, int C, int T, int stride) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, c = c0 + tx;
        tile[i][tx] = (c < C && t < T) ? src[((size_t)b * stride + t) * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, t = t0 + tx;
        if (t < T && c < C) dst[((size_t)b * C + c) * T + t] = tile[tx][i];
    }
}

// x_T ~ N(0,1) straight into the frame-major state (diffusion.py:265-268 with our Philox stream)
__global__ void k_x_init(float* __restrict__ x, int B, int T, int M, int stride, unsigned long long seed, const int* __restrict__ clipid) {
    const int quads = T * M / 4;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (q >= quads) return;
    float z[4];
    philox_normal4((unsigned)q, 0u, (unsigned)clipid[b], PURPOSE_X_INIT, seed, z);
    float* p = x + (size_t)b * stride * M + (size_t)q * 4;       // rows of a clip are contiguous: t*M+m == q*4
    *reinterpret_cast<float4*>(p) = make_float4(z[0], z[1], z[2], z[3]);
}

// K12: denorm + mask + layout: state [B*stride][M] -> mel_out [B][T][M]   (diffusion.py:279-290)
__global__ void k_finish_mel(const float* __restrict__ x, float* __restrict__ mel, const int* __restrict__ mel2ph,
                             const int* __restrict__ lens, const float* __restrict__ spec_min, const float* __restrict__ spec_max, int n_spec,
                             int B, int T, int M, int stride) {
    const size_t n = (size_t)B * T * M;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i % M);
        const size_t bt = i / M;
        const int t = (int)(bt % T), b = (int)(bt / T);
        const float lo = spec_min[n_spec == 1 ? 0 : m], hi = spec_max[n_spec == 1 ? 0 : m];
        float v = (x[((size_t)b * stride + t) * M + m] + 1.0f) / 2.0f * (hi - lo) + lo;
        if (mel2ph && mel2ph[bt] <= 0) v = v * 0.0f;
        if (t >= lens[b]) v = 0.0f;                          // zero padding beyond the clip's own length
        mel[i] = v;
    }
}

__global__ void k_set_int(int* p, int v) { *p = v; }
// p[i] = base + i*stride  (clip ids first_clip + b; stride 0 fills a constant: the default per-clip length)
__global__ void k_iota_int(int* p, int base, int stride, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = base + i * stride;
}
// p[i] = min(max(src[i], lo), hi)  (caller-supplied per-clip lengths, clamped to the workspace)
// dsvc_denoiser_forward's diffusion steps: a clamped copy (the FiLM table is tabulated for 0 .. K-1: nothing may index outside it) and a
// sticky error flag in host-mapped memory instead of a device-to-host check per call (the denoiser seam is called 1000 times per clip)
__global__ void k_clamp_steps(int* __restrict__ dst, const int* __restrict__ src, int K, int n, int* err_flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int t = src[i];
    if (t < 0 || t >= K) *err_flag = 1;
    dst[i] = t < 0 ? 0 : (t >= K ? K - 1 : t);
}

__global__ void k_clamp_copy_int(int* p, const int* __restrict__ src, int lo, int hi, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int v = src[i]; p[i] = v < lo ? lo : (v > hi ? hi : v); }
}
__global__ void k_add_int(int* p, int d) { *p += d; }

// ---- step-embedding tables (K2, K3): depend on the integer step only => built once per checkpoint ----
// emb[t][c] : SinusoidalPosEmb (net.py:32-44)
__global__ void k_sin_emb(float* __restrict__ emb, int K, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K * C) return;
    const int t = i / C, c = i % C;
    const int half = C / 2;
    const float scale = logf(10000.0f) / (float)(half - 1);
    const int k = c < half ? c : c - half;
    const float ang = (float)t * expf((float)k * -scale);
    emb[i] = c < half ? sinf(ang) : cosf(ang);
}

// out[r][o] = act( sum_i W[o][i] * in[r][i] + b[o] ),  act: 0 none, 1 Mish.  One thread per output,
// fp32 FMA chain -- run once at load time, not on the hot path.
__global__ void k_linear_rows(const float* __restrict__ in, const float* __restrict__ W, const float* __restrict__ b,
                              float* __restrict__ out, int R, int I, int O, int out_ld, int act) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (o >= O || r >= R) return;
    const float* w = W + (size_t)o * I;
    const float* x = in + (size_t)r * I;
    float s = 0.f;
    for (int i = 0; i < I; ++i) s = fmaf(w[i], x[i], s);
    s += b[o];
    if (act == 1) {
        const float sp = s > 20.f ? s : log1pf(expf(s));     // softplus (torch threshold 20)
        s = s * tanhf(sp);
    }
    out[(size_t)r * out_ld + o] = s;
}

// ---- PLMS (diffusion.py:165-198) on the frame-major state ----
struct PlmsArgs {
    float* x;                // [rows][M] state (in/out)
    const float* eps;        // [rows][M] current prediction
    float* hist;             // 4 slots of [rows][M]; slot of prediction n is n & 3
    float* x_pred;           // [rows][M] out of phase 0
    const float* alphas_cumprod;
    size_t n;                // rows*M
    int t, t_prev;           // t, max(t - interval, 0)
    int n_hist;              // predictions stored so far
    const int* state_dev;    // phase 2 under graph replay: {t, n_hist} live on the device (state_dev[0], state_dev[1]); null = use the fields
    int interval;            // ... and t_prev = max(t - interval, 0)
    int phase;               // 0: x_pred = xpred(x, eps, t)                 (first iteration, before the 2nd eval)
                             // 1: eps' = (hist[0] + eps)/2 ; x = xpred(x, eps', t)   (first iteration, after the 2nd eval; hist[0] holds eps_0)
                             // 2: eps' = AB(eps, hist); x = xpred(x, eps', t); push eps
    // round 6: the fp16 [hi | lo] planes of the value the NEXT evaluation's input projection reads (x_pred in phase 0, the new x otherwise), written
    // here instead of by a k_rows_to_half launch in front of every evaluation; null on the conv_gemm engine
    _Float16* xsh;           // [rows][2 * ldh]
    const int* rowclip;      // [rows]: < 0 on gap / padded rows (left alone, as k_rows_to_half leaves them)
    int M, ldh;
};

__device__ __forceinline__ float plms_xpred(float x, float e, float a_t, float a_p) {
    const float a_t_sq = sqrtf(a_t), a_p_sq = sqrtf(a_p);
    const float cx = 1.0f / (a_t_sq * (a_t_sq + a_p_sq));
    const float ce = 1.0f / (a_t_sq * (sqrtf((1.0f - a_p) * a_t) + sqrtf((1.0f - a_t) * a_p)));
    return x + (a_p - a_t) * (cx * x - ce * e);
}

__global__ void k_plms(const PlmsArgs a) {
    const int t = a.state_dev ? a.state_dev[0] : a.t;
    const int t_prev = a.state_dev ? (t - a.interval > 0 ? t - a.interval : 0) : a.t_prev;
    const int n_hist = a.state_dev ? a.state_dev[1] : a.n_hist;
    const float a_t = a.alphas_cumprod[t], a_p = a.alphas_cumprod[t_prev];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
        const float e = a.eps[i];
        const float x = a.x[i];
        float xn;                                            // what the next evaluation reads
        if (a.phase == 0) {
            xn = plms_xpred(x, e, a_t, a_p);
            a.x_pred[i] = xn;
            a.hist[i] = e;                                   // slot 0 <- eps_0
        } else if (a.phase == 1) {
            const float ep = (a.hist[i] + e) / 2.0f;
            xn = plms_xpred(x, ep, a_t, a_p);
            a.x[i] = xn;
        } else {
            const int nh = n_hist;
            const float h1 = a.hist[(size_t)((nh - 1) & 3) * a.n + i];
            float ep;
            if (nh == 1) {
                ep = (3.0f * e - h1) / 2.0f;
            } else if (nh == 2) {
                const float h2 = a.hist[(size_t)((nh - 2) & 3) * a.n + i];
                ep = (23.0f * e - 16.0f * h1 + 5.0f * h2) / 12.0f;
            } else {
                const float h2 = a.hist[(size_t)((nh - 2) & 3) * a.n + i];
                const float h3 = a.hist[(size_t)((nh - 3) & 3) * a.n + i];
                ep = (55.0f * e - 59.0f * h1 + 37.0f * h2 - 9.0f * h3) / 24.0f;
            }
            xn = plms_xpred(x, ep, a_t, a_p);
            a.x[i] = xn;
            a.hist[(size_t)(nh & 3) * a.n + i] = e;
        }
        if (a.xsh) {
            const size_t row = i / (size_t)a.M;
            if (a.rowclip[row] >= 0) {
                const int c = (int)(i - row * (size_t)a.M);
                const _Float16 h = (_Float16)xn;
                _Float16* q = a.xsh + row * (size_t)(2 * a.ldh) + c;
                q[0] = h;
                q[a.ldh] = (_Float16)(xn - (float)h);
            }
        }
    }
}

// norm_spec (diffusion.py:286-287) of a reference mel [B][T][M] (log10) into the frame-major state [B*stride][M]
__global__ void k_norm_ref_mel(const float* __restrict__ mel, float* __restrict__ x, const float* __restrict__ spec_min,
                               const float* __restrict__ spec_max, int n_spec, int B, int T, int M, int stride) {
    const size_t n = (size_t)B * T * M;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i % M);
        const size_t bt = i / M;
        const int t = (int)(bt % T), b = (int)(bt / T);
        const float lo = spec_min[n_spec == 1 ? 0 : m], hi = spec_max[n_spec == 1 ? 0 : m];
        x[((size_t)b * stride + t) * M + m] = (mel[i] - lo) / (hi - lo) * 2.0f - 1.0f;
    }
}

// q_sample (diffusion.py:200-205) in place on the frame-major state: x = sa*x0 + sb*noise(Philox X_INIT stream)
__global__ void k_q_sample(float* __restrict__ x, int B, int T, int M, int stride, float sa, float sb,
                           unsigned long long seed, const int* __restrict__ clipid) {
    const int quads = T * M / 4;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (q >= quads) return;
    float z[4];
    philox_normal4((unsigned)q, 0u, (unsigned)clipid[b], PURPOSE_X_INIT, seed, z);
    float4* p = reinterpret_cast<float4*>(x + (size_t)b * stride * M + (size_t)q * 4);
    float4 v = *p;
    v.x = sa * v.x + sb * z[0]; v.y = sa * v.y + sb * z[1]; v.z = sa * v.z + sb * z[2]; v.w = sa * v.w + sb * z[3];
    *p = v;
}

}  // namespace dsvc
