// HuBERT-soft content encoder behind the C ABI (include/dsvc.h: dsvc_hubert_*): 16 kHz waveform -> soft speech units [T, 256] at 50 Hz.
// Reference: network/hubert/hubert_model.py:16-160 (Hubert / HubertSoft.units, FeatureExtractor, FeatureProjection,
// PositionalConvEmbedding, TransformerEncoder of 12 nn.TransformerEncoderLayer(768, 12, 3072, gelu, batch_first, post-LN)), called through
// preprocessing/hubertinfer.py:30-42 (Hubertencoder.encode) -- the step in front of the hot path.
//
// Layout: fp32 frame-major rows [frame][channel], one utterance per call (the reference encodes one wav at a time).  Every dense
// contraction runs on the conv_gemm MFMA engine with split fp16 operands (fp32-class), weights packed into MFMA fragments on the device:
//   * a stride-2 conv over [L][512] is a dense conv over the SAME memory read as [L/2][1024] rows (two frames per row): k=3 -> two row
//     taps (conv_gemm's centred 3-tap form with a zero first tap), k=2 -> a plain 1x1 over the paired rows;
//   * the grouped positional conv (k=128, 16 groups of 48) is 16 launches on 48-channel column slices, 128 centred taps = padding 64 with
//     the surplus last frame never computed;
//   * attention: per head, K_h and V_h^T are packed as the "weights" of two GEMMs (scores = Q_h K_h^T / 8, out = softmax(scores) V_h).
// conv0 (1 -> 512, k=10, stride 5) is 10 FMAs per output: a direct fp32 kernel.  GroupNorm(512, 512) is a per-channel normalisation
// over time (double-precision moments), LayerNorm / softmax / GELU (exact erf form) are row kernels.
#include <math.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/dsvc.h"
#include "cg_util.h"
#include "rowops.h"

using namespace dsvc;

namespace {

constexpr int HB_C0 = 512, HB_D = 768, HB_FF = 3072, HB_HEADS = 12, HB_HD = 64, HB_LAYERS = 12, HB_OUT = 256, HB_GROUPS = 16, HB_PK = 128;
constexpr size_t HB_SCORE_BYTES = (size_t)1 << 30;      // budget of the heads' score matrices [heads in a launch][T][T] fp32: 12 heads fit up to T = 4 700 frames (94 s);
                                                        // longer utterances run the heads in groups (head_batch).  (4 GB held next to the denoiser / vocoder workspaces, grow-only)

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// out[row][col_off + col] = act(acc * scale + bias[col]) + res[row][col]   on rows < n_valid (others 0).  In a batched launch (ConvGemmArgs::nz)
// problem z = blockIdx.z writes out + z * out_z at columns col_off + z * col_z (bias and residual columns move with them): the heads of an
// attention layer land in their own score matrices / their own 64 columns of the context, the groups of the positional conv in their 48
struct EpLin {
    static constexpr bool PAIRED = false;
    struct Args { float* out; int ld; const float* bias; int cout; int act; const float* res; int ldres; int col_off; int n_valid; float scale; long long out_z; int col_z; };
    __device__ __forceinline__ void one(const Args& e, int row, int col, float v) const {
        if (col >= e.cout) return;
        const int z = blockIdx.z, co = e.col_off + z * e.col_z + col;
        v = v * e.scale + (e.bias ? e.bias[z * e.col_z + col] : 0.f);
        if (e.act == 1) v = gelu_exact(v);
        if (e.res) v += e.res[(size_t)row * e.ldres + co];
        e.out[(size_t)z * e.out_z + (size_t)row * e.ld + co] = row < e.n_valid ? v : 0.f;
    }
};

// wav [n] -> padded [pad | wav | pad]  (HubertSoft.units: F.pad(wav, (40, 40)), hubert_model.py:75)
__global__ void k_pad_wav(const float* __restrict__ wav, float* __restrict__ dst, long long n, int pad) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n + 2 * pad) return;
    dst[i] = (i >= pad && i < n + pad) ? wav[i - pad] : 0.f;
}

// conv0: Conv1d(1, 512, 10, stride 5, no bias)  (hubert_model.py:85)  out[t][c] = sum_j w[c][j] x[5t + j]
__global__ void k_conv0(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ out, int L) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)L * HB_C0) return;
    const int t = (int)(i / HB_C0), c = (int)(i - (long long)t * HB_C0);
    const float* xp = x + (long long)t * 5;
    const float* wp = w + c * 10;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 10; ++j) s = fmaf(wp[j], xp[j], s);
    out[i] = s;
}

// GroupNorm(512, 512): per-channel moments over time; grid (C/64, chunks), double atomics into sums[2][C]
__global__ void k_gn_stats(const float* __restrict__ x, double* __restrict__ sums, int L, int C, int rows_per_block) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int sub = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_block;
    double s = 0.0, q = 0.0;
    for (int r = r0 + sub; r < r0 + rows_per_block && r < L; r += 4) {
        const double v = (double)x[(size_t)r * C + c];
        s += v; q += v * v;
    }
    __shared__ double rs[4][64], rq[4][64];
    rs[sub][threadIdx.x & 63] = s; rq[sub][threadIdx.x & 63] = q;
    __syncthreads();
    if (sub == 0) {
        const int l = threadIdx.x;
        atomicAdd(sums + c, rs[0][l] + rs[1][l] + rs[2][l] + rs[3][l]);
        atomicAdd(sums + C + c, rq[0][l] + rq[1][l] + rq[2][l] + rq[3][l]);
    }
}
// x <- gelu((x - mean) * rstd * gamma + beta)   (biased variance, eps 1e-5: torch.nn.GroupNorm)
__global__ void k_gn_apply_gelu(float* __restrict__ x, const double* __restrict__ sums, const float* __restrict__ gamma,
                                const float* __restrict__ beta, int L, int C) {
    const long long n = (long long)L * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const double mean = sums[c] / L;
        const double var = sums[C + c] / L - mean * mean;
        const float rstd = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + 1e-5));
        x[i] = gelu_exact((x[i] - (float)mean) * rstd * gamma[c] + beta[c]);
    }
}

// row softmax over the first n columns of [rows][ld]; columns n..ld-1 are zeroed (they are K padding of the next GEMM)
// (blockIdx.y walks matrices z_stride floats apart: the heads of a layer in one launch)
__global__ void k_softmax_rows(float* __restrict__ s, int rows, int n, int ld, long long z_stride) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float* p = s + (size_t)blockIdx.y * z_stride + (size_t)row * ld;
    float m = -INFINITY;
    for (int c = lane; c < n; c += 64) m = fmaxf(m, p[c]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float sum = 0.f;
    for (int c = lane; c < n; c += 64) { const float e = expf(p[c] - m); p[c] = e; sum += e; }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
    for (int c = lane; c < ld; c += 64) p[c] = c < n ? p[c] * inv : 0.f;
}

struct Packed {
    DevBuf w;
    int n_ctiles = 0, taps = 1, cin_pad = 0;
};

}  // namespace

// =================================================================================================
struct dsvc_hubert {
    std::map<std::string, std::vector<float>> host;
    bool finalized = false;
    DevBuf conv0_w, gn_g, gn_b, fp_ln_g, fp_ln_b, fp_b, pos_w, pos_b, ln_g, ln_b, proj_b;
    Packed fe[6], fp_w, pos[HB_GROUPS], proj_w;
    struct Layer { Packed in_w, out_w, ff1, ff2; DevBuf in_b, out_b, b1, b2, n1g, n1b, n2g, n2b; } layers[HB_LAYERS];
    // workspace
    long long wsN = -1;
    int L[7] = {0, 0, 0, 0, 0, 0, 0};
    DevBuf wavp, c[2], gsum, h, h2, posb, qkv, S, attn, ffb, packK, packV;
    // round 5: the 12 heads of a layer and the 16 groups of the positional conv run as ONE batched launch each (conv_gemm.h: ConvGemmArgs::nz)
    DevBuf pos_all;                                // the groups' packed weights, pos_halfs apart
    size_t pos_halfs = 0, packK_halfs = 0, packV_halfs = 0;      // per group / per head
    DevBuf head_packs;                             // 2 x HB_HEADS PackDesc: K_h and V_h^T of every head out of the qkv buffer (k_pack_w_batch)
    int head_batch = HB_HEADS;                     // heads per batched launch: all 12 unless their score matrices would not fit HB_SCORE_BYTES (minutes-long utterances)

    ~dsvc_hubert() {
        for (DevBuf* b : {&conv0_w, &gn_g, &gn_b, &fp_ln_g, &fp_ln_b, &fp_b, &pos_w, &pos_b, &ln_g, &ln_b, &proj_b, &wavp, &c[0], &c[1], &gsum, &h, &h2,
                          &posb, &qkv, &S, &attn, &ffb, &packK, &packV, &pos_all, &head_packs})
            b->release();
        for (auto& p : fe) p.w.release();
        fp_w.w.release(); proj_w.w.release();
        for (auto& p : pos) p.w.release();
        for (auto& l : layers) {
            l.in_w.w.release(); l.out_w.w.release(); l.ff1.w.release(); l.ff2.w.release();
            for (DevBuf* b : {&l.in_b, &l.out_b, &l.b1, &l.b2, &l.n1g, &l.n1b, &l.n2g, &l.n2b}) b->release();
        }
    }
    const std::vector<float>* get(const std::string& k, size_t numel) {
        auto it = host.find(k);
        if (it == host.end()) { fail(DSVC_ESTATE, "hubert: tensor '%s' was never loaded", k.c_str()); return nullptr; }
        if (it->second.size() != numel) { fail(DSVC_EINVAL, "hubert: tensor '%s' has %zu elements, expected %zu", k.c_str(), it->second.size(), numel); return nullptr; }
        return &it->second;
    }
    int up(DevBuf& b, const std::string& k, size_t numel) {
        const std::vector<float>* v = get(k, numel);
        if (!v) return DSVC_ESTATE;
        DSVC_TRY(b.alloc(numel * 4));
        DSVC_HIP(hipMemcpy(b.p, v->data(), numel * 4, hipMemcpyHostToDevice));
        return DSVC_OK;
    }
    // pack a host fp32 matrix W(col, tap, ci) = src[col*s_col + ci*s_ci + tap*s_tap] into fragments on the device
    int pack(Packed& pk, const float* host_src, size_t numel, int cout, int taps, int cin, long long s_col, long long s_ci, long long s_tap) {
        DevBuf tmp;
        DSVC_TRY(tmp.alloc(numel * 4));
        DSVC_HIP(hipMemcpy(tmp.p, host_src, numel * 4, hipMemcpyHostToDevice));
        pk.n_ctiles = round_up(ceil_div(cout, 32), 2); pk.taps = taps; pk.cin_pad = round_up(cin, 16);
        const size_t halfs = packed_halfs(pk.n_ctiles, taps, pk.cin_pad, 2);
        DSVC_TRY(pk.w.alloc(halfs * 2));
        const long long total_el = (long long)halfs / 2;
        hipLaunchKernelGGL(k_pack_w, dim3((unsigned)((total_el + 255) / 256 < 8192 ? (total_el + 255) / 256 : 8192)), dim3(256), 0, 0, tmp.as<float>(),
                           (const int*)nullptr, pk.w.as<_Float16>(), pk.n_ctiles, taps, pk.cin_pad, cout, cin, s_col, s_ci, s_tap, 0, 1.0f);
        DSVC_HIP(hipGetLastError());
        DSVC_HIP(hipDeviceSynchronize());
        tmp.release();
        return DSVC_OK;
    }
    int finalize();
    int ensure_ws(long long n, hipStream_t st);
    int units(const float* wav, long long n, float* out, hipStream_t st);
};

static void hubert_lengths(long long n, int (&L)[7]) {
    const long long n0 = n + 80;
    long long l = n0 >= 10 ? (n0 - 10) / 5 + 1 : 0;
    L[0] = (int)l;
    const int k[6] = {3, 3, 3, 3, 2, 2};
    for (int i = 0; i < 6; ++i) { l = l >= k[i] ? (l - k[i]) / 2 + 1 : 0; L[i + 1] = (int)l; }
}

int dsvc_hubert::finalize() {
#define GET(var, key, n) const std::vector<float>* var = get(key, (size_t)(n)); if (!var) return DSVC_ESTATE
    DSVC_TRY(up(conv0_w, "feature_extractor.conv0.weight", (size_t)HB_C0 * 10));
    DSVC_TRY(up(gn_g, "feature_extractor.norm0.weight", HB_C0)); DSVC_TRY(up(gn_b, "feature_extractor.norm0.bias", HB_C0));
    for (int i = 1; i <= 6; ++i) {
        const int k = i <= 4 ? 3 : 2;
        GET(w, "feature_extractor.conv" + std::to_string(i) + ".weight", (size_t)HB_C0 * HB_C0 * k);
        if (k == 3) {
            // rows hold two frames [pos 0 | pos 1]; output q reads rows q (tap 1: j = pos) and q+1 (tap 2: j = 2 + pos, only pos 0 exists); tap 0 = 0
            std::vector<float> r((size_t)HB_C0 * 3 * 2 * HB_C0, 0.f);           // [o][tap 3][pos 2][c]
            for (int o = 0; o < HB_C0; ++o)
                for (int cc = 0; cc < HB_C0; ++cc)
                    for (int j = 0; j < 3; ++j) {
                        const int tap = 1 + j / 2, pos = j % 2;
                        r[(((size_t)o * 3 + tap) * 2 + pos) * HB_C0 + cc] = (*w)[((size_t)o * HB_C0 + cc) * 3 + j];
                    }
            DSVC_TRY(pack(fe[i - 1], r.data(), r.size(), HB_C0, 3, 2 * HB_C0, (long long)3 * 2 * HB_C0, 1, (long long)2 * HB_C0));
        } else {
            std::vector<float> r((size_t)HB_C0 * 2 * HB_C0);                     // [o][pos 2][c]
            for (int o = 0; o < HB_C0; ++o)
                for (int cc = 0; cc < HB_C0; ++cc)
                    for (int j = 0; j < 2; ++j) r[((size_t)o * 2 + j) * HB_C0 + cc] = (*w)[((size_t)o * HB_C0 + cc) * 2 + j];
            DSVC_TRY(pack(fe[i - 1], r.data(), r.size(), HB_C0, 1, 2 * HB_C0, (long long)2 * HB_C0, 1, 0));
        }
    }
    DSVC_TRY(up(fp_ln_g, "feature_projection.norm.weight", HB_C0)); DSVC_TRY(up(fp_ln_b, "feature_projection.norm.bias", HB_C0));
    {
        GET(w, "feature_projection.projection.weight", (size_t)HB_D * HB_C0);
        DSVC_TRY(pack(fp_w, w->data(), w->size(), HB_D, 1, HB_C0, HB_C0, 1, 0));
        DSVC_TRY(up(fp_b, "feature_projection.projection.bias", HB_D));
    }
    {   // positional conv: weight_norm(dim=2): w[o][c][k] = g[k] * v[o][c][k] / ||v[:, :, k]||   (hubert_model.py:128-137)
        const int gc = HB_D / HB_GROUPS;
        GET(g, "positional_embedding.conv.weight_g", HB_PK);
        GET(v, "positional_embedding.conv.weight_v", (size_t)HB_D * gc * HB_PK);
        std::vector<double> nrm(HB_PK, 0.0);
        for (size_t i = 0; i < v->size(); ++i) nrm[i % HB_PK] += (double)(*v)[i] * (double)(*v)[i];
        std::vector<float> w(v->size());
        for (size_t i = 0; i < v->size(); ++i) w[i] = (float)((double)(*g)[i % HB_PK] * (double)(*v)[i] / sqrt(nrm[i % HB_PK]));
        for (int gi = 0; gi < HB_GROUPS; ++gi)      // group gi: W(col = o_local, tap = k, ci = c) = w[(gc*gi + o_local)][c][k]
            DSVC_TRY(pack(pos[gi], w.data() + (size_t)gi * gc * gc * HB_PK, (size_t)gc * gc * HB_PK, gc, HB_PK, gc, (long long)gc * HB_PK, HB_PK, 1));
        // ... side by side, so that one batched launch walks the groups
        pos_halfs = packed_halfs(pos[0].n_ctiles, pos[0].taps, pos[0].cin_pad, 2);
        DSVC_TRY(pos_all.alloc(pos_halfs * 2 * HB_GROUPS));
        for (int gi = 0; gi < HB_GROUPS; ++gi) {
            DSVC_HIP(hipMemcpy(pos_all.as<_Float16>() + (size_t)gi * pos_halfs, pos[gi].w.p, pos_halfs * 2, hipMemcpyDeviceToDevice));
            pos[gi].w.release();
        }
        DSVC_TRY(up(pos_b, "positional_embedding.conv.bias", HB_D));
    }
    DSVC_TRY(up(ln_g, "norm.weight", HB_D)); DSVC_TRY(up(ln_b, "norm.bias", HB_D));
    for (int l = 0; l < HB_LAYERS; ++l) {
        const std::string q = "encoder.layers." + std::to_string(l) + ".";
        Layer& y = layers[l];
        GET(wi, q + "self_attn.in_proj_weight", (size_t)3 * HB_D * HB_D);
        DSVC_TRY(pack(y.in_w, wi->data(), wi->size(), 3 * HB_D, 1, HB_D, HB_D, 1, 0));
        DSVC_TRY(up(y.in_b, q + "self_attn.in_proj_bias", 3 * HB_D));
        GET(wo, q + "self_attn.out_proj.weight", (size_t)HB_D * HB_D);
        DSVC_TRY(pack(y.out_w, wo->data(), wo->size(), HB_D, 1, HB_D, HB_D, 1, 0));
        DSVC_TRY(up(y.out_b, q + "self_attn.out_proj.bias", HB_D));
        GET(w1, q + "linear1.weight", (size_t)HB_FF * HB_D);
        DSVC_TRY(pack(y.ff1, w1->data(), w1->size(), HB_FF, 1, HB_D, HB_D, 1, 0));
        DSVC_TRY(up(y.b1, q + "linear1.bias", HB_FF));
        GET(w2, q + "linear2.weight", (size_t)HB_D * HB_FF);
        DSVC_TRY(pack(y.ff2, w2->data(), w2->size(), HB_D, 1, HB_FF, HB_FF, 1, 0));
        DSVC_TRY(up(y.b2, q + "linear2.bias", HB_D));
        DSVC_TRY(up(y.n1g, q + "norm1.weight", HB_D)); DSVC_TRY(up(y.n1b, q + "norm1.bias", HB_D));
        DSVC_TRY(up(y.n2g, q + "norm2.weight", HB_D)); DSVC_TRY(up(y.n2b, q + "norm2.bias", HB_D));
    }
    {
        GET(w, "proj.weight", (size_t)HB_OUT * HB_D);
        DSVC_TRY(pack(proj_w, w->data(), w->size(), HB_OUT, 1, HB_D, HB_D, 1, 0));
        DSVC_TRY(up(proj_b, "proj.bias", HB_OUT));
    }
#undef GET
    host.clear();
    finalized = true;
    return DSVC_OK;
}

int dsvc_hubert::ensure_ws(long long n, hipStream_t st) {
    if (n == wsN) return DSVC_OK;
    hubert_lengths(n, L);
    if (L[6] < 1) return fail(DSVC_EINVAL, "hubert: %lld samples are too short for one output frame", n);
    const size_t T = (size_t)L[6];
    DSVC_TRY(wavp.alloc((size_t)(n + 80 + 16) * 4));
    const size_t crow = (size_t)round_up(L[0] + 2, 64);
    DSVC_TRY(c[0].alloc(crow * HB_C0 * 4)); DSVC_TRY(c[1].alloc(crow * HB_C0 * 4));
    DSVC_TRY(gsum.alloc(2 * HB_C0 * 8));
    const size_t Tr = (size_t)round_up((int)T, 32), Tp = Tr;
    DSVC_TRY(h.alloc(Tr * HB_D * 4)); DSVC_TRY(h2.alloc(Tr * HB_D * 4)); DSVC_TRY(posb.alloc(Tr * HB_D * 4)); DSVC_TRY(qkv.alloc(Tr * 3 * HB_D * 4));
    head_batch = (int)(HB_SCORE_BYTES / (Tr * Tp * 4));
    head_batch = head_batch < 1 ? 1 : (head_batch > HB_HEADS ? HB_HEADS : head_batch);
    DSVC_TRY(S.alloc(Tr * Tp * 4 * head_batch)); DSVC_TRY(attn.alloc(Tr * HB_D * 4)); DSVC_TRY(ffb.alloc(Tr * HB_FF * 4));
    const int nctK = round_up(ceil_div((int)T, 32), 2), Tk = round_up((int)T, 16);
    packK_halfs = packed_halfs(nctK, 1, HB_HD, 2); packV_halfs = packed_halfs(2, 1, Tk, 2);
    DSVC_TRY(packK.alloc(packK_halfs * 2 * HB_HEADS));
    DSVC_TRY(packV.alloc(packV_halfs * 2 * HB_HEADS));
    {   // K_h as the weights of scores = Q_h K_h^T: W(col = j, ci = d) = K[j][d]; V_h^T as those of out = P V_h: W(col = d, ci = j) = V[j][d] -- the
        // qkv buffer is the same for every layer, so the 24 descriptors are built once per utterance length
        std::vector<PackDesc> pd(2 * HB_HEADS);
        for (int hd = 0; hd < HB_HEADS; ++hd) {
            const float* K = qkv.as<float>() + HB_D + hd * HB_HD;
            const float* V = qkv.as<float>() + 2 * HB_D + hd * HB_HD;
            pd[hd] = PackDesc{K, nullptr, packK.as<_Float16>() + (size_t)hd * packK_halfs, nctK, 1, HB_HD, (int)T, HB_HD, (long long)3 * HB_D, 1LL, 0LL, 0, 1.0f};
            pd[HB_HEADS + hd] = PackDesc{V, nullptr, packV.as<_Float16>() + (size_t)hd * packV_halfs, 2, 1, Tk, HB_HD, (int)T, 1LL, (long long)3 * HB_D, 0LL, 0, 1.0f};
        }
        DSVC_TRY(head_packs.alloc(pd.size() * sizeof(PackDesc)));
        DSVC_HIP(hipMemcpyAsync(head_packs.p, pd.data(), pd.size() * sizeof(PackDesc), hipMemcpyHostToDevice, st));
        DSVC_HIP(hipStreamSynchronize(st));                // (pd is a host temporary)
    }
    wsN = n;
    return DSVC_OK;
}

int dsvc_hubert::units(const float* wav, long long n, float* out, hipStream_t st) {
    DSVC_TRY(ensure_ws(n, st));
    const int T = L[6];
    auto gemm = [&](const float* x, int ldx, int n_rows, int cin, const Packed& pk, int taps, const EpLin::Args& e) -> int {
        ConvGemmArgs a{};
        a.x = x; a.ldx = ldx; a.n_rows = n_rows; a.clip_stride = n_rows < 32 ? 32 : n_rows; a.clip_len = n_rows;
        a.cin = cin; a.taps = taps; a.dil = 1; a.w = pk.w.as<_Float16>(); a.n_ctiles = pk.n_ctiles; a.w_planes = 2; a.in_slope = 1.0f;
        return launch<EpLin>(a, e, st);
    };
    // ---- feature extractor (hubert_model.py:82-102) ----
    hipLaunchKernelGGL(k_pad_wav, dim3((unsigned)((n + 80 + 255) / 256)), dim3(256), 0, st, wav, wavp.as<float>(), n, 40);
    DSVC_HIP(hipMemsetAsync(c[0].p, 0, c[0].bytes, st));
    DSVC_HIP(hipMemsetAsync(c[1].p, 0, c[1].bytes, st));
    hipLaunchKernelGGL(k_conv0, dim3((unsigned)(((long long)L[0] * HB_C0 + 255) / 256)), dim3(256), 0, st, wavp.as<float>(), conv0_w.as<float>(), c[0].as<float>(), L[0]);
    DSVC_HIP(hipMemsetAsync(gsum.p, 0, 2 * HB_C0 * 8, st));
    hipLaunchKernelGGL(k_gn_stats, dim3(HB_C0 / 64, ceil_div(L[0], 1024)), dim3(256), 0, st, c[0].as<float>(), gsum.as<double>(), L[0], HB_C0, 1024);
    hipLaunchKernelGGL(k_gn_apply_gelu, dim3(4096), dim3(256), 0, st, c[0].as<float>(), gsum.as<double>(), gn_g.as<float>(), gn_b.as<float>(), L[0], HB_C0);
    int cur = 0;
    for (int i = 1; i <= 6; ++i) {
        // stride-2 conv: the input [L][512] read as [ceil(L/2)][1024]; rows >= L_out of the output are zeroed
        const int rows_in = (L[i - 1] + 1) / 2;
        if (i >= 2) DSVC_HIP(hipMemsetAsync(c[cur ^ 1].p, 0, (size_t)round_up(L[i - 1] + 2, 64) * HB_C0 * 4 < c[cur ^ 1].bytes ? (size_t)round_up(L[i - 1] + 2, 64) * HB_C0 * 4 : c[cur ^ 1].bytes, st));
        EpLin::Args e{c[cur ^ 1].as<float>(), HB_C0, nullptr, HB_C0, 1, nullptr, 0, 0, L[i], 1.0f};
        DSVC_TRY(gemm(c[cur].as<float>(), 2 * HB_C0, rows_in, 2 * HB_C0, fe[i - 1], fe[i - 1].taps, e));
        cur ^= 1;
    }
    const float* feat = c[cur].as<float>();                       // [T][512]
    // ---- feature projection: LayerNorm(512) -> Linear(512, 768)  (hubert_model.py:105-116) ----
    hipLaunchKernelGGL(k_layernorm, dim3(ceil_div(T, 4)), dim3(256), 0, st, feat, c[cur ^ 1].as<float>(), fp_ln_g.as<float>(), fp_ln_b.as<float>(), T, HB_C0, 1e-5f);
    {
        EpLin::Args e{h.as<float>(), HB_D, fp_b.as<float>(), HB_D, 0, nullptr, 0, 0, T, 1.0f};
        DSVC_TRY(gemm(c[cur ^ 1].as<float>(), HB_C0, T, HB_C0, fp_w, 1, e));
    }
    // ---- x + positional conv embedding (grouped k=128 conv, GELU), then LayerNorm(768)  (hubert_model.py:48-49,119-137) ----
    {   // the 16 groups are 16 problems of one launch: group z reads / writes columns 48 z .. 48 z + 47 (16 launches of 16 workgroups each took
        // 3.5 of HuBERT's 10.4 ms, profiles/r5h_kernel_stats_hubert.csv)
        const int gc = HB_D / HB_GROUPS;
        ConvGemmArgs a{};
        a.x = h.as<float>(); a.ldx = HB_D; a.n_rows = T; a.clip_stride = T < 32 ? 32 : T; a.clip_len = T;
        a.cin = gc; a.taps = HB_PK; a.dil = 1; a.w = pos_all.as<_Float16>(); a.n_ctiles = pos[0].n_ctiles; a.w_planes = 2; a.in_slope = 1.0f;
        a.nz = HB_GROUPS; a.x_z = gc; a.w_z = (long long)pos_halfs;
        EpLin::Args e{posb.as<float>(), HB_D, pos_b.as<float>(), gc, 1, h.as<float>(), HB_D, 0, T, 1.0f, 0LL, gc};
        DSVC_TRY(launch<EpLin>(a, e, st));
    }
    hipLaunchKernelGGL(k_layernorm, dim3(ceil_div(T, 4)), dim3(256), 0, st, posb.as<float>(), h.as<float>(), ln_g.as<float>(), ln_b.as<float>(), T, HB_D, 1e-5f);
    // ---- 12 post-LN transformer encoder layers (nn.TransformerEncoderLayer(768, 12, 3072, gelu, batch_first)) ----
    const int Tp = round_up(T, 32), Tk = round_up(T, 16);
    const int nctK = round_up(ceil_div(T, 32), 2);
    for (int l = 0; l < HB_LAYERS; ++l) {
        Layer& y = layers[l];
        {
            EpLin::Args e{qkv.as<float>(), 3 * HB_D, y.in_b.as<float>(), 3 * HB_D, 0, nullptr, 0, 0, T, 1.0f};
            DSVC_TRY(gemm(h.as<float>(), HB_D, T, HB_D, y.in_w, 1, e));
        }
        {   // all 12 heads at once (5 launches per layer instead of 60): K_h / V_h^T of every head packed as "weights" by one launch, then
            // scores_h = Q_h K_h^T / 8 -> softmax -> out_h = P_h V_h as batched launches (blockIdx.z = head; head h's scores in S + h * Tp * Tp)
            const long long sz = (long long)Tp * Tp;
            hipLaunchKernelGGL(k_pack_w_batch, dim3(128, 2 * HB_HEADS), dim3(256), 0, st, (const PackDesc*)head_packs.p);
            for (int h0 = 0; h0 < HB_HEADS; h0 += head_batch) {      // (one round unless the utterance is minutes long: head_batch)
                const int nb = HB_HEADS - h0 < head_batch ? HB_HEADS - h0 : head_batch;
                {
                    ConvGemmArgs a{};
                    a.x = qkv.as<float>() + h0 * HB_HD; a.ldx = 3 * HB_D; a.n_rows = T; a.clip_stride = T < 32 ? 32 : T; a.clip_len = T; a.cin = HB_HD; a.taps = 1; a.dil = 1;
                    a.w = packK.as<_Float16>() + (size_t)h0 * packK_halfs; a.n_ctiles = nctK; a.w_planes = 2; a.in_slope = 1.0f;
                    a.nz = nb; a.x_z = HB_HD; a.w_z = (long long)packK_halfs;
                    EpLin::Args e{S.as<float>(), Tp, nullptr, Tp, 0, nullptr, 0, 0, T, 0.125f, sz, 0};
                    DSVC_TRY(launch<EpLin>(a, e, st));
                }
                hipLaunchKernelGGL(k_softmax_rows, dim3(ceil_div(T, 4), nb), dim3(256), 0, st, S.as<float>(), T, T, Tp, sz);
                {
                    ConvGemmArgs a{};
                    a.x = S.as<float>(); a.ldx = Tp; a.n_rows = T; a.clip_stride = T < 32 ? 32 : T; a.clip_len = T; a.cin = Tk; a.taps = 1; a.dil = 1;
                    a.w = packV.as<_Float16>() + (size_t)h0 * packV_halfs; a.n_ctiles = 2; a.w_planes = 2; a.in_slope = 1.0f;
                    a.nz = nb; a.x_z = sz; a.w_z = (long long)packV_halfs;
                    EpLin::Args e{attn.as<float>(), HB_D, nullptr, HB_HD, 0, nullptr, 0, h0 * HB_HD, T, 1.0f, 0LL, HB_HD};
                    DSVC_TRY(launch<EpLin>(a, e, st));
                }
            }
        }
        {   // x = LayerNorm1(x + out_proj(attn))
            EpLin::Args e{h2.as<float>(), HB_D, y.out_b.as<float>(), HB_D, 0, h.as<float>(), HB_D, 0, T, 1.0f};
            DSVC_TRY(gemm(attn.as<float>(), HB_D, T, HB_D, y.out_w, 1, e));
            hipLaunchKernelGGL(k_layernorm, dim3(ceil_div(T, 4)), dim3(256), 0, st, h2.as<float>(), h.as<float>(), y.n1g.as<float>(), y.n1b.as<float>(), T, HB_D, 1e-5f);
        }
        {   // x = LayerNorm2(x + linear2(gelu(linear1(x))))
            EpLin::Args e1{ffb.as<float>(), HB_FF, y.b1.as<float>(), HB_FF, 1, nullptr, 0, 0, T, 1.0f};
            DSVC_TRY(gemm(h.as<float>(), HB_D, T, HB_D, y.ff1, 1, e1));
            EpLin::Args e2{h2.as<float>(), HB_D, y.b2.as<float>(), HB_D, 0, h.as<float>(), HB_D, 0, T, 1.0f};
            DSVC_TRY(gemm(ffb.as<float>(), HB_FF, T, HB_FF, y.ff2, 1, e2));
            hipLaunchKernelGGL(k_layernorm, dim3(ceil_div(T, 4)), dim3(256), 0, st, h2.as<float>(), h.as<float>(), y.n2g.as<float>(), y.n2b.as<float>(), T, HB_D, 1e-5f);
        }
    }
    {   // units = proj(x)  (hubert_model.py:74-77)
        EpLin::Args e{out, HB_OUT, proj_b.as<float>(), HB_OUT, 0, nullptr, 0, 0, T, 1.0f};
        DSVC_TRY(gemm(h.as<float>(), HB_D, T, HB_D, proj_w, 1, e));
    }
    DSVC_HIP(hipGetLastError());
    return DSVC_OK;
}

// =================================================================================================
extern "C" {

int dsvc_hubert_create(dsvc_hubert** out) {
    if (!out) return fail(DSVC_EINVAL, "null argument");
    int ndev = 0;
    DSVC_HIP(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(DSVC_EHIP, "no HIP device visible");
    *out = new dsvc_hubert();
    return DSVC_OK;
}

int dsvc_hubert_load_tensor(dsvc_hubert* h, const char* name, const float* host, int64_t numel) {
    if (!h || !name || !host || numel < 0) return fail(DSVC_EINVAL, "null argument");
    if (h->finalized) return fail(DSVC_ESTATE, "hubert already finalized");
    h->host[name].assign(host, host + numel);
    return DSVC_OK;
}

int dsvc_hubert_finalize(dsvc_hubert* h) {
    if (!h) return fail(DSVC_EINVAL, "null handle");
    if (h->finalized) return DSVC_OK;
    return h->finalize();
}

void dsvc_hubert_destroy(dsvc_hubert* h) { delete h; }

int dsvc_hubert_frames(int64_t n_samples, int32_t* frames) {
    if (!frames || n_samples < 0) return fail(DSVC_EINVAL, "bad argument");
    int L[7];
    hubert_lengths(n_samples, L);
    *frames = L[6];
    return DSVC_OK;
}

int dsvc_hubert_units(dsvc_hubert* h, const float* wav, int64_t n_samples, float* units, void* stream) {
    if (!h || !wav || !units) return fail(DSVC_EINVAL, "null argument");
    if (!h->finalized) return fail(DSVC_ESTATE, "hubert not finalized");
    // (320 samples already give one frame after the 40 + 40 sample padding of HubertSoft.units; units() rejects anything shorter)
    return h->units(wav, n_samples, units, (hipStream_t)stream);
}

}  // extern "C"
