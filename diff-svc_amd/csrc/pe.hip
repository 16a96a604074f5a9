// PitchExtractor (mel -> f0) behind the C ABI (include/dsvc.h: dsvc_pe_*).
// Reference: modules/fastspeech/pe.py:120-148 (PitchExtractor.forward) = Prenet (:7-42) -> ConvStacks (:81-117, ConvBlock :45-78 with
// GroupNorm(n_chans/16, n_chans)) -> PitchPredictor (modules/fastspeech/tts_modules.py:192-235; LayerNorm over channels eps 1e-12 :10-25;
// SinusoidalPositionalEmbedding + utils.make_positions, common_layers.py:88-143, utils/__init__.py:145-157) -> denorm_f0
// (utils/pitch_utils.py:63-76).  The 24 kHz demo path runs it on the sampler's mel to get the f0 the vocoder is driven with
// (infer_tools/infer_tool.py:134-136,165-166) -- SURVEY.md 8(f) rank 4.
//
// Layout: fp32 frame-major rows [clip * Ts + frame][channel], Ts = max(T + kernel/2, 32): conv_gemm's clip slots, whose gap rows read
// as zero -- the convolutions' zero padding, so a tap never reaches into the neighbouring clip.  Padding FRAMES (all-zero mel rows
// inside T) are ordinary rows, as in the reference: only the prenet masks them, ConvStacks and the predictor run over them and
// GroupNorm counts them.  Norms, positions and the f0 epilogue are row kernels.
// Round 5: every contraction runs on fp32 operands with FLOAT64 accumulation on the matrix cores (v_mfma_f64_16x16x4_f64, k_conv_f64acc below).
// Until round 4 they ran on the conv_gemm engine with split fp16 operands (22-bit operands: f0 1.3e-5 relative from the reference's).  That is
// nine times the distance the reference's own fp32 extractor keeps from a float64 evaluation of itself (1.5e-6, oracle/make_golden_cfg0.py) --
// and the NSF source of the vocoder INTEGRATES f0 into a phase: 1e-5 relative is 0.07 rad after 6 s at 200 Hz, 4e-3 RMS of PCM against a bar
// of 1e-4.  The extractor is 15 small convolutions (~6 GFLOP per 10 s clip, launch-bound), so there is one path, the exact one.
#include <math.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/dsvc.h"
#include "cg_util.h"
#include "rowops.h"

using namespace dsvc;

namespace {

// out[row][col] = ((relu?)(acc + bias[col]) * scale[col] + shift[col]) * keep[row]   on frames < T of each slot (others 0)
struct EpPe {
    static constexpr bool PAIRED = false;
    struct Args { float* out; int ld; const float* bias; int cout; int relu; const float* scale; const float* shift; const float* keep; int Ts, T; };
    __device__ __forceinline__ void one(const Args& e, int row, int col, float v) const {
        if (col >= e.cout) return;
        v += e.bias ? e.bias[col] : 0.f;
        if (e.relu) v = fmaxf(v, 0.f);
        if (e.scale) v = v * e.scale[col] + e.shift[col];
        if (e.keep) v *= e.keep[row];
        const int t = row - (row / e.Ts) * e.Ts;
        e.out[(size_t)row * e.ld + col] = t < e.T ? v : 0.f;
    }
};

// mel [B][T][M] -> rows [B*Ts][Mp] (channel pad zero, gap rows zero) and keep[row] = 1 - (sum |mel[row]| == 0)   (pe.py:30-31)
__global__ void k_pe_stage_mel(const float* __restrict__ mel, float* __restrict__ x, float* __restrict__ keep, int B, int T, int Ts, int M, int Mp) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B * Ts) return;
    const int b = row / Ts, t = row - b * Ts;
    float s = 0.f;
    for (int c = lane; c < Mp; c += 64) {
        const float v = (t < T && c < M) ? mel[((size_t)b * T + t) * M + c] : 0.f;
        x[(size_t)row * Mp + c] = v;
        s += fabsf(v);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) keep[row] = s == 0.f ? 0.f : 1.f;
}

// GroupNorm moments per (clip, group) over T frames x gs channels: sums[clip][G][2] doubles; grid (chunks, B), 256 threads
__global__ void k_pe_gn_stats(const float* __restrict__ y, double* __restrict__ sums, int T, int Ts, int C, int gs, int rows_per_block) {
    // per-channel partial moments of this block's rows, added up per GROUP in LDS before they go out: one pair of double atomics per group and block
    // (round 5; one pair per CHANNEL and block before -- 15 000 atomics on 16 addresses, 78 us for a 1.9 MB tensor, profiles/r5p_kernel_stats_pe.csv)
    __shared__ double sh[2][512];
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * rows_per_block;
    const int G = C / gs;
    for (int c0 = 0; c0 < C; c0 += 512) {
        for (int c = c0 + threadIdx.x; c < C && c < c0 + 512; c += blockDim.x) {
            double s = 0.0, q = 0.0;
            for (int t = t0; t < t0 + rows_per_block && t < T; ++t) {
                const double v = (double)y[((size_t)b * Ts + t) * C + c];
                s += v; q += v * v;
            }
            sh[0][c - c0] = s; sh[1][c - c0] = q;
        }
        __syncthreads();
        const int c_hi = C < c0 + 512 ? C : c0 + 512;
        if (gs <= 512 && c0 % gs == 0 && 512 % gs == 0) {
            for (int g = c0 / gs + threadIdx.x; g * gs < c_hi; g += blockDim.x) {      // whole groups inside this 512-channel chunk
                double s = 0.0, q = 0.0;
                for (int c = g * gs; c < (g + 1) * gs; ++c) { s += sh[0][c - c0]; q += sh[1][c - c0]; }
                atomicAdd(sums + ((size_t)b * G + g) * 2, s);
                atomicAdd(sums + ((size_t)b * G + g) * 2 + 1, q);
            }
        } else {
            for (int c = c0 + threadIdx.x; c < c_hi; c += blockDim.x) {
                atomicAdd(sums + ((size_t)b * G + c / gs) * 2, sh[0][c - c0]);
                atomicAdd(sums + ((size_t)b * G + c / gs) * 2 + 1, sh[1][c - c0]);
            }
        }
        __syncthreads();
    }
}
// x <- x + relu((y - mean) * rstd * gamma + beta)   (ConvBlock :69-78 + the residual of ConvStacks :108-110; biased variance, eps 1e-5)
__global__ void k_pe_gn_apply(float* __restrict__ x, const float* __restrict__ y, const double* __restrict__ sums, const float* __restrict__ gamma,
                              const float* __restrict__ beta, int B, int T, int Ts, int C, int gs) {
    const long long n = (long long)B * Ts * C;
    const int G = C / gs;
    const double cnt = (double)T * gs;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int row = (int)(i / C);
        const int b = row / Ts, t = row - b * Ts;
        if (t >= T) continue;
        const double* sp = sums + ((size_t)b * G + c / gs) * 2;
        const double mean = sp[0] / cnt;
        const double var = sp[1] / cnt - mean * mean;
        const float rstd = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + 1e-5));
        const float v = (y[i] - (float)mean) * rstd * gamma[c] + beta[c];
        x[i] += fmaxf(v, 0.f);
    }
}

// positions = cumsum(x[..., 0] != 0) * (x[..., 0] != 0)  (make_positions with padding_idx 0), then x += alpha * table[position]
// one wave per clip for the scan; pos[row] int
__global__ void k_pe_positions(const float* __restrict__ x, int* __restrict__ pos, int T, int Ts, int C) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    int run = 0;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        const bool m = t < T && x[((size_t)b * Ts + t) * C] != 0.f;
        const unsigned long long bal = __ballot(m);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (t < T) pos[b * Ts + t] = m ? run + before + 1 : 0;
        run += __popcll(bal);
    }
}
__global__ void k_pe_add_pos(float* __restrict__ x, const int* __restrict__ pos, const float* __restrict__ table, const float* __restrict__ alpha,
                             int B, int T, int Ts, int C) {
    const long long n = (long long)B * Ts * C;
    const float al = alpha[0];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int row = (int)(i / C);
        if (row - (row / Ts) * Ts >= T) continue;
        x[i] += al * table[(size_t)pos[row] * C + c];
    }
}

// pred rows [B*Ts][ldp] (2 used) -> pitch_pred [B][T][2], f0 [B][T]   (pe.py:141-147 + denorm_f0)
__global__ void k_pe_finish(const float* __restrict__ pred, int ldp, const float* __restrict__ keep, float* __restrict__ pitch_pred, float* __restrict__ f0,
                            int B, int T, int Ts, int pitch_norm, float f0_mean, float f0_std, int use_uv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * T) return;
    const int b = i / T, t = i - b * T;
    const int row = b * Ts + t;
    const float p0 = pred[(size_t)row * ldp], p1 = pred[(size_t)row * ldp + 1];
    if (pitch_pred) { pitch_pred[(size_t)i * 2] = p0; pitch_pred[(size_t)i * 2 + 1] = p1; }
    float f = p0;
    if (pitch_norm == 1) f = f * f0_std + f0_mean;
    if (pitch_norm == 0) f = exp2f(f);
    if (use_uv && p1 > 0.f) f = 0.f;
    if (keep[row] == 0.f) f = 0.f;
    f0[i] = f;
}

struct Packed {
    DevBuf w;       // float64 B fragments of v_mfma_f64_16x16x4_f64: [col tile 16][tap][cin_pad / 16][lane 64][4 doubles]
    int n_ctiles = 0, taps = 1, cin_pad = 0;
};

typedef double pe_f64x4 __attribute__((ext_vector_type(4)));
typedef float pe_f32x4 __attribute__((ext_vector_type(4)));

// out[row][col] = EPI( sum_tap sum_ci  x[row + tap - taps/2][ci] * W[col][ci][tap] ): fp32 operands, products and the whole K = taps * cin
// accumulation in FLOAT64 on the matrix cores (v_mfma_f64_16x16x4_f64; a product of two fp32 values is exact in float64), ONE rounding to
// fp32 per output.  Why not fp32 accumulation: measured on an intermediate build of round 5 -- with v_mfma_f32_32x32x2_f32, 1280-term fp32 chains in 15
// consecutive convolutions left f0 5e-6 ... 1.5e-5 from a float64 evaluation of the extractor where the reference's own fp32 CPU arithmetic keeps
// 1.5e-6, and the NSF source turns that into 1.4e-3 ... 4e-3 RMS of PCM.  With float64 accumulation the only fp32 roundings left are the ones
// every fp32 implementation has (one per layer output, the norms).
// One wave = 16 frames x 32 channels (two 16 x 16 accumulators of 4 doubles per lane), four waves per workgroup, no LDS: operands are small
// and L2-resident, the extractor is launch-bound.
//   A (activations): lane l supplies x[frame l & 15][k], k <-> q = l >> 4;  B (weights): W[k][channel l & 15].  An instruction contracts 4 k
//   (one per q); a block of 16 input channels is walked in 4 instructions, instruction j taking channels 4 q + j: lane (., q) needs channels
//   4 q .. 4 q + 3 of the block -- ONE 16-byte load of activations per block, converted to float64 in registers; weights are packed as doubles.
//   D: lane l holds channel l & 15 of frames 4 i + (l >> 4), i = 0..3 -- the f64 instruction interleaves the rows over the lane groups, unlike
//   the fp32 ones (measured: tools/micro/f64_probe.hip, profiles/r5f_f64_probe.txt; there is no ISA document in the image).
template <class Epi>
__global__ void __launch_bounds__(256) k_conv_f64acc(const float* __restrict__ x, int ldx, int n_rows, int cin_pad, int taps, const double* __restrict__ w,
                                                      int n_cpairs, const typename Epi::Args e) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int rt = wid / n_cpairs, cp = wid - rt * n_cpairs;
    const int row0 = rt * 16;
    if (row0 >= n_rows) return;
    const int r = lane & 15, q = lane >> 4;
    const int nkb = cin_pad >> 4, pad = taps >> 1;
    pe_f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    const size_t tile_doubles = (size_t)taps * nkb * 256;              // one 16-channel column tile
    const double* wp = w + (size_t)(2 * cp) * tile_doubles + lane * 4;
    for (int tap = 0; tap < taps; ++tap) {
        const int src = row0 + r + tap - pad;
        // a tap outside its clip's T frames is the conv's zero padding.  Tested here, not left to what the gap rows hold: LayerNorm writes
        // beta into them (the row kernels run over whole slots), and a slot ends in >= taps/2 gap rows, so no tap reaches another clip's frames
        const bool ok = src >= 0 && src < n_rows && (src - (src / e.Ts) * e.Ts) < e.T;
        const float* xp = x + (size_t)(ok ? src : 0) * ldx + 4 * q;
        const double* wt = wp + (size_t)tap * nkb * 256;
#pragma unroll 2
        for (int kb = 0; kb < nkb; ++kb) {
            pe_f32x4 a = *reinterpret_cast<const pe_f32x4*>(xp + 16 * kb);
            if (!ok) a = pe_f32x4{0.f, 0.f, 0.f, 0.f};
            const pe_f64x4 b0 = *reinterpret_cast<const pe_f64x4*>(wt + (size_t)kb * 256);
            const pe_f64x4 b1 = *reinterpret_cast<const pe_f64x4*>(wt + tile_doubles + (size_t)kb * 256);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double ad = (double)a[j];
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ad, b0[j], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ad, b1[j], acc1, 0, 0, 0);
            }
        }
    }
    Epi epi;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        epi.one(e, row0 + 4 * i + q, (2 * cp) * 16 + r, (float)acc0[i]);
        epi.one(e, row0 + 4 * i + q, (2 * cp + 1) * 16 + r, (float)acc1[i]);
    }
}

}  // namespace

// =================================================================================================
struct dsvc_pe {
    dsvc_pe_cfg cfg;
    std::map<std::string, std::vector<float>> host;
    bool finalized = false;
    int Mp = 0;
    struct Pre { Packed w; DevBuf b, scale, shift; };
    std::vector<Pre> pre;
    Packed pre_out_w, enc_in_w, enc_out_w, lin_w;
    DevBuf pre_out_b, enc_in_b, enc_out_b, lin_b, alpha, table;
    int table_rows = 0;
    struct Enc { Packed w; DevBuf b, g, be; };
    std::vector<Enc> enc;
    struct Pred { Packed w; DevBuf b, g, be; };
    std::vector<Pred> pred;
    // workspace
    long long wsB = -1, wsT = -1;
    DevBuf xm, keep, a, b2, gsum, posi, outp;

    ~dsvc_pe() {
        for (DevBuf* d : {&pre_out_b, &enc_in_b, &enc_out_b, &lin_b, &alpha, &table, &xm, &keep, &a, &b2, &gsum, &posi, &outp}) d->release();
        for (Packed* p : {&pre_out_w, &enc_in_w, &enc_out_w, &lin_w}) p->w.release();
        for (auto& p : pre) { p.w.w.release(); p.b.release(); p.scale.release(); p.shift.release(); }
        for (auto& p : enc) { p.w.w.release(); p.b.release(); p.g.release(); p.be.release(); }
        for (auto& p : pred) { p.w.w.release(); p.b.release(); p.g.release(); p.be.release(); }
    }
    const std::vector<float>* get(const std::string& k, size_t numel) {
        auto it = host.find(k);
        if (it == host.end()) { fail(DSVC_ESTATE, "pe: tensor '%s' was never loaded", k.c_str()); return nullptr; }
        if (it->second.size() != numel) { fail(DSVC_EINVAL, "pe: tensor '%s' has %zu elements, expected %zu", k.c_str(), it->second.size(), numel); return nullptr; }
        return &it->second;
    }
    int up(DevBuf& d, const float* src, size_t numel) {
        DSVC_TRY(d.alloc(numel * 4));
        DSVC_HIP(hipMemcpy(d.p, src, numel * 4, hipMemcpyHostToDevice));
        return DSVC_OK;
    }
    int up(DevBuf& d, const std::string& k, size_t numel) {
        const std::vector<float>* v = get(k, numel);
        if (!v) return DSVC_ESTATE;
        return up(d, v->data(), numel);
    }
    // torch Conv1d weight [cout][cin][k] (k = 1 for Linear [cout][cin]) -> float64 B fragments (k_conv_f64acc): lane (col, q) of block kb holds
    // W[col][16 kb + 4 q + j][tap], j = 0..3; column tiles of 16, an even number of them (a wave takes two)
    int pack(Packed& pk, const std::string& key, int cout, int cin, int k) {
        const std::vector<float>* w = get(key, (size_t)cout * cin * k);
        if (!w) return DSVC_ESTATE;
        pk.n_ctiles = round_up(ceil_div(cout, 16), 2); pk.taps = k; pk.cin_pad = round_up(cin, 16);
        const int nkb = pk.cin_pad / 16;
        std::vector<double> f((size_t)pk.n_ctiles * k * nkb * 256, 0.0);
        for (int ct = 0; ct < pk.n_ctiles; ++ct)
            for (int tap = 0; tap < k; ++tap)
                for (int kb = 0; kb < nkb; ++kb)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 4; ++j) {
                            const int col = ct * 16 + (l & 15), ci = 16 * kb + 4 * (l >> 4) + j;
                            if (col < cout && ci < cin) f[((((size_t)ct * k + tap) * nkb + kb) * 64 + l) * 4 + j] = (double)(*w)[((size_t)col * cin + ci) * k + tap];
                        }
        DSVC_TRY(pk.w.alloc(f.size() * 8));
        DSVC_HIP(hipMemcpy(pk.w.p, f.data(), f.size() * 8, hipMemcpyHostToDevice));
        return DSVC_OK;
    }
    int slot_rows(int T) const {
        const int halo = (cfg.kernel > cfg.predictor_kernel ? cfg.kernel : cfg.predictor_kernel) / 2;
        return T + halo < 32 ? 32 : T + halo;
    }
    int finalize();
    int ensure_ws(int B, int T);
    int run(const float* mel, int B, int T, float* pitch_pred, float* f0, hipStream_t st);
};

int dsvc_pe::finalize() {
    const int M = cfg.n_mel, H = cfg.hidden, P = cfg.predictor_hidden, K = cfg.kernel, PK = cfg.predictor_kernel;
    Mp = round_up(M, 16);
    pre.resize(cfg.prenet_layers);
    for (int l = 0; l < cfg.prenet_layers; ++l) {
        const std::string q = "mel_prenet.layers." + std::to_string(l) + ".";
        DSVC_TRY(pack(pre[l].w, q + "0.weight", H, l == 0 ? M : H, K));
        DSVC_TRY(up(pre[l].b, q + "0.bias", H));
        // eval BatchNorm1d as an affine: scale = gamma / sqrt(running_var + 1e-5), shift = beta - running_mean * scale
        const std::vector<float>* g = get(q + "2.weight", H); const std::vector<float>* be = get(q + "2.bias", H);
        const std::vector<float>* mu = get(q + "2.running_mean", H); const std::vector<float>* var = get(q + "2.running_var", H);
        if (!g || !be || !mu || !var) return DSVC_ESTATE;
        std::vector<float> sc(H), sh(H);
        for (int c = 0; c < H; ++c) {
            const double s = (double)(*g)[c] / sqrt((double)(*var)[c] + 1e-5);
            sc[c] = (float)s; sh[c] = (float)((double)(*be)[c] - (double)(*mu)[c] * s);
        }
        DSVC_TRY(up(pre[l].scale, sc.data(), H)); DSVC_TRY(up(pre[l].shift, sh.data(), H));
    }
    DSVC_TRY(pack(pre_out_w, "mel_prenet.out_proj.weight", H, H, 1)); DSVC_TRY(up(pre_out_b, "mel_prenet.out_proj.bias", H));
    enc.resize(cfg.conv_layers);
    if (cfg.conv_layers > 0) {
        DSVC_TRY(pack(enc_in_w, "mel_encoder.in_proj.weight", H, H, 1)); DSVC_TRY(up(enc_in_b, "mel_encoder.in_proj.bias", H));
        for (int l = 0; l < cfg.conv_layers; ++l) {
            const std::string q = "mel_encoder.conv." + std::to_string(l) + ".";
            DSVC_TRY(pack(enc[l].w, q + "conv.conv.weight", H, H, K)); DSVC_TRY(up(enc[l].b, q + "conv.conv.bias", H));
            DSVC_TRY(up(enc[l].g, q + "norm.weight", H)); DSVC_TRY(up(enc[l].be, q + "norm.bias", H));
        }
        DSVC_TRY(pack(enc_out_w, "mel_encoder.out_proj.weight", H, H, 1)); DSVC_TRY(up(enc_out_b, "mel_encoder.out_proj.bias", H));
    }
    pred.resize(cfg.predictor_layers);
    for (int l = 0; l < cfg.predictor_layers; ++l) {
        const std::string q = "pitch_predictor.conv." + std::to_string(l) + ".";
        DSVC_TRY(pack(pred[l].w, q + "1.weight", P, l == 0 ? H : P, PK)); DSVC_TRY(up(pred[l].b, q + "1.bias", P));
        DSVC_TRY(up(pred[l].g, q + "3.weight", P)); DSVC_TRY(up(pred[l].be, q + "3.bias", P));
    }
    DSVC_TRY(pack(lin_w, "pitch_predictor.linear.weight", 2, P, 1)); DSVC_TRY(up(lin_b, "pitch_predictor.linear.bias", 2));
    DSVC_TRY(up(alpha, "pitch_predictor.pos_embed_alpha", 1));
    host.clear();
    finalized = true;
    return DSVC_OK;
}

int dsvc_pe::ensure_ws(int B, int T) {
    if (B == wsB && T == wsT) return DSVC_OK;
    const int Ts = slot_rows(T);
    const size_t rows = (size_t)round_up(B * Ts, 128) + 128;
    const int Cmax = cfg.hidden > cfg.predictor_hidden ? cfg.hidden : cfg.predictor_hidden;
    DSVC_TRY(xm.alloc(rows * Mp * 4)); DSVC_TRY(keep.alloc(rows * 4));
    DSVC_TRY(a.alloc(rows * Cmax * 4)); DSVC_TRY(b2.alloc(rows * Cmax * 4));
    DSVC_TRY(gsum.alloc((size_t)B * (cfg.hidden / 16 + 1) * 2 * 8));
    DSVC_TRY(posi.alloc(rows * 4)); DSVC_TRY(outp.alloc(rows * 64 * 4));
    wsB = B; wsT = T;
    return DSVC_OK;
}

int dsvc_pe::run(const float* mel, int B, int T, float* pitch_pred, float* f0, hipStream_t st) {
    DSVC_TRY(ensure_ws(B, T));
    const int M = cfg.n_mel, H = cfg.hidden, P = cfg.predictor_hidden;
    const int Ts = slot_rows(T);
    const int rows = B * Ts;
    if (T + 1 > table_rows) return fail(DSVC_ESTATE, "pe: %d frames need %d position rows, dsvc_pe_set_positions gave %d", T, T + 1, table_rows);
    auto gemm = [&](const float* x, int ldx, int cin, const Packed& pk, const EpPe::Args& e) -> int {
        if (ldx % 4 || round_up(cin, 16) != pk.cin_pad || pk.cin_pad > ldx) return fail(DSVC_EINVAL, "pe: operand rows of %d floats (stride %d) against weights packed for %d", cin, ldx, pk.cin_pad);
        const int waves = ceil_div(rows, 16) * (pk.n_ctiles / 2);
        hipLaunchKernelGGL(k_conv_f64acc<EpPe>, dim3(ceil_div(waves, 4)), dim3(256), 0, st, x, ldx, rows, pk.cin_pad, pk.taps, pk.w.as<double>(), pk.n_ctiles / 2, e);
        DSVC_HIP(hipGetLastError());
        return DSVC_OK;
    };
    float* A = a.as<float>();
    float* Bf = b2.as<float>();
    const float* kp = keep.as<float>();
    // ---- Prenet (pe.py:23-42): 3 x [Conv1d k5 -> ReLU -> BatchNorm1d(eval)] * nonpadding, out_proj * nonpadding ----
    hipLaunchKernelGGL(k_pe_stage_mel, dim3(ceil_div(rows, 4)), dim3(256), 0, st, mel, xm.as<float>(), keep.as<float>(), B, T, Ts, M, Mp);
    const float* cur = xm.as<float>();
    int ld = Mp;
    for (size_t l = 0; l < pre.size(); ++l) {
        float* dst = (cur == A) ? Bf : A;
        EpPe::Args e{dst, H, pre[l].b.as<float>(), H, 1, pre[l].scale.as<float>(), pre[l].shift.as<float>(), kp, Ts, T};
        DSVC_TRY(gemm(cur, ld, ld, pre[l].w, e));
        cur = dst; ld = H;
    }
    {
        float* dst = (cur == A) ? Bf : A;
        EpPe::Args e{dst, H, pre_out_b.as<float>(), H, 0, nullptr, nullptr, kp, Ts, T};
        DSVC_TRY(gemm(cur, ld, ld, pre_out_w, e));
        cur = dst;
    }
    // ---- ConvStacks (pe.py:98-117): in_proj, L x [x += relu(GroupNorm(conv k5(x)))], out_proj ----
    if (cfg.conv_layers > 0) {
        float* x = (cur == A) ? Bf : A;
        float* y = (x == A) ? Bf : A;                                 // == cur's buffer, free after in_proj
        {
            EpPe::Args e{x, H, enc_in_b.as<float>(), H, 0, nullptr, nullptr, nullptr, Ts, T};
            DSVC_TRY(gemm(cur, H, H, enc_in_w, e));
        }
        const int G = H / 16, gs = 16;
        for (size_t l = 0; l < enc.size(); ++l) {
            EpPe::Args e{y, H, enc[l].b.as<float>(), H, 0, nullptr, nullptr, nullptr, Ts, T};
            DSVC_TRY(gemm(x, H, H, enc[l].w, e));
            DSVC_HIP(hipMemsetAsync(gsum.p, 0, (size_t)B * G * 2 * 8, st));
            hipLaunchKernelGGL(k_pe_gn_stats, dim3(ceil_div(T, 64), B), dim3(256), 0, st, y, gsum.as<double>(), T, Ts, H, gs, 64);
            hipLaunchKernelGGL(k_pe_gn_apply, dim3(2048), dim3(256), 0, st, x, y, gsum.as<double>(), enc[l].g.as<float>(), enc[l].be.as<float>(), B, T, Ts, H, gs);
        }
        {
            EpPe::Args e{y, H, enc_out_b.as<float>(), H, 0, nullptr, nullptr, nullptr, Ts, T};
            DSVC_TRY(gemm(x, H, H, enc_out_w, e));
        }
        cur = y;
    }
    // ---- PitchPredictor (tts_modules.py:222-235): x += alpha * sinusoid[positions]; 5 x [conv k5 -> ReLU -> LayerNorm(C, eps 1e-12)]; Linear -> 2 ----
    float* x = const_cast<float*>(cur);
    float* y = (x == A) ? Bf : A;
    hipLaunchKernelGGL(k_pe_positions, dim3(B), dim3(64), 0, st, x, posi.as<int>(), T, Ts, H);
    hipLaunchKernelGGL(k_pe_add_pos, dim3(2048), dim3(256), 0, st, x, posi.as<int>(), table.as<float>(), alpha.as<float>(), B, T, Ts, H);
    int cin = H;
    for (size_t l = 0; l < pred.size(); ++l) {
        EpPe::Args e{y, P, pred[l].b.as<float>(), P, 1, nullptr, nullptr, nullptr, Ts, T};
        DSVC_TRY(gemm(x, cin, cin, pred[l].w, e));
        hipLaunchKernelGGL(k_layernorm, dim3(ceil_div(rows, 4)), dim3(256), 0, st, y, y, pred[l].g.as<float>(), pred[l].be.as<float>(), rows, P, 1e-12f);
        float* t = x; x = y; y = t;
        cin = P;
    }
    {
        EpPe::Args e{outp.as<float>(), 64, lin_b.as<float>(), 2, 0, nullptr, nullptr, nullptr, Ts, T};
        DSVC_TRY(gemm(x, P, P, lin_w, e));
    }
    hipLaunchKernelGGL(k_pe_finish, dim3(ceil_div(B * T, 256)), dim3(256), 0, st, outp.as<float>(), 64, kp, pitch_pred, f0, B, T, Ts, cfg.pitch_norm, cfg.f0_mean,
                       cfg.f0_std, cfg.use_uv);
    DSVC_HIP(hipGetLastError());
    return DSVC_OK;
}

// =================================================================================================
extern "C" {

int dsvc_pe_create(const dsvc_pe_cfg* cfg, dsvc_pe** out) {
    if (!cfg || !out) return fail(DSVC_EINVAL, "null argument");
    if (cfg->n_mel < 1 || cfg->hidden < 16 || cfg->hidden % 16 || cfg->predictor_hidden < 16 || cfg->predictor_hidden % 16)
        return fail(DSVC_EINVAL, "pe: n_mel %d, hidden %d, predictor_hidden %d (the channel counts must be multiples of 16)", cfg->n_mel, cfg->hidden, cfg->predictor_hidden);
    if (cfg->kernel % 2 != 1 || cfg->predictor_kernel % 2 != 1 || cfg->kernel < 1 || cfg->predictor_kernel < 1)
        return fail(DSVC_EINVAL, "pe: odd kernel sizes only ('SAME' padding), got %d / %d", cfg->kernel, cfg->predictor_kernel);
    if (cfg->prenet_layers < 1 || cfg->conv_layers < 0 || cfg->predictor_layers < 1) return fail(DSVC_EINVAL, "pe: bad layer counts");
    if (cfg->pitch_norm < 0 || cfg->pitch_norm > 2) return fail(DSVC_EINVAL, "pe: pitch_norm must be 0 (log), 1 (standard) or 2 (none)");
    int ndev = 0;
    DSVC_HIP(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(DSVC_EHIP, "no HIP device visible");
    dsvc_pe* p = new dsvc_pe();
    p->cfg = *cfg;
    *out = p;
    return DSVC_OK;
}

int dsvc_pe_load_tensor(dsvc_pe* p, const char* name, const float* host, int64_t numel) {
    if (!p || !name || !host || numel < 0) return fail(DSVC_EINVAL, "null argument");
    if (p->finalized) return fail(DSVC_ESTATE, "pe already finalized");
    p->host[name].assign(host, host + numel);
    return DSVC_OK;
}

int dsvc_pe_finalize(dsvc_pe* p) {
    if (!p) return fail(DSVC_EINVAL, "null handle");
    if (p->finalized) return DSVC_OK;
    return p->finalize();
}

int dsvc_pe_set_positions(dsvc_pe* p, const float* host_table, int32_t n_rows) {
    if (!p || !host_table || n_rows < 1) return fail(DSVC_EINVAL, "bad argument");
    DSVC_HIP(hipDeviceSynchronize());
    DSVC_TRY(p->up(p->table, host_table, (size_t)n_rows * p->cfg.hidden));
    p->table_rows = n_rows;
    return DSVC_OK;
}

void dsvc_pe_destroy(dsvc_pe* p) { delete p; }

int dsvc_pe_run(dsvc_pe* p, const float* mel, int32_t B, int32_t T, float* pitch_pred, float* f0, void* stream) {
    if (!p || !mel || !f0) return fail(DSVC_EINVAL, "null argument");
    if (!p->finalized) return fail(DSVC_ESTATE, "pe not finalized");
    if (B < 1 || T < 1) return fail(DSVC_EINVAL, "pe: B %d, T %d", B, T);
    if ((long long)B * p->slot_rows(T) > (1ll << 30)) return fail(DSVC_EINVAL, "pe: batch too large");
    return p->run(mel, B, T, pitch_pred, f0, (hipStream_t)stream);
}

}  // extern "C"
