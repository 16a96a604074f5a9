// wgrad_fm_kernel (csrc/wgrad.h, round 5) on its own, without torch: (1) the lane mapping of ds_read_b64_tr_b16 the kernel relies on (there is no ISA
// document in the image), (2) the contraction from frame-major [hi | lo] planes -- three conv taps as row offsets + a second B source + the bias
// column sums -- against a float64 evaluation, (3) its time beside wgrad_nt_kernel's on fragment-tiled planes of the same problem.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o wgrad_fm_test wgrad_fm_test.hip ../../diff-svc_amd/csrc/common.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../diff-svc_amd/csrc/wgrad.h"

using namespace dsvc;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__global__ void k_tr_probe(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 4];
    const int l = threadIdx.x;
    for (int e = 0; e < 4; ++e) lds[l * 4 + e] = (unsigned short)(l * 4 + e);      // lane s's 8 bytes hold the values 4 s .. 4 s + 3
    __syncthreads();
    typedef wg_short4 __attribute__((address_space(3))) * lds_p;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)lds;
    const wg_short4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(size_t)(base + 8u * l));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}

// float64 evaluation: dw[o][k] = sum_n (ah + al)[n][o] * (bh + bl)[n + shift][k];  bias[o] = sum_n (ah + al)[n][o]
__global__ void k_ref(const _Float16* a, int a_ld, int a_lo, WgradFmSeg sg, int k_base, int n_total, int O, double* dw, int K, double* bias) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x, o = blockIdx.y;
    if (k >= sg.k_tiles * 128) return;
    double s = 0.0, sb = 0.0;
    for (int n = 0; n < n_total; ++n) {
        const double av = (double)(float)a[(size_t)n * a_ld + o] + (double)(float)a[(size_t)n * a_ld + a_lo + o];
        const _Float16* br = sg.b + (long long)(n + sg.shift) * sg.ld;
        const double bv = (double)(float)br[k] + (double)(float)br[sg.lo + k];
        s += av * bv; sb += av;
    }
    dw[(size_t)o * K + k_base + k] = s;
    if (bias && k == 0 && k_base == 0) bias[o] = sb;
}

static float frand() { return (float)((rand() & 0xffff) - 32768) / 32768.0f; }

int main() {
    // ---- (1) the transposing read ----
    {
        unsigned short* d; CK(hipMalloc(&d, 64 * 4 * 2));
        hipLaunchKernelGGL(k_tr_probe, dim3(1), dim3(64), 0, 0, d);
        unsigned short h[256]; CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                const int src_lane = 16 * (l >> 4) + 4 * j + ((l & 15) >> 2), expect = src_lane * 4 + (l & 3);
                if (h[l * 4 + j] != expect) ++bad;
            }
        printf("ds_read_b64_tr_b16: lane i of a 16-lane group receives element (i & 3) of the 8 bytes lanes 4 j + (i >> 2) addressed, j = 0..3: %s (%d of 256 differ)\n",
               bad ? "NO" : "yes", bad);
        for (int l = 0; l < 64; l += 7) printf("  lane %2d: %3d %3d %3d %3d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
        if (bad) return 1;
    }
    // ---- (2) parity ----
    const int GUARD = 64, rows = 2176, C = 384, O = 768, H = 256, dil = 4;           // (rows: 16 clips of 136)
    const int a_ld = 2 * O, a_lo = O, x_ld = 2 * C, x_lo = C, c_ld = 2 * H, c_lo = H;
    std::vector<_Float16> ha((size_t)rows * a_ld), hx((size_t)(rows + 2 * GUARD) * x_ld), hc((size_t)rows * c_ld);
    srand(7);
    for (size_t i = 0; i < ha.size(); ++i) {
        const bool lo = (i % a_ld) >= (size_t)a_lo, gap = (i / a_ld) % 136 >= 128;         // clips of 128 frames, 136 rows apart; gap rows are zero as in the trainer's planes
        ha[i] = gap ? (_Float16)0.f : (_Float16)(frand() * (lo ? 4e-4f : 1.0f));
    }
    for (size_t i = 0; i < hx.size(); ++i) { const bool lo = (i % x_ld) >= (size_t)x_lo; hx[i] = (_Float16)(frand() * (lo ? 4e-4f : 1.0f)); }
    for (size_t i = 0; i < hc.size(); ++i) { const bool lo = (i % c_ld) >= (size_t)c_lo; hc[i] = (_Float16)(frand() * (lo ? 4e-4f : 1.0f)); }
    _Float16 *da, *dx, *dc;
    CK(hipMalloc(&da, ha.size() * 2)); CK(hipMalloc(&dx, hx.size() * 2)); CK(hipMalloc(&dc, hc.size() * 2));
    CK(hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dc, hc.data(), hc.size() * 2, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)wgrad_fm_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, WG_STAGES * WG_STAGE_BYTES));
    CK(hipFuncSetAttribute((const void*)wgrad_fm_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, WG_STAGES * WG_STAGE_BYTES));
    CK(hipFuncSetAttribute((const void*)wgrad_nt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WG_STAGES * WG_STAGE_BYTES));

    // one launch of one or two contractions (the second, if any: the conditioner-like source `cond2` against the same A), checked against float64
    auto run_case = [&](const char* name, int n_seg, const WgradFmSeg* segs, int n_rows, int S, bool with_bias, int reps, int spc = 0, int clip_rows = 0,
                        const WgradFmSeg* second = nullptr) -> int {
        const int n_prob = second ? 2 : 1;
        const WgradFmSeg* sgs[2] = {segs, second};
        const int nsg[2] = {n_seg, 1};
        int kt[2] = {0, 0}, tiles[2] = {0, 0};
        for (int q = 0; q < n_prob; ++q) { for (int s = 0; s < nsg[q]; ++s) kt[q] += sgs[q][s].k_tiles; tiles[q] = 3 * kt[q]; }
        const int O_pad = 768, tsum = tiles[0] + tiles[1];
        WgradFmArgs a{};
        a.n_prob = n_prob; a.wgs0 = S * tiles[0]; a.n_stages = spc ? (n_rows / clip_rows) * spc : n_rows / 32; a.clip_rows = clip_rows; a.spc = spc; a.xcd_map = S % 8 == 0;
        float *part, *bpart, *dw[2], *db;
        CK(hipMalloc(&part, (size_t)S * tsum * 256 * 128 * 4)); CK(hipMalloc(&bpart, (size_t)S * tsum * 256 * 4)); CK(hipMalloc(&db, O * 4));
        float* pp = part; float* bp = bpart;
        for (int q = 0; q < n_prob; ++q) {
            WgradFmProb& P = a.p[q];
            P.a = da; P.a_ld = a_ld; P.a_lo = a_lo; P.n_seg = nsg[q];
            for (int s = 0; s < nsg[q]; ++s) P.seg[s] = sgs[q][s];
            P.slices = S; P.slice_stages = ceil_div(a.n_stages, S); P.part = pp; P.O_pad = O_pad; P.K_pad = kt[q] * 128; P.tiles = tiles[q];
            P.bias_part = (with_bias && q == 0) ? bp : nullptr;
            pp += (size_t)S * tiles[q] * 256 * 128; bp += (size_t)S * tiles[q] * 256;
            CK(hipMalloc(&dw[q], (size_t)O * P.K_pad * 4));
        }
        auto launch = [&]() {
            if (with_bias) hipLaunchKernelGGL(wgrad_fm_kernel<true>, dim3(tsum * S), dim3(512), WG_STAGES * WG_STAGE_BYTES, 0, a);
            else hipLaunchKernelGGL(wgrad_fm_kernel<false>, dim3(tsum * S), dim3(512), WG_STAGES * WG_STAGE_BYTES, 0, a);
        };
        launch();
        for (int q = 0; q < n_prob; ++q) {
            const WgradFmProb& P = a.p[q];
            WgradSegs rs{}; rs.n = 1; rs.s[0] = WgradSeg{dw[q], 0, P.K_pad, (long long)P.K_pad, 1, 0};
            hipLaunchKernelGGL(k_wgrad_nt_reduce, dim3(ceil_div(P.K_pad, 1024) + (P.bias_part ? 1 : 0), O), dim3(256), 0, 0, P.part, S, O_pad, P.K_pad, O, rs, 1.0f,
                               (const float*)P.bias_part, S * kt[q], db, (float*)nullptr);
        }
        CK(hipDeviceSynchronize());
        int rc = 0;
        if (n_rows <= 4096) {            // float64 check (the gap rows of A are zero, so the sum over all rows is the sum over the clips' frames)
            for (int q = 0; q < n_prob; ++q) {
                const int K_pad = a.p[q].K_pad;
                double *rw, *rb;
                CK(hipMalloc(&rw, (size_t)O * K_pad * 8)); CK(hipMalloc(&rb, O * 8));
                int kb = 0;
                for (int s = 0; s < nsg[q]; ++s) {
                    hipLaunchKernelGGL(k_ref, dim3(ceil_div(sgs[q][s].k_tiles * 128, 128), O), dim3(128), 0, 0, da, a_ld, a_lo, sgs[q][s], kb, n_rows, O, rw, K_pad, rb);
                    kb += sgs[q][s].k_tiles * 128;
                }
                CK(hipDeviceSynchronize());
                std::vector<float> w((size_t)O * K_pad), b(O);
                std::vector<double> w64((size_t)O * K_pad), b64(O);
                CK(hipMemcpy(w.data(), dw[q], w.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(w64.data(), rw, w64.size() * 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(b.data(), db, O * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b64.data(), rb, O * 8, hipMemcpyDeviceToHost));
                const bool chk_bias = with_bias && q == 0;
                double num = 0, den = 0, worst = 0, bnum = 0, bden = 0;
                for (size_t i = 0; i < w.size(); ++i) { const double d = w[i] - w64[i]; num += d * d; den += w64[i] * w64[i]; if (fabs(d) > worst) worst = fabs(d); }
                if (chk_bias) for (int o = 0; o < O; ++o) { const double d = b[o] - b64[o]; bnum += d * d; bden += b64[o] * b64[o]; }
                const double rel = sqrt(num / den), brel = chk_bias ? sqrt(bnum / bden) : 0.0;
                printf("%s [%d]: %d rows (%d stages), %d slices, K_pad %d: rel-L2 vs float64 %.3e (worst |d| %.3e at rms %.3e)\n", name, q, n_rows, a.n_stages, S, K_pad, rel,
                       worst, sqrt(den / w.size()));
                if (chk_bias) printf("    bias column sums: rel-L2 %.3e\n", brel);
                if (!(rel < 2e-6) || (chk_bias && !(brel < 2e-6))) { printf("    FAILED\n"); rc = 1; }
                CK(hipFree(rw)); CK(hipFree(rb));
            }
        }
        if (reps > 0) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int i = 0; i < 3; ++i) launch();
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) launch();
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double flop = 2.0 * a.n_stages * 32 * O_pad * (kt[0] + kt[1]) * 128 * 3;
            printf("%s: wgrad_fm_kernel %.1f us per launch (%d workgroups), %.0f TFLOP/s MFMA-equivalent\n", name, ms * 1000 / reps, tsum * S, flop / (ms / reps * 1e-3) / 1e12);
        }
        CK(hipFree(part)); CK(hipFree(bpart)); CK(hipFree(db));
        for (int q = 0; q < n_prob; ++q) CK(hipFree(dw[q]));
        return rc;
    };
    int rc = 0;
    const _Float16* x0 = dx + (size_t)GUARD * x_ld;
    WgradFmSeg taps[3] = {{x0, x_ld, x_lo, -dil, 3}, {x0, x_ld, x_lo, 0, 3}, {x0, x_ld, x_lo, dil, 3}};
    WgradFmSeg cond[1] = {{dc, c_ld, c_lo, 0, 2}};
    WgradFmSeg outp[1] = {{x0, x_ld, x_lo, 0, 3}};
    WgradFmSeg mixed[2] = {{x0, x_ld, x_lo, 8, 2}, {dc, c_ld, c_lo, 0, 1}};
    rc |= run_case("conv taps (3 x 384), bias", 3, taps, rows, 8, true, 0);
    rc |= run_case("cond (256)", 1, cond, rows, 40, false, 0);
    rc |= run_case("out projection (384), bias", 1, outp, rows, 24, true, 0);
    rc |= run_case("two sources, ragged slices", 2, mixed, rows - 64, 5, true, 0);
    rc |= run_case("conv taps, gap rows skipped", 3, taps, rows, 8, true, 0, 4, 136);
    rc |= run_case("out projection + cond in one launch, gap rows skipped", 1, outp, rows, 16, true, 0, 4, 136, cond);
    rc |= run_case("out projection + cond in one launch, 3 slices each", 1, outp, rows, 3, true, 0, 0, 0, cond);
    // ---- (3) time at the benchmarked size (64 clips x 136 rows) ----
    {
        const int big = 8704;
        CK(hipFree(da)); CK(hipFree(dx)); CK(hipFree(dc));
        CK(hipMalloc(&da, (size_t)big * a_ld * 2)); CK(hipMalloc(&dx, (size_t)(big + 2 * GUARD) * x_ld * 2)); CK(hipMalloc(&dc, (size_t)big * c_ld * 2));
        CK(hipMemset(da, 0x11, (size_t)big * a_ld * 2)); CK(hipMemset(dx, 0x12, (size_t)(big + 2 * GUARD) * x_ld * 2)); CK(hipMemset(dc, 0x13, (size_t)big * c_ld * 2));
        const _Float16* xb = dx + (size_t)GUARD * x_ld;
        WgradFmSeg t3[3] = {{xb, x_ld, x_lo, -dil, 3}, {xb, x_ld, x_lo, 0, 3}, {xb, x_ld, x_lo, dil, 3}};
        WgradFmSeg c1[1] = {{dc, c_ld, c_lo, 0, 2}};
        WgradFmSeg o1[1] = {{xb, x_ld, x_lo, 0, 3}};
        run_case("conv taps @ 8704 rows", 3, t3, big, 8, true, 50);
        run_case("conv taps @ 8192 frames of 8704 rows", 3, t3, big, 8, true, 50, 4, 136);
        run_case("cond @ 8704 rows", 1, c1, big, 40, false, 50);
        run_case("out projection @ 8704 rows", 1, o1, big, 24, true, 50);
        run_case("out projection + cond @ 8192 frames, one launch", 1, o1, big, 16, true, 50, 4, 136, c1);
        // wgrad_nt_kernel on fragment-tiled planes of the same problem (8192 real frames), for the kernel-to-kernel comparison
        const int ldT = 8192, a_rows = 768, b_rows = 3 * 384 + 256;
        _Float16 *at, *bt; float* part;
        CK(hipMalloc(&at, (size_t)2 * a_rows * ldT * 2)); CK(hipMalloc(&bt, (size_t)2 * b_rows * ldT * 2)); CK(hipMalloc(&part, (size_t)288 * 256 * 128 * 4));
        CK(hipMemset(at, 0x11, (size_t)2 * a_rows * ldT * 2)); CK(hipMemset(bt, 0x12, (size_t)2 * b_rows * ldT * 2));
        struct { const char* name; int K_pad, S, b_row0; } old[3] = {{"conv taps", 1152, 8, 0}, {"cond", 256, 40, 1152}, {"out projection", 384, 24, 0}};
        for (auto& c : old) {
            WgradNtArgs w{};
            w.at = at; w.bt = bt + (size_t)c.b_row0 * ldT; w.a_plane = (long long)a_rows * ldT; w.b_plane = (long long)b_rows * ldT; w.ldT = ldT; w.n_total = ldT;
            w.slice_len = round_up(ceil_div(ldT, c.S), 32); w.part = part; w.O_pad = 768; w.K_pad = c.K_pad; w.tiles = 3 * (c.K_pad / 128); w.xcd_map = 1;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(wgrad_nt_kernel, dim3(w.tiles * c.S), dim3(512), WG_STAGES * WG_STAGE_BYTES, 0, w);
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(wgrad_nt_kernel, dim3(w.tiles * c.S), dim3(512), WG_STAGES * WG_STAGE_BYTES, 0, w);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s @ 8192 frames: wgrad_nt_kernel %.1f us per launch (%d workgroups)\n", c.name, ms * 1000 / 50, w.tiles * c.S);
        }
    }
    printf(rc ? "FAILED\n" : "all cases passed\n");
    return rc;
}
