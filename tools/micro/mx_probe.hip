// Round 4 probe: can the w_lo * x correction of the f16_w2 scheme run on the block-scaled fp8 / fp6 matrix instructions of gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/mx_probe tools/micro/mx_probe.hip && tools/micro/mx_probe
// Answers, each against a host emulation:
//   A  v_cvt_scalef32_pk_fp8_f16 / _pk32_bf6_f16 / _pk32_fp6_f16: rounding, scale direction, saturation, element order
//   B  v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 x fp8, bf6 x bf6, fp6 x fp6): which byte / bit field pairs with which, where the per-lane
//      E8M0 scale applies, C/D layout
//   C  issue rates on all 256 CUs (2 waves per SIMD, random operands): 32 f16 MFMAs per group (today's f16_w2 gate loop) against
//      16 f16 + 4 fp8 (K = 64) and 16 f16 + 4 fp6, with and without the in-register fp16 -> fp8 / fp6 conversion of the B operand
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half32 __attribute__((ext_vector_type(32)));
typedef short short2_t __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v6i __attribute__((ext_vector_type(6)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// ---------------- host decoders ----------------
static float dec_e4m3(unsigned b) {
    const int s = (b >> 7) & 1, e = (b >> 3) & 15, m = b & 7;
    float v;
    if (e == 15 && m == 7) return NAN;
    if (e == 0) v = ldexpf((float)m / 8.f, -6); else v = ldexpf(1.f + (float)m / 8.f, e - 7);
    return s ? -v : v;
}
static float dec_e2m3(unsigned b) {                 // fp6
    const int s = (b >> 5) & 1, e = (b >> 3) & 3, m = b & 7;
    float v = e == 0 ? (float)m / 8.f : ldexpf(1.f + (float)m / 8.f, e - 1);
    return s ? -v : v;
}
static float dec_e3m2(unsigned b) {                 // bf6
    const int s = (b >> 5) & 1, e = (b >> 2) & 7, m = b & 3;
    float v = e == 0 ? ldexpf((float)m / 4.f, -2) : ldexpf(1.f + (float)m / 4.f, e - 3);
    return s ? -v : v;
}
static unsigned get6(const unsigned* w, int m) {    // 6-bit field m of a little-endian bit stream
    const int bit = 6 * m, q = bit >> 5, r = bit & 31;
    unsigned long long two = (unsigned long long)w[q] | ((unsigned long long)(q + 1 < 8 ? w[q + 1] : 0u) << 32);
    return (unsigned)((two >> r) & 63u);
}

// ---------------- A: conversions ----------------
__global__ void k_cvt8(const _Float16* __restrict__ src, unsigned* __restrict__ out, float scale) {
    const int l = threadIdx.x;
    half2_t v = {src[2 * l], src[2 * l + 1]};
    short2_t o = {0, 0};
    o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(o, v, scale, false);
    short2_t p = {0, 0};
    p = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p, v, scale, true);
    out[2 * l] = (unsigned)(unsigned short)o[0] | ((unsigned)(unsigned short)o[1] << 16);
    out[2 * l + 1] = (unsigned)(unsigned short)p[0] | ((unsigned)(unsigned short)p[1] << 16);
}
__global__ void k_cvt6(const _Float16* __restrict__ src, unsigned* __restrict__ out, float scale, int bf) {
    const int l = threadIdx.x;
    half32 v;
    for (int i = 0; i < 32; ++i) v[i] = src[l * 32 + i];
    v6i o = bf ? __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(v, scale) : __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v, scale);
    for (int i = 0; i < 6; ++i) out[l * 6 + i] = (unsigned)o[i];
}

// ---------------- B: the scaled MFMA ----------------
template <int FMT>
__global__ void k_mx(const v8i* __restrict__ a, const v8i* __restrict__ b, const int* __restrict__ sa, const int* __restrict__ sb, f32x16* __restrict__ d) {
    const int l = threadIdx.x;
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], c, FMT, FMT, 0, sa[l], 0, sb[l]);
    d[l] = c;
}

template <int FA, int FB>
__global__ void k_mx2(const v8i* __restrict__ a, const v8i* __restrict__ b, const int* __restrict__ sa, const int* __restrict__ sb, f32x16* __restrict__ d) {
    const int l = threadIdx.x;
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], c, FA, FB, 0, sa[l], 0, sb[l]);
    d[l] = c;
}

// D: the sequence the layer kernel would run: four fp16 B fragments of a lane (k = 16 kk + 8 (lane >> 5) + e) -> ONE pk32 conversion to bf6 ->
// K = 64 product with fp6 weight codes (A), uniform E8M0 scales
__global__ void k_inreg(const int* __restrict__ a6, const half8* __restrict__ hb, f32x16* __restrict__ d, float xscale, int sa, int sb) {
    const int l = threadIdx.x;
    half32 v;
    for (int kk = 0; kk < 4; ++kk) {
        const half8 f = hb[kk * 64 + l];
        for (int e = 0; e < 8; ++e) v[8 * kk + e] = f[e];
    }
    const v6i q = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(v, xscale);
    v6i w;                                              // (sizeof(v6i) is 32, not 24: index the packed stream by hand)
    for (int i = 0; i < 6; ++i) w[i] = a6[l * 6 + i];
    const v8i A = __builtin_shufflevector(w, w, 0, 1, 2, 3, 4, 5, -1, -1), B = __builtin_shufflevector(q, q, 0, 1, 2, 3, 4, 5, -1, -1);
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 2, 3, 0, sa, 0, sb);
    d[l] = c;
}

// ---------------- C: rates ----------------
// MODE 0: 32 f16 MFMAs per group.  1: 16 f16 + 4 fp8.  2: 16 f16 + 4 fp8 + 64 cvt.  3: 16 f16 + 4 bf6.  4: 16 f16 + 4 bf6 + 4 pk32 cvt.
// 5: 4 fp8 only.  6: 4 bf6 only.  7: 16 f16 only.
template <int MODE>
__global__ void __launch_bounds__(512, 2) k_rate(const _Float16* __restrict__ src, float* __restrict__ sink, int iters) {
    const int lane = threadIdx.x & 63;
    half8 a[4], b[4][4];
    v8i a8, b8[4];
    const half8* s8 = reinterpret_cast<const half8*>(src) + (size_t)blockIdx.x * 64 * 32 + lane;
    for (int i = 0; i < 4; ++i) a[i] = s8[64 * i];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) b[i][j] = s8[64 * (4 + 4 * i + j)];
    {
        const v8i* si = reinterpret_cast<const v8i*>(src) + (size_t)blockIdx.x * 64 * 16 + lane;
        a8 = si[64 * 10];
        for (int j = 0; j < 4; ++j) { b8[j] = si[64 * (11 + j)]; }
        for (int q = 0; q < 8; ++q) {                     // keep every byte a finite fp8 / fp6 pattern with a small exponent
            a8[q] &= 0x37373737; for (int j = 0; j < 4; ++j) b8[j][q] &= 0x37373737;
        }
    }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 2 || MODE == 4) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) asm volatile("" : "+v"(b[kk][nt]));      // "fresh" B fragments: the conversion cannot be hoisted
        }
        if constexpr (MODE != 5 && MODE != 6) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kk], b[kk][nt], acc[nt], 0, 0, 0);
                    if constexpr (MODE == 0) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(kk + 1) & 3], b[kk][nt], acc[nt], 0, 0, 0);
                    if constexpr (MODE == 2) {
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            half2_t v0 = {b[kk][nt][4 * p], b[kk][nt][4 * p + 1]}, v1 = {b[kk][nt][4 * p + 2], b[kk][nt][4 * p + 3]};
                            short2_t o = __builtin_bit_cast(short2_t, b8[nt][2 * kk + p]);
                            o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(o, v0, 1.0f, false);
                            o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(o, v1, 1.0f, true);
                            b8[nt][2 * kk + p] = __builtin_bit_cast(int, o);
                        }
                    }
                }
        }
        if constexpr (MODE == 4) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                half32 v;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[8 * kk + e] = b[kk][nt][e];
                v6i o = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(v, 1.0f);
#pragma unroll
                for (int q = 0; q < 6; ++q) b8[nt][q] = o[q];
            }
        }
        if constexpr (MODE == 1 || MODE == 2 || MODE == 5) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8[nt], acc[nt], 0, 0, 0, 127, 0, 127);
        }
        if constexpr (MODE == 3 || MODE == 4 || MODE == 6) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8[nt], acc[nt], 3, 3, 0, 127, 0, 127);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void rate(const char* what, const _Float16* d, float* sink, double ref_ms = 0) {
    const int iters = 4000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_rate<MODE>, dim3(256), dim3(512), 0, 0, d, sink, iters);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_rate<MODE>, dim3(256), dim3(512), 0, 0, d, sink, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    // per SIMD: 2 waves x iters groups
    printf("rate %-46s %8.3f ms   %7.1f ns per group per wave-pair\n", what, best, best * 1e6 / iters);
}

int main() {
    // ---- A ----
    {
        const float vals[] = {0.f, 0.1f, 0.5f, 1.0f, 1.0625f, 1.1875f, 3.f, 17.f, 100.f, 448.f, 464.f, 500.f, 1000.f, 60000.f, -2.3f, 1e-3f, 3e-3f, 0.0146f, -0.0009765625f,
                              7.5f, 28.f, 30.f, 0.03f, 0.06f, 0.12f, 0.3f, 5.1f, 6.9f, 11.f, 23.f, -27.f, 2.2f};
        const int n = sizeof(vals) / sizeof(float);
        std::vector<_Float16> h(128 * 32, (_Float16)0.f);
        for (int i = 0; i < n; ++i) h[i] = (_Float16)vals[i];
        _Float16* d; unsigned* o;
        CK(hipMalloc(&d, h.size() * 2)); CK(hipMalloc(&o, 64 * 8 * 4));
        CK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        for (float sc : {1.0f, 8.0f, 0.125f}) {
            hipLaunchKernelGGL(k_cvt8, dim3(1), dim3(64), 0, 0, d, o, sc);
            std::vector<unsigned> ho(128);
            CK(hipMemcpy(ho.data(), o, 128 * 4, hipMemcpyDeviceToHost));
            printf("A fp8  scale %.3f (dst_hi=false word, dst_hi=true word of lane 0: %08x %08x)\n   ", sc, ho[0], ho[1]);
            for (int i = 0; i < n; ++i) {
                const unsigned w = ho[2 * (i / 2)];
                printf("%g->%g  ", vals[i], dec_e4m3((w >> (8 * (i & 1))) & 255u));
            }
            printf("\n");
        }
        for (int bf = 0; bf < 2; ++bf)
            for (float sc : {1.0f, 4.0f}) {
                hipLaunchKernelGGL(k_cvt6, dim3(1), dim3(64), 0, 0, d, o, sc, bf);
                std::vector<unsigned> ho(8, 0u);
                CK(hipMemcpy(ho.data(), o, 6 * 4, hipMemcpyDeviceToHost));
                printf("A %s scale %.3f words %08x %08x %08x\n   ", bf ? "bf6" : "fp6", sc, ho[0], ho[1], ho[2]);
                for (int i = 0; i < n; ++i) printf("%g->%g  ", vals[i], bf ? dec_e3m2(get6(ho.data(), i)) : dec_e2m3(get6(ho.data(), i)));
                printf("\n");
            }
        CK(hipFree(d)); CK(hipFree(o));
    }
    // ---- B ----
    for (int fmt : {0, 2, 3}) {
        std::vector<unsigned> a(64 * 8), b(64 * 8);
        std::vector<int> sa(64), sb(64);
        unsigned x = 777u + fmt;
        auto rnd = [&]() { x = x * 1664525u + 1013904223u; return x >> 8; };
        for (auto& w : a) { w = 0; for (int q = 0; q < 4; ++q) { unsigned by = rnd() & 255u; if ((by & 0x7f) == 0x7f) by ^= 1; w |= by << (8 * q); } }
        for (auto& w : b) { w = 0; for (int q = 0; q < 4; ++q) { unsigned by = rnd() & 255u; if ((by & 0x7f) == 0x7f) by ^= 1; w |= by << (8 * q); } }
        for (int trial = 0; trial < 2; ++trial) {
            for (int l = 0; l < 64; ++l) { sa[l] = trial ? 120 + (int)(rnd() % 12) : 127; sb[l] = trial ? 124 + (int)(rnd() % 6) : 127; }
            v8i *da, *db; int *dsa, *dsb; f32x16* dd;
            CK(hipMalloc(&da, 64 * 32)); CK(hipMalloc(&db, 64 * 32)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dd, 64 * 64));
            CK(hipMemcpy(da, a.data(), 64 * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 64 * 32, hipMemcpyHostToDevice));
            CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
            if (fmt == 0) hipLaunchKernelGGL(k_mx<0>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
            if (fmt == 2) hipLaunchKernelGGL(k_mx<2>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
            if (fmt == 3) hipLaunchKernelGGL(k_mx<3>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
            std::vector<float> hd(64 * 16);
            CK(hipMemcpy(hd.data(), dd, 64 * 64, hipMemcpyDeviceToHost));
            auto elem = [&](const std::vector<unsigned>& v, int lane, int m) -> double {
                if (fmt == 0) return dec_e4m3((v[lane * 8 + (m >> 2)] >> (8 * (m & 3))) & 255u);
                const unsigned f = get6(&v[lane * 8], m);
                return fmt == 2 ? dec_e2m3(f) : dec_e3m2(f);
            };
            // H1: lane (i, h) pairs with lane (j, h) element by element, each lane's own scale on its 32 elements
            // H2: as H1 but the scales of lanes 0..31 serve both halves
            double e1 = 0, e2 = 0, mag = 0;
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 16; ++r) {
                    const int j = l & 31, i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                    double s1 = 0, s2 = 0;
                    for (int h = 0; h < 2; ++h) {
                        double p = 0;
                        for (int m = 0; m < 32; ++m) p += elem(a, i + 32 * h, m) * elem(b, j + 32 * h, m);
                        s1 += p * ldexp(1.0, sa[i + 32 * h] - 127) * ldexp(1.0, sb[j + 32 * h] - 127);
                        s2 += p * ldexp(1.0, sa[i] - 127) * ldexp(1.0, sb[j] - 127);
                    }
                    const double got = hd[l * 16 + r];
                    e1 = fmax(e1, fabs(got - s1)); e2 = fmax(e2, fabs(got - s2)); mag = fmax(mag, fabs(s1));
                }
            printf("B fmt %d %s scales: max|D| %.4g   H1 (own-lane scale) err %.3g   H2 (lanes 0..31 scale) err %.3g\n", fmt, trial ? "random" : "unit", mag, e1, e2);
            CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dsa)); CK(hipFree(dsb)); CK(hipFree(dd));
        }
    }
    // ---- B2: mixed formats (which of cbsz / blgp is the first operand's?) ----
    for (int order = 0; order < 2; ++order) {
        std::vector<unsigned> a(64 * 8), b(64 * 8);
        std::vector<int> sa(64), sb(64);
        unsigned x = 4242u;
        auto rnd = [&]() { x = x * 1664525u + 1013904223u; return x >> 8; };
        for (auto& w : a) w = rnd() | (rnd() << 24);
        for (auto& w : b) w = rnd() | (rnd() << 24);
        for (int l = 0; l < 64; ++l) { sa[l] = 120 + (int)(rnd() % 12); sb[l] = 124 + (int)(rnd() % 6); }
        v8i *da, *db; int *dsa, *dsb; f32x16* dd;
        CK(hipMalloc(&da, 64 * 32)); CK(hipMalloc(&db, 64 * 32)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dd, 64 * 64));
        CK(hipMemcpy(da, a.data(), 64 * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 64 * 32, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
        if (order == 0) hipLaunchKernelGGL((k_mx2<2, 3>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
        else hipLaunchKernelGGL((k_mx2<3, 2>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
        std::vector<float> hd(64 * 16);
        CK(hipMemcpy(hd.data(), dd, 64 * 64, hipMemcpyDeviceToHost));
        for (int hyp = 0; hyp < 2; ++hyp) {                 // hyp 0: cbsz is the FIRST operand's format; 1: the second's
            const int fa = (order == 0) == (hyp == 0) ? 2 : 3, fb = 5 - fa;
            double err = 0, mag = 0;
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 16; ++r) {
                    const int j = l & 31, i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                    double s1 = 0;
                    for (int h = 0; h < 2; ++h) {
                        double p = 0;
                        for (int m = 0; m < 32; ++m) {
                            const unsigned ca = get6(&a[(i + 32 * h) * 8], m), cb = get6(&b[(j + 32 * h) * 8], m);
                            p += (double)(fa == 2 ? dec_e2m3(ca) : dec_e3m2(ca)) * (double)(fb == 2 ? dec_e2m3(cb) : dec_e3m2(cb));
                        }
                        s1 += p * ldexp(1.0, sa[i + 32 * h] - 127) * ldexp(1.0, sb[j + 32 * h] - 127);
                    }
                    err = fmax(err, fabs(hd[l * 16 + r] - s1)); mag = fmax(mag, fabs(s1));
                }
            printf("B2 builtin(cbsz=%d, blgp=%d), hypothesis '%s': max|D| %.4g err %.3g\n", order == 0 ? 2 : 3, order == 0 ? 3 : 2,
                   hyp == 0 ? "cbsz = format of operand 1" : "cbsz = format of operand 2", mag, err);
        }
        CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dsa)); CK(hipFree(dsb)); CK(hipFree(dd));
    }
    // ---- D: fp16 fragments -> pk32 bf6 -> product with fp6 codes ----
    {
        std::vector<unsigned> a(64 * 6);
        std::vector<_Float16> hb(4 * 64 * 8);
        unsigned x = 99u;
        auto rnd = [&]() { x = x * 1664525u + 1013904223u; return x >> 8; };
        for (auto& w : a) w = rnd() | (rnd() << 24);
        for (auto& v : hb) v = (_Float16)(((int)(rnd() % 4001) - 2000) / 250.0f);       // [-8, 8]
        const float xscale = 4.0f; const int sa = 127 - 20, sb = 129;
        int* da; half8* dh; f32x16* dd;
        CK(hipMalloc(&da, 64 * 24)); CK(hipMalloc(&dh, hb.size() * 2)); CK(hipMalloc(&dd, 64 * 64));
        CK(hipMemcpy(da, a.data(), 64 * 24, hipMemcpyHostToDevice)); CK(hipMemcpy(dh, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_inreg, dim3(1), dim3(64), 0, 0, da, dh, dd, xscale, sa, sb);
        std::vector<float> hd(64 * 16);
        CK(hipMemcpy(hd.data(), dd, 64 * 64, hipMemcpyDeviceToHost));
        // host: bf6 nearest (ties to even) of x / xscale, saturating at 28
        auto q_bf6 = [&](float v) {
            float best = 0; double bd = 1e30; int bc = 0;
            for (int c = 0; c < 32; ++c) { const double g = dec_e3m2(c), dd_ = fabs(fabs((double)v) - g); if (dd_ < bd || (dd_ == bd && !(c & 1) && (bc & 1))) { bd = dd_; best = (float)g; bc = c; } }
            return v < 0 ? -best : best;
        };
        double err = 0, mag = 0, exact_err = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                const int j = l & 31, i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                double s = 0, se = 0;
                for (int h = 0; h < 2; ++h)
                    for (int kk = 0; kk < 4; ++kk)
                        for (int e = 0; e < 8; ++e) {
                            const double wv = dec_e2m3(get6(&a[(i + 32 * h) * 6], 8 * kk + e)) * ldexp(1.0, sa - 127);
                            const float xv = (float)hb[(kk * 64 + j + 32 * h) * 8 + e];
                            s += wv * (double)q_bf6(xv / xscale) * ldexp(1.0, sb - 127);
                            se += wv * (double)xv;
                        }
                err = fmax(err, fabs(hd[l * 16 + r] - s)); exact_err = fmax(exact_err, fabs(hd[l * 16 + r] - se)); mag = fmax(mag, fabs(se));
            }
        printf("D in-register fp16 -> bf6 -> fp6 x bf6 product: max|D| %.4g   err vs emulation %.3g   err vs unquantised x %.3g\n", mag, err, exact_err);
        CK(hipFree(da)); CK(hipFree(dh)); CK(hipFree(dd));
    }
    // ---- C ----
    {
        const size_t n = (size_t)256 * 64 * 32 * 8;
        std::vector<_Float16> h(n);
        unsigned x = 12345u;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (_Float16)(((int)(x >> 9) % 2001 - 1000) / 1000.0f); }
        _Float16* d; float* sink;
        CK(hipMalloc(&d, n * 2)); CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
        CK(hipMalloc(&sink, (size_t)256 * 512 * 4));
        rate<7>("16 f16", d, sink);
        rate<0>("32 f16 (f16_w2 group)", d, sink);
        rate<5>("4 fp8 K=64", d, sink);
        rate<6>("4 bf6 K=64", d, sink);
        rate<1>("16 f16 + 4 fp8", d, sink);
        rate<2>("16 f16 + 4 fp8 + 64 pk cvt", d, sink);
        rate<3>("16 f16 + 4 bf6", d, sink);
        rate<4>("16 f16 + 4 bf6 + 4 pk32 cvt", d, sink);
        rate<0>("32 f16 (again)", d, sink);
    }
    return 0;
}
