// Round 6 (second session) probe: does the ORDER in which the fused layer kernel issues its fp16 MFMAs and its block-scaled 6-bit MFMAs cost matrix-pipe time?
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/mx_order tools/micro/mx_order.hip && tools/micro/mx_order
// mx_probe.hip (round 4) measured 16 fp16 + 4 bf6 (K = 64) MFMAs per group at 920 ns per wave pair against 681 + 161 = 842 for the two kinds alone -- with all 16
// fp16 MFMAs first (N-tile inner: dependency distance 4) and the four 6-bit ones behind them.  The kernel (tlayer.h: tl_compute_group_w6) issues them N-tile by
// N-tile: four fp16 MFMAs into ONE accumulator, the conversion, the 6-bit MFMA into the same accumulator (G6: a second one) -- eight changes of MFMA kind per
// group instead of two, each 6-bit MFMA directly dependent on the fp16 MFMA in front of it.  This probe times the orders on all 256 CUs, 2 waves per SIMD, random
// operands, register-resident (no LDS, no global loads): what is left is issue order, dependencies and the change of kind.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half32 __attribute__((ext_vector_type(32)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v6i __attribute__((ext_vector_type(6)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ v8i cvt6(const half8 (&b)[4]) {
    half32 v;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[8 * kk + e] = b[kk][e];
    const v6i o = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(v, 1.0f);
    return __builtin_shufflevector(o, o, 0, 1, 2, 3, 4, 5, -1, -1);
}
#define HI(nt, kk) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kk], b[nt][kk], acc[nt], 0, 0, 0)
#define LO(nt, q) acc[nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, q, acc[nt], 2, 3, 0, 127, 0, 127)
#define LG(nt) acc[nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(g8, c8[nt], acc[nt], 2, 3, 0, 127, 0, 127)

// MODE 0: 16 hi, kk outer / N-tile inner, then 4 conversions, then 4 lo        (mx_probe's "16 f16 + 4 bf6 + 4 pk32 cvt")
//      1: per N-tile: 4 hi into one accumulator, conversion, lo                  (the kernel's W6 order)
//      2: per N-tile: 4 hi, conversion, lo, g6                                   (the kernel's G6 order, output phase)
//      3: pairs of N-tiles: 8 hi, 2 conversions, 2 lo
//      4: N-tile major: 16 hi (4 per accumulator in a row), 4 conversions, 4 lo
//      5: per N-tile, lo one N-tile late: hi(nt) x 4, lo(nt - 1), conversion(nt)
//      6: 16 hi only, N-tile major (4 dependent in a row)
//      7: 16 hi only, kk outer / N-tile inner
//      8: N-tile major 16 hi, 4 conversions, 4 lo, 4 g6                          (G6 with both 6-bit kinds batched at the end)
//      9: pairs of N-tiles: 8 hi, 2 conversions, 2 lo, 2 g6
template <int MODE>
__global__ void __launch_bounds__(512, 2) k_rate(const _Float16* __restrict__ src, float* __restrict__ sink, int iters) {
    const int lane = threadIdx.x & 63;
    half8 a[4], b[4][4];
    v8i a8, g8, c8[4];
    const half8* s8 = reinterpret_cast<const half8*>(src) + (size_t)blockIdx.x * 64 * 32 + lane;
    for (int i = 0; i < 4; ++i) a[i] = s8[64 * i];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) b[i][j] = s8[64 * (4 + 4 * i + j)];
    {
        const v8i* si = reinterpret_cast<const v8i*>(src) + (size_t)blockIdx.x * 64 * 16 + lane;
        a8 = si[64 * 10]; g8 = si[64 * 9];
        for (int j = 0; j < 4; ++j) c8[j] = si[64 * (11 + j)];
        for (int q = 0; q < 8; ++q) { a8[q] &= 0x37373737; g8[q] &= 0x37373737; for (int j = 0; j < 4; ++j) c8[j][q] &= 0x37373737; }
    }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(b[nt][kk]));      // "fresh" B fragments: the conversion cannot be hoisted
        SB();
        if constexpr (MODE == 0 || MODE == 7) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) HI(nt, kk);
            SB();
            if constexpr (MODE == 0) {
                v8i q[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) q[nt] = cvt6(b[nt]);
                SB();
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) LO(nt, q[nt]);
            }
        } else if constexpr (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) HI(nt, kk);
                const v8i q = cvt6(b[nt]);
                LO(nt, q);
                if constexpr (MODE == 2) LG(nt);
                SB();
            }
        } else if constexpr (MODE == 3 || MODE == 9) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
#pragma unroll
                for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) HI(2 * p + n2, kk);
                SB();
                const v8i q0 = cvt6(b[2 * p]), q1 = cvt6(b[2 * p + 1]);
                SB();
                LO(2 * p, q0); LO(2 * p + 1, q1);
                if constexpr (MODE == 9) { LG(2 * p); LG(2 * p + 1); }
                SB();
            }
        } else if constexpr (MODE == 4 || MODE == 6 || MODE == 8) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) HI(nt, kk);
            SB();
            if constexpr (MODE != 6) {
                v8i q[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) q[nt] = cvt6(b[nt]);
                SB();
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) LO(nt, q[nt]);
                if constexpr (MODE == 8) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) LG(nt);
                }
            }
        } else if constexpr (MODE == 5) {
            v8i qp = cvt6(b[3]);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) HI(nt, kk);
                if (nt > 0) LO(nt - 1, qp);
                qp = cvt6(b[nt]);
                SB();
            }
            LO(3, qp);
        }
        SB();
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void rate(const char* what, const _Float16* d, float* sink) {
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
    printf("order %-78s %8.3f ms   %7.1f ns per group per wave pair\n", what, best, best * 1e6 / iters);
}

int main() {
    const size_t n = (size_t)256 * 64 * 32 * 8;
    std::vector<_Float16> h(n);
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (_Float16)(((int)(x >> 9) % 2001 - 1000) / 1000.0f); }
    _Float16* d; float* sink;
    CK(hipMalloc(&d, n * 2)); CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&sink, (size_t)256 * 512 * 4));
    for (int round = 0; round < 2; ++round) {
        rate<7>("16 hi only, kk outer / N-tile inner", d, sink);
        rate<6>("16 hi only, N-tile major (4 dependent in a row)", d, sink);
        rate<0>("16 hi (N-tile inner), 4 cvt, 4 lo              [mx_probe's order]", d, sink);
        rate<4>("16 hi N-tile major, 4 cvt, 4 lo", d, sink);
        rate<1>("per N-tile: 4 hi, cvt, lo                      [the kernel's W6 order]", d, sink);
        rate<5>("per N-tile: 4 hi, lo of the N-tile before, cvt", d, sink);
        rate<3>("pairs of N-tiles: 8 hi, 2 cvt, 2 lo", d, sink);
        rate<2>("per N-tile: 4 hi, cvt, lo, g6                  [the kernel's G6 order]", d, sink);
        rate<9>("pairs of N-tiles: 8 hi, 2 cvt, 2 lo, 2 g6", d, sink);
        rate<8>("16 hi N-tile major, 4 cvt, 4 lo, 4 g6", d, sink);
    }
    return 0;
}
