// Which clock does the chip hold under a dense fp16 MFMA load, and what does one v_mfma_f32_32x32x16_f16 cost per SIMD then?
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_clock tools/micro/mfma_clock.hip && tools/micro/mfma_clock
// Register-resident loop (no memory in the loop): 4 independent accumulators per wave, random fp16 operands (or zeros: argv[1] = 0).
// s_memtime counts shader cycles, s_memrealtime a fixed 100 MHz clock: their ratio over the loop is the clock the CU actually ran at.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_mfma(const _Float16* __restrict__ src, float* __restrict__ sink, unsigned long long* __restrict__ stamps, int iters) {
    const int lane = threadIdx.x & 63;
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const half8*>(src + ((size_t)(blockIdx.x * 8 + i) * 64 + lane) * 8);
        b[i] = *reinterpret_cast<const half8*>(src + ((size_t)(blockIdx.x * 8 + 4 + i) * 64 + lane) * 8);
    }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it += 4) {               // 16 MFMAs per trip, every register index static (a dynamic one would go through scratch)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[(i + j) & 3], acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = c1 - c0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
    const int random_data = argc > 1 ? atoi(argv[1]) : 1;
    const int blocks = 256;
    std::vector<_Float16> h((size_t)blocks * 8 * 64 * 8);
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = random_data ? (_Float16)(((int)(x >> 9) % 2001 - 1000) / 1000.0f) : (_Float16)0.f; }
    _Float16* d; float* sink; unsigned long long* st;
    hipMalloc(&d, h.size() * 2); hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&sink, (size_t)blocks * 1024 * 4); hipMalloc(&st, blocks * 16);
    for (int threads : {256, 512}) {
        for (int iters : {2000, 20000, 200000}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(threads), 0, 0, d, sink, st, iters);      // warm-up
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(threads), 0, 0, d, sink, st, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> hs(blocks * 2);
            hipMemcpy(hs.data(), st, blocks * 16, hipMemcpyDeviceToHost);
            double cyc = 0, rt = 0;
            for (int b = 0; b < blocks; ++b) { cyc += hs[b * 2]; rt += hs[b * 2 + 1]; }
            cyc /= blocks; rt /= blocks;
            const double waves_per_simd = threads / 256.0;
            const double mfma_per_simd = 4.0 * iters * waves_per_simd;
            const double flop = 2.0 * 32 * 32 * 16 * 4.0 * iters * (threads / 64) * blocks;
            printf("data=%s waves/SIMD=%.0f iters=%6d: kernel %.3f ms  %.0f TFLOP/s | in-loop: %.1f shader cycles per MFMA per SIMD, clock %.2f GHz (s_memtime/s_memrealtime)\n",
                   random_data ? "random" : "zeros", waves_per_simd, iters, ms, flop / (ms * 1e-3) / 1e12, cyc / mfma_per_simd, cyc / (rt * 10.0) );
        }
    }
    return 0;
}
