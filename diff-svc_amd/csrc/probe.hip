// dsvc_probe_mfma: what does this chip SUSTAIN on dense fp16 MFMA right now?  (include/dsvc.h)
// A register-resident v_mfma_f32_32x32x16_f16 loop (no memory traffic inside it) on every CU, two waves per SIMD, random or zero
// operands.  The matrix pipe is paced at 32 shader cycles per instruction either way; what differs is the clock the chip holds:
// measured on MI355X 2.39 GHz / 2.49 PF/s with zero operands, 1.60-1.62 GHz / 1.65 PF/s with random ones (profiles/r2h_mfma_clock.txt)
// -- the datasheet's 2.5 PF/s is not reachable on real data, and a roofline fraction is worth reading against both numbers.
#include <vector>

#include "../../include/dsvc.h"
#include "common.h"

using namespace dsvc;

namespace {
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(512) k_probe_mfma(const _Float16* __restrict__ src, float* __restrict__ sink, unsigned long long* __restrict__ stamps, int iters) {
    const int lane = threadIdx.x & 63;
    half8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const half8*>(src + ((size_t)((blockIdx.x & 31) * 8 + i) * 64 + lane) * 8);
        b[i] = *reinterpret_cast<const half8*>(src + ((size_t)((blockIdx.x & 31) * 8 + 4 + i) * 64 + lane) * 8);
    }
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it += 4) {               // every register index static: a dynamic one would go through scratch
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[(i + j) & 3], acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = c1 - c0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

__global__ void k_probe_fill(_Float16* p, int n, int random_data) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)i * 2654435761u + 12345u;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15;
    p[i] = random_data ? (_Float16)(((int)(x >> 9) % 2001 - 1000) / 1000.0f) : (_Float16)0.f;
}
}  // namespace

static int probe_run(int32_t random_data, float* out4, void* stream);

extern "C" int dsvc_probe_mfma(int32_t random_data, float* tflops, float* clock_ghz, void* stream) {
    if (!tflops || !clock_ghz) return fail(DSVC_EINVAL, "null argument");
    float o[4];
    DSVC_TRY(probe_run(random_data, o, stream));
    *tflops = o[0]; *clock_ghz = o[2];
    return DSVC_OK;
}

// out4: [0] TFLOP/s over the kernel's wall time (HIP events), [1] TFLOP/s inside the timed loops (FLOPs / mean in-loop realtime: what the
// matrix pipes sustain once every wave is in its loop), [2] mean clock over the CUs in GHz, [3] lowest clock any CU held
extern "C" int dsvc_probe_mfma_detail(int32_t random_data, float* out4, void* stream) {
    if (!out4) return fail(DSVC_EINVAL, "null argument");
    return probe_run(random_data, out4, stream);
}

static int probe_run(int32_t random_data, float* out4, void* stream) {
    float tf_ev = 0.f, ghz = 0.f;
    float* tflops = &tf_ev; float* clock_ghz = &ghz;
    hipStream_t st = (hipStream_t)stream;
    hipDeviceProp_t prop;
    DSVC_HIP(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256, iters = 40000;
    const int n = 32 * 8 * 64 * 8;
    _Float16* src = nullptr; float* sink = nullptr; unsigned long long* stamps = nullptr;
    DSVC_HIP(hipMalloc(&src, (size_t)n * 2)); DSVC_HIP(hipMalloc(&sink, (size_t)blocks * 512 * 4)); DSVC_HIP(hipMalloc(&stamps, (size_t)blocks * 16));
    hipLaunchKernelGGL(k_probe_fill, dim3(ceil_div(n, 256)), dim3(256), 0, st, src, n, random_data);
    hipEvent_t e0, e1;
    DSVC_HIP(hipEventCreate(&e0)); DSVC_HIP(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_probe_mfma, dim3(blocks), dim3(512), 0, st, src, sink, stamps, iters);          // warm-up: lets the clock settle
    DSVC_HIP(hipEventRecord(e0, st));
    hipLaunchKernelGGL(k_probe_mfma, dim3(blocks), dim3(512), 0, st, src, sink, stamps, iters);
    DSVC_HIP(hipEventRecord(e1, st));
    DSVC_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    DSVC_HIP(hipEventElapsedTime(&ms, e0, e1));
    // every CU reports (shader cycles, 100 MHz ticks) over its loop: the clock is the mean over the CUs (the XCDs do not all hold the
    // same clock under this load; one workgroup's ratio is not the chip's)
    std::vector<unsigned long long> all((size_t)blocks * 2);
    DSVC_HIP(hipMemcpy(all.data(), stamps, (size_t)blocks * 16, hipMemcpyDeviceToHost));
    unsigned long long hs[2] = {0, 0};
    for (int b = 0; b < blocks; ++b) { hs[0] += all[2 * b]; hs[1] += all[2 * b + 1]; }
    // the loop runs iters / 4 trips of 16 MFMAs: 4 * iters MFMAs per wave, 8 waves per workgroup
    *tflops = (float)(2.0 * 32 * 32 * 16 * 4.0 * (double)iters * 8 * blocks / (ms * 1e-3) / 1e12);
    *clock_ghz = hs[1] ? (float)((double)hs[0] / ((double)hs[1] * 10.0)) : 0.f;            // shader cycles / (100 MHz ticks * 10 ns)
    double min_clk = 1e9;
    for (int b = 0; b < blocks; ++b)
        if (all[2 * b + 1]) { const double c = (double)all[2 * b] / ((double)all[2 * b + 1] * 10.0); if (c < min_clk) min_clk = c; }
    const double mean_ticks = (double)hs[1] / blocks;
    out4[0] = tf_ev;
    out4[1] = mean_ticks > 0 ? (float)(2.0 * 32 * 32 * 16 * 4.0 * (double)iters * 8 * blocks / (mean_ticks * 1e-8) / 1e12) : 0.f;
    out4[2] = ghz;
    out4[3] = min_clk < 1e9 ? (float)min_clk : 0.f;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(src); (void)hipFree(sink); (void)hipFree(stamps);
    return DSVC_OK;
}
