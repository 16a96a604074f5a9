// dsvc_probe_mfma: what does this chip SUSTAIN on dense fp16 MFMA right now?  (include/dsvc.h)
// A register-resident v_mfma_f32_32x32x16_f16 loop (no memory traffic inside it) on every CU, two waves per SIMD, random or zero
// operands, time-boxed.  The matrix pipe is paced at 32 shader cycles per instruction either way; what differs is the clock the chip holds:
// measured on MI355X 2.39 GHz / 2.49 PF/s with zero operands, 1.50-1.63 GHz / 1.56-1.71 PF/s with random ones (profiles/r2h_mfma_clock.txt,
// r3c_overlap.txt)
// -- the datasheet's 2.5 PF/s is not reachable on real data, and a roofline fraction is worth reading against both numbers.
#include <vector>

#include "../../include/dsvc.h"
#include "../../include/dsvc_debug.h"
#include "common.h"

using namespace dsvc;

namespace {
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Time-boxed: every wave issues MFMAs until a deadline on the fixed 100 MHz counter and reports how many it got through.  (A fixed amount
// of work per wave is the wrong probe: arbitration between the two waves of a SIMD is by age, the older wave of a pair runs at the full
// pipe rate and finishes in half the kernel's time -- its own loop time says 3.1 PF/s while the kernel as a whole delivers 1.2.)
__global__ void __launch_bounds__(512) k_probe_mfma(const _Float16* __restrict__ src, float* __restrict__ sink, unsigned long long* __restrict__ stamps,
                                                    unsigned long long window_ticks) {
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
    unsigned long long trips = 0;
    do {
        for (int it = 0; it < 32; ++it) {                  // 32 x 16 MFMAs between two looks at the clock; every register index static
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[(i + j) & 3], acc[i], 0, 0, 0);
        }
        trips += 32;
    } while (__builtin_amdgcn_s_memrealtime() - r0 < window_ticks);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) {                                       // per wave: trips, shader cycles, realtime ticks
        unsigned long long* o = stamps + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 3;
        o[0] = trips; o[1] = c1 - c0; o[2] = r1 - r0;
    }
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
    *tflops = o[1]; *clock_ghz = o[2];
    return DSVC_OK;
}

// out4: [0] TFLOP/s over the kernel's wall time (HIP events), [1] TFLOP/s inside the timed loops (FLOPs / mean in-loop realtime: what the
// matrix pipes sustain once every wave is in its loop), [2] mean clock over the CUs in GHz, [3] lowest clock any CU held
extern "C" int dsvc_probe_mfma_detail(int32_t random_data, float* out4, void* stream) {
    if (!out4) return fail(DSVC_EINVAL, "null argument");
    return probe_run(random_data, out4, stream);
}

static int probe_run(int32_t random_data, float* out4, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    hipDeviceProp_t prop;
    DSVC_HIP(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const unsigned long long window = 400000;             // 4 ms of the 100 MHz counter
    const int n = 32 * 8 * 64 * 8;
    _Float16* src = nullptr; float* sink = nullptr; unsigned long long* stamps = nullptr;
    DSVC_HIP(hipMalloc(&src, (size_t)n * 2)); DSVC_HIP(hipMalloc(&sink, (size_t)blocks * 512 * 4)); DSVC_HIP(hipMalloc(&stamps, (size_t)blocks * 8 * 24));
    hipLaunchKernelGGL(k_probe_fill, dim3(ceil_div(n, 256)), dim3(256), 0, st, src, n, random_data);
    hipEvent_t e0, e1;
    DSVC_HIP(hipEventCreate(&e0)); DSVC_HIP(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_probe_mfma, dim3(blocks), dim3(512), 0, st, src, sink, stamps, window);          // warm-up: lets the clock settle
    DSVC_HIP(hipEventRecord(e0, st));
    hipLaunchKernelGGL(k_probe_mfma, dim3(blocks), dim3(512), 0, st, src, sink, stamps, window);
    DSVC_HIP(hipEventRecord(e1, st));
    DSVC_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    DSVC_HIP(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> all((size_t)blocks * 8 * 3);
    DSVC_HIP(hipMemcpy(all.data(), stamps, all.size() * 8, hipMemcpyDeviceToHost));
    double mfmas = 0, cyc = 0, ticks = 0, span = 0, min_clk = 1e9;
    for (int w = 0; w < blocks * 8; ++w) {
        mfmas += (double)all[3 * w] * 16.0; cyc += (double)all[3 * w + 1]; ticks += (double)all[3 * w + 2];
        if ((double)all[3 * w + 2] > span) span = (double)all[3 * w + 2];
        if (all[3 * w + 2]) { const double c = (double)all[3 * w + 1] / ((double)all[3 * w + 2] * 10.0); if (c < min_clk) min_clk = c; }
    }
    const double flops = mfmas * 2.0 * 32 * 32 * 16;
    out4[0] = ms > 0 ? (float)(flops / (ms * 1e-3) / 1e12) : 0.f;                 // over the kernel's wall time (HIP events)
    out4[1] = span > 0 ? (float)(flops / (span * 1e-8) / 1e12) : 0.f;             // over the longest in-loop window of any wave
    out4[2] = ticks > 0 ? (float)(cyc / (ticks * 10.0)) : 0.f;                    // mean clock: shader cycles / (100 MHz ticks * 10 ns)
    out4[3] = min_clk < 1e9 ? (float)min_clk : 0.f;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(src); (void)hipFree(sink); (void)hipFree(stamps);
    return DSVC_OK;
}
