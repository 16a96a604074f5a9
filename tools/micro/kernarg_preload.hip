// What the kernel-argument fetch costs at the front of a latency-bound launch, and what preloading the leading arguments into SGPRs
// (-mllvm -amdgpu-kernarg-preload-count=N: only scalar / pointer parameters in front of the first by-value struct qualify) buys.
// The shape of the single clip's layer kernels: 224 workgroups x 576 threads, 110 KB of dynamic LDS, waves_per_eu(3,3); every thread's
// first action is a global load whose address comes from the arguments (the tile DMA / weight ring), the epilogue's pointers are needed last.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=14 -o tools/micro/kernarg_preload tools/micro/kernarg_preload.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Head { const float* x; int cin, swz, taps, dil; const float* w; int m_tiles, w_planes; long long variant; int n_variants; unsigned lo; };   // 56 B
struct Tail { const int* step; int off, rows; const unsigned* w6; int sc6; float xs; int kp; unsigned hi; };                                       // 40 B
struct Args { Head h; Tail t; };                                                                                                                      // 96 B
struct Args9 { Head h; Tail t; const int* ninth; };                                                                                                   // 104 B
struct Epi { const float* cproj; float* out; int ld, C; const int* rowclip; float s; int pad; };                                                      // 40 B

template <class A>
__device__ __forceinline__ void body(const Head& h, const A& t, const Epi& e) {
    extern __shared__ float sm[];
    const int tid = threadIdx.x, row = blockIdx.x * 32 + (tid & 31);
    float v = h.x[(size_t)row * h.cin + (tid >> 5) * h.dil] + h.w[(size_t)(blockIdx.y * h.m_tiles + (tid & 63)) * h.taps + h.variant];
    sm[tid ^ h.swz] = v;
    __syncthreads();
    v = sm[tid] * t.xs + (float)t.sc6;
    if (t.step) v += (float)t.step[0];
    if (e.rowclip[row] >= 0) e.out[(size_t)row * e.ld + (tid % e.C)] = v + e.cproj[tid & 63] * e.s;
}
__global__ void __launch_bounds__(576, 3) __attribute__((amdgpu_waves_per_eu(3, 3))) k_struct(const Args a, const Epi e) { body(a.h, a.t, e); }
__global__ void __launch_bounds__(576, 3) __attribute__((amdgpu_waves_per_eu(3, 3))) k_struct9(const Args9 a, const Epi e) { body(a.h, a.t, e); }
__global__ void __launch_bounds__(576, 3) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_flat(const float* x, int cin, int swz, int taps, int dil, const float* w, int m_tiles, int w_planes, long long variant, int n_variants, unsigned lo,
       const Tail t, const Epi e) {
    Head h{x, cin, swz, taps, dil, w, m_tiles, w_planes, variant, n_variants, lo};
    body(h, t, e);
}

template <class F>
float graph_time(const char* name, hipStream_t st, F&& launch) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 40; ++i) launch();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(a, st));
        for (int i = 0; i < 100; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(b, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const float us = ms * 1e3f / 4000;
        if (us < best) best = us;
        printf("%-44s %6.3f us per node\n", name, us);
    }
    return best;
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    float *x, *w, *out, *cproj; int* rc;
    const int rows = 28 * 32;
    CK(hipMalloc(&x, (size_t)rows * 768 * 4 + 4096)); CK(hipMemset(x, 0, (size_t)rows * 768 * 4 + 4096));
    CK(hipMalloc(&w, 1 << 22)); CK(hipMemset(w, 0, 1 << 22));
    CK(hipMalloc(&out, (size_t)rows * 768 * 4)); CK(hipMalloc(&cproj, 4096)); CK(hipMemset(cproj, 0, 4096));
    CK(hipMalloc(&rc, rows * 4)); CK(hipMemset(rc, 0, rows * 4));
    Args a{}; a.h = Head{x, 768, 7, 3, 2, w, 8, 2, 0, 1, 0}; a.t = Tail{nullptr, 0, 896, nullptr, 3, 0.5f, 0, 0};
    Args9 a9{}; a9.h = a.h; a9.t = a.t;
    Epi e{cproj, out, 768, 384, rc, 1.0f, 0};
    const size_t smem = 110 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_struct), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_struct9), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_flat), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const dim3 grid(28, 8), block(576);
    for (int round = 0; round < 2; ++round) {
        graph_time("by-value structs (96 + 40 B)", st, [&] { hipLaunchKernelGGL(k_struct, grid, block, smem, st, a, e); });
        graph_time("by-value structs (104 + 40 B)", st, [&] { hipLaunchKernelGGL(k_struct9, grid, block, smem, st, a9, e); });
        graph_time("14 leading dwords preloaded + structs", st, [&] {
            hipLaunchKernelGGL(k_flat, grid, block, smem, st, a.h.x, a.h.cin, a.h.swz, a.h.taps, a.h.dil, a.h.w, a.h.m_tiles, a.h.w_planes, a.h.variant,
                               a.h.n_variants, a.h.lo, a.t, e); });
    }
    return 0;
}
