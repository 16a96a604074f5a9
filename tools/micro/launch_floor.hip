// Per-launch cost of back-to-back dependent kernels on one stream (HIP events around N launches), for the shapes the
// B=1 denoiser step uses.  hipcc --offload-arch=gfx950 -O3 -o tools/micro/launch_floor tools/micro/launch_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Big { const void* p[12]; int v[24]; };

__global__ void k_trivial(int* p) { if (p && threadIdx.x == 12345) *p = 1; }
__global__ void __launch_bounds__(256) k_lds(int* p) {
    extern __shared__ char sm[];
    if (p && threadIdx.x == 12345) { sm[threadIdx.x] = 1; *p = sm[3]; }
}
__global__ void __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) k_attr(Big b, Big c) {
    extern __shared__ char sm[];
    if (b.v[3] == 12345) { sm[threadIdx.x] = 1; *(int*)b.p[0] = sm[3] + c.v[1]; }
    __syncthreads();
}
// touches memory like a real kernel: every thread writes one dword (dirty lines at the boundary)
__global__ void __launch_bounds__(256) k_store(float* p, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = (float)i;
}

template <class F>
float timeit(const char* name, int n, hipStream_t st, F&& launch) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 50; ++i) launch();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < n; ++i) launch();
    CK(hipEventRecord(b, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-58s %7.2f us/launch\n", name, ms * 1e3f / n);
    return ms * 1e3f / n;
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    int* d; CK(hipMalloc(&d, 1 << 24));
    Big b{}; b.p[0] = d; Big c{};
    const int N = 4000;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_attr), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    timeit("trivial <<<1,64>>>", N, st, [&] { hipLaunchKernelGGL(k_trivial, dim3(1), dim3(64), 0, st, nullptr); });
    timeit("trivial <<<256,256>>>", N, st, [&] { hipLaunchKernelGGL(k_trivial, dim3(256), dim3(256), 0, st, nullptr); });
    timeit("trivial <<<336,256>>>", N, st, [&] { hipLaunchKernelGGL(k_trivial, dim3(336), dim3(256), 0, st, nullptr); });
    timeit("trivial <<<(28,12),256>>>", N, st, [&] { hipLaunchKernelGGL(k_trivial, dim3(28, 12), dim3(256), 0, st, nullptr); });
    timeit("dyn LDS 37 KB <<<336,256>>>", N, st, [&] { hipLaunchKernelGGL(k_lds, dim3(336), dim3(256), 37 * 1024, st, nullptr); });
    timeit("dyn LDS 110 KB <<<224,256>>>", N, st, [&] { hipLaunchKernelGGL(k_lds, dim3(224), dim3(256), 110 * 1024, st, nullptr); });
    timeit("waves_per_eu(2,2), 2x288 B kernarg, 37 KB, barrier", N, st, [&] { hipLaunchKernelGGL(k_attr, dim3(28, 12), dim3(256), 37 * 1024, st, b, c); });
    timeit("store 1 dword/thread <<<336,256>>>", N, st, [&] { hipLaunchKernelGGL(k_store, dim3(336), dim3(256), 0, st, (float*)d, 336 * 256); });
    timeit("store 1 dword/thread <<<4096,256>>> (4 MB dirty)", N, st, [&] { hipLaunchKernelGGL(k_store, dim3(4096), dim3(256), 0, st, (float*)d, 4096 * 256); });
    // the same through a captured graph of 40 launches
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(k_attr, dim3(28, 12), dim3(256), 37 * 1024, st, b, c);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    float us = timeit("graph of 40 x (waves_per_eu kernel)", 100, st, [&] { CK(hipGraphLaunch(ge, st)); });
    printf("  -> %.2f us per kernel node\n", us / 40);
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(k_trivial, dim3(336), dim3(256), 0, st, nullptr);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    us = timeit("graph of 40 x trivial <<<336,256>>>", 100, st, [&] { CK(hipGraphLaunch(ge, st)); });
    printf("  -> %.2f us per kernel node\n", us / 40);
    // null stream for comparison
    timeit("trivial <<<336,256>>> on the NULL stream", N, 0, [&] { hipLaunchKernelGGL(k_trivial, dim3(336), dim3(256), 0, 0, nullptr); });
    return 0;
}
