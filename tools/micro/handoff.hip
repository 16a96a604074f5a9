// What would a flag-synchronised single-clip layer cost?  The B=1 residual layer is two kernels today: 224 workgroups compute the gate
// output g (each a 32-frame x 48-channel slice, 3 KB), the kernel boundary hands all of g to 224 workgroups of the output projection
// (each reads the 8 slices of its frame tile, 24 KB).  This micro-benchmark times that exchange both ways on the real geometry:
//   (a) two launches per layer (boundary = the hand-off), and
//   (b) ONE launch per layer: produce -> publish -> wait for the 8 producers of the tile -> consume, in three publishing flavours
//       (plain stores + __threadfence, write-through sc1 stores + drained flag, and both with all 8 workgroups of a tile on one XCD).
// Every wait is bounded (a broken assumption ends in a count of timeouts, not a hang) and every consumer checks every word it reads.
// hipcc --offload-arch=gfx950 -O3 -o tools/micro/handoff tools/micro/handoff.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int TILES = 28, SLICES = 8, WG = TILES * SLICES;      // 224 workgroups
constexpr int THREADS = 576;                                    // 9 waves, as the real kernels
constexpr int SLICE_WORDS = 768;                                // 3 KB per (tile, slice): 32 frames x 48 channels fp16
constexpr int WORK_WORDS = 8192;                                // 32 KB streamed per workgroup per phase (weights + tile stand-in)

struct Args {
    const unsigned* stream;      // [>= WG * WORK_WORDS] read-only traffic
    unsigned* g;                 // [2][TILES][SLICES][SLICE_WORDS] (double-buffered by layer parity)
    unsigned* out;               // [WG] checksums
    unsigned* flags;             // [2][TILES]
    unsigned* stats;             // [0] timeouts, [1] bad words, [2..] max wait (memrealtime ticks)
    unsigned layer;              // value tag: every word of slice s of tile t in layer l is (l << 16) | (t << 8) | s
    int mode;                    // 0 plain + threadfence, 1 sc1 write-through + drained flag
    int same_xcd;                // 1: the 8 workgroups of a tile sit on one XCD (block b runs on XCD b % 8)
};

__device__ __forceinline__ void map_block(int b, int same_xcd, int& tile, int& slice) {
    if (same_xcd) { const int xcd = b & 7, slot = b >> 3; tile = xcd + 8 * (slot >> 3); slice = slot & 7; }
    else { tile = b / SLICES; slice = b % SLICES; }
}

__device__ __forceinline__ unsigned stream_work(const Args& a, int b, unsigned salt) {
    unsigned s = salt;
    const unsigned* p = a.stream + (size_t)b * WORK_WORDS;
    for (int i = threadIdx.x; i < WORK_WORDS; i += THREADS) s += p[i];
    return s;
}

__device__ __forceinline__ void produce(const Args& a, int tile, int slice, unsigned keep) {
    unsigned* dst = a.g + (((size_t)(a.layer & 1) * TILES + tile) * SLICES + slice) * SLICE_WORDS;
    const unsigned v = (a.layer << 16) | ((unsigned)tile << 8) | (unsigned)slice;
    for (int i = threadIdx.x; i < SLICE_WORDS; i += THREADS) dst[i] = v + (keep & 0u);
}

__device__ __forceinline__ void produce_sc1(const Args& a, int tile, int slice) {
    unsigned* dst = a.g + (((size_t)(a.layer & 1) * TILES + tile) * SLICES + slice) * SLICE_WORDS;
    const unsigned v = (a.layer << 16) | ((unsigned)tile << 8) | (unsigned)slice;
    for (int i = threadIdx.x; i < SLICE_WORDS; i += THREADS) {
        unsigned* p = dst + i;
        asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__device__ __forceinline__ unsigned consume(const Args& a, int tile, unsigned* bad) {
    const unsigned* src = a.g + ((size_t)(a.layer & 1) * TILES + tile) * SLICES * SLICE_WORDS;
    unsigned s = 0, nb = 0;
    for (int i = threadIdx.x; i < SLICES * SLICE_WORDS; i += THREADS) {
        const unsigned w = src[i];
        const unsigned want = (a.layer << 16) | ((unsigned)tile << 8) | (unsigned)(i / SLICE_WORDS);
        nb += (w != want);
        s += w;
    }
    *bad = nb;
    return s;
}

// (a) two launches
__global__ void __launch_bounds__(THREADS) k_produce(Args a) {
    int tile, slice; map_block(blockIdx.x, a.same_xcd, tile, slice);
    if (tile >= TILES) return;
    const unsigned keep = stream_work(a, blockIdx.x, 1u);
    produce(a, tile, slice, keep);
}
__global__ void __launch_bounds__(THREADS) k_consume(Args a) {
    int tile, slice; map_block(blockIdx.x, a.same_xcd, tile, slice);
    if (tile >= TILES) return;
    unsigned keep = stream_work(a, blockIdx.x, 2u);
    unsigned bad;
    keep += consume(a, tile, &bad);
    if (bad) atomicAdd(a.stats + 1, bad);
    if (threadIdx.x == 0) a.out[tile * SLICES + slice] = keep;
}

// (b) one launch: produce -> publish -> wait -> consume
__global__ void __launch_bounds__(THREADS) k_fused(Args a) {
    int tile, slice; map_block(blockIdx.x, a.same_xcd, tile, slice);
    if (tile >= TILES) return;
    unsigned* flag = a.flags + (a.layer & 1) * TILES + tile;
    if (slice == 0 && threadIdx.x == 0) a.flags[((a.layer + 1) & 1) * TILES + tile] = 0;      // the NEXT layer's counter (nobody touches it in this launch)
    unsigned keep = stream_work(a, blockIdx.x, 1u);
    if (a.mode == 1) produce_sc1(a, tile, slice); else produce(a, tile, slice, keep);
    __syncthreads();
    unsigned long long t0 = 0;
    if (threadIdx.x == 0) {
        if (a.mode == 0) __threadfence();                                                     // release: plain stores -> visible device-wide
        t0 = __builtin_amdgcn_s_memrealtime();
        __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the consumer's own weight stream would be prefetched here in the real kernel
        int spins = 0;
        const bool broken = __hip_atomic_load(a.stats, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;    // one timeout anywhere: stop waiting everywhere
        while (!broken && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)SLICES) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > 50000) { atomicAdd(a.stats, 1u); break; }
        }
        const unsigned dt = (unsigned)(__builtin_amdgcn_s_memrealtime() - t0);
        atomicMax(a.stats + 2, dt);
        atomicAdd(a.stats + 3, dt);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                                        // every wave: drop stale L1 / L2 lines before reading g
    keep += stream_work(a, blockIdx.x, 2u);
    unsigned bad;
    keep += consume(a, tile, &bad);
    if (bad) atomicAdd(a.stats + 1, bad);
    if (threadIdx.x == 0) a.out[tile * SLICES + slice] = keep;
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    Args a{};
    unsigned* stream; CK(hipMalloc(&stream, (size_t)256 * WORK_WORDS * 4)); CK(hipMemset(stream, 1, (size_t)256 * WORK_WORDS * 4));
    a.stream = stream;
    CK(hipMalloc(&a.g, (size_t)2 * TILES * SLICES * SLICE_WORDS * 4)); CK(hipMemset(a.g, 0xff, (size_t)2 * TILES * SLICES * SLICE_WORDS * 4));
    CK(hipMalloc(&a.out, 256 * 4)); CK(hipMalloc(&a.flags, 2 * TILES * 4)); CK(hipMemset(a.flags, 0, 2 * TILES * 4));
    CK(hipMalloc(&a.stats, 64)); CK(hipMemset(a.stats, 0, 64));
    const int LAYERS = 2000;                                   // even: the flag double-buffer closes on itself
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto report = [&](const char* name, float us_per_layer) {
        unsigned h[4]; CK(hipMemcpy(h, a.stats, 16, hipMemcpyDeviceToHost));
        printf("%-78s %7.2f us/layer   timeouts %u  bad words %u", name, us_per_layer, h[0], h[1]);
        if (h[3]) printf("  wait: mean %.2f us, max %.2f us", h[3] / (double)LAYERS / WG * 0.01, h[2] * 0.01);
        printf("\n");
        CK(hipMemset(a.stats, 0, 64));
    };
    auto run_graph = [&](auto&& enqueue) -> float {          // all launches of a measurement as ONE graph (no host pacing), replayed once timed
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        enqueue();
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        CK(hipMemset(a.stats, 0, 64));
        CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        return ms;
    };
    for (int same = 0; same <= 1; ++same) {
        a.same_xcd = same;
        const int grid = same ? 256 : WG;
        {   // (a) two launches per layer
            a.mode = 0;
            const float ms = run_graph([&] {
                for (int l = 0; l < LAYERS; ++l) { a.layer = l; hipLaunchKernelGGL(k_produce, dim3(grid), dim3(THREADS), 0, st, a); hipLaunchKernelGGL(k_consume, dim3(grid), dim3(THREADS), 0, st, a); }
            });
            report(same ? "two launches per layer, tile's workgroups on one XCD" : "two launches per layer, workgroups in block order", ms * 1e3f / LAYERS);
        }
        for (int mode = 0; mode <= 1; ++mode) {   // (b) one launch per layer
            a.mode = mode;
            CK(hipMemset(a.flags, 0, 2 * TILES * 4));
            const float ms = run_graph([&] {
                for (int l = 0; l < LAYERS; ++l) { a.layer = l; hipLaunchKernelGGL(k_fused, dim3(grid), dim3(THREADS), 0, st, a); }
            });
            char name[160];
            snprintf(name, sizeof name, "ONE launch per layer, %s, %s", mode ? "sc1 write-through stores + drained flag" : "plain stores + __threadfence",
                     same ? "tile's workgroups on one XCD" : "block order");
            report(name, ms * 1e3f / LAYERS);
        }
    }
    return 0;
}
