// Can this chip run dense fp16 MFMA and an HBM stream AT THE SAME TIME, and at what joint rates?  
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/overlap tools/micro/overlap.hip && tools/micro/overlap
//
// One workgroup per CU.  Waves [0, NM) sit in the register-resident v_mfma_f32_32x32x16_f16 loop of csrc/probe.hip (random fp16
// operands, 4 independent accumulators, no memory traffic); waves [NM, NM + NS) stream HBM with the residual layer's access pattern
// (tlayer.h): fully coalesced 16-byte-per-lane non-temporal loads and stores, 1 KiB per wave-instruction, two bytes read per byte
// written, every wave walking its own 4 MB region of a 16 + 8 GB footprint (far beyond the 256 MB Infinity Cache).  Both roles run
// against the SAME wall-clock window (a deadline on s_memrealtime, the fixed 100 MHz counter), so "joint" rates are rates over the same
// interval; each wave reports its trips, its shader cycles (s_memtime) and its realtime ticks, which gives the clock the CU held.
// Waves are dealt to the 4 SIMDs round-robin, so with NM = NS = 4 every SIMD hosts one MFMA wave and one streaming wave -- the shape a
// software-pipelined layer kernel would have (dedicated memory waves under the gate MFMAs).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Rec { unsigned long long trips, cycles, ticks, role; };

constexpr size_t REGION = 4u << 20;          // bytes of the read stream one wave walks (its write region is half of that)
constexpr int MAX_NS = 16;                   // streaming waves per CU that have their own region

__global__ void __launch_bounds__(1024) k_overlap(const _Float16* __restrict__ ops, const float* __restrict__ rd, float* __restrict__ wr,
                                                  float* __restrict__ sink, Rec* __restrict__ rec, int nm, int ns, unsigned long long window_ticks, int mode) {
    // mode bit 0: streaming waves raise their priority (s_setprio 3); bit 1: the MFMA waves are the YOUNGER waves of the workgroup
    // (arbitration between co-resident waves is by priority, then age -- MI355X_MICROARCH.md "Two waves per SIMD");
    // bit 2: every streaming wave walks an 8 KB region only (L2-resident: is it HBM or the memory pipeline that MFMAs exclude?);
    // bit 3: the MFMA waves run at ~50 % duty (a burst of 512 MFMAs, then s_sleep for about as long): is the trade linear in time?
    const int lane = threadIdx.x & 63;
    const int wave_hw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = (mode & 2) ? (wave_hw + nm) % (nm + ns) : wave_hw;      // role index: < nm = MFMA
    const int gw = blockIdx.x * (nm + ns) + wave;
    if ((mode & 1) && wave >= nm) __builtin_amdgcn_s_setprio(3);
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long deadline = r0 + window_ticks;
    unsigned long long trips = 0;
    if (wave < nm) {
        half8 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = *reinterpret_cast<const half8*>(ops + ((size_t)((gw & 63) * 8 + i) * 64 + lane) * 8);
            b[i] = *reinterpret_cast<const half8*>(ops + ((size_t)((gw & 63) * 8 + 4 + i) * 64 + lane) * 8);
        }
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        do {
            for (int it = 0; it < 32; ++it) {                       // 32 x 16 MFMAs between two looks at the clock
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[(i + j) & 3], acc[i], 0, 0, 0);
            }
            trips += 32;
            if (mode & 8) {                                          // ~512 MFMAs x 32 cycles x (2 waves per SIMD when nm == 8) of idling
                const int naps = nm > 4 ? 4 : 2;
                for (int z = 0; z < naps; ++z) __builtin_amdgcn_s_sleep(127);           // 127 x 64 cycles each
            }
        } while (__builtin_amdgcn_s_memrealtime() < deadline);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][r];
        sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        const size_t sw = (size_t)blockIdx.x * ns + (wave - nm);     // streaming-wave index
        const float* rp = rd + sw * (REGION / 4);
        float* wp = wr + sw * (REGION / 8);
        size_t off = 0;                                              // in floats, within the region
        f32x4 keep = {0.f, 0.f, 0.f, 0.f};
        do {                                                         // one trip = 8 KiB read + 4 KiB written per wave, 8 loads in flight
            const float* p = rp + off + (size_t)lane * 4;
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + u * 256));
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                f32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = v[2 * u][i] + v[2 * u + 1][i];
                __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(wp + (off >> 1) + u * 256 + (size_t)lane * 4));
                keep[0] += o[0];
            }
            off += 2048;
            if (off >= ((mode & 4) ? (size_t)2048 : REGION / 4)) off = 0;
            trips += 1;
        } while (__builtin_amdgcn_s_memrealtime() < deadline);
        if (keep[0] == 123.456f) sink[0] = keep[0];
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) rec[gw] = Rec{trips, c1 - c0, r1 - r0, (unsigned long long)(wave < nm ? 0 : 1)};
}

int main(int argc, char** argv) {
    const double window_ms = argc > 1 ? atof(argv[1]) : 4.0;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const int max_stream_waves = cus * MAX_NS;
    std::vector<_Float16> h((size_t)64 * 8 * 64 * 8);
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (_Float16)(((int)(x >> 9) % 2001 - 1000) / 1000.0f); }
    _Float16* ops; float *rd, *wr, *sink; Rec* rec;
    hipMalloc(&ops, h.size() * 2); hipMemcpy(ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    if (hipMalloc(&rd, (size_t)max_stream_waves * REGION) != hipSuccess || hipMalloc(&wr, (size_t)max_stream_waves * REGION / 2) != hipSuccess) {
        printf("allocation failed\n"); return 1;
    }
    hipMemset(rd, 0x3c, (size_t)max_stream_waves * REGION);         // finite floats
    hipMalloc(&sink, (size_t)cus * 1024 * 4); hipMalloc(&rec, (size_t)cus * 32 * sizeof(Rec));
    printf("%d CUs, window %.1f ms per configuration; MFMA = v_mfma_f32_32x32x16_f16 on random operands, stream = nt dwordx4, 2 B read per B written\n", cus, window_ms);
    printf("%-36s %10s %10s %10s | %9s %9s\n", "waves per CU (MFMA + stream)", "TFLOP/s", "read+write", "GB/s", "MFMA GHz", "strm GHz");
    struct Cfg { int nm, ns; const char* what; int mode; };
    const Cfg cfgs[] = {{8, 0, "MFMA alone, 2 waves/SIMD", 0}, {4, 0, "MFMA alone, 1 wave/SIMD", 0}, {0, 8, "stream alone, 2 waves/SIMD", 0},
                        {0, 4, "stream alone, 1 wave/SIMD", 0}, {0, 16, "stream alone, 4 waves/SIMD", 0}, {4, 4, "joint 4 + 4 (1 + 1 per SIMD)", 0},
                        {8, 8, "joint 8 + 8 (2 + 2 per SIMD)", 0}, {4, 8, "joint 4 + 8", 0}, {8, 4, "joint 8 + 4", 0}, {4, 12, "joint 4 + 12", 0},
                        {4, 4, "joint 4 + 4, stream waves prio 3", 1}, {8, 8, "joint 8 + 8, stream waves prio 3", 1}, {4, 8, "joint 4 + 8, stream prio 3", 1},
                        {8, 4, "joint 8 + 4, stream prio 3", 1}, {4, 12, "joint 4 + 12, stream prio 3", 1},
                        {4, 4, "joint 4 + 4, MFMA waves younger", 2}, {8, 8, "joint 8 + 8, MFMA waves younger", 2}, {4, 8, "joint 4 + 8, MFMA younger", 2},
                        {4, 4, "joint 4 + 4, younger + prio", 3}, {8, 8, "joint 8 + 8, younger + prio", 3},
                        {0, 8, "stream alone, L2-resident (8 KB/wave)", 4}, {4, 4, "joint 4 + 4, L2-resident stream", 4},
                        {8, 8, "joint 8 + 8, L2-resident stream", 4}, {4, 8, "joint 4 + 8, L2-resident stream", 4},
                        {8, 0, "MFMA alone at ~50 % duty", 8}, {8, 8, "joint 8 + 8, MFMA ~50 % duty", 8}, {4, 4, "joint 4 + 4, MFMA ~50 % duty", 8},
                        {4, 8, "joint 4 + 8, MFMA ~50 % duty", 8}, {8, 8, "joint 8 + 8, 50 % duty, stream prio 3", 9},
                        {8, 0, "MFMA alone again (chip now warm)", 0}};
    for (const Cfg& c : cfgs) {
        const int waves = c.nm + c.ns;
        if (waves > 16 || c.ns > MAX_NS) continue;
        for (int rep = 0; rep < 2; ++rep) {                           // first pass warms up / settles the clock
            hipMemset(rec, 0, (size_t)cus * 32 * sizeof(Rec));
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_overlap, dim3(cus), dim3(64 * waves), 0, 0, ops, rd, wr, sink, rec, c.nm, c.ns,
                               (unsigned long long)(window_ms * 1e5), c.mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            if (hipGetLastError() != hipSuccess) { printf("%-36s launch failed\n", c.what); break; }
            if (rep == 0) continue;
            std::vector<Rec> r((size_t)cus * 32);
            hipMemcpy(r.data(), rec, r.size() * sizeof(Rec), hipMemcpyDeviceToHost);
            const int per_cu = c.nm + c.ns;
            double mf = 0, by = 0, cm = 0, tm = 0, cs = 0, ts = 0, span = 0;
            for (int i = 0; i < cus * per_cu; ++i) {
                const double sec = r[i].ticks * 1e-8;
                if (sec > span) span = sec;
                if (r[i].role == 0) { mf += (double)r[i].trips * 16 * 2.0 * 32 * 32 * 16; cm += r[i].cycles; tm += r[i].ticks; }
                else { by += (double)r[i].trips * 12288.0; cs += r[i].cycles; ts += r[i].ticks; }
            }
            printf("%-36s %10.0f %10s %10.0f | %9.2f %9.2f   (kernel %.2f ms)\n", c.what, mf / span / 1e12, "", by / span / 1e9,
                   tm > 0 ? cm / (tm * 10.0) : 0.0, ts > 0 ? cs / (ts * 10.0) : 0.0, ms);
        }
    }
    return 0;
}
