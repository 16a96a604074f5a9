// Shared device/host helpers for the diff-svc gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/dsvc.h"

namespace dsvc {

// ---- error plumbing (C functions return int status; message kept per thread) ----
void set_error(const std::string& msg);
int fail(int code, const char* fmt, ...);

#define DSVC_HIP(call)                                                                      \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return ::dsvc::fail(DSVC_EHIP, "%s failed: %s (%s:%d)", #call,          \
                                hipGetErrorString(e_), __FILE__, __LINE__);                 \
    } while (0)

#define DSVC_TRY(expr)                      \
    do {                                    \
        int rc_ = (expr);                   \
        if (rc_ != DSVC_OK) return rc_; \
    } while (0)

// ---- Philox4x32-10 counter-based generator (identical to oracle/dsvc_oracle.py) ----
enum : uint32_t { PURPOSE_DDPM_NOISE = 1, PURPOSE_X_INIT = 2, PURPOSE_SINE_NOISE = 3, PURPOSE_SINE_PHASE = 4 };

struct u32x4 { uint32_t x, y, z, w; };

__host__ __device__ inline u32x4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint64_t seed) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

// (0,1] / [0,1) from the top 24 bits: exact in fp32, same as the oracle's float64 expressions.
__host__ __device__ inline float u01_open_low(uint32_t r) { return (float)((r >> 8) + 1u) * (1.0f / 16777216.0f); }
__host__ __device__ inline float u01_open_high(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

// lane-th of the four Box-Muller normals of one Philox call
__device__ inline float philox_normal_lane(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint64_t seed, int lane) {
    const u32x4 r = philox4x32(c0, c1, c2, c3, seed);
    const uint32_t ra = (lane & 2) ? r.z : r.x;
    const uint32_t rb = (lane & 2) ? r.w : r.y;
    const float rad = sqrtf(-2.0f * logf(u01_open_low(ra)));
    float s, c;
    sincospif(2.0f * u01_open_high(rb), &s, &c);
    return rad * ((lane & 1) ? s : c);
}

__device__ inline void philox_normal4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint64_t seed, float out[4]) {
    const u32x4 r = philox4x32(c0, c1, c2, c3, seed);
    const float ra = sqrtf(-2.0f * logf(u01_open_low(r.x)));
    const float rb = sqrtf(-2.0f * logf(u01_open_low(r.z)));
    float s, c;
    sincospif(2.0f * u01_open_high(r.y), &s, &c);
    out[0] = ra * c; out[1] = ra * s;
    sincospif(2.0f * u01_open_high(r.w), &s, &c);
    out[2] = rb * c; out[3] = rb * s;
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

}  // namespace dsvc
