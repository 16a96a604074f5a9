// Helpers shared by the epilogues of the tgemm engine -- the sampler's (diffnet_t.h) and the training forward's (train.hip): the
// accumulator-tiled fp32 layout, 16-byte loads / stores, the fp16 hi|lo split of a lane's 16 channels, the pre-scaled gate activation.
#pragma once
#include "tgemm.h"

namespace dsvc {

// "accumulator-tiled" fp32 layout of the buffers that only the tgemm epilogues touch (residual stream, skip sum,
// hoisted conditioner projection): [frame tile of 32][m_tile][q = reg/4][lane 64][4 floats], i.e. exactly the order
// in which a wave's accumulator registers hold a 32-channel x 32-frame tile.  Every accumulator-init load and every
// epilogue store is then ONE fully coalesced 1 KiB access per instruction (a frame-major layout would touch 32
// different 128-B lines with 32 useful bytes each).  Element (frame, m_tile, reg r, half h) lives at
//   ((frame/32 * n_mtiles + m_tile) * 4 + r/4) * 256 + ((frame%32) + 32*h) * 4 + r%4
__host__ __device__ __forceinline__ size_t tiled_off(int frame, int n_mtiles, int mt, int r, int h) {
    return (((size_t)(frame >> 5) * n_mtiles + mt) * 4 + (r >> 2)) * 256 + (size_t)((frame & 31) + 32 * h) * 4 + (r & 3);
}
// pointer to register quad 0 of (frame tile containing `frame`, m_tile) for this lane; quad q is at + q*256 floats
__device__ __forceinline__ size_t tiled_lane_base(int frame, int n_mtiles, int mt, int lane) {
    return ((size_t)(frame >> 5) * n_mtiles + mt) * 1024 + (size_t)lane * 4;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4_nt(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void st4_nt(float* p, f32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); }

// split 16 fp32 values into fp16 hi and lo (= fp16(v - hi)) and store both planes: dst[0..15] and dst[lo_off..lo_off+15]
__device__ __forceinline__ void store_hi_lo16(_Float16* dst, int lo_off, const float (&v)[16]) {
    half8 h0, h1, l0, l1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h0[i] = (_Float16)v[i]; h1[i] = (_Float16)v[8 + i];
        l0[i] = (_Float16)(v[i] - (float)h0[i]); l1[i] = (_Float16)(v[8 + i] - (float)h1[i]);
    }
    *reinterpret_cast<half8*>(dst) = h0;
    *reinterpret_cast<half8*>(dst + 8) = h1;
    *reinterpret_cast<half8*>(dst + lo_off) = l0;
    *reinterpret_cast<half8*>(dst + lo_off + 8) = l1;
}

// sigmoid(a) * tanh(b) = (1 - E2) / ((1 + E1) * (1 + E2)),  E1 = exp(-a), E2 = exp(-2b)   (net.py:73-77).
// The gate kernel's weights and conditioner projection are packed PRE-SCALED (gate rows by -log2(e), filter rows by
// -2 log2(e)), so its accumulators already hold ag = -a*log2(e) and bf = -2b*log2(e): 9 VALU per output, 3 of them
// transcendental (v_exp_f32 x2, v_rcp_f32), instead of the ~30 the libm forms expand to.
constexpr float GATE_SCALE = -1.4426950408889634f;        // -log2(e)
constexpr float FILT_SCALE = -2.8853900817779268f;        // -2 log2(e)
__device__ __forceinline__ float gate_act_scaled(float ag, float bf) {
    bf = __builtin_amdgcn_fmed3f(bf, -43.28f, 43.28f);     // |b| <= 15: tanh is +-1 to fp32 precision beyond |b| ~ 9; keeps E2 finite
    const float e1 = __builtin_amdgcn_exp2f(ag);
    const float e2 = __builtin_amdgcn_exp2f(bf);
    return (1.0f - e2) * __builtin_amdgcn_rcpf((1.0f + e1) * (1.0f + e2));
}

}  // namespace dsvc
