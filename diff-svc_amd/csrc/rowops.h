// rowops.h -- row kernels shared by the conv_gemm-based translation units (hubert.hip, pe.hip).
#pragma once
#include "common.h"

namespace dsvc {
namespace {

// LayerNorm over the channel axis, one wave per row; out may alias in
__global__ void k_layernorm(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ gamma, const float* __restrict__ beta,
                            int rows, int C, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* p = in + (size_t)row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += p[c];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = p[c] - mean; q += d * d; }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
    float* o_ = out + (size_t)row * C;
    for (int c = lane; c < C; c += 64) o_[c] = (p[c] - mean) * rstd * gamma[c] + beta[c];
}

}  // namespace
}  // namespace dsvc
