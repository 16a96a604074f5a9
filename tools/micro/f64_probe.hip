// Accumulator layout of v_mfma_f64_16x16x4_f64 (no ISA document in the image): D[i][j] with A[i][0] = i, B[0][j] = 1 gives the ROW each
// (lane, register) holds; with A[i][0] = 1, B[0][j] = j the COLUMN.   hipcc --offload-arch=gfx950 -O2 -o f64_probe f64_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(double* out) {
    const int l = threadIdx.x;
    const int i = l & 15, q = l >> 4;
    d4 r = {0, 0, 0, 0}, c = {0, 0, 0, 0}, kk = {0, 0, 0, 0};
    r = __builtin_amdgcn_mfma_f64_16x16x4f64(q == 0 ? (double)i : 0.0, q == 0 ? 1.0 : 0.0, r, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(q == 0 ? 1.0 : 0.0, q == 0 ? (double)i : 0.0, c, 0, 0, 0);
    // k pairing: A[i][k] = 10^k (lane group q <-> k), B[k][j] = (k == 2): D = A[i][k paired with B's group 2]
    kk = __builtin_amdgcn_mfma_f64_16x16x4f64(q == 0 ? 1.0 : q == 1 ? 10.0 : q == 2 ? 100.0 : 1000.0, q == 2 ? 1.0 : 0.0, kk, 0, 0, 0);
    for (int v = 0; v < 4; ++v) { out[(l * 4 + v) * 3] = r[v]; out[(l * 4 + v) * 3 + 1] = c[v]; out[(l * 4 + v) * 3 + 2] = kk[v]; }
}
int main() {
    double* d; hipMalloc(&d, 64 * 4 * 3 * 8);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    double h[64 * 4 * 3]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 5)
        printf("lane %2d: rows %g %g %g %g | cols %g %g %g %g | k-pair %g\n", l, h[(l*4+0)*3], h[(l*4+1)*3], h[(l*4+2)*3], h[(l*4+3)*3],
               h[(l*4+0)*3+1], h[(l*4+1)*3+1], h[(l*4+2)*3+1], h[(l*4+3)*3+1], h[(l*4)*3+2]);
    return 0;
}
