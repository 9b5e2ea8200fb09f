// Output layout of v_mfma_f64_16x16x4_f64 on gfx950: D[i][j] = 16 i + j from A[i][0] = i, A[i][1] = 1, B[0][j] = 16, B[1][j] = j.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(double* out) {
    const int l = threadIdx.x, i = l % 16, kk = l / 16;
    const double a = kk == 0 ? (double)i : (kk == 1 ? 1.0 : 0.0);      // A[i][kk], lane = i + 16 kk
    const double b = kk == 0 ? 16.0 : (kk == 1 ? (double)i : 0.0);      // B[kk][j = l % 16]
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
    double* d; hipMalloc(&d, 256 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l : {0, 1, 15, 16, 17, 32, 48, 63}) printf("lane %2d: (%d,%d) (%d,%d) (%d,%d) (%d,%d)\n", l, (int)h[l*4]/16, (int)h[l*4]%16, (int)h[l*4+1]/16, (int)h[l*4+1]%16, (int)h[l*4+2]/16, (int)h[l*4+2]%16, (int)h[l*4+3]/16, (int)h[l*4+3]%16);
    return 0;
}
