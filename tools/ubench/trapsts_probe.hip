// Does TRAPSTS.EXCP accumulate the overflow of a float -> half conversion with exception traps disabled (the default)?
// If so, an unguarded hi/lo split can be CHECKED per wave after the fact at zero cost per value.
//   hipcc --offload-arch=gfx950 -O2 -o trapsts_probe tools/ubench/trapsts_probe.hip && ./trapsts_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void probe(const float* x, unsigned* out, _Float16* sink) {
    const int lane = threadIdx.x;
    __builtin_amdgcn_s_setreg(3 | (8 << 11), 0);                              // TRAPSTS[8:0] = 0
    const float v = x[lane];
    const float t0 = __uint_as_float(__float_as_uint(v) & 0xFFFFE000u);
    const _Float16 hi = (_Float16)t0, lo = (_Float16)(v - t0);
    sink[2 * (blockIdx.x * 64 + lane)] = hi; sink[2 * (blockIdx.x * 64 + lane) + 1] = lo;
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    const unsigned e = __builtin_amdgcn_s_getreg(3 | (8 << 11));
    if (lane == 0) out[blockIdx.x] = e;
}
int main() {
    float h[4][64];
    for (int c = 0; c < 4; ++c) for (int i = 0; i < 64; ++i) h[c][i] = 1.0f + i;
    h[1][37] = 7.0e4f;                 // overflows fp16
    h[2][5] = NAN;                     // quiet NaN
    h[3][9] = INFINITY;                // inf - inf in the lo half
    float* dx; unsigned* dout; _Float16* dsink;
    hipMalloc(&dx, sizeof(h)); hipMalloc(&dout, 16); hipMalloc(&dsink, 4 * 64 * 2 * 2);
    hipMemcpy(dx, h, sizeof(h), hipMemcpyHostToDevice);
    unsigned r[4];
    const char* what[4] = {"in range", "one lane 7e4 (overflow)", "one lane NaN", "one lane inf"};
    for (int c = 0; c < 4; ++c) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dx + c * 64, dout + c, dsink + c * 128);
    }
    hipDeviceSynchronize();
    hipMemcpy(r, dout, 16, hipMemcpyDeviceToHost);
    for (int c = 0; c < 4; ++c) printf("%-26s TRAPSTS.EXCP = 0x%03x (invalid %u, overflow %u, underflow %u, inexact %u)\n", what[c], r[c], r[c] & 1, (r[c] >> 3) & 1, (r[c] >> 4) & 1, (r[c] >> 5) & 1);
    return 0;
}
