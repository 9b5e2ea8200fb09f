// Sustained matrix-pipe rate under full-chip load and the unit of s_memtime: every SIMD of every CU issues N back-to-back
// v_mfma_f32_16x16x32_f16 (4 independent accumulators, operands in registers).  Reports wall time per MFMA per SIMD (-> the
// clock if one MFMA is 16 cycles), s_memtime ticks per MFMA, and the same with two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* ticks, int n) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int i = 0; i < n; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    f4 s = c0 + c1 + c2 + c3;
    if (s[0] == 12345.0f) out[threadIdx.x] = s[1];
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
int main() {
    float* out; unsigned long long* ticks; hipMalloc(&out, 4096); hipMalloc(&ticks, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 50000;                                     // 200,000 MFMAs per wave
    for (int waves = 4; waves <= 8; waves += 4)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 0, 0, out, ticks, n);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long tk; hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost);
            const double mf = 4.0 * n * (waves / 4);         // MFMAs per SIMD
            printf("%d wave(s)/SIMD: %.3f ms, %.2f ns per MFMA per SIMD -> %.3f GHz at 16 cycles/MFMA; s_memtime: %.2f ticks per MFMA of the wave -> tick rate %.3f GHz; chip %.0f TFLOP/s\n",
                   waves / 4, ms, ms * 1e6 / mf, 16.0 * mf / (ms * 1e6), (double)tk / (4.0 * n), (double)tk / (ms * 1e6), mf * 1024 * 16384 / (ms * 1e-3) / 1e12);
        }
    return 0;
}
