// Probe (GPU box): what does a K = 16 matrix instruction cost next to the K = 32 one, chip-wide and under the power cap?
// The 3x3 ring convolution spends 5 K = 32 steps on the 9 taps x 16 channels = 144 k-values of a chunk (the tenth half-step multiplies
// a zero-weight pad tap).  Alternative: 4 steps of v_mfma_f32_16x16x32_f16 + 1 step of v_mfma_f32_16x16x16_f16.  Every SIMD of every CU
// runs `waves` waves issuing chunks of (steps x 18 matrix instructions: 3 x 2 accumulators, 3 split terms), operands in registers.
//   mode 0: 5 x K32            (today)
//   mode 1: 4 x K32 + 1 x K16  (candidate)
//   mode 2: 4 x K32            (what the pad step costs at all)
//   mode 3: 9 x K16            (all-K16: is the legacy instruction half the cycles?)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* ticks, int n) {
    h8 a[3], b[2];
    for (int m = 0; m < 3; ++m) for (int i = 0; i < 8; ++i) a[m][i] = (_Float16)(0.001f * (threadIdx.x + i + m));
    for (int q = 0; q < 2; ++q) for (int i = 0; i < 8; ++i) b[q][i] = (_Float16)(0.002f * (threadIdx.x - i + q));
    f4 c[3][2];
    for (int m = 0; m < 3; ++m) for (int q = 0; q < 2; ++q) c[m][q] = f4{0, 0, 0, 0};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int i = 0; i < n; ++i) {
        constexpr int N32 = MODE == 0 ? 5 : (MODE == 3 ? 0 : 4), N16 = MODE == 1 ? 1 : (MODE == 3 ? 9 : 0);
#pragma unroll
        for (int s = 0; s < N32; ++s)
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int q = 0; q < 2; ++q) c[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], b[q], c[m][q], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < N16; ++s)
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const h4 a4 = {a[m][0], a[m][1], a[m][2], a[m][3]}, b4 = {b[q][4], b[q][5], b[q][6], b[q][7]};
                        c[m][q] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c[m][q], 0, 0, 0);
                    }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    f4 s = c[0][0] + c[0][1] + c[1][0] + c[1][1] + c[2][0] + c[2][1];
    if (s[0] == 12345.0f) out[threadIdx.x] = s[1];
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

int main() {
    float* out; unsigned long long* ticks; hipMalloc(&out, 4096); hipMalloc(&ticks, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 6000;
    const char* names[4] = {"5 x K32", "4 x K32 + 1 x K16", "4 x K32", "9 x K16"};
    for (int waves = 4; waves <= 8; waves += 4)
        for (int rep = 0; rep < 2; ++rep)
            for (int mode = 0; mode < 4; ++mode) {
                hipEventRecord(e0);
                for (int l = 0; l < 4; ++l) {
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * waves), 0, 0, out, ticks, n);
                    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * waves), 0, 0, out, ticks, n);
                    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(64 * waves), 0, 0, out, ticks, n);
                    if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(64 * waves), 0, 0, out, ticks, n);
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                unsigned long long tk; hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost);
                const double chunks = 4.0 * n * (waves / 4);                     // chunks per SIMD
                printf("%d wave(s)/SIMD  %-18s %8.3f ms  %7.1f ns per chunk per SIMD  (s_memtime ticks per chunk of a wave: %.1f)\n",
                       waves / 4, names[mode], ms, ms * 1e6 / chunks, (double)tk / n);
            }
    return 0;
}
