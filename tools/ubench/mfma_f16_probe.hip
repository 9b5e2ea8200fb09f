// Probe (GPU box): issue rate of f16 MFMA shapes, fp16-denormal handling of MFMA inputs, and accuracy of the
// 3 x fp16 split (hi*hi + hi*lo + lo*hi, fp32 accumulate) against fp64 and against the exact fp32 MFMA.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int WHICH>
__global__ void rate_kernel(float* out, unsigned long long* cyc, int iters) {
    f4 acc[6];
    for (int i = 0; i < 6; ++i) acc[i] = f4{0, 0, 0, 0};
    h4 a = {(_Float16)1.0f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)2.0f}, b = a;
    h8 a8 = {1, 2, 3, 4, 5, 6, 7, 8}, b8 = a8;
    float af = threadIdx.x * 0.001f, bf = 1.0f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (WHICH == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, acc[i], 0, 0, 0);
            if (WHICH == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i], 0, 0, 0);
            if (WHICH == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[i], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// D[16x16] = A[16xK] * B[Kx16], K = 16 per MFMA step; mode 0: exact fp32 MFMA, 1: fp16 inputs, 2: 3xfp16 split
__global__ void gemm_kernel(const float* A, const float* B, float* D, int K, int mode, float scale_a) {
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    f4 acc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 16) {
        if (mode == 0) {
            for (int kk = 0; kk < 16; kk += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * K + k0 + kk + g], B[(k0 + kk + g) * 16 + i], acc, 0, 0, 0);
        } else {
            h4 ah, al, bh, bl;
            for (int j = 0; j < 4; ++j) {
                float a = A[i * K + k0 + 4 * g + j] * scale_a, b = B[(k0 + 4 * g + j) * 16 + i];
                ah[j] = (_Float16)a; al[j] = (_Float16)(a - (float)ah[j]);
                bh[j] = (_Float16)b; bl[j] = (_Float16)(b - (float)bh[j]);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, acc, 0, 0, 0);
            if (mode == 2) {
                acc = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bh, acc, 0, 0, 0);
            }
        }
    }
    for (int r = 0; r < 4; ++r) D[(g * 4 + r) * 16 + i] = acc[r] / (mode == 0 ? 1.0f : scale_a);
}

__global__ void denorm_kernel(float* out) {
    h4 a = {(_Float16)0.0f, 0, 0, 0}, b = {(_Float16)1.0f, 0, 0, 0};
    a[0] = (_Float16)3.0e-6f;                     // fp16 subnormal (min normal 6.1e-5)
    f4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)a[0]; }
}

int main() {
    float* d; unsigned long long* c;
    hipMalloc(&d, 4096); hipMalloc(&c, 64);
    const char* names[3] = {"v_mfma_f32_16x16x16_f16", "v_mfma_f32_16x16x32_f16", "v_mfma_f32_16x16x4_f32"};
    for (int w = 0; w < 3; ++w) {
        unsigned long long h = 0; const int iters = 20000;
        if (w == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(1), dim3(64), 0, 0, d, c, iters);
        if (w == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(1), dim3(64), 0, 0, d, c, iters);
        if (w == 2) hipLaunchKernelGGL(rate_kernel<2>, dim3(1), dim3(64), 0, 0, d, c, iters);
        hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf("%-28s %.2f cycles per MFMA (one wave, 6 independent accumulators)\n", names[w], (double)h / (iters * 6.0));
    }
    float hd[2];
    hipLaunchKernelGGL(denorm_kernel, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(hd, d, 8, hipMemcpyDeviceToHost);
    printf("fp16 subnormal input 3e-6 through MFMA: acc = %.4e (operand value %.4e) -> %s\n", hd[0], hd[1], hd[0] != 0.0f ? "NOT flushed" : "FLUSHED");
    // accuracy
    const int K = 864;
    std::mt19937 rng(1); std::uniform_real_distribution<float> ua(-1.f, 1.f); std::normal_distribution<float> nw(0.f, 0.03f);
    std::vector<float> A(16 * K), B(K * 16), D(256);
    for (auto& v : A) v = nw(rng);                 // weights
    for (auto& v : B) v = std::max(0.f, ua(rng)) * ((rng() & 7) == 0 ? 0.01f : 1.f);    // ReLU-like activations, some tiny
    float *dA, *dB, *dD; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    std::vector<double> ref(256, 0.0); double mag = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * B[k * 16 + j]; ref[i * 16 + j] = s; mag += fabs(s); }
    mag /= 256;
    const char* mn[4] = {"exact fp32 MFMA", "fp16 inputs (1 MFMA)", "3 x fp16 split", "3 x fp16 split, weights x1024"};
    for (int m = 0; m < 4; ++m) {
        hipLaunchKernelGGL(gemm_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD, K, m == 3 ? 2 : m, m == 3 ? 1024.f : 1.f);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
        double e = 0, emax = 0; for (int q = 0; q < 256; ++q) { double x = fabs(D[q] - ref[q]); e += x; emax = std::max(emax, x); }
        printf("%-32s mean |err| %.3e  max %.3e   (mean |ref| %.3e)\n", mn[m], e / 256, emax, mag);
    }
    return 0;
}
