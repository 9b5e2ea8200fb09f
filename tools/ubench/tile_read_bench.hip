// Cold-HBM read bandwidth of "every thread reads an 8x8-pixel block of a 2304x3840 fp32 plane" access patterns (the PCA
// projection's input side) against a linear stream of the same bytes.  6 planes per pass, 4 rotating buffers (850 MB) so
// that nothing is served from the 256-MiB Infinity Cache.  Why does the projection's min/max pass read at 1-2 TB/s?
#include <hip/hip_runtime.h>
#include <cstdio>
#define H 2304
#define W 3840
#define P 6
// A: lane = block (32-B lane stride), 8 rows x 2 dwordx4; 256 consecutive blocks per workgroup (the PCA kernels' pattern)
__global__ __launch_bounds__(256) void kA(const float* __restrict__ in, float* __restrict__ out) {
    const long b = (long)blockIdx.x * 256 + threadIdx.x; const int BW = W / 8, BH = H / 8;
    const int p = b / (BH * BW), r = b % (BH * BW), by = r / BW, bx = r % BW;
    const float* q = in + (long)p * H * W + (long)by * 8 * W + bx * 8;
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { float4 a = *(const float4*)(q + (long)i * W), c = *(const float4*)(q + (long)i * W + 4); s += a.x + a.y + a.z + a.w + c.x + c.y + c.z + c.w; }
    if (s == 12345.678f) out[b] = s;
}
// B: as A but each wave's loads are fully contiguous 1-KB segments (wrong pixels per lane, same bytes per workgroup)
__global__ __launch_bounds__(256) void kB(const float* __restrict__ in, float* __restrict__ out) {
    const long b = (long)blockIdx.x * 256 + threadIdx.x; const int BW = W / 8, BH = H / 8;
    const long b0 = b - (threadIdx.x & 63);
    const int p = b0 / (BH * BW), r = b0 % (BH * BW), by = r / BW, bx = r % BW;
    const float* q = in + (long)p * H * W + (long)by * 8 * W + bx * 8 + (threadIdx.x & 63) * 4;
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { float4 a = *(const float4*)(q + (long)i * W), c = *(const float4*)(q + (long)i * W + 256); s += a.x + a.y + a.z + a.w + c.x + c.y + c.z + c.w; }
    if (s == 12345.678f) out[b] = s;
}
// C: linear stream, 16 B per lane, 2 x dwordx4 x 8 per thread (same bytes per thread, consecutive 128-B... per thread 256 B contiguous chunks strided by the wave)
__global__ __launch_bounds__(256) void kC(const float* __restrict__ in, float* __restrict__ out) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const float* q = in + (long)blockIdx.x * 256 * 64 + threadIdx.x * 4;
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { float4 a = *(const float4*)(q + i * 1024); s += a.x + a.y + a.z + a.w; }
    if (s == 12345.678f) out[t] = s;
}
// D: as A with a grid-stride persistent loop (2048 workgroups), i.e. fewer resident waves start at the same time
__global__ __launch_bounds__(256) void kD(const float* __restrict__ in, float* __restrict__ out, long nb) {
    const int BW = W / 8, BH = H / 8; float s = 0;
    for (long b = (long)blockIdx.x * 256 + threadIdx.x; b < nb; b += (long)gridDim.x * 256) {
        const int p = b / (BH * BW), r = b % (BH * BW), by = r / BW, bx = r % BW;
        const float* q = in + (long)p * H * W + (long)by * 8 * W + bx * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) { float4 a = *(const float4*)(q + (long)i * W), c = *(const float4*)(q + (long)i * W + 4); s += a.x + a.y + a.z + a.w + c.x + c.y + c.z + c.w; }
    }
    if (s == 12345.678f) out[blockIdx.x] = s;
}
// E: as A but the workgroup is 64 blocks wide x 4 block-rows tall (32 image rows x 2 KB)
__global__ __launch_bounds__(256) void kE(const float* __restrict__ in, float* __restrict__ out) {
    const int BW = W / 8, BH = H / 8; const int gx = (BW + 63) / 64, gy = BH / 4;
    const int p = blockIdx.x / (gx * gy), r = blockIdx.x % (gx * gy), by = (r / gx) * 4 + (threadIdx.x >> 6), bx = (r % gx) * 64 + (threadIdx.x & 63);
    if (bx >= BW) return;
    const float* q = in + (long)p * H * W + (long)by * 8 * W + bx * 8;
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { float4 a = *(const float4*)(q + (long)i * W), c = *(const float4*)(q + (long)i * W + 4); s += a.x + a.y + a.z + a.w + c.x + c.y + c.z + c.w; }
    if (s == 12345.678f) out[blockIdx.x] = s;
}
int main() {
    const size_t bytes = (size_t)P * H * W * 4; const long nb = (long)P * (H / 8) * (W / 8);
    float* buf[4]; float* out;
    for (int i = 0; i < 4; ++i) { hipMalloc(&buf[i], bytes); hipMemset(buf[i], 0, bytes); }
    hipMalloc(&out, nb * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"A lane=block, 32-B lane stride (PCA pattern)", "B same rows, contiguous 1-KB loads", "C linear stream", "D pattern A, persistent 2048 workgroups", "E pattern A, workgroup 64 blocks x 4 block-rows"};
    for (int k = 0; k < 5; ++k) {
        float best = 1e9, sum = 0; const int reps = 8;
        for (int r = 0; r < reps + 2; ++r) {
            const float* in = buf[r % 4];
            hipEventRecord(e0);
            if (k == 0) hipLaunchKernelGGL(kA, dim3((nb + 255) / 256), dim3(256), 0, 0, in, out);
            if (k == 1) hipLaunchKernelGGL(kB, dim3((nb + 255) / 256), dim3(256), 0, 0, in, out);
            if (k == 2) hipLaunchKernelGGL(kC, dim3(bytes / (256 * 256)), dim3(256), 0, 0, in, out);
            if (k == 3) hipLaunchKernelGGL(kD, dim3(2048), dim3(256), 0, 0, in, out, nb);
            if (k == 4) hipLaunchKernelGGL(kE, dim3(P * ((W / 8 + 63) / 64) * (H / 8 / 4)), dim3(256), 0, 0, in, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (r >= 2) { sum += ms; if (ms < best) best = ms; }
        }
        printf("%-52s avg %7.1f us  best %7.1f us  %.2f TB/s (avg)\n", names[k], sum / reps * 1e3, best * 1e3, bytes / (sum / reps * 1e-3) / 1e12);
    }
    return 0;
}
