// Effective HBM bandwidth of "read P planes, write Q planes" per-pixel kernels at 2304x3840 as a function of P, Q, bytes per lane and
// workgroup shape.  (Why do the multi-plane kernels of this repo sit at ~2.5 TB/s when a stream reaches 6?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int VEC, int ROWS>
__global__ __launch_bounds__(256) void k(const float* __restrict__ in, float* __restrict__ out, int P, int Q, int H, int W) {
    // workgroup = (256 / ROWS) * VEC pixels wide x ROWS rows
    constexpr int TX = 256 / ROWS;
    const int x = (blockIdx.x * TX + threadIdx.x % TX) * VEC, y = blockIdx.y * ROWS + threadIdx.x / TX;
    if (x >= W || y >= H) return;
    const size_t HW = (size_t)H * W, pix = (size_t)y * W + x;
    float acc[VEC];
    for (int v = 0; v < VEC; ++v) acc[v] = 0.0f;
    for (int p = 0; p < P; ++p) {
        if (VEC == 4) { const float4 t = *reinterpret_cast<const float4*>(in + p * HW + pix); acc[0] += t.x; acc[1 % VEC] += t.y; acc[2 % VEC] += t.z; acc[3 % VEC] += t.w; }
        else if (VEC == 2) { const float2 t = *reinterpret_cast<const float2*>(in + p * HW + pix); acc[0] += t.x; acc[1 % VEC] += t.y; }
        else acc[0] += in[p * HW + pix];
    }
    for (int q = 0; q < Q; ++q) {
        if (VEC == 4) *reinterpret_cast<float4*>(out + q * HW + pix) = make_float4(acc[0] + q, acc[1 % VEC], acc[2 % VEC], acc[3 % VEC]);
        else if (VEC == 2) *reinterpret_cast<float2*>(out + q * HW + pix) = make_float2(acc[0] + q, acc[1 % VEC]);
        else out[q * HW + pix] = acc[0] + q;
    }
}
int main() {
    const int H = 2304, W = 3840; const size_t HW = (size_t)H * W;
    float *in, *out; hipMalloc(&in, 26 * HW * 4); hipMalloc(&out, 18 * HW * 4); hipMemset(in, 0, 26 * HW * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int cfg[][2] = {{1, 1}, {6, 0}, {6, 3}, {26, 0}, {26, 4}, {6, 18}, {0, 18}, {18, 3}};
    for (auto& c : cfg) for (int mode = 0; mode < 5; ++mode) {
        const int P = c[0], Q = c[1];
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL((k<1, 4>), dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, 0, in, out, P, Q, H, W);
            if (mode == 1) hipLaunchKernelGGL((k<1, 1>), dim3((W + 255) / 256, H), dim3(256), 0, 0, in, out, P, Q, H, W);
            if (mode == 2) hipLaunchKernelGGL((k<4, 1>), dim3((W + 1023) / 1024, H), dim3(256), 0, 0, in, out, P, Q, H, W);
            if (mode == 3) hipLaunchKernelGGL((k<2, 8>), dim3((W + 63) / 64, (H + 7) / 8), dim3(256), 0, 0, in, out, P, Q, H, W);
            if (mode == 4) hipLaunchKernelGGL((k<4, 8>), dim3((W + 127) / 128, (H + 7) / 8), dim3(256), 0, 0, in, out, P, Q, H, W);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        printf("P %2d Q %2d %-22s: %7.1f us  %.2f TB/s\n", P, Q, mode == 0 ? "4 B/lane, 64x4 tile" : mode == 1 ? "4 B/lane, 256x1 tile" : mode == 2 ? "16 B/lane, 1024x1 tile" : mode == 3 ? "8 B/lane, 64x8 tile" : "16 B/lane, 128x8 tile",
               ms * 1e3, (P + Q) * HW * 4 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
