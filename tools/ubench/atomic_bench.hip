// Micro-benchmark: float atomicAdd throughput for the splat's access shapes (run on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// each thread: one pixel of a W x H plane, adds to NCH planes at (x + shift) ; variant selects pattern
template <int NCH, int MODE>
__global__ __launch_bounds__(256) void k(float* acc, const float* src, int W, int H, int shift) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    size_t HW = (size_t)W * H;
    float v = src[(size_t)y * W + x];
    int tx = x + shift; if (tx >= W) tx -= W;
    size_t o = (size_t)y * W + tx;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (MODE == 0) atomicAdd(acc + c * HW + o, v);                     // 1 atomic / px / ch
        if (MODE == 1) { acc[c * HW + o] = v; }                            // plain store
        if (MODE == 2) { atomicAdd(acc + c * HW + o, v); atomicAdd(acc + c * HW + o + 1, v);   // 4 corners
                         if (y + 1 < H) { atomicAdd(acc + c * HW + o + W, v); atomicAdd(acc + c * HW + o + W + 1, v); } }
        if (MODE == 3) { float r = atomicAdd(acc + c * HW + o, v); if (r == 12345.f) acc[0] = r; }   // returning
    }
}

int main() {
    const int W = 3840, H = 2304;
    size_t HW = (size_t)W * H;
    float *acc, *src;
    CK(hipMalloc(&acc, HW * 4 * 4 + 4096)); CK(hipMalloc(&src, HW * 4));
    CK(hipMemset(acc, 0, HW * 16)); CK(hipMemset(src, 0, HW * 4));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid((W + 63) / 64, (H + 3) / 4), blk(256);
    auto run = [&](const char* name, auto kern, int shift, double atoms_per_px) {
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, grid, blk, 0, 0, acc, src, W, H, shift);
        hipEventRecord(e0);
        const int n = 10;
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(kern, grid, blk, 0, 0, acc, src, W, H, shift);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= n;
        printf("%-44s shift %2d : %8.1f us  %7.1f GB/s of atomic/store bytes\n", name, shift, ms * 1e3, atoms_per_px * HW * 4 / ms / 1e6);
    };
    run("1 ch, 1 atomic/px (aligned)", k<1, 0>, 0, 1);
    run("1 ch, 1 atomic/px (shift 3)", k<1, 0>, 3, 1);
    run("4 ch, 1 atomic/px/ch (aligned)", k<4, 0>, 0, 4);
    run("4 ch, 1 atomic/px/ch (shift 3)", k<4, 0>, 3, 4);
    run("4 ch, plain store", k<4, 1>, 3, 4);
    run("4 ch, 4 corners (16 atomics/px)", k<4, 2>, 3, 16);
    run("1 ch, 4 corners", k<1, 2>, 3, 4);
    run("4 ch, returning atomic", k<4, 3>, 3, 4);
    return 0;
}
