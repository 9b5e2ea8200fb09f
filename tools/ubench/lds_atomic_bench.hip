// LDS atomic throughput on gfx950: ds_add_f32 vs ds_add_u32 vs ds_write_b32 (conflict-free, one dword per lane).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc) {
    __shared__ float acc[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) acc[i] = 0.0f;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    const int base = (threadIdx.x >> 6) * 1024 + (threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float* p = acc + base + j * 64 + ((it & 1) ? 1 : 0);
            if (OP == 0) atomicAdd(p, 1.0f);
            else if (OP == 1) atomicAdd(reinterpret_cast<unsigned int*>(p), 1u);
            else *reinterpret_cast<volatile float*>(p) = (float)it;
        }
    }
    __syncthreads();
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = acc[threadIdx.x];
}
int main() {
    float* out; unsigned long long* cyc; hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 8);
    const int iters = 1000;
    for (int wgs : {1, 256, 1024}) for (int op = 0; op < 3; ++op) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (op == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, out, iters, cyc);
            if (op == 1) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, out, iters, cyc);
            if (op == 2) hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(256), 0, 0, out, iters, cyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double instr = (double)iters * 16 * 4;     // wave-instructions per workgroup
        printf("wgs %4d op %s: %.1f us, %.1f ref-cycles per wave-instruction per WG (counter), %.2f ns per wave-instr per WG\n", wgs,
               op == 0 ? "ds_add_f32" : op == 1 ? "ds_add_u32" : "ds_write  ", ms * 1e3, (double)c / instr, ms * 1e6 / instr);
    }
    return 0;
}
