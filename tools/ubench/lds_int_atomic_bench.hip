// LDS atomic throughput on gfx950 by type and address pattern: ds_add_u32 / ds_add_u64 / ds_add_f32 / ds_add_f64.
// Patterns: 0 = one element per lane, consecutive (conflict-free); 1 = neighbouring lanes share an element (2-way same address);
// 2 = pseudo-random elements in a 2048-element window; 3 = all lanes of a wave on 4 elements (heavy same-address).
// Build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -o lds_int_atomic_bench lds_int_atomic_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <typename T> struct One;
template <> struct One<unsigned int> { static __device__ unsigned int v() { return 1u; } };
template <> struct One<unsigned long long> { static __device__ unsigned long long v() { return 1ull; } };
template <> struct One<float> { static __device__ float v() { return 1.0f; } };
template <> struct One<double> { static __device__ double v() { return 1.0; } };

template <typename T, int PAT>
__global__ __launch_bounds__(256) void k(T* out, int iters) {
    __shared__ T acc[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) acc[i] = T(0);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned int h = threadIdx.x * 2654435761u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int idx;
            if (PAT == 0) idx = wave * 1024 + j * 64 + lane + (it & 1);
            else if (PAT == 1) idx = wave * 1024 + j * 64 + (lane >> 1) + (it & 1);
            else if (PAT == 2) { h = h * 1664525u + 1013904223u; idx = (h >> 12) & 2047; }
            else idx = wave * 1024 + (lane & 3) + j * 4;
            atomicAdd(acc + idx, One<T>::v());
        }
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = acc[threadIdx.x];
}

template <typename T, int PAT>
static void run(const char* name, T* out) {
    const int iters = 2000;
    for (int wgs : {256, 1024}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k<T, PAT>), dim3(wgs), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double instr_per_cu = (double)iters * 8 * 4 * (wgs / 256.0);   // wave-instructions per CU (256 CUs)
        printf("%-10s pattern %d wgs %4d: %8.1f us  %.2f ns per wave-instruction per CU\n", name, PAT, wgs, ms * 1e3, ms * 1e6 / instr_per_cu);
    }
}

int main() {
    void* out; hipMalloc(&out, 1024 * 256 * 8);
#define ALLPAT(T, name) run<T, 0>(name, (T*)out); run<T, 1>(name, (T*)out); run<T, 2>(name, (T*)out); run<T, 3>(name, (T*)out);
    ALLPAT(unsigned int, "ds_add_u32")
    ALLPAT(unsigned long long, "ds_add_u64")
    ALLPAT(float, "ds_add_f32")
    ALLPAT(double, "ds_add_f64")
    return 0;
}
