// Low-dimensional feature projection (pca_comp.py:473-528): an 8x8 / stride-8 block projection onto K<=16
// components in fp64, followed by a GLOBAL min/max rescale to [-1,1].
//
// HBM-bound: 64 fp32 pixels in, K values out per block.  One thread per 8x8 block; a wave covers 64
// consecutive blocks of one block-row, so each of the 8 row reads of a wave is a contiguous 2-KiB
// segment (2 x dwordx4 per lane) and each component store is a contiguous 256-B (fp32) segment.  The
// projection matrix is read through wave-uniform (scalar) loads.  Two passes over the planes instead of
// one pass + an fp64 intermediate: the second read of a 4K level-0 input (212 MB) is served largely from
// the 256-MiB Infinity Cache, and no P*K*H*W/64 fp64 scratch is needed.
#include "common.h"

#define PCA_MAXK 16

// Hardware fp64 atomics at the L2 (global_atomic_min_f64 / max_f64), fire and forget: see pca_pyramid_kernels.hip for what
// the compare-and-swap loop they replace cost.
__device__ __forceinline__ void atomic_min_f64(double* addr, double v) {
    (void)__hip_atomic_fetch_min(addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void atomic_max_f64(double* addr, double v) {
    (void)__hip_atomic_fetch_max(addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void pca_init_minmax(double* mm) { mm[0] = 1.0e300; mm[1] = -1.0e300; }

// y[k] for the 8x8 block (bx,by) of plane p, fp64; same code in both passes => bit-identical values.
template <int K>
__device__ __forceinline__ void pca_block(const float* __restrict__ plane, int W, int bx, int by,
                                          const double* __restrict__ ev, const double* __restrict__ mean,
                                          const double* __restrict__ mv, double (&y)[K]) {
#pragma unroll
    for (int k = 0; k < K; ++k) y[k] = 0.0;
    const float* p = plane + (int64_t)by * 8 * W + (int64_t)bx * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float4 a = *reinterpret_cast<const float4*>(p + (int64_t)i * W);
        float4 b = *reinterpret_cast<const float4*>(p + (int64_t)i * W + 4);
        double d[8] = {(double)a.x, (double)a.y, (double)a.z, (double)a.w, (double)b.x, (double)b.y, (double)b.z, (double)b.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] -= mean[i * 8 + j];           // pca_comp.py:502
#pragma unroll
        for (int k = 0; k < K; ++k) {
#pragma unroll
            for (int j = 0; j < 8; ++j) y[k] = fma(d[j], ev[k * 64 + i * 8 + j], y[k]);   // :507
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) y[k] = y[k] / mv[k];                  // :511
}

template <int K, int PASS>
__global__ __launch_bounds__(256) void pca_kernel(const float* __restrict__ planes, const double* __restrict__ ev,
                                                  const double* __restrict__ mean, const double* __restrict__ mv,
                                                  double* __restrict__ mm, float* __restrict__ out32,
                                                  double* __restrict__ out64, int H, int W, int Ktot, int nq) {
    // K components per thread; a plane's Ktot = K * nq components are spread over nq threads (blockIdx.z = p * nq + q):
    // on the coarse pyramid levels one thread per block is a single wave's chain of 1,024 dependent fp64 FMAs
    const int BW = W >> 3, BH = H >> 3;
    int bx = blockIdx.x * 64 + (threadIdx.x & 63);
    int by = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int p = blockIdx.z / nq, k0 = (blockIdx.z - p * nq) * K;
    bool live = bx < BW && by < BH;
    double y[K];
    if (live) pca_block<K>(planes + (int64_t)p * H * W, W, bx, by, ev + k0 * 64, mean, mv + k0, y);
    if (PASS == 0 || PASS == 2) {
        double lo = 1.0e300, hi = -1.0e300;
        if (live) {
#pragma unroll
            for (int k = 0; k < K; ++k) { lo = y[k] < lo ? y[k] : lo; hi = y[k] > hi ? y[k] : hi; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {                        // wave64 shuffle reduction
            double ol = __shfl_xor(lo, off), oh = __shfl_xor(hi, off);
            lo = ol < lo ? ol : lo; hi = oh > hi ? oh : hi;
        }
        __shared__ double slo[4], shi[4];
        if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int i = 1; i < 4; ++i) { lo = slo[i] < lo ? slo[i] : lo; hi = shi[i] > hi ? shi[i] : hi; }
            atomic_min_f64(mm, lo);
            atomic_max_f64(mm + 1, hi);
        }
        if (PASS == 2 && live) {                                        // raw projections for pca_rescale_kernel
            const int64_t BHW = (int64_t)BH * BW;
            const int64_t o = ((int64_t)p * Ktot + k0) * BHW + (int64_t)by * BW + bx;
#pragma unroll
            for (int k = 0; k < K; ++k) out64[o + (int64_t)k * BHW] = y[k];
        }
    } else {
#pragma clang fp contract(off)
        if (!live) return;
        const double mi = mm[0], range = mm[1] - mm[0];
        const int64_t BHW = (int64_t)BH * BW;
        const int64_t o = ((int64_t)p * Ktot + k0) * BHW + (int64_t)by * BW + bx;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double v = ((y[k] - mi) / range) * 2.0 - 1.0;               // pca_comp.py:523-526
            if (out32) out32[o + (int64_t)k * BHW] = (float)v;          // fLDRnet.py:146 .float()
            if (out64) out64[o + (int64_t)k * BHW] = v;
        }
    }
}

// Second pass when the caller provides the fp64 output buffer: pass 0 leaves the raw projections y there and this
// kernel rescales them in place (pca_comp.py:521-526) and emits the fp32 cast (fLDRnet.py:146) and, optionally, the
// split-packed twin the convolutions consume — a pure stream (8 B in, 8 + 4 [+ 4] B out per value) instead of a second
// projection of the input planes.  The arithmetic on y is the one pass 1 applies, so results are identical.
// One thread = one feature-map pixel x 8 consecutive channels (one split-packed group).
__global__ __launch_bounds__(256) void pca_rescale_kernel(double* __restrict__ y64, const double* __restrict__ mm,
                                                          float* __restrict__ out32, unsigned char* __restrict__ spk,
                                                          int C, int64_t BHW) {
#pragma clang fp contract(off)
    const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int g = blockIdx.y;
    if (pix >= BHW) return;
    const double mi = mm[0], range = mm[1] - mm[0];
    float f[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = g * 8 + k;
        f[k] = 0.0f;
        if (c < C) {
            const int64_t o = (int64_t)c * BHW + pix;
            const double v = ((y64[o] - mi) / range) * 2.0 - 1.0;
            y64[o] = v;
            f[k] = (float)v;
            if (out32) out32[o] = f[k];
        }
    }
    if (spk) {
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        h8 hi, lo;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float t = __uint_as_float(__float_as_uint(f[k]) & 0xFFFFE000u);      // the split of conv_spk_kernels.hip
            hi[k] = (_Float16)t;
            lo[k] = (_Float16)(f[k] - t);
        }
        unsigned char* d = spk + ((int64_t)g * 2 * BHW + pix) * 16;
        *reinterpret_cast<h8*>(d) = hi;
        *reinterpret_cast<h8*>(d + BHW * 16) = lo;
    }
}

template <int K>
static void pca_launch_stream(const float* planes, const double* ev, const double* mean, const double* mv, float* o32,
                              double* o64, void* spk, double* mm, int P, int H, int W, hipStream_t s) {
    dim3 grid(fldr_cdiv(W / 8, 64), fldr_cdiv(H / 8, 4), P);
    hipLaunchKernelGGL(pca_init_minmax, dim3(1), dim3(1), 0, s, mm);
    if ((int64_t)grid.x * grid.y * P < 64 && K % 4 == 0 && K > 4) {        // coarse levels (<= 36 x 60 blocks): 4 components per thread
        grid.z = P * (K / 4);                                               // (measured 19/17/16 -> 11/11/10 us; 2 per thread is slower again)
        hipLaunchKernelGGL((pca_kernel<4, 2>), grid, dim3(256), 0, s, planes, ev, mean, mv, mm, (float*)nullptr, o64, H, W, K, K / 4);
    } else {
        hipLaunchKernelGGL((pca_kernel<K, 2>), grid, dim3(256), 0, s, planes, ev, mean, mv, mm, (float*)nullptr, o64, H, W, K, 1);
    }
    const int64_t BHW = (int64_t)(H / 8) * (W / 8);
    const int C = P * K;
    hipLaunchKernelGGL(pca_rescale_kernel, dim3(fldr_cdiv(BHW, 256), (C + 7) / 8), dim3(256), 0, s, o64, mm, o32,
                       reinterpret_cast<unsigned char*>(spk), C, BHW);
}

template <int K>
static void pca_launch(const float* planes, const double* ev, const double* mean, const double* mv, float* o32, double* o64,
                       double* mm, int P, int H, int W, hipStream_t s) {
    dim3 grid(fldr_cdiv(W / 8, 64), fldr_cdiv(H / 8, 4), P);
    hipLaunchKernelGGL(pca_init_minmax, dim3(1), dim3(1), 0, s, mm);
    hipLaunchKernelGGL((pca_kernel<K, 0>), grid, dim3(256), 0, s, planes, ev, mean, mv, mm, o32, o64, H, W, K, 1);
    hipLaunchKernelGGL((pca_kernel<K, 1>), grid, dim3(256), 0, s, planes, ev, mean, mv, mm, o32, o64, H, W, K, 1);
}

// fldr_pca_project with the fp64 output buffer as the intermediate (one projection instead of two) and an optional
// split-packed twin of the fp32 output (fldr_spk_bytes(P*K, H/8, W/8) bytes) for the convolutions that follow.
extern "C" int fldr_pca_project_stream(const float* planes, const double* ev, const double* mean, const double* meanvec,
                                       float* out_f32_or_null, double* out_f64, void* out_spk_or_null, double* minmax_ws,
                                       int P, int K, int H, int W, fldr_stream_t stream) {
    FLDR_CHECK_ARG(planes && ev && mean && meanvec && minmax_ws && out_f64 && P > 0 && H > 0 && W > 0);
    if (H % 8 != 0 || W % 8 != 0) return FLDR_E_SHAPE;
    if (((uintptr_t)planes & 15) != 0) return FLDR_E_ARG;
    hipStream_t s = fldr_s(stream);
    switch (K) {
        case 16: pca_launch_stream<16>(planes, ev, mean, meanvec, out_f32_or_null, out_f64, out_spk_or_null, minmax_ws, P, H, W, s); break;
        case 8:  pca_launch_stream<8>(planes, ev, mean, meanvec, out_f32_or_null, out_f64, out_spk_or_null, minmax_ws, P, H, W, s); break;
        case 4:  pca_launch_stream<4>(planes, ev, mean, meanvec, out_f32_or_null, out_f64, out_spk_or_null, minmax_ws, P, H, W, s); break;
        default: return FLDR_E_ARG;
    }
    FLDR_LAUNCH_RET();
}

extern "C" int fldr_pca_project(const float* planes, const double* ev, const double* mean, const double* meanvec,
                                float* out_f32, double* out_f64_or_null, double* minmax_ws,
                                int P, int K, int H, int W, fldr_stream_t stream) {
    FLDR_CHECK_ARG(planes && ev && mean && meanvec && minmax_ws && (out_f32 || out_f64_or_null) && P > 0 && H > 0 && W > 0);
    if (H % 8 != 0 || W % 8 != 0) return FLDR_E_SHAPE;                // pca_comp.py:486-487
    if (((uintptr_t)planes & 15) != 0) return FLDR_E_ARG;              // dwordx4 row loads
    hipStream_t s = fldr_s(stream);
    switch (K) {
        case 16: pca_launch<16>(planes, ev, mean, meanvec, out_f32, out_f64_or_null, minmax_ws, P, H, W, s); break;
        case 8:  pca_launch<8>(planes, ev, mean, meanvec, out_f32, out_f64_or_null, minmax_ws, P, H, W, s); break;
        case 4:  pca_launch<4>(planes, ev, mean, meanvec, out_f32, out_f64_or_null, minmax_ws, P, H, W, s); break;
        default: return FLDR_E_ARG;
    }
    FLDR_LAUNCH_RET();
}
