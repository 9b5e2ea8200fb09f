// PWC cost volume (OpticalFlow/correlation.py, forward): 81 displacements, radius 4, mean over channels.
//
// The reference launches one 32-thread block per pixel, re-reads the 9x9xC neighbourhood from global
// memory for every pixel and lets thread 0 add the 32 partial sums serially (correlation.py:44-112).
// Here a 576-thread workgroup (9 waves) owns an 8x32-pixel tile: per 8-channel chunk the 8x32 tile of
// `a` and the 16x40 window of `b` (tile + radius-4 halo, zero padded) are staged once in LDS; wave `w`
// owns displacement row dy = w-4, lane = (row, pixel-quad), and every lane keeps a 4-pixel x 9-dx
// register tile (36 accumulators), so one channel step costs 4 ds_read_b128 for 36 FMAs.  No NHWC
// rearranged copies (the reference's rbot0/rbot1) are made; inputs stay NCHW and are read coalesced.
// HBM-bound by its output (81 planes out for 2 C planes in): rocprof round 2 (PWC-Net shapes of a 4K pair, N = 2):
// 617 us for the 544x960x32 level = 605 MB -> 1.0 TB/s with scalar stores / staging loads and 16-channel chunks, 267 us
// (2.3 TB/s) now; the stores are 16-byte
// (4 pixels per lane and displacement) and the staging loads 16-byte where the row alignment allows, the 36 divisions by C
// one exact reciprocal-multiply-correct sequence each (fldr_div_by: the quotient of a true division).
#include "common.h"

#define CORR_TH 8
#define CORR_TW 32
#ifndef CORR_CC
#define CORR_CC 8                  // channels per staged chunk: 28 KB of LDS -> 4 workgroups per CU (16: 2 per CU, 343 us at 544x960x32; 8: 267; 4: 422)
#endif
#define CORR_BH (CORR_TH + 8)
#define CORR_BW (CORR_TW + 8)

__global__ __launch_bounds__(576) void correlation_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          float* __restrict__ out, int C, int H, int W) {
    __shared__ __attribute__((aligned(16))) float sa[CORR_CC][CORR_TH][CORR_TW];
    __shared__ __attribute__((aligned(16))) float sb[CORR_CC][CORR_BH][CORR_BW];
    const int tid = threadIdx.x;
    const int dyi = tid >> 6;              // 0..8  -> dy = dyi - 4   (wave-uniform)
    const int lane = tid & 63;
    const int row = lane >> 3, quad = lane & 7;
    const int x0 = blockIdx.x * CORR_TW, y0 = blockIdx.y * CORR_TH, n = blockIdx.z;
    const int64_t HW = (int64_t)H * W;
    const float* an = a + (int64_t)n * C * HW;
    const float* bn = b + (int64_t)n * C * HW;
    const bool vec4 = (W & 3) == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;   // uniform

    float acc[9][4];
#pragma unroll
    for (int d = 0; d < 9; ++d)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[d][p] = 0.0f;

    for (int c0 = 0; c0 < C; c0 += CORR_CC) {
        // staging in quads of 4 consecutive pixels: one 16-byte load where the quad is inside the image and W % 4 == 0
        // (tile and window origins are multiples of 4), element-wise with zero fill at the borders otherwise
        for (int e = tid; e < CORR_CC * CORR_TH * (CORR_TW / 4); e += 576) {
            const int c = e / (CORR_TH * (CORR_TW / 4)), r = e % (CORR_TH * (CORR_TW / 4));
            const int y = y0 + r / (CORR_TW / 4), x = x0 + (r % (CORR_TW / 4)) * 4;
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (c0 + c < C && y < H) {
                const float* p = an + (int64_t)(c0 + c) * HW + (int64_t)y * W + x;
                if (vec4 && x + 3 < W) v = *reinterpret_cast<const float4*>(p);
                else { if (x < W) v.x = p[0]; if (x + 1 < W) v.y = p[1]; if (x + 2 < W) v.z = p[2]; if (x + 3 < W) v.w = p[3]; }
            }
            reinterpret_cast<float4*>(&sa[0][0][0])[e] = v;
        }
        for (int e = tid; e < CORR_CC * CORR_BH * (CORR_BW / 4); e += 576) {
            const int c = e / (CORR_BH * (CORR_BW / 4)), r = e % (CORR_BH * (CORR_BW / 4));
            const int y = y0 - 4 + r / (CORR_BW / 4), x = x0 - 4 + (r % (CORR_BW / 4)) * 4;
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);                                   // zero padding: correlation.py:297-298
            if (c0 + c < C && y >= 0 && y < H) {
                const float* p = bn + (int64_t)(c0 + c) * HW + (int64_t)y * W + x;
                if (vec4 && x >= 0 && x + 3 < W) v = *reinterpret_cast<const float4*>(p);
                else { if (x >= 0 && x < W) v.x = p[0]; if (x + 1 >= 0 && x + 1 < W) v.y = p[1]; if (x + 2 >= 0 && x + 2 < W) v.z = p[2]; if (x + 3 >= 0 && x + 3 < W) v.w = p[3]; }
            }
            reinterpret_cast<float4*>(&sb[0][0][0])[e] = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int c = 0; c < CORR_CC; ++c) {
            float4 f1 = *reinterpret_cast<const float4*>(&sa[c][row][quad * 4]);
            const float* brow = &sb[c][row + dyi][quad * 4];
            float4 g0 = *reinterpret_cast<const float4*>(brow);
            float4 g1 = *reinterpret_cast<const float4*>(brow + 4);
            float4 g2 = *reinterpret_cast<const float4*>(brow + 8);
            float f[4] = {f1.x, f1.y, f1.z, f1.w};
            float g[12] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w, g2.x, g2.y, g2.z, g2.w};
#pragma unroll
            for (int d = 0; d < 9; ++d)
#pragma unroll
                for (int p = 0; p < 4; ++p) acc[d][p] = fmaf(f[p], g[p + d], acc[d][p]);
        }
        __syncthreads();
    }
    const int y = y0 + row, x = x0 + quad * 4;
    if (y >= H) return;
    const float cf = (float)C, rcf = 1.0f / cf;
    float* on = out + (int64_t)n * 81 * HW + (int64_t)y * W + x;
#pragma unroll
    for (int d = 0; d < 9; ++d) {
        float* o = on + (int64_t)(dyi * 9 + d) * HW;                  // channel (dy+4)*9 + (dx+4)
        float q[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) q[p] = fldr_div_by(acc[d][p], cf, rcf);       // == acc / C (correlation.py:108)
        if (vec4 && x + 3 < W) *reinterpret_cast<float4*>(o) = make_float4(q[0], q[1], q[2], q[3]);
        else {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (x + p < W) o[p] = q[p];
        }
    }
}

extern "C" int fldr_correlation_fwd(const float* a, const float* b, float* out, int N, int C, int H, int W,
                                    fldr_stream_t stream) {
    FLDR_CHECK_ARG(a && b && out && N > 0 && C > 0 && H > 0 && W > 0);
    dim3 grid(fldr_cdiv(W, CORR_TW), fldr_cdiv(H, CORR_TH), N);
    hipLaunchKernelGGL(correlation_kernel, grid, dim3(576), 0, fldr_s(stream), a, b, out, C, H, W);
    FLDR_LAUNCH_RET();
}
