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

// ------------------------------------------------------------------------------------------------
// Round 3: asynchronous staging.  The kernel above stages synchronously (load -> LDS -> barrier -> FMAs -> barrier, nothing in
// flight during the FMAs: 0.28 of the HBM rate at the largest PWC level, 0.01-0.15 on the small ones).  Here three more waves
// are LOADERS: they move chunk k + 1 global -> LDS by LDS-DMA (global_load_lds_dwordx4, 28 instructions of 1 KB per 8-channel
// chunk) into the other of two stages and wait for it while the nine compute waves run the FMAs of chunk k; one barrier per
// chunk hands the stage over.  (The DMAs cannot be issued by the compute waves themselves: the compiler orders every LDS read behind
// all outstanding LDS-DMA of its wave — s_waitcnt vmcnt(0) in front of the first ds_read — so nothing would overlap.)  Quads
// outside the image (and channels past C) read a 16-byte zero block instead (the zero padding of correlation.py:297-298).
// Needs W % 4 == 0 and 16-byte aligned tensors (whole quads are then inside or outside the image); anything else takes the
// kernel above.  Same arithmetic and order per accumulator as above: identical results.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) const void* corr_gptr_t;
typedef __attribute__((address_space(3))) void* corr_lptr_t;
__device__ __attribute__((aligned(16))) float corr_zero_block[4];

#ifndef CORR_NLOAD
#define CORR_NLOAD 3                   // loader waves (one wave issuing all 28 DMAs of a chunk took 2 us per chunk: the compute waves need 1)
#endif
struct CorrGrid { int tiles_x, per_sample, total, per_xcd; };      // per_xcd 0: plain row-major grid (test hook)

template <int CC>
__global__ __launch_bounds__(64 * (9 + CORR_NLOAD)) void correlation_dma_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              float* __restrict__ out, int C, int H, int W, CorrGrid gr) {
    constexpr int NA = CC * CORR_TH * (CORR_TW / 4), NB = CC * CORR_BH * (CORR_BW / 4);     // 16-byte pieces of a chunk
    constexpr int STAGE = (NA + NB) * 16;
    constexpr int NJ = (NA + NB) / 64;                                    // DMA wave-instructions per chunk
    constexpr int NJL = (NJ + CORR_NLOAD - 1) / CORR_NLOAD;               // ... per loader wave
    static_assert(NA % 64 == 0 && NB % 64 == 0, "whole wave-instructions of pieces");
    extern __shared__ __attribute__((aligned(16))) unsigned char corr_smem[];              // 2 stages of {sa[CC][TH][TW], sb[CC][BH][BW]}
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);            // 0..8: compute wave of dy = wave - 4; 9..: loaders
    const int lane = tid & 63;
    // Tiles are dealt to the XCDs in contiguous row-major ranges (workgroup b runs on XCD b & 7): a tile's 16 x 40 window of the
    // second map overlaps its neighbours' (2.5 x the tile's own pixels), and neighbours then share it in one L2.
    const int lin = gr.per_xcd ? (int)(blockIdx.x & 7) * gr.per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if ((gr.per_xcd && (int)(blockIdx.x >> 3) >= gr.per_xcd) || lin >= gr.total) return;      // workgroup-uniform, before any barrier
    const int n = lin / gr.per_sample, trem = lin - n * gr.per_sample, tyi = trem / gr.tiles_x;
    const int x0 = (trem - tyi * gr.tiles_x) * CORR_TW, y0 = tyi * CORR_TH;
    const int64_t HW = (int64_t)H * W;

    if (wave >= 9) {
        // ================================= loaders =================================
        // loader w issues the wave-instructions j = w, w + NLOAD, ... of every chunk; the geometry of its pieces (channel inside the
        // chunk, plane offset of the quad or -1 outside the image) does not depend on the chunk
        const int lw = wave - 9;
        const float* an = a + (int64_t)n * C * HW;
        const float* bn = b + (int64_t)n * C * HW;
        const float* zero = corr_zero_block;
        int pch[NJL], poff[NJL];
#pragma unroll
        for (int i = 0; i < NJL; ++i) {
            const int j = lw + i * CORR_NLOAD;
            const int e = j * 64 + lane;
            if (e < NA) {
                pch[i] = e / (CORR_TH * (CORR_TW / 4));
                const int r = e % (CORR_TH * (CORR_TW / 4));
                const int y = y0 + r / (CORR_TW / 4), x = x0 + (r % (CORR_TW / 4)) * 4;
                poff[i] = (y < H && x < W) ? y * W + x : -1;
            } else {
                const int eb = e - NA;
                pch[i] = eb / (CORR_BH * (CORR_BW / 4));
                const int r = eb % (CORR_BH * (CORR_BW / 4));
                const int y = y0 - 4 + r / (CORR_BW / 4), x = x0 - 4 + (r % (CORR_BW / 4)) * 4;
                poff[i] = (y >= 0 && y < H && x >= 0 && x < W) ? y * W + x : -1;
            }
        }
        auto issue = [&](int c0, int buf) __attribute__((always_inline)) {
            unsigned char* st = corr_smem + buf * STAGE;
#pragma unroll
            for (int i = 0; i < NJL; ++i) {
                const int j = lw + i * CORR_NLOAD;
                if (j < NJ) {                                             // wave-uniform
                    const float* pl = j * 64 < NA ? an : bn;
                    const float* src = (poff[i] >= 0 && c0 + pch[i] < C) ? pl + (int64_t)(c0 + pch[i]) * HW + poff[i] : zero;
                    __builtin_amdgcn_global_load_lds((corr_gptr_t)src, (corr_lptr_t)(st + j * 1024), 16, 0, 0);
                }
            }
        };
        issue(0, 0);
        __builtin_amdgcn_s_waitcnt(0x0F70);                               // vmcnt(0): my part of chunk 0 has landed
        __builtin_amdgcn_s_barrier();
        int buf = 0;
        for (int c0 = 0; c0 < C; c0 += CC) {
            if (c0 + CC < C) { issue(c0 + CC, buf ^ 1); __builtin_amdgcn_s_waitcnt(0x0F70); }    // fetched while the compute waves work on chunk c0
            __builtin_amdgcn_s_barrier();                                 // chunk c0 consumed by everybody, chunk c0 + CC in place
            buf ^= 1;
        }
        return;
    }

    // ================================= compute waves =================================
    const int dyi = wave;
    const int row = lane >> 3, quad = lane & 7;
    float acc[9][4];
#pragma unroll
    for (int d = 0; d < 9; ++d)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[d][p] = 0.0f;
    __builtin_amdgcn_s_barrier();                                         // chunk 0 is in stage 0
    int buf = 0;
    for (int c0 = 0; c0 < C; c0 += CC) {
        const float4* sa4 = reinterpret_cast<const float4*>(corr_smem + buf * STAGE);
        const float4* sb4 = sa4 + NA;
#pragma unroll 4
        for (int c = 0; c < CC; ++c) {
            const float4 f1 = sa4[(c * CORR_TH + row) * (CORR_TW / 4) + quad];
            const float4* brow = sb4 + (c * CORR_BH + row + dyi) * (CORR_BW / 4) + quad;
            const float4 g0 = brow[0], g1 = brow[1], g2 = brow[2];
            const float f[4] = {f1.x, f1.y, f1.z, f1.w};
            const float g[12] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w, g2.x, g2.y, g2.z, g2.w};
#pragma unroll
            for (int d = 0; d < 9; ++d)
#pragma unroll
                for (int p = 0; p < 4; ++p) acc[d][p] = fmaf(f[p], g[p + d], acc[d][p]);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                               // lgkmcnt(0): my LDS reads of this stage are done before the loader may refill it
        __builtin_amdgcn_s_barrier();
        buf ^= 1;
    }
    const int y = y0 + row, x = x0 + quad * 4;
    if (y >= H || x >= W) return;
    const float cf = (float)C, rcf = 1.0f / cf;
    float* on = out + (int64_t)n * 81 * HW + (int64_t)y * W + x;
#pragma unroll
    for (int d = 0; d < 9; ++d) {
        float q[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) q[p] = fldr_div_by(acc[d][p], cf, rcf);       // == acc / C (correlation.py:108)
        *reinterpret_cast<float4*>(on + (int64_t)(dyi * 9 + d) * HW) = make_float4(q[0], q[1], q[2], q[3]);   // channel (dy+4)*9 + (dx+4)
    }
}

static int g_corr_variant = 1;                       // 1: LDS-DMA double buffer where the shape allows; 0: the synchronous kernel
static int g_corr_cc = 8;
static int g_corr_xcd = 1;                           // 1: contiguous tile ranges per XCD; 0: plain row-major grid
FLDR_HOOK int fldr_debug_corr_xcd(int v) { if (v == 0 || v == 1) g_corr_xcd = v; return g_corr_xcd; }
FLDR_HOOK int fldr_debug_corr_variant(int v) { if (v == 0 || v == 1) g_corr_variant = v; return g_corr_variant; }
FLDR_HOOK int fldr_debug_corr_chunk(int v) { if (v == 8 || v == 16) g_corr_cc = v; return g_corr_cc; }

template <int CC>
static int corr_dma_launch(const float* a, const float* b, float* out, int N, int C, int H, int W, hipStream_t s) {
    constexpr int LDS = 2 * (CC * CORR_TH * (CORR_TW / 4) + CC * CORR_BH * (CORR_BW / 4)) * 16;
    static std::atomic<uint64_t> attr_done{0};
    if (int e = fldr_set_max_lds(reinterpret_cast<const void*>(&correlation_dma_kernel<CC>), LDS, attr_done)) return e;
    CorrGrid gr;
    gr.tiles_x = fldr_cdiv(W, CORR_TW);
    gr.per_sample = gr.tiles_x * fldr_cdiv(H, CORR_TH);
    if ((int64_t)gr.per_sample * N > (1ll << 30)) return FLDR_E_SHAPE;
    gr.total = gr.per_sample * N;
    gr.per_xcd = g_corr_xcd ? (gr.total + 7) / 8 : 0;
    hipLaunchKernelGGL(correlation_dma_kernel<CC>, dim3(g_corr_xcd ? 8 * gr.per_xcd : gr.total), dim3(64 * (9 + CORR_NLOAD)), LDS, s, a, b, out, C, H, W, gr);
    return 0;
}

extern "C" int fldr_correlation_fwd(const float* a, const float* b, float* out, int N, int C, int H, int W,
                                    fldr_stream_t stream) {
    FLDR_CHECK_ARG(a && b && out && N > 0 && C > 0 && H > 0 && W > 0);
    const bool vec4 = (W & 3) == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (g_corr_variant == 1 && vec4) {
        if (int e = g_corr_cc == 16 ? corr_dma_launch<16>(a, b, out, N, C, H, W, fldr_s(stream)) : corr_dma_launch<8>(a, b, out, N, C, H, W, fldr_s(stream))) return e;
        FLDR_LAUNCH_RET();
    }
    dim3 grid(fldr_cdiv(W, CORR_TW), fldr_cdiv(H, CORR_TH), N);
    hipLaunchKernelGGL(correlation_kernel, grid, dim3(576), 0, fldr_s(stream), a, b, out, C, H, W);
    FLDR_LAUNCH_RET();
}
