// Definitions shared by the destination-owned splat kernels (splat_tile_kernels.hip: band / LDS-f32-atomic tiles;
// splat_acc64_kernels.hip: LDS-f64-atomic tiles): the source-block geometry of the flow-bounds tables, the splat geometry of
// one source pixel and the conservative reach test.
#pragma once
#include "common.h"

#define ST_BW 64                    // source block
#define ST_BH 4
#define ST_SBX 4                    // blocks per super-block
#define ST_SBY 16
#define ST_SB_BLOCKS (ST_SBX * ST_SBY)
#define ST_Q 1024                   // block queue entries per tile
#define ST_SBQ 64                   // super-block queue entries per tile
#define ST_U 4                      // source blocks in flight per iteration (memory-level parallelism)

struct StGeom {
    int   x0, y0;
    float wnw, wne, wsw, wse;
};

// identical arithmetic to splat_geom (warp_kernels.hip): softSplat.py:23-38
__device__ __forceinline__ StGeom st_geom(int x, int y, float fx, float fy, int W, int H) {
#pragma clang fp contract(off)
    StGeom g;
    float ox = (float)x + fx;
    float oy = (float)y + fy;
    float xf = floorf(ox), yf = floorf(oy);
    float x1 = xf + 1.0f, y1 = yf + 1.0f;
    g.wnw = (x1 - ox) * (y1 - oy);
    g.wne = (ox - xf) * (y1 - oy);
    g.wsw = (x1 - ox) * (oy - yf);
    g.wse = (ox - xf) * (oy - yf);
    xf = fminf(fmaxf(xf, -2.0f), (float)W + 1.0f);
    yf = fminf(fmaxf(yf, -2.0f), (float)H + 1.0f);
    g.x0 = (int)xf; g.y0 = (int)yf;
    return g;
}

// acc / norm of the splat normalisation (softSplat.py:343-349) for SEVERAL accumulators over one normaliser: the reciprocal once
// (v_rcp_f32 + one Newton step), then per value q = v r, one exact residual v - q n and one correction (Markstein): the correctly
// rounded quotient except on rare double-rounding ties (<= 1 ulp there) — 3 full-rate instructions per value instead of the ~10 of the
// IEEE division expansion (16 quotients per cell in the feature configuration: a third of that kernel's finish).  Normalisers outside
// [2^-100, 2^100] (never a sum of bilinear weights x exp(metric) of this model) take the true division: wave-uniform branch in the callers.
struct StRecip { float n, r; };
__device__ __forceinline__ StRecip st_recip(float n) {
    StRecip q;
    q.n = n;
    const float r0 = __builtin_amdgcn_rcpf(n);
    q.r = __builtin_fmaf(__builtin_fmaf(-n, r0, 1.0f), r0, r0);
    return q;
}
__device__ __forceinline__ bool st_recip_safe(float n) { const float a = fabsf(n); return a >= 7.888609e-31f && a <= 1.2676506e30f; }
__device__ __forceinline__ float st_div(float v, const StRecip& d) {
    const float q = v * d.r;
    return __builtin_fmaf(__builtin_fmaf(-q, d.n, v), d.r, q);
}

// Can a source region [rx0, rx1] x [ry0, ry1] (inclusive pixel coordinates) with flow bounds b touch the tile?
// Target corner columns of a source: floor(x + fx) and floor(x + fx) + 1.  Conservative by one cell.
__device__ __forceinline__ bool st_match(const float4 b, float rx0, float rx1, float ry0, float ry1, float tx0, float tx1,
                                         float ty0, float ty1) {
    return (rx1 + b.y >= tx0 - 2.0f) && (rx0 + b.x <= tx1 + 1.0f) && (ry1 + b.w >= ty0 - 2.0f) && (ry0 + b.z <= ty1 + 1.0f);
}


// Flow-bounds table of a flow that is the bilinear upsampling of a low-resolution field (flow = up(scale * lo) * mul), computed from the
// low-resolution field alone: one workgroup of ST_UP_WAVES waves per super-block (see splat_tile_kernels.hip for the derivation).
// (a device function: the kernel of splat_tile_kernels.hip and, round 4, the fused flow-resize + bounds launch of warp_kernels.hip)
#define ST_UP_WAVES 16
__device__ __forceinline__ void splat_bounds_up_body(const float* __restrict__ lo, int64_t lo_bstride, const float* __restrict__ tv,
                                                     int smode, float mul, float* __restrict__ blk, float* __restrict__ sbt,
                                                     int h, int w, int H, int W, float sy, float sx, int nsb_x, int nsb,
                                                     int pair, int N, int sb, int n, float (*red)[4]) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tab = n;                                                  // table sample
    if (pair) {
        const int k = n / N;
        n -= k * N;
        const bool second_half = (pair == 1) == (k == 0);             // channels 2-3 (flow_01)
        lo += second_half ? 2 * (int64_t)h * w : 0;
        smode = pair == 1 ? (k == 0 ? 1 : 2) : 0;
    }
    const float INF = __builtin_inff();
    const float scale = smode == 0 ? 1.0f : (smode == 1 ? tv[n] : 1.0f - tv[n]);
    const float* px = lo + (int64_t)n * lo_bstride;
    const float* py = px + (int64_t)h * w;
    float sxmin = INF, sxmax = -INF, symin = INF, symax = -INF;
    // Footprint of a block: at most ST_BH + 1 rows and ST_BW + 2 columns when upsampling (H >= h, W >= w: host-checked).  The
    // loads of ALL FOUR blocks of the wave go out first, from clamped indices instead of loop bounds (a repeated pixel does not
    // change a minimum; blocks outside the image read block 0's footprint and are discarded): with data-dependent loop bounds, or
    // with the blocks one after the other, every round waited for its own loads.  (A workgroup still needs ~10 us whatever
    // its footprint — sixteen waves, four shuffle reductions each, one barrier — see the lane-per-block kernel below.)
    constexpr int NBW = ST_SB_BLOCKS / ST_UP_WAVES, NV = (ST_BH + 1) * 2;
    float vx[NBW][NV], vy[NBW][NV];
    bool live[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        const int b = wv + j * ST_UP_WAVES;                             // block of the super-block (wave-uniform)
        int X0 = ((sb % nsb_x) * ST_SBX + b % ST_SBX) * ST_BW, Y0 = ((sb / nsb_x) * ST_SBY + b / ST_SBX) * ST_BH;
        live[j] = X0 < W && Y0 < H;
        if (!live[j]) { X0 = 0; Y0 = 0; }
        int c0, c1, r0, r1, d0, d1; float l;
        fldr_lin_src(X0, sx, w, c0, d0, l); fldr_lin_src(min(X0 + ST_BW - 1, W - 1), sx, w, d1, c1, l);
        fldr_lin_src(Y0, sy, h, r0, d0, l); fldr_lin_src(min(Y0 + ST_BH - 1, H - 1), sy, h, d1, r1, l);
#pragma unroll
        for (int k = 0; k < ST_BH + 1; ++k) {
            const int r = min(r0 + k, r1);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = min(c0 + lane + 64 * i, c1);
                vx[j][k * 2 + i] = px[(int64_t)r * w + c]; vy[j][k * 2 + i] = py[(int64_t)r * w + c];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        const int b = wv + j * ST_UP_WAVES;
        float xmin = INF, xmax = -INF, ymin = INF, ymax = -INF;
        if (live[j]) {                                                  // wave-uniform
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const float ax = (scale * vx[j][k]) * mul, ay = (scale * vy[j][k]) * mul;
                xmin = fminf(xmin, ax); xmax = fmaxf(xmax, ax); ymin = fminf(ymin, ay); ymax = fmaxf(ymax, ay);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                xmin = fminf(xmin, __shfl_xor(xmin, o)); xmax = fmaxf(xmax, __shfl_xor(xmax, o));
                ymin = fminf(ymin, __shfl_xor(ymin, o)); ymax = fmaxf(ymax, __shfl_xor(ymax, o));
            }
            const float ex = fmaxf(fabsf(xmin), fabsf(xmax)) * 2.0e-6f, ey = fmaxf(fabsf(ymin), fabsf(ymax)) * 2.0e-6f;
            xmin -= ex; xmax += ex; ymin -= ey; ymax += ey;
        }
        if (lane == 0) *reinterpret_cast<float4*>(blk + (((int64_t)tab * nsb + sb) * ST_SB_BLOCKS + b) * 4) = make_float4(xmin, xmax, ymin, ymax);
        sxmin = fminf(sxmin, xmin); sxmax = fmaxf(sxmax, xmax); symin = fminf(symin, ymin); symax = fmaxf(symax, ymax);
    }
    if (lane == 0) { red[wv][0] = sxmin; red[wv][1] = sxmax; red[wv][2] = symin; red[wv][3] = symax; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 1; k < ST_UP_WAVES; ++k) {
            sxmin = fminf(sxmin, red[k][0]); sxmax = fmaxf(sxmax, red[k][1]); symin = fminf(symin, red[k][2]); symax = fmaxf(symax, red[k][3]);
        }
        *reinterpret_cast<float4*>(sbt + ((int64_t)tab * nsb + sb) * 4) = make_float4(sxmin, sxmax, symin, symax);
    }
}



// splat_tile_kernels.hip: the exact bounds pre-pass over a flow tensor (samples flow_bstride floats apart)
void fldr_splat_bounds_launch(const float* flow, int64_t flow_bstride, float* blk, float* sbt, int N, int H, int W, int nsb_x, int nsb,
                              hipStream_t s);
