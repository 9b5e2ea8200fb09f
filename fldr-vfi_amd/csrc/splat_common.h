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

// Can a source region [rx0, rx1] x [ry0, ry1] (inclusive pixel coordinates) with flow bounds b touch the tile?
// Target corner columns of a source: floor(x + fx) and floor(x + fx) + 1.  Conservative by one cell.
__device__ __forceinline__ bool st_match(const float4 b, float rx0, float rx1, float ry0, float ry1, float tx0, float tx1,
                                         float ty0, float ty1) {
    return (rx1 + b.y >= tx0 - 2.0f) && (rx0 + b.x <= tx1 + 1.0f) && (ry1 + b.w >= ty0 - 2.0f) && (ry0 + b.z <= ty1 + 1.0f);
}


// splat_tile_kernels.hip: the exact bounds pre-pass over a flow tensor (samples flow_bstride floats apart)
void fldr_splat_bounds_launch(const float* flow, int64_t flow_bstride, float* blk, float* sbt, int N, int H, int W, int nsb_x, int nsb,
                              hipStream_t s);
