// Softmax splatting (softSplat.py:12-52, 320-352) with DESTINATION-OWNED tiles: no global atomics, no zeroed
// accumulator tensor, no separate normalisation pass.
//
// The strip kernel (warp_kernels.hip) scatters with global float atomics and is bound by the per-CU atomic issue
// rate (~1.3 TB/s of atomic bytes chip-wide: 292 us per 4-channel 2304x3840 splat, plus a memset of the accumulator
// and a normalisation pass).  Here every workgroup OWNS a TW x TH tile of the output: it finds the source pixels
// whose bilinear footprint can touch the tile, accumulates them with LDS atomics (ds_add_f32: conflict-free for
// neighbouring cells), and writes the finished tile ((acc / norm - 0.5) * 2, softSplat.py:343-349) with plain
// coalesced stores.  Sources near tile borders are evaluated by every tile they touch (1.7-2.5x redundant reads and
// geometry, served by L2) — cheap next to the atomics they replace.
//
// Finding the sources: a pre-pass reduces the flow to bounds (min/max of fx, fy) per source block (64 x 4 pixels = one
// 256-B row segment per wave) and per super-block (4 x 16 blocks = 256 x 64 pixels).  A tile tests the super-block
// table in parallel, then the blocks of the matching super-blocks, and queues the blocks whose target bounding box
// overlaps it.  Exact for arbitrary flows: the test is conservative, every queued source is re-tested per corner, and
// a queue overflow (pathological flows that collapse a large area into one tile) falls back to a scan of all blocks.
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

// ------------------------------------------------------------------------------------------------
// pre-pass: flow bounds per block and per super-block
//   blk[n][sb][block in sb][4] = {fxmin, fxmax, fymin, fymax};  sbt[n][sb][4]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void splat_bounds_kernel(const float* __restrict__ flow, float* __restrict__ blk,
                                                           float* __restrict__ sbt, int H, int W, int nsb_x, int nsb) {
    const int sb = blockIdx.x, n = blockIdx.y;
    const int sbx = sb % nsb_x, sby = sb / nsb_x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = sbx * (ST_SBX * ST_BW) + threadIdx.x;
    const int64_t HW = (int64_t)H * W;
    const float* fl = flow + (int64_t)n * 2 * HW;
    const float INF = __builtin_inff();
    __shared__ float red[4][4];
    float sxmin = INF, sxmax = -INF, symin = INF, symax = -INF;
    float* bo = blk + ((int64_t)n * nsb + sb) * ST_SB_BLOCKS * 4;
    // 4 block rows (16 pixel rows) per trip: all 32 loads are issued before the first reduction
    for (int bq = 0; bq < ST_SBY; bq += 4) {
        float vx[4][ST_BH], vy[4][ST_BH];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < ST_BH; ++r) {
                const int y = sby * (ST_SBY * ST_BH) + (bq + k) * ST_BH + r;
                const bool ok = x < W && y < H;
                const int64_t pix = ok ? (int64_t)y * W + x : 0;
                vx[k][r] = fl[pix]; vy[k][r] = fl[HW + pix];
            }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < ST_BH; ++r) { fldr_pin(vx[k][r]); fldr_pin(vy[k][r]); }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float xmin = INF, xmax = -INF, ymin = INF, ymax = -INF;
#pragma unroll
            for (int r = 0; r < ST_BH; ++r) {                // pixels outside the image do not count (fminf/fmaxf drop NaN)
                const int y = sby * (ST_SBY * ST_BH) + (bq + k) * ST_BH + r;
                const bool ok = x < W && y < H;
                xmin = fminf(xmin, ok ? vx[k][r] : INF); xmax = fmaxf(xmax, ok ? vx[k][r] : -INF);
                ymin = fminf(ymin, ok ? vy[k][r] : INF); ymax = fmaxf(ymax, ok ? vy[k][r] : -INF);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                xmin = fminf(xmin, __shfl_xor(xmin, o)); xmax = fmaxf(xmax, __shfl_xor(xmax, o));
                ymin = fminf(ymin, __shfl_xor(ymin, o)); ymax = fmaxf(ymax, __shfl_xor(ymax, o));
            }
            if (lane == 0) {
                float4 v = make_float4(xmin, xmax, ymin, ymax);
                *reinterpret_cast<float4*>(bo + ((bq + k) * ST_SBX + wv) * 4) = v;
            }
            sxmin = fminf(sxmin, xmin); sxmax = fmaxf(sxmax, xmax);
            symin = fminf(symin, ymin); symax = fmaxf(symax, ymax);
        }
    }
    if (lane == 0) { red[wv][0] = sxmin; red[wv][1] = sxmax; red[wv][2] = symin; red[wv][3] = symax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float4 v = make_float4(fminf(fminf(red[0][0], red[1][0]), fminf(red[2][0], red[3][0])),
                               fmaxf(fmaxf(red[0][1], red[1][1]), fmaxf(red[2][1], red[3][1])),
                               fminf(fminf(red[0][2], red[1][2]), fminf(red[2][2], red[3][2])),
                               fmaxf(fmaxf(red[0][3], red[1][3]), fmaxf(red[2][3], red[3][3])));
        *reinterpret_cast<float4*>(sbt + ((int64_t)n * nsb + sb) * 4) = v;
    }
}

// Can a source region [rx0, rx1] x [ry0, ry1] (inclusive pixel coordinates) with flow bounds b touch the tile?
// Target corner columns of a source: floor(x + fx) and floor(x + fx) + 1.  Conservative by one cell.
__device__ __forceinline__ bool st_match(const float4 b, float rx0, float rx1, float ry0, float ry1, float tx0, float tx1,
                                         float ty0, float ty1) {
    return (rx1 + b.y >= tx0 - 2.0f) && (rx0 + b.x <= tx1 + 1.0f) && (ry1 + b.w >= ty0 - 2.0f) && (ry0 + b.z <= ty1 + 1.0f);
}

// mode: 0 summation; 1 average; 2 linear; 3 softmax.  CB value channels per workgroup (+ 1 normalisation accumulator
// when MODE >= 1); channel group = blockIdx.z % groups.
template <int MODE, int CB, int TW, int TH>
__global__ __launch_bounds__(256) void splat_tile_kernel(const float* __restrict__ in, const float* __restrict__ flow,
                                                         const float* __restrict__ metric, const float* __restrict__ blk,
                                                         const float* __restrict__ sbt, float* __restrict__ out,
                                                         int C, int H, int W, int groups, int nsb_x, int nsb) {
#pragma clang fp contract(off)
    constexpr int CA = MODE >= 1 ? CB + 1 : CB;
    constexpr int CELLS = TW * TH;
    __shared__ float acc[CA * CELLS];
    __shared__ unsigned short sbq[ST_SBQ];
    __shared__ int blkq[ST_Q];
    __shared__ int q_count[2];                                  // [0] super-blocks, [1] blocks
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = blockIdx.z / groups, grp = blockIdx.z % groups;
    const int cbase = grp * CB;
    const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
    const int64_t HW = (int64_t)H * W;
    const float ftx0 = (float)tx0, ftx1 = (float)(tx0 + TW - 1), fty0 = (float)ty0, fty1 = (float)(ty0 + TH - 1);

    for (int i = tid; i < CA * CELLS / 4; i += 256) reinterpret_cast<float4*>(acc)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (tid < 2) q_count[tid] = 0;
    __syncthreads();

    // ---- which super-blocks, then which blocks, can reach this tile ----
    const float* sbn = sbt + (int64_t)n * nsb * 4;
    const float* bkn = blk + (int64_t)n * nsb * ST_SB_BLOCKS * 4;
    for (int s = tid; s < nsb; s += 256) {
        const float4 b = *reinterpret_cast<const float4*>(sbn + s * 4);
        const int sx = (s % nsb_x) * (ST_SBX * ST_BW), sy = (s / nsb_x) * (ST_SBY * ST_BH);
        if (st_match(b, (float)sx, (float)(sx + ST_SBX * ST_BW - 1), (float)sy, (float)(sy + ST_SBY * ST_BH - 1), ftx0, ftx1, fty0, fty1)) {
            const int i = atomicAdd(&q_count[0], 1);
            if (i < ST_SBQ) sbq[i] = (unsigned short)s;
        }
    }
    __syncthreads();
    const int n_sb = q_count[0];
    bool overflow = n_sb > ST_SBQ;
    if (!overflow) {
        for (int i = wv; i < n_sb; i += 4) {                    // one matching super-block per wave, one block per lane
            const int s = sbq[i];
            const float4 b = *reinterpret_cast<const float4*>(bkn + ((int64_t)s * ST_SB_BLOCKS + lane) * 4);
            const int bx = (s % nsb_x) * ST_SBX + (lane % ST_SBX), by = (s / nsb_x) * ST_SBY + (lane / ST_SBX);
            const int sx = bx * ST_BW, sy = by * ST_BH;
            if (st_match(b, (float)sx, (float)(sx + ST_BW - 1), (float)sy, (float)(sy + ST_BH - 1), ftx0, ftx1, fty0, fty1)) {
                const int k = atomicAdd(&q_count[1], 1);
                if (k < ST_Q) blkq[k] = (by << 16) | bx;
            }
        }
        __syncthreads();
        overflow = q_count[1] > ST_Q;
    }
    const int nbx = nsb_x * ST_SBX;
    const int nby = (nsb / nsb_x) * ST_SBY;
    const int n_blk = overflow ? nbx * nby : q_count[1];

    const float* fl = flow + (int64_t)n * 2 * HW;
    const float* mt = metric ? metric + (int64_t)n * HW : nullptr;
    const float* inn = in + (int64_t)n * C * HW;

    // ---- accumulate: thread = one source pixel of a block (wave = one 64-pixel row), ST_U blocks in flight ----
    for (int q0 = 0; q0 < n_blk; q0 += ST_U) {
        float fx[ST_U], fy[ST_U], mv[ST_U], val[ST_U][CB];
        int px[ST_U], py[ST_U];
        bool ok[ST_U];
#pragma unroll
        for (int u = 0; u < ST_U; ++u) {
            const int q = q0 + u;
            int bx, by;
            if (!overflow) { const int e = blkq[q < n_blk ? q : 0]; bx = e & 0xFFFF; by = e >> 16; }
            else { const int qq = q < n_blk ? q : 0; bx = qq % nbx; by = qq / nbx; }
            px[u] = bx * ST_BW + lane; py[u] = by * ST_BH + wv;
            ok[u] = q < n_blk && px[u] < W && py[u] < H;
            const int64_t pix = ok[u] ? (int64_t)py[u] * W + px[u] : 0;
            fx[u] = fl[pix]; fy[u] = fl[HW + pix];
            mv[u] = 0.0f;
            if ((MODE == 2 || MODE == 3) && mt != nullptr) mv[u] = mt[pix];
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                const int cc = cbase + c < C ? cbase + c : C - 1;
                val[u][c] = inn[(int64_t)cc * HW + pix];
            }
        }
#pragma unroll
        for (int u = 0; u < ST_U; ++u) {
            fldr_pin(fx[u]); fldr_pin(fy[u]); fldr_pin(mv[u]);
#pragma unroll
            for (int c = 0; c < CB; ++c) fldr_pin(val[u][c]);
        }
#pragma unroll
        for (int u = 0; u < ST_U; ++u) {
            if (!ok[u]) continue;
            const StGeom g = st_geom(px[u], py[u], fx[u], fy[u], W, H);
            const int lx = g.x0 - tx0, ly = g.y0 - ty0;                    // tile-local north-west corner
            if (lx < -1 || lx >= TW || ly < -1 || ly >= TH) continue;     // footprint misses the tile
            // a corner counts here when it lies in this tile AND in the image (edge tiles extend past it; tile origins
            // are >= 0, so "in the tile" implies a non-negative coordinate)
            const bool cx0 = lx >= 0 && g.x0 < W, cx1 = lx + 1 < TW && g.x0 + 1 < W;
            const bool cy0 = ly >= 0 && g.y0 < H, cy1 = ly + 1 < TH && g.y0 + 1 < H;
            const bool vnw = cx0 && cy0, vne = cx1 && cy0, vsw = cx0 && cy1, vse = cx1 && cy1;
            float wgt = 1.0f;
            if (MODE == 2) wgt = mv[u];
            if (MODE == 3 && mt != nullptr) wgt = expf(mv[u]);
            float* cell = acc + ly * TW + lx;
#pragma unroll
            for (int c = 0; c < CA; ++c) {
                float v;
                if (c < CB) {
                    v = val[u][c];
                    if (MODE == 3) v = (v + 1.0f) / 2.0f;                  // softSplat.py:334
                    if (MODE >= 2) v = v * wgt;                            // :328 / :338
                    if (cbase + c >= C) v = 0.0f;
                } else {
                    v = wgt;                                               // normalisation accumulator
                }
                float* pc = cell + c * CELLS;
                if (vnw) atomicAdd(pc, v * g.wnw);
                if (vne) atomicAdd(pc + 1, v * g.wne);
                if (vsw) atomicAdd(pc + TW, v * g.wsw);
                if (vse) atomicAdd(pc + TW + 1, v * g.wse);
            }
        }
    }
    __syncthreads();

    // ---- finish and write the tile: (acc / norm - 0.5) * 2, norm 0 -> 1 (softSplat.py:343-349) ----
    float* on = out + (int64_t)n * C * HW;
    for (int i = tid; i < CELLS; i += 256) {
        const int x = tx0 + i % TW, y = ty0 + i / TW;
        if (x >= W || y >= H) continue;
        float norm = 1.0f;
        if (MODE >= 1) { norm = acc[CB * CELLS + i]; if (norm == 0.0f) norm = 1.0f; }
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            if (cbase + c >= C) break;
            float v = acc[c * CELLS + i];
            if (MODE >= 1) v = v / norm;
            on[(int64_t)(cbase + c) * HW + (int64_t)y * W + x] = (v - 0.5f) * 2.0f;
        }
    }
}

extern "C" int64_t fldr_softsplat_tile_ws_floats(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return FLDR_E_ARG;
    const int64_t nsb = (int64_t)fldr_cdiv(W, ST_SBX * ST_BW) * fldr_cdiv(H, ST_SBY * ST_BH);
    return (int64_t)N * nsb * (ST_SB_BLOCKS + 1) * 4;
}

template <int MODE>
static void splat_tile_launch(const float* img, const float* flow, const float* metric, const float* blk, const float* sbt,
                              float* out, int N, int C, int H, int W, int nsb_x, int nsb, hipStream_t s) {
    if (C <= 3) {                  // images: every channel + the normalisation accumulator in one 128 x 32 tile (64 KB of LDS)
        dim3 grid(fldr_cdiv(W, 128), fldr_cdiv(H, 32), N);
        hipLaunchKernelGGL((splat_tile_kernel<MODE, 3, 128, 32>), grid, dim3(256), 0, s, img, flow, metric, blk, sbt, out, C, H, W, 1, nsb_x, nsb);
    } else {                       // feature maps: groups of 12 channels, 64 x 16 tiles (52 KB)
        const int groups = fldr_cdiv(C, 12);
        dim3 grid(fldr_cdiv(W, 64), fldr_cdiv(H, 16), N * groups);
        hipLaunchKernelGGL((splat_tile_kernel<MODE, 12, 64, 16>), grid, dim3(256), 0, s, img, flow, metric, blk, sbt, out, C, H, W, groups, nsb_x, nsb);
    }
}

// FunctionSoftsplat (softSplat.py:320-352) end to end, destination-owned.  ws: fldr_softsplat_tile_ws_floats floats.
extern "C" int fldr_softsplat_tile(const float* img, const float* flow, const float* metric, float* out, float* ws,
                                   int N, int C, int H, int W, int mode, fldr_stream_t stream) {
    FLDR_CHECK_ARG(img && flow && out && ws && N > 0 && C > 0 && H > 0 && W > 0 && mode >= 0 && mode <= 3);
    FLDR_CHECK_ARG(mode != 2 || metric != nullptr);
    if (W > 65535 * ST_BW || H > 32767 * ST_BH) return FLDR_E_SHAPE;
    const int nsb_x = fldr_cdiv(W, ST_SBX * ST_BW), nsb = nsb_x * fldr_cdiv(H, ST_SBY * ST_BH);
    float* blk = ws;
    float* sbt = ws + (int64_t)N * nsb * ST_SB_BLOCKS * 4;
    hipStream_t s = fldr_s(stream);
    hipLaunchKernelGGL(splat_bounds_kernel, dim3(nsb, N), dim3(256), 0, s, flow, blk, sbt, H, W, nsb_x, nsb);
    switch (mode) {
        case 0: splat_tile_launch<0>(img, flow, metric, blk, sbt, out, N, C, H, W, nsb_x, nsb, s); break;
        case 1: splat_tile_launch<1>(img, flow, metric, blk, sbt, out, N, C, H, W, nsb_x, nsb, s); break;
        case 2: splat_tile_launch<2>(img, flow, metric, blk, sbt, out, N, C, H, W, nsb_x, nsb, s); break;
        default: splat_tile_launch<3>(img, flow, metric, blk, sbt, out, N, C, H, W, nsb_x, nsb, s); break;
    }
    FLDR_LAUNCH_RET();
}
