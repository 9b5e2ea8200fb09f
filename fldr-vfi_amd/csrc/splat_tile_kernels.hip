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
#include "splat_common.h"

// ------------------------------------------------------------------------------------------------
// pre-pass: flow bounds per block and per super-block
//   blk[n][sb][block in sb][4] = {fxmin, fxmax, fymin, fymax};  sbt[n][sb][4]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void splat_bounds_kernel(const float* __restrict__ flow, float* __restrict__ blk,
                                                           float* __restrict__ sbt, int H, int W, int nsb_x, int nsb,
                                                           int64_t flow_bstride) {
    const int sb = blockIdx.x, n = blockIdx.y;
    const int sbx = sb % nsb_x, sby = sb / nsb_x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = sbx * (ST_SBX * ST_BW) + threadIdx.x;
    const int64_t HW = (int64_t)H * W;
    const float* fl = flow + (int64_t)n * flow_bstride;          // samples flow_bstride floats apart ([2,H,W] each)
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

// (also used by splat_acc64_kernels.hip)
void fldr_splat_bounds_launch(const float* flow, int64_t flow_bstride, float* blk, float* sbt, int N, int H, int W, int nsb_x, int nsb,
                              hipStream_t s) {
    hipLaunchKernelGGL(splat_bounds_kernel, dim3(nsb, N), dim3(256), 0, s, flow, blk, sbt, H, W, nsb_x, nsb, flow_bstride);
}

// The same table for a flow that is the bilinear upsampling of a low-resolution field — flow = up(scale * lo) * mul, what
// fldr_level0_prep writes as flow_t0 / flow_t1 (fLDRnet.py:404-405,419-422) — computed from the low-resolution field alone:
// bilinear weights are a convex combination, so every value of a 64x4 block lies between the extremes of the low-resolution
// pixels its rows and columns interpolate between (fldr_lin_src is monotone: the footprint is the index range of the block's
// first and last pixel), widened by 2e-6 of the magnitude for the roundings of the three interpolation steps.  The bounds
// only select candidate sources (st_match / the trimmed walk): a superset is exact.  2 MB read instead of the two
// full-resolution flow planes.
// pair (0: off): blockIdx.y runs over the 2 N flows of BOTH problems of a pair call, lo = the [N,4,h,w] level flow (channels
// 0-1 flow_10, 2-3 flow_01), table sample k N + n = problem k, sample n.  pair 1 = the level-0 image splats (fLDRnet.py:
// 404-405, 449-450: problem 0 = t * flow_01, problem 1 = (1 - t) * flow_10), pair 2 = the feature splats of a level (:386-387:
// problem 0 = flow_10 — feat1's flow —, problem 1 = flow_01, unscaled).
// One workgroup of 16 waves per super-block: wave v takes blocks v, v + 16, v + 32, v + 48, lanes run over the columns of a
// block's low-resolution footprint (all loads of a wave's four blocks are independent), wave shuffles reduce, LDS joins the
// 64 block bounds into the super-block's.  (The first version ran one LANE per block over its footprint — 99 dependent
// iterations for a x2 upsampling: 12-22 us per launch whatever the size, rocprofv3 round 3; this one is ~3 us.)
__global__ __launch_bounds__(64 * ST_UP_WAVES) void splat_bounds_up_kernel(const float* __restrict__ lo, int64_t lo_bstride, const float* __restrict__ tv,
                                                             int smode, float mul, float* __restrict__ blk, float* __restrict__ sbt,
                                                             int h, int w, int H, int W, float sy, float sx, int nsb_x, int nsb,
                                                             int pair, int N) {
    __shared__ float red[ST_UP_WAVES][4];
    splat_bounds_up_body(lo, lo_bstride, tv, smode, mul, blk, sbt, h, w, H, W, sy, sx, nsb_x, nsb, pair, N, blockIdx.x, blockIdx.y, red);
}

// The same table with ONE LANE per block and one wave per super-block (no LDS, no barrier): the better shape when the
// upsampling factor is large — a x8 block's footprint is 10 x 2 low-resolution pixels, a 20-iteration loop per lane: 7.7 us for
// the level-0 pair of a 4K frame, where the workgroup-per-super-block kernel above needs 20-45 us (one 1024-thread workgroup
// per CU at a time, ~10 us each whatever the footprint).  For x2 (99 iterations per lane: 12-22 us) the kernel above wins.
__global__ __launch_bounds__(64) void splat_bounds_up_lane_kernel(const float* __restrict__ lo, int64_t lo_bstride, const float* __restrict__ tv,
                                                                  int smode, float mul, float* __restrict__ blk, float* __restrict__ sbt,
                                                                  int h, int w, int H, int W, float sy, float sx, int nsb_x, int nsb,
                                                                  int pair, int N) {
#pragma clang fp contract(off)
    const int sb = blockIdx.x, lane = threadIdx.x;
    int n = blockIdx.y;
    const int tab = n;                                                  // table sample
    if (pair) {
        const int k = n / N;
        n -= k * N;
        const bool second_half = (pair == 1) == (k == 0);             // channels 2-3 (flow_01)
        lo += second_half ? 2 * (int64_t)h * w : 0;
        smode = pair == 1 ? (k == 0 ? 1 : 2) : 0;
    }
    const int X0 = ((sb % nsb_x) * ST_SBX + lane % ST_SBX) * ST_BW, Y0 = ((sb / nsb_x) * ST_SBY + lane / ST_SBX) * ST_BH;
    const float INF = __builtin_inff();
    float xmin = INF, xmax = -INF, ymin = INF, ymax = -INF;
    if (X0 < W && Y0 < H) {
        const float scale = smode == 0 ? 1.0f : (smode == 1 ? tv[n] : 1.0f - tv[n]);
        int c0, c1, r0, r1, d0, d1; float l;
        fldr_lin_src(X0, sx, w, c0, d0, l); fldr_lin_src(min(X0 + ST_BW - 1, W - 1), sx, w, d1, c1, l);
        fldr_lin_src(Y0, sy, h, r0, d0, l); fldr_lin_src(min(Y0 + ST_BH - 1, H - 1), sy, h, d1, r1, l);
        const float* px = lo + (int64_t)n * lo_bstride;
        const float* py = px + (int64_t)h * w;
        for (int r = r0; r <= r1; ++r)
            for (int c = c0; c <= c1; ++c) {
                const float vx = (scale * px[(int64_t)r * w + c]) * mul, vy = (scale * py[(int64_t)r * w + c]) * mul;
                xmin = fminf(xmin, vx); xmax = fmaxf(xmax, vx); ymin = fminf(ymin, vy); ymax = fmaxf(ymax, vy);
            }
        const float ex = fmaxf(fabsf(xmin), fabsf(xmax)) * 2.0e-6f, ey = fmaxf(fabsf(ymin), fabsf(ymax)) * 2.0e-6f;
        xmin -= ex; xmax += ex; ymin -= ey; ymax += ey;
    }
    *reinterpret_cast<float4*>(blk + (((int64_t)tab * nsb + sb) * ST_SB_BLOCKS + lane) * 4) = make_float4(xmin, xmax, ymin, ymax);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        xmin = fminf(xmin, __shfl_xor(xmin, o)); xmax = fmaxf(xmax, __shfl_xor(xmax, o));
        ymin = fminf(ymin, __shfl_xor(ymin, o)); ymax = fmaxf(ymax, __shfl_xor(ymax, o));
    }
    if (lane == 0) *reinterpret_cast<float4*>(sbt + ((int64_t)tab * nsb + sb) * 4) = make_float4(xmin, xmax, ymin, ymax);
}

// which of the two: the lane-per-block kernel from x4 upwards
static void splat_bounds_up_launch(const float* lo, int64_t lo_bstride, const float* t, int smode, float mul, float* blk, float* sbt,
                                   int h, int w, int H, int W, int nsb_x, int nsb, int pair, int N, int samples, hipStream_t s) {
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    if ((int64_t)W >= 4 * (int64_t)w)
        hipLaunchKernelGGL(splat_bounds_up_lane_kernel, dim3(nsb, samples), dim3(64), 0, s, lo, lo_bstride, t, smode, mul, blk, sbt, h, w, H, W, sy, sx, nsb_x, nsb, pair, N);
    else
        hipLaunchKernelGGL(splat_bounds_up_kernel, dim3(nsb, samples), dim3(64 * ST_UP_WAVES), 0, s, lo, lo_bstride, t, smode, mul, blk, sbt, h, w, H, W, sy, sx, nsb_x, nsb, pair, N);
}

#ifdef FLDR_TEST_HOOKS        // the LDS-f32-atomic tile kernel (ds_add_f32: 80 ns per wave-instruction): measured and retired, test build only
// mode: 0 summation; 1 average; 2 linear; 3 softmax.  CB value channels per workgroup (+ 1 normalisation accumulator
// when MODE >= 1); channel group = blockIdx.z % groups.
template <int MODE, int CB, int TW, int TH>
__global__ __launch_bounds__(256) void splat_tile_kernel(const float* __restrict__ in, int64_t in_bstride, int64_t in_cstride,
                                                         const float* __restrict__ flow,
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
    const float* inn = in + (int64_t)n * in_bstride;              // channel planes in_cstride floats apart (views of [B,3,2,H,W])

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
                val[u][c] = inn[(int64_t)cc * in_cstride + pix];
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
#if defined(BAND_NT) && (BAND_NT & 1)
            __builtin_nontemporal_store((v - 0.5f) * 2.0f, &on[(int64_t)(cbase + c) * HW + (int64_t)y * W + x]);
#else
            on[(int64_t)(cbase + c) * HW + (int64_t)y * W + x] = (v - 0.5f) * 2.0f;
#endif
        }
    }
}

// (the test-build guard continues through the band kernel below)

// ------------------------------------------------------------------------------------------------
// Band splat: destination-owned, NO atomics at all.
//
// ds_add_f32 retires one wave-instruction per ~190 cycles per CU on gfx950 (tools/ubench/lds_atomic_bench.hip; the
// integer ds_add_u32 takes 4), so the tile kernel above is slower than the global-atomic strip kernel.  Plain LDS
// read-modify-write is fast, but only legal when no two lanes of a wave-instruction hit the same cell and no other
// wave touches it.  Hence:
//   * every WAVE owns a private TW x TH band of the output in LDS (no other wave ever touches it);
//   * per source row (64 pixels, one per lane) and bilinear corner, the lanes CLAIM their cell (ds_write of the lane id
//     to a claim array, read back: for lanes that collide exactly one reads its own id); owners add their contribution
//     with ds_read / v_add / ds_write, the others retry in the next round.  Coherent flows need one round; any flow
//     terminates (each contested cell gets one owner per round) and the sum is exact.
// The candidate search (flow bounds per block / super-block) is the tile kernel's, done per wave with ballots.
// ------------------------------------------------------------------------------------------------

#define SB_LIST 128
#ifndef ST_IMG_TW
#define ST_IMG_TW 56            // band of the image instance: 56 + 3 + (flow spread <= 5) columns of candidates = ONE 64-pixel chunk per
#define ST_IMG_TH 12            // row in the trimmed walk; measured in the 4K forward: 128x6 366 us, 120x6 336 us, 56x12 317 us
#endif

#ifdef ST_STAMPS
__device__ unsigned long long st_stamp_buf[16];
#define TSTAMP(var) unsigned long long var; { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
FLDR_HOOK int fldr_debug_read_st_stamps(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(st_stamp_buf), sizeof(unsigned long long) * 16); }
#else
#define TSTAMP(var)
#endif

// Wave-wide shift by one lane as a DPP move (gfx9 wave_shr:1 / wave_shl:1): a VALU cycle instead of the LDS round trip of
// ds_bpermute (__shfl_up / __shfl_down).  Lane 0 (shr) / lane 63 (shl) keep their own value.  All lanes must be active.
__device__ __forceinline__ int st_shr1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int st_shl1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ float st_shr1(float v) { return __int_as_float(st_shr1(__float_as_int(v))); }

template <int MODE, int CB, int TW, int TH>
__global__ __launch_bounds__(256) void splat_band_kernel(const float* __restrict__ in, int64_t in_bstride, int64_t in_cstride,
                                                         const float* __restrict__ flow,
                                                         const float* __restrict__ metric, const float* __restrict__ blk,
                                                         const float* __restrict__ sbt, float* __restrict__ out,
                                                         int C, int H, int W, int groups, int nsb_x, int nsb) {
#pragma clang fp contract(off)
    constexpr int CA = MODE >= 1 ? CB + 1 : CB;
    constexpr int CELLS = TW * TH;
    __shared__ __attribute__((aligned(16))) float acc_all[4][CA * CELLS];
    __shared__ unsigned char claim_all[4][CELLS];                 // lane ids (< 64)
    __shared__ unsigned short sbl_all[4][SB_LIST];                // matching super-blocks of each wave (nsb < 65536: host-checked)
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.z / groups, grp = blockIdx.z % groups;
    const int cbase = grp * CB;
    const int tx0 = blockIdx.x * TW, ty0 = (blockIdx.y * 4 + wv) * TH;
    if (ty0 >= H) return;                                         // wave-uniform; the kernel has no workgroup barrier
    float* acc = acc_all[wv];
    // accumulator of (channel c, cell): cell-major when a cell's CA values are one 16-byte record (the image splat: 3 channels
    // + the normalisation sum), so that a read-modify-write of a cell is ONE ds_read_b128 + ONE ds_write_b128 instead of
    // four of each; channel-major otherwise
    auto ai = [](int c, int cell) __attribute__((always_inline)) { return CA == 4 ? cell * 4 + c : c * CELLS + cell; };
    volatile unsigned char* claim = claim_all[wv];                // volatile: the write / read-back pair must reach the LDS
    unsigned short* sbl = sbl_all[wv];
    const int64_t HW = (int64_t)H * W;
    const float ftx0 = (float)tx0, ftx1 = (float)(tx0 + TW - 1), fty0 = (float)ty0, fty1 = (float)(ty0 + TH - 1);

    TSTAMP(t_begin)
    for (int i = lane; i < CA * CELLS / 4; i += 64) reinterpret_cast<float4*>(acc)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    TSTAMP(t_init)
#ifdef ST_STAMPS
    unsigned long long c_proc = 0, c_blocks = 0, c_rows = 0, c_fast = 0, c_slow = 0, c_walk = 0;
#endif

    const float* sbn = sbt + (int64_t)n * nsb * 4;
    const float* bkn = blk + (int64_t)n * nsb * ST_SB_BLOCKS * 4;

    const float* fl = flow + (int64_t)n * 2 * HW;
    const float* mt = metric ? metric + (int64_t)n * HW : nullptr;
    const float* inn = in + (int64_t)n * in_bstride;              // channel planes in_cstride floats apart (views of [B,3,2,H,W])

    // one corner of one source row: claim, then read-modify-write by the owners; repeat for the lanes that lost
    auto scatter = [&](bool active, int cell, float w, const float (&v)[CA]) __attribute__((always_inline)) {
        bool pending = active;
        while (__ballot(pending)) {
            if (pending) claim[cell] = (unsigned char)lane;
            bool owner = false;
            if (pending) owner = claim[cell] == (unsigned char)lane;
            if (owner) {
                float cur[CA];
#pragma unroll
                for (int c = 0; c < CA; ++c) cur[c] = acc[ai(c, cell)];
#pragma unroll
                for (int c = 0; c < CA; ++c) acc[ai(c, cell)] = cur[c] + v[c] * w;
            }
            pending = pending && !owner;
        }
    };

    // ---- accumulate: lane = one source pixel of a 64-pixel row; the next block's 4 rows are loaded ahead ----
    float fx[2][ST_BH], fy[2][ST_BH], mv[2][ST_BH], val[2][ST_BH][CB];
    int bxy[2] = {0, 0};
    auto load_block = [&](int buf, int e) __attribute__((always_inline)) {   // e = (first row << 16) | first column: 4 rows x 64 pixels
        bxy[buf] = e;
        const int px = (e & 0xFFFF) + lane;
#pragma unroll
        for (int r = 0; r < ST_BH; ++r) {
            const int py = (e >> 16) + r;
            const bool ok = px < W && py < H;
            const int64_t pix = ok ? (int64_t)py * W + px : 0;
#if defined(BAND_NT) && (BAND_NT & 2)
            fx[buf][r] = __builtin_nontemporal_load(&fl[pix]); fy[buf][r] = __builtin_nontemporal_load(&fl[HW + pix]);
            mv[buf][r] = 0.0f;
            if ((MODE == 2 || MODE == 3) && mt != nullptr) mv[buf][r] = __builtin_nontemporal_load(&mt[pix]);
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                const int cc = cbase + c < C ? cbase + c : C - 1;
                val[buf][r][c] = __builtin_nontemporal_load(&inn[(int64_t)cc * in_cstride + pix]);
            }
#elif defined(BAND_ABLATE) && BAND_ABLATE == 1                         // diagnostic: only the flow is read
            fx[buf][r] = fl[pix]; fy[buf][r] = fl[HW + pix];
            mv[buf][r] = 0.25f;
#pragma unroll
            for (int c = 0; c < CB; ++c) val[buf][r][c] = 0.5f;
#else
            fx[buf][r] = fl[pix]; fy[buf][r] = fl[HW + pix];
            mv[buf][r] = 0.0f;
            if ((MODE == 2 || MODE == 3) && mt != nullptr) mv[buf][r] = mt[pix];
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                const int cc = cbase + c < C ? cbase + c : C - 1;
                val[buf][r][c] = inn[(int64_t)cc * in_cstride + pix];
            }
#endif
        }
    };
    // plain read-modify-write of one cell per active lane: ONLY for sets of lanes whose cells are pairwise distinct
    auto rmw = [&](bool active, int cell, const float (&v)[CA]) __attribute__((always_inline)) {
        if (!__ballot(active)) return;                            // wave-uniform
        if (active) {
            float cur[CA];
#pragma unroll
            for (int c = 0; c < CA; ++c) cur[c] = acc[ai(c, cell)];
#pragma unroll
            for (int c = 0; c < CA; ++c) acc[ai(c, cell)] = cur[c] + v[c];
        }
    };

    // One block = 4 consecutive source rows of 64 pixels.  Fast path per row (the strip kernel's register merging, with
    // LDS read-modify-write instead of atomics), valid when the target columns x0 are non-decreasing over the lanes
    // with no run of three equal values — any locally smooth flow:
    //   * a lane whose right neighbour lands in the SAME cell (x0, y0 equal: compression) hands it all four corners;
    //   * a lane whose right neighbour lands one cell to the right hands it its right-hand column (NE, SE), which is
    //     the neighbour's left-hand column;
    //   * the bottom-left cell stays pending in registers and joins the next row's top-left cell when that lands one
    //     row lower (the normal case), so a row costs ONE read-modify-write per channel.
    // Under the precondition the emitting lanes of each set address pairwise distinct cells (x0 strictly increases over
    // them), so no claim round is needed.  Any other row takes the claim path above (exact for every flow).
    auto process_block = [&](int buf) __attribute__((always_inline)) {
        const int e = bxy[buf];
        const int px = (e & 0xFFFF) + lane;
        float pend[CA];
        int pend_cell = -1;
#pragma unroll
        for (int c = 0; c < CA; ++c) pend[c] = 0.0f;
#pragma unroll
        for (int r = 0; r < ST_BH; ++r) {
            const int py = (e >> 16) + r;
            const bool ok = px < W && py < H;
            const StGeom g = st_geom(px, py, fx[buf][r], fy[buf][r], W, H);
            const int lx = g.x0 - tx0, ly = g.y0 - ty0;          // band-local north-west corner
            const bool hit = ok && lx >= -1 && lx < TW && ly >= -1 && ly < TH;
            // a corner counts here when it lies in this band AND in the image (band origins are >= 0)
            const bool cx0 = hit && lx >= 0 && g.x0 < W, cx1 = hit && lx + 1 < TW && g.x0 + 1 < W;
            const bool cy0 = ly >= 0 && g.y0 < H, cy1 = ly + 1 < TH && g.y0 + 1 < H;
            const bool vnw = cx0 && cy0, vne = cx1 && cy0, vsw = cx0 && cy1, vse = cx1 && cy1;
            if (!__ballot(vnw || vne || vsw || vse)) {            // wave-uniform: this row misses the band
                rmw(pend_cell >= 0, pend_cell, pend);
                pend_cell = -1;
                continue;
            }
#ifdef ST_STAMPS
            ++c_rows;
#endif
            float wgt = 1.0f;
            if (MODE == 2) wgt = mv[buf][r];
            if (MODE == 3 && mt != nullptr) wgt = expf(mv[buf][r]);
            float v[CA];
#pragma unroll
            for (int c = 0; c < CA; ++c) {
                if (c < CB) {
                    float t = val[buf][r][c];
                    if (MODE == 3) t = (t + 1.0f) / 2.0f;         // softSplat.py:334
                    if (MODE >= 2) t = t * wgt;                   // :328 / :338
                    if (cbase + c >= C) t = 0.0f;
                    v[c] = t;
                } else {
                    v[c] = wgt;                                   // normalisation accumulator
                }
            }
            const int cell = ly * TW + lx;
            // precondition of the fast path (lanes past the right image border count as far to the right)
            const int xk = ok ? g.x0 : 0x3fffffff;
            const int xl = st_shr1(xk), xll = st_shr1(xl);       // (lane 1's xll is lane 0's value: the test below needs lane >= 2)
            const bool bad = (lane >= 1 && xk < xl) || (lane >= 2 && xk == xl && xl == xll && xk != 0x3fffffff);
#ifdef ST_STAMPS
            if (__ballot(bad)) ++c_slow; else ++c_fast;
#endif
            if (__ballot(bad)) {                                  // wave-uniform: claim path for this row
                scatter(pend_cell >= 0, pend_cell, 1.0f, pend);
                pend_cell = -1;
                scatter(vnw, cell, g.wnw, v);
                scatter(vne, cell + 1, g.wne, v);
                scatter(vsw, cell + TW, g.wsw, v);
                scatter(vse, cell + TW + 1, g.wse, v);
                continue;
            }
            // (every lane shift below is executed by ALL lanes)
            const int xr = st_shl1(g.x0), yr = st_shl1(g.y0);
            const int hr_i = st_shl1((int)hit);
            const bool hr = hr_i != 0 && lane < 63;
            const bool giver = hit && hr && xr == g.x0 && yr == g.y0;             // right neighbour: same cell
            const bool mright = hit && hr && xr == g.x0 + 1 && yr == g.y0;        // right neighbour: one cell to the right
            const int lg_i = st_shr1((int)giver), lm_i = st_shr1((int)mright);
            const bool lgiver = lane > 0 && lg_i != 0;
            const bool lmright = lane > 0 && lm_i != 0;
            float top[CA], bot[CA], rtop[CA], rbot[CA];
#pragma unroll
            for (int c = 0; c < CA; ++c) {
                float nw = vnw ? v[c] * g.wnw : 0.0f, ne = vne ? v[c] * g.wne : 0.0f;
                float sw = vsw ? v[c] * g.wsw : 0.0f, se = vse ? v[c] * g.wse : 0.0f;
                // (1) same-cell neighbour: right-hand column first ...
                const float dne = st_shr1(giver ? ne : 0.0f), dse = st_shr1(giver ? se : 0.0f);
                if (giver) { ne = 0.0f; se = 0.0f; }
                if (lgiver) { ne += dne; se += dse; }
                // (2) ... then right-hand columns move into the neighbour's left-hand column ...
                const float ane = st_shr1(mright ? ne : 0.0f), ase = st_shr1(mright ? se : 0.0f);
                if (mright) { ne = 0.0f; se = 0.0f; }
                if (lmright) { nw += ane; sw += ase; }
                // (3) ... then the left-hand column of a same-cell giver (with what it received in (2))
                const float dnw = st_shr1(giver ? nw : 0.0f), dsw = st_shr1(giver ? sw : 0.0f);
                if (giver) { nw = 0.0f; sw = 0.0f; }
                if (lgiver) { nw += dnw; sw += dsw; }
                top[c] = nw; bot[c] = sw; rtop[c] = ne; rbot[c] = se;
            }
            const bool emit = hit && !giver;
            // pending bottom cells of the previous row: join this row's top-left cell when aligned, else go out on their own
            const bool aligned = emit && vnw && pend_cell == cell;
            rmw(pend_cell >= 0 && !aligned, pend_cell, pend);
            if (aligned) {
#pragma unroll
                for (int c = 0; c < CA; ++c) top[c] += pend[c];
            }
            rmw(emit && vnw, cell, top);
            rmw(emit && vne && !mright, cell + 1, rtop);
            rmw(emit && vse && !mright, cell + TW + 1, rbot);
            pend_cell = (emit && vsw) ? cell + TW : -1;
#pragma unroll
            for (int c = 0; c < CA; ++c) pend[c] = bot[c];
        }
        rmw(pend_cell >= 0, pend_cell, pend);
    };

    // ---- find the super-blocks whose flow bounds reach this band (8 x 64 tested per trip, all loads issued first), list
    //      them in LDS, then walk their blocks (the next super-block's block bounds and the next block's rows are loaded
    //      ahead).  Any number of candidates is handled: when the list is full the scan stops, the list is walked, and
    //      the scan resumes where it stopped. ----
    auto sb_blocks = [&](int sbi, const float4 bb) __attribute__((always_inline)) -> unsigned long long {
        const int sx = ((sbi % nsb_x) * ST_SBX + lane % ST_SBX) * ST_BW, sy = ((sbi / nsb_x) * ST_SBY + lane / ST_SBX) * ST_BH;
        return __ballot(st_match(bb, (float)sx, (float)(sx + ST_BW - 1), (float)sy, (float)(sy + ST_BH - 1), ftx0, ftx1, fty0, fty1));
    };
    TSTAMP(t_scan0)
    int s_start = 0;
    do {
        // scan [s_start, nsb) until the list is full
        int n_list = 0, s_resume = nsb;
        for (int s0 = s_start; s0 < nsb && s_resume == nsb; s0 += 64 * 8) {
            float4 sb8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int s = s0 + k * 64 + lane;
                sb8[k] = *reinterpret_cast<const float4*>(sbn + (s < nsb ? s : nsb - 1) * 4);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int s = s0 + k * 64 + lane;
                const int sx = (s % nsb_x) * (ST_SBX * ST_BW), sy = (s / nsb_x) * (ST_SBY * ST_BH);
                const bool m = s < nsb && st_match(sb8[k], (float)sx, (float)(sx + ST_SBX * ST_BW - 1), (float)sy,
                                                   (float)(sy + ST_SBY * ST_BH - 1), ftx0, ftx1, fty0, fty1);
                const unsigned long long mask = __ballot(m);
                if (mask && s_resume == nsb) {                    // wave-uniform
                    if (n_list + __popcll(mask) > SB_LIST) {
                        s_resume = s0 + k * 64;                   // resume with this chunk after the walk
                    } else {
                        if (m) sbl[n_list + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)s;
                        n_list += __popcll(mask);
                    }
                }
            }
        }
        // walk the list
        TSTAMP(t_w0)
#ifdef ST_STAMPS
        c_walk -= t_w0;
#endif
        // Trimmed walk (the common case: one scan pass found every candidate).  Aligned 64 x 4 source blocks overhang a band
        // on both sides — a 128-wide band shifted by the flow is hit by three of them, and every row of a hit block costs
        // ~450 VALU instructions — so the candidates are re-cut to the band: the bounding box of the matching blocks,
        // intersected with the pixel range their joint flow bounds allow (the st_match test for a single pixel), walked in
        // 64-pixel chunks that start at its left edge.  Exact: a source outside the range cannot touch the band, and
        // the chunks / row groups are disjoint.  Taken when it needs no more 64 x 4 pieces than there are matching blocks.
        bool trimmed = false;
        int it_k = 0, it_nb = 0, it_nx = 1, it_X0 = 0, it_Y0 = 0;
        if (s_start == 0 && s_resume == nsb && n_list > 0) {
            const float INF = __builtin_inff();
            float rxmin = INF, rxmax = -INF, rymin = INF, rymax = -INF;     // flow bounds of the matching blocks
            int cx0 = 0x7fffffff, cx1 = -1, cy0 = 0x7fffffff, cy1 = -1;     // their bounding box (pixels, inclusive)
            int n_match = 0;
            for (int i = 0; i < n_list; ++i) {
                const int sbi = sbl[i];
                const float4 bb = *reinterpret_cast<const float4*>(bkn + ((int64_t)sbi * ST_SB_BLOCKS + lane) * 4);
                const int sx = ((sbi % nsb_x) * ST_SBX + lane % ST_SBX) * ST_BW, sy = ((sbi / nsb_x) * ST_SBY + lane / ST_SBX) * ST_BH;
                const bool hit = st_match(bb, (float)sx, (float)(sx + ST_BW - 1), (float)sy, (float)(sy + ST_BH - 1), ftx0, ftx1, fty0, fty1);
                n_match += __popcll(__ballot(hit));
                if (hit) {
                    rxmin = fminf(rxmin, bb.x); rxmax = fmaxf(rxmax, bb.y); rymin = fminf(rymin, bb.z); rymax = fmaxf(rymax, bb.w);
                    cx0 = min(cx0, sx); cx1 = max(cx1, sx + ST_BW - 1); cy0 = min(cy0, sy); cy1 = max(cy1, sy + ST_BH - 1);
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                rxmin = fminf(rxmin, __shfl_xor(rxmin, o)); rxmax = fmaxf(rxmax, __shfl_xor(rxmax, o));
                rymin = fminf(rymin, __shfl_xor(rymin, o)); rymax = fmaxf(rymax, __shfl_xor(rymax, o));
                cx0 = min(cx0, __shfl_xor(cx0, o)); cx1 = max(cx1, __shfl_xor(cx1, o));
                cy0 = min(cy0, __shfl_xor(cy0, o)); cy1 = max(cy1, __shfl_xor(cy1, o));
            }
            if (cx1 >= 0 && fabsf(rxmin) < 1.0e6f && fabsf(rxmax) < 1.0e6f && fabsf(rymin) < 1.0e6f && fabsf(rymax) < 1.0e6f) {   // (finite, int-safe bounds; wave-uniform)
                // x + fx >= tx0 - 2 and x + fx <= tx1 + 1 for some fx in [rxmin, rxmax]  (st_match with rx0 = rx1 = x)
                const int X0 = max(cx0, (int)floorf(ftx0 - 2.0f - rxmax)), X1 = min(min(cx1, W - 1), (int)ceilf(ftx1 + 1.0f - rxmin));
                const int Y0 = max(cy0, (int)floorf(fty0 - 2.0f - rymax)), Y1 = min(min(cy1, H - 1), (int)ceilf(fty1 + 1.0f - rymin));
                const int nx = X1 >= X0 ? (X1 - X0) / ST_BW + 1 : 0, ny = Y1 >= Y0 ? (Y1 - Y0) / ST_BH + 1 : 0;
                if (nx * ny <= n_match) {                 // never more blocks than the walk below would process
                    trimmed = true;
                    it_nx = nx > 0 ? nx : 1; it_nb = nx * ny; it_X0 = X0; it_Y0 = Y0;
                }
            }
        }
        // One generator for both walks (a single pair of load / process sites keeps the kernel inside the instruction
        // cache): the chunks of the trimmed rectangle, or the matching blocks of the listed super-blocks.
        int li = 0, wbx0 = 0, wby0 = 0;
        unsigned long long mk = 0ull;
        auto next_block = [&](int& e) __attribute__((always_inline)) -> bool {
            if (trimmed) {
                if (it_k >= it_nb) return false;
                const int gy = it_k / it_nx;
                e = ((it_Y0 + gy * ST_BH) << 16) | (it_X0 + (it_k - gy * it_nx) * ST_BW);
                ++it_k;
                return true;
            }
            while (!mk) {
                if (li >= n_list) return false;
                const int sbi = sbl[li++];
                const float4 bb = *reinterpret_cast<const float4*>(bkn + ((int64_t)sbi * ST_SB_BLOCKS + lane) * 4);
                mk = sb_blocks(sbi, bb);
                wbx0 = (sbi % nsb_x) * ST_SBX; wby0 = (sbi / nsb_x) * ST_SBY;
            }
            const int bit = __builtin_ctzll(mk);
            mk &= mk - 1;
            e = (((wby0 + bit / ST_SBX) * ST_BH) << 16) | ((wbx0 + bit % ST_SBX) * ST_BW);
            return true;
        };
        {
            int e0 = 0, e1 = 0;
            bool have = next_block(e0);
            if (have) load_block(0, e0);
            while (have) {
                const bool more = next_block(e1);
                if (more) load_block(1, e1);
                { TSTAMP(p0) process_block(0); TSTAMP(p1)
#ifdef ST_STAMPS
                  c_proc += p1 - p0; ++c_blocks;
#endif
                }
                if (!more) break;
                have = next_block(e0);
                if (have) load_block(0, e0);
                { TSTAMP(p0) process_block(1); TSTAMP(p1)
#ifdef ST_STAMPS
                  c_proc += p1 - p0; ++c_blocks;
#endif
                }
            }
        }
        TSTAMP(t_w1)
#ifdef ST_STAMPS
        c_walk += t_w1;
#endif
        s_start = s_resume;
    } while (s_start < nsb);
    TSTAMP(t_scan1)

    // ---- finish and write the band: (acc / norm - 0.5) * 2, norm 0 -> 1 (softSplat.py:343-349) ----
    float* on = out + (int64_t)n * C * HW;
    for (int i = lane; i < CELLS; i += 64) {
        const int x = tx0 + i % TW, y = ty0 + i / TW;
        if (x >= W || y >= H) continue;
        float norm = 1.0f;
        if (MODE >= 1) { norm = acc[ai(CB, i)]; if (norm == 0.0f) norm = 1.0f; }
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            if (cbase + c >= C) break;
            float v = acc[ai(c, i)];
            if (MODE >= 1) v = v / norm;
#if defined(BAND_NT) && (BAND_NT & 1)
            __builtin_nontemporal_store((v - 0.5f) * 2.0f, &on[(int64_t)(cbase + c) * HW + (int64_t)y * W + x]);
#else
            on[(int64_t)(cbase + c) * HW + (int64_t)y * W + x] = (v - 0.5f) * 2.0f;
#endif
        }
    }
#ifdef ST_STAMPS
    TSTAMP(t_end)
    if (blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2 && wv == 1 && lane == 0) {
        unsigned long long* o = st_stamp_buf;
        o[0] = t_init - t_begin; o[1] = t_scan1 - t_scan0; o[2] = c_walk; o[3] = c_proc; o[4] = c_blocks; o[5] = c_rows;
        o[6] = c_fast; o[7] = c_slow; o[8] = t_end - t_scan1; o[9] = t_end - t_begin;
    }
#endif
}

template <int MODE>
static void splat_band_launch(const float* img, int64_t ibs, int64_t ics, const float* flow, const float* metric, const float* blk, const float* sbt,
                              float* out, int N, int C, int H, int W, int nsb_x, int nsb, hipStream_t s) {
    if (C <= 3) {                  // images: 3 channels + normalisation, 56 x 12 band per wave (11.4 KB of LDS per wave: 3 workgroups per CU)
        dim3 grid(fldr_cdiv(W, ST_IMG_TW), fldr_cdiv(H, 4 * ST_IMG_TH), N);
        hipLaunchKernelGGL((splat_band_kernel<MODE, 3, ST_IMG_TW, ST_IMG_TH>), grid, dim3(256), 0, s, img, ibs, ics, flow, metric, blk, sbt, out, C, H, W, 1, nsb_x, nsb);
    } else {                       // feature maps: groups of 12 channels, 64 x 4 band per wave (13.5 KB)
        const int groups = fldr_cdiv(C, 12);
        dim3 grid(fldr_cdiv(W, 64), fldr_cdiv(H, 4 * 4), N * groups);
        hipLaunchKernelGGL((splat_band_kernel<MODE, 12, 64, 4>), grid, dim3(256), 0, s, img, ibs, ics, flow, metric, blk, sbt, out, C, H, W, groups, nsb_x, nsb);
    }
}

#endif  // FLDR_TEST_HOOKS (tile + band kernels)

// floats of one flow-bounds table (block + super-block intervals): the workspace of fldr_splat_bounds_upsampled[_pair] /
// fldr_softsplat_acc64 (and of the test build's fldr_softsplat_tile)
extern "C" int64_t fldr_softsplat_tile_ws_floats(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return FLDR_E_ARG;
    const int64_t nsb = (int64_t)fldr_cdiv(W, ST_SBX * ST_BW) * fldr_cdiv(H, ST_SBY * ST_BH);
    return (int64_t)N * nsb * (ST_SB_BLOCKS + 1) * 4;
}

#ifdef FLDR_TEST_HOOKS
template <int MODE>
static void splat_tile_launch(const float* img, int64_t ibs, int64_t ics, const float* flow, const float* metric, const float* blk, const float* sbt,
                              float* out, int N, int C, int H, int W, int nsb_x, int nsb, hipStream_t s) {
    if (C <= 3) {                  // images: every channel + the normalisation accumulator in one 128 x 32 tile (64 KB of LDS)
        dim3 grid(fldr_cdiv(W, 128), fldr_cdiv(H, 32), N);
        hipLaunchKernelGGL((splat_tile_kernel<MODE, 3, 128, 32>), grid, dim3(256), 0, s, img, ibs, ics, flow, metric, blk, sbt, out, C, H, W, 1, nsb_x, nsb);
    } else {                       // feature maps: groups of 12 channels, 64 x 16 tiles (52 KB)
        const int groups = fldr_cdiv(C, 12);
        dim3 grid(fldr_cdiv(W, 64), fldr_cdiv(H, 16), N * groups);
        hipLaunchKernelGGL((splat_tile_kernel<MODE, 12, 64, 16>), grid, dim3(256), 0, s, img, ibs, ics, flow, metric, blk, sbt, out, C, H, W, groups, nsb_x, nsb);
    }
}

// FunctionSoftsplat (softSplat.py:320-352) end to end, destination-owned.  ws: fldr_softsplat_tile_ws_floats floats.
static int g_splat_tile_variant = 1;     // 0: LDS-atomic tiles, 1: claim-and-add bands
FLDR_HOOK int fldr_debug_splat_tile_variant(int v) { if (v == 0 || v == 1) g_splat_tile_variant = v; return g_splat_tile_variant; }

extern "C" int fldr_softsplat_tile_strided(const float* img, int64_t img_bstride, int64_t img_cstride, const float* flow,
                                           const float* metric, float* out, float* ws, int N, int C, int H, int W, int mode,
                                           fldr_stream_t stream);

extern "C" int fldr_softsplat_tile(const float* img, const float* flow, const float* metric, float* out, float* ws,
                                   int N, int C, int H, int W, int mode, fldr_stream_t stream) {
    return fldr_softsplat_tile_strided(img, (int64_t)C * H * W, (int64_t)H * W, flow, metric, out, ws, N, C, H, W, mode, stream);
}

// img: sample n, channel c at img + n*img_bstride + c*img_cstride (floats), each [H,W] plane contiguous.
// mode_flags bit 0: ws already holds the bounds table (fldr_splat_bounds_upsampled)
static int splat_tile_run(const float* img, int64_t img_bstride, int64_t img_cstride, const float* flow,
                          const float* metric, float* out, float* ws, int N, int C, int H, int W, int mode, int mode_flags,
                          fldr_stream_t stream) {
    FLDR_CHECK_ARG(img && flow && out && ws && N > 0 && C > 0 && H > 0 && W > 0 && mode >= 0 && mode <= 3);
    FLDR_CHECK_ARG(mode != 2 || metric != nullptr);
    if (W > 65535 * ST_BW || H > 32767 * ST_BH) return FLDR_E_SHAPE;
    if ((int64_t)fldr_cdiv(W, ST_SBX * ST_BW) * fldr_cdiv(H, ST_SBY * ST_BH) > 65535) return FLDR_E_SHAPE;
    const int nsb_x = fldr_cdiv(W, ST_SBX * ST_BW), nsb = nsb_x * fldr_cdiv(H, ST_SBY * ST_BH);
    float* blk = ws;
    float* sbt = ws + (int64_t)N * nsb * ST_SB_BLOCKS * 4;
    hipStream_t s = fldr_s(stream);
    if (!(mode_flags & 1)) fldr_splat_bounds_launch(flow, 2 * (int64_t)H * W, blk, sbt, N, H, W, nsb_x, nsb, s);
    if (g_splat_tile_variant == 1) {
        switch (mode) {
            case 0: splat_band_launch<0>(img, img_bstride, img_cstride, flow, metric, blk, sbt, out, N, C, H, W, nsb_x, nsb, s); break;
            case 1: splat_band_launch<1>(img, img_bstride, img_cstride, flow, metric, blk, sbt, out, N, C, H, W, nsb_x, nsb, s); break;
            case 2: splat_band_launch<2>(img, img_bstride, img_cstride, flow, metric, blk, sbt, out, N, C, H, W, nsb_x, nsb, s); break;
            default: splat_band_launch<3>(img, img_bstride, img_cstride, flow, metric, blk, sbt, out, N, C, H, W, nsb_x, nsb, s); break;
        }
        FLDR_LAUNCH_RET();
    }
    switch (mode) {
        case 0: splat_tile_launch<0>(img, img_bstride, img_cstride, flow, metric, blk, sbt, out, N, C, H, W, nsb_x, nsb, s); break;
        case 1: splat_tile_launch<1>(img, img_bstride, img_cstride, flow, metric, blk, sbt, out, N, C, H, W, nsb_x, nsb, s); break;
        case 2: splat_tile_launch<2>(img, img_bstride, img_cstride, flow, metric, blk, sbt, out, N, C, H, W, nsb_x, nsb, s); break;
        default: splat_tile_launch<3>(img, img_bstride, img_cstride, flow, metric, blk, sbt, out, N, C, H, W, nsb_x, nsb, s); break;
    }
    FLDR_LAUNCH_RET();
}

extern "C" int fldr_softsplat_tile_strided(const float* img, int64_t img_bstride, int64_t img_cstride, const float* flow,
                                           const float* metric, float* out, float* ws, int N, int C, int H, int W, int mode,
                                           fldr_stream_t stream) {
    return splat_tile_run(img, img_bstride, img_cstride, flow, metric, out, ws, N, C, H, W, mode, 0, stream);
}

// fldr_softsplat_tile_strided with the bounds table already in `ws` (any table whose block / super-block intervals CONTAIN the
// flow values of their pixels gives the exact result: the table only selects candidate sources).
extern "C" int fldr_softsplat_tile_prebounded(const float* img, int64_t img_bstride, int64_t img_cstride, const float* flow,
                                              const float* metric, float* out, float* ws, int N, int C, int H, int W, int mode,
                                              fldr_stream_t stream) {
    return splat_tile_run(img, img_bstride, img_cstride, flow, metric, out, ws, N, C, H, W, mode, 1, stream);
}
#endif  // FLDR_TEST_HOOKS (the destination-owned splats of rounds 1-2: fldr_softsplat_tile*)

// The bounds table of `ws` for flow = F.interpolate(scale * flow_lo, (H, W), bilinear) * mul, from flow_lo alone (conservative:
// see splat_bounds_up_kernel).  flow_lo: sample n at flow_lo + n*lo_bstride, [2,h,w] contiguous; scale_mode 0: 1, 1: t[n],
// 2: 1 - t[n] (t may be null for mode 0).
extern "C" int fldr_splat_bounds_upsampled(const float* flow_lo, int64_t lo_bstride, const float* t, int scale_mode, float mul,
                                           float* ws, int N, int h, int w, int H, int W, fldr_stream_t stream) {
    FLDR_CHECK_ARG(flow_lo && ws && N > 0 && h > 0 && w > 0 && H >= h && W >= w && scale_mode >= 0 && scale_mode <= 2 && mul > 0.0f);
    FLDR_CHECK_ARG(scale_mode == 0 || t != nullptr);
    if (W > 65535 * ST_BW || H > 32767 * ST_BH) return FLDR_E_SHAPE;
    if ((int64_t)fldr_cdiv(W, ST_SBX * ST_BW) * fldr_cdiv(H, ST_SBY * ST_BH) > 65535) return FLDR_E_SHAPE;
    const int nsb_x = fldr_cdiv(W, ST_SBX * ST_BW), nsb = nsb_x * fldr_cdiv(H, ST_SBY * ST_BH);
    float* blk = ws;
    float* sbt = ws + (int64_t)N * nsb * ST_SB_BLOCKS * 4;
    splat_bounds_up_launch(flow_lo, lo_bstride, t, scale_mode, mul, blk, sbt, h, w, H, W, nsb_x, nsb, 0, N, N, fldr_s(stream));
    FLDR_LAUNCH_RET();
}

// Both bounds tables of a pair call of fldr_softsplat_acc64 (flags bit 1) in ONE launch, from the [N,4,h,w] flow of a pyramid
// level (channels 0-1 flow_10, 2-3 flow_01; samples lo_bstride floats apart, 0 = 4*h*w): pair 1 = the level-0 image splats
// (problem 0: flow = up(t * flow_01) * mul, problem 1: up((1 - t) * flow_10) * mul), pair 2 = the feature splats (problem 0:
// up(flow_10) * mul, problem 1: up(flow_01) * mul).  ws: 2 * fldr_softsplat_tile_ws_floats(N, H, W) floats.
extern "C" int fldr_splat_bounds_upsampled_pair(const float* flow_l, int64_t lo_bstride, const float* t, int pair, float mul, float* ws,
                                                int N, int h, int w, int H, int W, fldr_stream_t stream) {
    FLDR_CHECK_ARG(flow_l && ws && N > 0 && h > 0 && w > 0 && H >= h && W >= w && (pair == 1 || pair == 2) && mul > 0.0f);
    FLDR_CHECK_ARG(pair == 2 || t != nullptr);
    if (W > 65535 * ST_BW || H > 32767 * ST_BH) return FLDR_E_SHAPE;
    if ((int64_t)fldr_cdiv(W, ST_SBX * ST_BW) * fldr_cdiv(H, ST_SBY * ST_BH) > 65535) return FLDR_E_SHAPE;
    const int nsb_x = fldr_cdiv(W, ST_SBX * ST_BW), nsb = nsb_x * fldr_cdiv(H, ST_SBY * ST_BH);
    float* blk = ws;
    float* sbt = ws + (int64_t)2 * N * nsb * ST_SB_BLOCKS * 4;
    splat_bounds_up_launch(flow_l, lo_bstride ? lo_bstride : 4 * (int64_t)h * w, t, 0, mul, blk, sbt, h, w, H, W, nsb_x, nsb, pair, N, 2 * N, fldr_s(stream));
    FLDR_LAUNCH_RET();
}
