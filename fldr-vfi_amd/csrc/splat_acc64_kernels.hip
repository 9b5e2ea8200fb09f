// Softmax splatting (softSplat.py:12-52, 320-352) with destination-owned tiles and FP64 LDS ATOMICS.
//
// Measured on gfx950 (tools/ubench/lds_int_atomic_bench.hip, ns per wave-instruction per CU, conflict-free / two lanes per
// cell / random cells): ds_add_f32 80 / 94 / 80 — the reason the f32 tile kernel of splat_tile_kernels.hip lost to global
// atomics and the band kernel went to claim rounds and DPP merging (~450 vector instructions per source row) — but
// ds_add_f64 3.5 / 8.4 / 8.7 and ds_add_u64 2.6 / 5.1 / 4.5: the 64-bit LDS atomics run at (nearly) the plain LDS rate.
// So: every workgroup owns a TW x TH tile of the output as fp64 accumulators in LDS (channel-major: neighbouring cells are
// neighbouring 8-byte words, conflict-free for a coherent flow), every source pixel whose footprint can reach the tile adds
// its four corner contributions (the fp32 products of the reference kernel, softSplat.py:40-51, widened to fp64) with one
// ds_add_f64 each, and the finished tile is normalised and written once.  No global atomics, no accumulator tensor, no memset,
// no normalisation pass, no conflict handling — and the fp64 sums make the result independent of the summation order to
// ~1e-16 relative (run-to-run differences only where a sum lands within that of a rounding boundary of the fp32 output).
//
// Candidate sources: the flow-bounds tables of splat_tile_kernels.hip (per 64x4 block and 256x64 super-block; from the flow
// itself or, for an upsampled flow, from its low-resolution source).  A tile tests the super-blocks, then the blocks of the
// matching ones, and then walks EITHER the rectangle of source pixels those blocks' joint bounds allow (the normal case: a
// smooth flow; no overhang beyond the bounds' slack, threads are laid out along the rectangle's rows) OR the queued blocks
// (scattered candidates).  Queue overflow or too many super-blocks: the rectangle grows to the blocks' bounding box / the
// image.  Every visited source is re-tested per corner, so any superset of the true candidates gives the exact result.
#include "splat_common.h"

#define SA_SBQ 64                    // matching super-blocks a workgroup can list
#define SA_Q 512                     // matching blocks a workgroup can queue
#ifndef SA_WHOLE_PIXELS
#define SA_WHOLE_PIXELS 2304         // up to 36 x 64: every tile walks the whole map instead of reading bounds tables
#endif
#ifndef SA_IMG_TW
#define SA_IMG_TW 64                 // image configuration: tile, tiles per workgroup, source pixels per thread in flight
#define SA_IMG_TH 24
#define SA_IMG_K 3
#define SA_IMG_U 4
#endif
#ifndef SA_IMGQ_U
#define SA_IMGQ_U 1                  // 4-pixel runs per thread in flight (QUAD walk)
#endif
#ifndef SA_FEAT_TW
#define SA_FEAT_TW 32                // 16-channel-group configuration
#define SA_FEAT_TH 8
#define SA_FEAT_K 1
#define SA_FEAT_U 1
#endif

typedef _Float16 sa_h8 __attribute__((ext_vector_type(8)));

struct SaProblem {
    const float* img;                // sample n, channel c at img + n * img_bstride + c * img_cstride; [H,W] planes contiguous
    const float* flow;               // [N,2,H,W], samples flow_bstride floats apart
    const float* metric;             // [N,1,H,W] or null
    const float* blk;                // bounds tables (splat_tile_kernels.hip)
    const float* sbt;
    float* out_f32;                  // [N,C,H,W] or null
    unsigned char* out_spk;          // packed [N,C,H,W] or null
    int64_t img_bstride, img_cstride, flow_bstride;
};
struct SaArgs {
    SaProblem p[2];
    int N, C, H, W, groups, nsb_x, nsb;
    int whole;                                    // 1: no bounds tables, every tile walks the whole map
    int gpw;                                      // channel groups per workgroup (1 or `groups`)
    int tiles_x, st_y, n_st, total, per_xcd;      // tiles per row, super-tile rows, super-tiles per (problem, sample, group), all work items, items per XCD
};

template <int CB, int U, bool QUAD>
struct SaBuf {
    int x[U], y[U];
    bool ok[U];
    float fx[U], fy[U], mv[U], val[U][CB];
};
// QUAD: an item is a run of four horizontally adjacent source pixels (x a multiple of 4), loaded with one 16-byte access per plane
template <int CB, int U>
struct SaBuf<CB, U, true> {
    int x[U], y[U];
    bool ok[U];
    float4 fx[U], fy[U], mv[U], val[U][CB];
};
__device__ __forceinline__ void sa_pin4(float4& v) { fldr_pin(v.x); fldr_pin(v.y); fldr_pin(v.z); fldr_pin(v.w); }
__device__ __forceinline__ float sa_at(const float4& v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }

// MODE: 0 summation; 1 average; 2 linear; 3 softmax.  CB value channels per workgroup (+ the normalisation accumulator when
// MODE >= 1).  A workgroup owns a column of K vertically adjacent TW x TH tiles: ONE candidate search for the whole column
// (the search is a chain of two dependent table reads and three barriers: ~40 % of the kernel when done per tile), then the
// tiles one after the other through the same LDS accumulator.  U source pixels per thread and iteration; all loads of
// iteration i + 1 — and of the next tile's first iteration — are in flight under iteration i / under the tile's write-out.
// Work items are dealt to the XCDs in contiguous row-major ranges (blockIdx & 7 = XCD): neighbouring tiles read overlapping
// source rows and the two halves of partly covered 128-byte lines, which then hit in the same L2.
//
// QUAD (images whose rows are a multiple of 4 pixels, 16-byte aligned planes): a thread's item is a run of FOUR adjacent source
// pixels.  (1) Every plane is read with 16-byte lane accesses — this chip streams 3.9-4.1 TB/s through 4-byte lanes and
// 5.5-5.9 TB/s through 16-byte ones (tools/ubench/plane_bw_bench), and the 4-byte walk was pinned there.  (2) Under a coherent
// flow pixel j's east corners ARE pixel j + 1's west corners: the thread adds the two fp32 products in fp64 (exactly) in
// registers and issues one atomic — 10 instead of 16 per run and channel.  (3) Lanes are then 4 cells apart, so the
// accumulator rows are stored de-interleaved by 4 (cell x at (x & 3) * TW / 4 + x / 4): the lanes of one atomic instruction
// hit consecutive 8-byte words again.
template <int MODE, int CB, int TW, int TH, int K, int U, bool QUAD>
__global__ __launch_bounds__(256) void splat_acc64_kernel(SaArgs a) {
#pragma clang fp contract(off)
    constexpr int CA = MODE >= 1 ? CB + 1 : CB;
    constexpr int CELLS = TW * TH;
    extern __shared__ __attribute__((aligned(16))) unsigned char sa_smem[];
    double* acc = reinterpret_cast<double*>(sa_smem);                           // [CA][CELLS]
    int* blkq = reinterpret_cast<int*>(sa_smem + (size_t)CA * CELLS * 8);       // [SA_Q]
    float* redf = reinterpret_cast<float*>(blkq + SA_Q);                        // [4][4]
    int* redi = reinterpret_cast<int*>(redf + 16);                              // [4][4]
    int* cnt = redi + 16;                                                       // [2]
    unsigned short* sbq = reinterpret_cast<unsigned short*>(cnt + 2);           // [SA_SBQ]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lin = (int)(blockIdx.x & 7) * a.per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= a.per_xcd || lin >= a.total) return;           // workgroup-uniform
    const int zi = lin / a.n_st, st = lin - zi * a.n_st;
    const int sty = st / a.tiles_x, stx = st - sty * a.tiles_x;
    // (a.gpw channel groups per workgroup: 1, or all of them — then the candidate search and the tile plans serve every group)
    const int wg_groups = a.groups / a.gpw;                             // group slots in the work-item index
    const int per_prob = a.N * wg_groups;
    const int prob = zi / per_prob;
    const int rem = zi - prob * per_prob;
    const int n = rem / wg_groups, grp0 = (rem - n * wg_groups) * a.gpw;
    const SaProblem& P = a.p[prob];
    const int C = a.C, H = a.H, W = a.W, nsb_x = a.nsb_x, nsb = a.nsb;
    int cbase = grp0 * CB;                                              // first channel of the group being accumulated
    const int tx0 = stx * TW, sy0 = sty * (K * TH);
    const int64_t HW = (int64_t)H * W;
    const float ftx0 = (float)tx0, ftx1 = (float)(tx0 + TW - 1), fsy0 = (float)sy0, fsy1 = (float)(min(sy0 + K * TH, H) - 1);
    const float INF = __builtin_inff();

    // ---- which super-blocks, then which blocks, can reach this column of tiles ----
    const float* sbn = P.sbt + (int64_t)n * nsb * 4;
    const float* bkn = P.blk + (int64_t)n * nsb * ST_SB_BLOCKS * 4;
    // a.whole: no table at all, every tile walks the whole map (maps of a few thousand pixels: cheaper than a bounds launch
    // and the search's two dependent table reads).  nsb <= SA_SBQ (feature maps): the super-block level is skipped, every
    // super-block's blocks are tested directly.
    const bool small = nsb <= SA_SBQ;                                   // kernel-uniform
    if (tid < 2) cnt[tid] = small && tid == 0 ? nsb : 0;
    if (small && tid < nsb) sbq[tid] = (unsigned short)tid;
    float4 b_first = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (!small && !a.whole) b_first = *reinterpret_cast<const float4*>(sbn + (tid < nsb ? tid : 0) * 4);   // in flight while the accumulator is cleared
    for (int i = tid; i < CA * CELLS / 2; i += 256) reinterpret_cast<double2*>(acc)[i] = make_double2(0.0, 0.0);
    if ((CA * CELLS) & 1) { if (tid == 0) acc[CA * CELLS - 1] = 0.0; }
    __syncthreads();
    if (!small && !a.whole) {
        for (int s = tid; s < nsb; s += 256) {
            const float4 b = s == tid ? b_first : *reinterpret_cast<const float4*>(sbn + s * 4);
            const int sx = (s % nsb_x) * (ST_SBX * ST_BW), sy = (s / nsb_x) * (ST_SBY * ST_BH);
            if (st_match(b, (float)sx, (float)(sx + ST_SBX * ST_BW - 1), (float)sy, (float)(sy + ST_SBY * ST_BH - 1), ftx0, ftx1, fsy0, fsy1)) {
                const int i = atomicAdd(&cnt[0], 1);
                if (i < SA_SBQ) sbq[i] = (unsigned short)s;
            }
        }
        __syncthreads();
    }
    const int n_sb = cnt[0];
    const bool sb_over = a.whole || n_sb > SA_SBQ;                      // workgroup-uniform: walk the whole map
    float rxmin = INF, rxmax = -INF, rymin = INF, rymax = -INF;         // joint flow bounds of the matching blocks
    int cx0 = 0x7fffffff, cx1 = -1, cy0 = 0x7fffffff, cy1 = -1;         // their bounding box (pixels, inclusive)
    if (!sb_over) {
        for (int i = wv; i < n_sb; i += 4) {                            // one matching super-block per wave, one block per lane
            const int s = sbq[i];
            const float4 b = *reinterpret_cast<const float4*>(bkn + ((int64_t)s * ST_SB_BLOCKS + lane) * 4);
            const int bx = (s % nsb_x) * ST_SBX + (lane % ST_SBX), by = (s / nsb_x) * ST_SBY + (lane / ST_SBX);
            const int sx = bx * ST_BW, sy = by * ST_BH;
            if (st_match(b, (float)sx, (float)(sx + ST_BW - 1), (float)sy, (float)(sy + ST_BH - 1), ftx0, ftx1, fsy0, fsy1)) {
                rxmin = fminf(rxmin, b.x); rxmax = fmaxf(rxmax, b.y); rymin = fminf(rymin, b.z); rymax = fmaxf(rymax, b.w);
                cx0 = min(cx0, sx); cx1 = max(cx1, sx + ST_BW - 1); cy0 = min(cy0, sy); cy1 = max(cy1, sy + ST_BH - 1);
                const int k = atomicAdd(&cnt[1], 1);
                if (k < SA_Q) blkq[k] = (by << 16) | bx;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            rxmin = fminf(rxmin, __shfl_xor(rxmin, o)); rxmax = fmaxf(rxmax, __shfl_xor(rxmax, o));
            rymin = fminf(rymin, __shfl_xor(rymin, o)); rymax = fmaxf(rymax, __shfl_xor(rymax, o));
            cx0 = min(cx0, __shfl_xor(cx0, o)); cx1 = max(cx1, __shfl_xor(cx1, o));
            cy0 = min(cy0, __shfl_xor(cy0, o)); cy1 = max(cy1, __shfl_xor(cy1, o));
        }
        if (lane == 0) {
            redf[wv * 4 + 0] = rxmin; redf[wv * 4 + 1] = rxmax; redf[wv * 4 + 2] = rymin; redf[wv * 4 + 3] = rymax;
            redi[wv * 4 + 0] = cx0; redi[wv * 4 + 1] = cx1; redi[wv * 4 + 2] = cy0; redi[wv * 4 + 3] = cy1;
        }
    }
    __syncthreads();
    int n_match = 0;
    bool bounded = false;                                               // the joint flow bounds are finite and int-safe
    if (!sb_over) {
        n_match = cnt[1];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            rxmin = fminf(rxmin, redf[k * 4 + 0]); rxmax = fmaxf(rxmax, redf[k * 4 + 1]);
            rymin = fminf(rymin, redf[k * 4 + 2]); rymax = fmaxf(rymax, redf[k * 4 + 3]);
            cx0 = min(cx0, redi[k * 4 + 0]); cx1 = max(cx1, redi[k * 4 + 1]); cy0 = min(cy0, redi[k * 4 + 2]); cy1 = max(cy1, redi[k * 4 + 3]);
        }
        bounded = fabsf(rxmin) < 1.0e6f && fabsf(rxmax) < 1.0e6f && fabsf(rymin) < 1.0e6f && fabsf(rymax) < 1.0e6f;
    }

    const float* fl = P.flow + (int64_t)n * P.flow_bstride;
    const float* mt = P.metric ? P.metric + (int64_t)n * HW : nullptr;
    const float* inn = P.img + (int64_t)n * P.img_bstride;
    float* on = P.out_f32 ? P.out_f32 + (int64_t)n * C * HW : nullptr;
    const int G = (C + 7) >> 3;
    unsigned char* sp = P.out_spk ? P.out_spk + (int64_t)n * G * 2 * HW * 16 : nullptr;
    bool bad = false;

    // ---- the walk of one tile: a rectangle of source pixels, or the queued blocks (everything here is workgroup-uniform) ----
    int ty0 = 0;                                                        // tile being accumulated
    int X0 = 0, Y0 = 0, Rw = 1, Rh = 0, n_chunks = 0;
    bool rect = true;
    int rx = 0, ry = 0, dqx = 0, dqy = 0;                               // rectangle walk: this thread's next pixel (column, row) and the step of 256 pixels
    auto plan_tile = [&](int k) __attribute__((always_inline)) {
        ty0 = sy0 + k * TH;
        const float fty0 = (float)ty0, fty1 = (float)(ty0 + TH - 1);
        int X1 = W - 1, Y1 = H - 1;
        X0 = 0; Y0 = 0; rect = true; n_chunks = 0;
        if (sb_over) {
            n_chunks = (int)((HW + 255) / 256);
        } else if (n_match > 0) {
            X0 = cx0; X1 = min(cx1, W - 1); Y0 = cy0; Y1 = min(cy1, H - 1);
            if (bounded) {
                // x + fx >= tx0 - 2 and x + fx <= tx1 + 1 for some fx in [rxmin, rxmax]  (st_match for a single pixel)
                X0 = max(X0, (int)floorf(ftx0 - 2.0f - rxmax)); X1 = min(X1, (int)ceilf(ftx1 + 1.0f - rxmin));
                Y0 = max(Y0, (int)floorf(fty0 - 2.0f - rymax)); Y1 = min(Y1, (int)ceilf(fty1 + 1.0f - rymin));
            }
            const int64_t area = (int64_t)max(X1 - X0 + 1, 0) * max(Y1 - Y0 + 1, 0);
            rect = n_match > SA_Q || area <= (int64_t)256 * n_match;    // never more pixels than the block walk would visit
            n_chunks = rect ? (int)((area + 255) / 256) : n_match;
        }
        Rw = max(X1 - X0 + 1, 1); Rh = max(Y1 - Y0 + 1, 0);
        if constexpr (QUAD) {
            // the walk's items are runs of 4 pixels: the rectangle widens to multiples of 4 (W % 4 == 0), the block walk
            // takes four blocks (one per wave: 64 runs) per iteration
            X0 &= ~3; X1 = min(X1 | 3, W - 1);
            Rw = max((X1 - X0 + 1) >> 2, 1);
            if (n_chunks > 0) n_chunks = rect ? (int)(((int64_t)Rw * Rh + 255) / 256) : (n_match + 3) >> 2;
        }
        rx = tid % Rw; ry = tid / Rw;
        dqx = 256 % Rw; dqy = 256 / Rw;
#if defined(SA_ABLATE) && SA_ABLATE == 1                              // diagnostic: search + zero + finish only
        n_chunks = 0;
#endif
    };

    auto load_chunk = [&](SaBuf<CB, U, QUAD>& b, int u, int k) __attribute__((always_inline)) {
        int x, y;
        bool ok;
        if (rect) {
            x = X0 + (QUAD ? 4 * rx : rx); y = Y0 + ry;
            ok = k < n_chunks && ry < Rh;
            rx += dqx; ry += dqy;
            if (rx >= Rw) { rx -= Rw; ++ry; }
        } else if constexpr (QUAD) {
            const int bi = 4 * k + wv;
            const int e = blkq[bi < n_match ? bi : 0];
            x = (e & 0xFFFF) * ST_BW + (lane & 15) * 4; y = (e >> 16) * ST_BH + (lane >> 4);
            ok = bi < n_match && x < W && y < H;
        } else {
            const int e = blkq[k < n_chunks ? k : 0];
            x = (e & 0xFFFF) * ST_BW + lane; y = (e >> 16) * ST_BH + wv;
            ok = k < n_chunks && x < W && y < H;
        }
        b.x[u] = x; b.y[u] = y; b.ok[u] = ok;
        const int64_t pix = ok ? (int64_t)y * W + x : 0;                // clamped: every load below is unconditional
        if constexpr (QUAD) {
            b.fx[u] = *reinterpret_cast<const float4*>(fl + pix); b.fy[u] = *reinterpret_cast<const float4*>(fl + HW + pix);
            b.mv[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if ((MODE == 2 || MODE == 3) && mt != nullptr) b.mv[u] = *reinterpret_cast<const float4*>(mt + pix);
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                const int cc = cbase + c < C ? cbase + c : C - 1;
                b.val[u][c] = *reinterpret_cast<const float4*>(inn + (int64_t)cc * P.img_cstride + pix);
            }
        } else {
            b.fx[u] = fl[pix]; b.fy[u] = fl[HW + pix];
            b.mv[u] = 0.0f;
            if ((MODE == 2 || MODE == 3) && mt != nullptr) b.mv[u] = mt[pix];
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                const int cc = cbase + c < C ? cbase + c : C - 1;
                b.val[u][c] = inn[(int64_t)cc * P.img_cstride + pix];
            }
        }
    };
    auto load_iter = [&](SaBuf<CB, U, QUAD>& b, int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; ++u) load_chunk(b, u, k0 + u);
    };
    auto process = [&](SaBuf<CB, U, QUAD>& b) __attribute__((always_inline)) {
      if constexpr (QUAD) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            sa_pin4(b.fx[u]); sa_pin4(b.fy[u]); sa_pin4(b.mv[u]);
#pragma unroll
            for (int c = 0; c < CB; ++c) sa_pin4(b.val[u][c]);
        }
#if defined(SA_ABLATE) && SA_ABLATE == 3                              // diagnostic: the walk's loads only
        return;
#endif
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!b.ok[u]) continue;
            int x0[4], y0[4], iw[4], ie[4];
            float wnw[4], wne[4], wsw[4], wse[4], wg[4];
            bool xa[4], xb[4], ya[4], yb[4];
            bool any = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const StGeom g = st_geom(b.x[u] + j, b.y[u], sa_at(b.fx[u], j), sa_at(b.fy[u], j), W, H);     // x0 in [-2, W + 1], y0 in [-2, H + 1]
                x0[j] = g.x0; y0[j] = g.y0; wnw[j] = g.wnw; wne[j] = g.wne; wsw[j] = g.wsw; wse[j] = g.wse;
                const uint32_t ulx = (uint32_t)(g.x0 - tx0), uly = (uint32_t)(g.y0 - ty0), ux1 = ulx + 1u;      // tile-local north-west corner (wraps when outside)
                // a corner counts when it lies in this tile (cells right of / below the image exist in a partial tile and are never written out)
                xa[j] = ulx < (uint32_t)TW; xb[j] = ux1 < (uint32_t)TW; ya[j] = uly < (uint32_t)TH; yb[j] = uly + 1u < (uint32_t)TH;
                iw[j] = (int)(uly * (uint32_t)TW + (ulx & 3u) * (uint32_t)(TW / 4) + (ulx >> 2));              // west / east column, row y0 (+ TW: row y0 + 1)
                ie[j] = (int)(uly * (uint32_t)TW + (ux1 & 3u) * (uint32_t)(TW / 4) + (ux1 >> 2));
                any = any || ((xa[j] || xb[j]) && (ya[j] || yb[j]));
                wg[j] = 1.0f;
                if (MODE == 2) wg[j] = sa_at(b.mv[u], j);
                if (MODE == 3 && mt != nullptr) wg[j] = expf(sa_at(b.mv[u], j));
            }
            if (!any) continue;                                           // no footprint of the run touches the tile
            float v[4][CA];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < CA; ++c) {
                    float t = wg[j];                                       // normalisation accumulator
                    if (c < CB) {
                        t = sa_at(b.val[u][c < CB ? c : 0], j);
                        if (MODE == 3) t = (t + 1.0f) / 2.0f;              // softSplat.py:334
                        if (MODE >= 2) t = t * wg[j];                      // :328 / :338
                        if (cbase + c >= C) t = 0.0f;
                    }
                    v[j][c] = t;
                }
            // hand-off j -> j + 1: pixel j's east column is pixel j + 1's west column, same rows (then their validity agrees too)
            bool gn[4], gs[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool coh = j < 3 && x0[j] + 1 == x0[j < 3 ? j + 1 : j] && y0[j] == y0[j < 3 ? j + 1 : j];
                gn[j] = coh && xb[j] && ya[j]; gs[j] = coh && xb[j] && yb[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int jp = j > 0 ? j - 1 : 0;
                if (xa[j] && ya[j]) {
#pragma unroll
                    for (int c = 0; c < CA; ++c) {                         // the reference's fp32 products (softSplat.py:40-51), summed in fp64
                        double d = (double)(v[j][c] * wnw[j]);
                        if (j > 0) d += (double)(gn[jp] ? v[jp][c] * wne[jp] : 0.0f);
                        atomicAdd(acc + c * CELLS + iw[j], d);
                    }
                }
                if (xa[j] && yb[j]) {
#pragma unroll
                    for (int c = 0; c < CA; ++c) {
                        double d = (double)(v[j][c] * wsw[j]);
                        if (j > 0) d += (double)(gs[jp] ? v[jp][c] * wse[jp] : 0.0f);
                        atomicAdd(acc + c * CELLS + iw[j] + TW, d);
                    }
                }
                if (xb[j] && ya[j] && !gn[j]) {
#pragma unroll
                    for (int c = 0; c < CA; ++c) atomicAdd(acc + c * CELLS + ie[j], (double)(v[j][c] * wne[j]));
                }
                if (xb[j] && yb[j] && !gs[j]) {
#pragma unroll
                    for (int c = 0; c < CA; ++c) atomicAdd(acc + c * CELLS + ie[j] + TW, (double)(v[j][c] * wse[j]));
                }
            }
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            fldr_pin(b.fx[u]); fldr_pin(b.fy[u]); fldr_pin(b.mv[u]);
#pragma unroll
            for (int c = 0; c < CB; ++c) fldr_pin(b.val[u][c]);
        }
#if defined(SA_ABLATE) && SA_ABLATE == 3                              // diagnostic: the walk's loads only
        return;
#endif
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!b.ok[u]) continue;
            const StGeom g = st_geom(b.x[u], b.y[u], b.fx[u], b.fy[u], W, H);
            const int lx = g.x0 - tx0, ly = g.y0 - ty0;                    // tile-local north-west corner
            if (lx < -1 || lx >= TW || ly < -1 || ly >= TH) continue;     // footprint misses the tile
            // a corner counts here when it lies in this tile AND in the image (tile origins are >= 0)
            const bool cxa = lx >= 0 && g.x0 < W, cxb = lx + 1 < TW && g.x0 + 1 < W;
            const bool cya = ly >= 0 && g.y0 < H, cyb = ly + 1 < TH && g.y0 + 1 < H;
            const bool vnw = cxa && cya, vne = cxb && cya, vsw = cxa && cyb, vse = cxb && cyb;
            float wgt = 1.0f;
            if (MODE == 2) wgt = b.mv[u];
            if (MODE == 3 && mt != nullptr) wgt = expf(b.mv[u]);
            double* cell = acc + (ly * TW + lx);
#pragma unroll
            for (int c = 0; c < CA; ++c) {
                float v;
                if (c < CB) {
                    v = b.val[u][c];
                    if (MODE == 3) v = (v + 1.0f) / 2.0f;                  // softSplat.py:334
                    if (MODE >= 2) v = v * wgt;                            // :328 / :338
                    if (cbase + c >= C) v = 0.0f;
                } else {
                    v = wgt;                                               // normalisation accumulator
                }
                double* pc = cell + c * CELLS;
#if defined(SA_ABLATE) && SA_ABLATE == 2                              // diagnostic: everything but the LDS atomics
                asm volatile("" :: "v"(pc), "v"(v * g.wnw), "v"(v * g.wne), "v"(v * g.wsw), "v"(v * g.wse), "v"((int)vnw + (int)vne + (int)vsw + (int)vse));
#else
                if (vnw) atomicAdd(pc, (double)(v * g.wnw));               // the reference's fp32 products (softSplat.py:40-51), summed in fp64
                if (vne) atomicAdd(pc + 1, (double)(v * g.wne));
                if (vsw) atomicAdd(pc + TW, (double)(v * g.wsw));
                if (vse) atomicAdd(pc + TW + 1, (double)(v * g.wse));
#endif
            }
        }
      }
    };
    // finish and write tile (tx0, fy0): (acc / norm - 0.5) * 2, norm 0 -> 1 (softSplat.py:343-349); the cells are left zeroed
    auto finish = [&](int fy0, int cbase) __attribute__((always_inline)) {
      if constexpr (QUAD) {
        static_assert(!QUAD || (CB % 8 != 0 && TW % 4 == 0), "the run walk serves the image configuration (fp32 output)");
        for (int i = tid; i < CELLS / 4; i += 256) {
            const int xq = i % (TW / 4), yy = i / (TW / 4);
            const int x = tx0 + 4 * xq, y = fy0 + yy;
            const int cell = yy * TW + xq;                               // cell (4 xq + j, yy) lives at cell + j * TW / 4
            float norm[4] = {1.0f, 1.0f, 1.0f, 1.0f};
            if (MODE >= 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    norm[j] = (float)acc[CB * CELLS + cell + j * (TW / 4)]; acc[CB * CELLS + cell + j * (TW / 4)] = 0.0;
                    if (norm[j] == 0.0f) norm[j] = 1.0f;
                }
            }
            float o[CB][4];
            // (quotients over one normaliser: reciprocal + corrected product, splat_common.h; true division for normalisers out of its range)
            StRecip rn[4];
            bool fast = true;
            if (MODE >= 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { rn[j] = st_recip(norm[j]); fast = fast && st_recip_safe(norm[j]); }
                fast = __builtin_amdgcn_ballot_w64(!fast) == 0ull;           // wave-uniform
            }
#pragma unroll
            for (int c = 0; c < CB; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = (float)acc[c * CELLS + cell + j * (TW / 4)];
                    acc[c * CELLS + cell + j * (TW / 4)] = 0.0;
                    if (MODE >= 1) v = fast ? st_div(v, rn[j]) : v / norm[j];
                    o[c][j] = (v - 0.5f) * 2.0f;
                }
            if (x >= W || y >= H || !on) continue;                       // (W % 4 == 0: the whole run is inside or outside)
            const int64_t pix = (int64_t)y * W + x;
#pragma unroll
            for (int c = 0; c < CB; ++c)
                if (cbase + c < C) *reinterpret_cast<float4*>(on + (int64_t)(cbase + c) * HW + pix) = make_float4(o[c][0], o[c][1], o[c][2], o[c][3]);
        }
      } else {
        for (int i = tid; i < CELLS; i += 256) {
            const int x = tx0 + i % TW, y = fy0 + i / TW;
            float norm = 1.0f;
            if (MODE >= 1) { norm = (float)acc[CB * CELLS + i]; acc[CB * CELLS + i] = 0.0; if (norm == 0.0f) norm = 1.0f; }
            float o[CB];
            const StRecip rn = st_recip(norm);
            const bool fast = MODE >= 1 && __builtin_amdgcn_ballot_w64(!st_recip_safe(norm)) == 0ull;        // wave-uniform
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                float v = (float)acc[c * CELLS + i];
                acc[c * CELLS + i] = 0.0;
                if (MODE >= 1) v = fast ? st_div(v, rn) : v / norm;
                o[c] = cbase + c < C ? (v - 0.5f) * 2.0f : 0.0f;
            }
            if (x >= W || y >= H) continue;
            const int64_t pix = (int64_t)y * W + x;
            if (on) {
#pragma unroll
                for (int c = 0; c < CB; ++c)
                    if (cbase + c < C) on[(int64_t)(cbase + c) * HW + pix] = o[c];
            }
            if constexpr (CB % 8 == 0) {
                if (sp) {
#pragma unroll
                    for (int g8 = 0; g8 < CB / 8; ++g8) {
                        const int gi = (cbase >> 3) + g8;
                        if (gi >= G) break;
                        sa_h8 hi, lo;
                        {
                            float xs[8];
                            _Float16 hs[8], ls[8];
#pragma unroll
                            for (int k = 0; k < 8; ++k) xs[k] = o[g8 * 8 + k];
                            fldr_split_hl_group(xs, hs, ls, bad);
#pragma unroll
                            for (int k = 0; k < 8; ++k) { hi[k] = hs[k]; lo[k] = ls[k]; }
                        }
                        unsigned char* d = sp + ((int64_t)gi * 2 * HW + pix) * 16;
                        *reinterpret_cast<sa_h8*>(d) = hi;
                        *reinterpret_cast<sa_h8*>(d + HW * 16) = lo;
                    }
                }
            }
        }
      }
    };

    // units of this workgroup: (tile k, group g) in k-major order; a unit's first loads go out before the previous unit is written
    SaBuf<CB, U, QUAD> b0, b1;
    const int n_units = K * a.gpw;
    auto begin_unit = [&](int u) __attribute__((always_inline)) {
        const int k = u / a.gpw;
        cbase = (grp0 + (u - k * a.gpw)) * CB;
        plan_tile(k);
        if (n_chunks > 0) load_iter(b0, 0);
    };
    begin_unit(0);
#pragma unroll 1
    for (int u = 0; u < n_units; ++u) {
        const int cur_y0 = ty0, cur_cbase = cbase;
        if (cur_y0 >= H) break;                                         // workgroup-uniform
        // this unit's first iteration's loads are already in flight in b0
        for (int k0 = 0; k0 < n_chunks; k0 += 2 * U) {
            const bool more = k0 + U < n_chunks;
            if (more) load_iter(b1, k0 + U);
            process(b0);
            if (!more) break;
            if (k0 + 2 * U < n_chunks) load_iter(b0, k0 + 2 * U);
            process(b1);
        }
        const bool next = u + 1 < n_units && sy0 + ((u + 1) / a.gpw) * TH < H;
        if (next) begin_unit(u + 1);
        __syncthreads();
        finish(cur_y0, cur_cbase);
        if (!next) break;
        __syncthreads();
    }
    fldr_note_range(bad);
}

FLDR_TU_STATUS(acc64)

#ifndef SA_FOLD_MIN_TILES
#define SA_FOLD_MIN_TILES 512
#endif
#ifndef SA_QUAD_DEFAULT
#define SA_QUAD_DEFAULT 1
#endif
static int g_sa_quad = SA_QUAD_DEFAULT;                // images: runs of four pixels per thread where the geometry allows (see the kernel); 0: one pixel per item
FLDR_HOOK int fldr_debug_splat_quad(int v) { if (v == 0 || v == 1) g_sa_quad = v; return g_sa_quad; }
static int g_sa_group_fold = 0;          // measured at 288x480x48, both directions: 72.6 us folded vs 67.1 (1080 workgroups on 1024 slots: a second, nearly empty round)
FLDR_HOOK int fldr_debug_splat_group_fold(int v) { if (v == 0 || v == 1) g_sa_group_fold = v; return g_sa_group_fold; }

template <int MODE, int CB, int TW, int TH, int K, int U, bool QUAD = false>
static int sa_launch2(SaArgs& a, int nprob, hipStream_t s) {
    constexpr int CA = MODE >= 1 ? CB + 1 : CB;
    constexpr int LDS = CA * TW * TH * 8 + SA_Q * 4 + 16 * 4 + 16 * 4 + 2 * 4 + SA_SBQ * 2;
    static_assert(LDS <= 160 * 1024, "tile does not fit the LDS");
    static std::atomic<uint64_t> attr_done{0};
    if (int e = fldr_set_max_lds(reinterpret_cast<const void*>(&splat_acc64_kernel<MODE, CB, TW, TH, K, U, QUAD>), LDS, attr_done)) return e;
    a.tiles_x = fldr_cdiv(a.W, TW);
    a.st_y = fldr_cdiv(a.H, K * TH);
    a.n_st = a.tiles_x * a.st_y;
    const int64_t total = (int64_t)nprob * a.N * (a.groups / a.gpw) * a.n_st;
    if (total > (1ll << 30)) return FLDR_E_SHAPE;
    a.total = (int)total;
    a.per_xcd = (a.total + 7) / 8;
    hipLaunchKernelGGL((splat_acc64_kernel<MODE, CB, TW, TH, K, U, QUAD>), dim3(8 * a.per_xcd), dim3(256), LDS, s, a);
    return 0;
}

// Images (<= 3 channels): every channel + the normalisation sum in one 64 x 24 tile (48 KB: three workgroups per CU; rows are
// 256-byte aligned), three tiles per workgroup (72 rows: 2304 / 72 = 32), four source pixels per thread in flight.  Anything wider: groups of 16
// channels (two packed groups), 32 x 8 tiles (35 KB: four workgroups per CU), one pixel per thread in flight.
template <int MODE>
static int sa_launch(SaArgs& a, int nprob, hipStream_t s) {
    if (a.C <= 3) {
        a.groups = 1; a.gpw = 1;
        // runs of four pixels per thread (16-byte accesses, in-register hand-off between neighbouring pixels) where rows and planes allow
        bool quad = g_sa_quad && a.W % 4 == 0;
        for (int k = 0; k < nprob && quad; ++k) {
            const SaProblem& p = a.p[k];
            quad = ((reinterpret_cast<uintptr_t>(p.img) | reinterpret_cast<uintptr_t>(p.flow) | reinterpret_cast<uintptr_t>(p.metric) |
                     reinterpret_cast<uintptr_t>(p.out_f32)) & 15) == 0 && ((p.img_bstride | p.img_cstride | p.flow_bstride) & 3) == 0 && p.out_spk == nullptr;
        }
        if (quad) return sa_launch2<MODE, 3, SA_IMG_TW, SA_IMG_TH, SA_IMG_K, SA_IMGQ_U, true>(a, nprob, s);
        return sa_launch2<MODE, 3, SA_IMG_TW, SA_IMG_TH, SA_IMG_K, SA_IMG_U>(a, nprob, s);
    }
    a.groups = fldr_cdiv(a.C, 16);
    // optionally (test-build hook; off: see g_sa_group_fold) all channel groups of a tile in ONE workgroup: one candidate search, the
    // flow of a source pixel fetched by one workgroup instead of `groups`
    const int64_t tiles = (int64_t)nprob * a.N * fldr_cdiv(a.W, SA_FEAT_TW) * fldr_cdiv(a.H, SA_FEAT_TH * SA_FEAT_K);
    a.gpw = (g_sa_group_fold && tiles >= SA_FOLD_MIN_TILES) ? a.groups : 1;
    return sa_launch2<MODE, 16, SA_FEAT_TW, SA_FEAT_TH, SA_FEAT_K, SA_FEAT_U>(a, nprob, s);
}

extern "C" int fldr_softsplat_acc64(const fldr_splat_acc_desc* d, fldr_stream_t stream) {
    FLDR_CHECK_ARG(d && d->nprob >= 1 && d->nprob <= 2 && d->N > 0 && d->C > 0 && d->H > 0 && d->W > 0 && d->mode >= 0 && d->mode <= 3);
    // flags: bit 0 tables of ws[j] already filled; bit 1 ws[0] holds the pair tables of fldr_splat_bounds_upsampled_pair (always laid
    // out for TWO problems: nprob must be 2, and bit 0 would name a different layout); bit 2 whole-map walk, no tables
    FLDR_CHECK_ARG(!(d->flags & ~7) && (!(d->flags & 2) || (d->nprob == 2 && !(d->flags & 1))));
    if (d->W > 65535 * ST_BW || d->H > 32767 * ST_BH || (int64_t)d->H * d->W * 4 >= (1ll << 32)) return FLDR_E_SHAPE;
    if ((int64_t)fldr_cdiv(d->W, ST_SBX * ST_BW) * fldr_cdiv(d->H, ST_SBY * ST_BH) > 65535) return FLDR_E_SHAPE;
    SaArgs a;
    a.N = d->N; a.C = d->C; a.H = d->H; a.W = d->W; a.groups = 1;
    // maps of at most SA_WHOLE_PIXELS pixels (the coarse pyramid levels) need no tables: flags bit 2 forces that walk for any size
    a.whole = ((d->flags & 4) || (!(d->flags & 3) && (int64_t)d->H * d->W <= SA_WHOLE_PIXELS)) ? 1 : 0;
    a.nsb_x = fldr_cdiv(d->W, ST_SBX * ST_BW);
    a.nsb = a.nsb_x * fldr_cdiv(d->H, ST_SBY * ST_BH);
    hipStream_t s = fldr_s(stream);
    const int64_t HW = (int64_t)d->H * d->W;
    for (int k = 0; k < 2; ++k) {
        const int j = k < d->nprob ? k : 0;
        FLDR_CHECK_ARG(d->img[j] && d->flow[j] && (a.whole || d->ws[(d->flags & 2) ? 0 : j]) && (d->out_f32[j] || d->out_spk[j]));
        FLDR_CHECK_ARG(d->mode != 2 || d->metric[j] != nullptr);
        FLDR_CHECK_ARG(!d->out_spk[j] || d->C > 3);                     // packed output: the 16-channel configuration only
        SaProblem& p = a.p[k];
        p.img = d->img[j]; p.flow = d->flow[j]; p.metric = d->metric[j];
        if (a.whole) {
            p.blk = p.sbt = nullptr;
        } else if (d->flags & 2) {                                             // ws[0]: tables of the 2 N flows of both problems (fldr_splat_bounds_upsampled_pair)
            p.blk = d->ws[0] + (int64_t)k * d->N * a.nsb * ST_SB_BLOCKS * 4;
            p.sbt = d->ws[0] + (int64_t)d->nprob * d->N * a.nsb * ST_SB_BLOCKS * 4 + (int64_t)k * d->N * a.nsb * 4;
        } else {
            p.blk = d->ws[j]; p.sbt = d->ws[j] + (int64_t)d->N * a.nsb * ST_SB_BLOCKS * 4;
        }
        p.out_f32 = d->out_f32[j]; p.out_spk = reinterpret_cast<unsigned char*>(d->out_spk[j]);
        p.img_bstride = d->img_bstride[j];
        p.img_cstride = d->img_cstride[j] ? d->img_cstride[j] : HW;
        p.flow_bstride = d->flow_bstride[j] ? d->flow_bstride[j] : 2 * HW;
        if (k < d->nprob && !a.whole && !(d->flags & 3))
            fldr_splat_bounds_launch(p.flow, p.flow_bstride, const_cast<float*>(p.blk), const_cast<float*>(p.sbt), d->N, d->H, d->W, a.nsb_x, a.nsb, s);
    }
    int e;
    switch (d->mode) {
        case 0: e = sa_launch<0>(a, d->nprob, s); break;
        case 1: e = sa_launch<1>(a, d->nprob, s); break;
        case 2: e = sa_launch<2>(a, d->nprob, s); break;
        default: e = sa_launch<3>(a, d->nprob, s); break;
    }
    if (e) return e;
    FLDR_LAUNCH_RET();
}
