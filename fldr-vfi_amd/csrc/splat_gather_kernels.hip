// Softmax splatting of FEATURE MAPS as a deterministic gather (FunctionSoftsplat, softSplat.py:320-352, on the warped
// features of fLDRnet.py:386-387): no atomics, no accumulator tensor, no memset, no normalisation pass, same sums.
//
// The scatter kernel (warp_kernels.hip) is bound by the memory-side float-atomic rate (49 accumulator channels per source
// pixel; rocprof at 4K: 2 x (42 us scatter + 12 us normalise + 9.5 us memset) at the 288x480 level) and its result depends
// on the order in which atomics land.  Here every thread OWNS one destination pixel and looks for the sources
// whose bilinear footprint covers it:
//   * pass 1 (sg_bounds_kernel): min / max of the flow over every 16x16 SOURCE tile;
//   * pass 2 (sg_gather_kernel): a workgroup = one 16x16 destination tile first lists, in index order, the source tiles whose
//     reach [x0 + fxmin, x1 + fxmax + 1] x [...] meets it (one tile per thread and pass, ballot-compacted), then walks that
//     list with the tile's 256 targets staged in LDS (double buffered); inside
//     a reaching tile a destination pixel d only visits the sources that CAN land on it, sx in [d - 1 - fxmax, d + 1 - fxmin)
//     (clamped to the tile; likewise y): a window of (flow spread within the tile + 2 + margins)^2 sources whatever the flow's
//     magnitude — a few dozen for the smooth flows of video, all of them for pathological ones (still exact).  Each visited
//     source is tested exactly like kernel_Softsplat_updateOutput places it (floor of x + fx; the corner weights
//     (x1 - ox) * (y1 - oy) ... in fp32, contraction off) and, when one of its four corners is d, its C channels are added.
//   The accumulation order per destination is fixed (tiles in index order, sources row-major), so the result is
//   run-to-run identical; it differs from the scatter kernel's only by the order of the fp32 sums (~1e-6 relative).
// Both directions of a level (feat1 by flow_10, feat0 by flow_01) and all samples run in ONE launch pair.
// Output: (acc / norm - 0.5) * 2 with norm 0 -> 1 (no division for 'summation'), fp32 NCHW and / or the split-packed layout.
#include "common.h"

#define SG_T 16                     // tile edge (source tiles of the bounds pass, destination tiles of the gather pass)
#define SG_MAXDIR 2

struct SgArgs {
    const float* img[SG_MAXDIR];    // [N,C,H,W] fp32, channel planes contiguous (stride H*W)
    int64_t img_bstride[SG_MAXDIR]; // floats between samples
    const float* flow[SG_MAXDIR];   // [N,2,H,W]: x plane, y plane
    int64_t flow_bstride[SG_MAXDIR];
    const float* metric[SG_MAXDIR]; // [N,1,H,W] or null (softmax / linear weights)
    float* out_f32[SG_MAXDIR];      // [N,C,H,W] or null
    unsigned char* out_spk[SG_MAXDIR];   // split-packed [N][ceil(C/8)][hi,lo][H*W][8 halves] or null
    float4* bounds;                 // [ndir * N][tiles]: (fxmin, fxmax, fymin, fymax)
    int32_t ndir, N, C, H, W, tiles_x, tiles_y, mode;       // mode: 1 average, 2 linear, 3 softmax (0 summation: no normalisation)
};

__global__ __launch_bounds__(256) void sg_bounds_kernel(SgArgs a) {
    const int lx = threadIdx.x & (SG_T - 1), ly = threadIdx.x / SG_T;
    const int x = blockIdx.x * SG_T + lx, y = blockIdx.y * SG_T + ly;
    const int dn = blockIdx.z, d = dn / a.N, n = dn - d * a.N;
    const int64_t HW = (int64_t)a.H * a.W;
    float xmin = 3.0e38f, xmax = -3.0e38f, ymin = 3.0e38f, ymax = -3.0e38f;
    if (x < a.W && y < a.H) {
        const float* f = a.flow[d] + (int64_t)n * a.flow_bstride[d] + (int64_t)y * a.W + x;
        xmin = xmax = f[0];
        ymin = ymax = f[HW];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        xmin = fminf(xmin, __shfl_xor(xmin, off)); xmax = fmaxf(xmax, __shfl_xor(xmax, off));
        ymin = fminf(ymin, __shfl_xor(ymin, off)); ymax = fmaxf(ymax, __shfl_xor(ymax, off));
    }
    __shared__ float red[4][4];
    if ((threadIdx.x & 63) == 0) { float* r = red[threadIdx.x >> 6]; r[0] = xmin; r[1] = xmax; r[2] = ymin; r[3] = ymax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) { xmin = fminf(xmin, red[i][0]); xmax = fmaxf(xmax, red[i][1]); ymin = fminf(ymin, red[i][2]); ymax = fmaxf(ymax, red[i][3]); }
        a.bounds[(int64_t)dn * a.tiles_x * a.tiles_y + blockIdx.y * a.tiles_x + blockIdx.x] = make_float4(xmin, xmax, ymin, ymax);
    }
}

#define SG_MAXTILES 4096             // source tiles per map (host-checked): capacity of the reach list
#define SG_SLOTS 6                   // matches recorded per scan round
#define SG_R 32                      // fast path: edge of the staged source region (16 destination pixels + flow spread + margins)

template <int CT>
__global__ __launch_bounds__(256) void sg_gather_kernel(SgArgs a) {
#pragma clang fp contract(off)
    __shared__ unsigned short s_list[SG_MAXTILES];                   // source tiles that reach this destination tile, ascending
    __shared__ int s_wcnt[4];
    __shared__ float s_ox[2][SG_T * SG_T], s_oy[2][SG_T * SG_T], s_wg[2][SG_T * SG_T];   // targets / weights of the staged source tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lx = tid & (SG_T - 1), ly = tid / SG_T;
    const int dx0 = blockIdx.x * SG_T, dy0 = blockIdx.y * SG_T;
    const int dx = dx0 + lx, dy = dy0 + ly;
    const int dn = blockIdx.z, d = dn / a.N, n = dn - d * a.N;
    const int H = a.H, W = a.W, C = a.C;
    const int64_t HW = (int64_t)H * W;
    const float* img = a.img[d] + (int64_t)n * a.img_bstride[d];
    const float* fxp = a.flow[d] + (int64_t)n * a.flow_bstride[d];
    const float* fyp = fxp + HW;
    const float* mtp = (a.mode >= 2 && a.metric[d]) ? a.metric[d] + (int64_t)n * HW : nullptr;
    const int ntiles = a.tiles_x * a.tiles_y;
    const float4* bnd = a.bounds + (int64_t)dn * ntiles;
    const float fdx0 = (float)dx0, fdx1 = (float)(min(dx0 + SG_T, W) - 1), fdy0 = (float)dy0, fdy1 = (float)(min(dy0 + SG_T, H) - 1);
    const bool live = dx < W && dy < H;

    // ---- 1. which source tiles can land on this destination tile?  256 tiles per pass, one per thread; the list keeps the
    // tile order (ballot + per-wave offsets), so the accumulation order below is fixed.  fp32 addition is monotone, hence the
    // tile-level sums (float)x0 + fxmin ... bound every per-pixel (float)sx + fx exactly. ----
    int cnt = 0;
    for (int base = 0; base < ntiles; base += 256) {
        const int t = base + tid;
        bool reach = false;
        if (t < ntiles) {
            const float4 b = bnd[t];
            const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
            const int sx0 = tx * SG_T, sy0 = ty * SG_T;
            const int sx1 = min(sx0 + SG_T, W) - 1, sy1 = min(sy0 + SG_T, H) - 1;
            reach = floorf((float)sx0 + b.x) <= fdx1 && floorf((float)sx1 + b.y) + 1.0f >= fdx0 &&
                    floorf((float)sy0 + b.z) <= fdy1 && floorf((float)sy1 + b.w) + 1.0f >= fdy0;
        }
        const unsigned long long m = __ballot(reach);
        if (lane == 0) s_wcnt[wave] = __popcll(m);
        __syncthreads();
        int off = cnt;
        for (int w = 0; w < wave; ++w) off += s_wcnt[w];
        if (reach) s_list[off + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)t;
        cnt += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        __syncthreads();
    }

    float acc[CT], norm = 0.0f;
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = 0.0f;
    // the channels of one matching source (exact placement test done by the caller)
    auto add_source = [&](int sx, int sy, float w, float wgt) {
        const float* s = img + (int64_t)sy * W + sx;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            if (c < C) {
                float v = s[(int64_t)c * HW];
                if (a.mode == 3) v = (v + 1.0f) / 2.0f;                     // softSplat.py:334
                if (a.mode >= 2) v = v * wgt;                               // :328 / :338
                acc[c] += v * w;
            }
        }
        norm += wgt * w;
    };
    // does the source with target (ox, oy) land on (dx, dy), and with which corner weight?  (kernel_Softsplat_updateOutput:
    // floor, the four corners, (x1 - fx) * (y1 - fy) ... in fp32)
    auto corner = [&](float ox, float oy, float& w) -> bool {
        const float xf = floorf(ox), yf = floorf(oy);
        const int ix = (int)fminf(fmaxf(xf, -2.0f), (float)W + 1.0f), iy = (int)fminf(fmaxf(yf, -2.0f), (float)H + 1.0f);
        const bool west = ix == dx, east = ix + 1 == dx, north = iy == dy, south = iy + 1 == dy;
        const float x1 = xf + 1.0f, y1 = yf + 1.0f;
        const float wxv = west ? (x1 - ox) : (ox - xf);                      // softSplat.py:35-38
        const float wyv = north ? (y1 - oy) : (oy - yf);
        w = wxv * wyv;
        return (west || east) && (north || south);
    };

    // ---- 2. union of the flow bounds over the reaching tiles -> the source REGION that can land on this destination tile ----
    __shared__ float s_red[4][4];
    {
        float xmin = 3.0e38f, xmax = -3.0e38f, ymin = 3.0e38f, ymax = -3.0e38f;
        for (int li = tid; li < cnt; li += 256) {
            const float4 b = bnd[s_list[li]];
            xmin = fminf(xmin, b.x); xmax = fmaxf(xmax, b.y); ymin = fminf(ymin, b.z); ymax = fmaxf(ymax, b.w);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            xmin = fminf(xmin, __shfl_xor(xmin, off)); xmax = fmaxf(xmax, __shfl_xor(xmax, off));
            ymin = fminf(ymin, __shfl_xor(ymin, off)); ymax = fmaxf(ymax, __shfl_xor(ymax, off));
        }
        if (lane == 0) { s_red[wave][0] = xmin; s_red[wave][1] = xmax; s_red[wave][2] = ymin; s_red[wave][3] = ymax; }
        __syncthreads();
    }
    const float uxmin = fminf(fminf(s_red[0][0], s_red[1][0]), fminf(s_red[2][0], s_red[3][0]));
    const float uxmax = fmaxf(fmaxf(s_red[0][1], s_red[1][1]), fmaxf(s_red[2][1], s_red[3][1]));
    const float uymin = fminf(fminf(s_red[0][2], s_red[1][2]), fminf(s_red[2][2], s_red[3][2]));
    const float uymax = fmaxf(fmaxf(s_red[0][3], s_red[1][3]), fmaxf(s_red[2][3], s_red[3][3]));
    // sources that can reach column X: sx + fx in [X - 1, X + 1), fx in [uxmin, uxmax]  ->  sx in [X - 1 - uxmax, X + 1 - uxmin);
    // one pixel of margin on either side absorbs the rounding of these bounds (the exact test follows per source)
    const float rxl = floorf(fdx0 - 1.0f - uxmax) - 1.0f, rxh = ceilf(fdx1 + 1.0f - uxmin) + 1.0f;
    const float ryl = floorf(fdy0 - 1.0f - uymax) - 1.0f, ryh = ceilf(fdy1 + 1.0f - uymin) + 1.0f;
    const bool fast = cnt > 0 && rxh - rxl < (float)SG_R && ryh - ryl < (float)SG_R;       // workgroup-uniform (NaN bounds: false)

    if (fast) {
        // ---- 3a. FAST PATH (smooth flows: the region fits SG_R x SG_R): the region's targets and weights staged once; every
        // destination pixel scans the same relative window (so the lanes of a wave find their sources in the same iterations) ----
        __shared__ float r_ox[SG_R * SG_R], r_oy[SG_R * SG_R], r_wg[SG_R * SG_R];
        const int rx0 = (int)rxl, ry0 = (int)ryl;
        for (int e = tid; e < SG_R * SG_R; e += 256) {
            const int ry = e / SG_R, rx = e - ry * SG_R;
            const int sx = rx0 + rx, sy = ry0 + ry;
            const bool in = sx >= 0 && sx < W && sy >= 0 && sy < H && (float)sx <= rxh && (float)sy <= ryh;
            const int64_t sp = in ? (int64_t)sy * W + sx : 0;
            const float fx = fxp[sp], fy = fyp[sp];
            float wg = 1.0f;
            if (mtp) wg = a.mode == 3 ? expf(mtp[sp]) : mtp[sp];
            r_ox[e] = in ? (float)sx + fx : -1.0e30f;                    // out-of-image slots never match
            r_oy[e] = in ? (float)sy + fy : -1.0e30f;
            r_wg[e] = wg;
        }
        __syncthreads();
        if (live) {
            // this pixel's window inside the region, as offsets relative to the tile-level window start (equal for all lanes)
            const int wx0 = max(0, (int)(floorf((float)(dx - 1) - uxmax) - 1.0f) - rx0), wx1 = min(SG_R - 1, (int)(ceilf((float)(dx + 1) - uxmin) + 1.0f) - rx0);
            const int wy0 = max(0, (int)(floorf((float)(dy - 1) - uymax) - 1.0f) - ry0), wy1 = min(SG_R - 1, (int)(ceilf((float)(dy + 1) - uymin) + 1.0f) - ry0);
            // Two phases, so that a wave runs the 48-load channel body once per match SLOT and not once per window position at
            // which any of its lanes has a match (where the flow is not uniform over a wave's 16 x 4 pixels — the coarse levels —
            // the lanes' matches sit at different positions: measured ~33 body executions per wave instead of ~5): first scan
            // the window (LDS only) recording up to SG_SLOTS matches, then add them slot by slot; a pixel with more matches
            // (converging flows) simply takes another round.  The order of additions per pixel is still the scan order.
            const int KX = wx1 - wx0 + 1, KT = KX * (wy1 - wy0 + 1);
            int k = 0;
            while (k < KT) {
                int m_e[SG_SLOTS]; float m_w[SG_SLOTS];
                int nm = 0;
#pragma unroll
                for (int j = 0; j < SG_SLOTS; ++j) { m_e[j] = 0; m_w[j] = 0.0f; }
                for (; k < KT && nm < SG_SLOTS; ++k) {
                    const int ry = wy0 + k / KX, rx = wx0 + k % KX;
                    const int e = ry * SG_R + rx;
                    float w;
                    if (!corner(r_ox[e], r_oy[e], w)) continue;
#pragma unroll
                    for (int j = 0; j < SG_SLOTS; ++j) { const bool h = nm == j; m_e[j] = h ? e : m_e[j]; m_w[j] = h ? w : m_w[j]; }
                    ++nm;
                }
#pragma unroll
                for (int j = 0; j < SG_SLOTS; ++j)
                    if (j < nm) { const int e = m_e[j]; add_source(rx0 + e % SG_R, ry0 + e / SG_R, m_w[j], r_wg[e]); }
            }
        }
    } else {
    // ---- 3b. GENERAL PATH (any flow): walk the reaching tiles; a tile's 256 targets and weights are staged in LDS (double
    // buffered: tile i+1 is requested while tile i is scanned), then every destination pixel scans its window of the tile ----
    float n_ox = 0.0f, n_oy = 0.0f, n_wg = 1.0f;
    float4 n_b = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    auto request = [&](int li) {
        const int t = s_list[li];
        const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
        const int sx = min(tx * SG_T + lx, W - 1), sy = min(ty * SG_T + ly, H - 1);     // clamped: out-of-image slots are never visited
        const int64_t sp = (int64_t)sy * W + sx;
        n_ox = (float)sx + fxp[sp];                                  // softSplat.py:23-24
        n_oy = (float)sy + fyp[sp];
        n_wg = 1.0f;
        if (mtp) n_wg = a.mode == 3 ? expf(mtp[sp]) : mtp[sp];
        n_b = bnd[t];
    };
    if (cnt > 0) request(0);
    for (int li = 0; li < cnt; ++li) {
        const int buf = li & 1;
        s_ox[buf][tid] = n_ox; s_oy[buf][tid] = n_oy; s_wg[buf][tid] = n_wg;
        const float4 b = n_b;
        const int t = s_list[li];
        __syncthreads();                                             // staged tile visible; the other buffer's readers (li-1) are done
        if (li + 1 < cnt) request(li + 1);
        if (!live) continue;
        const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
        const int sx0 = tx * SG_T, sy0 = ty * SG_T;
        const int sx1 = min(sx0 + SG_T, W) - 1, sy1 = min(sy0 + SG_T, H) - 1;
        const int wx0 = max(sx0, (int)fmaxf(floorf((float)(dx - 1) - b.y) - 1.0f, -1.0e9f)), wx1 = min(sx1, (int)fminf(ceilf((float)(dx + 1) - b.x) + 1.0f, 1.0e9f));
        const int wy0 = max(sy0, (int)fmaxf(floorf((float)(dy - 1) - b.w) - 1.0f, -1.0e9f)), wy1 = min(sy1, (int)fminf(ceilf((float)(dy + 1) - b.z) + 1.0f, 1.0e9f));
        const int KX = wx1 - wx0 + 1, KT = KX > 0 ? KX * (wy1 - wy0 + 1) : 0;
        int k = 0;
        while (k < KT) {                                             // the two phases of the fast path
            int m_e[SG_SLOTS]; float m_w[SG_SLOTS];
            int nm = 0;
#pragma unroll
            for (int j = 0; j < SG_SLOTS; ++j) { m_e[j] = 0; m_w[j] = 0.0f; }
            for (; k < KT && nm < SG_SLOTS; ++k) {
                const int sy = wy0 + k / KX, sx = wx0 + k % KX;
                const int si = (sy - sy0) * SG_T + (sx - sx0);
                float w;
                if (!corner(s_ox[buf][si], s_oy[buf][si], w)) continue;
#pragma unroll
                for (int j = 0; j < SG_SLOTS; ++j) { const bool h = nm == j; m_e[j] = h ? si : m_e[j]; m_w[j] = h ? w : m_w[j]; }
                ++nm;
            }
#pragma unroll
            for (int j = 0; j < SG_SLOTS; ++j)
                if (j < nm) { const int si = m_e[j]; add_source(sx0 + si % SG_T, sy0 + si / SG_T, m_w[j], s_wg[buf][si]); }
        }
    }
    }
    if (!live) return;
    // ---- normalise, emit (softSplat.py:343-349) ----
    const int64_t pix = (int64_t)dy * W + dx;
    const bool normalise = a.mode >= 1;
    if (norm == 0.0f) norm = 1.0f;
    float* o32 = a.out_f32[d] ? a.out_f32[d] + (int64_t)n * C * HW + pix : nullptr;
    unsigned char* osp = a.out_spk[d] ? a.out_spk[d] + (int64_t)n * ((C + 7) >> 3) * 2 * HW * 16 : nullptr;
    bool bad = false;
#pragma unroll
    for (int g = 0; g < (CT + 7) / 8; ++g) {
        if (g * 8 >= C) break;
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        h8 hi, lo;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = g * 8 + k;
            float v = 0.0f;
            if (c < CT && c < C) {
                v = acc[c < CT ? c : 0];
                if (normalise) v = v / norm;
                v = (v - 0.5f) * 2.0f;                               // softSplat.py:352 (every mode)
                if (o32) o32[(int64_t)c * HW] = v;
            }
            _Float16 h_, l_;
            fldr_split_hl(v, h_, l_, bad);
            hi[k] = h_; lo[k] = l_;
        }
        if (osp) {
            unsigned char* q = osp + ((int64_t)g * 2 * HW + pix) * 16;
            *reinterpret_cast<h8*>(q) = hi;
            *reinterpret_cast<h8*>(q + HW * 16) = lo;
        }
    }
    if (osp) fldr_note_range(bad);
}

FLDR_TU_STATUS(gather)

extern "C" int64_t fldr_softsplat_gather_ws_floats(int ndir, int N, int H, int W) {
    if (ndir <= 0 || ndir > SG_MAXDIR || N <= 0 || H <= 0 || W <= 0) return FLDR_E_ARG;
    return (int64_t)ndir * N * fldr_cdiv(W, SG_T) * fldr_cdiv(H, SG_T) * 4;
}

extern "C" int fldr_softsplat_gather(const fldr_splat_gather_desc* d, fldr_stream_t stream) {
    FLDR_CHECK_ARG(d && d->ws && d->ndir >= 1 && d->ndir <= SG_MAXDIR && d->N > 0 && d->C > 0 && d->C <= 48 && d->H > 0 && d->W > 0);
    FLDR_CHECK_ARG(d->mode >= 0 && d->mode <= 3);
    // every destination tile walks every source tile: meant for feature maps (<= 288 x 512 at 4K), not for frames
    if ((int64_t)fldr_cdiv(d->W, SG_T) * fldr_cdiv(d->H, SG_T) > SG_MAXTILES) return FLDR_E_SHAPE;
    SgArgs a;
    for (int k = 0; k < SG_MAXDIR; ++k) {
        const bool liv = k < d->ndir;
        if (liv) { FLDR_CHECK_ARG(d->img[k] && d->flow[k] && (d->out_f32[k] || d->out_spk[k])); }
        a.img[k] = liv ? d->img[k] : nullptr; a.img_bstride[k] = liv ? d->img_bstride[k] : 0;
        a.flow[k] = liv ? d->flow[k] : nullptr; a.flow_bstride[k] = liv ? d->flow_bstride[k] : 0;
        a.metric[k] = liv ? d->metric[k] : nullptr;
        a.out_f32[k] = liv ? d->out_f32[k] : nullptr;
        a.out_spk[k] = liv ? reinterpret_cast<unsigned char*>(d->out_spk[k]) : nullptr;
    }
    a.bounds = reinterpret_cast<float4*>(d->ws);
    a.ndir = d->ndir; a.N = d->N; a.C = d->C; a.H = d->H; a.W = d->W; a.mode = d->mode;
    a.tiles_x = fldr_cdiv(d->W, SG_T); a.tiles_y = fldr_cdiv(d->H, SG_T);
    hipStream_t s = fldr_s(stream);
    dim3 grid(a.tiles_x, a.tiles_y, d->ndir * d->N);
    hipLaunchKernelGGL(sg_bounds_kernel, grid, dim3(256), 0, s, a);
    if (d->C <= 4)       hipLaunchKernelGGL(sg_gather_kernel<4>, grid, dim3(256), 0, s, a);
    else if (d->C <= 16) hipLaunchKernelGGL(sg_gather_kernel<16>, grid, dim3(256), 0, s, a);
    else                 hipLaunchKernelGGL(sg_gather_kernel<48>, grid, dim3(256), 0, s, a);
    FLDR_LAUNCH_RET();
}
