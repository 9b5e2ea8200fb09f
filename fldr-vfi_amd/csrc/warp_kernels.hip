// Forward splat, backward warp, splat metric, bilinear resize and the fp64 synthesis tail.
// All HBM-bound gather/scatter/elementwise work: one thread per pixel, channels looped in registers so
// that the per-pixel geometry (corner indices, bilinear weights, mask) is computed once; a wave covers
// 64 consecutive x of one row, so loads are 256-B coalesced and the float atomics of the splat go out as
// (near-)contiguous 256-B wave-instructions, the shape the gfx950 memory-side atomic units run at full rate.
#include "common.h"
#include "splat_common.h"

// ------------------------------------------------------------------------------------------------
// softmax splatting (softSplat.py)
// ------------------------------------------------------------------------------------------------
struct SplatGeom {
    int   x0, y0;
    float wnw, wne, wsw, wse;
    bool  vnw, vne, vsw, vse;
};

__device__ __forceinline__ SplatGeom splat_geom(int x, int y, float fx, float fy, int W, int H) {
#pragma clang fp contract(off)
    SplatGeom g;
    float ox = (float)x + fx;                 // softSplat.py:23-24
    float oy = (float)y + fy;
    float xf = floorf(ox), yf = floorf(oy);
    float x1 = xf + 1.0f, y1 = yf + 1.0f;
    g.wnw = (x1 - ox) * (y1 - oy);            // softSplat.py:35-38
    g.wne = (ox - xf) * (y1 - oy);
    g.wsw = (x1 - ox) * (oy - yf);
    g.wse = (ox - xf) * (oy - yf);
    xf = fminf(fmaxf(xf, -2.0f), (float)W + 1.0f);
    yf = fminf(fmaxf(yf, -2.0f), (float)H + 1.0f);
    g.x0 = (int)xf; g.y0 = (int)yf;
    bool x0v = g.x0 >= 0 && g.x0 < W, x1v = g.x0 + 1 >= 0 && g.x0 + 1 < W;
    bool y0v = g.y0 >= 0 && g.y0 < H, y1v = g.y0 + 1 >= 0 && g.y0 + 1 < H;
    g.vnw = x0v && y0v; g.vne = x1v && y0v; g.vsw = x0v && y1v; g.vse = x1v && y1v;
    return g;
}

__device__ __forceinline__ void splat_add(float* __restrict__ plane, const SplatGeom& g, int W, float v) {
    float* p = plane + (int64_t)g.y0 * W + g.x0;
    if (g.vnw) atomicAdd(p, v * g.wnw);
    if (g.vne) atomicAdd(p + 1, v * g.wne);
    if (g.vsw) atomicAdd(p + W, v * g.wsw);
    if (g.vse) atomicAdd(p + W + 1, v * g.wse);
}

// mode: -1 raw (in has C channels, out has C channels); 0 summation; 1 average; 2 linear; 3 softmax
//
// Strip scatter with in-register merging.  The splat is bound by the memory-side float-atomic rate
// (MI355X: ~1.3 TB/s of atomic bytes), so the kernel minimises atomic requests instead of issuing one per
// (source, corner): a wave walks DOWN a 64-pixel-wide strip, R rows per wave, one source pixel per lane.
//   * horizontal merge: where lane l+1 lands exactly one cell to the right of lane l (the normal case for
//     a locally smooth flow), lane l hands its right-hand corner column (NE,SE) to lane l+1 with a wave
//     shuffle, which adds it to its own left-hand column (NW,SW): same cells;
//   * vertical merge: the bottom cell of the previous row is kept pending in registers and, where the
//     current row lands exactly one cell lower, added to the current top cell before ONE atomic is issued.
// Aligned regions cost ~(1+1/R)(1+1/64) atomics per source instead of 4; any lane that is not aligned with
// its neighbour simply emits its own corners, so the result is exact for arbitrary flows.  Every atomic
// wave-instruction addresses (mostly) consecutive cells of one row: the contiguous 256-B shape.
__device__ __forceinline__ void emit_cell(float* __restrict__ plane, int cx, int cy, int W, int H, float v) {
    if (cx >= 0 && cx < W && cy >= 0 && cy < H) atomicAdd(plane + (int64_t)cy * W + cx, v);
}

template <int MODE, int CB, int R>
__global__ __launch_bounds__(256) void splat_strip_kernel(const float* __restrict__ in, const float* __restrict__ flow,
                                                          const float* __restrict__ metric, float* __restrict__ acc,
                                                          int C, int H, int W, int groups) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = blockIdx.z / groups, grp = blockIdx.z % groups;
    const int x = blockIdx.x * 64 + lane;
    const int ybase = (blockIdx.y * 4 + wv) * R;
    if (ybase >= H) return;                                   // wave-uniform
    const int64_t HW = (int64_t)H * W;
    const int CA = (MODE >= 1) ? C + 1 : C;                   // accumulator channels
    const int cbase = grp * CB;
    const float* fl = flow + (int64_t)n * 2 * HW;
    float* ap = acc + ((int64_t)n * CA + cbase) * HW;
    const bool xin = x < W;

    // phase A: issue every load of the strip segment (R rows) before any atomic, for memory-level parallelism
    float fx[R], fy[R], mt[R], val[R][CB];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int y = ybase + r;
        const bool ok = xin && y < H;
        const int64_t pix = ok ? (int64_t)y * W + x : 0;      // clamped: every load below is unconditional
        fx[r] = fl[pix];
        fy[r] = fl[HW + pix];
        mt[r] = 0.0f;
        if ((MODE == 2 || MODE == 3) && metric != nullptr) mt[r] = metric[(int64_t)n * HW + pix];   // wave-uniform test
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const int cc = cbase + c < C ? cbase + c : C - 1;
            val[r][c] = in[((int64_t)n * C + cc) * HW + pix];
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {                             // pin the batch, then mask
        const bool ok = xin && ybase + r < H;
        fldr_pin(fx[r]); fldr_pin(fy[r]); fldr_pin(mt[r]);
        fx[r] = ok ? fx[r] : 0.0f;
        fy[r] = ok ? fy[r] : 0.0f;
        mt[r] = ok ? mt[r] : 0.0f;
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            fldr_pin(val[r][c]);
            val[r][c] = (ok && cbase + c < C) ? val[r][c] : 0.0f;
        }
    }

    bool pend_valid = false;
    int pend_x = 0, pend_y = 0;
    float pend_v[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) pend_v[c] = 0.0f;

#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int y = ybase + r;
        if (y >= H) break;                                    // wave-uniform
        SplatGeom g;
        g.x0 = g.y0 = 0; g.wnw = g.wne = g.wsw = g.wse = 0.0f;
        g.vnw = g.vne = g.vsw = g.vse = false;
        if (xin) g = splat_geom(x, y, fx[r], fy[r], W, H);
        const bool valid = xin && (g.vnw | g.vne | g.vsw | g.vse);
        float wgt = 1.0f;
        if (MODE == 2) wgt = mt[r];
        if (MODE == 3 && metric != nullptr) wgt = expf(mt[r]);
        // neighbour alignment
        const int nx0 = __shfl_down(g.x0, 1), ny0 = __shfl_down(g.y0, 1);
        const int nvalid = __shfl_down((int)valid, 1);
        const bool merge_right = valid && lane < 63 && nvalid && nx0 == g.x0 + 1 && ny0 == g.y0;
        const int left_merges = __shfl_up((int)merge_right, 1);      // executed by ALL lanes (no short-circuit)
        const bool from_left = lane > 0 && left_merges != 0;
        const bool aligned = pend_valid && valid && pend_x == g.x0 && pend_y == g.y0;
        const bool emit_right = valid && !merge_right;
        const bool emit_pend = pend_valid && !aligned;
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            if (cbase + c >= CA) break;                       // wave-uniform
            float v = 0.0f;
            if (valid) {
                if (cbase + c < C) {
                    v = val[r][c];
                    if (MODE == 3) v = (v + 1.0f) / 2.0f;     // softSplat.py:334
                    if (MODE >= 2) v = v * wgt;               // :328 / :338
                } else {
                    v = wgt;                                  // normalisation channel
                }
            }
            const float ne = g.vne ? v * g.wne : 0.0f, se = g.vse ? v * g.wse : 0.0f;
            const float lt = __shfl_up(ne, 1), lb = __shfl_up(se, 1);
            float top = (g.vnw ? v * g.wnw : 0.0f) + (from_left ? lt : 0.0f);
            const float bot = (g.vsw ? v * g.wsw : 0.0f) + (from_left ? lb : 0.0f);
            float* plane = ap + (int64_t)c * HW;
            if (emit_right) {
                emit_cell(plane, g.x0 + 1, g.y0, W, H, ne);
                emit_cell(plane, g.x0 + 1, g.y0 + 1, W, H, se);
            }
            if (emit_pend) emit_cell(plane, pend_x, pend_y, W, H, pend_v[c]);
            if (aligned) top += pend_v[c];
            if (valid) emit_cell(plane, g.x0, g.y0, W, H, top);
            pend_v[c] = bot;
        }
        pend_valid = valid;
        pend_x = g.x0;
        pend_y = g.y0 + 1;
    }
    if (pend_valid) {
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            if (cbase + c >= CA) break;
            emit_cell(ap + (int64_t)c * HW, pend_x, pend_y, W, H, pend_v[c]);
        }
    }
}

template <int MODE>
static void splat_launch(const float* in, const float* flow, const float* metric, float* acc, int N, int C, int H, int W,
                         hipStream_t s) {
    const int CA = (MODE >= 1) ? C + 1 : C;
    if (CA <= 4) {        // images: all (<=4) accumulator channels in one wave, 8 rows per wave
        dim3 grid(fldr_cdiv(W, 64), fldr_cdiv(H, 4 * 8), N);
        hipLaunchKernelGGL((splat_strip_kernel<MODE, 4, 8>), grid, dim3(256), 0, s, in, flow, metric, acc, C, H, W, 1);
    } else {              // feature maps: channel groups of 7 (49 = 7 x 7) for parallelism, 8 rows per wave
        const int groups = fldr_cdiv(CA, 7);
        if ((int64_t)H * W > 40000) {
            dim3 grid(fldr_cdiv(W, 64), fldr_cdiv(H, 4 * 8), N * groups);
            hipLaunchKernelGGL((splat_strip_kernel<MODE, 7, 8>), grid, dim3(256), 0, s, in, flow, metric, acc, C, H, W, groups);
        } else {
            // coarse pyramid levels (<= 144 x 240): a wave's rows are a chain of dependent L2 atomics (measured: 16 us per
            // launch whatever the size with 8 rows per wave; 5.5 us with 2 rows per wave and 4x the waves; at 288 x 480 the
            // extra atomics of the shorter vertical merge chains cost more than the parallelism gives, 47 vs 41 us)
            dim3 grid(fldr_cdiv(W, 64), fldr_cdiv(H, 4 * 2), N * groups);
            hipLaunchKernelGGL((splat_strip_kernel<MODE, 7, 2>), grid, dim3(256), 0, s, in, flow, metric, acc, C, H, W, groups);
        }
    }
}

// out = (acc[c] / norm - 0.5) * 2, norm = acc[C] with 0 -> 1 (softSplat.py:343-349)
template <bool NORMALISE>
__global__ __launch_bounds__(256) void splat_finish_kernel(const float* __restrict__ acc, float* __restrict__ out,
                                                           int C, int64_t HW) {
#pragma clang fp contract(off)
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int n = blockIdx.y;
    if (i >= HW) return;
    const int CA = NORMALISE ? C + 1 : C;
    const float* a = acc + (int64_t)n * CA * HW + i;
    float* o = out + (int64_t)n * C * HW + i;
    float norm = 1.0f;
    if (NORMALISE) { norm = a[(int64_t)C * HW]; if (norm == 0.0f) norm = 1.0f; }
    for (int c = 0; c < C; ++c) {
        float v = a[(int64_t)c * HW];
        if (NORMALISE) v = v / norm;
        o[(int64_t)c * HW] = (v - 0.5f) * 2.0f;
    }
}

// The same normalisation writing the split-packed layout the convolutions consume (conv_spk_kernels.hip) instead of
// fp32 NCHW: thread = one pixel x 8 consecutive channels.
template <bool NORMALISE>
__global__ __launch_bounds__(256) void splat_finish_spk_kernel(const float* __restrict__ acc, unsigned char* __restrict__ out,
                                                               int C, int64_t HW) {
#pragma clang fp contract(off)
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int g = blockIdx.y, n = blockIdx.z;
    if (i >= HW) return;
    const int CA = NORMALISE ? C + 1 : C;
    const int G = (C + 7) >> 3;
    const float* a = acc + (int64_t)n * CA * HW + i;
    float norm = 1.0f;
    if (NORMALISE) { norm = a[(int64_t)C * HW]; if (norm == 0.0f) norm = 1.0f; }
    h8 hi, lo;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = g * 8 + k;
        float v = 0.0f;
        if (c < C) {
            v = a[(int64_t)c * HW];
            if (NORMALISE) v = v / norm;
            v = (v - 0.5f) * 2.0f;
        }
        _Float16 h_, l_;
        fldr_split_hl(v, h_, l_, bad);                                   // the guarded split of common.h
        hi[k] = h_; lo[k] = l_;
    }
    fldr_note_range(bad);
    unsigned char* d = out + (((int64_t)n * G + g) * 2 * HW + i) * 16;
    *reinterpret_cast<h8*>(d) = hi;
    *reinterpret_cast<h8*>(d + HW * 16) = lo;
}

FLDR_TU_STATUS(warp)

extern "C" int fldr_softsplat_fwd(const float* in, const float* flow, float* out_zeroed,
                                  int N, int C, int H, int W, fldr_stream_t stream) {
    FLDR_CHECK_ARG(in && flow && out_zeroed && N > 0 && C > 0 && H > 0 && W > 0);
    splat_launch<-1>(in, flow, nullptr, out_zeroed, N, C, H, W, fldr_s(stream));
    FLDR_LAUNCH_RET();
}

extern "C" int fldr_softsplat_fused(const float* img, const float* flow, const float* metric, float* out, float* scratch,
                                    int N, int C, int H, int W, int mode, fldr_stream_t stream) {
    FLDR_CHECK_ARG(img && flow && out && scratch && N > 0 && C > 0 && H > 0 && W > 0 && mode >= 0 && mode <= 3);
    FLDR_CHECK_ARG(mode != 2 || metric != nullptr);
    const int64_t HW = (int64_t)H * W;
    const int CA = mode >= 1 ? C + 1 : C;
    hipError_t e = hipMemsetAsync(scratch, 0, sizeof(float) * (size_t)N * CA * HW, fldr_s(stream));
    if (e != hipSuccess) return (int)e;
    switch (mode) {
        case 0: splat_launch<0>(img, flow, metric, scratch, N, C, H, W, fldr_s(stream)); break;
        case 1: splat_launch<1>(img, flow, metric, scratch, N, C, H, W, fldr_s(stream)); break;
        case 2: splat_launch<2>(img, flow, metric, scratch, N, C, H, W, fldr_s(stream)); break;
        default: splat_launch<3>(img, flow, metric, scratch, N, C, H, W, fldr_s(stream)); break;
    }
    dim3 g2(fldr_cdiv(HW, 256), N);
    if (mode == 0) hipLaunchKernelGGL(splat_finish_kernel<false>, g2, dim3(256), 0, fldr_s(stream), scratch, out, C, HW);
    else           hipLaunchKernelGGL(splat_finish_kernel<true>, g2, dim3(256), 0, fldr_s(stream), scratch, out, C, HW);
    FLDR_LAUNCH_RET();
}

// fldr_softsplat_fused with a split-packed result (no fp32 tensor is written): the warped features feed conv_flow1 only.
extern "C" int fldr_softsplat_fused_spk(const float* img, const float* flow, const float* metric, void* out_spk, float* scratch,
                                        int N, int C, int H, int W, int mode, fldr_stream_t stream) {
    FLDR_CHECK_ARG(img && flow && out_spk && scratch && N > 0 && C > 0 && H > 0 && W > 0 && mode >= 0 && mode <= 3);
    FLDR_CHECK_ARG(mode != 2 || metric != nullptr);
    const int64_t HW = (int64_t)H * W;
    const int CA = mode >= 1 ? C + 1 : C;
    hipError_t e = hipMemsetAsync(scratch, 0, sizeof(float) * (size_t)N * CA * HW, fldr_s(stream));
    if (e != hipSuccess) return (int)e;
    switch (mode) {
        case 0: splat_launch<0>(img, flow, metric, scratch, N, C, H, W, fldr_s(stream)); break;
        case 1: splat_launch<1>(img, flow, metric, scratch, N, C, H, W, fldr_s(stream)); break;
        case 2: splat_launch<2>(img, flow, metric, scratch, N, C, H, W, fldr_s(stream)); break;
        default: splat_launch<3>(img, flow, metric, scratch, N, C, H, W, fldr_s(stream)); break;
    }
    dim3 g2(fldr_cdiv(HW, 256), (C + 7) / 8, N);
    unsigned char* o = reinterpret_cast<unsigned char*>(out_spk);
    if (mode == 0) hipLaunchKernelGGL(splat_finish_spk_kernel<false>, g2, dim3(256), 0, fldr_s(stream), scratch, o, C, HW);
    else           hipLaunchKernelGGL(splat_finish_spk_kernel<true>, g2, dim3(256), 0, fldr_s(stream), scratch, o, C, HW);
    FLDR_LAUNCH_RET();
}

// The two feature splats of a pyramid level (fLDRnet.py:386-387: feat1 by flow_10, feat0 by flow_01; one sample each) with
// ONE memset and ONE normalisation launch: scratch holds both accumulators back to back, out_spk both packed results as a
// batch of two (sample 0 = the first problem) — what the batched conv_flow1 launch consumes.  Same kernels, same results as two
// fldr_softsplat_fused_spk calls; 4 launches instead of 6.
extern "C" int fldr_softsplat_pair_spk(const float* img_a, const float* flow_a, const float* img_b, const float* flow_b,
                                       void* out_spk, float* scratch, int C, int H, int W, int mode, fldr_stream_t stream) {
    FLDR_CHECK_ARG(img_a && flow_a && img_b && flow_b && out_spk && scratch && C > 0 && H > 0 && W > 0 && mode >= 0 && mode <= 3 && mode != 2);
    const int64_t HW = (int64_t)H * W;
    const int CA = mode >= 1 ? C + 1 : C;
    hipError_t e = hipMemsetAsync(scratch, 0, sizeof(float) * (size_t)2 * CA * HW, fldr_s(stream));
    if (e != hipSuccess) return (int)e;
    for (int k = 0; k < 2; ++k) {
        const float* img = k ? img_b : img_a;
        const float* flow = k ? flow_b : flow_a;
        float* acc = scratch + (int64_t)k * CA * HW;
        switch (mode) {
            case 0: splat_launch<0>(img, flow, nullptr, acc, 1, C, H, W, fldr_s(stream)); break;
            case 1: splat_launch<1>(img, flow, nullptr, acc, 1, C, H, W, fldr_s(stream)); break;
            default: splat_launch<3>(img, flow, nullptr, acc, 1, C, H, W, fldr_s(stream)); break;
        }
    }
    dim3 g2(fldr_cdiv(HW, 256), (C + 7) / 8, 2);
    unsigned char* o = reinterpret_cast<unsigned char*>(out_spk);
    if (mode == 0) hipLaunchKernelGGL(splat_finish_spk_kernel<false>, g2, dim3(256), 0, fldr_s(stream), scratch, o, C, HW);
    else           hipLaunchKernelGGL(splat_finish_spk_kernel<true>, g2, dim3(256), 0, fldr_s(stream), scratch, o, C, HW);
    FLDR_LAUNCH_RET();
}

// ------------------------------------------------------------------------------------------------
// backward warp (fLDRnet.py:546-581) and the splat metric built on it (fLDRnet.py:442-446)
// ------------------------------------------------------------------------------------------------
// xs_mode / fs_mode: 0 -> 1, 1 -> t[n], 2 -> (1 - t[n]); used for flowback = bwarp(t*flow_10, (1-t)*flow_01)
// (fLDRnet.py:474-475) without materialising the scaled flows.  The scale is applied to the fetched values
// before the bilinear weights, which is the rounding order of scaling the tensors first.
__device__ __forceinline__ float tscale(int mode, const float* __restrict__ t, int n) {
#pragma clang fp contract(off)
    return mode == 0 ? 1.0f : (mode == 1 ? t[n] : 1.0f - t[n]);
}

__global__ __launch_bounds__(256) void bwarp_kernel(const float* __restrict__ x, const float* __restrict__ flo,
                                                    float* __restrict__ out, int C, int H, int W, int withmask,
                                                    float inv_wm1, float inv_hm1, const float* __restrict__ t,
                                                    int xs_mode, int fs_mode) {
#pragma clang fp contract(off)
    int px = blockIdx.x * 64 + (threadIdx.x & 63);
    int py = blockIdx.y * 4 + (threadIdx.x >> 6);
    int n = blockIdx.z;
    if (px >= W || py >= H) return;
    const int64_t HW = (int64_t)H * W;
    const int64_t pix = (int64_t)py * W + px;
    const float* fl = flo + (int64_t)n * 2 * HW;
    const float xs = tscale(xs_mode, t, n), fs = tscale(fs_mode, t, n);
    FldrTap tp = fldr_grid_tap((float)px, (float)py, fs_mode ? fs * fl[pix] : fl[pix], fs_mode ? fs * fl[HW + pix] : fl[HW + pix],
                               W, H, inv_wm1, inv_hm1, 1.0f / inv_wm1, 1.0f / inv_hm1);
    float m = withmask ? fldr_tap_mask(tp) : 1.0f;
    const float* xp = x + (int64_t)n * C * HW;
    float* op = out + (int64_t)n * C * HW + pix;
    if (xs_mode == 0) {
        for (int c = 0; c < C; ++c) op[(int64_t)c * HW] = fldr_tap_sample(tp, xp + (int64_t)c * HW, W, H) * m;
    } else {
        for (int c = 0; c < C; ++c) {
            const float* pl = xp + (int64_t)c * HW;
            const int xa = min(max(tp.x0, 0), W - 1), xb = min(max(tp.x0 + 1, 0), W - 1);
            const int ya = min(max(tp.y0, 0), H - 1), yb = min(max(tp.y0 + 1, 0), H - 1);
            float pnw = pl[(int64_t)ya * W + xa], pne = pl[(int64_t)ya * W + xb];
            float psw = pl[(int64_t)yb * W + xa], pse = pl[(int64_t)yb * W + xb];
            fldr_pin(pnw); fldr_pin(pne); fldr_pin(psw); fldr_pin(pse);
            float v = 0.0f;
            v += tp.vnw ? (pnw * xs) * tp.wnw : 0.0f;
            v += tp.vne ? (pne * xs) * tp.wne : 0.0f;
            v += tp.vsw ? (psw * xs) * tp.wsw : 0.0f;
            v += tp.vse ? (pse * xs) * tp.wse : 0.0f;
            op[(int64_t)c * HW] = v * m;
        }
    }
}

extern "C" int fldr_bwarp(const float* x, const float* flo, float* out, int N, int C, int H, int W, int withmask,
                          fldr_stream_t stream) {
    FLDR_CHECK_ARG(x && flo && out && N > 0 && C > 0 && H > 0 && W > 0);
    dim3 grid(fldr_cdiv(W, 64), fldr_cdiv(H, 4), N);
    float iw = (float)(W - 1 > 1 ? W - 1 : 1), ih = (float)(H - 1 > 1 ? H - 1 : 1);
    hipLaunchKernelGGL(bwarp_kernel, grid, dim3(256), 0, fldr_s(stream), x, flo, out, C, H, W, withmask, iw, ih,
                       (const float*)nullptr, 0, 0);
    FLDR_LAUNCH_RET();
}

extern "C" int fldr_bwarp_tscaled(const float* x, const float* flo, float* out, const float* t, int x_mode, int flo_mode,
                                  int N, int C, int H, int W, int withmask, fldr_stream_t stream) {
    FLDR_CHECK_ARG(x && flo && out && t && N > 0 && C > 0 && H > 0 && W > 0);
    FLDR_CHECK_ARG(x_mode >= 0 && x_mode <= 2 && flo_mode >= 0 && flo_mode <= 2);
    dim3 grid(fldr_cdiv(W, 64), fldr_cdiv(H, 4), N);
    float iw = (float)(W - 1 > 1 ? W - 1 : 1), ih = (float)(H - 1 > 1 ? H - 1 : 1);
    hipLaunchKernelGGL(bwarp_kernel, grid, dim3(256), 0, fldr_s(stream), x, flo, out, C, H, W, withmask, iw, ih, t, x_mode, flo_mode);
    FLDR_LAUNCH_RET();
}

__global__ __launch_bounds__(256) void zmetric_kernel(const float* __restrict__ self_img, const float* __restrict__ other,
                                                      const float* __restrict__ flow, float alpha, float* __restrict__ z,
                                                      int C, int H, int W, float inv_wm1, float inv_hm1) {
#pragma clang fp contract(off)
    int px = blockIdx.x * 64 + (threadIdx.x & 63);
    int py = blockIdx.y * 4 + (threadIdx.x >> 6);
    int n = blockIdx.z;
    if (px >= W || py >= H) return;
    const int64_t HW = (int64_t)H * W;
    const int64_t pix = (int64_t)py * W + px;
    const float* fl = flow + (int64_t)n * 2 * HW;
    FldrTap t = fldr_grid_tap((float)px, (float)py, fl[pix], fl[HW + pix], W, H, inv_wm1, inv_hm1, 1.0f / inv_wm1, 1.0f / inv_hm1);
    float m = fldr_tap_mask(t);
    const float* sp = self_img + (int64_t)n * C * HW + pix;
    const float* op = other + (int64_t)n * C * HW;
    float acc = 0.0f;
    for (int c = 0; c < C; ++c) {
        float wv = fldr_tap_sample(t, op + (int64_t)c * HW, W, H) * m;
        acc += alpha * fabsf(sp[(int64_t)c * HW] - wv);
    }
    z[(int64_t)n * HW + pix] = acc / (float)C;
}

extern "C" int fldr_zmetric(const float* self_img, const float* other_img, const float* flow, float alpha, float* z,
                            int N, int C, int H, int W, fldr_stream_t stream) {
    FLDR_CHECK_ARG(self_img && other_img && flow && z && N > 0 && C > 0 && H > 0 && W > 0);
    dim3 grid(fldr_cdiv(W, 64), fldr_cdiv(H, 4), N);
    float iw = (float)(W - 1 > 1 ? W - 1 : 1), ih = (float)(H - 1 > 1 ? H - 1 : 1);
    hipLaunchKernelGGL(zmetric_kernel, grid, dim3(256), 0, fldr_s(stream), self_img, other_img, flow, alpha, z, C, H, W, iw, ih);
    FLDR_LAUNCH_RET();
}

// ------------------------------------------------------------------------------------------------
// F.interpolate(bilinear, align_corners=False) * mul   (fLDRnet.py:384-385, 419-422)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              int h, int w, int H, int W, float sy, float sx, float mul) {
#pragma clang fp contract(off)
    int X = blockIdx.x * 64 + (threadIdx.x & 63);
    int Y = blockIdx.y * 4 + (threadIdx.x >> 6);
    int c = blockIdx.z;
    if (X >= W || Y >= H) return;
    int x0, x1, y0, y1; float lx, ly;
    fldr_lin_src(X, sx, w, x0, x1, lx);
    fldr_lin_src(Y, sy, h, y0, y1, ly);
    const float* p = in + (int64_t)c * h * w;
    float wx0 = 1.0f - lx, wy0 = 1.0f - ly;
    float top = wx0 * p[(int64_t)y0 * w + x0] + lx * p[(int64_t)y0 * w + x1];
    float bot = wx0 * p[(int64_t)y1 * w + x0] + lx * p[(int64_t)y1 * w + x1];
    out[((int64_t)c * H + Y) * W + X] = (wy0 * top + ly * bot) * mul;
}

// The same resize for a tensor of C <= 8 channels that also emits its split-packed twin (one group): the upsampled flow of
// fLDRnet.py:384-385 is consumed as fp32 (splats, residual) AND as the third packed source of conv_flow2.0.
__device__ __forceinline__ void resize_spk_pixel(const float* __restrict__ in, float* __restrict__ out, unsigned char* __restrict__ spk,
                                                 int C, int h, int w, int H, int W, float sy, float sx, float mul, int n, int X, int Y) {
#pragma clang fp contract(off)
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    int x0, x1, y0, y1; float lx, ly;
    fldr_lin_src(X, sx, w, x0, x1, lx);
    fldr_lin_src(Y, sy, h, y0, y1, ly);
    const float wx0 = 1.0f - lx, wy0 = 1.0f - ly;
    const int64_t HW = (int64_t)H * W, pix = (int64_t)Y * W + X;
    h8 hi, lo;
    bool bad = false;
    float vs[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float v = 0.0f;
        if (c < C) {
            const float* p = in + ((int64_t)n * C + c) * h * w;
            const float top = wx0 * p[(int64_t)y0 * w + x0] + lx * p[(int64_t)y0 * w + x1];
            const float bot = wx0 * p[(int64_t)y1 * w + x0] + lx * p[(int64_t)y1 * w + x1];
            v = (wy0 * top + ly * bot) * mul;
            out[((int64_t)n * C + c) * HW + pix] = v;
        }
        vs[c] = v;
    }
    {
        _Float16 hs[8], ls[8];
        fldr_split_hl_group(vs, hs, ls, bad);
#pragma unroll
        for (int c = 0; c < 8; ++c) { hi[c] = hs[c]; lo[c] = ls[c]; }
    }
    fldr_note_range(bad);
    unsigned char* d = spk + ((int64_t)n * 2 * HW + pix) * 16;
    *reinterpret_cast<h8*>(d) = hi;
    *reinterpret_cast<h8*>(d + HW * 16) = lo;
}

__global__ __launch_bounds__(256) void resize_bilinear_spk_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                  unsigned char* __restrict__ spk, int C, int h, int w,
                                                                  int H, int W, float sy, float sx, float mul) {
    const int X = blockIdx.x * 64 + (threadIdx.x & 63);
    const int Y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (X >= W || Y >= H) return;
    resize_spk_pixel(in, out, spk, C, h, w, H, W, sy, sx, mul, blockIdx.z, X, Y);
}

// The upsampled [N,4,H,W] flow of a pyramid level (fp32 + packed twin, as above) AND the two flow-bounds tables of the feature splats
// that consume it (fldr_splat_bounds_upsampled_pair, pair 2), in ONE launch: both read only the low-resolution flow, and the bounds
// pass was a 10 us launch of 2 N nsb workgroups between the resize and the splat on every level.  The first n_bounds workgroups run
// the bounds body (splat_common.h), the others resize a 64 x 16 pixel tile each with the pixel function of the kernel above —
// the same instructions on the same operands as the two separate launches, so both outputs are bit-identical to them.
__global__ __launch_bounds__(64 * ST_UP_WAVES) void resize_spk_bounds_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                             unsigned char* __restrict__ spk, float* __restrict__ blk,
                                                                             float* __restrict__ sbt, int N, int h, int w, int H, int W,
                                                                             float sy, float sx, float mul, int nsb_x, int nsb,
                                                                             int n_bounds, int tiles_x, int tiles_y) {
    __shared__ float red[ST_UP_WAVES][4];
    int b = blockIdx.x;
    if (b < n_bounds) {                                                 // workgroup-uniform
        splat_bounds_up_body(in, 4 * (int64_t)h * w, nullptr, 0, mul, blk, sbt, h, w, H, W, sy, sx, nsb_x, nsb, 2, N, b % nsb, b / nsb, red);
        return;
    }
    b -= n_bounds;
    const int tx = b % tiles_x, ty = (b / tiles_x) % tiles_y, n = b / (tiles_x * tiles_y);
    const int X = tx * 64 + (threadIdx.x & 63);
    const int Y = ty * ST_UP_WAVES + (threadIdx.x >> 6);
    if (X >= W || Y >= H) return;
    resize_spk_pixel(in, out, spk, 4, h, w, H, W, sy, sx, mul, n, X, Y);
}

extern "C" int fldr_resize_bilinear_spk(const float* in, float* out, void* out_spk, int N, int C, int h, int w, int H, int W,
                                        float mul, fldr_stream_t stream) {
    FLDR_CHECK_ARG(in && out && out_spk && N > 0 && C > 0 && C <= 8 && h > 0 && w > 0 && H > 0 && W > 0);
    dim3 grid(fldr_cdiv(W, 64), fldr_cdiv(H, 4), N);
    hipLaunchKernelGGL(resize_bilinear_spk_kernel, grid, dim3(256), 0, fldr_s(stream), in, out, reinterpret_cast<unsigned char*>(out_spk),
                       C, h, w, H, W, (float)h / (float)H, (float)w / (float)W, mul);
    FLDR_LAUNCH_RET();
}

// fldr_resize_bilinear_spk (C = 4) + fldr_splat_bounds_upsampled_pair (pair 2) of the same low-resolution flow in one launch.
// ws: 2 * fldr_softsplat_tile_ws_floats(N, H, W) floats, laid out as fldr_splat_bounds_upsampled_pair leaves them.  W < 4 w (the
// workgroup-per-super-block bounds body; the model upsamples by 2), else FLDR_E_SHAPE: call the two entry points instead.
extern "C" int fldr_resize_bilinear_spk_bounds(const float* in, float* out, void* out_spk, float* ws, int N, int h, int w, int H, int W,
                                               float mul, fldr_stream_t stream) {
    FLDR_CHECK_ARG(in && out && out_spk && ws && N > 0 && h > 0 && w > 0 && H >= h && W >= w && mul > 0.0f);
    if ((int64_t)W >= 4 * (int64_t)w) return FLDR_E_SHAPE;
    if (W > 65535 * ST_BW || H > 32767 * ST_BH) return FLDR_E_SHAPE;
    const int nsb_x = fldr_cdiv(W, ST_SBX * ST_BW), nsb = nsb_x * fldr_cdiv(H, ST_SBY * ST_BH);
    const int tiles_x = fldr_cdiv(W, 64), tiles_y = fldr_cdiv(H, ST_UP_WAVES);
    const int64_t n_bounds = (int64_t)2 * N * nsb, blocks = n_bounds + (int64_t)N * tiles_x * tiles_y;
    if (blocks > 0x7fffffff) return FLDR_E_SHAPE;
    float* blk = ws;
    float* sbt = ws + (int64_t)2 * N * nsb * ST_SB_BLOCKS * 4;
    hipLaunchKernelGGL(resize_spk_bounds_kernel, dim3((unsigned)blocks), dim3(64 * ST_UP_WAVES), 0, fldr_s(stream), in, out,
                       reinterpret_cast<unsigned char*>(out_spk), blk, sbt, N, h, w, H, W, (float)h / (float)H, (float)w / (float)W, mul,
                       nsb_x, nsb, (int)n_bounds, tiles_x, tiles_y);
    FLDR_LAUNCH_RET();
}

extern "C" int fldr_resize_bilinear(const float* in, float* out, int NC, int h, int w, int H, int W, float mul,
                                    fldr_stream_t stream) {
    FLDR_CHECK_ARG(in && out && NC > 0 && h > 0 && w > 0 && H > 0 && W > 0);
    dim3 grid(fldr_cdiv(W, 64), fldr_cdiv(H, 4), NC);
    hipLaunchKernelGGL(resize_bilinear_kernel, grid, dim3(256), 0, fldr_s(stream), in, out, h, w, H, W,
                       (float)h / (float)H, (float)w / (float)W, mul);
    FLDR_LAUNCH_RET();
}

// ------------------------------------------------------------------------------------------------
// occlusion softmax + synthesis, fp64 (fLDRnet.py:511-524)
// ------------------------------------------------------------------------------------------------
struct SynthArgs {
    const float* cand[6];
    int64_t bstride[6];
};

template <typename OUT>
__global__ __launch_bounds__(256) void synth_tail_kernel(const float* __restrict__ refine, SynthArgs a,
                                                         const float* __restrict__ tv, const float* __restrict__ poison, double T,
                                                         OUT* __restrict__ out, int64_t HW) {
#pragma clang fp contract(off)
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int n = blockIdx.y;
    if (i >= HW) return;
    const float* r = refine + (int64_t)n * 6 * HW + i;
    double s[6], mx = -1.0e300;
    for (int k = 0; k < 6; ++k) { s[k] = (double)r[(int64_t)k * HW] / T; mx = s[k] > mx ? s[k] : mx; }
    double sum = 0.0;
    for (int k = 0; k < 6; ++k) { s[k] = exp(s[k] - mx); sum += s[k]; }
    const float t = tv[n] + *poison;                          // (*poison: 0.0f, NaN once a ring wait expired — common.h)
    const double w1 = (double)t, w0 = (double)(1.0f - t);     // (1 - t_value) is formed in fp32 (fLDRnet.py:517)
    double wo[6];
    for (int k = 0; k < 6; ++k) wo[k] = ((k & 1) ? w1 : w0) * (s[k] / sum);
    double div = ((wo[0] + wo[1]) + wo[2]) + wo[3];            // :517
    div = div + (wo[4] + wo[5]);                               // :522
    for (int c = 0; c < 3; ++c) {
        double v[6];
        for (int k = 0; k < 6; ++k) v[k] = wo[k] * (double)a.cand[k][(int64_t)n * a.bstride[k] + (int64_t)c * HW + i];
        double o = v[0] + v[1];                                // :518
        o = o + (v[2] + v[3]);                                 // :520
        o = o + (v[4] + v[5]);                                 // :521
        out[((int64_t)n * 3 + c) * HW + i] = (OUT)(o / div);   // :524
    }
}

extern "C" int fldr_synth_tail(const float* refine, const float* const cand[6], const int64_t cand_bstride[6],
                               const float* t, double T_param, double* out_f64, float* out_f32,
                               int N, int H, int W, fldr_stream_t stream) {
    FLDR_CHECK_ARG(refine && cand && cand_bstride && t && N > 0 && H > 0 && W > 0);
    FLDR_CHECK_ARG((out_f64 != nullptr) != (out_f32 != nullptr));
    SynthArgs a;
    for (int k = 0; k < 6; ++k) { FLDR_CHECK_ARG(cand[k]); a.cand[k] = cand[k]; a.bstride[k] = cand_bstride[k]; }
    const int64_t HW = (int64_t)H * W;
    const float* poison = fldr_status_poison_ptr();
    if (!poison) return FLDR_E_STATUS;
    dim3 grid(fldr_cdiv(HW, 256), N);
    if (out_f64) hipLaunchKernelGGL(synth_tail_kernel<double>, grid, dim3(256), 0, fldr_s(stream), refine, a, t, poison, T_param, out_f64, HW);
    else         hipLaunchKernelGGL(synth_tail_kernel<float>, grid, dim3(256), 0, fldr_s(stream), refine, a, t, poison, T_param, out_f32, HW);
    FLDR_LAUNCH_RET();
}
