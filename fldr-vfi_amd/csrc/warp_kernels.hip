// Forward splat, backward warp, splat metric, bilinear resize and the fp64 synthesis tail.
// All HBM-bound gather/scatter/elementwise work: one thread per pixel, channels looped in registers so
// that the per-pixel geometry (corner indices, bilinear weights, mask) is computed once; a wave covers
// 64 consecutive x of one row, so loads are 256-B coalesced and the float atomics of the splat go out as
// (near-)contiguous 256-B wave-instructions, the shape the gfx950 memory-side atomic units run at full rate.
#include "common.h"

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
// LDS-binned scatter.  A workgroup owns a TY x 64 SOURCE tile and a group of CB accumulator channels.
// Flows are locally coherent (they are bilinear upsamplings of a coarse field), so almost all targets of a
// tile fall into a small window around the tile's displaced position: the window [oy,oy+WH) x [ox,ox+WW)
// (origin = minimum target corner of the tile) lives in LDS, contributions are accumulated there with
// ds_add_f32, and the window is then flushed with ONE global float atomic per touched cell, issued as
// contiguous 256-B wave-instructions.  Targets outside the window (diverging flow) go straight to global
// atomics, so the result is exact for any flow.  Versus one global atomic per (source, corner) this cuts
// the memory-side atomic traffic ~3-4x, which is what bounds the splat (MI355X: ~1.3 TB/s of atomic bytes).
template <int MODE, int CB, int TY, int WH, int WW>
__global__ __launch_bounds__(256) void splat_tile_kernel(const float* __restrict__ in, const float* __restrict__ flow,
                                                         const float* __restrict__ metric, float* __restrict__ acc,
                                                         int C, int H, int W, int groups) {
    constexpr int PPT = TY / 4;                         // source pixels per thread (rows ly, ly+4, ...)
    __shared__ float win[CB][WH * WW];
    __shared__ int smin[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = blockIdx.z / groups, grp = blockIdx.z % groups;
    const int x = blockIdx.x * 64 + lane;
    const int64_t HW = (int64_t)H * W;
    const int CA = (MODE >= 1) ? C + 1 : C;             // accumulator channels
    const float* fl = flow + (int64_t)n * 2 * HW;

    for (int i = tid; i < CB * WH * WW; i += 256) (&win[0][0])[i] = 0.0f;

    SplatGeom g[PPT];
    int mnx = 0x7fffffff, mny = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const int y = blockIdx.y * TY + wv + 4 * q;
        g[q].vnw = g[q].vne = g[q].vsw = g[q].vse = false;
        if (x < W && y < H) {
            const int64_t pix = (int64_t)y * W + x;
            g[q] = splat_geom(x, y, fl[pix], fl[HW + pix], W, H);
            if (g[q].vnw | g[q].vne | g[q].vsw | g[q].vse) {
                mnx = min(mnx, max(g[q].x0, 0));
                mny = min(mny, max(g[q].y0, 0));
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mnx = min(mnx, __shfl_xor(mnx, off));
        mny = min(mny, __shfl_xor(mny, off));
    }
    if (lane == 0) { smin[0][wv] = mnx; smin[1][wv] = mny; }
    __syncthreads();
    const int ox = min(min(smin[0][0], smin[0][1]), min(smin[0][2], smin[0][3]));
    const int oy = min(min(smin[1][0], smin[1][1]), min(smin[1][2], smin[1][3]));
    if (ox == 0x7fffffff) return;                       // nothing of this tile lands inside the image

    float* ap = acc + (int64_t)n * CA * HW;
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        if (!(g[q].vnw | g[q].vne | g[q].vsw | g[q].vse)) continue;
        const int y = blockIdx.y * TY + wv + 4 * q;
        const int64_t pix = (int64_t)y * W + x;
        float wgt = 1.0f;
        if (MODE == 2) wgt = metric[(int64_t)n * HW + pix];
        if (MODE == 3 && metric != nullptr) wgt = expf(metric[(int64_t)n * HW + pix]);
        const int wx = g[q].x0 - ox, wy = g[q].y0 - oy;                 // window coords of the NW corner (>= -1)
        const bool in_win = wx >= 0 && wy >= 0 && wx + 1 < WW && wy + 1 < WH;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const int c = grp * CB + cb;
            if (c >= CA) break;
            float v;
            if (c < C) {
                v = in[((int64_t)n * C + c) * HW + pix];
                if (MODE == 3) v = (v + 1.0f) / 2.0f;                     // softSplat.py:334
                if (MODE >= 2) v = v * wgt;                               // :328 / :338
            } else {
                v = wgt;                                                  // normalisation channel
            }
            if (in_win) {
                float* wp = &win[cb][wy * WW + wx];
                if (g[q].vnw) atomicAdd(wp, v * g[q].wnw);
                if (g[q].vne) atomicAdd(wp + 1, v * g[q].wne);
                if (g[q].vsw) atomicAdd(wp + WW, v * g[q].wsw);
                if (g[q].vse) atomicAdd(wp + WW + 1, v * g[q].wse);
            } else {
                splat_add(ap + (int64_t)c * HW, g[q], W, v);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < WH * WW; i += 256) {
        const int gy = oy + i / WW, gx = ox + i % WW;
        if (gy >= H || gx >= W) continue;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const int c = grp * CB + cb;
            if (c >= CA) break;
            const float v = win[cb][i];
            if (v != 0.0f) atomicAdd(ap + (int64_t)c * HW + (int64_t)gy * W + gx, v);
        }
    }
}

template <int MODE>
static void splat_launch(const float* in, const float* flow, const float* metric, float* acc, int N, int C, int H, int W,
                         hipStream_t s) {
    const int CA = (MODE >= 1) ? C + 1 : C;
    if (CA <= 4) {        // images: 4 accumulator channels per workgroup, 16x64 source tile, 32x96 window (48 KiB)
        dim3 grid(fldr_cdiv(W, 64), fldr_cdiv(H, 16), N);
        hipLaunchKernelGGL((splat_tile_kernel<MODE, 4, 16, 32, 96>), grid, dim3(256), 0, s, in, flow, metric, acc, C, H, W, 1);
    } else {              // feature maps: channel groups of 7 (49 = 7 x 7), 8x64 tile, 20x96 window (52.5 KiB)
        const int groups = fldr_cdiv(CA, 7);
        dim3 grid(fldr_cdiv(W, 64), fldr_cdiv(H, 8), N * groups);
        hipLaunchKernelGGL((splat_tile_kernel<MODE, 7, 8, 20, 96>), grid, dim3(256), 0, s, in, flow, metric, acc, C, H, W, groups);
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
                               W, H, inv_wm1, inv_hm1);
    float m = withmask ? fldr_tap_mask(tp) : 1.0f;
    const float* xp = x + (int64_t)n * C * HW;
    float* op = out + (int64_t)n * C * HW + pix;
    if (xs_mode == 0) {
        for (int c = 0; c < C; ++c) op[(int64_t)c * HW] = fldr_tap_sample(tp, xp + (int64_t)c * HW, W) * m;
    } else {
        for (int c = 0; c < C; ++c) {
            const float* p = xp + (int64_t)c * HW + (int64_t)tp.y0 * W + tp.x0;
            float v = 0.0f;
            if (tp.vnw) v += (p[0] * xs) * tp.wnw;
            if (tp.vne) v += (p[1] * xs) * tp.wne;
            if (tp.vsw) v += (p[W] * xs) * tp.wsw;
            if (tp.vse) v += (p[W + 1] * xs) * tp.wse;
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
    FldrTap t = fldr_grid_tap((float)px, (float)py, fl[pix], fl[HW + pix], W, H, inv_wm1, inv_hm1);
    float m = fldr_tap_mask(t);
    const float* sp = self_img + (int64_t)n * C * HW + pix;
    const float* op = other + (int64_t)n * C * HW;
    float acc = 0.0f;
    for (int c = 0; c < C; ++c) {
        float wv = fldr_tap_sample(t, op + (int64_t)c * HW, W) * m;
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
                                                         const float* __restrict__ tv, double T, OUT* __restrict__ out,
                                                         int64_t HW) {
#pragma clang fp contract(off)
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int n = blockIdx.y;
    if (i >= HW) return;
    const float* r = refine + (int64_t)n * 6 * HW + i;
    double s[6], mx = -1.0e300;
    for (int k = 0; k < 6; ++k) { s[k] = (double)r[(int64_t)k * HW] / T; mx = s[k] > mx ? s[k] : mx; }
    double sum = 0.0;
    for (int k = 0; k < 6; ++k) { s[k] = exp(s[k] - mx); sum += s[k]; }
    const float t = tv[n];
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
    dim3 grid(fldr_cdiv(HW, 256), N);
    if (out_f64) hipLaunchKernelGGL(synth_tail_kernel<double>, grid, dim3(256), 0, fldr_s(stream), refine, a, t, T_param, out_f64, HW);
    else         hipLaunchKernelGGL(synth_tail_kernel<float>, grid, dim3(256), 0, fldr_s(stream), refine, a, t, T_param, out_f32, HW);
    FLDR_LAUNCH_RET();
}
