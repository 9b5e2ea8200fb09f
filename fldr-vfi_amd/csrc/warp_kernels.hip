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
template <int MODE>
__global__ __launch_bounds__(256) void splat_scatter_kernel(const float* __restrict__ in, const float* __restrict__ flow,
                                                            const float* __restrict__ metric, float* __restrict__ acc,
                                                            int C, int H, int W) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    int n = blockIdx.z;
    if (x >= W || y >= H) return;
    const int64_t HW = (int64_t)H * W;
    const int64_t pix = (int64_t)y * W + x;
    const float* fl = flow + (int64_t)n * 2 * HW;
    SplatGeom g = splat_geom(x, y, fl[pix], fl[HW + pix], W, H);
    if (!(g.vnw | g.vne | g.vsw | g.vse)) return;
    const int CA = (MODE >= 1) ? C + 1 : C;           // accumulator channels
    const float* ip = in + (int64_t)n * C * HW + pix;
    float* ap = acc + (int64_t)n * CA * HW;
    float wgt = 1.0f;
    if (MODE == 2) wgt = metric[(int64_t)n * HW + pix];
    if (MODE == 3 && metric != nullptr) wgt = expf(metric[(int64_t)n * HW + pix]);
    for (int c = 0; c < C; ++c) {
        float v = ip[(int64_t)c * HW];
        if (MODE == 3) v = (v + 1.0f) / 2.0f;          // softSplat.py:334
        if (MODE >= 2) v = v * wgt;                    // :328 / :338
        splat_add(ap + (int64_t)c * HW, g, W, v);
    }
    if (MODE >= 1) splat_add(ap + (int64_t)C * HW, g, W, wgt);
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
    dim3 grid(fldr_cdiv(W, 64), fldr_cdiv(H, 4), N);
    hipLaunchKernelGGL(splat_scatter_kernel<-1>, grid, dim3(256), 0, fldr_s(stream), in, flow, nullptr, out_zeroed, C, H, W);
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
    dim3 grid(fldr_cdiv(W, 64), fldr_cdiv(H, 4), N);
    switch (mode) {
        case 0: hipLaunchKernelGGL(splat_scatter_kernel<0>, grid, dim3(256), 0, fldr_s(stream), img, flow, metric, scratch, C, H, W); break;
        case 1: hipLaunchKernelGGL(splat_scatter_kernel<1>, grid, dim3(256), 0, fldr_s(stream), img, flow, metric, scratch, C, H, W); break;
        case 2: hipLaunchKernelGGL(splat_scatter_kernel<2>, grid, dim3(256), 0, fldr_s(stream), img, flow, metric, scratch, C, H, W); break;
        default: hipLaunchKernelGGL(splat_scatter_kernel<3>, grid, dim3(256), 0, fldr_s(stream), img, flow, metric, scratch, C, H, W); break;
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
