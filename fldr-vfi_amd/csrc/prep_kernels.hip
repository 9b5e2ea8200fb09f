// Level-0 synthesis inputs in ONE pass over the frame (fLDRnet.py:400-479): the x8 bilinear flow upsampling (:419-422),
// the splat metrics z0 / z1 (:442-446), the t-scaled flows flow_t0 / flow_t1 (:404-405), the backward flows flowback_0 /
// flowback_1 (:474-475) and the backward-warped frames im0_tot / im1_tot (:478-479).
//
// The unfused path runs 3 resizes, 2 zmetric, 2 bwarp_tscaled and 2 bwarp kernels and moves ~60 full-resolution planes
// through HBM; here the two upsampled flows are never materialised (every full-resolution flow value — at the pixel and at
// the 4 taps of each backward warp — is re-evaluated from the cache-resident 288x480 flow), each frame plane is read once
// directly plus through L2-friendly gathers, and only the 16 (+2) planes the consumers need are written.
// Every value is produced by the SAME device functions and operation order as the unfused kernels (fldr_lin_src,
// fldr_grid_tap, fldr_tap_sample, fldr_tap_mask; contraction off), so the results are bit-identical to them.
#include "common.h"

struct PrepArgs {
    const float* flow_lo;          // [N,4,h,w]: flow_10 (x,y), flow_01 (x,y)
    const float* I0; const float* I1;
    int64_t i0_bstride, i1_bstride;   // floats between samples ([3,H,W] blocks are contiguous)
    const float* t;                // [N]
    float* z0; float* z1;          // [N,1,H,W] or null (both or neither)
    float* flow_t0; float* flow_t1; float* flowback_0; float* flowback_1;   // [N,2,H,W]
    float* im0_tot; float* im1_tot;                                         // [N,3,H,W]
    int h, w, H, W;
    float sy, sx, mul, inv_wm1, inv_hm1, za0, za1;
    int withmask;
};

// F.interpolate(bilinear, align_corners=False)(scale * plane)[Y, X] * mul — the arithmetic of resize_bilinear_kernel on
// a low-resolution plane that was first multiplied by `scale` (pre != 0) in fp32, as `t4 * flow_01_lo` does.
__device__ __forceinline__ float prep_up(const float* __restrict__ p, int h, int w, int X, int Y, float sx, float sy, float mul,
                                         int pre, float scale) {
#pragma clang fp contract(off)
    int x0, x1, y0, y1; float lx, ly;
    fldr_lin_src(X, sx, w, x0, x1, lx);
    fldr_lin_src(Y, sy, h, y0, y1, ly);
    float a00 = p[(int64_t)y0 * w + x0], a01 = p[(int64_t)y0 * w + x1];
    float a10 = p[(int64_t)y1 * w + x0], a11 = p[(int64_t)y1 * w + x1];
    if (pre) { a00 = scale * a00; a01 = scale * a01; a10 = scale * a10; a11 = scale * a11; }
    const float wx0 = 1.0f - lx, wy0 = 1.0f - ly;
    const float top = wx0 * a00 + lx * a01;
    const float bot = wx0 * a10 + lx * a11;
    return (wy0 * top + ly * bot) * mul;
}

// bwarp_tscaled of a full-resolution flow field that only exists as its low-resolution source: sample (xs * up(plane)) at
// the tap `tp` with the arithmetic of bwarp_kernel's scaled branch.
__device__ __forceinline__ float prep_sample_up(const FldrTap& tp, const float* __restrict__ p, const PrepArgs& a, float xs) {
#pragma clang fp contract(off)
    const int xa = min(max(tp.x0, 0), a.W - 1), xb = min(max(tp.x0 + 1, 0), a.W - 1);
    const int ya = min(max(tp.y0, 0), a.H - 1), yb = min(max(tp.y0 + 1, 0), a.H - 1);
    const float pnw = prep_up(p, a.h, a.w, xa, ya, a.sx, a.sy, a.mul, 0, 1.0f), pne = prep_up(p, a.h, a.w, xb, ya, a.sx, a.sy, a.mul, 0, 1.0f);
    const float psw = prep_up(p, a.h, a.w, xa, yb, a.sx, a.sy, a.mul, 0, 1.0f), pse = prep_up(p, a.h, a.w, xb, yb, a.sx, a.sy, a.mul, 0, 1.0f);
    float v = 0.0f;
    v += tp.vnw ? (pnw * xs) * tp.wnw : 0.0f;
    v += tp.vne ? (pne * xs) * tp.wne : 0.0f;
    v += tp.vsw ? (psw * xs) * tp.wsw : 0.0f;
    v += tp.vse ? (pse * xs) * tp.wse : 0.0f;
    return v;
}

__global__ __launch_bounds__(256) void level0_prep_kernel(PrepArgs a) {
#pragma clang fp contract(off)
    const int px = blockIdx.x * 64 + (threadIdx.x & 63);
    const int py = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int n = blockIdx.z;
    if (px >= a.W || py >= a.H) return;
    const int64_t HW = (int64_t)a.H * a.W, hw = (int64_t)a.h * a.w;
    const int64_t pix = (int64_t)py * a.W + px;
    const float* lo = a.flow_lo + (int64_t)n * 4 * hw;
    const float* l10x = lo, *l10y = lo + hw, *l01x = lo + 2 * hw, *l01y = lo + 3 * hw;
    const float* i0 = a.I0 + (int64_t)n * a.i0_bstride;
    const float* i1 = a.I1 + (int64_t)n * a.i1_bstride;
    const float tv = a.t[n], omt = 1.0f - tv;

    // the frames at this pixel (direct reads, issued first)
    float c0[3], c1[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { c0[c] = i0[(int64_t)c * HW + pix]; c1[c] = i1[(int64_t)c * HW + pix]; }

    // upsampled flows at this pixel (fLDRnet.py:419-422)
    const float f10x = prep_up(l10x, a.h, a.w, px, py, a.sx, a.sy, a.mul, 0, 1.0f), f10y = prep_up(l10y, a.h, a.w, px, py, a.sx, a.sy, a.mul, 0, 1.0f);
    const float f01x = prep_up(l01x, a.h, a.w, px, py, a.sx, a.sy, a.mul, 0, 1.0f), f01y = prep_up(l01y, a.h, a.w, px, py, a.sx, a.sy, a.mul, 0, 1.0f);

    // splat metrics (fLDRnet.py:442-446 = zmetric_kernel): z0 from I0 and bwarp(I1, flow_01); z1 from I1 and bwarp(I0, flow_10)
    if (a.z0) {
        const FldrTap t0 = fldr_grid_tap((float)px, (float)py, f01x, f01y, a.W, a.H, a.inv_wm1, a.inv_hm1);
        const FldrTap t1 = fldr_grid_tap((float)px, (float)py, f10x, f10y, a.W, a.H, a.inv_wm1, a.inv_hm1);
        const float m0 = fldr_tap_mask(t0), m1 = fldr_tap_mask(t1);
        float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float w0 = fldr_tap_sample(t0, i1 + (int64_t)c * HW, a.W, a.H) * m0;
            const float w1 = fldr_tap_sample(t1, i0 + (int64_t)c * HW, a.W, a.H) * m1;
            acc0 += a.za0 * fabsf(c0[c] - w0);
            acc1 += a.za1 * fabsf(c1[c] - w1);
        }
        a.z0[(int64_t)n * HW + pix] = acc0 / 3.0f;
        a.z1[(int64_t)n * HW + pix] = acc1 / 3.0f;
    }

    // t-scaled forward flows (fLDRnet.py:404-405,419-422): upsampling of (t * flow_01_lo) and ((1-t) * flow_10_lo)
    const int64_t o2 = (int64_t)n * 2 * HW + pix;
    a.flow_t0[o2] = prep_up(l01x, a.h, a.w, px, py, a.sx, a.sy, a.mul, 1, tv);
    a.flow_t0[o2 + HW] = prep_up(l01y, a.h, a.w, px, py, a.sx, a.sy, a.mul, 1, tv);
    a.flow_t1[o2] = prep_up(l10x, a.h, a.w, px, py, a.sx, a.sy, a.mul, 1, omt);
    a.flow_t1[o2 + HW] = prep_up(l10y, a.h, a.w, px, py, a.sx, a.sy, a.mul, 1, omt);

    // backward flows (fLDRnet.py:474-475 = bwarp_kernel with scales): flowback_0 = bwarp(t * flow_10, (1-t) * flow_01),
    // flowback_1 = bwarp((1-t) * flow_01, t * flow_10)
    const FldrTap tb0 = fldr_grid_tap((float)px, (float)py, omt * f01x, omt * f01y, a.W, a.H, a.inv_wm1, a.inv_hm1);
    const FldrTap tb1 = fldr_grid_tap((float)px, (float)py, tv * f10x, tv * f10y, a.W, a.H, a.inv_wm1, a.inv_hm1);
    const float mb0 = a.withmask ? fldr_tap_mask(tb0) : 1.0f, mb1 = a.withmask ? fldr_tap_mask(tb1) : 1.0f;
    const float fb0x = prep_sample_up(tb0, l10x, a, tv) * mb0, fb0y = prep_sample_up(tb0, l10y, a, tv) * mb0;
    const float fb1x = prep_sample_up(tb1, l01x, a, omt) * mb1, fb1y = prep_sample_up(tb1, l01y, a, omt) * mb1;
    a.flowback_0[o2] = fb0x; a.flowback_0[o2 + HW] = fb0y;
    a.flowback_1[o2] = fb1x; a.flowback_1[o2 + HW] = fb1y;

    // backward-warped frames (fLDRnet.py:478-479 = bwarp_kernel)
    const FldrTap ti0 = fldr_grid_tap((float)px, (float)py, fb0x, fb0y, a.W, a.H, a.inv_wm1, a.inv_hm1);
    const FldrTap ti1 = fldr_grid_tap((float)px, (float)py, fb1x, fb1y, a.W, a.H, a.inv_wm1, a.inv_hm1);
    const float mi0 = a.withmask ? fldr_tap_mask(ti0) : 1.0f, mi1 = a.withmask ? fldr_tap_mask(ti1) : 1.0f;
    const int64_t o3 = (int64_t)n * 3 * HW + pix;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        a.im0_tot[o3 + (int64_t)c * HW] = fldr_tap_sample(ti0, i0 + (int64_t)c * HW, a.W, a.H) * mi0;
        a.im1_tot[o3 + (int64_t)c * HW] = fldr_tap_sample(ti1, i1 + (int64_t)c * HW, a.W, a.H) * mi1;
    }
}

extern "C" int fldr_level0_prep(const fldr_prep_desc* d, fldr_stream_t stream) {
    FLDR_CHECK_ARG(d && d->flow_lo && d->I0 && d->I1 && d->t && d->flow_t0 && d->flow_t1 && d->flowback_0 && d->flowback_1);
    FLDR_CHECK_ARG(d->im0_tot && d->im1_tot && (!d->z0 == !d->z1) && d->N > 0 && d->h > 0 && d->w > 0 && d->H > 0 && d->W > 0);
    PrepArgs a;
    a.flow_lo = d->flow_lo; a.I0 = d->I0; a.I1 = d->I1; a.i0_bstride = d->i0_bstride; a.i1_bstride = d->i1_bstride;
    a.t = d->t; a.z0 = d->z0; a.z1 = d->z1; a.flow_t0 = d->flow_t0; a.flow_t1 = d->flow_t1;
    a.flowback_0 = d->flowback_0; a.flowback_1 = d->flowback_1; a.im0_tot = d->im0_tot; a.im1_tot = d->im1_tot;
    a.h = d->h; a.w = d->w; a.H = d->H; a.W = d->W;
    a.sy = (float)d->h / (float)d->H; a.sx = (float)d->w / (float)d->W; a.mul = d->mul;
    a.inv_wm1 = (float)(d->W - 1 > 1 ? d->W - 1 : 1); a.inv_hm1 = (float)(d->H - 1 > 1 ? d->H - 1 : 1);
    a.za0 = d->z_alpha0; a.za1 = d->z_alpha1; a.withmask = d->withmask;
    dim3 grid(fldr_cdiv(d->W, 64), fldr_cdiv(d->H, 4), d->N);
    hipLaunchKernelGGL(level0_prep_kernel, grid, dim3(256), 0, fldr_s(stream), a);
    FLDR_LAUNCH_RET();
}
