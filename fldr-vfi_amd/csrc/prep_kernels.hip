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
    const float2* flow_lo2;        // [2][N,h,w] x float2: flow_10 (x,y) of every sample, then flow_01 (x,y): one 8-B load per
                                   // low-resolution pixel and flow
    int N;
    const float* I0; const float* I1;
    int64_t i0_bstride, i1_bstride;   // floats between samples
    int64_t i0_cstride, i1_cstride;   // floats between channel planes
    const float* t;                // [N]
    float* z0; float* z1;          // [N,1,H,W] or null (both or neither)
    float* flow_t0; float* flow_t1; float* flowback_0; float* flowback_1;   // [N,2,H,W]
    float* im0_tot; float* im1_tot;                                         // [N,3,H,W]
    int h, w, H, W;
    float sy, sx, mul, inv_wm1, inv_hm1, r_wm1, r_hm1, za0, za1;     // inv_*: the grid normalisation divisors max(S-1,1); r_*: their reciprocals
    int withmask;
    int phase;                     // bit 0: z0 / z1 + flow_t0 / flow_t1; bit 1: flowback_0 / _1 + im0_tot / im1_tot (3 = everything)
    int kx, ky;                    // sx == 2^-kx / sy == 2^-ky exactly (integer source-index arithmetic), else -1
    float rkx, rky;                // 2^-(kx+1), 2^-(ky+1)
};

// Uniform base pointer + 32-bit byte offset: one global_load / global_store with an SGPR base and a VGPR offset, no 64-bit
// address arithmetic per access (the kernel is bound by its VALU instruction count).  Planes are < 4 GB (host-checked).
__device__ __forceinline__ float prep_ldf(const float* __restrict__ base, uint32_t boff) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + boff); }
__device__ __forceinline__ float2 prep_ldf2(const float2* __restrict__ base, uint32_t boff) { return *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(base) + boff); }
// The 16 output planes (566 MB at 4K, read back by the splats / enc1 / dec3 only after hundreds of MB of other traffic) are
// stored with the streaming hint so that they do not displace what the Infinity Cache can actually keep (the packed
// activations the convolutions hand to each other, enc1's output for enc2): +0.6-1.0 % frame pairs/s, A/B on one box.
// The same hint on dec3's loads / stores and on the band splat's stores measured neutral, on the band splat's loads -1.3 %,
// on enc1's stores -3.3 % (enc2 reads them right away).
#ifndef PREP_NT
#define PREP_NT 1
#endif
__device__ __forceinline__ void prep_stf(float* __restrict__ base, uint32_t boff, float v) {
#if PREP_NT
    __builtin_nontemporal_store(v, reinterpret_cast<float*>(reinterpret_cast<char*>(base) + boff));
#else
    *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + boff) = v;
#endif
}

// Source indices / weight of F.interpolate(bilinear, align_corners=False) along one axis (fldr_lin_src), computed once
// and shared by every plane and tap that is evaluated at the same coordinate.
struct PrepLin { int i0, i1; float l; };
// kshift >= 0: scale == 2^-kshift exactly (the model's x8 upsampling: 0.125).  Then r = scale * (o + 0.5) - 0.5 =
// (2 o + 1 - 2^k) / 2^(k+1) is exact in fp32 and fldr_lin_src's floor / fraction are a shift and a mask of the integer
// numerator — the same i0, i1 and l bit for bit, in 6 integer instructions instead of 12 (the kernel is bound by its
// vector-instruction count, and this runs ten times per pixel).
__device__ __forceinline__ PrepLin prep_lin(int o, float scale, int in_size, int kshift, float rk) {
    PrepLin r;
    if (kshift >= 0) {                                                   // uniform
        const int t = 2 * o + 1 - (1 << kshift);                         // numerator of r over 2^(k+1); r < 0 clamps to 0
        const int tc = t < 0 ? 0 : t;
        const int i = tc >> (kshift + 1);
        r.i0 = i < in_size - 1 ? i : in_size - 1;
        r.i1 = r.i0 + (r.i0 < in_size - 1 ? 1 : 0);
        // r - i0: the fraction while i <= in_size - 1 (always, for o inside the upsampled image); the general clamp otherwise
        const float l = (float)(tc - (r.i0 << (kshift + 1))) * rk;        // rk = 2^-(kshift+1) from the host (a reciprocal computed here is a 10-instruction division, ten times per pixel)
        r.l = l > 1.0f ? 1.0f : l;
    } else {
        fldr_lin_src(o, scale, in_size, r.i0, r.i1, r.l);
    }
    return r;
}

// The four low-resolution neighbours of one full-resolution coordinate, all 4 flow channels each (8 loads of 8 B).
struct PrepQuad { float4 a00, a01, a10, a11; };
__device__ __forceinline__ PrepQuad prep_quad(const float2* __restrict__ p10, const float2* __restrict__ p01, int w, const PrepLin& ix,
                                              const PrepLin& iy) {
    PrepQuad q;
    const uint32_t y0 = __umul24((uint32_t)iy.i0, (uint32_t)w), y1 = __umul24((uint32_t)iy.i1, (uint32_t)w);   // full-rate 24-bit multiplies
    const uint32_t o00 = (y0 + (uint32_t)ix.i0) * 8u, o01 = (y0 + (uint32_t)ix.i1) * 8u;
    const uint32_t o10 = (y1 + (uint32_t)ix.i0) * 8u, o11 = (y1 + (uint32_t)ix.i1) * 8u;
    const float2 b00 = prep_ldf2(p10, o00), b01 = prep_ldf2(p10, o01), b10 = prep_ldf2(p10, o10), b11 = prep_ldf2(p10, o11);
    const float2 c00 = prep_ldf2(p01, o00), c01 = prep_ldf2(p01, o01), c10 = prep_ldf2(p01, o10), c11 = prep_ldf2(p01, o11);
    q.a00 = make_float4(b00.x, b00.y, c00.x, c00.y); q.a01 = make_float4(b01.x, b01.y, c01.x, c01.y);
    q.a10 = make_float4(b10.x, b10.y, c10.x, c10.y); q.a11 = make_float4(b11.x, b11.y, c11.x, c11.y);
    return q;
}
__device__ __forceinline__ float prep_ch(const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }

// F.interpolate(bilinear, align_corners=False)(scale * plane)[Y, X] * mul — the arithmetic of resize_bilinear_kernel on
// channel c of the low-resolution flow, first multiplied by `scale` (pre != 0) in fp32 as `t4 * flow_01_lo` does.
__device__ __forceinline__ float prep_up(const PrepQuad& q, int c, const PrepLin& ix, const PrepLin& iy, float mul, int pre, float scale) {
#pragma clang fp contract(off)
    float a00 = prep_ch(q.a00, c), a01 = prep_ch(q.a01, c), a10 = prep_ch(q.a10, c), a11 = prep_ch(q.a11, c);
    if (pre) { a00 = scale * a00; a01 = scale * a01; a10 = scale * a10; a11 = scale * a11; }
    const float wx0 = 1.0f - ix.l, wy0 = 1.0f - iy.l;
    const float top = wx0 * a00 + ix.l * a01;
    const float bot = wx0 * a10 + ix.l * a11;
    return (wy0 * top + iy.l * bot) * mul;
}

// bwarp_tscaled of a full-resolution 2-channel flow field that only exists as its low-resolution source `lo2` (x, y per
// low-resolution pixel): sample (xs * up(channel)) at the tap `tp` with the arithmetic of bwarp_kernel's scaled branch.
// The tap's four corners are adjacent full-resolution pixels, and when upsampling (scale <= 1) adjacent pixels start
// their low-resolution neighbourhoods at most one cell apart: the four 2x2 neighbourhoods lie in ONE 3x3 block, loaded
// once (9 loads of 8 B instead of 16 of 16 B) and picked apart with selects.  (The kernel is bound by its VALU
// instruction count: ~1,100 per pixel after this and the prepared taps of common.h, 1,411 before.)
__device__ __forceinline__ void prep_sample_up2(const FldrTap& tp, const FldrTapP& tpp, const float2* __restrict__ lo2, const PrepArgs& a,
                                                float xs, float& ox, float& oy) {
#pragma clang fp contract(off)
    const int xa = min(max(tp.x0, 0), a.W - 1), xb = min(max(tp.x0 + 1, 0), a.W - 1);
    const int ya = min(max(tp.y0, 0), a.H - 1), yb = min(max(tp.y0 + 1, 0), a.H - 1);
    const PrepLin lxa = prep_lin(xa, a.sx, a.w, a.kx, a.rkx), lxb = prep_lin(xb, a.sx, a.w, a.kx, a.rkx);
    const PrepLin lya = prep_lin(ya, a.sy, a.h, a.ky, a.rky), lyb = prep_lin(yb, a.sy, a.h, a.ky, a.rky);
    // columns lxa.i0 + {0,1,2} and rows lya.i0 + {0,1,2}, clamped like fldr_lin_src's i1: (i0, i1) of xa is columns (0,1),
    // of xb columns (dx, dx+1) with dx = lxb.i0 - lxa.i0 in {0,1}; rows alike
    const int c0 = lxa.i0, c1 = min(c0 + 1, a.w - 1), c2 = min(c0 + 2, a.w - 1);
    const int r0 = lya.i0, r1 = min(r0 + 1, a.h - 1), r2 = min(r0 + 2, a.h - 1);
    const uint32_t q0 = __umul24((uint32_t)r0, (uint32_t)a.w) * 8u, q1 = __umul24((uint32_t)r1, (uint32_t)a.w) * 8u, q2 = __umul24((uint32_t)r2, (uint32_t)a.w) * 8u;
    const uint32_t k0 = (uint32_t)c0 * 8u, k1 = (uint32_t)c1 * 8u, k2 = (uint32_t)c2 * 8u;
    const float2 m[3][3] = {{prep_ldf2(lo2, q0 + k0), prep_ldf2(lo2, q0 + k1), prep_ldf2(lo2, q0 + k2)},
                            {prep_ldf2(lo2, q1 + k0), prep_ldf2(lo2, q1 + k1), prep_ldf2(lo2, q1 + k2)},
                            {prep_ldf2(lo2, q2 + k0), prep_ldf2(lo2, q2 + k1), prep_ldf2(lo2, q2 + k2)}};
    const bool dx = lxb.i0 != lxa.i0, dy = lyb.i0 != lya.i0;
    const float wxa = 1.0f - lxa.l, wxb = 1.0f - lxb.l, wya = 1.0f - lya.l, wyb = 1.0f - lyb.l;
    float o[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        // prep_up's arithmetic per corner — top = wx0 * a00 + lx * a01, bot likewise, (wy0 * top + ly * bot) * mul — with the
        // horizontal interpolations done once per neighbourhood row (west pair of columns for xa, east pair for xb) and the
        // vertical pairs of rows picked afterwards: same operands, same operations, 20 selects instead of 56
        float tw[3], te[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float v0 = k == 0 ? m[r][0].x : m[r][0].y, v1 = k == 0 ? m[r][1].x : m[r][1].y, v2 = k == 0 ? m[r][2].x : m[r][2].y;
            tw[r] = wxa * v0 + lxa.l * v1;
            te[r] = wxb * (dx ? v1 : v0) + lxb.l * (dx ? v2 : v1);
        }
        const float pnw = (wya * tw[0] + lya.l * tw[1]) * a.mul;
        const float pne = (wya * te[0] + lya.l * te[1]) * a.mul;
        const float psw = (wyb * (dy ? tw[1] : tw[0]) + lyb.l * (dy ? tw[2] : tw[1])) * a.mul;
        const float pse = (wyb * (dy ? te[1] : te[0]) + lyb.l * (dy ? te[2] : te[1])) * a.mul;
        float v = 0.0f;
        v += (pnw * xs) * tpp.wnw;
        v += (pne * xs) * tpp.wne;
        v += (psw * xs) * tpp.wsw;
        v += (pse * xs) * tpp.wse;
        o[k] = v;
    }
    ox = o[0]; oy = o[1];
}

// [N,4,h,w] -> [2][N,h,w] x float2
__global__ __launch_bounds__(256) void prep_interleave_kernel(const float* __restrict__ lo, float2* __restrict__ lo2, int64_t hw) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y, N = gridDim.y;
    if (i >= hw) return;
    const float* p = lo + (int64_t)n * 4 * hw + i;
    lo2[(int64_t)n * hw + i] = make_float2(p[0], p[hw]);
    lo2[(int64_t)(N + n) * hw + i] = make_float2(p[2 * hw], p[3 * hw]);
}

// (An output path through an LDS tile — every plane written as 16-byte pieces after one barrier instead of 4-byte lanes straight from
// the pixel's thread — was built and measured in round 2: the kernel alone 316 vs 321 us, but 433 vs 439 pairs/s with three pairs in
// flight; the kernel is bound by its ~1,000 vector instructions per pixel, not by its stores.  Removed.)
#define PREP_NPL 16                // z0, z1, flow_t0 (x, y), flow_t1, flowback_0, flowback_1, im0_tot (3), im1_tot (3)
// PH: the phases this instantiation carries (1: z0 / z1 + flow_t; 2: flowback + im_tot; 3: both) — a template parameter so that a
// one-phase launch has the registers (and with them the waves in flight: the kernel waits on chains of dependent gathers) of its phase
template <int PH>
__global__ __launch_bounds__(256) void level0_prep_kernel(PrepArgs a) {
#pragma clang fp contract(off)
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int px = blockIdx.x * 64 + tx;
    const int py = blockIdx.y * 4 + ty;
    const int n = blockIdx.z;
    const bool live = px < a.W && py < a.H;
    const int64_t HW = (int64_t)a.H * a.W, hw = (int64_t)a.h * a.w;
    constexpr bool ph1 = (PH & 1) != 0, ph2 = (PH & 2) != 0;
    const int64_t o1 = (int64_t)n * HW, o2 = (int64_t)n * 2 * HW, o3 = (int64_t)n * 3 * HW;
    float* const dst[PREP_NPL] = {
        a.z0 ? a.z0 + o1 : nullptr, a.z1 ? a.z1 + o1 : nullptr,
        a.flow_t0 + o2, a.flow_t0 + o2 + HW, a.flow_t1 + o2, a.flow_t1 + o2 + HW,
        a.flowback_0 + o2, a.flowback_0 + o2 + HW, a.flowback_1 + o2, a.flowback_1 + o2 + HW,
        a.im0_tot + o3, a.im0_tot + o3 + HW, a.im0_tot + o3 + 2 * HW, a.im1_tot + o3, a.im1_tot + o3 + HW, a.im1_tot + o3 + 2 * HW};
    const uint32_t pix_off = (__umul24((uint32_t)(live ? py : 0), (uint32_t)a.W) + (uint32_t)(live ? px : 0)) * 4u;
    auto put = [&](int plane, float v) __attribute__((always_inline)) {
        prep_stf(dst[plane], pix_off, v);                               // 4-byte lanes, straight from the pixel's thread
    };
    if (live) {
        const uint32_t pixb = (__umul24((uint32_t)py, (uint32_t)a.W) + (uint32_t)px) * 4u;      // byte offset of this pixel inside a plane
        const float2* lo10 = a.flow_lo2 + (int64_t)n * hw;       // flow_10 (x,y)
        const float2* lo01 = a.flow_lo2 + (int64_t)(a.N + n) * hw;   // flow_01 (x,y); quad channels: 0,1 = flow_10, 2,3 = flow_01
        const float* i0 = a.I0 + (int64_t)n * a.i0_bstride;
        const float* i1 = a.I1 + (int64_t)n * a.i1_bstride;
        const float tv = a.t[n], omt = 1.0f - tv;

        // the frames at this pixel (direct reads, issued first; only the splat metrics use them)
        float c0[3] = {0.0f, 0.0f, 0.0f}, c1[3] = {0.0f, 0.0f, 0.0f};
        if (ph1 && a.z0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { c0[c] = prep_ldf(i0 + (int64_t)c * a.i0_cstride, pixb); c1[c] = prep_ldf(i1 + (int64_t)c * a.i1_cstride, pixb); }
        }

        // upsampled flows at this pixel (fLDRnet.py:419-422)
        const PrepLin lx = prep_lin(px, a.sx, a.w, a.kx, a.rkx), ly = prep_lin(py, a.sy, a.h, a.ky, a.rky);
        const PrepQuad q = prep_quad(lo10, lo01, a.w, lx, ly);
        const float f10x = prep_up(q, 0, lx, ly, a.mul, 0, 1.0f), f10y = prep_up(q, 1, lx, ly, a.mul, 0, 1.0f);
        const float f01x = prep_up(q, 2, lx, ly, a.mul, 0, 1.0f), f01y = prep_up(q, 3, lx, ly, a.mul, 0, 1.0f);

        // splat metrics (fLDRnet.py:442-446 = zmetric_kernel): z0 from I0 and bwarp(I1, flow_01); z1 from I1 and bwarp(I0, flow_10)
        if (ph1 && a.z0) {
            const FldrTapP t0 = fldr_tap_prepare(fldr_grid_tap((float)px, (float)py, f01x, f01y, a.W, a.H, a.inv_wm1, a.inv_hm1, a.r_wm1, a.r_hm1), a.W, a.H);
            const FldrTapP t1 = fldr_tap_prepare(fldr_grid_tap((float)px, (float)py, f10x, f10y, a.W, a.H, a.inv_wm1, a.inv_hm1, a.r_wm1, a.r_hm1), a.W, a.H);
            const float m0 = fldr_tap_mask_p(t0), m1 = fldr_tap_mask_p(t1);
            float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float w0 = fldr_tap_sample_p(t0, i1 + (int64_t)c * a.i1_cstride) * m0;
                const float w1 = fldr_tap_sample_p(t1, i0 + (int64_t)c * a.i0_cstride) * m1;
                acc0 += a.za0 * fabsf(c0[c] - w0);
                acc1 += a.za1 * fabsf(c1[c] - w1);
            }
            put(0, fldr_div_by(acc0, 3.0f, 1.0f / 3.0f));          // == acc0 / 3.0f (the mean over the 3 channels)
            put(1, fldr_div_by(acc1, 3.0f, 1.0f / 3.0f));
        }

        // t-scaled forward flows (fLDRnet.py:404-405,419-422): upsampling of (t * flow_01_lo) and ((1-t) * flow_10_lo)
        if (ph1) {
            put(2, prep_up(q, 2, lx, ly, a.mul, 1, tv));
            put(3, prep_up(q, 3, lx, ly, a.mul, 1, tv));
            put(4, prep_up(q, 0, lx, ly, a.mul, 1, omt));
            put(5, prep_up(q, 1, lx, ly, a.mul, 1, omt));
        }
        if (ph2) {
            // backward flows (fLDRnet.py:474-475 = bwarp_kernel with scales): flowback_0 = bwarp(t * flow_10, (1-t) * flow_01),
            // flowback_1 = bwarp((1-t) * flow_01, t * flow_10)
            const FldrTap tb0 = fldr_grid_tap((float)px, (float)py, omt * f01x, omt * f01y, a.W, a.H, a.inv_wm1, a.inv_hm1, a.r_wm1, a.r_hm1);
            const FldrTap tb1 = fldr_grid_tap((float)px, (float)py, tv * f10x, tv * f10y, a.W, a.H, a.inv_wm1, a.inv_hm1, a.r_wm1, a.r_hm1);
            const FldrTapP tb0p = fldr_tap_prepare(tb0, a.W, a.H), tb1p = fldr_tap_prepare(tb1, a.W, a.H);
            const float mb0 = a.withmask ? fldr_tap_mask_p(tb0p) : 1.0f, mb1 = a.withmask ? fldr_tap_mask_p(tb1p) : 1.0f;
            float fb0x, fb0y, fb1x, fb1y;
            prep_sample_up2(tb0, tb0p, lo10, a, tv, fb0x, fb0y);
            prep_sample_up2(tb1, tb1p, lo01, a, omt, fb1x, fb1y);
            fb0x = fb0x * mb0; fb0y = fb0y * mb0; fb1x = fb1x * mb1; fb1y = fb1y * mb1;
            put(6, fb0x); put(7, fb0y); put(8, fb1x); put(9, fb1y);

            // backward-warped frames (fLDRnet.py:478-479 = bwarp_kernel)
            const FldrTapP ti0 = fldr_tap_prepare(fldr_grid_tap((float)px, (float)py, fb0x, fb0y, a.W, a.H, a.inv_wm1, a.inv_hm1, a.r_wm1, a.r_hm1), a.W, a.H);
            const FldrTapP ti1 = fldr_tap_prepare(fldr_grid_tap((float)px, (float)py, fb1x, fb1y, a.W, a.H, a.inv_wm1, a.inv_hm1, a.r_wm1, a.r_hm1), a.W, a.H);
            const float mi0 = a.withmask ? fldr_tap_mask_p(ti0) : 1.0f, mi1 = a.withmask ? fldr_tap_mask_p(ti1) : 1.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                put(10 + c, fldr_tap_sample_p(ti0, i0 + (int64_t)c * a.i0_cstride) * mi0);
                put(13 + c, fldr_tap_sample_p(ti1, i1 + (int64_t)c * a.i1_cstride) * mi1);
            }
        }
    }
}

#ifdef FLDR_TEST_HOOKS
// ---- LDS-staged gather windows (round 4; measured, not faster on coherent flows: TEST BUILD ONLY, cross-check of the kernel above) ----
// Hypothesis: the kernel above is bound by its 48 four-byte image gathers per pixel (with the lanes of a gather 2 / 4 pixels apart —
// a thread owning a run of 2 / 4 pixels for 16-byte plane loads / stores — it takes 395 / 518 us instead of 328: the time follows
// the cache lines per gather instruction).  So here the gathers leave the vector-memory path: a workgroup owns a 64 x 16 tile; for each
// of its four backward taps it finds the exact bounding box of the tile's corner pixels (packed-int16 min / max reduction over the
// workgroup), stages that box of the three planes — tile + flow SPREAD, not magnitude: (64 + 16) x (16 + 8) — in LDS with 16-byte
// LDS-DMA and samples from LDS; a box that does not fit falls back to global gathers for that tile and tap pair.  Same device
// functions, operands and operation order per value: bit-identical planes (test_level0_prep_lds_windows_bit_identical).
// MEASURED (tools/kernel_bench.py prep, 2304x3840): rigid-shift flows 350 us (global gathers) vs 368-380 us (this kernel, tiles of
// 8 / 16 rows, windows 72 / 80 wide; 32 rows: 471); incoherent flows (low-resolution noise 0.3 px) 495 vs 385 us.  Per phase:
// z + flow_t 148-157 vs 160-169 us, flowback + im_tot 237-241 vs 237-250 us — i.e. on coherent flows the image gathers are NOT
// what bounds the kernel (they hit L1 / L2 lines the direct reads brought in), and staging them costs what it saves.  Kept out of
// the product; the hypothesis stands only for incoherent flows.
#define PL_TW 64
#ifndef PL_TH
#define PL_TH 16
#endif
#ifndef PL_WW
#define PL_WW 80                              // window width (floats): tile + 16
#endif
#ifndef PL_WH
#define PL_WH (PL_TH + 8)                     // window height: tile + 8
#endif
#define PL_PLANE (PL_WW * PL_WH)              // floats per window plane (7,680 B)
#define PL_RPT (PL_TH / 4)                    // rows per thread
#define PL_LDS_BYTES (6 * PL_PLANE * 4 + 256)

typedef __attribute__((address_space(1))) const void* pl_gptr_t;
typedef __attribute__((address_space(3))) void* pl_lptr_t;
typedef short pl_s2 __attribute__((ext_vector_type(2)));

// componentwise min / max of two int16 pairs (x in the low half, y in the high half; coordinates < 32768, host-checked)
__device__ __forceinline__ uint32_t pl_min2(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(pl_s2, a), __builtin_bit_cast(pl_s2, b))); }
__device__ __forceinline__ uint32_t pl_max2(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(pl_s2, a), __builtin_bit_cast(pl_s2, b))); }
__device__ __forceinline__ uint32_t pl_pack(int x, int y) { return ((uint32_t)y << 16) | (uint32_t)(x & 0xffff); }

// the clamped corner box of a tap (fldr_tap_prepare's xa, xb, ya, yb)
struct PlBox { int xa, xb, ya, yb; };
__device__ __forceinline__ PlBox pl_box(const FldrTap& t, int W, int H) {
    PlBox b;
    b.xa = min(max(t.x0, 0), W - 1); b.xb = min(max(t.x0 + 1, 0), W - 1);
    b.ya = min(max(t.y0, 0), H - 1); b.yb = min(max(t.y0 + 1, 0), H - 1);
    return b;
}
// fldr_tap_sample_p on a window: corner (x, y) of the image is cell (y - oy) * PL_WW + (x - ox); same products, same order
__device__ __forceinline__ float pl_sample(const FldrTap& t, const PlBox& b, const float* __restrict__ win, int ox, int oy) {
#pragma clang fp contract(off)
    const float* ra = win + (b.ya - oy) * PL_WW - ox;
    const float* rb = win + (b.yb - oy) * PL_WW - ox;
    const float pnw = ra[b.xa], pne = ra[b.xb], psw = rb[b.xa], pse = rb[b.xb];
    const float wnw = t.vnw ? t.wnw : 0.0f, wne = t.vne ? t.wne : 0.0f, wsw = t.vsw ? t.wsw : 0.0f, wse = t.vse ? t.wse : 0.0f;
    float v = 0.0f;
    v += pnw * wnw;
    v += pne * wne;
    v += psw * wsw;
    v += pse * wse;
    return v;
}

__global__ __launch_bounds__(256) void level0_prep_lds_kernel(PrepArgs a) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) unsigned char pl_smem[];
    float* const win = reinterpret_cast<float*>(pl_smem);                    // [6][PL_PLANE]: planes 0-2 window A, 3-5 window B
    uint32_t* const red_base = reinterpret_cast<uint32_t*>(pl_smem + 6 * PL_PLANE * 4);   // 2 x [4 waves][4] packed corner bounds (one set per phase)
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6, lane = tx;
    const int px = blockIdx.x * PL_TW + tx;
    const int py0 = blockIdx.y * PL_TH + ty;                                 // this thread's rows: py0 + 4 k
    const int n = blockIdx.z;
    const bool xin = px < a.W;
    const int64_t HW = (int64_t)a.H * a.W, hw = (int64_t)a.h * a.w;
    const bool ph1 = (a.phase & 1) != 0, ph2 = (a.phase & 2) != 0;           // uniform
    const int64_t o1 = (int64_t)n * HW, o2 = (int64_t)n * 2 * HW, o3 = (int64_t)n * 3 * HW;
    const float2* lo10 = a.flow_lo2 + (int64_t)n * hw;
    const float2* lo01 = a.flow_lo2 + (int64_t)(a.N + n) * hw;
    const float* i0 = a.I0 + (int64_t)n * a.i0_bstride;
    const float* i1 = a.I1 + (int64_t)n * a.i1_bstride;
    const float tv = a.t[n], omt = 1.0f - tv;
    const bool want_z = ph1 && a.z0;
    const float fpx = (float)px;

    // upsampled flows at this thread's pixels (fLDRnet.py:419-422); out-of-image pixels of a partial tile idle (live[k] false)
    float f10x[PL_RPT], f10y[PL_RPT], f01x[PL_RPT], f01y[PL_RPT];
    bool live[PL_RPT];
    const PrepLin lx = prep_lin(xin ? px : 0, a.sx, a.w, a.kx, a.rkx);
#pragma unroll
    for (int k = 0; k < PL_RPT; ++k) {
        const int py = py0 + 4 * k;
        live[k] = xin && py < a.H;
        const PrepLin ly = prep_lin(live[k] ? py : 0, a.sy, a.h, a.ky, a.rky);
        const PrepQuad q = prep_quad(lo10, lo01, a.w, lx, ly);
        f10x[k] = prep_up(q, 0, lx, ly, a.mul, 0, 1.0f); f10y[k] = prep_up(q, 1, lx, ly, a.mul, 0, 1.0f);
        f01x[k] = prep_up(q, 2, lx, ly, a.mul, 0, 1.0f); f01y[k] = prep_up(q, 3, lx, ly, a.mul, 0, 1.0f);
        if (ph1 && live[k]) {                                               // t-scaled forward flows (fLDRnet.py:404-405,419-422)
            const uint32_t pixb = (__umul24((uint32_t)py, (uint32_t)a.W) + (uint32_t)px) * 4u;
            prep_stf(a.flow_t0 + o2, pixb, prep_up(q, 2, lx, ly, a.mul, 1, tv));
            prep_stf(a.flow_t0 + o2 + HW, pixb, prep_up(q, 3, lx, ly, a.mul, 1, tv));
            prep_stf(a.flow_t1 + o2, pixb, prep_up(q, 0, lx, ly, a.mul, 1, omt));
            prep_stf(a.flow_t1 + o2 + HW, pixb, prep_up(q, 1, lx, ly, a.mul, 1, omt));
        }
    }

    // Workgroup-wide corner box of two taps (A, B): packed (x, y) min of the north-west corners, max of the south-east ones
    // -> window origins (x a multiple of 4: 16-byte chunks) and whether both boxes fit their windows.
    int oxA = 0, oyA = 0, oxB = 0, oyB = 0;
    auto reduce_boxes = [&](uint32_t* red, uint32_t mnA, uint32_t mxA, uint32_t mnB, uint32_t mxB) -> bool {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            mnA = pl_min2(mnA, (uint32_t)__shfl_xor((int)mnA, off)); mxA = pl_max2(mxA, (uint32_t)__shfl_xor((int)mxA, off));
            mnB = pl_min2(mnB, (uint32_t)__shfl_xor((int)mnB, off)); mxB = pl_max2(mxB, (uint32_t)__shfl_xor((int)mxB, off));
        }
        if (lane == 0) { red[ty * 4 + 0] = mnA; red[ty * 4 + 1] = mxA; red[ty * 4 + 2] = mnB; red[ty * 4 + 3] = mxB; }
        __syncthreads();                                                     // (also: every wave is done with the previous windows)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            mnA = pl_min2(mnA, red[w * 4 + 0]); mxA = pl_max2(mxA, red[w * 4 + 1]);
            mnB = pl_min2(mnB, red[w * 4 + 2]); mxB = pl_max2(mxB, red[w * 4 + 3]);
        }
        oxA = (int)(mnA & 0xffffu) & ~3; oyA = (int)(mnA >> 16);
        oxB = (int)(mnB & 0xffffu) & ~3; oyB = (int)(mnB >> 16);
        return (int)(mxA & 0xffffu) - oxA < PL_WW && (int)(mxA >> 16) - oyA < PL_WH &&
               (int)(mxB & 0xffffu) - oxB < PL_WW && (int)(mxB >> 16) - oyB < PL_WH;
    };
    // stage window A <- planes of imgA at (oxA, oyA), window B <- imgB at (oxB, oyB): 16-byte LDS-DMA chunks, clamped into the image
    auto fill_windows = [&](const float* imgA, int64_t csA, const float* imgB, int64_t csB) {
        constexpr int CH = PL_PLANE / 4;                                    // 16-byte chunks per plane (480)
#pragma unroll
        for (int j = 0; j < (CH + 255) / 256; ++j) {
            const int c = tid + 256 * j;
            const int row = c / (PL_WW / 4), col = (c - row * (PL_WW / 4)) * 4;
            if (c < CH) {                                                    // (wave-uniform except in the last, partial wave)
                const uint32_t ga = (uint32_t)(min(oyA + row, a.H - 1) * a.W + min(oxA + col, a.W - 4)) * 4u;
                const uint32_t gb = (uint32_t)(min(oyB + row, a.H - 1) * a.W + min(oxB + col, a.W - 4)) * 4u;
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    __builtin_amdgcn_global_load_lds((pl_gptr_t)(reinterpret_cast<const char*>(imgA + (int64_t)p * csA) + ga),
                                                     (pl_lptr_t)(win + p * PL_PLANE + (c & ~63) * 4), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((pl_gptr_t)(reinterpret_cast<const char*>(imgB + (int64_t)p * csB) + gb),
                                                     (pl_lptr_t)(win + (3 + p) * PL_PLANE + (c & ~63) * 4), 16, 0, 0);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                                  // vmcnt(0): my chunks have landed
        __syncthreads();
    };

    // ---- phase 1: splat metrics (fLDRnet.py:442-446): z0 from I0 and bwarp(I1, flow_01); z1 from I1 and bwarp(I0, flow_10) ----
    if (want_z) {
        uint32_t mnA = 0x7fff7fffu, mxA = 0u, mnB = 0x7fff7fffu, mxB = 0u;   // A: I1 at flow_01, B: I0 at flow_10
#pragma unroll
        for (int k = 0; k < PL_RPT; ++k) {
            if (!live[k]) continue;
            const float fpy = (float)(py0 + 4 * k);
            const PlBox b0 = pl_box(fldr_grid_tap(fpx, fpy, f01x[k], f01y[k], a.W, a.H, a.inv_wm1, a.inv_hm1, a.r_wm1, a.r_hm1), a.W, a.H);
            const PlBox b1 = pl_box(fldr_grid_tap(fpx, fpy, f10x[k], f10y[k], a.W, a.H, a.inv_wm1, a.inv_hm1, a.r_wm1, a.r_hm1), a.W, a.H);
            mnA = pl_min2(mnA, pl_pack(b0.xa, b0.ya)); mxA = pl_max2(mxA, pl_pack(b0.xb, b0.yb));
            mnB = pl_min2(mnB, pl_pack(b1.xa, b1.ya)); mxB = pl_max2(mxB, pl_pack(b1.xb, b1.yb));
        }
        const bool fit = reduce_boxes(red_base, mnA, mxA, mnB, mxB);         // workgroup-uniform
        if (fit) fill_windows(i1, a.i1_cstride, i0, a.i0_cstride);
#pragma unroll
        for (int k = 0; k < PL_RPT; ++k) {
            if (!live[k]) continue;
            const int py = py0 + 4 * k;
            const float fpy = (float)py;
            const uint32_t pixb = (__umul24((uint32_t)py, (uint32_t)a.W) + (uint32_t)px) * 4u;
            const FldrTap t0 = fldr_grid_tap(fpx, fpy, f01x[k], f01y[k], a.W, a.H, a.inv_wm1, a.inv_hm1, a.r_wm1, a.r_hm1);
            const FldrTap t1 = fldr_grid_tap(fpx, fpy, f10x[k], f10y[k], a.W, a.H, a.inv_wm1, a.inv_hm1, a.r_wm1, a.r_hm1);
            const FldrTapP t0p = fldr_tap_prepare(t0, a.W, a.H), t1p = fldr_tap_prepare(t1, a.W, a.H);
            const PlBox b0 = pl_box(t0, a.W, a.H), b1 = pl_box(t1, a.W, a.H);
            const float m0 = fldr_tap_mask_p(t0p), m1 = fldr_tap_mask_p(t1p);
            float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float c0 = prep_ldf(i0 + (int64_t)c * a.i0_cstride, pixb), c1 = prep_ldf(i1 + (int64_t)c * a.i1_cstride, pixb);
                const float w0 = (fit ? pl_sample(t0, b0, win + c * PL_PLANE, oxA, oyA) : fldr_tap_sample_p(t0p, i1 + (int64_t)c * a.i1_cstride)) * m0;
                const float w1 = (fit ? pl_sample(t1, b1, win + (3 + c) * PL_PLANE, oxB, oyB) : fldr_tap_sample_p(t1p, i0 + (int64_t)c * a.i0_cstride)) * m1;
                acc0 += a.za0 * fabsf(c0 - w0);
                acc1 += a.za1 * fabsf(c1 - w1);
            }
            prep_stf(a.z0 + o1, pixb, fldr_div_by(acc0, 3.0f, 1.0f / 3.0f));
            prep_stf(a.z1 + o1, pixb, fldr_div_by(acc1, 3.0f, 1.0f / 3.0f));
        }
    }
    if (!ph2) return;

    // ---- phase 2: backward flows (fLDRnet.py:474-475) from the low-resolution field, then the backward-warped frames (:478-479) ----
    float fb0x[PL_RPT], fb0y[PL_RPT], fb1x[PL_RPT], fb1y[PL_RPT];
    uint32_t mnA = 0x7fff7fffu, mxA = 0u, mnB = 0x7fff7fffu, mxB = 0u;       // A: I0 at flowback_0, B: I1 at flowback_1
#pragma unroll
    for (int k = 0; k < PL_RPT; ++k) {
        fb0x[k] = fb0y[k] = fb1x[k] = fb1y[k] = 0.0f;
        if (!live[k]) continue;
        const int py = py0 + 4 * k;
        const float fpy = (float)py;
        const uint32_t pixb = (__umul24((uint32_t)py, (uint32_t)a.W) + (uint32_t)px) * 4u;
        const FldrTap tb0 = fldr_grid_tap(fpx, fpy, omt * f01x[k], omt * f01y[k], a.W, a.H, a.inv_wm1, a.inv_hm1, a.r_wm1, a.r_hm1);
        const FldrTap tb1 = fldr_grid_tap(fpx, fpy, tv * f10x[k], tv * f10y[k], a.W, a.H, a.inv_wm1, a.inv_hm1, a.r_wm1, a.r_hm1);
        const FldrTapP tb0p = fldr_tap_prepare(tb0, a.W, a.H), tb1p = fldr_tap_prepare(tb1, a.W, a.H);
        const float mb0 = a.withmask ? fldr_tap_mask_p(tb0p) : 1.0f, mb1 = a.withmask ? fldr_tap_mask_p(tb1p) : 1.0f;
        float x0, y0, x1, y1;
        prep_sample_up2(tb0, tb0p, lo10, a, tv, x0, y0);
        prep_sample_up2(tb1, tb1p, lo01, a, omt, x1, y1);
        x0 = x0 * mb0; y0 = y0 * mb0; x1 = x1 * mb1; y1 = y1 * mb1;
        fb0x[k] = x0; fb0y[k] = y0; fb1x[k] = x1; fb1y[k] = y1;
        prep_stf(a.flowback_0 + o2, pixb, x0); prep_stf(a.flowback_0 + o2 + HW, pixb, y0);
        prep_stf(a.flowback_1 + o2, pixb, x1); prep_stf(a.flowback_1 + o2 + HW, pixb, y1);
        const PlBox b0 = pl_box(fldr_grid_tap(fpx, fpy, x0, y0, a.W, a.H, a.inv_wm1, a.inv_hm1, a.r_wm1, a.r_hm1), a.W, a.H);
        const PlBox b1 = pl_box(fldr_grid_tap(fpx, fpy, x1, y1, a.W, a.H, a.inv_wm1, a.inv_hm1, a.r_wm1, a.r_hm1), a.W, a.H);
        mnA = pl_min2(mnA, pl_pack(b0.xa, b0.ya)); mxA = pl_max2(mxA, pl_pack(b0.xb, b0.yb));
        mnB = pl_min2(mnB, pl_pack(b1.xa, b1.ya)); mxB = pl_max2(mxB, pl_pack(b1.xb, b1.yb));
    }
    const bool fit = reduce_boxes(red_base + 16, mnA, mxA, mnB, mxB);
    if (fit) fill_windows(i0, a.i0_cstride, i1, a.i1_cstride);
#pragma unroll
    for (int k = 0; k < PL_RPT; ++k) {
        if (!live[k]) continue;
        const int py = py0 + 4 * k;
        const float fpy = (float)py;
        const uint32_t pixb = (__umul24((uint32_t)py, (uint32_t)a.W) + (uint32_t)px) * 4u;
        const FldrTap ti0 = fldr_grid_tap(fpx, fpy, fb0x[k], fb0y[k], a.W, a.H, a.inv_wm1, a.inv_hm1, a.r_wm1, a.r_hm1);
        const FldrTap ti1 = fldr_grid_tap(fpx, fpy, fb1x[k], fb1y[k], a.W, a.H, a.inv_wm1, a.inv_hm1, a.r_wm1, a.r_hm1);
        const FldrTapP ti0p = fldr_tap_prepare(ti0, a.W, a.H), ti1p = fldr_tap_prepare(ti1, a.W, a.H);
        const PlBox b0 = pl_box(ti0, a.W, a.H), b1 = pl_box(ti1, a.W, a.H);
        const float mi0 = a.withmask ? fldr_tap_mask_p(ti0p) : 1.0f, mi1 = a.withmask ? fldr_tap_mask_p(ti1p) : 1.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v0 = (fit ? pl_sample(ti0, b0, win + c * PL_PLANE, oxA, oyA) : fldr_tap_sample_p(ti0p, i0 + (int64_t)c * a.i0_cstride)) * mi0;
            const float v1 = (fit ? pl_sample(ti1, b1, win + (3 + c) * PL_PLANE, oxB, oyB) : fldr_tap_sample_p(ti1p, i1 + (int64_t)c * a.i1_cstride)) * mi1;
            prep_stf(a.im0_tot + o3 + (int64_t)c * HW, pixb, v0);
            prep_stf(a.im1_tot + o3 + (int64_t)c * HW, pixb, v1);
        }
    }
}

#endif  // FLDR_TEST_HOOKS (LDS-staged gather windows)

static int g_prep_variant = 0;                   // test build: 1 = LDS-staged gather windows where the geometry allows; 0 (default, product): the kernel with global gathers
FLDR_HOOK int fldr_debug_prep_variant(int v) { if (v == 0 || v == 1) g_prep_variant = v; return g_prep_variant; }

extern "C" int fldr_level0_prep(const fldr_prep_desc* d, fldr_stream_t stream) {
    FLDR_CHECK_ARG(d && d->ws && d->flow_lo && d->I0 && d->I1 && d->t && d->flow_t0 && d->flow_t1 && d->flowback_0 && d->flowback_1);
    FLDR_CHECK_ARG(d->im0_tot && d->im1_tot && (!d->z0 == !d->z1) && d->N > 0 && d->h > 0 && d->w > 0 && d->H > 0 && d->W > 0);
    if (d->H < d->h || d->W < d->w || (int64_t)d->H * d->W * 4 >= (1ll << 32)) return FLDR_E_SHAPE;          // upsampling only (prep_sample_up2's 3x3 neighbourhood)
    PrepArgs a;
    a.flow_lo2 = reinterpret_cast<const float2*>(d->ws); a.N = d->N; a.I0 = d->I0; a.I1 = d->I1; a.i0_bstride = d->i0_bstride; a.i1_bstride = d->i1_bstride;
    a.i0_cstride = d->i0_cstride ? d->i0_cstride : (int64_t)d->H * d->W; a.i1_cstride = d->i1_cstride ? d->i1_cstride : (int64_t)d->H * d->W;
    a.t = d->t; a.z0 = d->z0; a.z1 = d->z1; a.flow_t0 = d->flow_t0; a.flow_t1 = d->flow_t1;
    a.flowback_0 = d->flowback_0; a.flowback_1 = d->flowback_1; a.im0_tot = d->im0_tot; a.im1_tot = d->im1_tot;
    a.h = d->h; a.w = d->w; a.H = d->H; a.W = d->W;
    a.sy = (float)d->h / (float)d->H; a.sx = (float)d->w / (float)d->W; a.mul = d->mul;
    a.inv_wm1 = (float)(d->W - 1 > 1 ? d->W - 1 : 1); a.inv_hm1 = (float)(d->H - 1 > 1 ? d->H - 1 : 1);
    a.r_wm1 = 1.0f / a.inv_wm1; a.r_hm1 = 1.0f / a.inv_hm1;
    a.za0 = d->z_alpha0; a.za1 = d->z_alpha1; a.withmask = d->withmask;
    a.phase = (d->phase & 3) ? (d->phase & 3) : 3;
    a.kx = a.ky = -1;
    for (int k = 0; k <= 6; ++k) {                                   // (2 o + 1 stays far inside int for any plane < 4 GB)
        if (a.sx == 1.0f / (float)(1 << k)) a.kx = k;
        if (a.sy == 1.0f / (float)(1 << k)) a.ky = k;
    }
    a.rkx = a.kx >= 0 ? 1.0f / (float)(1 << (a.kx + 1)) : 0.0f;
    a.rky = a.ky >= 0 ? 1.0f / (float)(1 << (a.ky + 1)) : 0.0f;
    const int64_t hw = (int64_t)d->h * d->w;
    if (!(d->phase & 4))                                          // bit 2: d->ws already holds the interleaved flow (second phase of a split call)
        hipLaunchKernelGGL(prep_interleave_kernel, dim3(fldr_cdiv(hw, 256), d->N), dim3(256), 0, fldr_s(stream), d->flow_lo,
                           reinterpret_cast<float2*>(d->ws), hw);
    // LDS windows: 16-byte DMA chunks need W % 4 == 0 and 16-byte aligned planes; corner coordinates are packed as int16
    const bool al16 = !((reinterpret_cast<uintptr_t>(a.I0) | reinterpret_cast<uintptr_t>(a.I1)) & 15) && !((a.i0_cstride | a.i1_cstride | a.i0_bstride | a.i1_bstride) & 3);
#ifdef FLDR_TEST_HOOKS
    if (g_prep_variant == 1 && !(d->W & 3) && d->W >= 4 && d->W < 32768 && d->H < 32768 && al16) {
        static std::atomic<uint64_t> attr_done{0};
        if (int e = fldr_set_max_lds(reinterpret_cast<const void*>(&level0_prep_lds_kernel), PL_LDS_BYTES, attr_done)) return e;
        dim3 grid(fldr_cdiv(d->W, PL_TW), fldr_cdiv(d->H, PL_TH), d->N);
        hipLaunchKernelGGL(level0_prep_lds_kernel, grid, dim3(256), PL_LDS_BYTES, fldr_s(stream), a);
        FLDR_LAUNCH_RET();
    }
#endif
    (void)al16;
    dim3 grid(fldr_cdiv(d->W, 64), fldr_cdiv(d->H, 4), d->N);
    if (a.phase == 1) hipLaunchKernelGGL(level0_prep_kernel<1>, grid, dim3(256), 0, fldr_s(stream), a);
    else if (a.phase == 2) hipLaunchKernelGGL(level0_prep_kernel<2>, grid, dim3(256), 0, fldr_s(stream), a);
    else hipLaunchKernelGGL(level0_prep_kernel<3>, grid, dim3(256), 0, fldr_s(stream), a);
    FLDR_LAUNCH_RET();
}
