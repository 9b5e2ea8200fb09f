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
    // 1-D grid, XCD-aware (round 4): workgroups b and b + 8 share an XCD (and its L2), so XCD x = b & 7 walks the CONTIGUOUS
    // range of tiles [x * tiles_per_xcd, (x + 1) * tiles_per_xcd) in row-major order — a band of rows.  Every frame plane is read
    // twice directly and eight times through gathers a few pixels away: dealt round-robin, each of the 8 L2s fetched nearly the
    // whole frame again (647 MB fetched for 212 MB of planes, profiles/r03_forward_pmc.txt).
    int tiles_x, n_tiles, tiles_per_xcd;
    uint32_t m_tiles_x;            // floor(2^32 / tiles_x) + 1
};
// tile index of this workgroup, or -1
__device__ __forceinline__ int prep_tile(const PrepArgs& a, int& bx, int& by) {
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int tile = a.tiles_per_xcd ? xcd * a.tiles_per_xcd + k : (int)blockIdx.x;     // (0: row-major deal, the A/B setting)
    if ((a.tiles_per_xcd && k >= a.tiles_per_xcd) || tile >= a.n_tiles) return -1;
    by = a.tiles_x == 1 ? tile : (int)__umulhi((uint32_t)tile, a.m_tiles_x);
    bx = tile - by * a.tiles_x;
    return tile;
}

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
__global__ __launch_bounds__(256) void level0_prep_kernel(PrepArgs a) {
#pragma clang fp contract(off)
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    int bx, by;
    if (prep_tile(a, bx, by) < 0) return;
    const int px = bx * 64 + tx;
    const int py = by * 4 + ty;
    const int n = blockIdx.y;
    const bool live = px < a.W && py < a.H;
    const int64_t HW = (int64_t)a.H * a.W, hw = (int64_t)a.h * a.w;
    const bool ph1 = (a.phase & 1) != 0, ph2 = (a.phase & 2) != 0;        // uniform
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

// ---- runs of four pixels (round 4) -------------------------------------------------------------------------------------
// The thread-per-pixel kernel above moves its 6 + 16 planes through 4-byte lanes, and this chip streams planes at 3.9-4.1 TB/s
// that way against 5.2-6.0 TB/s through 16-byte lanes (tools/ubench/plane_bw_bench): its 1.21 GB in 311 us sat exactly on that
// cap.  Here a thread owns FOUR horizontally adjacent pixels (x = 4 m .. 4 m + 3): the frames at the run are six 16-byte loads,
// every output plane one 16-byte streaming store, and everything that is constant over the run is evaluated once — the row's
// vertical source indices, and, for upsampling factors >= 8, the horizontal ones too: with sx = 2^-k, k >= 3, the source index
// (2 x + 1 - 2^k) >> (k + 1) is the same for the four pixels of an aligned run (2 x + 1 = 8 m + r, r in {1,3,5,7}: the shift drops
// r), so the 2x2 low-resolution neighbourhood — 8 loads and their addresses — is shared and only the fraction differs.  The
// per-pixel arithmetic (taps, gathers, sums) is the SAME device functions in the same order: bit-identical outputs.
// Requirements (else the kernel above runs): W % 4 == 0, kx >= 3, every plane base / stride a multiple of 16 bytes.
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
template <int R> struct PrepVec;
template <> struct PrepVec<4> { typedef f4v T; };
template <> struct PrepVec<2> { typedef f2v T; };
template <int R>
__device__ __forceinline__ typename PrepVec<R>::T prep_ldv(const float* __restrict__ base, uint32_t boff) {
    return *reinterpret_cast<const typename PrepVec<R>::T*>(reinterpret_cast<const char*>(base) + boff);
}
template <int R>
__device__ __forceinline__ void prep_stv(float* __restrict__ base, uint32_t boff, const float (&v)[R]) {
    typename PrepVec<R>::T x;
#pragma unroll
    for (int j = 0; j < R; ++j) x[j] = v[j];
#if PREP_NT
    __builtin_nontemporal_store(x, reinterpret_cast<typename PrepVec<R>::T*>(reinterpret_cast<char*>(base) + boff));
#else
    *reinterpret_cast<typename PrepVec<R>::T*>(reinterpret_cast<char*>(base) + boff) = x;
#endif
}

#ifndef PREP_QUAD_WAVES
#define PREP_QUAD_WAVES 2                        // minimum waves per SIMD the register allocation is held to
#endif
// R: pixels per thread (4: 16-byte lanes; 2: 8-byte lanes, half the registers, lanes of a gather two pixels apart instead of four)
template <int R>
__global__ __launch_bounds__(256, R == 4 ? PREP_QUAD_WAVES : 4) void level0_prep_run_kernel(PrepArgs a) {
#pragma clang fp contract(off)
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    int bx, by;
    if (prep_tile(a, bx, by) < 0) return;
    const int px0 = (bx * 64 + tx) * R;
    const int py = by * 4 + ty;
    const int n = blockIdx.y;
    if (px0 >= a.W || py >= a.H) return;
    const int64_t HW = (int64_t)a.H * a.W, hw = (int64_t)a.h * a.w;
    const bool ph1 = (a.phase & 1) != 0, ph2 = (a.phase & 2) != 0;        // uniform
    const int64_t o1 = (int64_t)n * HW, o2 = (int64_t)n * 2 * HW, o3 = (int64_t)n * 3 * HW;
    const uint32_t pixb = (__umul24((uint32_t)py, (uint32_t)a.W) + (uint32_t)px0) * 4u;      // byte offset of the run inside a plane
    const float2* lo10 = a.flow_lo2 + (int64_t)n * hw;
    const float2* lo01 = a.flow_lo2 + (int64_t)(a.N + n) * hw;
    const float* i0 = a.I0 + (int64_t)n * a.i0_bstride;
    const float* i1 = a.I1 + (int64_t)n * a.i1_bstride;
    const float tv = a.t[n], omt = 1.0f - tv;
    const bool want_z = ph1 && a.z0;

    // the frames at the run (direct reads, issued first; only the splat metrics use them)
    typename PrepVec<R>::T c0[3], c1[3];
    if (want_z) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { c0[c] = prep_ldv<R>(i0 + (int64_t)c * a.i0_cstride, pixb); c1[c] = prep_ldv<R>(i1 + (int64_t)c * a.i1_cstride, pixb); }
    }

    // upsampled flows at the run: one neighbourhood (see above), R fractions
    const PrepLin ly = prep_lin(py, a.sy, a.h, a.ky, a.rky);
    PrepLin lx[R];
#pragma unroll
    for (int j = 0; j < R; ++j) lx[j] = prep_lin(px0 + j, a.sx, a.w, a.kx, a.rkx);
    const PrepQuad q = prep_quad(lo10, lo01, a.w, lx[0], ly);
    float f10x[R], f10y[R], f01x[R], f01y[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        f10x[j] = prep_up(q, 0, lx[j], ly, a.mul, 0, 1.0f); f10y[j] = prep_up(q, 1, lx[j], ly, a.mul, 0, 1.0f);
        f01x[j] = prep_up(q, 2, lx[j], ly, a.mul, 0, 1.0f); f01y[j] = prep_up(q, 3, lx[j], ly, a.mul, 0, 1.0f);
    }
    const float fpy = (float)py;

    if (want_z) {
        float z0v[R], z1v[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const float fpx = (float)(px0 + j);
            const FldrTapP t0 = fldr_tap_prepare(fldr_grid_tap(fpx, fpy, f01x[j], f01y[j], a.W, a.H, a.inv_wm1, a.inv_hm1, a.r_wm1, a.r_hm1), a.W, a.H);
            const FldrTapP t1 = fldr_tap_prepare(fldr_grid_tap(fpx, fpy, f10x[j], f10y[j], a.W, a.H, a.inv_wm1, a.inv_hm1, a.r_wm1, a.r_hm1), a.W, a.H);
            const float m0 = fldr_tap_mask_p(t0), m1 = fldr_tap_mask_p(t1);
            float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float w0 = fldr_tap_sample_p(t0, i1 + (int64_t)c * a.i1_cstride) * m0;
                const float w1 = fldr_tap_sample_p(t1, i0 + (int64_t)c * a.i0_cstride) * m1;
                acc0 += a.za0 * fabsf(c0[c][j] - w0);
                acc1 += a.za1 * fabsf(c1[c][j] - w1);
            }
            z0v[j] = fldr_div_by(acc0, 3.0f, 1.0f / 3.0f);
            z1v[j] = fldr_div_by(acc1, 3.0f, 1.0f / 3.0f);
        }
        prep_stv<R>(a.z0 + o1, pixb, z0v);
        prep_stv<R>(a.z1 + o1, pixb, z1v);
    }
    if (ph1) {
        float v[4][R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            v[0][j] = prep_up(q, 2, lx[j], ly, a.mul, 1, tv);  v[1][j] = prep_up(q, 3, lx[j], ly, a.mul, 1, tv);
            v[2][j] = prep_up(q, 0, lx[j], ly, a.mul, 1, omt); v[3][j] = prep_up(q, 1, lx[j], ly, a.mul, 1, omt);
        }
        prep_stv<R>(a.flow_t0 + o2, pixb, v[0]); prep_stv<R>(a.flow_t0 + o2 + HW, pixb, v[1]);
        prep_stv<R>(a.flow_t1 + o2, pixb, v[2]); prep_stv<R>(a.flow_t1 + o2 + HW, pixb, v[3]);
    }
    if (ph2) {
        float fb[4][R];                                                    // flowback_0 (x, y), flowback_1 (x, y) of the run's pixels
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const float fpx = (float)(px0 + j);
            const FldrTap tb0 = fldr_grid_tap(fpx, fpy, omt * f01x[j], omt * f01y[j], a.W, a.H, a.inv_wm1, a.inv_hm1, a.r_wm1, a.r_hm1);
            const FldrTap tb1 = fldr_grid_tap(fpx, fpy, tv * f10x[j], tv * f10y[j], a.W, a.H, a.inv_wm1, a.inv_hm1, a.r_wm1, a.r_hm1);
            const FldrTapP tb0p = fldr_tap_prepare(tb0, a.W, a.H), tb1p = fldr_tap_prepare(tb1, a.W, a.H);
            const float mb0 = a.withmask ? fldr_tap_mask_p(tb0p) : 1.0f, mb1 = a.withmask ? fldr_tap_mask_p(tb1p) : 1.0f;
            float fb0x, fb0y, fb1x, fb1y;
            prep_sample_up2(tb0, tb0p, lo10, a, tv, fb0x, fb0y);
            prep_sample_up2(tb1, tb1p, lo01, a, omt, fb1x, fb1y);
            fb[0][j] = fb0x * mb0; fb[1][j] = fb0y * mb0; fb[2][j] = fb1x * mb1; fb[3][j] = fb1y * mb1;
        }
        prep_stv<R>(a.flowback_0 + o2, pixb, fb[0]); prep_stv<R>(a.flowback_0 + o2 + HW, pixb, fb[1]);
        prep_stv<R>(a.flowback_1 + o2, pixb, fb[2]); prep_stv<R>(a.flowback_1 + o2 + HW, pixb, fb[3]);
        float im[6][R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const float fpx = (float)(px0 + j);
            const FldrTapP ti0 = fldr_tap_prepare(fldr_grid_tap(fpx, fpy, fb[0][j], fb[1][j], a.W, a.H, a.inv_wm1, a.inv_hm1, a.r_wm1, a.r_hm1), a.W, a.H);
            const FldrTapP ti1 = fldr_tap_prepare(fldr_grid_tap(fpx, fpy, fb[2][j], fb[3][j], a.W, a.H, a.inv_wm1, a.inv_hm1, a.r_wm1, a.r_hm1), a.W, a.H);
            const float mi0 = a.withmask ? fldr_tap_mask_p(ti0) : 1.0f, mi1 = a.withmask ? fldr_tap_mask_p(ti1) : 1.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                im[c][j] = fldr_tap_sample_p(ti0, i0 + (int64_t)c * a.i0_cstride) * mi0;
                im[3 + c][j] = fldr_tap_sample_p(ti1, i1 + (int64_t)c * a.i1_cstride) * mi1;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            prep_stv<R>(a.im0_tot + o3 + (int64_t)c * HW, pixb, im[c]);
            prep_stv<R>(a.im1_tot + o3 + (int64_t)c * HW, pixb, im[3 + c]);
        }
    }
}

static int g_prep_quad = 0;                      // pixels per thread where the geometry allows: 0 / 1 one (the kernel above), 2, 4
static int g_prep_xcd = 0;                       // 0: tiles dealt to the workgroups in row-major order (cross-check / A-B)
FLDR_HOOK int fldr_debug_prep_xcd(int v) { if (v == 0 || v == 1) g_prep_xcd = v; return g_prep_xcd; }
FLDR_HOOK int fldr_debug_prep_quad(int v) { if (v == 0 || v == 1 || v == 2 || v == 4) g_prep_quad = v; return g_prep_quad; }

static inline bool prep_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

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
    const int64_t HW = (int64_t)d->H * d->W;
    const int R = g_prep_quad == 2 ? 2 : 4;
    const bool quad = g_prep_quad > 1 && a.kx >= 3 && !(d->W & 3) && !(HW & 3) && !(a.i0_cstride & 3) && !(a.i1_cstride & 3) && !(a.i0_bstride & 3) &&
                      !(a.i1_bstride & 3) && prep_al16(a.I0) && prep_al16(a.I1) && (!a.z0 || (prep_al16(a.z0) && prep_al16(a.z1))) && prep_al16(a.flow_t0) &&
                      prep_al16(a.flow_t1) && prep_al16(a.flowback_0) && prep_al16(a.flowback_1) && prep_al16(a.im0_tot) && prep_al16(a.im1_tot);
    a.tiles_x = fldr_cdiv(d->W, quad ? 64 * R : 64);
    a.n_tiles = a.tiles_x * fldr_cdiv(d->H, 4);
    if ((int64_t)a.n_tiles * a.tiles_x >= (1ll << 32)) return FLDR_E_SHAPE;                      // exactness of the umulhi division
    a.m_tiles_x = (uint32_t)((1ull << 32) / (uint32_t)a.tiles_x) + 1u;
    // whole tile rows per XCD (a band of rows); g_prep_xcd == 0: tiles dealt round-robin as before (tiles_per_xcd = n_tiles on "XCD" 0 .. 7
    // is expressed by one tile per step: see prep_tile)
    a.tiles_per_xcd = g_prep_xcd ? fldr_cdiv(fldr_cdiv(a.n_tiles, a.tiles_x), 8) * a.tiles_x : 0;
    dim3 grid(8 * (g_prep_xcd ? a.tiles_per_xcd : fldr_cdiv(a.n_tiles, 8)), d->N);
    if (quad && R == 4) hipLaunchKernelGGL(level0_prep_run_kernel<4>, grid, dim3(256), 0, fldr_s(stream), a);
    else if (quad) hipLaunchKernelGGL(level0_prep_run_kernel<2>, grid, dim3(256), 0, fldr_s(stream), a);
    else hipLaunchKernelGGL(level0_prep_kernel, grid, dim3(256), 0, fldr_s(stream), a);
    FLDR_LAUNCH_RET();
}
