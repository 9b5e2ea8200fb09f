// Device functions of the level-0 synthesis inputs (fLDRnet.py:400-479), shared by level0_prep_kernel (prep_kernels.hip) and the
// fused enc1 kernel (enc1_fused_kernels.hip: the second half of the prep work evaluated inside enc1's staging).
//
// Every value is produced with the operands and the operation order of the unfused kernels (resize_bilinear_kernel, zmetric_kernel,
// bwarp_kernel: fldr_lin_src, fldr_grid_tap, fldr_tap_sample, fldr_tap_mask of common.h; contraction off), so the results are
// bit-identical to them (tests/test_gpu_parity.py::test_level0_prep_bit_identical_to_unfused).  What differs is the instruction
// count — the prep kernel is bound by its vector instructions (911 per pixel in round 4) and runs below the board's power limit, so
// instructions removed here return as time:
//   * source indices of the x2^k upsampling from integer arithmetic without the clamps that cannot trigger for coordinates inside
//     the upsampled image (prep_lin_in: 6 instead of 11 instructions, ten uses per pixel);
//   * a tap is built prepared (clamped corner offsets + masked weights) in one go: integer clamps as v_med3_i32, corner validity as
//     one unsigned compare per coordinate, the float clamp in front of the int conversion as v_med3_f32;
//   * both channels of a 2-channel field ride in one <2 x float> (v_pk_mul_f32 / v_pk_add_f32: the same two fp32 operations per
//     lane, one instruction);
//   * row-dependent quantities of a wave that covers one image row are wave-uniform (scalar) in the caller.
#pragma once
#include "common.h"

typedef float prep_f2 __attribute__((ext_vector_type(2)));

// Uniform base pointer + 32-bit byte offset: one global_load / global_store with an SGPR base and a VGPR offset, no 64-bit
// address arithmetic per access.  Planes are < 4 GB (host-checked).
__device__ __forceinline__ float prep_ldf(const float* __restrict__ base, uint32_t boff) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + boff); }
__device__ __forceinline__ prep_f2 prep_ldf2(const float2* __restrict__ base, uint32_t boff) { return *reinterpret_cast<const prep_f2*>(reinterpret_cast<const char*>(base) + boff); }

// clamp(x, 0, hi) for 0 <= hi: one v_med3_i32 (LLVM only forms it from min / max when both bounds are constants)
__device__ __forceinline__ int prep_clamp0(int x, int hi) {
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "v"(hi));
    return r;
}

// Source indices / weight of F.interpolate(bilinear, align_corners=False) along one axis (fldr_lin_src), computed once
// and shared by every plane and tap that is evaluated at the same coordinate.
struct PrepLin { int i0, i1; float l; };
// For an output coordinate o INSIDE the upsampled axis (0 <= o < in_size << kshift):
// kshift >= 0: scale == 2^-kshift exactly (the model's x8 upsampling: 0.125).  Then r = scale * (o + 0.5) - 0.5 =
// (2 o + 1 - 2^k) / 2^(k+1) is exact in fp32 and fldr_lin_src's floor / fraction are a shift and a mask of the integer
// numerator t: i = t >> (k+1) <= in_size - 1 (so fldr_lin_src's clamp of i0 never triggers), the fraction (t & (2^(k+1) - 1)) /
// 2^(k+1) < 1 (nor does its clamp of l) — the same i0, i1 and l bit for bit.  kshift < 0: fldr_lin_src itself.
template <bool P2>
__device__ __forceinline__ PrepLin prep_lin_in(int o, float scale, int in_size, int kshift, float rk) {
    PrepLin r;
    if constexpr (P2) {
        const int t = 2 * o + 1 - (1 << kshift);                         // numerator of r over 2^(k+1); r < 0 clamps to 0
        const int tc = t < 0 ? 0 : t;
        r.i0 = tc >> (kshift + 1);
        r.i1 = min(r.i0 + 1, in_size - 1);
        r.l = (float)(tc & ((2 << kshift) - 1)) * rk;                    // rk = 2^-(kshift+1) from the host
    } else {
        fldr_lin_src(o, scale, in_size, r.i0, r.i1, r.l);
    }
    return r;
}

// The four low-resolution neighbours of one full-resolution coordinate, both 2-channel flows (flow_10 = a, flow_01 = b): 8 loads of 8 B.
struct PrepQuad { prep_f2 a00, a01, a10, a11, b00, b01, b10, b11; };
__device__ __forceinline__ PrepQuad prep_quad(const float2* __restrict__ p10, const float2* __restrict__ p01, int w, const PrepLin& ix,
                                              const PrepLin& iy) {
    PrepQuad q;
    const uint32_t y0 = __umul24((uint32_t)iy.i0, (uint32_t)w), y1 = __umul24((uint32_t)iy.i1, (uint32_t)w);   // full-rate 24-bit multiplies
    const uint32_t o00 = (y0 + (uint32_t)ix.i0) * 8u, o01 = (y0 + (uint32_t)ix.i1) * 8u;
    const uint32_t o10 = (y1 + (uint32_t)ix.i0) * 8u, o11 = (y1 + (uint32_t)ix.i1) * 8u;
    q.a00 = prep_ldf2(p10, o00); q.a01 = prep_ldf2(p10, o01); q.a10 = prep_ldf2(p10, o10); q.a11 = prep_ldf2(p10, o11);
    q.b00 = prep_ldf2(p01, o00); q.b01 = prep_ldf2(p01, o01); q.b10 = prep_ldf2(p01, o10); q.b11 = prep_ldf2(p01, o11);
    return q;
}

// F.interpolate(bilinear, align_corners=False)(scale * field)[Y, X] * mul for both channels of a 2-channel field — the arithmetic of
// resize_bilinear_kernel per channel, the field first multiplied by `scale` (pre) in fp32 as `t4 * flow_01_lo` does.
template <bool PRE>
__device__ __forceinline__ prep_f2 prep_up2(prep_f2 a00, prep_f2 a01, prep_f2 a10, prep_f2 a11, float lx, float ly, float mul, float scale) {
#pragma clang fp contract(off)
    if constexpr (PRE) { a00 = scale * a00; a01 = scale * a01; a10 = scale * a10; a11 = scale * a11; }
    const float wx0 = 1.0f - lx, wy0 = 1.0f - ly;
    const prep_f2 top = wx0 * a00 + lx * a01;
    const prep_f2 bot = wx0 * a10 + lx * a11;
    return (wy0 * top + ly * bot) * mul;
}

// A backward-warp tap (fldr_grid_tap + fldr_tap_prepare of common.h in one go): north-west integer corner (for the flow
// re-evaluation of prep_sample_up2), the four CLAMPED corner positions as 32-bit byte offsets into a contiguous [H,W] fp32 plane and
// the corner weights with out-of-bounds corners zeroed.  Same position arithmetic, operation for operation (see fldr_grid_tap).
struct PrepTap {
    int xa, xb, ya, yb;          // clamped corner coordinates
    uint32_t onw, one, osw, ose;
    float wnw, wne, wsw, wse;    // masked
};
template <class A>
__device__ __forceinline__ PrepTap prep_tap(float px, float py, float fx, float fy, const A& a) {
#pragma clang fp contract(off)
    PrepTap t;
    const float vx = px + fx;
    const float vy = py + fy;
    const float gx = fldr_div_by(2.0f * vx, a.inv_wm1, a.r_wm1) - 1.0f;   // == (2 vx) / wm1: torch's div(Tensor, Scalar) is a true fp32 division
    const float gy = fldr_div_by(2.0f * vy, a.inv_hm1, a.r_hm1) - 1.0f;
    const float ix = (gx + 1.0f) * ((float)a.W * 0.5f) - 0.5f;
    const float iy = (gy + 1.0f) * ((float)a.H * 0.5f) - 0.5f;
    float xf = floorf(ix), yf = floorf(iy);
    const float w = ix - xf, e = 1.0f - w;
    const float n = iy - yf, s = 1.0f - n;
    // clamp before the int conversion: wild flows must not overflow int (v_med3_f32; a NaN gives the lower bound like fminf(fmaxf()))
    xf = __builtin_amdgcn_fmed3f(xf, -2.0f, (float)a.W + 1.0f);
    yf = __builtin_amdgcn_fmed3f(yf, -2.0f, (float)a.H + 1.0f);
    const int x0 = (int)xf, y0 = (int)yf;
    const bool x0v = (uint32_t)x0 < (uint32_t)a.W, x1v = (uint32_t)(x0 + 1) < (uint32_t)a.W;
    const bool y0v = (uint32_t)y0 < (uint32_t)a.H, y1v = (uint32_t)(y0 + 1) < (uint32_t)a.H;
    const float wnw = s * e, wne = s * w, wsw = n * e, wse = n * w;
    t.wnw = (x0v && y0v) ? wnw : 0.0f; t.wne = (x1v && y0v) ? wne : 0.0f;
    t.wsw = (x0v && y1v) ? wsw : 0.0f; t.wse = (x1v && y1v) ? wse : 0.0f;
    t.xa = prep_clamp0(x0, a.W - 1); t.xb = prep_clamp0(x0 + 1, a.W - 1);
    t.ya = prep_clamp0(y0, a.H - 1); t.yb = prep_clamp0(y0 + 1, a.H - 1);
    const uint32_t ra = __umul24((uint32_t)t.ya, (uint32_t)a.W), rb = __umul24((uint32_t)t.yb, (uint32_t)a.W);   // full-rate 24-bit multiply (coordinates < 2^24)
    t.onw = (ra + (uint32_t)t.xa) * 4u; t.one = (ra + (uint32_t)t.xb) * 4u;
    t.osw = (rb + (uint32_t)t.xa) * 4u; t.ose = (rb + (uint32_t)t.xb) * 4u;
    return t;
}
// fldr_tap_mask from the masked weights (same sum: the skipped corners add +0)
__device__ __forceinline__ float prep_tap_mask(const PrepTap& p) {
#pragma clang fp contract(off)
    float m = 0.0f;
    m += p.wnw; m += p.wne; m += p.wsw; m += p.wse;
    return m < 0.999f ? 0.0f : 1.0f;
}
// fldr_tap_sample against a wave-uniform pointer to a contiguous [H,W] plane, in two halves: prep_tap_gather requests the four corners,
// prep_tap_blend forms the sample (a masked corner contributes p * 0 = +-0 instead of a literal +0, which never changes a sum that
// starts at +0).  The callers request ALL samples of a phase before blending any: the kernel is bound by the latency of its dependent
// round trips — until round 5 every sample's four loads were pinned (fldr_pin, against LLVM sinking clamped loads into selects) right
// behind their issue, which made each sample a round trip of its own: 12 of the ~14 serialised round trips of a wave.  The prepared taps
// have no selects on loaded values, so nothing needs pinning.
struct PrepCorners { float nw, ne, sw, se; };
__device__ __forceinline__ PrepCorners prep_tap_gather(const PrepTap& p, const float* __restrict__ plane) {
    PrepCorners c;
#if defined(PREP_ABLATE) && PREP_ABLATE == 5            // diagnostic build: no image gathers at all
    c.nw = p.wnw; c.ne = p.wne; c.sw = p.wsw; c.se = p.wse; (void)plane;
#else
    c.nw = prep_ldf(plane, p.onw); c.ne = prep_ldf(plane, p.one); c.sw = prep_ldf(plane, p.osw); c.se = prep_ldf(plane, p.ose);
#endif
    return c;
}
__device__ __forceinline__ float prep_tap_blend(const PrepTap& p, const PrepCorners& c) {
#pragma clang fp contract(off)
    float v = 0.0f;
    v += c.nw * p.wnw;
    v += c.ne * p.wne;
    v += c.sw * p.wsw;
    v += c.se * p.wse;
    return v;
}

// ---- wave-private LDS windows of the low-resolution flow (round 5) --------------------------------------------------------------------
// A vector-memory instruction occupies the CU's address unit for ~16 cycles per 4 bytes per lane whatever it hits (measured on the prep
// kernel by removing instruction classes: a 4-byte gather ~17 cycles per wave, an 8-byte low-resolution load ~30, a 4-byte store ~22),
// and the kernel is bound by that unit — of its 96 vector-memory instructions per pixel, 26 were 8-byte loads of the low-resolution flow
// (the pixel's own 2 x 2 x 2 neighbours and the 3 x 3 neighbourhoods of the two backward-flow taps) that the 64 pixels of a wave fetch
// from a handful of cells.  So a wave stages those cells ONCE in LDS — 2 + 2 + 2 eight-byte loads — and every lane reads its
// neighbours from there (ds_read: not the address unit).  Windows are anchored at lane 0's neighbourhood; if any lane's neighbourhood
// falls outside (incoherent flows: wave-uniform test) the wave takes the global loads.  Same values either way: bit-identical.
#define PREP_QW 20                 // quad window: 2 source rows (a wave covers ONE image row) x PREP_QW source columns x float4 (f10 | f01)
#define PREP_UW 20                 // tap window: PREP_UH x PREP_UW cells of one 2-channel field
#define PREP_UH 6
#define PREP_WAVE_LDS (2 * PREP_QW * 16 + 2 * PREP_UW * PREP_UH * 8)       // bytes per wave: 640 + 1920
typedef __attribute__((address_space(3))) unsigned char* prep_lds_t;      // an LDS pointer by type: its accesses are ds_ instructions, never merged with the global path
#define PREP_LDS_F2(p) (*reinterpret_cast<__attribute__((address_space(3))) prep_f2*>(p))
__device__ __forceinline__ void prep_wave_lds_sync() {
    // other lanes' LDS writes -> this lane's reads: LDS executes a wave's operations in order; the fences keep the compiler from moving them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The pixel's own four low-resolution neighbours of both flows through the wave's quad window (iy wave-uniform).  Returns false (wave-
// uniform) when some lane's columns fall outside the window: the caller then uses prep_quad.
__device__ __forceinline__ bool prep_quad_window(const float2* __restrict__ p10, const float2* __restrict__ p01, int w, const PrepLin& ix, const PrepLin& iy,
                                                 prep_lds_t win, int lane, PrepQuad& q) {
    const int C0 = __builtin_amdgcn_readfirstlane(ix.i0);
    const bool inside = ix.i0 >= C0 && ix.i1 - C0 < PREP_QW;
    if (__builtin_amdgcn_ballot_w64(!inside) != 0ull) return false;
    {   // lanes 0 .. 2 QW - 1: cell (row lane / QW, column C0 + lane % QW, clamped: never read beyond the last real column), one field per instruction
        const int r = lane >= PREP_QW ? 1 : 0, c = lane - r * PREP_QW;
        const uint32_t off = (__umul24((uint32_t)(r ? iy.i1 : iy.i0), (uint32_t)w) + (uint32_t)min(C0 + c, w - 1)) * 8u;
        if (lane < 2 * PREP_QW) {
            const prep_f2 va = prep_ldf2(p10, off), vb = prep_ldf2(p01, off);
            PREP_LDS_F2(win + lane * 16) = va;
            PREP_LDS_F2(win + lane * 16 + 8) = vb;
        }
    }
    prep_wave_lds_sync();
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int o0 = (ix.i0 - C0) * 16, o1 = (ix.i1 - C0) * 16;
    typedef __attribute__((address_space(3))) f4* lf4;
    const f4 c00 = *reinterpret_cast<lf4>(win + o0), c01 = *reinterpret_cast<lf4>(win + o1);
    const f4 c10 = *reinterpret_cast<lf4>(win + PREP_QW * 16 + o0), c11 = *reinterpret_cast<lf4>(win + PREP_QW * 16 + o1);
    q.a00 = c00.xy; q.b00 = c00.zw; q.a01 = c01.xy; q.b01 = c01.zw;
    q.a10 = c10.xy; q.b10 = c10.zw; q.a11 = c11.xy; q.b11 = c11.zw;
    return true;
}

// bwarp_tscaled of a full-resolution 2-channel flow field that only exists as its low-resolution source `lo2` (x, y per
// low-resolution pixel): sample (xs * up(field)) at the tap `tp` with the arithmetic of bwarp_kernel's scaled branch.
// The tap's four corners are adjacent full-resolution pixels, and when upsampling (scale <= 1) adjacent pixels start
// their low-resolution neighbourhoods at most one cell apart: the four 2x2 neighbourhoods lie in ONE 3x3 block, loaded
// once (9 loads of 8 B instead of 16 of 16 B) and picked apart with selects; the horizontal interpolations are done once per
// neighbourhood row (west pair of columns for xa, east pair for xb) and the vertical pairs of rows picked afterwards: the operands
// and operations of prep_up2 per corner.  WIN: the 3x3 block comes from the wave's tap window; if some lane's block does not fit (wave-
// uniform), `fail` is set and the result is void — the caller redoes the pixel with WIN = false (global loads).
template <bool P2, bool WIN, class A>
__device__ __forceinline__ prep_f2 prep_sample_up2(const PrepTap& tp, const float2* __restrict__ lo2, const A& a, float xs,
                                                   prep_lds_t win, int lane, bool& fail) {
#pragma clang fp contract(off)
    const PrepLin lxa = prep_lin_in<P2>(tp.xa, a.sx, a.w, a.kx, a.rkx), lxb = prep_lin_in<P2>(tp.xb, a.sx, a.w, a.kx, a.rkx);
    const PrepLin lya = prep_lin_in<P2>(tp.ya, a.sy, a.h, a.ky, a.rky), lyb = prep_lin_in<P2>(tp.yb, a.sy, a.h, a.ky, a.rky);
    // columns lxa.i0 + {0,1,2} and rows lya.i0 + {0,1,2}, clamped like fldr_lin_src's i1: (i0, i1) of xa is columns (0,1),
    // of xb columns (dx, dx+1) with dx = lxb.i0 - lxa.i0 in {0,1}; rows alike
    const int c0 = lxa.i0, c1 = lxa.i1, c2 = min(c0 + 2, a.w - 1);
    const int r0 = lya.i0, r1 = lya.i1, r2 = min(r0 + 2, a.h - 1);
    prep_f2 m[3][3];
    if constexpr (WIN) {
        const int C0 = __builtin_amdgcn_readfirstlane(c0) - 3, R0 = __builtin_amdgcn_readfirstlane(r0) - 2;     // may be negative: the fill clamps
        const bool inside = c0 >= C0 && c0 + 2 - C0 < PREP_UW && r0 >= R0 && r0 + 2 - R0 < PREP_UH;
        if (__builtin_amdgcn_ballot_w64(!inside) != 0ull) { fail = true; return prep_f2{0.0f, 0.0f}; }      // wave-uniform: the caller redoes the pixel on the global path
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cell = lane + 64 * i;
            if (cell < PREP_UW * PREP_UH) {
                const int r = cell / PREP_UW, c = cell - r * PREP_UW;
                const int gr = min(max(R0 + r, 0), a.h - 1), gc = min(max(C0 + c, 0), a.w - 1);
                PREP_LDS_F2(win + cell * 8) = prep_ldf2(lo2, (__umul24((uint32_t)gr, (uint32_t)a.w) + (uint32_t)gc) * 8u);
            }
        }
        prep_wave_lds_sync();
        const int b0 = ((r0 - R0) * PREP_UW - C0) * 8, b1 = ((r1 - R0) * PREP_UW - C0) * 8, b2 = ((r2 - R0) * PREP_UW - C0) * 8;
        const int k0 = c0 * 8, k1 = c1 * 8, k2 = c2 * 8;
        m[0][0] = PREP_LDS_F2(win + b0 + k0); m[0][1] = PREP_LDS_F2(win + b0 + k1); m[0][2] = PREP_LDS_F2(win + b0 + k2);
        m[1][0] = PREP_LDS_F2(win + b1 + k0); m[1][1] = PREP_LDS_F2(win + b1 + k1); m[1][2] = PREP_LDS_F2(win + b1 + k2);
        m[2][0] = PREP_LDS_F2(win + b2 + k0); m[2][1] = PREP_LDS_F2(win + b2 + k1); m[2][2] = PREP_LDS_F2(win + b2 + k2);
    } else {
        const uint32_t q0 = __umul24((uint32_t)r0, (uint32_t)a.w) * 8u, q1 = __umul24((uint32_t)r1, (uint32_t)a.w) * 8u, q2 = __umul24((uint32_t)r2, (uint32_t)a.w) * 8u;
        const uint32_t k0 = (uint32_t)c0 * 8u, k1 = (uint32_t)c1 * 8u, k2 = (uint32_t)c2 * 8u;
        m[0][0] = prep_ldf2(lo2, q0 + k0); m[0][1] = prep_ldf2(lo2, q0 + k1); m[0][2] = prep_ldf2(lo2, q0 + k2);
        m[1][0] = prep_ldf2(lo2, q1 + k0); m[1][1] = prep_ldf2(lo2, q1 + k1); m[1][2] = prep_ldf2(lo2, q1 + k2);
        m[2][0] = prep_ldf2(lo2, q2 + k0); m[2][1] = prep_ldf2(lo2, q2 + k1); m[2][2] = prep_ldf2(lo2, q2 + k2);
    }
    const bool dx = lxb.i0 != lxa.i0, dy = lyb.i0 != lya.i0;
    const float wxa = 1.0f - lxa.l, wxb = 1.0f - lxb.l, wya = 1.0f - lya.l, wyb = 1.0f - lyb.l;
    prep_f2 tw[3], te[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        tw[r] = wxa * m[r][0] + lxa.l * m[r][1];
        te[r] = wxb * (dx ? m[r][1] : m[r][0]) + lxb.l * (dx ? m[r][2] : m[r][1]);
    }
    const prep_f2 pnw = (wya * tw[0] + lya.l * tw[1]) * a.mul;
    const prep_f2 pne = (wya * te[0] + lya.l * te[1]) * a.mul;
    const prep_f2 psw = (wyb * (dy ? tw[1] : tw[0]) + lyb.l * (dy ? tw[2] : tw[1])) * a.mul;
    const prep_f2 pse = (wyb * (dy ? te[1] : te[0]) + lyb.l * (dy ? te[2] : te[1])) * a.mul;
    prep_f2 v = {0.0f, 0.0f};
    v += (pnw * xs) * tp.wnw;
    v += (pne * xs) * tp.wne;
    v += (psw * xs) * tp.wsw;
    v += (pse * xs) * tp.wse;
    return v;
}

// Second half of the prep work at ONE full-resolution pixel (fLDRnet.py:474-479): from the upsampled flows f10 / f01 at the pixel,
// flowback_0 = bwarp(t * flow_10, (1-t) * flow_01), flowback_1 = bwarp((1-t) * flow_01, t * flow_10) (bwarp_kernel with scales) and
// the backward-warped frames im0_tot = bwarp(I0, flowback_0), im1_tot = bwarp(I1, flowback_1) (bwarp_kernel).
// i0 / i1: sample base pointers (wave-uniform), c-strides from `a`.
struct PrepP2 { prep_f2 fb0, fb1; float im0[3], im1[3]; };
template <bool P2, bool WIN, class A>
__device__ __forceinline__ PrepP2 prep_phase2_pixel(const A& a, float fpx, float fpy, prep_f2 f10, prep_f2 f01, const float2* __restrict__ lo10,
                                                    const float2* __restrict__ lo01, const float* __restrict__ i0, const float* __restrict__ i1,
                                                    float tv, float omt, prep_lds_t win0, prep_lds_t win1, int lane, bool& fail) {
#pragma clang fp contract(off)
    PrepP2 r;
    const prep_f2 s01 = omt * f01, s10 = tv * f10;
    const PrepTap tb0 = prep_tap(fpx, fpy, s01.x, s01.y, a);
    const PrepTap tb1 = prep_tap(fpx, fpy, s10.x, s10.y, a);
    const float mb0 = a.withmask ? prep_tap_mask(tb0) : 1.0f, mb1 = a.withmask ? prep_tap_mask(tb1) : 1.0f;
    r.fb0 = prep_sample_up2<P2, WIN>(tb0, lo10, a, tv, win0, lane, fail) * mb0;
    r.fb1 = prep_sample_up2<P2, WIN>(tb1, lo01, a, omt, win1, lane, fail) * mb1;
    if (WIN && fail) return r;                                            // wave-uniform
    const PrepTap ti0 = prep_tap(fpx, fpy, r.fb0.x, r.fb0.y, a);
    const PrepTap ti1 = prep_tap(fpx, fpy, r.fb1.x, r.fb1.y, a);
    const float mi0 = a.withmask ? prep_tap_mask(ti0) : 1.0f, mi1 = a.withmask ? prep_tap_mask(ti1) : 1.0f;
    PrepCorners g0[3], g1[3];                                               // all 24 gathers in flight together
#pragma unroll
    for (int c = 0; c < 3; ++c) { g0[c] = prep_tap_gather(ti0, i0 + (int64_t)c * a.i0_cstride); g1[c] = prep_tap_gather(ti1, i1 + (int64_t)c * a.i1_cstride); }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        r.im0[c] = prep_tap_blend(ti0, g0[c]) * mi0;
        r.im1[c] = prep_tap_blend(ti1, g1[c]) * mi1;
    }
    return r;
}
