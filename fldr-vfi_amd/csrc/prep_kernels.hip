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
#include "prep_device.h"

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

// The 16 output planes (566 MB at 4K, read back by the splats / enc1 / dec3 only after hundreds of MB of other traffic) are
// stored with the streaming hint so that they do not displace what the Infinity Cache can actually keep (the packed
// activations the convolutions hand to each other, enc1's output for enc2): +0.6-1.0 % frame pairs/s, A/B on one box.
#ifndef PREP_NT
#define PREP_NT 1
#endif
__device__ __forceinline__ void prep_stf(float* __restrict__ base, uint32_t boff, float v) {
#if defined(PREP_ABLATE) && PREP_ABLATE == 1            // diagnostic build: no plane stores (the values stay live through an opaque use)
    asm volatile("" :: "v"(v), "v"(boff), "s"(base));
    return;
#endif
#if defined(PREP_ABLATE) && PREP_ABLATE == 4            // diagnostic build: every store instruction issued, all into the first 64 KB of the plane (no HBM write stream)
    boff &= 0xFFFCu;
#endif
#if PREP_NT
    __builtin_nontemporal_store(v, reinterpret_cast<float*>(reinterpret_cast<char*>(base) + boff));
#else
    *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + boff) = v;
#endif
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
// one-phase launch has the registers (and with them the waves in flight: the kernel waits on chains of dependent gathers) of its phase.
// P2: both scales are exact powers of two (the model's x8 upsampling): integer source-index arithmetic (prep_lin_in).
// A wave covers 64 pixels of ONE row: everything that depends on the row only (its source rows and weight, its byte offset) is
// wave-uniform and lives in scalar registers.
// PREP_TAP_WINDOWS (default 1): the 3 x 3 low-resolution neighbourhoods of the two backward-flow taps through the wave's tap windows
// (round 5: -15 us).  THE LIBRARY IS COMPILED WITHOUT PACKED-FP32 INSTRUCTIONS (csrc/hipcc_flags.rsp: -target-feature -packed-fp32-ops) because of this kernel:
// with the windows hipcc 7.2 forms `v_pk_add_f32 v[36:37], v[36:37], v[4:5] op_sel:[0,1]` for the taps' y coordinates (fb.y + fpy, the row as the
// high register of the (fpx, fpy) pair), and on gfx950 that operand form — a packed fp32 add / mul / fma whose first vector-register source is read
// straight and whose second through op_sel — returns low half = src0 + 0 in lanes 48-63 now and then while waves of ANOTHER kernel issue matrix
// instructions on the same SIMD: im1_tot came out as runs of 16 zero pixels whenever enc1 / dec1 / dec0 of another stream shared the CUs (1-6 wrong
// frames in 12 with three pairs in flight).  Found by bench.py's deferred replay check, traced to the instruction with probes edited into the
// kernel's assembly (tools/asm_pad_variant.py, tools/asm_edits/), reproduced stand-alone (tools/ubench/pk_opsel_probe.hip: 7 of 25 operand forms,
// only beside matrix instructions, only lanes 48-63, only the low half): profiles/r06_prep_concurrency.txt, profiles/r06_pk_opsel_probe.txt.
// The kernel is bound by its vector-memory address traffic, not by arithmetic: without packed instructions it takes the same time, and the whole
// forward is 1.4 % faster without them (same-box A/B).  Appending `-Xclang -target-feature -Xclang +packed-fp32-ops` rebuilds the failing kernel
// (tools/README.md).
// tools/check_pk_opsel.py (CPU test) keeps every kernel of the library free of the affected forms.
#ifndef PREP_TAP_WINDOWS
#define PREP_TAP_WINDOWS 1
#endif
#ifndef PREP_WPE
#define PREP_WPE 7                 // <= 72 registers: seven waves per SIMD (8 spills; 6 measured 2 % slower)
#endif
template <int PH, bool P2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PREP_WPE))) void level0_prep_kernel(PrepArgs a) {
#pragma clang fp contract(off)
    constexpr int WAVE_LDS = PREP_TAP_WINDOWS ? PREP_WAVE_LDS : 2 * PREP_QW * 16;      // (without the tap windows: the quad window alone, 640 B per wave)
    __shared__ __attribute__((aligned(16))) unsigned char prep_lds[4 * WAVE_LDS];
    const int tx = threadIdx.x & 63, ty = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const prep_lds_t wq = (prep_lds_t)prep_lds + ty * WAVE_LDS;          // this wave's windows (prep_device.h): quad, tap 0, tap 1
    const prep_lds_t wu0 = wq + 2 * PREP_QW * 16;
    const prep_lds_t wu1 = wu0 + PREP_UW * PREP_UH * 8;
    const int px_raw = blockIdx.x * 64 + tx;
    const int py = blockIdx.y * 4 + ty;
    const int n = blockIdx.z;
    if (py >= a.H) return;                                               // wave-uniform
    // (the lanes of a partial wave beyond the image compute on its last column — the wave stages its low-resolution windows together —
    // and store nothing)
    const bool live = px_raw < a.W;
    const int px = live ? px_raw : a.W - 1;
    const int64_t HW = (int64_t)a.H * a.W, hw = (int64_t)a.h * a.w;
    constexpr bool ph1 = (PH & 1) != 0, ph2 = (PH & 2) != 0;
    const int64_t o1 = (int64_t)n * HW, o2 = (int64_t)n * 2 * HW, o3 = (int64_t)n * 3 * HW;
    float* const dst[PREP_NPL] = {
        a.z0 ? a.z0 + o1 : nullptr, a.z1 ? a.z1 + o1 : nullptr,
        a.flow_t0 + o2, a.flow_t0 + o2 + HW, a.flow_t1 + o2, a.flow_t1 + o2 + HW,
        a.flowback_0 + o2, a.flowback_0 + o2 + HW, a.flowback_1 + o2, a.flowback_1 + o2 + HW,
        a.im0_tot + o3, a.im0_tot + o3 + HW, a.im0_tot + o3 + 2 * HW, a.im1_tot + o3, a.im1_tot + o3 + HW, a.im1_tot + o3 + 2 * HW};
    {
        const uint32_t pixb = ((uint32_t)py * (uint32_t)a.W + (uint32_t)px) * 4u;      // byte offset of this pixel inside a plane
        auto put = [&](int plane, float v) __attribute__((always_inline)) {
            if (live) prep_stf(dst[plane], pixb, v);                     // 4-byte lanes, straight from the pixel's thread
        };
        const float2* lo10 = a.flow_lo2 + (int64_t)n * hw;       // flow_10 (x,y)
        const float2* lo01 = a.flow_lo2 + (int64_t)(a.N + n) * hw;   // flow_01 (x,y)
        const float* i0 = a.I0 + (int64_t)n * a.i0_bstride;
        const float* i1 = a.I1 + (int64_t)n * a.i1_bstride;
        const float tv = a.t[n], omt = 1.0f - tv;
        const float fpx = (float)px, fpy = (float)py;

        // the frames at this pixel (direct reads, issued first; only the splat metrics use them)
        float c0[3] = {0.0f, 0.0f, 0.0f}, c1[3] = {0.0f, 0.0f, 0.0f};
        if (ph1 && a.z0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { c0[c] = prep_ldf(i0 + (int64_t)c * a.i0_cstride, pixb); c1[c] = prep_ldf(i1 + (int64_t)c * a.i1_cstride, pixb); }
        }

        // upsampled flows at this pixel (fLDRnet.py:419-422); the row's source rows / weight are scalar; the neighbours come from the
        // wave's quad window (global loads where the window does not cover the wave: x2 upsampling, non-power-of-two scales)
        const PrepLin lx = prep_lin_in<P2>(px, a.sx, a.w, a.kx, a.rkx), ly = prep_lin_in<P2>(py, a.sy, a.h, a.ky, a.rky);
        PrepQuad q;
        if (!prep_quad_window(lo10, lo01, a.w, lx, ly, wq, tx, q)) q = prep_quad(lo10, lo01, a.w, lx, ly);
        const prep_f2 f10 = prep_up2<false>(q.a00, q.a01, q.a10, q.a11, lx.l, ly.l, a.mul, 1.0f);
        const prep_f2 f01 = prep_up2<false>(q.b00, q.b01, q.b10, q.b11, lx.l, ly.l, a.mul, 1.0f);

        // t-scaled forward flows (fLDRnet.py:404-405,419-422): upsampling of (t * flow_01_lo) and ((1-t) * flow_10_lo); first, so that
        // the eight low-resolution neighbours are dead before the gathers below
        if (ph1) {
            const prep_f2 ft0 = prep_up2<true>(q.b00, q.b01, q.b10, q.b11, lx.l, ly.l, a.mul, tv);
            const prep_f2 ft1 = prep_up2<true>(q.a00, q.a01, q.a10, q.a11, lx.l, ly.l, a.mul, omt);
            put(2, ft0.x); put(3, ft0.y); put(4, ft1.x); put(5, ft1.y);
        }
        // splat metrics (fLDRnet.py:442-446 = zmetric_kernel): z0 from I0 and bwarp(I1, flow_01); z1 from I1 and bwarp(I0, flow_10)
        if (ph1 && a.z0) {
            const PrepTap t0 = prep_tap(fpx, fpy, f01.x, f01.y, a);
            const PrepTap t1 = prep_tap(fpx, fpy, f10.x, f10.y, a);
            const float m0 = prep_tap_mask(t0), m1 = prep_tap_mask(t1);
            float acc0 = 0.0f, acc1 = 0.0f;
            PrepCorners g0[3], g1[3];                                       // all 24 gathers in flight together
#pragma unroll
            for (int c = 0; c < 3; ++c) { g0[c] = prep_tap_gather(t0, i1 + (int64_t)c * a.i1_cstride); g1[c] = prep_tap_gather(t1, i0 + (int64_t)c * a.i0_cstride); }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float w0 = prep_tap_blend(t0, g0[c]) * m0;
                const float w1 = prep_tap_blend(t1, g1[c]) * m1;
                acc0 += a.za0 * fabsf(c0[c] - w0);
                acc1 += a.za1 * fabsf(c1[c] - w1);
            }
            put(0, fldr_div_by(acc0, 3.0f, 1.0f / 3.0f));          // == acc0 / 3.0f (the mean over the 3 channels)
            put(1, fldr_div_by(acc1, 3.0f, 1.0f / 3.0f));
        }

        if (ph2) {
            // backward flows (fLDRnet.py:474-475) and backward-warped frames (:478-479)
            // (the taps' low-resolution neighbourhoods through the wave's windows; a wave whose flows are too incoherent for them redoes
            // its pixels on the global path)
            bool fail = false;
#if PREP_TAP_WINDOWS
            PrepP2 r = prep_phase2_pixel<P2, true>(a, fpx, fpy, f10, f01, lo10, lo01, i0, i1, tv, omt, wu0, wu1, tx, fail);
            if (fail) r = prep_phase2_pixel<P2, false>(a, fpx, fpy, f10, f01, lo10, lo01, i0, i1, tv, omt, wu0, wu1, tx, fail);
#else
            PrepP2 r = prep_phase2_pixel<P2, false>(a, fpx, fpy, f10, f01, lo10, lo01, i0, i1, tv, omt, wu0, wu1, tx, fail);
#endif
            put(6, r.fb0.x); put(7, r.fb0.y); put(8, r.fb1.x); put(9, r.fb1.y);
#pragma unroll
            for (int c = 0; c < 3; ++c) { put(10 + c, r.im0[c]); put(13 + c, r.im1[c]); }
        }
    }
}

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
    dim3 grid(fldr_cdiv(d->W, 64), fldr_cdiv(d->H, 4), d->N);
    const bool p2 = a.kx >= 0 && a.ky >= 0;                     // (then H == h << ky and W == w << kx: prep_lin_in's precondition)
    hipStream_t s = fldr_s(stream);
#define PREP_LAUNCH(PH) do { if (p2) hipLaunchKernelGGL((level0_prep_kernel<PH, true>), grid, dim3(256), 0, s, a); \
                             else hipLaunchKernelGGL((level0_prep_kernel<PH, false>), grid, dim3(256), 0, s, a); } while (0)
    if (a.phase == 1) PREP_LAUNCH(1);
    else if (a.phase == 2) PREP_LAUNCH(2);
    else PREP_LAUNCH(3);
#undef PREP_LAUNCH
    FLDR_LAUNCH_RET();
}
