// Shared host/device helpers for libfldr_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "fldr_hip.h"

// Tuning / cross-check hooks (fldr_debug_*, include/fldr_hip_test_hooks.h) and the retired kernel generations that only serve as
// bit-exact cross-checks are compiled into the TEST build only (make hooks: -DFLDR_TEST_HOOKS -> libfldr_hip_test.so, loaded by
// the test suite through fldr_hip.test_hooks()).  The product library exports the integration ABI of include/fldr_hip.h and
// nothing else; in it the hook functions are file-local and unused, so the state they would change stays at its default.
#ifdef FLDR_TEST_HOOKS
#include "fldr_hip_test_hooks.h"             // the declarations carry the visibility of the test build's extra exports
#define FLDR_HOOK extern "C" FLDR_API
#else
#define FLDR_HOOK __attribute__((unused)) static
#endif

#define FLDR_CHECK_ARG(cond) do { if (!(cond)) return FLDR_E_ARG; } while (0)
#define FLDR_LAUNCH_RET() do { hipError_t e_ = hipGetLastError(); return e_ == hipSuccess ? 0 : (int)e_; } while (0)

static inline hipStream_t fldr_s(fldr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline int fldr_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device only: remember it per (kernel
// instantiation, device ordinal) — `done` is one function-local static bit mask per instantiation — so that a process
// driving several GPUs raises the limit on each of them.  Returns 0 or the hipError_t.
static inline int fldr_set_max_lds(const void* fn, int bytes, std::atomic<uint64_t>& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return 0;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    done.fetch_or(bit, std::memory_order_release);
    return 0;
}

// ---- fp16 hi/lo split with a range guard ---------------------------------------------------------------------------
// x = hi + lo, hi = x truncated to 11 significant bits (exact in fp16), lo = fp16(x - hi): the operand form of the
// 3 x fp16-split MFMA convolutions.  fp16 ends at 65504: beyond it a plain conversion gives hi = inf and the convolution
// NaN where the fp32 reference stays finite.  Guard: both halves are clamped to +-65504 (|x| < 131008 is then still
// represented, to fp16 precision of the excess: absolute error <= 16; beyond that — and for NaN inputs — the value
// SATURATES to a finite number) and the event is recorded in a sticky per-library flag that fldr_range_status()
// reports, so it is never silent: the accurate remedy for such data is FLDR_CONV_PRECISION=fp32.  In-range values produce the same
// bits as the unguarded split (2 v_med3 + 1 compare per element more).
#define FLDR_F16_MAX 65504.0f
static __device__ int fldr_tu_range_flag;                 // one per translation unit (read by fldr_tu_range_read below)
__device__ __forceinline__ void fldr_split_hl(float x, _Float16& hi, _Float16& lo, bool& bad) {
#ifdef FLDR_NO_RANGE_GUARD                               // A/B builds only (tools/build_lib_variant.sh): the unguarded split
    const float t0 = __uint_as_float(__float_as_uint(x) & 0xFFFFE000u);
    hi = (_Float16)t0; lo = (_Float16)(x - t0);
    return;
#endif
    bad |= !(fabsf(x) <= FLDR_F16_MAX);                  // overflow or NaN
    const float t = __builtin_amdgcn_fmed3f(__uint_as_float(__float_as_uint(x) & 0xFFFFE000u), -FLDR_F16_MAX, FLDR_F16_MAX);
    hi = (_Float16)t;
    lo = (_Float16)__builtin_amdgcn_fmed3f(x - t, -FLDR_F16_MAX, FLDR_F16_MAX);
}
// Pieces of the wave-uniform guard for callers that decide once for MANY values (a whole unit's epilogue): sum |x| while the values
// are produced, ask fldr_guard_trips once, then split with fldr_split_hl (tripped: exact per-value semantics, flag included) or
// fldr_split_plain (no lane's sum left the range, so every value is in range and the plain split gives the same bits).
__device__ __forceinline__ void fldr_split_plain(float x, _Float16& hi, _Float16& lo) {
    const float t0 = __uint_as_float(__float_as_uint(x) & 0xFFFFE000u);
    hi = (_Float16)t0; lo = (_Float16)(x - t0);
}
__device__ __forceinline__ bool fldr_guard_trips(float abs_sum) {        // wave-uniform; inf and NaN propagate through the sum
#if defined(FLDR_PER_VALUE_GUARD) || defined(FLDR_NO_RANGE_GUARD)
    return true;
#else
    return __builtin_expect(__builtin_amdgcn_ballot_w64(!(abs_sum <= FLDR_F16_MAX)) != 0ull, 0);
#endif
}
// The same split for a GROUP of values with the guard decided per WAVE: one |x| sum per lane (an add per value) and one compare per
// group instead of a compare and two clamps per value.  If no lane's sum exceeds the range (the sum bounds every |x|; inf and NaN
// propagate through it) every value is in range and the plain split gives the guarded split's bits; otherwise — rare: activations of
// this network are O(1..100) — the whole wave takes the guarded per-value path above, flag included.  Same results and the same flag
// semantics as fldr_split_hl in every case; ~2 vector instructions per value less (the step runs at the board's power limit, where
// instructions saved return as time: bench +2 % with the guards compiled out entirely, DESIGN_LOG round 4).
template <int N>
__device__ __forceinline__ void fldr_split_hl_group(const float (&x)[N], _Float16 (&hi)[N], _Float16 (&lo)[N], bool& bad) {
#if defined(FLDR_PER_VALUE_GUARD) || defined(FLDR_NO_RANGE_GUARD)
#pragma unroll
    for (int i = 0; i < N; ++i) fldr_split_hl(x[i], hi[i], lo[i], bad);
#else
    float s = fabsf(x[0]);
#pragma unroll
    for (int i = 1; i < N; ++i) s += fabsf(x[i]);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(s <= FLDR_F16_MAX)) != 0ull, 0)) {   // wave-uniform, rare
#pragma unroll
        for (int i = 0; i < N; ++i) fldr_split_hl(x[i], hi[i], lo[i], bad);
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float t0 = __uint_as_float(__float_as_uint(x[i]) & 0xFFFFE000u);
            hi[i] = (_Float16)t0; lo[i] = (_Float16)(x[i] - t0);
        }
    }
#endif
}
// ---- status a host can see WITHOUT synchronising (fldr_status_word, ABI 105) -----------------------------------------------------
// The sticky flags above / below live in device memory: reading them costs a device synchronisation, which a frame loop only pays at
// its end (fldr_range_status / fldr_ring_status).  A drop-in caller that never asks would ship saturated activations — or, after an
// expired ring wait, a frame computed from operands that never landed — unnoticed.  So every event is ALSO stored, system scope, into a
// block of pinned, mapped host memory (fldr_status_block: one word per kind, plain stores of 1 — no PCIe atomics needed) that the host
// polls at the start of its next forward, and an expired ring wait additionally sets a float in DEVICE memory to NaN (`poison`,
// 0.0f otherwise) that every frame-writing kernel adds to its blend weight t: each frame produced after the fault is NaN (8-bit
// form: black) until fldr_ring_status(reset = 1) — a never-checked run cannot ship a plausible wrong frame.
struct fldr_status_block { int range; int ring; int pad[14]; };
struct fldr_tu_status_t { fldr_status_block* host; float* poison; };
static __device__ fldr_tu_status_t fldr_tu_status;        // one per translation unit, bound by fldr_status_word() (null before)
// (explicit global-address-space pointers: through generic pointers these stores became flat_store instructions)
typedef __attribute__((address_space(1))) int* fldr_gint_t;
typedef __attribute__((address_space(1))) float* fldr_gfloat_t;
__device__ __forceinline__ void fldr_status_raise_range() {
#ifdef FLDR_NO_HOST_STATUS                               // A/B builds only: the range flag stays in device memory (what its cold path costs the kernels around it)
    return;
#endif
    fldr_status_block* h = fldr_tu_status.host;
    if (h) __hip_atomic_store((fldr_gint_t)&h->range, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void fldr_status_raise_ring() {
    fldr_status_block* h = fldr_tu_status.host;
    if (h) __hip_atomic_store((fldr_gint_t)&h->ring, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    float* p = fldr_tu_status.poison;
    if (p) *(fldr_gfloat_t)p = __builtin_nanf("");
}
__device__ __forceinline__ void fldr_note_range(bool bad) { if (bad) { fldr_tu_range_flag = 1; fldr_status_raise_range(); } }
static inline int fldr_tu_range_read(int reset) {
    int v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(fldr_tu_range_flag), sizeof(int)) != hipSuccess) return -1;
    if (v && reset) { const int z = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(fldr_tu_range_flag), &z, sizeof(int)) != hipSuccess) return -1; }
    return v;
}
static inline int fldr_tu_status_bind(fldr_status_block* host_dev, float* poison) {
    const fldr_tu_status_t v = {host_dev, poison};
    return hipMemcpyToSymbol(HIP_SYMBOL(fldr_tu_status), &v, sizeof(v)) == hipSuccess ? 0 : -1;
}
// translation units that split: each exports its flag and its binding of the status block through this pair (aggregated by
// fldr_range_status / fldr_status_word in conv_spk_kernels.hip)
#define FLDR_TU_STATUS(name)                                                                             \
    int fldr_range_read_##name(int reset) { return fldr_tu_range_read(reset); }                          \
    int fldr_status_bind_##name(fldr_status_block* h, float* p) { return fldr_tu_status_bind(h, p); }
#define FLDR_TU_STATUS_DECL(name) int fldr_range_read_##name(int reset); int fldr_status_bind_##name(fldr_status_block* h, float* p);
FLDR_TU_STATUS_DECL(spk) FLDR_TU_STATUS_DECL(ring) FLDR_TU_STATUS_DECL(conv) FLDR_TU_STATUS_DECL(s2) FLDR_TU_STATUS_DECL(split)
FLDR_TU_STATUS_DECL(warp) FLDR_TU_STATUS_DECL(gather) FLDR_TU_STATUS_DECL(acc64) FLDR_TU_STATUS_DECL(dec23)
int fldr_ring_timeouts_read(int reset);             // conv_ring_kernels.hip: expired ring waits (fldr_ring_status)
// Device pointer of the current device's poison float (see above) for the launchers of the frame-writing kernels; allocates and binds
// the status block on first use.  null on failure (the launcher then returns an error: a frame without its fault guard is not written).
const float* fldr_status_poison_ptr(void);

// Opaque use of a loaded value: stops LLVM from sinking an unconditional (clamped-address) load back into
// the select that consumes it, which would re-create `branch + load + s_waitcnt vmcnt(0)` per element.
// Call it on a whole batch of loaded values AFTER all loads of the batch have been written down.
__device__ __forceinline__ void fldr_pin(float& v) { asm volatile("" : "+v"(v)); }

// ---- bilinear sampling with F.grid_sample(align_corners=False, zeros) semantics ----------------
// Position arithmetic follows the op sequence of DCTVFInet.bwarp (fLDRnet.py:561-565) followed by
// PyTorch's unnormalisation ((g+1)*size/2 - 0.5) with FMA contraction disabled, so that sample
// positions agree with the fp32 reference to the last bit wherever possible: a 1-ulp change of the
// normalised coordinate is 1e-4 px at 4K and would show up at sharp edges and at the mask threshold.
struct FldrTap {
    int   x0, y0;            // north-west integer corner
    float wnw, wne, wsw, wse;
    bool  vnw, vne, vsw, vse; // corner in bounds
};

// x / c, correctly rounded, from the correctly rounded reciprocal rc = RN(1 / c) of a wave-uniform divisor: q = x * rc, one
// residual r = x - q * c (exact in an FMA) and one correction q + r * rc (Markstein).  Three full-rate instructions instead of
// the ~10 (one of them a quarter-rate v_rcp_f32) of the IEEE division expansion, whose scaling / fix-up steps only matter for
// denormal or overflowing quotients; checked against true division on 10^8 numerators for every divisor this path uses.
__device__ __forceinline__ float fldr_div_by(float x, float c, float rc) {
    const float q = x * rc;
    const float r = __builtin_fmaf(-q, c, x);
    return __builtin_fmaf(r, rc, q);
}

// rwm1 / rhm1: 1.0f / wm1, 1.0f / hm1 computed by a true fp32 division (host side, or once per thread)
__device__ __forceinline__ FldrTap fldr_grid_tap(float px, float py, float fx, float fy, int W, int H,
                                                 float wm1, float hm1, float rwm1, float rhm1) {
#pragma clang fp contract(off)
    FldrTap t;
    float vx = px + fx;
    float vy = py + fy;
    float gx = fldr_div_by(2.0f * vx, wm1, rwm1) - 1.0f;   // == (2 vx) / wm1: torch's div(Tensor, Scalar) is a true fp32 division
    float gy = fldr_div_by(2.0f * vy, hm1, rhm1) - 1.0f;
    float ix = (gx + 1.0f) * ((float)W * 0.5f) - 0.5f;
    float iy = (gy + 1.0f) * ((float)H * 0.5f) - 0.5f;
    float xf = floorf(ix), yf = floorf(iy);
    float w = ix - xf, e = 1.0f - w;
    float n = iy - yf, s = 1.0f - n;
    t.wnw = s * e; t.wne = s * w; t.wsw = n * e; t.wse = n * w;
    // clamp before the int conversion: wild flows must not overflow int
    xf = fminf(fmaxf(xf, -2.0f), (float)W + 1.0f);
    yf = fminf(fmaxf(yf, -2.0f), (float)H + 1.0f);
    t.x0 = (int)xf; t.y0 = (int)yf;
    bool x0v = t.x0 >= 0 && t.x0 < W, x1v = t.x0 + 1 >= 0 && t.x0 + 1 < W;
    bool y0v = t.y0 >= 0 && t.y0 < H, y1v = t.y0 + 1 >= 0 && t.y0 + 1 < H;
    t.vnw = x0v && y0v; t.vne = x1v && y0v; t.vsw = x0v && y1v; t.vse = x1v && y1v;
    return t;
}

__device__ __forceinline__ float fldr_tap_mask(const FldrTap& t) {
#pragma clang fp contract(off)
    // grid_sample of a ones tensor: sum of the in-bounds corner weights, accumulated nw,ne,sw,se
    float m = 0.0f;
    if (t.vnw) m += t.wnw;
    if (t.vne) m += t.wne;
    if (t.vsw) m += t.wsw;
    if (t.vse) m += t.wse;
    return m < 0.999f ? 0.0f : 1.0f;          // fLDRnet.py:573-574
}

__device__ __forceinline__ float fldr_tap_sample(const FldrTap& t, const float* __restrict__ plane, int W, int H) {
#pragma clang fp contract(off)
    // The four taps are fetched UNCONDITIONALLY from clamped addresses and masked afterwards: a predicated
    // load compiles to branch + load + s_waitcnt vmcnt(0), which serialises the gathers of a wave.
    const int xa = min(max(t.x0, 0), W - 1), xb = min(max(t.x0 + 1, 0), W - 1);
    const int ya = min(max(t.y0, 0), H - 1), yb = min(max(t.y0 + 1, 0), H - 1);
    const float* ra = plane + (int64_t)ya * W;
    const float* rb = plane + (int64_t)yb * W;
    float pnw = ra[xa], pne = ra[xb], psw = rb[xa], pse = rb[xb];
    fldr_pin(pnw); fldr_pin(pne); fldr_pin(psw); fldr_pin(pse);
    float v = 0.0f;
    v += t.vnw ? pnw * t.wnw : 0.0f;
    v += t.vne ? pne * t.wne : 0.0f;
    v += t.vsw ? psw * t.wsw : 0.0f;
    v += t.vse ? pse * t.wse : 0.0f;
    return v;
}

// A tap prepared for sampling several planes: the four CLAMPED corner positions as 32-bit byte offsets into a contiguous
// [H,W] fp32 plane (H*W*4 < 2^32, host-checked) and the corner weights with out-of-bounds corners zeroed.  Against a
// wave-uniform plane pointer each gather is then one global_load with an SGPR base and a VGPR offset — no per-load
// address arithmetic — and a sample is 4 multiplies + 4 adds.  Values are identical to fldr_tap_sample for finite
// planes: a masked corner contributes p * 0 = +-0 instead of a literal +0, which never changes a sum that starts at +0.
struct FldrTapP {
    uint32_t onw, one, osw, ose;
    float wnw, wne, wsw, wse;
};

__device__ __forceinline__ FldrTapP fldr_tap_prepare(const FldrTap& t, int W, int H) {
    FldrTapP p;
    const int xa = min(max(t.x0, 0), W - 1), xb = min(max(t.x0 + 1, 0), W - 1);
    const int ya = min(max(t.y0, 0), H - 1), yb = min(max(t.y0 + 1, 0), H - 1);
    const uint32_t ra = __umul24((uint32_t)ya, (uint32_t)W), rb = __umul24((uint32_t)yb, (uint32_t)W);   // full-rate 24-bit multiply (coordinates < 2^24)
    p.onw = (ra + (uint32_t)xa) * 4u; p.one = (ra + (uint32_t)xb) * 4u;
    p.osw = (rb + (uint32_t)xa) * 4u; p.ose = (rb + (uint32_t)xb) * 4u;
    p.wnw = t.vnw ? t.wnw : 0.0f; p.wne = t.vne ? t.wne : 0.0f;
    p.wsw = t.vsw ? t.wsw : 0.0f; p.wse = t.vse ? t.wse : 0.0f;
    return p;
}

// fldr_tap_mask from the masked weights (same sum: the skipped corners add +0)
__device__ __forceinline__ float fldr_tap_mask_p(const FldrTapP& p) {
#pragma clang fp contract(off)
    float m = 0.0f;
    m += p.wnw; m += p.wne; m += p.wsw; m += p.wse;
    return m < 0.999f ? 0.0f : 1.0f;
}

// plane: wave-uniform pointer to a contiguous [H,W] plane
__device__ __forceinline__ float fldr_tap_sample_p(const FldrTapP& p, const float* __restrict__ plane) {
#pragma clang fp contract(off)
    const char* b = reinterpret_cast<const char*>(plane);
    float pnw = *reinterpret_cast<const float*>(b + p.onw), pne = *reinterpret_cast<const float*>(b + p.one);
    float psw = *reinterpret_cast<const float*>(b + p.osw), pse = *reinterpret_cast<const float*>(b + p.ose);
    fldr_pin(pnw); fldr_pin(pne); fldr_pin(psw); fldr_pin(pse);
    float v = 0.0f;
    v += pnw * p.wnw;
    v += pne * p.wne;
    v += psw * p.wsw;
    v += pse * p.wse;
    return v;
}

// ---- F.interpolate(bilinear, align_corners=False) source index / lambda -------------------------
__device__ __forceinline__ void fldr_lin_src(int o, float scale, int in_size, int& i0, int& i1, float& l1) {
#pragma clang fp contract(off)
    float r = scale * ((float)o + 0.5f) - 0.5f;
    r = r < 0.0f ? 0.0f : r;
    int i = (int)r;
    i0 = i < in_size - 1 ? i : in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    float l = r - (float)i0;
    l1 = l < 0.0f ? 0.0f : (l > 1.0f ? 1.0f : l);
}
