// dec2 -> dec3 -> occlusion softmax / blend in ONE persistent kernel (round 5; fLDRnet.py:638-643, 511-524): dec2's output tile never
// leaves the CU.
//
// Why a producer / consumer workgroup and not "dec2 inside dec3's tile loop": dec3_synth_kernel (final_kernels.hip) runs three
// workgroups per CU so that one's tile fetch, another's matrix phases and a third's fp64 tail overlap; a fused workgroup needs dec2's
// inputs, its weights and the tile in LDS (> 90 KB): one per CU, its phases in series — no faster than the two kernels.  Here the
// overlap is INSIDE the workgroup: 768 threads = 3 waves per SIMD.  Waves 0-7 are the CONSUMER (dec3's four phase convolutions and the
// fp64 softmax / T / blend tail of tile k, read from an LDS tile buffer; wave c owns tile rows 2 (c >> 1), + 1 and the output rows of
// parity c & 1, so one lane = one half-resolution pixel = one output row of its 2 x 2 quad), waves 8-11 the PRODUCER (dec2 of tile
// k + 1 on the matrix cores into the other tile buffer, from input windows whose LDS-DMA was issued one tile earlier).  ONE workgroup
// barrier per tile:
//     period k   consumer: candidate row requested, phase convolutions + tail of tile k, frame stores
//                producer: DMA of tile k + 2's windows issued, dec2 of tile k + 1, vmcnt(0)
// Workgroups are persistent (one per CU, XCD-contiguous tile ranges): dec2's weights (28 KB of A operands) are fetched once per
// workgroup, dec3's table lives in the consumer waves' registers (32 VGPRs: the two phases of the wave's row parity).
//
// dec2 = ReLU(conv3x3(cat(nearest-x2(dec1) [32 ch, quarter resolution], enc1 [16 ch, half resolution])) + bias), 48 -> 16 channels, on
// the fp16 matrix cores with the 3 x fp16 split of the other convolutions (hi*hi + hi*lo + lo*hi, fp32 accumulation): per 16-pixel
// block D[co][px] += A[co][k] B[k][px] in 14 K-steps of v_mfma_f32_16x16x32_f16.  Steps 0-8: tap = step, lane group lg = dec1's 8-channel
// group lg — the tap is the same for the whole wave, so a B operand's LDS address is one of four per-block registers
// ((y + a) >> 1, (x + b) >> 1 for a, b in {0, 1}: the nearest upsampling is an index) plus an immediate: no address arithmetic in the
// loop.  Steps 9-13: enc1, lane group lg = (tap 2 (step - 9) + (lg >> 1), group lg & 1), one add per read (tap 9 has zero weights).
// B operands are fetched half a K-step (three blocks) ahead of the MFMAs that use them.  The tile the producer writes is the
// (8 + 2) x (32 + 2) pixels dec3 reads, in dec3's LDS layout (four planes of packed records), with exact zeros outside the image
// (dec3's zero padding).  Summation order differs from conv3x3_ring_kernel's: results agree with the two-kernel path to fp32
// accumulation rounding (tests: error vs fp64 torch, whole-model goldens).
#include "common.h"

#define D23_TH 8
#define D23_TW 32
#define D23_OW (D23_TW + 2)                   // dec2 tile incl. halo: 10 x 34
#define D23_OH (D23_TH + 2)
#define D23_OSLOTS (D23_OH * D23_OW)          // 340
#define D23_OPLANE 5632                       // bytes per plane of the tile buffer (340 slots of 16 B, padded: dec3's D3M_PLANE)
#define D23_EH 12                             // enc1 window (half resolution)
#define D23_EW 36
#define D23_EPLANE (D23_EH * D23_EW * 16)     // 6,912
#define D23_QH 6                              // dec1 window (quarter resolution)
#define D23_QW 18
#define D23_QPLANE (D23_QH * D23_QW * 16)     // 1,728
#define D23_KSTEPS 14                         // 9 dec1 taps + 5 enc1 tap pairs
#define D23_W2_BYTES (D23_KSTEPS * 2 * 1024)  // dec2 A operands: [step][hi, lo][lane][8 halves]
#define D23_HDR 16                            // floats before the dec2 operands: {1 / scale, scale, max |w|, 0, zero block (4 floats), ...}
#define D23_WIN (4 * D23_EPLANE + 8 * D23_QPLANE)   // one set of input windows: 41,472 B
#define D23_OFF_W2 0
#define D23_OFF_WIN (D23_OFF_W2 + D23_W2_BYTES)
#define D23_OFF_B (D23_OFF_WIN + 2 * D23_WIN)
#define D23_OFF_L (D23_OFF_B + 2 * 4 * D23_OPLANE)
#define D23_LWAVE (6 * 32 * 4)                // logits exchange per consumer wave: [co][32 pixels]
#define D23_LDS (D23_OFF_L + 8 * D23_LWAVE)   // 162,816 B
#define D23_THREADS 768
#ifndef D23_CPIECES
#define D23_CPIECES 6                         // window pieces per consumer wave (of 44; the producer waves share the rest)
#endif
#ifndef D23_EXP
#define D23_EXP d23_exp_nonpos
#endif
#ifndef D23_EXP_F32
#define D23_EXP_F32 1                         // softmax weights by the fp32 transcendental unit (see the tail of `consume`); 0: the fp64 exp below
#endif
#ifndef D23_NB
#define D23_NB 3                              // pixel blocks per fetch / MFMA unit of the producer (6 blocks per wave and K-step)
#endif
static_assert(D23_LDS <= 160 * 1024, "dec23: LDS");

typedef _Float16 d23_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 d23_h4 __attribute__((ext_vector_type(4)));
typedef float d23_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* d23_gptr_t;
typedef __attribute__((address_space(3))) void* d23_lptr_t;

#ifdef FLDR_STAMPS
// Diagnostic build only (tools/stamps): per-phase s_memtime sums of one workgroup's consumer wave 0 and producer wave 8.
#ifndef FLDR_STAMP_BLOCK
#define FLDR_STAMP_BLOCK 100
#endif
__device__ unsigned long long fldr_d23_stamp_buf[2 * 8];
__device__ unsigned long long fldr_d23_wave_buf[12 * 2];            // per wave of the stamped workgroup: {cycles waiting at the tile barrier, cycles of the tile loop}
FLDR_HOOK int fldr_debug_read_d23_wave_stamps(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fldr_d23_wave_buf), sizeof(unsigned long long) * 24);
}
#define D23_STAMP(var) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
FLDR_HOOK int fldr_debug_read_d23_stamps(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fldr_d23_stamp_buf), sizeof(unsigned long long) * 16);
}
#else
#define D23_STAMP(var)
#endif

struct D23Args {
    const unsigned char* dec1;     // packed [N][4][hi, lo][h4 * w4][16 B]
    const unsigned char* enc1;     // packed [N][2][hi, lo][h * w][16 B]
    const float* w2;               // fldr_dec23_prepack
    const float* bias2;            // [16]
    const float* w3;               // fldr_dec3_prepack_spk
    const float* bias3;            // [6]
    const float* cand[6];
    int64_t cand_bstride_b[6];     // batch / channel strides in bytes (a candidate may be a view: I0 / I1 are planes of the frame pair tensor)
    uint32_t cand_cstride_b[6];
    const float* t;                // [N]
    const float* poison;           // common.h: 0.0f, NaN once a ring wait expired (added to t: every frame written after the fault is NaN); + 64 bytes: the store sink of lanes without a pixel
    double T;
    void* out;                     // [N,3,H,W] fp64 or fp32, or the rounded 8-bit frame [N,3,Hc,Wc] (cropped; Wc even)
    int Hc, Wc;
    int N, H, W;                   // full resolution
    int tiles_x, per_sample, total, per_xcd, wgs_per_xcd;
};

// one 16-byte slot per lane from `g` (or the zero block) into 64 consecutive slots at `l`
__device__ __forceinline__ void d23_dma(const unsigned char* g, unsigned char* l) {
    __builtin_amdgcn_global_load_lds((d23_gptr_t)g, (d23_lptr_t)l, 16, 0, 0);
}
// a packed 16-byte record at LDS byte offset `off` (+ an immediate)
__device__ __forceinline__ d23_h8 d23_lds(const unsigned char* smem, int off) { return *reinterpret_cast<const d23_h8*>(smem + off); }

// workgroup barrier usable from wave-uniform branches (every wave of the workgroup executes the same number of them)
__device__ __forceinline__ void d23_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// 1 / x for a finite positive x of ordinary magnitude (a sum of softmax weights): the hardware estimate + two Newton steps (full fp64
// precision; no scaling / fix-up of the IEEE division sequence, whose special cases cannot occur here — a zero or non-finite x gives a
// non-finite result in both)
__device__ __forceinline__ double d23_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}

// exp(x) for x <= 0 (a softmax argument after the maximum is subtracted) to ~1e-13 relative — three orders below what the fp32 logits carry —
// in 16 instructions: n = round(x / ln 2), r = x - n ln 2 in two steps (|r| <= 0.347), the Taylor polynomial of degree 10 in Horner form
// (|r|^11 / 11! < 2.2e-13), scaled by 2^n (v_ldexp_f64: anything below the denormals becomes 0, as exp's own underflow).  The library
// exp costs ~24 with its range checks and its last two ulps.
__device__ __forceinline__ double d23_exp_nonpos(double x) {
    const double n = __builtin_rint(x * 1.4426950408889634074);
    double r = __builtin_fma(n, -6.93147180369123816490e-01, x);
    r = __builtin_fma(n, -1.90821492927058770002e-10, r);
    double p = 2.7557319223985890653e-07;                               // 1 / 10!
    p = __builtin_fma(p, r, 2.7557319223985892511e-06);                 // 1 / 9!
    p = __builtin_fma(p, r, 2.4801587301587301566e-05);                 // 1 / 8!
    p = __builtin_fma(p, r, 1.9841269841269841253e-04);                 // 1 / 7!
    p = __builtin_fma(p, r, 1.3888888888888889419e-03);                 // 1 / 6!
    p = __builtin_fma(p, r, 8.3333333333333332177e-03);                 // 1 / 5!
    p = __builtin_fma(p, r, 4.1666666666666664354e-02);                 // 1 / 4!
    p = __builtin_fma(p, r, 1.6666666666666665741e-01);                 // 1 / 3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)n);
}

template <int I, int N, typename F>
__device__ __forceinline__ void d23_static_for(F& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); d23_static_for<I + 1, N>(f); }
}

template <typename OUT>
__global__ __launch_bounds__(D23_THREADS) void dec23_synth_kernel(D23Args a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wv < 8;                                        // wave-uniform
    const int h = a.H >> 1, w = a.W >> 1, h4 = a.H >> 2, w4 = a.W >> 2;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int t_end = min((xcd + 1) * a.per_xcd, a.total);
    const int t_first = xcd * a.per_xcd + slot;
    if (t_first >= t_end) return;                                        // workgroup-uniform
    const int my_tiles = (t_end - t_first + a.wgs_per_xcd - 1) / a.wgs_per_xcd;
    const unsigned char* zero = reinterpret_cast<const unsigned char*>(a.w2) + 16;      // 16 zero bytes (header floats 4 .. 7)

    // ---- once per workgroup: dec2's A operands ----
    {
        const unsigned char* g2 = reinterpret_cast<const unsigned char*>(a.w2 + D23_HDR);
        for (int piece = wv; piece < D23_W2_BYTES / 1024; piece += D23_THREADS / 64) d23_dma(g2 + (piece * 64 + lane) * 16, smem + D23_OFF_W2 + piece * 1024);
    }
    auto tile_of = [&](int k, int& n, int& i0, int& j0) __attribute__((always_inline)) {
        const int lin = t_first + k * a.wgs_per_xcd;
        n = lin / a.per_sample;
        const int trem = lin - n * a.per_sample, tyi = trem / a.tiles_x;
        i0 = tyi * D23_TH; j0 = (trem - tyi * a.tiles_x) * D23_TW;
    };
    const int ln = lane & 15, lg = lane >> 4;
    // Arguments that are used at one place of the tile loop are read from the kernel-argument segment THERE (scalar loads, hits in the
    // constant cache): kept in registers across the loop they were ~40 SGPRs too many, spilled to vector lanes and read back with ~75
    // v_readlane per tile.  The empty asm makes the pointer opaque, so the loads cannot be hoisted out of the loop again.
    auto args = [&]() __attribute__((always_inline)) {
        auto kp = __builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        return (const __attribute__((address_space(4))) D23Args*)kp;
    };
    unsigned long long st6 = 0, sf = 0, st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, st5 = 0, sa = 0, sb = 0, sc = 0, sd = 0, se = 0;
    (void)st0; (void)st1; (void)st2; (void)st3; (void)st4; (void)st5; (void)sa; (void)sb; (void)sc; (void)sd; (void)se; (void)st6; (void)sf;

    // The inputs of tile k go into window set k & 1: enc1 rows i0 - 2 .. i0 + 9, columns j0 - 2 .. j0 + 33 (4 planes of 432 slots, 7 pieces of
    // 64 slots each, the last one overlapping); dec1 rows i0 / 2 - 1 .. + 4, columns j0 / 2 - 1 .. + 16 (8 planes of 108 slots, 2 pieces):
    // 44 pieces of 1 KB.  An LDS-DMA instruction holds its wave until every earlier vector-memory operation of the wave AND the piece
    // itself have returned (phase stamps: 600-1,300 cycles per piece in a wave with nothing else in flight, 2,800 behind the candidates'
    // loads), so only the producer waves — no other global traffic — use it, for 3 pieces each; the consumer waves fetch theirs (4 each)
    // with ordinary loads at the start of their period and write them to LDS at its end.
    struct Piece { const unsigned char* g; int lds; };
    // piece e of a tile's windows; `rc` = (row << 8 | column) of this lane's slot inside the window
    auto piece_rc = [&](int e, int ln_) __attribute__((always_inline)) -> int {
        if (e < 28) {
            const int i = e - ((e * 37) >> 8) * 7;                       // e % 7 for e < 28
            const int sl = (i < 6 ? i * 64 : D23_EH * D23_EW - 64) + ln_;
            const int r = (sl * 57) >> 11;                               // sl / 36 for sl < 432
            return (r << 8) | (sl - r * D23_EW);
        }
        const int sl = ((e & 1) ? D23_QH * D23_QW - 64 : 0) + ln_;
        const int r = (sl * 57) >> 10;                                   // sl / 18 for sl < 108
        return (r << 8) | (sl - r * D23_QW);
    };
    auto piece_of = [&](int e, int rc, int n, int i0, int j0, int set) __attribute__((always_inline)) -> Piece {
        Piece r;
        if (e < 28) {
            const int plane = (e * 37) >> 8, i = e - plane * 7;          // e / 7, e % 7 for e < 28
            const int piece = i < 6 ? i * 64 : D23_EH * D23_EW - 64;
            const int gy = i0 - 2 + (rc >> 8), gx = j0 - 2 + (rc & 255);
            const bool ok = gy >= 0 && gy < h && gx >= 0 && gx < w;
            const unsigned char* pl = args()->enc1 + ((int64_t)n * 4 + plane) * (int64_t)h * w * 16;
            r.g = ok ? pl + (uint32_t)(gy * w + gx) * 16u : zero;
            r.lds = D23_OFF_WIN + set * D23_WIN + plane * D23_EPLANE + piece * 16;
        } else {
            const int plane = (e - 28) >> 1;
            const int piece = (e & 1) ? D23_QH * D23_QW - 64 : 0;
            const int gy = (i0 >> 1) - 1 + (rc >> 8), gx = (j0 >> 1) - 1 + (rc & 255);
            const bool ok = gy >= 0 && gy < h4 && gx >= 0 && gx < w4;
            const unsigned char* pl = args()->dec1 + ((int64_t)n * 8 + plane) * (int64_t)h4 * w4 * 16;
            r.g = ok ? pl + (uint32_t)(gy * w4 + gx) * 16u : zero;
            r.lds = D23_OFF_WIN + set * D23_WIN + 4 * D23_EPLANE + plane * D23_QPLANE + piece * 16;
        }
        return r;
    };
    // The pieces of every later tile: with D23_CPIECES = 6 the consumer waves take all 44 (wave c: c + 8 j while < 44 — their period has
    // slack at the barrier, the producer's has none), otherwise the producer waves share the rest.  The slot coordinates are recomputed
    // per tile from an opaque copy of the lane number (a handful of vector instructions): left to itself the compiler hoists them and the
    // address arithmetic out of the tile loop into a dozen live registers — those were spilled, and every reload of a spill is an
    // s_waitcnt vmcnt(0) in the middle of the tile.
    constexpr int PREST = (44 - 8 * D23_CPIECES + 3) / 4;                // pieces per producer wave (0 when the consumers take all)
    const int my_e0 = consumer ? wv : 8 * D23_CPIECES + (wv - 8), my_de = consumer ? 8 : 4;
    auto stage_mine = [&](int k) __attribute__((always_inline)) {
        int n, i0, j0;
        tile_of(k, n, i0, j0);
        int ln_op = lane;
        asm volatile("" : "+v"(ln_op));                                  // (opaque: the slot arithmetic below stays inside the tile loop)
#pragma unroll
        for (int j = 0; j < D23_CPIECES; ++j) {
            const int e = my_e0 + my_de * j;
            if (e < 44 && (consumer || j < PREST)) {                     // wave-uniform
                const Piece pc = piece_of(e, piece_rc(e, ln_op), n, i0, j0, k & 1);
                d23_dma(pc.g, smem + pc.lds);
            }
        }
    };
    // tile 0's windows: every wave by LDS-DMA (nothing else is in flight yet)
    {
        int n, i0, j0;
        tile_of(0, n, i0, j0);
        for (int e = wv; e < 44; e += D23_THREADS / 64) {
            const Piece pc = piece_of(e, piece_rc(e, lane), n, i0, j0, 0);
            d23_dma(pc.g, smem + pc.lds);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                                  // vmcnt(0): my DMA pieces (weights, tile 0's windows)
    d23_barrier();

#ifndef D23_ABLATE
#define D23_ABLATE 0
#endif
    // The two roles are separate code paths (their register sets never overlap in one wave); both execute the same my_tiles + 2 barriers.
    if (consumer) {
    // =============================== CONSUMER (waves 0-7): dec3_synth_kernel's body on a tile buffer ===============================
        const int ra = wv & 1, r0 = 2 * ((wv >> 1) & 3);                    // output-row parity, first tile row of the wave
        const int tx = lane & 31, ty = r0 + (lane >> 5);
        const float inv_scale3 = a.w3[0];
        const int64_t HW = (int64_t)a.H * a.W;
        d23_h8 tab[2][2][2];                                                 // [pb][group][hi, lo] of phase 2 ra + pb
        {
            const unsigned char* g3 = reinterpret_cast<const unsigned char*>(a.w3 + 16);    // D3M_HDR floats, then [phase][group][hi, lo][lane][8 halves]
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) tab[pb][g][kk] = *reinterpret_cast<const d23_h8*>(g3 + ((((2 * ra + pb) * 2 + g) * 2 + kk) * 64 + lane) * 16);
        }
        float bias3_r[6];
#pragma unroll
        for (int co = 0; co < 6; ++co) bias3_r[co] = a.bias3[co];
        float2 cv[6][3];
        float acc3[2][6];
        auto consume = [&](int k) __attribute__((always_inline)) {
            int c_n, i0, j0;
            tile_of(k, c_n, i0, j0);
            const int c_li = i0 + ty, c_lj = j0 + tx;
            // the candidates' row 2 li + ra, columns 2 lj, 2 lj + 1: one 32-bit offset per channel, one scalar base per candidate; channels 0
            // and 1 are requested now, channel 2 after the matrix phase (its operand registers are free then)
            const int lic = min(c_li, h - 1), ljc = min(c_lj, w - 1);
            const uint32_t pob = (__umul24((uint32_t)(2 * lic + ra), (uint32_t)a.W) + (uint32_t)(2 * ljc)) * 4u;
            auto load_cands = [&](int ch) __attribute__((always_inline)) {
                const auto ap = args();
#pragma unroll
                for (int kc = 0; kc < 6; ++kc)
                    cv[kc][ch] = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(ap->cand[kc]) + (int64_t)c_n * ap->cand_bstride_b[kc] + (pob + (uint32_t)ch * ap->cand_cstride_b[kc]));
            };
            load_cands(0);
            load_cands(1);
            D23_STAMP(st6)
            const unsigned char* tb = smem + D23_OFF_B + (k & 1) * 4 * D23_OPLANE;
            float* lg_s = reinterpret_cast<float*>(smem + D23_OFF_L + wv * D23_LWAVE);     // [co][32 pixels of one tile row]
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {                         // one tile row = two 16-pixel blocks: 8 requests, then 4 independent MFMA chains
                    d23_h8 xh[2][2], xl[2][2];
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
                        const int sl = (r0 + rr + ra + (lg >> 1)) * D23_OW + 16 * qq + ln + pb + (lg & 1);
#pragma unroll
                        for (int g = 0; g < 2; ++g) {
                            xh[qq][g] = *reinterpret_cast<const d23_h8*>(tb + (2 * g) * D23_OPLANE + sl * 16);
                            xl[qq][g] = *reinterpret_cast<const d23_h8*>(tb + (2 * g + 1) * D23_OPLANE + sl * 16);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    d23_f4 c4[2][2];
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                        for (int g = 0; g < 2; ++g) c4[qq][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tab[pb][g][0], xh[qq][g], d23_f4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                        for (int g = 0; g < 2; ++g) c4[qq][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tab[pb][g][0], xl[qq][g], c4[qq][g], 0, 0, 0);
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                        for (int g = 0; g < 2; ++g) c4[qq][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tab[pb][g][1], xh[qq][g], c4[qq][g], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    // matrix columns -> the lanes that own the row's pixels (wave-scope exchange, no workgroup barrier)
                    if (lg < 2) {
#pragma unroll
                        for (int qq = 0; qq < 2; ++qq) {
                            const d23_f4 dd = c4[qq][0] + c4[qq][1];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int co = 4 * lg + r;
                                // (columns of channels 4, 5 — lane group 1 — are stored with their 16-pixel halves swapped: lane groups 0 and 1 would
                                // otherwise write channels c and c + 4 of the same pixel, 128 floats = the same bank apart, in one instruction)
                                if (co < 6) lg_s[co * 32 + ((16 * qq + ln) ^ ((co >> 2) << 4))] = dd[r];
                            }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if ((lane >> 5) == rr) {
#pragma unroll
                        for (int co = 0; co < 6; ++co) acc3[pb][co] = lg_s[co * 32 + (tx ^ ((co >> 2) << 4))] * inv_scale3 + bias3_r[co];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            }
            load_cands(2);
            D23_STAMP(st1)
            // fp64 softmax / T + blend of the quad's row `ra` (final_kernels.hip, same arithmetic).  The softmax half needs no candidate; by
            // its end the candidates have returned, and that is where this wave's LDS-DMA pieces of tile k + 2 go: an LDS-DMA instruction
            // waits for every earlier vector-memory operation of its wave, so here it costs its own round trip only.
            const bool live = c_li < h && c_lj < w && (sizeof(OUT) != 1 || (2 * c_li + ra < a.Hc && 2 * c_lj < a.Wc));
            // t and the fault poison (common.h) by SCALAR loads, spelled out: as ordinary loads of a may-alias pointer they became vector-memory
            // loads, and their first use — behind the LDS-DMA pieces the compiler sinks this code below — was an s_waitcnt vmcnt(0) that made
            // every consumer wave sit out the round trip of the pieces it had just issued
            float t_s, p_s;
            {
                const float* tp = args()->t + __builtin_amdgcn_readfirstlane(c_n);
                const float* pp = args()->poison;
                asm volatile("s_load_dword %0, %2, 0x0\n\ts_load_dword %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(t_s), "=&s"(p_s) : "s"(tp), "s"(pp) : "memory");
            }
            const float t = t_s + p_s;
            const double w1 = (double)t, w0 = (double)(1.0f - t);
            const double inv_T = 1.0 / a.T;
            // out = sum_k wo_k cand_k / sum_k wo_k with wo_k = w_k softmax_k (fLDRnet.py:517-524): the softmax's own normalisation cancels
            // in that quotient, so the weights stay unnormalised (w_k exp(s_k - max)) — one fp64 division and six products per pixel less
            // than the literal order; the quotient differs from it by rounding only (~1e-16 relative)
            // Instruction diet of the same quotient (every step changes the literal order's result by rounding only, ~1e-16 relative, against
            // a 3e-6 bound): the maximum is taken on the fp32 logits (x -> x / T is monotone), logit / T - max is one fma, and the blend below
            // is an fma chain.
            // Round 6: the softmax weights themselves are fp32 — exp2((logit - max) * (log2 e / T)) on the transcendental unit, converted once —
            // and everything they enter (the t weights, the normalisation, the blend, the quotient) stays fp64 as in the reference.  The
            // logits are fp32 accumulations of split-fp16 products (error ~1e-6 of their magnitude): a weight computed from them is
            // uncertain by ~1e-6 relative whatever the exp's precision, and the fp32 form is within 1e-7 (dominant weights) .. 1e-6 (weights
            // below 1e-5 of the sum) of the fp64 one — a frame difference of ~1e-7, against the 3e-6 bound of the kernel's test and the
            // reference's own fp32 convolutions in front of this tail.  It takes 12 x 18 double-precision instructions per lane out of the
            // ~430 that paced the consumer waves (D23_EXP_F32=0: the 16-instruction fp64 exp above).
            double wo[2][6], inv_div[2];
#if D23_EXP_F32
            const float sc2 = (float)(inv_T * 1.4426950408889634074);
#endif
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                float mx32 = acc3[rb][0];
#pragma unroll
                for (int kc = 1; kc < 6; ++kc) mx32 = fmaxf(mx32, acc3[rb][kc]);
#if D23_EXP_F32
#pragma unroll
                for (int kc = 0; kc < 6; ++kc) wo[rb][kc] = ((kc & 1) ? w1 : w0) * (double)__builtin_amdgcn_exp2f((acc3[rb][kc] - mx32) * sc2);
#else
                const double nmx = -((double)mx32 * inv_T);
#pragma unroll
                for (int kc = 0; kc < 6; ++kc) wo[rb][kc] = ((kc & 1) ? w1 : w0) * D23_EXP(__builtin_fma((double)acc3[rb][kc], inv_T, nmx));
#endif
                double div = ((wo[rb][0] + wo[rb][1]) + wo[rb][2]) + wo[rb][3];      // fLDRnet.py:517
                div = div + (wo[rb][4] + wo[rb][5]);                                // :522
                inv_div[rb] = d23_rcp(div);
            }
            D23_STAMP(st4)
            // (pinned between the candidates' last use above... and the frame stores below: the barrier's vmcnt(3) relies on exactly this order)
            // The candidates are waited for HERE, before the pieces go out (they have had the matrix phase and the softmax to return): the
            // opaque uses make the compiler place its wait for them now — behind the pieces it can only wait vmcnt(0), the number of pieces
            // being a run-time value, and every consumer wave then sat out its pieces' round trip in front of the blend.
            // (not in the fp32-output instantiation — tests only —, where the pinned live range costs five spilled registers: it keeps the
            // compiler's own vmcnt(0) behind the pieces)
            if constexpr (sizeof(OUT) != 4) {
#pragma unroll
                for (int kc = 0; kc < 6; ++kc)
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) { fldr_pin(cv[kc][ch].x); fldr_pin(cv[kc][ch].y); }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (k + 2 < my_tiles) stage_mine(k + 2);                        // into the window set produce(k) read before the last barrier
            __builtin_amdgcn_sched_barrier(0);
            D23_STAMP(st5)
            {
                // Every lane blends and stores: a lane without a pixel of the frame (border tiles, the crop of the 8-bit form) works on its
                // clamped candidates and stores into the library's 16-byte sink instead of the frame.  No divergent region, so every wave
                // issues EXACTLY three vector-memory instructions behind its DMA pieces on every path — what the barrier's vmcnt(3) counts on
                // (tools/check_dma_waits.py checks the listing: with the stores inside `if (live)` the compiler merged the two paths' pending
                // loads at the join and put an s_waitcnt vmcnt(0) in front of the counted wait: every tile waited for its frame stores).
#pragma clang fp contract(off)
                OUT* out = static_cast<OUT*>(args()->out);
                char* const sink = const_cast<char*>(reinterpret_cast<const char*>(args()->poison)) + 64;
                // channel by channel: blend both pixels of the row, store (two values live, not six)
                const uint32_t po8 = (uint32_t)(2 * c_li + ra) * (uint32_t)a.Wc + (uint32_t)(2 * c_lj);
                const int64_t po = (int64_t)(2 * c_li + ra) * a.W + 2 * c_lj;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    double res[2];
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb) {
                        double o = wo[rb][0] * (double)(rb ? cv[0][ch].y : cv[0][ch].x);       // :518-521
#pragma unroll
                        for (int kc = 1; kc < 6; ++kc) o = __builtin_fma(wo[rb][kc], (double)(rb ? cv[kc][ch].y : cv[kc][ch].x), o);
                        res[rb] = o * inv_div[rb];                              // :524
                    }
                    if constexpr (sizeof(OUT) == 1) {
                        // the rounded 8-bit frame (fldr_frame_metrics' arithmetic: utils.py:685-688, np.around), cropped to Hc x Wc: two pixels =
                        // one 16-bit store per channel — three stores behind the DMA pieces, as the frame's
                        unsigned q[2];
#pragma unroll
                        for (int rb = 0; rb < 2; ++rb) {
                            double v = (res[rb] + 1.0) / 2.0;
                            v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
                            q[rb] = (unsigned)(int)rint(v * 255.0);
                        }
                        unsigned char* o = reinterpret_cast<unsigned char*>(out) + ((int64_t)c_n * 3 + ch) * ((int64_t)a.Hc * a.Wc) + po8;
                        o = live ? o : reinterpret_cast<unsigned char*>(sink);
                        *reinterpret_cast<unsigned short*>(o) = (unsigned short)(q[0] | (q[1] << 8));
                    } else {
                        char* o = reinterpret_cast<char*>(out + ((int64_t)c_n * 3 + ch) * HW) + (uint32_t)po * (uint32_t)sizeof(OUT);
                        o = live ? o : sink;
                        if constexpr (sizeof(OUT) == 8) *reinterpret_cast<double2*>(o) = make_double2(res[0], res[1]);
                        else *reinterpret_cast<float2*>(o) = make_float2((float)res[0], (float)res[1]);
                    }
                }
            }
        };
        if (1 < my_tiles) stage_mine(1);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        d23_barrier();
        for (int k = 0; k < my_tiles; ++k) {
            D23_STAMP(st0)
            consume(k);
            D23_STAMP(st2)
            // my DMA pieces have landed when at most the three frame stores issued after them are outstanding (vmcnt counts in order): the
            // barrier does not wait for the stores
            asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            D23_STAMP(st3)
            sf += st6 - st0; sa += st1 - st0; se += st4 - st1; sd += st5 - st4; sb += st2 - st5; sc += st3 - st2;
        }
#ifdef FLDR_STAMPS
        if (blockIdx.x == FLDR_STAMP_BLOCK && lane == 0) { fldr_d23_wave_buf[wv * 2] = sc; fldr_d23_wave_buf[wv * 2 + 1] = sa + se + sd + sb + sc; }
        if (blockIdx.x == FLDR_STAMP_BLOCK && tid == 0) { fldr_d23_stamp_buf[0] = sa; fldr_d23_stamp_buf[1] = sb; fldr_d23_stamp_buf[2] = sc; fldr_d23_stamp_buf[3] = my_tiles; fldr_d23_stamp_buf[4] = sd; fldr_d23_stamp_buf[5] = se; fldr_d23_stamp_buf[6] = sf; }
#endif
    } else {
    // =============================== PRODUCER (waves 8-11): dec2 of one tile ===============================
        const int pw = wv - 8;                                               // producer wave 0 .. 3
        // Which tile pixel a lane's MFMA column is: the (8 + 2) x (32 + 2) tile splits by ROW parity into two sets of 5 x 17 pixel PAIRS
        // (cy, cx) -> pixels (2 cy + pyb, 2 cx), (2 cy + pyb, 2 cx + 1); block i of a wave serves row parity pyb = i & 1 (a compile-time
        // constant of the unrolled block loop), its 16 lanes = 8 consecutive pairs x the two columns of a pair (pxb = ln & 1): pair index
        // 8 (block >> 1) + (ln >> 1), 11 blocks per parity = 22 blocks (+ 2 of padding) over 4 waves.  The nearest-x2 index of a dec1 tap
        // (dy, dx) is (cy + ((pyb + dy) >> 1), cx + ((pxb + dx) >> 1)): the row part is an IMMEDIATE per block, the column part one register
        // per lane (qx) — one register per block (qr) + one per lane + an immediate.
        // Round 6: this replaces four parity classes per block (pyb = ln & 1, pxb = (ln >> 1) & 1, four pairs per block).  There the two
        // rows of a block's pixels lay 16 banks apart in the enc1 window (36-record rows) and 8 banks apart in the dec1 window for the odd
        // kernel row: a two-way conflict on three quarters of the producer's B-operand reads (22 % of the kernel's LDS-active cycles were
        // bank conflicts, and padding the rows needs 6 KB of LDS that do not exist).  Now the 16 lanes of a block read 16 consecutive records
        // of ONE row (256 contiguous bytes: every bank once) — except where the pair index wraps to the next row.  Same values per pixel:
        // the bits of dec2's tile are unchanged.
        const int pxb = ln & 1;
        const int qx = pxb * 16;
        int pyx[6], qr[6], eoff[6], soff[6];
        const bool etap1 = (lg >> 1) != 0;                                   // enc1 steps: this lane group's tap is 2 j + 1 (else 2 j)
        const int egrp = (lg & 1) * 2 * D23_EPLANE;
        float bias2_r[4];
        float inv_scale2 = 0.0f;
        {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int idx = min(((pw * 6 + i) >> 1) * 8 + (ln >> 1), 84);
                const int cy = (idx * 241) >> 12, cx = idx - cy * 17;    // idx / 17, idx % 17 for idx < 128
                const int py = 2 * cy + (i & 1), px = 2 * cx + pxb;      // (pw * 6 is even: the block's row parity is i & 1)
                pyx[i] = (py << 8) | px;
                eoff[i] = D23_OFF_WIN + (py * D23_EW + px) * 16;
                soff[i] = (lg >> 1) * 2 * D23_OPLANE + (py * D23_OW + px) * 16 + (lg & 1) * 8;     // this lane's piece of the pixel's record in a tile buffer
                qr[i] = D23_OFF_WIN + 4 * D23_EPLANE + 2 * lg * D23_QPLANE + (cy * D23_QW + cx) * 16;
            }
            inv_scale2 = a.w2[0];
#pragma unroll
            for (int r = 0; r < 4; ++r) bias2_r[r] = a.bias2[4 * lg + r];
        }
        // dec2 of tile k from window set k & 1 into tile buffer k & 1
        auto produce = [&](int k) __attribute__((always_inline)) {
            int n, i0, j0;
            tile_of(k, n, i0, j0);
            d23_f4 acc[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[i] = d23_f4{0.0f, 0.0f, 0.0f, 0.0f};
            // B operands are requested one unit (D23_NB blocks of one K-step) ahead of the MFMAs that use them, into register set u & 1
            constexpr int NB = D23_NB, UPS = 6 / NB, NU = UPS * D23_KSTEPS;
            d23_h8 bh[2][NB], bl[2][NB], ah[2], al[2];
            auto fetch = [&](auto uc) __attribute__((always_inline)) {
                constexpr int u = decltype(uc)::value, s = u / UPS, part = u % UPS, set = u & 1;
#pragma unroll
                for (int ib = 0; ib < NB; ++ib) {
                    const int i = NB * part + ib;
                    if constexpr (s < 9) {
                        constexpr int dy = s / 3, dx = s % 3;
                        const int row = (((i & 1) + dy) >> 1) * D23_QW * 16;     // (pyb + dy) >> 1 rows with pyb = i & 1: an immediate once unrolled
                        const int off = dx == 1 ? qr[i] + qx : qr[i];
                        bh[set][ib] = d23_lds(smem, off + row + (dx >> 1) * 16);
                        bl[set][ib] = d23_lds(smem, off + row + (dx >> 1) * 16 + D23_QPLANE);
                    } else {
                        constexpr int t0 = 2 * (s - 9), t1 = t0 + 1 < 9 ? t0 + 1 : 8;   // (tap 9 has zero weights: tap 8 is read again)
                        const int off = eoff[i] + egrp + (etap1 ? ((t1 / 3) * D23_EW + t1 % 3) * 16 : ((t0 / 3) * D23_EW + t0 % 3) * 16);
                        bh[set][ib] = d23_lds(smem, off);
                        bl[set][ib] = d23_lds(smem, off + D23_EPLANE);
                    }
                }
            };
            auto fetch_a = [&](auto sc) __attribute__((always_inline)) {
                constexpr int s = decltype(sc)::value;
                ah[s & 1] = d23_lds(smem, D23_OFF_W2 + (s * 2 + 0) * 1024 + lane * 16);
                al[s & 1] = d23_lds(smem, D23_OFF_W2 + (s * 2 + 1) * 1024 + lane * 16);
            };
            auto unit = [&](auto uc) __attribute__((always_inline)) {
                constexpr int u = decltype(uc)::value, s = u / UPS, part = u % UPS, set = u & 1;
                // the requests of the next unit go out after this one's first MFMA pass: whatever the compiler's wait before that pass covers
                // was issued 2 NB MFMAs earlier
#pragma unroll
                for (int ib = 0; ib < NB; ++ib) acc[NB * part + ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s & 1], bh[set][ib], acc[NB * part + ib], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (u + 1 < NU) fetch(std::integral_constant<int, u + 1>{});
                if constexpr (part == UPS - 1 && s + 1 < D23_KSTEPS) fetch_a(std::integral_constant<int, s + 1>{});
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ib = 0; ib < NB; ++ib) acc[NB * part + ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s & 1], bl[set][ib], acc[NB * part + ib], 0, 0, 0);
#pragma unroll
                for (int ib = 0; ib < NB; ++ib) acc[NB * part + ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[s & 1], bh[set][ib], acc[NB * part + ib], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            fetch_a(std::integral_constant<int, 0>{});
            fetch(std::integral_constant<int, 0>{});
            d23_static_for<0, NU>(unit);
            D23_STAMP(st2)
            // epilogue: scale, bias, ReLU, exact zeros outside the image (tiles on the frame's border only: wave-uniform test), split with ONE
            // range guard for the wave's 24 values, one 8-byte piece (4 channels) of each pixel's packed record
            unsigned char* tb = smem + D23_OFF_B + (k & 1) * 4 * D23_OPLANE;
            const bool border = i0 < 1 || i0 + D23_TH + 1 > h || j0 < 1 || j0 + D23_TW + 1 > w;     // the 10 x 34 pixels are not all inside the image
            float xs[24];
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) xs[4 * i + r] = fmaxf(acc[i][r] * inv_scale2 + bias2_r[r], 0.0f);
            if (border) {
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const int gy = i0 - 1 + (pyx[i] >> 8), gx = j0 - 1 + (pyx[i] & 255);
                    const bool inside = gy >= 0 && gy < h && gx >= 0 && gx < w;
#pragma unroll
                    for (int r = 0; r < 4; ++r) xs[4 * i + r] = inside ? xs[4 * i + r] : 0.0f;
                }
            }
            _Float16 hs[24], ls[24];
            bool bad = false;
            fldr_split_hl_group(xs, hs, ls, bad);
            fldr_note_range(bad);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                if (((pw * 6 + i) >> 1) * 8 + (ln >> 1) < 85) {
                    // channels 4 lg .. 4 lg + 3: group lg >> 1, second half of the record for odd lg; planes: [g0 hi, g0 lo, g1 hi, g1 lo]
                    unsigned char* d = tb + soff[i];
                    *reinterpret_cast<d23_h4*>(d) = d23_h4{hs[4 * i], hs[4 * i + 1], hs[4 * i + 2], hs[4 * i + 3]};
                    *reinterpret_cast<d23_h4*>(d + D23_OPLANE) = d23_h4{ls[4 * i], ls[4 * i + 1], ls[4 * i + 2], ls[4 * i + 3]};
                }
            }
            // the next tile reads the other window set
            const int delta = (k & 1) ? -D23_WIN : D23_WIN;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                eoff[i] += delta;
                qr[i] += delta;
            }
        };

        if (1 < my_tiles) stage_mine(1);
        if (!(D23_ABLATE & 16)) produce(0);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        d23_barrier();
        for (int k = 0; k < my_tiles; ++k) {
            D23_STAMP(st0)
            if (k + 2 < my_tiles) stage_mine(k + 2);                          // into the window set produce(k) read before the last barrier
            D23_STAMP(st1)
            st2 = st1;
            if (k + 1 < my_tiles && !(D23_ABLATE & 1)) produce(k + 1);
            D23_STAMP(st3)
            __builtin_amdgcn_s_waitcnt(0x0F70);
            D23_STAMP(st4)
            d23_barrier();
            D23_STAMP(st5)
            sa += st1 - st0; sb += st2 - st1; sc += st3 - st2; sd += st4 - st3; se += st5 - st4;
        }
#ifdef FLDR_STAMPS
        if (blockIdx.x == FLDR_STAMP_BLOCK && lane == 0) { fldr_d23_wave_buf[wv * 2] = se; fldr_d23_wave_buf[wv * 2 + 1] = sa + sb + sc + sd + se; }
        if (blockIdx.x == FLDR_STAMP_BLOCK && tid == 512) { unsigned long long* o = fldr_d23_stamp_buf + 8; o[0] = sa; o[1] = sb; o[2] = sc; o[3] = sd; o[4] = se; o[5] = my_tiles; }
#endif
    }
}

FLDR_TU_STATUS(dec23)

// dec2's weights [16, 48, 3, 3] as A operands: D23_HDR floats {1 / scale, scale, max |w|, 0, zero block ...} + [step][hi, lo][lane][8 halves],
// lane = (output channel = lane & 15, lane group lg = lane >> 4); steps 0 .. 8: tap = step, 8-channel group lg (dec1); steps 9 .. 13: tap =
// 2 (step - 9) + (lg >> 1), group 4 + (lg & 1) (enc1; tap 9: zeros); element j = input channel 8 group + j; scaled by a power of two into
// the fp16 range, split hi + lo.
__global__ void dec23_prepack_kernel(const float* __restrict__ w, float* __restrict__ wp) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    float m = 0.0f;
    for (int i = tid; i < 16 * 48 * 9; i += 256) m = fmaxf(m, fabsf(w[i]));
    red[tid] = m;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) red[tid] = fmaxf(red[tid], red[tid + st]);
        __syncthreads();
    }
    const float mx = red[0];
    float scale = 1.0f;                                                  // largest power of two with mx * scale <= 8192 (as spk_absmax_kernel)
    if (mx > 0.0f) scale = exp2f(floorf(log2f(8192.0f / mx)));
    if (tid < D23_HDR) wp[tid] = tid == 0 ? 1.0f / scale : (tid == 1 ? scale : (tid == 2 ? mx : 0.0f));
    d23_h8* frag = reinterpret_cast<d23_h8*>(wp + D23_HDR);
    for (int i = tid; i < D23_KSTEPS * 2 * 64; i += 256) {
        const int lane = i & 63, kind = (i >> 6) & 1, s = i >> 7;
        const int co = lane & 15, lg = lane >> 4;
        const int tap = s < 9 ? s : 2 * (s - 9) + (lg >> 1), g = s < 9 ? lg : 4 + (lg & 1);
        d23_h8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = tap < 9 ? w[((int64_t)co * 48 + g * 8 + j) * 9 + tap] * scale : 0.0f;
            const _Float16 hh = (_Float16)x;
            v[j] = kind == 0 ? hh : (_Float16)(x - (float)hh);
        }
        frag[i] = v;
    }
}

extern "C" int64_t fldr_dec23_prepack_size(void) { return D23_HDR + D23_W2_BYTES / 4; }

extern "C" int fldr_dec23_prepack(const float* dec2_weight, float* wpack, fldr_stream_t stream) {
    FLDR_CHECK_ARG(dec2_weight && wpack && (reinterpret_cast<uintptr_t>(wpack) & 15) == 0);
    hipLaunchKernelGGL(dec23_prepack_kernel, dim3(1), dim3(256), 0, fldr_s(stream), dec2_weight, wpack);
    FLDR_LAUNCH_RET();
}

extern "C" int fldr_dec23_synth(const void* dec1_spk, const void* enc1_spk, const float* w2pack, const float* bias2, const float* w3m, const float* bias3,
                                const float* const cand[6], const int64_t cand_bstride[6], const int64_t cand_cstride[6], const float* t, double T_param,
                                double* out_f64, float* out_f32, uint8_t* out_u8, int H_u8, int W_u8, int N, int H, int W, fldr_stream_t stream) {
    FLDR_CHECK_ARG(dec1_spk && enc1_spk && w2pack && bias2 && w3m && bias3 && cand && cand_bstride && cand_cstride && t && N > 0 && H > 0 && W > 0);
    FLDR_CHECK_ARG((out_f64 != nullptr) + (out_f32 != nullptr) + (out_u8 != nullptr) == 1);
    if (out_u8) {                                                        // the cropped 8-bit frame: pairs of pixels are stored together
        FLDR_CHECK_ARG(H_u8 > 0 && W_u8 > 0 && H_u8 <= H && W_u8 <= W && (reinterpret_cast<uintptr_t>(out_u8) & 1) == 0);
        if (W_u8 & 1) return FLDR_E_SHAPE;
        if ((int64_t)H_u8 * W_u8 >= (1ll << 31)) return FLDR_E_SHAPE;
    }
    if ((H | W) & 3) return FLDR_E_SHAPE;                                // dec1 lives at a quarter of the resolution
    if ((int64_t)H * W * 8 >= (1ll << 32)) return FLDR_E_SHAPE;          // 32-bit byte offsets inside a plane
    FLDR_CHECK_ARG(((reinterpret_cast<uintptr_t>(out_f64) & 15) | (reinterpret_cast<uintptr_t>(out_f32) & 7)) == 0);
    FLDR_CHECK_ARG(((reinterpret_cast<uintptr_t>(dec1_spk) | reinterpret_cast<uintptr_t>(enc1_spk) | reinterpret_cast<uintptr_t>(w2pack) | reinterpret_cast<uintptr_t>(w3m)) & 15) == 0);
    D23Args a;
    a.dec1 = static_cast<const unsigned char*>(dec1_spk); a.enc1 = static_cast<const unsigned char*>(enc1_spk);
    a.w2 = w2pack; a.bias2 = bias2; a.w3 = w3m; a.bias3 = bias3;
    for (int k = 0; k < 6; ++k) {
        FLDR_CHECK_ARG(cand[k] && (((uintptr_t)cand[k]) & 7) == 0 && (cand_bstride[k] & 1) == 0 && (cand_cstride[k] & 1) == 0);
        if (cand_cstride[k] < 0 || cand_bstride[k] < 0 || 2 * cand_cstride[k] * 4 + (int64_t)H * W * 4 >= (1ll << 32)) return FLDR_E_SHAPE;   // 32-bit offsets inside a sample
        a.cand[k] = cand[k]; a.cand_bstride_b[k] = cand_bstride[k] * 4; a.cand_cstride_b[k] = (uint32_t)(cand_cstride[k] * 4);
    }
    a.t = t; a.poison = fldr_status_poison_ptr(); a.T = T_param;
    if (!a.poison) return FLDR_E_STATUS;
    a.out = out_f64 ? static_cast<void*>(out_f64) : (out_f32 ? static_cast<void*>(out_f32) : static_cast<void*>(out_u8));
    a.Hc = out_u8 ? H_u8 : H; a.Wc = out_u8 ? W_u8 : W;
    a.N = N; a.H = H; a.W = W;
    a.tiles_x = fldr_cdiv(W / 2, D23_TW);
    a.per_sample = a.tiles_x * fldr_cdiv(H / 2, D23_TH);
    if ((int64_t)a.per_sample * N > (1ll << 30)) return FLDR_E_SHAPE;
    a.total = a.per_sample * N;
    a.per_xcd = (a.total + 7) / 8;
    a.wgs_per_xcd = a.per_xcd < 32 ? a.per_xcd : 32;                     // one workgroup per CU (159 KB of LDS)
    static std::atomic<uint64_t> attr64{0}, attr32{0}, attr8{0};
    if (out_f64) {
        if (int e = fldr_set_max_lds(reinterpret_cast<const void*>(&dec23_synth_kernel<double>), D23_LDS, attr64)) return e;
        hipLaunchKernelGGL((dec23_synth_kernel<double>), dim3(8 * a.wgs_per_xcd), dim3(D23_THREADS), D23_LDS, fldr_s(stream), a);
    } else if (out_f32) {
        if (int e = fldr_set_max_lds(reinterpret_cast<const void*>(&dec23_synth_kernel<float>), D23_LDS, attr32)) return e;
        hipLaunchKernelGGL((dec23_synth_kernel<float>), dim3(8 * a.wgs_per_xcd), dim3(D23_THREADS), D23_LDS, fldr_s(stream), a);
    } else {
        if (int e = fldr_set_max_lds(reinterpret_cast<const void*>(&dec23_synth_kernel<uint8_t>), D23_LDS, attr8)) return e;
        hipLaunchKernelGGL((dec23_synth_kernel<uint8_t>), dim3(8 * a.wgs_per_xcd), dim3(D23_THREADS), D23_LDS, fldr_s(stream), a);
    }
    FLDR_LAUNCH_RET();
}
