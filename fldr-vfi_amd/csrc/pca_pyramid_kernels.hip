// Low-dimensional feature projection of ALL pyramid levels in two launches (pca_comp.py:473-528 called once per level
// at fLDRnet.py:146): same arithmetic as pca_kernels.hip (fp64, (x - mean) * ev summed over the 64 pixels in row-major
// order, / meanvec, global min/max per level, ((y - min) / range) * 2 - 1, cast to fp32), bit-identical results.
//
// What was slow in the per-level kernels (rocprof, round 2 start: 383 us per 4K forward for 353 MB of algorithmic traffic):
//   * 18 launches (init + projection + rescale per level), five levels too small to fill the chip;
//   * the projection parked 106 MB of raw fp64 at level 0 and a streaming kernel read it back (and wrote the rescaled
//     fp64 in place although the model only consumes the fp32 cast and its split-packed twin);
//   * the coefficient matrix was read k-major: per (row, component) one 64-byte scalar load feeding EIGHT DEPENDENT
//     FMAs on one accumulator — the fp64 pipe ran latency-bound at ~45 % (ablation: 62 us with the pixel loads
//     compiled out, against ~28 us of issue time for 0.95 G fp64 operations at level 0).
// Here:
//   * a prepacked table holds the coefficients PIXEL-major (64 rows of K coefficients + the pixel's mean): one row of
//     scalar loads per pixel feeds K INDEPENDENT FMAs (one per accumulator), so the FMAs issue back to back;
//   * pass A (one launch over the blocks of all levels) only reduces min / max per level — no stores; pass B (one
//     launch) recomputes the projection (same code => same bits) and emits fp32 + split-packed directly.  The raw fp64
//     never touches memory: level 0 moves 2 x 212 MB in (the second read mostly from the Infinity Cache) + 106 MB out
//     instead of 212 + 106 + 106 in / out + 212 of rescale traffic;
//   * y / meanvec[k] and (y - min) / range divide by wave-uniform values: the quotient is formed by the Markstein
//     sequence q = y * r, e = fma(-q, c, y), q' = fma(e, r, q) on the correctly rounded reciprocal r = RN(1 / c),
//     which is the correctly rounded quotient (checked against true division: tests/test_host_cpu.py on exact rational
//     arithmetic, tests/test_gpu_parity.py bit-for-bit against pca_kernels.hip) — 3 FMAs instead of ~14 instructions
//     of which one quarter-rate.
#include "common.h"

#define PCAP_MAX_LEVELS 8
#ifndef PCAP_REVERSE
#define PCAP_REVERSE 1
#endif
#define PCAP_MM 16                      // doubles between the per-level bounds in the workspace: every bound on a 128-byte line of its own (all
                                        // 12 on one line serialised the ~25,000 atomics of the matrix-core pass A at one L2 channel: 139 vs 96 us)
#ifndef PCAP_H
#define PCAP_H 16                       // coefficients per scalar request (16: whole rows for K = 16, 64 cycles of FMAs per request)
#endif

struct PcapLevel {
    const float* planes;
    float* out32;
    unsigned char* spk;
    int32_t P, H, W;
    int32_t wg_start;                 // first workgroup of the level in the launch
    int64_t nblocks;                  // P * (H/8) * (W/8)
    int32_t item_start;               // matrix-core kernel: first 32-block item of the level
    int32_t wg_start_b, wg_start_r;   // first item of the level in the emit pass (levels with `raw` have none there) / in the rescale launch (the others have none)
    double* raw;                      // nblocks * K doubles or null: pass A parks the un-normalised projections here and a streaming launch rescales them
};
struct PcapArgs {
    PcapLevel lv[PCAP_MAX_LEVELS];
    const double* table;              // 64 rows {K coefficients, mean, 0}, meanvec[K], RN(1 / meanvec)[K] (fldr_pca_prepack)
    double* mm;                       // min of level l at [32 l], max at [32 l + 16]
    int32_t n_levels;
};

// Per-level min / max: ONE hardware fp64 atomic per workgroup and bound (global_atomic_min_f64 / max_f64, executed at the
// L2, nothing returned, nothing waited for).  The compare-and-swap loop of pca_kernels.hip pre-reads the current value
// through the per-CU L1, which is not coherent with the L2 where atomics execute: nearly every workgroup then saw a stale
// value, entered the loop, and 2 x 3,240 dependent same-line CAS round trips serialised — that, not HBM or the fp64 pipe,
// was most of the 125-148 us of a level-0 projection (the same kernel's emit pass, without atomics: 66 us).
__device__ __forceinline__ void pcap_atomic_min(double* addr, double v) {
    (void)__hip_atomic_fetch_min(addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void pcap_atomic_max(double* addr, double v) {
    (void)__hip_atomic_fetch_max(addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// x / c for a wave-uniform c with r = RN(1 / c): correctly rounded (Markstein), see the file header
__device__ __forceinline__ double pcap_div(double x, double c, double r) {
    const double q = x * r;
    const double e = fma(-q, c, x);
    return fma(e, r, q);
}

__global__ void pcap_init_kernel(double* mm, int n) {
    if ((int)threadIdx.x < n) { mm[PCAP_MM * 2 * threadIdx.x] = 1.0e300; mm[PCAP_MM * (2 * threadIdx.x + 1)] = -1.0e300; }
}

// table from the module's parameters (EV8 [K,64] k-major, Mean8 [64], meanVec8 [K]); one block of 64 x K threads
template <int K>
__global__ void pcap_prepack_kernel(const double* __restrict__ ev, const double* __restrict__ mean, const double* __restrict__ mv,
                                    double* __restrict__ tab) {
    const int i = threadIdx.x;                       // pixel
    for (int k = 0; k < K; ++k) tab[i * (K + 2) + k] = ev[k * 64 + i];
    tab[i * (K + 2) + K] = mean[i];
    tab[i * (K + 2) + K + 1] = 0.0;
    if (i < K) { tab[64 * (K + 2) + i] = mv[i]; tab[64 * (K + 2) + K + i] = 1.0 / mv[i]; }
}

// Pixels of one 8x8 block: 16 x dwordx4, all issued back to back (the consumer runs a whole block later).
__device__ __forceinline__ void pcap_load(const float* __restrict__ p, int W, float (&x)[64]) {
#if defined(PCAP_ABLATE) && PCAP_ABLATE == 1                           // diagnostic: no pixel traffic
    for (int i = 0; i < 64; ++i) asm volatile("v_mov_b32 %0, 1.0" : "=v"(x[i]));
    return;
#endif
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#if defined(PCAP_NT) && PCAP_NT
        typedef float pcap_f4 __attribute__((ext_vector_type(4)));
        const pcap_f4 a_ = __builtin_nontemporal_load(reinterpret_cast<const pcap_f4*>(p + (int64_t)i * W));
        const pcap_f4 b_ = __builtin_nontemporal_load(reinterpret_cast<const pcap_f4*>(p + (int64_t)i * W + 4));
        const float4 a = make_float4(a_[0], a_[1], a_[2], a_[3]), b = make_float4(b_[0], b_[1], b_[2], b_[3]);
#else
        const float4 a = *reinterpret_cast<const float4*>(p + (int64_t)i * W);
        const float4 b = *reinterpret_cast<const float4*>(p + (int64_t)i * W + 4);
#endif
        x[i * 8 + 0] = a.x; x[i * 8 + 1] = a.y; x[i * 8 + 2] = a.z; x[i * 8 + 3] = a.w;
        x[i * 8 + 4] = b.x; x[i * 8 + 5] = b.y; x[i * 8 + 6] = b.z; x[i * 8 + 7] = b.w;
    }
}

// y[k] of one 8x8 block, fp64: the accumulation order per component (pixels row-major) and the operations are those of
// pca_block in pca_kernels.hip.  Table (fldr_pca_prepack): 64 rows {K coefficients, mean, 0} of K + 2 doubles, then
// meanvec[K], RN(1 / meanvec)[K].
//
// The coefficient stream is hand-issued scalar loads in HALF rows of H = min(K, 8) coefficients, one half row ahead of the
// FMAs that consume it: {request half h+1; H FMAs with half h; s_waitcnt lgkmcnt(0)}.  Two half rows + the mean = 34
// SGPRs.  Left to the compiler, inside the item loop below, the 140 scalar loads of a block are all placed before the
// first FMA and parked in VGPR lanes (observed: 4,400 v_writelane / v_readlane per block against 1,024 FMAs), or become
// vector loads when the pointer is not provably un-aliased by the kernel's own stores.
typedef double pcap_d8 __attribute__((ext_vector_type(8)));
typedef double pcap_d4 __attribute__((ext_vector_type(4)));

// `after`: accumulators whose pending FMAs must be placed BEFORE this request (dummy register operands, not referenced by
// the instruction).  asm volatile statements keep their order among themselves, but the plain FMAs between them do not:
// without the operands the compiler moves all 128 request / wait pairs of a block in front of the first FMA.
template <int H> struct PcapHalf;
template <> struct PcapHalf<8> {
    pcap_d8 c;
    __device__ __forceinline__ void request(const double* tab, int byte_off, const double* after) {   // byte_off: compile-time after unrolling
        if (after) asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(c) : "s"(tab), "i"(byte_off), "v"(after[0]), "v"(after[1]), "v"(after[2]),
                                "v"(after[3]), "v"(after[4]), "v"(after[5]), "v"(after[6]), "v"(after[7]));
        else asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(c) : "s"(tab), "i"(byte_off));
    }
    __device__ __forceinline__ void arrived() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(c)); }
};
template <> struct PcapHalf<16> {                                    // a whole row of K = 16: two 16-dword tuples
    struct { pcap_d8 lo, hi; __device__ __forceinline__ double operator[](int k) const { return k < 8 ? lo[k] : hi[k - 8]; } } c;
    __device__ __forceinline__ void request(const double* tab, int byte_off, const double* after) {
        if (after) asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %4" : "=s"(c.lo), "=s"(c.hi) : "s"(tab), "i"(byte_off), "i"(byte_off + 64),
                                "v"(after[0]), "v"(after[1]), "v"(after[2]), "v"(after[3]), "v"(after[4]), "v"(after[5]), "v"(after[6]), "v"(after[7]),
                                "v"(after[8]), "v"(after[9]), "v"(after[10]), "v"(after[11]), "v"(after[12]), "v"(after[13]), "v"(after[14]), "v"(after[15]));
        else asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %4" : "=s"(c.lo), "=s"(c.hi) : "s"(tab), "i"(byte_off), "i"(byte_off + 64));
    }
    __device__ __forceinline__ void arrived() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(c.lo), "+s"(c.hi)); }
};
template <> struct PcapHalf<4> {
    pcap_d4 c;
    __device__ __forceinline__ void request(const double* tab, int byte_off, const double* after) {
        if (after) asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(c) : "s"(tab), "i"(byte_off), "v"(after[0]), "v"(after[1]), "v"(after[2]), "v"(after[3]));
        else asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(c) : "s"(tab), "i"(byte_off));
    }
    __device__ __forceinline__ void arrived() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(c)); }
};

template <int K>
__device__ __forceinline__ void pcap_project(const float (&x)[64], const double* tab, double (&y)[K]) {
    constexpr int H = PCAP_H < K ? PCAP_H : K, NH = K / H, ROW = (K + 2) * 8;      // coefficients per request; requests per pixel; bytes per table row
#pragma unroll
    for (int k = 0; k < K; ++k) y[k] = 0.0;
    PcapHalf<H> buf[2];
    double mean = 0.0, mean_next = 0.0;
    // prologue: half 0 of pixel 0 and its mean
    asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(mean) : "s"(tab), "i"(K * 8));
    buf[0].request(tab, 0, nullptr);
    buf[0].arrived();
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(mean));
#pragma unroll
    for (int h = 0; h < 64 * NH; ++h) {
        const int i = h / NH, j = h % NH;                               // pixel, half
        if (h + 1 < 64 * NH) {
            const int i1 = (h + 1) / NH, j1 = (h + 1) % NH;
            // ordered after the FMAs of half h-1 (they wrote y[jp*H ...]): keeps the stream one half row ahead, no more
            const int jp = (h + NH - 1) % NH;
            buf[(h + 1) & 1].request(tab, i1 * ROW + j1 * H * 8, h > 0 ? &y[jp * H] : nullptr);
            if (j1 == 0) asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(mean_next) : "s"(tab), "i"(i1 * ROW + K * 8));
        }
        const double d = (double)x[i] - mean;                           // pca_comp.py:502
#pragma unroll
        for (int k = 0; k < H; ++k) y[j * H + k] = fma(d, buf[h & 1].c[k], y[j * H + k]);   // :507
        if (h + 1 < 64 * NH) {
            buf[(h + 1) & 1].arrived();
            if ((h + 1) % NH == 0) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(mean_next)); mean = mean_next; }
        }
    }
    // meanvec and its reciprocals behind the rows
    const double* tail = tab + 64 * (K + 2);
#pragma unroll
    for (int j = 0; j < NH; ++j) {
        PcapHalf<H> mv, rmv;
        mv.request(tail, j * H * 8, nullptr);
        rmv.request(tail, (K + j * H) * 8, nullptr);
        mv.arrived();
        rmv.arrived();
#pragma unroll
        for (int k = 0; k < H; ++k) y[j * H + k] = pcap_div(y[j * H + k], mv.c[k], rmv.c[k]);   // :511
    }
}

// Work item = 256 consecutive blocks of one level (one block per thread); items are numbered level by level.
struct PcapWhere { int level; bool live; int p; int64_t pix, BHW, b, nb; const float* src; int W; double* raw; };   // b: block index in the level (0 for a dead lane) of nb

// WHICH: the item numbering — 0 pass A (every level), 1 emit pass (levels without parked projections), 2 rescale launch (levels with).
// A level without items in a numbering starts where the next one does and is never selected.
template <int WHICH>
__device__ __forceinline__ PcapWhere pcap_locate(const PcapArgs& a, int item) {
    PcapWhere w;
    auto start_of = [&](int l) { return WHICH == 0 ? a.lv[l].wg_start : (WHICH == 1 ? a.lv[l].wg_start_b : a.lv[l].wg_start_r); };
    w.level = 0;
    int start = start_of(0);
    w.raw = a.lv[0].raw;
#pragma unroll
    for (int l = 1; l < PCAP_MAX_LEVELS; ++l)
        if (l < a.n_levels && item >= start_of(l)) { w.level = l; start = start_of(l); w.raw = a.lv[l].raw; }   // workgroup-uniform
    const PcapLevel& L = a.lv[w.level];
    const int64_t b = (int64_t)(item - start) * 256 + threadIdx.x;
    w.live = b < L.nblocks;
    w.b = w.live ? b : 0; w.nb = L.nblocks;
    const int BW = L.W >> 3;
    w.BHW = (int64_t)(L.H >> 3) * BW;
    const int64_t bb = w.live ? b : 0;                                  // dead lanes read block 0 of the level
    w.p = (int)(bb / w.BHW);
    w.pix = bb - (int64_t)w.p * w.BHW;
    const int by = (int)(w.pix / BW), bx = (int)(w.pix - (int64_t)by * BW);
    w.W = L.W;
    w.src = L.planes + (int64_t)w.p * L.H * L.W + (int64_t)by * 8 * L.W + (int64_t)bx * 8;
    return w;
}

// Rescale and emit one block's K projections (pca_comp.py:523-526, fLDRnet.py:146): fp32 NCHW and / or the split-packed twin.
template <int K>
__device__ __forceinline__ void pcap_emit(const PcapArgs& a, const PcapWhere& w, const double (&y)[K]) {
#pragma clang fp contract(off)
    {
        {
            const PcapLevel& L = a.lv[w.level];
            const double mi = a.mm[PCAP_MM * 2 * w.level], range = a.mm[PCAP_MM * (2 * w.level + 1)] - mi;
            const double rr = 1.0 / range;                               // one true division per block; K Markstein quotients
            float f[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const double v = pcap_div(y[k] - mi, range, rr) * 2.0 - 1.0;    // pca_comp.py:523-526
                f[k] = (float)v;                                         // fLDRnet.py:146 .float()
            }
            if (L.out32) {
                float* o = L.out32 + (int64_t)w.p * K * w.BHW + w.pix;
#pragma unroll
                for (int k = 0; k < K; ++k) o[(int64_t)k * w.BHW] = f[k];
            }
            if (L.spk) {
                // channel c = p*K + k lives in group c >> 3, slot c & 7: [group][hi, lo][pixel][8 halves]
                if constexpr (K % 8 == 0) {
                    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#pragma unroll
                    for (int g = 0; g < K / 8; ++g) {
                        h8 vh, vl;
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const float t = __uint_as_float(__float_as_uint(f[g * 8 + k]) & 0xFFFFE000u);   // the split of conv_spk_kernels.hip
                            vh[k] = (_Float16)t;
                            vl[k] = (_Float16)(f[g * 8 + k] - t);
                        }
                        unsigned char* d = L.spk + (((int64_t)w.p * (K / 8) + g) * 2 * w.BHW + w.pix) * 16;
                        *reinterpret_cast<h8*>(d) = vh;
                        *reinterpret_cast<h8*>(d + w.BHW * 16) = vl;
                    }
                } else {
                    static_assert(K == 4, "split-packed output: K must be 4 or a multiple of 8");
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                    h4 vh, vl;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float t = __uint_as_float(__float_as_uint(f[k]) & 0xFFFFE000u);
                        vh[k] = (_Float16)t;
                        vl[k] = (_Float16)(f[k] - t);
                    }
                    unsigned char* d = L.spk + (((int64_t)(w.p >> 1)) * 2 * w.BHW + w.pix) * 16 + (w.p & 1) * 8;
                    *reinterpret_cast<h4*>(d) = vh;
                    *reinterpret_cast<h4*>(d + w.BHW * 16) = vl;
                    if ((L.P & 1) && w.p == L.P - 1) {                   // padding half of the last group: zeros, never uninitialised
                        const h4 z = {(_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0};
                        *reinterpret_cast<h4*>(d + 8) = z;
                        *reinterpret_cast<h4*>(d + w.BHW * 16 + 8) = z;
                    }
                }
            }
        }
    }
}

// Both passes as ONE persistent, double-buffered loop: a workgroup walks the items blockIdx.x, + gridDim.x, ... and the 16
// row loads of item i+1 are in flight while item i's ~1,200 fp64 operations run.  (One block per thread without the
// prefetch ran as lock-stepped generations — every resident wave loading, then every wave computing — and reached
// 107 us for the level-0 min/max pass against 58 us of arithmetic and 37 us of cold HBM read measured separately.)
//   EMIT = false: pass A, min / max per level (flushed with one hardware atomic pair whenever the level changes);
//   EMIT = true : pass B, recompute (same code => same bits), rescale, emit fp32 NCHW and / or the split-packed twin.
template <int K, bool EMIT>
__global__ __launch_bounds__(256, 2) void pcap_kernel(PcapArgs a, const double* __restrict__ table, int total_items) {
#pragma clang fp contract(off)
    __shared__ double slo[4], shi[4];
    float xa[64], xb[64];
    double lo = 1.0e300, hi = -1.0e300;
    int cur_level = -1;

    auto flush = [&]() {                                                // workgroup-uniform call sites only
        if (cur_level < 0) return;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {                        // wave64 shuffle reduction
            const double ol = __shfl_xor(lo, off), oh = __shfl_xor(hi, off);
            lo = ol < lo ? ol : lo; hi = oh > hi ? oh : hi;
        }
        __syncthreads();                                                // slo / shi of the previous flush have been read
        if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int i = 1; i < 4; ++i) { lo = slo[i] < lo ? slo[i] : lo; hi = shi[i] > hi ? shi[i] : hi; }
            pcap_atomic_min(a.mm + PCAP_MM * 2 * cur_level, lo);
            pcap_atomic_max(a.mm + PCAP_MM * (2 * cur_level + 1), hi);
        }
        lo = 1.0e300; hi = -1.0e300;
    };

    auto process = [&](const PcapWhere& w, const float (&x)[64]) {
        double y[K];
        // The table pointer is laundered once per block: its 1,100 scalar loads are loop-invariant, and hoisted out of the
        // item loop they would need ~2,200 SGPRs (observed: thousands of spills).
#if defined(PCAP_ABLATE) && PCAP_ABLATE == 2                           // diagnostic: no projection arithmetic
        for (int k = 0; k < K; ++k) y[k] = (double)(x[k] + x[k + 16] + x[k + 32] + x[k + 48]);
#else
        pcap_project<K>(x, table, y);
#endif
        if constexpr (!EMIT) {
            if (w.level != cur_level) { flush(); cur_level = w.level; }
            if (w.live) {
#pragma unroll
                for (int k = 0; k < K; ++k) { lo = y[k] < lo ? y[k] : lo; hi = y[k] > hi ? y[k] : hi; }
                // One read of the frames (w.raw, workgroup-uniform): y[] is parked and pcap_rescale_kernel streams it back instead of
                // 256 bytes of pixels + 64 K FMAs per block.
                if (w.raw != nullptr) {                                  // component-major: the lanes of a store write consecutive doubles
#pragma unroll
                    for (int k = 0; k < K; ++k) w.raw[(int64_t)k * w.nb + w.b] = y[k];
                }
            }
        } else {
            if (!w.live) return;
            pcap_emit<K>(a, w, y);
        }
    };

    const int stride = gridDim.x;
    int item = blockIdx.x;
    if (item >= total_items) return;
    // Pass B walks the items in the opposite direction: what pass A read LAST is still in the Infinity Cache (the 267 MB of a
    // 4K pyramid exceed its 256 MB, so a second scan in the same direction would miss everywhere).
    auto phys = [&](int i) { return (EMIT && PCAP_REVERSE) ? total_items - 1 - i : i; };
    PcapWhere wa = pcap_locate<EMIT ? 1 : 0>(a, phys(item)), wb = wa;
    pcap_load(wa.src, wa.W, xa);
    while (true) {
        int next = item + stride;                                        // workgroup-uniform control flow throughout
        if (next < total_items) { wb = pcap_locate<EMIT ? 1 : 0>(a, phys(next)); pcap_load(wb.src, wb.W, xb); }
        __builtin_amdgcn_sched_barrier(0);                               // the prefetch is issued before the arithmetic below
        process(wa, xa);
        if (next >= total_items) break;
        item = next;
        next = item + stride;
        if (next < total_items) { wa = pcap_locate<EMIT ? 1 : 0>(a, phys(next)); pcap_load(wa.src, wa.W, xa); }
        __builtin_amdgcn_sched_barrier(0);
        process(wb, xb);
        if (next >= total_items) break;
        item = next;
    }
    if constexpr (!EMIT) flush();
}

// The emit pass of the levels whose projections pass A parked (PcapLevel::raw, K planes of nblocks doubles): a streaming kernel — K
// doubles in, fp32 / split-packed out, no arithmetic beyond the rescale.  One block per thread,
// one 256-block item per workgroup, the items in REVERSE order (the tail of what pass A wrote is still in the Infinity Cache).
template <int K>
__global__ __launch_bounds__(256) void pcap_rescale_kernel(PcapArgs a, int total_items) {
#pragma clang fp contract(off)
    const PcapWhere w = pcap_locate<2>(a, total_items - 1 - (int)blockIdx.x);
    if (!w.live) return;
    double y[K];
#pragma unroll
    for (int k = 0; k < K; ++k) y[k] = w.raw[(int64_t)k * w.nb + w.b];
    pcap_emit<K>(a, w, y);
}

#ifdef FLDR_TEST_HOOKS                 // (the matrix-core formulation is a measured alternative, not a product path: test build only)
// ------------------------------------------------------------------------------------------------
// K = 16 on the fp64 matrix cores (opt-in: fldr_debug_pca_variant(1); the scalar-fed kernel above stays the default because it
// is bit-identical to the per-level kernels and this one is no faster — 196 vs 202 us per 4K pyramid, both passes within
// ~10 % of what their 267 + 400 MB cost at the 4.2-4.5 TB/s this access pattern streams at; pass A / B with the pixel loads
// compiled out: 65 / 62 us, without the matrix instructions: 75 / 80 us, without pass B's stores: 65 us).
//
// What the ablations of the kernel above say (4K pyramid, two passes = 200 us): 141 us with the pixel loads compiled out —
// the 1,100 scalar loads per block bound the fp64 pipe at half its rate — and 152 us with the arithmetic compiled out:
// one block per lane means 16-byte loads at a 32-byte lane stride, two instructions over the same lines (3.5 TB/s).
// Here the projection is the GEMM it is:  Y[16 components][16 blocks] += EV[16][4 pixels] * (X - mean)[4 pixels][16 blocks]
// as v_mfma_f64_16x16x4_f64, 16 steps per 8x8 block:
//   * A (coefficients, 16 steps x 1 double per lane) and the pixel means live in registers for the whole kernel — no scalar
//     stream; matrix row i' holds component 4 (i' % 4) + i' / 4, so that a lane's four results (rows q + 4 r,
//     MI355X_MICROARCH.md) are the components 4 q .. 4 q + 3: one 8-byte piece of the split-packed record;
//   * B: lane (block j = lane % 16, quarter q = lane / 16) supplies, for every pair of rows, the four pixels 4 (q & 1) .. + 3 of
//     row 2 r' + (q >> 1) (steps 4 r' .. 4 r' + 3): ONE 16-byte load per row pair and lane, and the 64 lanes of a load read
//     two runs of 512 contiguous bytes (16 adjacent blocks, two image rows) — the access width this chip streams
//     fastest (tools/ubench/plane_bw_bench: 5.5-5.9 TB/s against 3.9-4.1 for 4-byte lanes);
//   * a wave owns 32 consecutive blocks per item = two independent accumulation chains; the next item's 16 loads are in
//     flight while the 32 matrix instructions of the current one run.
// The summation order over the 64 pixels differs from the scalar kernels (and the matrix instruction's internal order is
// the hardware's): results agree with them to fp64 rounding (~1e-15 relative; the fp32 casts the model consumes are
// equal except where that moves a value across a rounding boundary) — tests/test_gpu_parity.py bounds both.
// ------------------------------------------------------------------------------------------------
typedef double pcam_d4 __attribute__((ext_vector_type(4)));

struct PcamWhere { int level; int p[2]; int64_t pix[2]; bool live[2]; const float* src[2]; };

__device__ __forceinline__ void pcam_locate(const PcapArgs& a, int item, int j, int q, PcamWhere& w) {
    w.level = 0;
#pragma unroll
    for (int l = 1; l < PCAP_MAX_LEVELS; ++l)
        if (l < a.n_levels && item >= a.lv[l].item_start) w.level = l;          // wave-uniform
    const PcapLevel& L = a.lv[w.level];
    const int BW = L.W >> 3;
    const int BHW = (L.H >> 3) * BW;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int64_t b = (int64_t)(item - L.item_start) * 32 + c * 16 + j;
        w.live[c] = b < L.nblocks;
        const uint32_t bb = w.live[c] ? (uint32_t)b : 0u;                        // dead lanes read block 0 of the level
        // (floor of a correctly rounded fp64 quotient of two integers < 2^31 is the integer quotient: twice per 32 matrix instructions)
#if defined(PCAM_ABLATE) && PCAM_ABLATE == 3                           // diagnostic: cheap (inexact) block decode
        const uint32_t p = (uint32_t)((float)bb * (1.0f / (float)BHW));
        const uint32_t pix = bb - p * (uint32_t)BHW;
        const uint32_t by = (uint32_t)((float)pix * (1.0f / (float)BW)), bx = pix - by * (uint32_t)BW;
#else
        const uint32_t p = (uint32_t)((double)bb / (double)BHW);
        const uint32_t pix = bb - p * (uint32_t)BHW;
        const uint32_t by = (uint32_t)((double)pix / (double)BW), bx = pix - by * (uint32_t)BW;
#endif
        w.p[c] = (int)p; w.pix[c] = pix;
        w.src[c] = L.planes + (int64_t)p * L.H * L.W + ((int64_t)by * 8 + (q >> 1)) * L.W + (int64_t)bx * 8 + 4 * (q & 1);
    }
}

__device__ __forceinline__ void pcam_load(const PcamWhere& w, int W, float4 (&x)[2][4]) {
#if defined(PCAM_ABLATE) && PCAM_ABLATE == 1                           // diagnostic: no pixel traffic
    for (int c = 0; c < 2; ++c) for (int r = 0; r < 4; ++r) { asm volatile("v_mov_b32 %0, 1.0" : "=v"(x[c][r].x)); x[c][r].y = x[c][r].z = x[c][r].w = x[c][r].x; }
    return;
#endif
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) x[c][r] = *reinterpret_cast<const float4*>(w.src[c] + (int64_t)(2 * r) * W);
}

template <bool EMIT>
__global__ __launch_bounds__(256, 2) void pcam_kernel(PcapArgs a, const double* __restrict__ table, int total_items) {
#pragma clang fp contract(off)
    constexpr int K = 16;
    const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
    const int n_waves = gridDim.x * 4;
    int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= total_items) return;                                            // wave-uniform; the kernel has no workgroup barrier

    // A operand and means: step s = 4 r' + t  <->  pixel (row 2 r' + (q >> 1), column 4 (q & 1) + t); matrix row i' = lane % 16 <-> component 4 (i' % 4) + i' / 4
    const int comp_a = 4 * (j & 3) + (j >> 2);
    double aev[16], bm[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int pix = 8 * (2 * (s >> 2) + (q >> 1)) + 4 * (q & 1) + (s & 3);
        aev[s] = table[pix * (K + 2) + comp_a];
        bm[s] = table[pix * (K + 2) + K];
    }
    double mvr[4], rmvr[4];                                                     // this lane's components 4 q + r
#pragma unroll
    for (int r = 0; r < 4; ++r) { mvr[r] = table[64 * (K + 2) + 4 * q + r]; rmvr[r] = table[64 * (K + 2) + K + 4 * q + r]; }

    double lo = 1.0e300, hi = -1.0e300;
    int cur_level = -1;
    auto flush = [&]() {                                                        // wave-uniform call sites only
        if (cur_level < 0) return;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ol = __shfl_xor(lo, off), oh = __shfl_xor(hi, off);
            lo = ol < lo ? ol : lo; hi = oh > hi ? oh : hi;
        }
        if (lane == 0) { pcap_atomic_min(a.mm + PCAP_MM * 2 * cur_level, lo); pcap_atomic_max(a.mm + PCAP_MM * (2 * cur_level + 1), hi); }
        lo = 1.0e300; hi = -1.0e300;
    };

    auto process = [&](const PcamWhere& w, const float4 (&x)[2][4]) {
        pcam_d4 acc[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float4 xq = x[c][s >> 2];
                const float xv = (s & 3) == 0 ? xq.x : ((s & 3) == 1 ? xq.y : ((s & 3) == 2 ? xq.z : xq.w));
                const double d = (double)xv - bm[s];                            // pca_comp.py:502
#if defined(PCAM_ABLATE) && PCAM_ABLATE == 2                           // diagnostic: no matrix instructions
                acc[c][s & 3] += d * aev[s];
#else
                acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(aev[s], d, acc[c], 0, 0, 0);   // :507
#endif
            }
        if constexpr (!EMIT) {
            if (w.level != cur_level) { flush(); cur_level = w.level; }
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if (w.live[c]) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double y = pcap_div(acc[c][r], mvr[r], rmvr[r]);  // :511
                        lo = y < lo ? y : lo; hi = y > hi ? y : hi;
                    }
                }
        } else {
            const PcapLevel& L = a.lv[w.level];
            const int64_t BHW = (int64_t)(L.H >> 3) * (L.W >> 3);
            const double mi = a.mm[PCAP_MM * 2 * w.level], range = a.mm[PCAP_MM * (2 * w.level + 1)] - mi;
            const double rr = 1.0 / range;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (!w.live[c]) continue;
                float f[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double y = pcap_div(acc[c][r], mvr[r], rmvr[r]);
                    f[r] = (float)(pcap_div(y - mi, range, rr) * 2.0 - 1.0);    // :523-526, fLDRnet.py:146 .float()
                }
#if defined(PCAM_ABLATE) && PCAM_ABLATE == 4                           // diagnostic: pass B stores nothing
                asm volatile("" :: "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]));
                continue;
#endif
                if (L.out32) {
                    float* o = L.out32 + ((int64_t)w.p[c] * K + 4 * q) * BHW + w.pix[c];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[(int64_t)r * BHW] = f[r];
                }
                if (L.spk) {
                    // channel p*16 + 4q + r: group 2p + (q >> 1), halves 4 (q & 1) .. + 3 of the pixel's 16-byte record
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                    h4 vh, vl;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float t = __uint_as_float(__float_as_uint(f[r]) & 0xFFFFE000u);   // the split of conv_spk_kernels.hip
                        vh[r] = (_Float16)t;
                        vl[r] = (_Float16)(f[r] - t);
                    }
                    unsigned char* d = L.spk + (((int64_t)w.p[c] * 2 + (q >> 1)) * 2 * BHW + w.pix[c]) * 16 + (q & 1) * 8;
                    *reinterpret_cast<h4*>(d) = vh;
                    *reinterpret_cast<h4*>(d + BHW * 16) = vl;
                }
            }
        }
    };

    PcamWhere wa, wb;
    float4 xa[2][4], xb[2][4];
    pcam_locate(a, item, j, q, wa);
    pcam_load(wa, a.lv[wa.level].W, xa);
    while (true) {
        int next = item + n_waves;                                              // wave-uniform control flow throughout
        if (next < total_items) { pcam_locate(a, next, j, q, wb); pcam_load(wb, a.lv[wb.level].W, xb); }
        __builtin_amdgcn_sched_barrier(0);                                       // the prefetch is issued before the arithmetic below
        process(wa, xa);
        if (next >= total_items) break;
        item = next;
        next = item + n_waves;
        if (next < total_items) { pcam_locate(a, next, j, q, wa); pcam_load(wa, a.lv[wa.level].W, xa); }
        __builtin_amdgcn_sched_barrier(0);
        process(wb, xb);
        if (next >= total_items) break;
        item = next;
    }
    if constexpr (!EMIT) flush();
}

static int g_pcap_variant = 0;                    // 1: fp64 matrix cores (K = 16), 0 (default): scalar-fed vector kernel
FLDR_HOOK int fldr_debug_pca_variant(int v) { if (v == 0 || v == 1) g_pcap_variant = v; return g_pcap_variant; }
#else
static const int g_pcap_variant = 0;
#endif  // FLDR_TEST_HOOKS

extern "C" int64_t fldr_pca_table_size(int K) {
    if (K != 4 && K != 8 && K != 16) return FLDR_E_ARG;
    return 64 * (K + 2) + 2 * K;
}

extern "C" int fldr_pca_prepack(const double* ev, const double* mean, const double* meanvec, double* table, int K,
                                fldr_stream_t stream) {
    FLDR_CHECK_ARG(ev && mean && meanvec && table);
    hipStream_t s = fldr_s(stream);
    switch (K) {
        case 16: hipLaunchKernelGGL(pcap_prepack_kernel<16>, dim3(1), dim3(64), 0, s, ev, mean, meanvec, table); break;
        case 8:  hipLaunchKernelGGL(pcap_prepack_kernel<8>, dim3(1), dim3(64), 0, s, ev, mean, meanvec, table); break;
        case 4:  hipLaunchKernelGGL(pcap_prepack_kernel<4>, dim3(1), dim3(64), 0, s, ev, mean, meanvec, table); break;
        default: return FLDR_E_ARG;
    }
    FLDR_LAUNCH_RET();
}

static int g_pcap_wgs = 512;                      // persistent workgroups (2 per CU: ~190 VGPRs per thread with both pixel buffers)
FLDR_HOOK int fldr_debug_pca_workgroups(int v) { if (v > 0) g_pcap_wgs = v; return g_pcap_wgs; }

template <int K>
static void pcap_launch(const PcapArgs& a, int total_items, int items_emit, int items_rescale, hipStream_t s) {
    const int grid = total_items < g_pcap_wgs ? total_items : g_pcap_wgs;
    hipLaunchKernelGGL(pcap_init_kernel, dim3(1), dim3(64), 0, s, a.mm, a.n_levels);
    hipLaunchKernelGGL((pcap_kernel<K, false>), dim3(grid), dim3(256), 0, s, a, a.table, total_items);
    // the levels with parked projections first: the streaming rescale starts on what pass A wrote last
    if (items_rescale > 0) hipLaunchKernelGGL((pcap_rescale_kernel<K>), dim3(items_rescale), dim3(256), 0, s, a, items_rescale);
    if (items_emit > 0) {
        const int grid_b = items_emit < g_pcap_wgs ? items_emit : g_pcap_wgs;
        hipLaunchKernelGGL((pcap_kernel<K, true>), dim3(grid_b), dim3(256), 0, s, a, a.table, items_emit);
    }
}

extern "C" int fldr_pca_project_pyramid(const fldr_pca_level* levels, int n_levels, const double* table, int K,
                                        double* minmax_ws, fldr_stream_t stream) {
    FLDR_CHECK_ARG(levels && table && minmax_ws && n_levels >= 1 && n_levels <= PCAP_MAX_LEVELS);
    PcapArgs a;
    int64_t wg = 0, items = 0, wg_b = 0, wg_r = 0;
    bool mfma_ok = true;
    for (int l = 0; l < n_levels; ++l) {
        const fldr_pca_level& in = levels[l];
        FLDR_CHECK_ARG(in.planes && (in.out_f32 || in.out_spk) && in.P > 0 && in.H > 0 && in.W > 0);
        if (in.H % 8 != 0 || in.W % 8 != 0) return FLDR_E_SHAPE;       // pca_comp.py:486-487
        if (((uintptr_t)in.planes & 15) != 0) return FLDR_E_ARG;       // dwordx4 row loads
        PcapLevel& L = a.lv[l];
        L.planes = in.planes; L.out32 = in.out_f32; L.spk = reinterpret_cast<unsigned char*>(in.out_spk);
        L.P = in.P; L.H = in.H; L.W = in.W;
        L.nblocks = (int64_t)in.P * (in.H / 8) * (in.W / 8);
        L.wg_start = (int)wg;
        wg += (L.nblocks + 255) / 256;
        if (wg >= (1ll << 30)) return FLDR_E_SHAPE;
        // (the matrix-core variant recomputes everywhere)
        L.raw = (K == 16 && g_pcap_variant == 1) ? nullptr : in.raw_ws;
        if (((uintptr_t)in.raw_ws & 15) != 0) return FLDR_E_ARG;
        L.wg_start_b = (int)wg_b; L.wg_start_r = (int)wg_r;
        (L.raw ? wg_r : wg_b) += (L.nblocks + 255) / 256;
        L.item_start = (int)items;
        items += (L.nblocks + 31) / 32;
        if (L.nblocks + 32 >= (1ll << 31) || items >= (1ll << 30)) mfma_ok = false;
    }
    for (int l = n_levels; l < PCAP_MAX_LEVELS; ++l) {
        a.lv[l] = a.lv[0];
        a.lv[l].wg_start = a.lv[l].wg_start_b = a.lv[l].wg_start_r = 0x7fffffff; a.lv[l].item_start = 0x7fffffff; a.lv[l].nblocks = 0;
    }
    a.table = table; a.mm = minmax_ws; a.n_levels = n_levels;
    hipStream_t s = fldr_s(stream);
#ifdef FLDR_TEST_HOOKS
    if (K == 16 && g_pcap_variant == 1 && mfma_ok) {
        const int waves = (int)items, wgs = (waves + 3) / 4;
        const int grid = wgs < g_pcap_wgs ? wgs : g_pcap_wgs;
        hipLaunchKernelGGL(pcap_init_kernel, dim3(1), dim3(64), 0, s, a.mm, a.n_levels);
        hipLaunchKernelGGL((pcam_kernel<false>), dim3(grid), dim3(256), 0, s, a, a.table, (int)items);
        hipLaunchKernelGGL((pcam_kernel<true>), dim3(grid), dim3(256), 0, s, a, a.table, (int)items);
        FLDR_LAUNCH_RET();
    }
#else
    (void)mfma_ok; (void)items;
#endif
    switch (K) {
        case 16: pcap_launch<16>(a, (int)wg, (int)wg_b, (int)wg_r, s); break;
        case 8:  pcap_launch<8>(a, (int)wg, (int)wg_b, (int)wg_r, s); break;
        case 4:  pcap_launch<4>(a, (int)wg, (int)wg_b, (int)wg_r, s); break;
        default: return FLDR_E_ARG;
    }
    FLDR_LAUNCH_RET();
}
