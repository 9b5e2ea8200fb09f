// 3x3 / stride 1 / pad 1 convolutions on the fp16 matrix cores with fp32-equivalent accuracy ("3 x fp16 split").
//
// Every fp32 operand x is split into two halves  x = hi + lo,  hi = fp16(x), lo = fp16(x - hi)  (22 significant
// bits together) and each product is formed as  a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  by three
// v_mfma_f32_16x16x32_f16 with fp32 accumulation; the dropped a_lo*b_lo term is below 2^-22 relative.  Weights are
// pre-scaled by a power of two (undone exactly in the epilogue) so that their low halves stay normal.  Measured on
// MI355X (tools/ubench/mfma_f16_probe.hip, K = 864): mean |error| 6.8e-8 against fp64 versus 9.8e-8 for the exact
// v_mfma_f32_16x16x4_f32 chain and 7.6e-5 for plain fp16 inputs; fp16 MFMA inputs are not denormal-flushed.
// The f16 MFMA issues in 16 cycles for 16x16x32 (fp32 16x16x4: 32 cycles for 1/8 of the K depth), so the matrix
// time of a layer drops 16/3 = 5.3x; what remains is operand delivery, which this kernel organises as follows.
//
// Workgroup = 8 waves (two per SIMD, so one wave's LDS/VALU phases hide under the other's MFMAs) = an 8 x 32 output
// tile x MTOT = 16*NMT output channels; wave w owns output row w (two 16-pixel MFMA column tiles).  The input
// channels are processed in chunks of 16; one MFMA K-step (K = 32) covers TWO filter taps x 16 channels, so a chunk
// is 5 steps (9 taps + one zero-weight pad tap).  Per chunk, double buffered in LDS:
//   * input tile (10 x 34 pixels) as two planes of 16-byte elements [8 consecutive channels as halves] for hi and
//     for lo: a lane's B operand is ONE ds_read_b128 per plane; the channel-group stride is a multiple of 256 B,
//     which makes the b128 lane groups conflict free;
//   * weights already in MFMA A-operand order (prepacked on the device, hi and lo): a wave reads base + lane*16,
//     and the slab is filled by LDS-DMA (global_load_lds_dwordx4) with no registers involved;
//   * inputs are prefetched into registers during the previous chunk's MFMAs (multi-source concat, nearest-x2 and
//     zero padding resolved at load time), then split into hi/lo and written with two ds_write_b128.
#include "common.h"
#include <hip/hip_fp16.h>

#ifdef FLDR_STAMPS
// Diagnostic build only (tools/stamps): per-phase s_memtime sums of one workgroup, written to a buffer no kernel reads.
#ifndef FLDR_STAMP_BLOCK
#define FLDR_STAMP_BLOCK 0
#endif
__device__ unsigned long long fldr_split_stamp_buf[8 * 8];
#define SSTAMP(var) unsigned long long var; { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
FLDR_HOOK int fldr_debug_read_split_stamps(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fldr_split_stamp_buf), sizeof(unsigned long long) * 64);
}
#else
#define SSTAMP(var)
#endif

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* sgptr_t;
typedef __attribute__((address_space(3))) void* slptr_t;

#define SP_MAX_CIN 112
#define SP_TH 8
#define SP_TW 32
#define SP_IH (SP_TH + 2)
#define SP_IW (SP_TW + 2)
#define SP_CG_STRIDE 5632                       // bytes per 8-channel plane: 340 px * 16 B rounded up to a multiple of 256
#define SP_IN_BYTES (4 * SP_CG_STRIDE)          // hi[2 planes] + lo[2 planes]
#define SP_STEPS 5                              // tap pairs per 16-channel chunk (9 taps + 1 zero tap)
#define SP_HDR 4                                // floats before the packed weights: {1/scale, -, -, -}

struct SplitArgs {
    // per input channel (after concatenation) of sample 0: plane pointer, bit 0 = "stored at half resolution, read
    // through nearest x2"; 0 = padding channel.  Built on the host, read with scalar loads.
    unsigned long long chan[SP_MAX_CIN];
    int64_t chan_bstride[SP_MAX_CIN];
    const float* wpack;        // {header, halves...}
    const float* bias;
    const float* residual;
    float* out;
    int32_t cin, cout, cout_store;
    int32_t H, W;
    int32_t relu;
    int32_t tiles_x;
    int32_t groups;
    int32_t n_tiles, tiles_per_xcd;
};

template <int NMT>
struct SplitCfg {
    static constexpr int W_BYTES = SP_STEPS * NMT * 2 * 1024;           // one chunk of one group, hi + lo
    static constexpr int PIECES = W_BYTES / 16;                         // 16-B LDS-DMA pieces (a multiple of 64)
    static constexpr int NWI = (PIECES + 511) / 512;                    // sweeps of the 512-thread workgroup
    static constexpr int W_OFF = 0;                                     // 3-stage weight ring
    static constexpr int X_OFF = 3 * W_BYTES;                           // 2 input stages
    static constexpr int LDS_BYTES = X_OFF + 2 * SP_IN_BYTES;
    static_assert(PIECES % 64 == 0, "weight slab must be a whole number of wave-wide DMA pieces");
};

// x = hi + lo with hi = x truncated to 11 significant bits (exactly representable in fp16 for normal-range x, so its
// conversion is exact) and lo = fp16(x - hi): 3 VALU ops per element instead of 5-6 for the round-to-nearest split.
__device__ __forceinline__ void split8(const float (&x)[8], h8& hi, h8& lo, bool& bad) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        _Float16 h, l;
        fldr_split_hl(x[k], h, l, bad);                                  // the guarded split of common.h
        hi[k] = h; lo[k] = l;
    }
}

FLDR_TU_STATUS(split)

// TERMS = 3: hi*hi + hi*lo + lo*hi (fp32-equivalent); TERMS = 1: hi*hi only = plain fp16 inputs, fp32 accumulate
// (BASELINE config 5, "fp16 path with MFMA convs"; 11-bit operands, error ~1e-3 relative).
template <int NMT, int TERMS>
__global__ __launch_bounds__(512, 2) void conv3x3_split_kernel(SplitArgs a) {
    using Cfg = SplitCfg<NMT>;
    constexpr int MTOT = 16 * NMT;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n = blockIdx.y;
    // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2), so XCD x gets the
    // contiguous tile range [x*tpx, (x+1)*tpx) and all output-channel groups of a tile back to back: the groups
    // re-read the same input tile and vertically adjacent tiles share halo rows through that XCD's L2.
    const int xcd = blockIdx.x & 7, kx = blockIdx.x >> 3;
    const int grp = kx % a.groups;
    const int tile = xcd * a.tiles_per_xcd + kx / a.groups;
    if (tile >= a.n_tiles) return;                                    // workgroup-uniform
    const int cbase = grp * MTOT;
    const int oy0 = (tile / a.tiles_x) * SP_TH, ox0 = (tile % a.tiles_x) * SP_TW;
    const int lj = lane & 15, lg = lane >> 4;
    const int cin_pad = (a.cin + 15) / 16 * 16;
    const int n_chunks = cin_pad / 16;

    // staging geometry: waves 0-3 stage channels 0-7 of a chunk, waves 4-7 channels 8-15; 340 pixels over 256 threads
    const int scg = __builtin_amdgcn_readfirstlane(wave >> 2);      // wave-uniform 8-channel plane (in an SGPR)
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);                    // the younger wave of each SIMD loses MFMA arbitration otherwise
    const int st = tid & 255;
    int g_full[2], g_half[2], l_off[2];
    bool s_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = st + 256 * i;
        const int y = e / SP_IW, x = e % SP_IW;
        const int gy = oy0 - 1 + y, gx = ox0 - 1 + x;
        s_ok[i] = e < SP_IH * SP_IW && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        g_full[i] = s_ok[i] ? gy * a.W + gx : 0;
        g_half[i] = s_ok[i] ? (gy >> 1) * (a.W >> 1) + (gx >> 1) : 0;
        l_off[i] = e < SP_IH * SP_IW ? scg * SP_CG_STRIDE + e * 16 : -1;
    }
    // operand geometry: lane (pixel lj, k-group lg): plane lg&1, tap of the pair lg>>1
    int boff[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) boff[p] = (lg & 1) * SP_CG_STRIDE + (wave * SP_IW + p * 16 + lj) * 16;
    const int tap_sel = lg >> 1;

    f4 acc[NMT][2];
#pragma unroll
    for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int p = 0; p < 2; ++p) acc[m][p] = f4{0.0f, 0.0f, 0.0f, 0.0f};

    float pre[2][8];
    const float* wsrc = a.wpack + SP_HDR + (int64_t)grp * n_chunks * (Cfg::W_BYTES / 4);

    auto issue_weights = [&](int chunk) {                       // -> weight ring stage chunk % 3
        const float* g = wsrc + (int64_t)chunk * (Cfg::W_BYTES / 4);
        unsigned char* stage = smem + Cfg::W_OFF + (chunk % 3) * Cfg::W_BYTES;
#pragma unroll
        for (int i = 0; i < Cfg::NWI; ++i) {
            const int piece = i * 512 + wave * 64;                                    // wave-uniform, x16 bytes
            if (piece < Cfg::PIECES)
                __builtin_amdgcn_global_load_lds((sgptr_t)(g + (piece + lane) * 4), (slptr_t)(stage + piece * 16), 16, 0, 0);
        }
    };
    const float* zero_word = a.wpack + 3;                          // header slot holding 0.0f
    auto load_inputs = [&](int chunk) {
        // Every load is unconditional: padding pixels and padding channels read a zero word instead of being masked
        // (a predicated load compiles to branch + load + s_waitcnt vmcnt(0)); bases are SGPRs, offsets precomputed.
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int cg = chunk * 16 + scg * 8 + k;                 // wave-uniform
            const unsigned long long e = a.chan[cg];
            const bool ok = e != 0ull;
            const bool up2 = (e & 1ull) != 0ull;
            const float* base = ok ? reinterpret_cast<const float*>(static_cast<uintptr_t>(e & ~1ull)) + (int64_t)n * a.chan_bstride[cg]
                                   : zero_word;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float* p = s_ok[i] ? base + (ok ? (up2 ? g_half[i] : g_full[i]) : 0) : zero_word;
                pre[i][k] = *p;
            }
        }
    };
    bool range_bad = false;
    auto store_inputs = [&](int chunk) {                        // -> input stage chunk & 1
        unsigned char* stage = smem + Cfg::X_OFF + (chunk & 1) * SP_IN_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (l_off[i] < 0) continue;
            h8 hi, lo;
            split8(pre[i], hi, lo, range_bad);
            *reinterpret_cast<h8*>(stage + l_off[i]) = hi;
            *reinterpret_cast<h8*>(stage + 2 * SP_CG_STRIDE + l_off[i]) = lo;
        }
    };

    // Pipeline (per chunk k): MFMA steps on {weights ring k%3, input stage k&1}; before the last step: write the
    // registers prefetched for chunk k+1 to input stage (k+1)&1, THEN issue the LDS-DMA of chunk k+2's weights and
    // the register prefetch of chunk k+2's inputs.  The only vmcnt wait is the one in front of that LDS write (all
    // older loads were issued a full chunk earlier); the chunk barrier is a raw s_barrier behind lgkmcnt(0), so the
    // freshly issued DMA and loads stay in flight across it.
    issue_weights(0);
    load_inputs(0);
    store_inputs(0);
    if (n_chunks > 1) { issue_weights(1); load_inputs(1); }
    __syncthreads();                                   // (drains everything once; fine in the prologue)

#ifdef FLDR_STAMPS
    unsigned long long st_mfma = 0, st_stage = 0, st_last = 0, st_bar = 0;
    SSTAMP(tb0)
#endif
    for (int ch = 0; ch < n_chunks; ++ch) {
        SSTAMP(t0)
        const unsigned char* win = smem + Cfg::W_OFF + (ch % 3) * Cfg::W_BYTES + lane * 16;
        const unsigned char* xin = smem + Cfg::X_OFF + (ch & 1) * SP_IN_BYTES;
        h8 bh[2][2], bl[2][2], ah[2][NMT], al[2][NMT];
        auto ld = [&](int buf, int s) {
            // taps of step s: tA = 2s, tB = 2s+1 (tB = 9 is the zero-weight pad tap: re-read tap 8's pixels, finite)
            const int tA = 2 * s, tB = 2 * s + 1 < 9 ? 2 * s + 1 : 8;
            const int offA = ((tA / 3) * SP_IW + tA % 3) * 16, offB = ((tB / 3) * SP_IW + tB % 3) * 16;
            const int toff = tap_sel ? offB : offA;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                bh[buf][p] = *reinterpret_cast<const h8*>(xin + boff[p] + toff);
                bl[buf][p] = *reinterpret_cast<const h8*>(xin + 2 * SP_CG_STRIDE + boff[p] + toff);
            }
#pragma unroll
            for (int m = 0; m < NMT; ++m) {
                ah[buf][m] = *reinterpret_cast<const h8*>(win + ((s * NMT + m) * 2 + 0) * 1024);
                al[buf][m] = *reinterpret_cast<const h8*>(win + ((s * NMT + m) * 2 + 1) * 1024);
            }
        };
        ld(0, 0);
#pragma unroll
        for (int s = 0; s < SP_STEPS; ++s) {
            if (s == SP_STEPS - 1) {
                SSTAMP(t1)
#if defined(SPLIT_ABLATE) && SPLIT_ABLATE >= 5
#elif defined(SPLIT_ABLATE) && SPLIT_ABLATE == 3
                if (ch + 2 < n_chunks) { issue_weights(ch + 2); }
#elif defined(SPLIT_ABLATE) && SPLIT_ABLATE == 4
                if (ch + 1 < n_chunks) store_inputs(ch + 1);
                if (ch + 2 < n_chunks) { load_inputs(ch + 2); }
#else
                if (ch + 1 < n_chunks) store_inputs(ch + 1);
                if (ch + 2 < n_chunks) { issue_weights(ch + 2); load_inputs(ch + 2); }
#endif
                SSTAMP(t2)
#ifdef FLDR_STAMPS
                st_mfma += t1 - t0; st_stage += t2 - t1; tb0 = t2;
#endif
            }
            // term-major order: the 2*NMT accumulators are independent within a term, so no MFMA waits for its predecessor
#pragma unroll
            for (int term = 0; term < TERMS; ++term)
#pragma unroll
                for (int m = 0; m < NMT; ++m)
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const h8 av = term == 2 ? al[s & 1][m] : ah[s & 1][m];
                        const h8 bv = term == 1 ? bl[s & 1][p] : bh[s & 1][p];
#if !defined(SPLIT_ABLATE) || SPLIT_ABLATE != 2
                        acc[m][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc[m][p], 0, 0, 0);
#else
                        asm volatile("" :: "v"(av), "v"(bv));
#endif
                    }
            // the operands of step s+1 are requested AFTER the MFMAs of step s have been issued: they return while the
            // matrix pipe works, and the lgkmcnt wait in front of step s+1 covers reads that are a whole step old
#if !defined(SPLIT_ABLATE) || SPLIT_ABLATE != 1
            if (s + 1 < SP_STEPS) ld((s + 1) & 1, s + 1);
#else
            if (s + 1 < SP_STEPS) { for (int p = 0; p < 2; ++p) { bh[(s + 1) & 1][p] = bh[s & 1][p]; bl[(s + 1) & 1][p] = bl[s & 1][p]; }
                                    for (int m = 0; m < NMT; ++m) { ah[(s + 1) & 1][m] = ah[s & 1][m]; al[(s + 1) & 1][m] = al[s & 1][m]; } }
#endif
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * TERMS * NMT, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4 + 2 * NMT, 0);
        }
        SSTAMP(t3)
        __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): my LDS writes are done (vmcnt left alone)
#if !defined(SPLIT_ABLATE) || SPLIT_ABLATE != 6
        __builtin_amdgcn_s_barrier();
#endif
        SSTAMP(t4)
#ifdef FLDR_STAMPS
        st_last += t3 - tb0; st_bar += t4 - t3;
#endif
    }
#ifdef FLDR_STAMPS
    if (blockIdx.x == FLDR_STAMP_BLOCK && blockIdx.y == 0 && lane == 0) {
        unsigned long long* o = fldr_split_stamp_buf + wave * 8;
        o[0] = st_mfma; o[1] = st_stage; o[2] = st_last; o[3] = st_bar; o[4] = n_chunks;
    }
#endif

    // ---- epilogue: undo the weight scale, bias, ReLU, residual, coalesced 64-B row segments ----
    const float inv_scale = a.wpack[0];
    const int64_t HW = (int64_t)a.H * a.W;
    float* outn = a.out + (int64_t)n * a.cout_store * HW;
    const float* resn = a.residual ? a.residual + (int64_t)n * a.cout_store * HW : nullptr;
    float bias_r[NMT][4];
#pragma unroll
    for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias_r[m][r] = 0.0f;
    if (a.bias) {
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int co = cbase + m * 16 + lg * 4 + r;
                co = co < a.cout ? co : a.cout - 1;
                bias_r[m][r] = a.bias[co];
            }
    }
    const int oy = oy0 + wave;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int ox = ox0 + p * 16 + lj;
        const bool pix_ok = oy < a.H && ox < a.W;
        const int64_t po = pix_ok ? (int64_t)oy * a.W + ox : 0;
        float res_r[NMT][4];
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) res_r[m][r] = 0.0f;
        if (resn) {
#pragma unroll
            for (int m = 0; m < NMT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int co = cbase + m * 16 + lg * 4 + r;
                    co = co < a.cout_store ? co : a.cout_store - 1;
                    res_r[m][r] = resn[(int64_t)co * HW + po];
                }
#pragma unroll
            for (int m = 0; m < NMT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) fldr_pin(res_r[m][r]);
        }
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = cbase + m * 16 + lg * 4 + r;
                float v = acc[m][p][r] * inv_scale + bias_r[m][r];
                if (a.relu) v = fmaxf(v, 0.0f);
                v += res_r[m][r];
                if (co < a.cout_store && pix_ok) outn[(int64_t)co * HW + po] = v;
            }
    }
    fldr_note_range(range_bad);
}

// ------------------------------------------------------------------------------------------------
// prepack: max|w| -> power-of-two scale -> hi/lo halves in MFMA A-operand order
//   layout after the 4-float header: [group][chunk][step][m][kind hi|lo][lane][8 halves]
// ------------------------------------------------------------------------------------------------
static inline void split_geometry(int cout, int& nmt, int& groups) {
    if (cout <= 16)      { nmt = 1; groups = 1; }
    else if (cout <= 32) { nmt = 2; groups = 1; }
    else if (cout <= 48) { nmt = 3; groups = 1; }
    else if (cout <= 64) { nmt = 2; groups = 2; }
    else                 { nmt = 3; groups = (cout + 47) / 48; }
}

__global__ void split_absmax_kernel(const float* __restrict__ w, int64_t n, float* __restrict__ hdr) {
    __shared__ float red[256];
    float m = 0.0f;
    for (int64_t i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(w[i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mx = red[0];
        // largest power of two with mx * scale <= 8192 (fp16 max 65504); all-zero weights: scale 1
        float scale = 1.0f;
        if (mx > 0.0f) scale = exp2f(floorf(log2f(8192.0f / mx)));
        hdr[0] = 1.0f / scale; hdr[1] = scale; hdr[2] = mx; hdr[3] = 0.0f;
    }
}

__global__ void split_prepack_kernel(const float* __restrict__ w, float* __restrict__ wp, int cout, int cin, int nmt,
                                     int n_chunks, int64_t total_h8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one 8-half (16-byte) element per thread
    if (i >= total_h8) return;
    const float scale = wp[1];
    const int lane = (int)(i % 64);
    const int kind = (int)((i / 64) % 2);
    const int m = (int)((i / 128) % nmt);
    const int s = (int)((i / (128 * nmt)) % SP_STEPS);
    const int ch = (int)((i / (128 * nmt * SP_STEPS)) % n_chunks);
    const int gr = (int)(i / ((int64_t)128 * nmt * SP_STEPS * n_chunks));
    const int li = lane & 15, lg = lane >> 4;
    const int co = gr * 16 * nmt + m * 16 + li;
    const int tap = 2 * s + (lg >> 1);
    h8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = ch * 16 + (lg & 1) * 8 + j;
        float x = (tap < 9 && co < cout && c < cin) ? w[((int64_t)co * cin + c) * 9 + tap] * scale : 0.0f;
        const _Float16 h = (_Float16)x;
        v[j] = kind == 0 ? h : (_Float16)(x - (float)h);
    }
    reinterpret_cast<h8*>(wp + SP_HDR)[i] = v;
}

extern "C" int64_t fldr_conv_split_prepack_size(int cout, int cin) {
    if (cout <= 0 || cin <= 0 || cout > 96 || cin > SP_MAX_CIN) return FLDR_E_ARG;
    int nmt, groups;
    split_geometry(cout, nmt, groups);
    const int n_chunks = (cin + 15) / 16;
    return SP_HDR + (int64_t)groups * n_chunks * SP_STEPS * nmt * 2 * 64 * 4;     // floats (8 halves = 4 floats)
}

extern "C" int fldr_conv_split_prepack(const float* weight, float* wpack, int cout, int cin, fldr_stream_t stream) {
    FLDR_CHECK_ARG(weight && wpack);
    const int64_t total = fldr_conv_split_prepack_size(cout, cin);
    if (total < 0) return (int)total;
    int nmt, groups;
    split_geometry(cout, nmt, groups);
    const int n_chunks = (cin + 15) / 16;
    const int64_t total_h8 = (total - SP_HDR) / 4;
    hipLaunchKernelGGL(split_absmax_kernel, dim3(1), dim3(256), 0, fldr_s(stream), weight, (int64_t)cout * cin * 9, wpack);
    hipLaunchKernelGGL(split_prepack_kernel, dim3(fldr_cdiv(total_h8, 256)), dim3(256), 0, fldr_s(stream), weight, wpack, cout, cin,
                       nmt, n_chunks, total_h8);
    FLDR_LAUNCH_RET();
}

template <int NMT, int TERMS>
static int split_launch(const SplitArgs& a, int N, hipStream_t s) {
    using Cfg = SplitCfg<NMT>;
    SplitArgs b = a;
    b.tiles_x = fldr_cdiv(a.W, SP_TW);
    const int tiles_y = fldr_cdiv(a.H, SP_TH);
    static std::atomic<uint64_t> attr_done{0};
    if (int e = fldr_set_max_lds(reinterpret_cast<const void*>(&conv3x3_split_kernel<NMT, TERMS>), Cfg::LDS_BYTES, attr_done)) return e;
    b.n_tiles = b.tiles_x * tiles_y;
    b.tiles_per_xcd = (b.n_tiles + 7) / 8;
    hipLaunchKernelGGL((conv3x3_split_kernel<NMT, TERMS>), dim3(8 * b.tiles_per_xcd * b.groups, N), dim3(512), Cfg::LDS_BYTES, s, b);
    FLDR_LAUNCH_RET();
}

// Same descriptor as fldr_conv2d; only ksize 3 / stride 1; d->wpack must come from fldr_conv_split_prepack.
extern "C" int fldr_conv2d_split(const fldr_conv_desc* d, fldr_stream_t stream) {
    FLDR_CHECK_ARG(d && d->wpack && d->out && d->n_src >= 1 && d->n_src <= FLDR_CONV_MAX_SRC);
    FLDR_CHECK_ARG(d->N > 0 && d->cin > 0 && d->cin <= SP_MAX_CIN && d->cout > 0 && d->cout <= 96);
    FLDR_CHECK_ARG(d->cout_store > 0 && d->cout_store <= d->cout && d->ksize == 3 && d->stride == 1);
    if (d->Hout != d->Hin || d->Wout != d->Win) return FLDR_E_SHAPE;
    for (int k = 0; k < d->n_src; ++k) if (d->src_cstride[k] != 0) return FLDR_E_ARG;      // channel-strided sources: fldr_conv2d_s2_split only
    SplitArgs a;
    int csum = 0;
    for (int s = 0; s < d->n_src; ++s) {
        FLDR_CHECK_ARG(d->src[s] && d->src_c[s] > 0);
        if (d->src_up2[s] && ((d->Hin | d->Win) & 1)) return FLDR_E_SHAPE;
        if (csum + d->src_c[s] > SP_MAX_CIN) return FLDR_E_ARG;
        const int64_t plane = d->src_up2[s] ? (int64_t)(d->Hin >> 1) * (d->Win >> 1) : (int64_t)d->Hin * d->Win;
        for (int c = 0; c < d->src_c[s]; ++c) {
            a.chan[csum + c] = (unsigned long long)reinterpret_cast<uintptr_t>(d->src[s] + (int64_t)c * plane) | (d->src_up2[s] ? 1ull : 0ull);
            a.chan_bstride[csum + c] = d->src_bstride[s];
        }
        csum += d->src_c[s];
    }
    if (csum != d->cin) return FLDR_E_SHAPE;
    for (int c = csum; c < SP_MAX_CIN; ++c) { a.chan[c] = 0ull; a.chan_bstride[c] = 0; }
    a.wpack = d->wpack; a.bias = d->bias; a.residual = d->residual; a.out = d->out;
    a.cin = d->cin; a.cout = d->cout; a.cout_store = d->cout_store;
    a.H = d->Hin; a.W = d->Win; a.relu = d->relu; a.tiles_x = 0;
    int nmt, groups;
    split_geometry(d->cout, nmt, groups);
    a.groups = groups;
    hipStream_t s = fldr_s(stream);
    if (d->precision == 1) {                           // plain fp16 inputs
        if (nmt == 1) return split_launch<1, 1>(a, d->N, s);
        if (nmt == 2) return split_launch<2, 1>(a, d->N, s);
        return split_launch<3, 1>(a, d->N, s);
    }
    if (nmt == 1) return split_launch<1, 3>(a, d->N, s);
    if (nmt == 2) return split_launch<2, 3>(a, d->N, s);
    return split_launch<3, 3>(a, d->N, s);
}
