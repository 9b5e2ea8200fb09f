// Convolutions of fLDRnet (3x3 s1 p1 and 4x4 s2 p1) as implicit GEMMs on the gfx950 matrix cores.
//
// GEMM view: D[cout][pixel] = sum_{cin,tap} Wt[cout][cin,tap] * X[cin,tap][pixel].  The weights are the A
// operand (rows = output channels) and the pixels the B operand (columns), so that in the MFMA result
// layout a lane owns one PIXEL column and the registers walk the output channels: every store
// instruction of the epilogue writes 32 (or 16) consecutive pixels of one NCHW plane per half/quarter
// wave, i.e. full 128-B (64-B) segments, and the input tile is read from LDS with consecutive lanes on
// consecutive addresses (conflict free).
//
//   precision 0: v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 — exact fp32 products, fp32
//                accumulation (bitwise an fmaf chain), the parity path.
//
// A 256-thread workgroup (4 waves, one per SIMD) owns a TH x 32 output tile and ALL output channels
// (<= 96), loops over the input channels in chunks of CC, and per chunk stages in LDS
//   * the input tile with its halo, assembled on the fly from up to 12 sources (torch.cat is never
//     materialised; a source may be read through nearest x2 upsampling; zero padding by predication),
//   * the CC x taps x cout slice of the prepacked weights.
// cout <= 16 uses the 16x16x4 shape (M = 16), otherwise 32x32x2 with ceil(cout/32) M-tiles.
#include "common.h"
#include <type_traits>

#ifdef FLDR_STAMPS
// Diagnostic build only (tools/stamps): per-phase s_memtime sums of workgroup 0, written to a buffer no kernel reads.
#ifndef FLDR_STAMP_BLOCK
#define FLDR_STAMP_BLOCK 0
#endif
__device__ unsigned long long fldr_stamp_buf[4 * 8];
#define STAMP(var) unsigned long long var; { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
FLDR_HOOK int fldr_debug_read_stamps(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fldr_stamp_buf), sizeof(unsigned long long) * 32);
}
#else
#define STAMP(var)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
    const float* src[FLDR_CONV_MAX_SRC];
    int64_t src_bstride[FLDR_CONV_MAX_SRC];
    int32_t src_cbegin[FLDR_CONV_MAX_SRC + 1];
    int32_t src_up2[FLDR_CONV_MAX_SRC];
    int32_t n_src;
    const float* wpack;
    const float* bias;
    const float* residual;
    float* out;                // fp32 NCHW output or null
    unsigned char* out_spk;    // split-packed output (conv_spk_kernels.hip) or null
    int32_t cin, cout, cout_store;
    int32_t Hin, Win, Hout, Wout;
    int32_t relu;
    int32_t tiles_x;
    int32_t groups;            // output-channel groups of MTOT channels; blockIdx.x = tile * groups + group
};

constexpr int pad16mod32(int v) { return v + ((16 - (v % 32)) + 32) % 32; }

template <int KS, int STRIDE, int MT, int NMT, int PT, int CC_>
struct ConvCfg {
    static constexpr int CC = CC_;
    static constexpr int TW = 32;
    static constexpr int TPR = TW / MT;                  // pixel tiles per tile row (1 or 2)
    static constexpr int RPW = PT / TPR;                 // rows per wave
    static constexpr int TH = 4 * RPW;
    static constexpr int IH = (TH - 1) * STRIDE + KS;
    static constexpr int IW = (TW - 1) * STRIDE + KS;
    static constexpr int IWH = (IW + 1) / 2;             // stride 2: even/odd columns de-interleaved
    static constexpr int IWP = STRIDE == 2 ? 2 * IWH : IW;
    static constexpr int ICH = MT == 16 ? pad16mod32(IH * IWP) : IH * IWP;      // channel stride (floats)
    static constexpr int TAPS = KS * KS;
    static constexpr int MTOT = NMT * MT;
    static constexpr int WCH = MT == 16 ? pad16mod32(TAPS * MTOT) : TAPS * MTOT;   // also the stride of wpack
    static constexpr int KG = MT == 16 ? 4 : 2;          // input channels per MFMA
    static constexpr int NI = (IH * IW + 63) / 64;       // input elements per lane per channel
    static constexpr int CPW = CC / 4;                   // channels staged per wave
    static constexpr int WSLAB = CC * WCH;               // weight floats per chunk (contiguous in wpack)
    static constexpr int NWI = (WSLAB / 4 + 255) / 256;  // 16-B LDS-DMA pieces per thread per chunk
    static constexpr int W_LDS = (WSLAB + 63) / 64 * 64;  // weight region (the DMA lanes past the slab are masked off): a tight
                                                         // region lets three enc1 workgroups share a CU's 160 KB instead of two
    static constexpr int BUF_FLOATS = W_LDS + CC * ICH;  // one pipeline stage: [weights | input]
    static constexpr int TAB_FLOATS = 2 * 112;           // per-channel plane pointers (64-bit), cin_pad <= 112
    static constexpr int LDS_FLOATS = 2 * BUF_FLOATS + TAB_FLOATS;
    static constexpr int KSTEPS = CC / KG * TAPS;        // MFMA k-steps per chunk
};

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// The chunk loop is software pipelined over two LDS stages: while the MFMAs consume stage k,
//   * the weight slab of chunk k+1 streams global -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPRs),
//   * the input tile of chunk k+1 is prefetched into registers (zero padding / nearest-x2 / source
//     selection resolved at load time) and written to LDS after the MFMA phase,
// and ONE barrier per chunk separates the stages.
template <int KS, int STRIDE, int MT, int NMT, int PT, int CC_>
__global__ __launch_bounds__(256, ((MT == 16 && NMT == 3) || (KS == 4 && MT == 16)) ? 3 : 2) void conv_mfma_kernel(ConvArgs a) {
    using Cfg = ConvCfg<KS, STRIDE, MT, NMT, PT, CC_>;
    constexpr int CC = Cfg::CC, IH = Cfg::IH, IW = Cfg::IW, IWP = Cfg::IWP, IWH = Cfg::IWH, ICH = Cfg::ICH;
    constexpr int TAPS = Cfg::TAPS, MTOT = Cfg::MTOT, WCH = Cfg::WCH, KG = Cfg::KG;
    constexpr int NI = Cfg::NI, CPW = Cfg::CPW, NWI = Cfg::NWI, W_LDS = Cfg::W_LDS, BUF = Cfg::BUF_FLOATS;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n = blockIdx.y;
    const int grp = blockIdx.x % a.groups, tile = blockIdx.x / a.groups;
    const int cbase = grp * MTOT;                            // first output channel of this workgroup
    const int tile_y = tile / a.tiles_x, tile_x = tile % a.tiles_x;
    const int oy0 = tile_y * Cfg::TH, ox0 = tile_x * Cfg::TW;
    const int iy0 = oy0 * STRIDE - 1, ix0 = ox0 * STRIDE - 1;          // pad = 1

    const int lj = lane & (MT - 1);          // pixel within pixel tile (B column) / cout within M tile (A row)
    const int lk = lane / MT;                // k index within the MFMA (0..KG-1)

    int boff[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        int r = wave * Cfg::RPW + p / Cfg::TPR;
        int c0 = (p % Cfg::TPR) * MT;
        boff[p] = W_LDS + lk * ICH + (r * STRIDE) * IWP + (c0 + lj);
    }
    const int aoff = lk * WCH + lj;

    // per-lane geometry of the staged input elements (identical for every channel and chunk)
    int g_full[NI], g_half[NI], l_off[NI];
    unsigned vmask = 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int e = lane + 64 * i;
        const int y = e / IW, x = e % IW;
        const int gy = iy0 + y, gx = ix0 + x;
        const bool ok = e < IH * IW && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
        vmask |= ok ? (1u << i) : 0u;
        g_full[i] = ok ? gy * a.Win + gx : 0;
        g_half[i] = ok ? (gy >> 1) * (a.Win >> 1) + (gx >> 1) : 0;
        l_off[i] = e < IH * IW ? y * IWP + (STRIDE == 2 ? (x & 1) * IWH + (x >> 1) : x) : -1;
    }

    typedef typename std::conditional<MT == 32, f32x16, f32x4>::type acc_t;
    acc_t acc[NMT][PT];
#pragma unroll
    for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int r = 0; r < (MT == 32 ? 16 : 4); ++r) acc[m][p][r] = 0.0f;

    float pre[CPW][NI];
    const int cin_pad = (a.cin + CC - 1) / CC * CC;

    // Per-channel plane pointer table (bit 0 = "stored at half resolution"), built once: resolving the
    // source of a channel from the kernel arguments needs dynamically indexed loads, which must not sit
    // in the pipelined loop (their vmcnt(0) would drain the prefetches every chunk).
    unsigned long long* ctab = reinterpret_cast<unsigned long long*>(smem + 2 * BUF);
    if (tid < 112) {
        unsigned long long e = 0;
        if (tid < a.cin) {
            int s = 0;
            while (s + 1 < a.n_src && tid >= a.src_cbegin[s + 1]) ++s;
            const int up2 = a.src_up2[s];
            const int64_t plane = up2 ? (int64_t)(a.Hin >> 1) * (a.Win >> 1) : (int64_t)a.Hin * a.Win;
            const float* base = a.src[s] + (int64_t)n * a.src_bstride[s] + (int64_t)(tid - a.src_cbegin[s]) * plane;
            e = (unsigned long long)reinterpret_cast<uintptr_t>(base) | (unsigned long long)(up2 ? 1 : 0);
        }
        ctab[tid] = e;
    }
    __syncthreads();

    auto issue_weights = [&](int c0, float* stage) {
        const float* g = a.wpack + ((int64_t)grp * cin_pad + c0) * WCH;
#pragma unroll
        for (int i = 0; i < NWI; ++i) {
            const int piece = i * 256 + wave * 64;                    // wave-uniform piece base (x16 B)
            const int src = (piece + lane) * 4;
            if (src < Cfg::WSLAB)                                     // lanes past the slab write nothing
                __builtin_amdgcn_global_load_lds((gptr_t)(g + src), (lptr_t)(stage + piece * 4), 16, 0, 0);
        }
    };
    auto load_inputs = [&](int c0) {
#pragma unroll
        for (int q = 0; q < CPW; ++q) {
            const unsigned long long e = ctab[c0 + wave + 4 * q];
            const bool live = e != 0ull;
            const bool up2 = (e & 1ull) != 0ull;
            const float* base = reinterpret_cast<const float*>(static_cast<uintptr_t>(e & ~1ull));
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const bool ok = live && ((vmask >> i) & 1u);
                const int off = up2 ? g_half[i] : g_full[i];
                float v = 0.0f;
                if (ok) v = base[off];
                pre[q][i] = v;
            }
        }
    };
    auto store_inputs = [&](float* stage) {
#pragma unroll
        for (int q = 0; q < CPW; ++q) {
            float* dst = stage + W_LDS + (wave + 4 * q) * ICH;
#pragma unroll
            for (int i = 0; i < NI; ++i)
                if (l_off[i] >= 0) dst[l_off[i]] = pre[q][i];
        }
    };

    issue_weights(0, smem);
    load_inputs(0);
    store_inputs(smem);
    __syncthreads();

#ifdef FLDR_STAMPS
    unsigned long long s_issue = 0, s_mfma = 0, s_store = 0, s_bar = 0, s_n = 0;
    STAMP(t_begin)
#endif
    int stage = 0;
    for (int c0 = 0; c0 < cin_pad; c0 += CC, stage ^= 1) {
        float* cur = smem + stage * BUF;
        float* nxt = smem + (stage ^ 1) * BUF;
        const bool more = c0 + CC < cin_pad;
        STAMP(t0)
        STAMP(t1)
        {
            float av[3][NMT], bv[3][PT];
            auto ld = [&](int buf, int k) {
                const int cg = (k / TAPS) * KG, t = k % TAPS;
                const int dy = t / KS, dx = t % KS;
                const int toff = dy * IWP + (STRIDE == 2 ? (dx & 1) * IWH + (dx >> 1) : dx);
                const float* wc = cur + cg * WCH + aoff + t * MTOT;
                const float* ic = cur + cg * ICH + toff;
#pragma unroll
                for (int m = 0; m < NMT; ++m) av[buf][m] = wc[m * MT];
#pragma unroll
                for (int p = 0; p < PT; ++p) bv[buf][p] = ic[boff[p]];
            };
            ld(0, 0);
            if (Cfg::KSTEPS > 1) ld(1, 1);
#pragma unroll
            for (int k = 0; k < Cfg::KSTEPS; ++k) {
                // staging of the NEXT chunk is issued from inside the MFMA stream so that it overlaps matrix work
                if (k == 1 && more) issue_weights(c0 + CC, nxt);
                if (k == 2 && more) load_inputs(c0 + CC);
                if (k == Cfg::KSTEPS - 2 && more) store_inputs(nxt);
                if (k + 2 < Cfg::KSTEPS) ld((k + 2) % 3, k + 2);       // operands of step k+2 fly under the MFMAs of steps k, k+1
#pragma unroll
                for (int m = 0; m < NMT; ++m)
#pragma unroll
                    for (int p = 0; p < PT; ++p) {
                        if constexpr (MT == 32) acc[m][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k % 3][m], bv[k % 3][p], acc[m][p], 0, 0, 0);
                        else                    acc[m][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k % 3][m], bv[k % 3][p], acc[m][p], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_group_barrier(0x100, NMT + PT, 0);   // DS reads of step k+2 ...
                __builtin_amdgcn_sched_group_barrier(0x008, NMT * PT, 0);   // ... then the MFMAs of step k
            }
        }
        STAMP(t2)
        STAMP(t3)
        __syncthreads();
        STAMP(t4)
#ifdef FLDR_STAMPS
        s_issue += t1 - t0; s_mfma += t2 - t1; s_store += t3 - t2; s_bar += t4 - t3; s_n += 1;
#endif
    }
#ifdef FLDR_STAMPS
    STAMP(t_loop_end)
#endif

    // ---- epilogue: bias, ReLU, residual, store (lane = pixel column, registers = output channels) ----
#ifdef FLDR_STAMPS
    if (blockIdx.x == FLDR_STAMP_BLOCK && blockIdx.y == 0 && lane == 0) {
        unsigned long long* o = fldr_stamp_buf + wave * 8;
        o[0] = s_issue; o[1] = s_mfma; o[2] = s_store; o[3] = s_bar; o[4] = s_n; o[5] = t_loop_end - t_begin;
    }
#endif
    // All bias / residual loads are issued first (clamped addresses, no branches), then combined and stored.
    constexpr int NR = MT == 32 ? 16 : 4;
    const int64_t HWo = (int64_t)a.Hout * a.Wout;
    float* outn = a.out ? a.out + (int64_t)n * a.cout_store * HWo : nullptr;
    unsigned char* spkn = a.out_spk ? a.out_spk + (int64_t)n * ((a.cout_store + 7) >> 3) * 2 * HWo * 16 : nullptr;
    bool range_bad = false;
    const float* resn = a.residual ? a.residual + (int64_t)n * a.cout_store * HWo : nullptr;
    float bias_r[NMT][NR];
#pragma unroll
    for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int r = 0; r < NR; ++r) bias_r[m][r] = 0.0f;
    if (a.bias) {                                            // wave-uniform: one batch of loads, no per-element branch
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                int co = cbase + (MT == 32 ? m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk : m * 16 + lk * 4 + r);
                co = co < a.cout ? co : a.cout - 1;
                bias_r[m][r] = a.bias[co];
            }
    }
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int oy = oy0 + wave * Cfg::RPW + p / Cfg::TPR;
        const int ox = ox0 + (p % Cfg::TPR) * MT + lj;
        const bool pix_ok = oy < a.Hout && ox < a.Wout;
        const int64_t po = pix_ok ? (int64_t)oy * a.Wout + ox : 0;
        float res_r[NMT][NR];
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#pragma unroll
            for (int r = 0; r < NR; ++r) res_r[m][r] = 0.0f;
        if (resn) {                                          // wave-uniform
#pragma unroll
            for (int m = 0; m < NMT; ++m)
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    int co = cbase + (MT == 32 ? m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk : m * 16 + lk * 4 + r);
                    co = co < a.cout_store ? co : a.cout_store - 1;
                    res_r[m][r] = resn[(int64_t)co * HWo + po];
                }
#pragma unroll
            for (int m = 0; m < NMT; ++m)
#pragma unroll
                for (int r = 0; r < NR; ++r) fldr_pin(res_r[m][r]);
        }
#pragma unroll
        for (int m = 0; m < NMT; ++m) {
            float vv[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int co = cbase + (MT == 32 ? m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk : m * 16 + lk * 4 + r);
                float v = acc[m][p][r] + bias_r[m][r];
                if (a.relu) v = fmaxf(v, 0.0f);
                v += res_r[m][r];
                vv[r] = v;
                if (outn && co < a.cout_store && pix_ok) outn[(int64_t)co * HWo + po] = v;
            }
            if (spkn) {
                // registers r0..r0+3 are four consecutive channels = half of a split-packed group: one 8-byte store of
                // the hi halves and one of the lo halves (the same split as conv_spk_kernels.hip)
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
#pragma unroll
                for (int r0 = 0; r0 < NR; r0 += 4) {
                    const int co0 = cbase + (MT == 32 ? m * 32 + 8 * (r0 >> 2) + 4 * lk : m * 16 + lk * 4);
                    h4 hi, lo;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float x = co0 + r < a.cout_store ? vv[r0 + r] : 0.0f;
                        _Float16 h_, l_;
                        fldr_split_hl(x, h_, l_, range_bad);
                        hi[r] = h_; lo[r] = l_;
                    }
                    if ((co0 >> 3) < ((a.cout_store + 7) >> 3) && pix_ok) {     // every quad of a stored group: padding channels are written as zeros, never left uninitialised
                        unsigned char* q = spkn + ((int64_t)(co0 >> 3) * 2 * HWo + po) * 16 + ((co0 >> 2) & 1) * 8;
                        *reinterpret_cast<h4*>(q) = hi;
                        *reinterpret_cast<h4*>(q + HWo * 16) = lo;
                    }
                }
            }
        }
    }
    if (spkn) fldr_note_range(range_bad);
}

FLDR_TU_STATUS(conv)

// ------------------------------------------------------------------------------------------------
// weight prepack: [cout,cin,k,k] -> [cin_pad][taps][mtot], zero padded
// ------------------------------------------------------------------------------------------------
// Instance selection: M shape (16|32), M tiles per workgroup, output-channel groups, cin chunk.
//   cout <= 16 : 16x16x4, 1 tile           cout <= 32 : 32x32x2, 1 tile        cout <= 48 : 16x16x4, 3 tiles
//   cout <= 64 : 32x32x2, 2 tiles          cout <= 96 : 16x16x4, 3 tiles x 2 groups (half-size quanta: the
//   96-channel layers dominate and 2160 half tiles balance over 256 CUs far better than 1080 full ones)
static inline void conv_geometry(int cout, int ksize, int& mt, int& nmt, int& groups, int& cc) {
    groups = 1;
    if (cout <= 16)      { mt = 16; nmt = 1; }
    else if (cout <= 32) { mt = 32; nmt = 1; }
    else if (cout <= 48) { mt = 16; nmt = 3; }
    else if (cout <= 64) { mt = 32; nmt = 2; }
    else                 { mt = 16; nmt = 3; groups = 2; }
    cc = ksize == 4 ? 4 : 8;
}

static inline int conv_wch(int ksize, int mt, int nmt) {
    int wch = ksize * ksize * nmt * mt;
    return mt == 16 ? pad16mod32(wch) : wch;
}

// wpack layout: [group][cin_pad][WCH], WCH = taps * MTOT (+ padding), element (t, m) of a row at t * MTOT + m
__global__ void conv_prepack_kernel(const float* __restrict__ w, float* __restrict__ wp, int cout, int cin, int cin_pad,
                                    int taps, int mtot, int wch, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    int r = (int)(i % wch);
    int c = (int)((i / wch) % cin_pad);
    int g = (int)(i / ((int64_t)wch * cin_pad));
    int m = g * mtot + r % mtot, t = r / mtot;
    wp[i] = (t < taps && m < cout && c < cin) ? w[((int64_t)m * cin + c) * taps + t] : 0.0f;
}

extern "C" int64_t fldr_conv_prepack_size(int cout, int cin, int ksize) {
    if (cout <= 0 || cin <= 0 || (ksize != 3 && ksize != 4) || cout > 96 || cin > 112) return FLDR_E_ARG;
    int mt, nmt, groups, cc;
    conv_geometry(cout, ksize, mt, nmt, groups, cc);
    int cin_pad = (cin + cc - 1) / cc * cc;
    return (int64_t)groups * cin_pad * conv_wch(ksize, mt, nmt);
}

extern "C" int fldr_conv_prepack(const float* weight, float* wpack, int cout, int cin, int ksize, fldr_stream_t stream) {
    FLDR_CHECK_ARG(weight && wpack);
    int64_t total = fldr_conv_prepack_size(cout, cin, ksize);
    if (total < 0) return (int)total;
    int mt, nmt, groups, cc;
    conv_geometry(cout, ksize, mt, nmt, groups, cc);
    int cin_pad = (cin + cc - 1) / cc * cc;
    hipLaunchKernelGGL(conv_prepack_kernel, dim3(fldr_cdiv(total, 256)), dim3(256), 0, fldr_s(stream), weight, wpack, cout, cin,
                       cin_pad, ksize * ksize, nmt * mt, conv_wch(ksize, mt, nmt), total);
    FLDR_LAUNCH_RET();
}

template <int KS, int STRIDE, int MT, int NMT, int PT, int CC>
static int conv_launch(const ConvArgs& a, int N, hipStream_t s) {
    using Cfg = ConvCfg<KS, STRIDE, MT, NMT, PT, CC>;
    ConvArgs b = a;
    b.tiles_x = fldr_cdiv(a.Wout, Cfg::TW);
    const int tiles_y = fldr_cdiv(a.Hout, Cfg::TH);
    const size_t lds = sizeof(float) * Cfg::LDS_FLOATS;
    static std::atomic<uint64_t> attr_done{0};
    if (lds > 64 * 1024)
        if (int e = fldr_set_max_lds(reinterpret_cast<const void*>(&conv_mfma_kernel<KS, STRIDE, MT, NMT, PT, CC>), (int)lds, attr_done)) return e;
    hipLaunchKernelGGL((conv_mfma_kernel<KS, STRIDE, MT, NMT, PT, CC>), dim3(b.tiles_x * tiles_y * b.groups, N), dim3(256), lds, s, b);
    FLDR_LAUNCH_RET();
}

extern "C" int fldr_conv2d(const fldr_conv_desc* d, fldr_stream_t stream) {
    FLDR_CHECK_ARG(d && d->wpack && (d->out || d->out_spk) && d->n_src >= 1 && d->n_src <= FLDR_CONV_MAX_SRC);
    FLDR_CHECK_ARG(!d->residual || d->out);
    FLDR_CHECK_ARG(d->N > 0 && d->cin > 0 && d->cout > 0 && d->cout <= 96 && d->cout_store > 0 && d->cout_store <= d->cout);
    FLDR_CHECK_ARG((d->ksize == 3 && d->stride == 1) || (d->ksize == 4 && d->stride == 2));
    FLDR_CHECK_ARG(d->Hin > 0 && d->Win > 0);
    FLDR_CHECK_ARG(d->precision == 0);
    const int Ho = d->ksize == 3 ? d->Hin : (d->Hin + 2 - 4) / 2 + 1;
    const int Wo = d->ksize == 3 ? d->Win : (d->Win + 2 - 4) / 2 + 1;
    if (Ho != d->Hout || Wo != d->Wout) return FLDR_E_SHAPE;
    for (int k = 0; k < d->n_src; ++k) if (d->src_cstride[k] != 0) return FLDR_E_ARG;      // channel-strided sources: fldr_conv2d_s2_split only
    ConvArgs a;
    int csum = 0;
    for (int s = 0; s < FLDR_CONV_MAX_SRC; ++s) {
        bool live = s < d->n_src;
        if (live) {
            FLDR_CHECK_ARG(d->src[s] && d->src_c[s] > 0);
            if (d->src_up2[s] && ((d->Hin | d->Win) & 1)) return FLDR_E_SHAPE;
        }
        a.src[s] = live ? d->src[s] : nullptr;
        a.src_bstride[s] = live ? d->src_bstride[s] : 0;
        a.src_up2[s] = live ? d->src_up2[s] : 0;
        a.src_cbegin[s] = csum;
        if (live) csum += d->src_c[s];
    }
    a.src_cbegin[FLDR_CONV_MAX_SRC] = csum;
    if (csum != d->cin) return FLDR_E_SHAPE;
    a.n_src = d->n_src;
    a.wpack = d->wpack; a.bias = d->bias; a.residual = d->residual; a.out = d->out;
    a.out_spk = reinterpret_cast<unsigned char*>(d->out_spk);
    a.cin = d->cin; a.cout = d->cout; a.cout_store = d->cout_store;
    a.Hin = d->Hin; a.Win = d->Win; a.Hout = d->Hout; a.Wout = d->Wout; a.relu = d->relu; a.tiles_x = 0;
    hipStream_t s = fldr_s(stream);
    int mt, nmt, groups, cc;
    conv_geometry(d->cout, d->ksize, mt, nmt, groups, cc);
    a.groups = groups;
    if (d->ksize == 3) {
        if (mt == 16 && nmt == 1) return conv_launch<3, 1, 16, 1, 4, 8>(a, d->N, s);
        if (mt == 16)             return conv_launch<3, 1, 16, 3, 2, 8>(a, d->N, s);
        if (nmt == 1)             return conv_launch<3, 1, 32, 1, 2, 8>(a, d->N, s);
        return conv_launch<3, 1, 32, 2, 1, 8>(a, d->N, s);
    } else {
        if (mt == 16 && nmt == 1) return conv_launch<4, 2, 16, 1, 4, 4>(a, d->N, s);
        if (mt == 32 && nmt == 1) return conv_launch<4, 2, 32, 1, 2, 4>(a, d->N, s);
        if (mt == 32 && nmt == 2) return conv_launch<4, 2, 32, 2, 2, 4>(a, d->N, s);
        return FLDR_E_ARG;
    }
}

extern "C" int fldr_version(void) { return FLDR_VERSION; }

extern "C" const char* fldr_error_string(int code) {
    if (code == 0) return "success";
    if (code == FLDR_E_ARG) return "fldr: bad argument";
    if (code == FLDR_E_SHAPE) return "fldr: shape constraint violated";
    if (code == FLDR_E_STATUS) return "fldr: the status block of the device could not be allocated or bound";
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "fldr: unknown error";
}

// Diagnostic: resident workgroups per CU the runtime reports for the main 3x3 instances at their LDS sizes.
FLDR_HOOK int fldr_debug_conv_occupancy(int* out4) {
    int n = 0;
    hipError_t e;
    {
        using Cfg = ConvCfg<3, 1, 16, 3, 2, 8>;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_mfma_kernel<3, 1, 16, 3, 2, 8>, 256, sizeof(float) * Cfg::LDS_FLOATS);
        if (e != hipSuccess) return (int)e;
        out4[0] = n; out4[1] = (int)(sizeof(float) * Cfg::LDS_FLOATS);
    }
    {
        using Cfg = ConvCfg<3, 1, 32, 2, 1, 8>;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_mfma_kernel<3, 1, 32, 2, 1, 8>, 256, sizeof(float) * Cfg::LDS_FLOATS);
        if (e != hipSuccess) return (int)e;
        out4[2] = n; out4[3] = (int)(sizeof(float) * Cfg::LDS_FLOATS);
    }
    return 0;
}
