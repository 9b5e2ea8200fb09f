// 4x4 / stride 2 / pad 1 convolutions (the UNet encoders, fLDRnet.py:611-613) on the fp16 matrix cores with the 3 x fp16
// split of conv_split_kernels.hip (x = hi + lo, hi*hi + hi*lo + lo*hi, fp32 accumulation: fp32-equivalent accuracy).
//
// The exact fp32-MFMA kernel (conv_kernels.hip) is matrix-pipe bound on these layers (enc1: 76 % pipe busy at the clock the
// chip holds, 474 us against a 210 us HBM floor); here the matrix time drops 5.3x and the layer is bound by its 1 GB of
// plane traffic.  Same workgroup geometry as that kernel: 256 threads = 4 waves, an 8 x 32 output tile x all (<= 64)
// output channels, input channels in chunks of 4 staged through two LDS stages (weights by LDS-DMA, inputs prefetched into
// registers one chunk ahead: multi-source concat and zero padding resolved at load time).
//
// LDS input image per channel and per half (hi, lo): dwords [row pair][column parity][column / 2], each dword = the two rows of
// the pair as halves.  An output pixel (y, x) reads input rows 2y-1..2y+2 = row pairs y, y+1 and columns 2x-1..2x+2 =
// {even plane, odd plane} x indices {x, x+1} of the tile-local image, so ONE MFMA B operand (8 halves = 4 dwords) is a row
// pair x 4 columns = two ds_read2_b32 with no VALU work: the split happens once per staged element, not per use.
//   16x16x32 (cout <= 16): lane group g = lane / 16 -> (channel g / 2 of the step's pair, row pair g % 2): 2 steps per chunk
//   32x32x16 (cout <= 64): lane group g = lane / 32 -> row pair g of the step's channel:                    4 steps per chunk
// k slot j of a lane: dword j / 2 in the order (even x, even x+1, odd x, odd x+1) = kernel column dx {1, 3, 0, 2}... see
// s2_tap() — the weight prepack uses the same function, so any consistent order works.
#include "common.h"
#include <type_traits>

typedef _Float16 s2_h8 __attribute__((ext_vector_type(8)));
typedef float s2_f4 __attribute__((ext_vector_type(4)));
typedef float s2_f16 __attribute__((ext_vector_type(16)));
typedef int s2_i4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* s2_gptr_t;
typedef __attribute__((address_space(3))) void* s2_lptr_t;

#define S2_HDR 8                       // floats before the packed weights: {1/scale, scale, max|w|, 0, 0,0,0,0}
#define S2_CC 4                        // input channels per chunk
#define S2_TH 8
#define S2_TW 32
#define S2_IH 18                       // (TH-1)*2 + 4
#define S2_IW 66                       // (TW-1)*2 + 4
#define S2_RP 9                        // row pairs
#define S2_NI 10                       // ceil(RP*IW / 64) staged (row pair, column) items per lane and channel: 2 loads, 2 dword LDS writes each

struct S2Args {
    const float* src[FLDR_CONV_MAX_SRC];
    int64_t src_bstride[FLDR_CONV_MAX_SRC];
    int64_t src_cstride[FLDR_CONV_MAX_SRC];            // floats between channel planes
    int32_t src_cbegin[FLDR_CONV_MAX_SRC + 1];
    int32_t n_src;
    const float* wpack;
    const float* bias;
    float* out;                // fp32 NCHW or null
    unsigned char* out_spk;    // split-packed or null
    int32_t cin, cout, cout_store;
    int32_t Hin, Win, Hout, Wout;
    int32_t relu;
    int32_t tiles_x, n_tiles, tiles_per_xcd;
    int32_t N, wgs_per_xcd;    // persistent kernel: samples (tiles are numbered over all samples), workgroups per XCD
    int32_t x_shift;           // persistent kernel: the tile grid starts x_shift output columns left of the image (see s2_launch_pers)
    // second problem of a pair launch (conv4x4s2_pers_spk_kernel, gridDim.y == 2): the same source and geometry, other output channels
    const float* wpack2;
    const float* bias2;
    float* out2;
    unsigned char* out_spk2;
};

// kernel tap (dy, dx) of k slot j (0..7) within row pair rp (0..1): dword d = j / 2 -> (parity, index offset); half j % 2 -> row
__host__ __device__ __forceinline__ void s2_tap(int rp, int j, int& dy, int& dx) {
    const int d = j >> 1;                      // 0: even plane, x; 1: even plane, x+1; 2: odd plane, x; 3: odd plane, x+1
    const int parity = d >> 1, off = d & 1;
    // tile-local column = 2*xl + dx with parity dx & 1 and index xl + (dx >> 1)
    dx = parity + 2 * off;
    dy = 2 * rp + (j & 1);
}

template <int MT, int NMT>
struct S2Cfg {
    static constexpr int KL = 64 / MT;                         // lane groups (k octets) per MFMA: 4 (16x16x32) or 2 (32x32x16)
    static constexpr int CPS = KL / 2;                         // channels per MFMA step
    static constexpr int STEPS = S2_CC / CPS;                  // 2 or 4
    static constexpr int IWHP = 40;                            // dwords per (row pair, parity) line (33 used): the lane groups of a
                                                               // 16x16x32 operand read land 16 banks apart (32x32x16: half overlap)
    static constexpr int KIND = S2_RP * 2 * IWHP;              // dwords per channel and half
    static constexpr int CHS = 2 * KIND;                       // dwords per channel (hi, lo): 1440 = 32 mod 64, so the four lane groups
                                                               // of a 16x16x32 operand read ((channel, row pair) = +32, +16 banks) never collide
    static constexpr int X_DW = S2_CC * CHS;                   // input dwords per stage
    static constexpr int W_BYTES = STEPS * NMT * 2 * 1024;     // weights per chunk: [step][m][kind][lane][16 B]
    static constexpr int STAGE_BYTES = W_BYTES + X_DW * 4;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;          // 54,272 B for cout <= 16: three workgroups per CU
    static constexpr int NWI = (W_BYTES / 16 + 255) / 256;     // 16-B DMA pieces per thread per chunk
};

template <int MT, int NMT, int PT>
__global__ __launch_bounds__(256, 2) void conv4x4s2_split_kernel(S2Args a) {
    using Cfg = S2Cfg<MT, NMT>;
    constexpr int KL = Cfg::KL, STEPS = Cfg::STEPS, IWHP = Cfg::IWHP, KIND = Cfg::KIND, CHS = Cfg::CHS;
    constexpr int TPR = S2_TW / MT, RPW = PT / TPR;            // pixel tiles per tile row, output rows per wave
    static_assert(4 * RPW == S2_TH, "tile height");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n = blockIdx.y;
    // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2), so XCD x gets the contiguous
    // tile range [x*tpx, (x+1)*tpx): neighbouring tiles share their halo rows / columns (18 x 66 inputs per 16 x 64 core) in ONE L2.
    const int tile = (blockIdx.x & 7) * a.tiles_per_xcd + (blockIdx.x >> 3);
    if (tile >= a.n_tiles) return;                              // workgroup-uniform
    const int tile_y = tile / a.tiles_x, tile_x = tile % a.tiles_x;
    const int oy0 = tile_y * S2_TH, ox0 = tile_x * S2_TW;
    const int iy0 = oy0 * 2 - 1, ix0 = ox0 * 2 - 1;            // pad = 1
    const int lj = lane & (MT - 1), lg = lane / MT;

    // ---- staging geometry (identical for every channel and chunk): an item = one column of one row pair, i.e. exactly one
    //      dword of the hi plane and one of the lo plane (the staging was the bottleneck of this kernel when every element was
    //      converted and written on its own: 2 ds_write_b16 + 2 cvt per element against a 384-cycle MFMA phase per chunk) ----
    uint32_t g_off0[S2_NI], g_off1[S2_NI];                    // element offsets inside a channel plane (host-checked < 2^30)
    int l_dw[S2_NI];
    unsigned vmask0 = 0, vmask1 = 0;
#pragma unroll
    for (int i = 0; i < S2_NI; ++i) {
        const int e = lane + 64 * i;
        const int pr = e / S2_IW, x = e % S2_IW;
        const int gy0 = iy0 + 2 * pr, gy1 = gy0 + 1, gx = ix0 + x;
        const bool okx = e < S2_RP * S2_IW && gx >= 0 && gx < a.Win;
        const bool ok0 = okx && gy0 >= 0 && gy0 < a.Hin, ok1 = okx && gy1 >= 0 && gy1 < a.Hin;
        vmask0 |= ok0 ? (1u << i) : 0u;
        vmask1 |= ok1 ? (1u << i) : 0u;
        g_off0[i] = ok0 ? (uint32_t)(gy0 * a.Win + gx) : 0u;
        g_off1[i] = ok1 ? (uint32_t)(gy1 * a.Win + gx) : 0u;
        l_dw[i] = e < S2_RP * S2_IW ? (pr * 2 + (x & 1)) * IWHP + (x >> 1) : -1;      // dword inside a (channel, half) plane
    }

    // ---- operand geometry ----
    const int cps = KL / 2;
    const int b_ch = cps == 2 ? (lg >> 1) : 0;                 // channel of the step's pair held by this lane group
    const int b_rp = cps == 2 ? (lg & 1) : lg;                 // row pair
    int boff[PT];                                             // dword offset of (row pair r + b_rp, even plane, xl) in a channel's hi plane
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int r = wave * RPW + p / TPR, xl = (p % TPR) * MT + lj;
        boff[p] = b_ch * CHS + ((r + b_rp) * 2) * IWHP + xl;
    }

    typedef typename std::conditional<MT == 32, s2_f16, s2_f4>::type acc_t;
    acc_t acc[NMT][PT];
#pragma unroll
    for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int r = 0; r < (MT == 32 ? 16 : 4); ++r) acc[m][p][r] = 0.0f;

    const int cin_pad = (a.cin + S2_CC - 1) / S2_CC * S2_CC;
    const int n_chunks = cin_pad / S2_CC;

    // per-channel plane pointers in VGPR lanes (lane l = channel l and l + 64), read back with v_readlane: no LDS table, no
    // dynamically indexed kernel-argument reads in the loop
    unsigned long long ctab_lo = 0, ctab_hi = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c = lane + 64 * h;
        unsigned long long e = 0;
        if (c < a.cin) {
            int s = 0;
            while (s + 1 < a.n_src && c >= a.src_cbegin[s + 1]) ++s;
            const float* base = a.src[s] + (int64_t)n * a.src_bstride[s] + (int64_t)(c - a.src_cbegin[s]) * a.src_cstride[s];
            e = (unsigned long long)reinterpret_cast<uintptr_t>(base);
        }
        if (h == 0) ctab_lo = e; else ctab_hi = e;
    }
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    float pre[2 * S2_NI];
    auto issue_weights = [&](int chunk, unsigned char* stage) {
        const unsigned char* g = reinterpret_cast<const unsigned char*>(a.wpack + S2_HDR) + (int64_t)chunk * Cfg::W_BYTES;
#pragma unroll
        for (int i = 0; i < Cfg::NWI; ++i) {
            const int piece = i * 256 + wave * 64;                    // wave-uniform, x16 bytes
            if ((piece + lane) * 16 < Cfg::W_BYTES)
                __builtin_amdgcn_global_load_lds((s2_gptr_t)(g + (piece + lane) * 16), (s2_lptr_t)(stage + piece * 16), 16, 0, 0);
        }
    };
    auto load_inputs = [&](int chunk, float (&dst)[2 * S2_NI]) {  // wave w stages channel w of the chunk
        const int c = chunk * S2_CC + wave_u;                     // wave-uniform
        const unsigned long long t = c < 64 ? ctab_lo : ctab_hi;
        const unsigned long long e = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(t >> 32), c & 63) << 32) |
                                     (unsigned)__builtin_amdgcn_readlane((int)(unsigned)t, c & 63);
        const bool live = e != 0ull;
        // a GLOBAL pointer (address space 1), wave-uniform: global_load_dword with an SGPR base and a 32-bit lane offset
        // instead of flat_load (a flat load also occupies the LDS counter the operand reads below wait on)
        const auto* base = (const __attribute__((address_space(1))) float*)e;
#pragma unroll
        for (int i = 0; i < S2_NI; ++i) {
            float v0 = 0.0f, v1 = 0.0f;
            if (live && ((vmask0 >> i) & 1u)) v0 = base[(uint64_t)g_off0[i]];
            if (live && ((vmask1 >> i) & 1u)) v1 = base[(uint64_t)g_off1[i]];
            dst[2 * i] = v0; dst[2 * i + 1] = v1;
        }
    };
    auto store_inputs = [&](unsigned char* stage, const float (&src)[2 * S2_NI]) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2* hi = reinterpret_cast<h2*>(stage + Cfg::W_BYTES) + wave * CHS;
        h2* lo = hi + KIND;
        bool bad = false;                                                          // (one flag raise per call: the raise is cold code, kept out of the unrolled body)
#pragma unroll
        for (int i = 0; i < S2_NI; ++i) {
            if (l_dw[i] < 0) continue;
            const float x0 = src[2 * i], x1 = src[2 * i + 1];
            h2 h, l;                                                               // the guarded split of common.h
            {
                const float xs[2] = {x0, x1};
                _Float16 hs[2], ls[2];
                fldr_split_hl_group(xs, hs, ls, bad);
                h[0] = hs[0]; h[1] = hs[1]; l[0] = ls[0]; l[1] = ls[1];
            }
            hi[l_dw[i]] = h;                                                       // half 0 = the even row of the pair
            lo[l_dw[i]] = l;
        }
        fldr_note_range(bad);
    };

    issue_weights(0, smem);
    load_inputs(0, pre);
    store_inputs(smem, pre);
    __syncthreads();

    // iteration ch: MFMAs on stage ch&1 while the weights of chunk ch+1 stream into the other stage by LDS-DMA and its inputs
    // are prefetched into registers; after the MFMAs the registers are split and written; one barrier per chunk.
    // (Built and measured equal or slower: a two-chunk-deep register prefetch with counted waits, 16-byte staging loads,
    // unconditional clamped loads.  enc1 moves 1.06 GB in ~400 us.)
    for (int ch = 0; ch < n_chunks; ++ch) {
        unsigned char* cur = smem + (ch & 1) * Cfg::STAGE_BYTES;
        unsigned char* nxt = smem + ((ch & 1) ^ 1) * Cfg::STAGE_BYTES;
        const bool more = ch + 1 < n_chunks;
        if (more) {                                               // stage nxt was last read in iteration ch-1 (barrier since)
            issue_weights(ch + 1, nxt);
            load_inputs(ch + 1, pre);
        }
        const int* xin = reinterpret_cast<const int*>(cur + Cfg::W_BYTES);
        const unsigned char* win = cur + lane * 16;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            s2_h8 ah[NMT], al[NMT];
#pragma unroll
            for (int m = 0; m < NMT; ++m) {
                ah[m] = *reinterpret_cast<const s2_h8*>(win + ((s * NMT + m) * 2 + 0) * 1024);
                al[m] = *reinterpret_cast<const s2_h8*>(win + ((s * NMT + m) * 2 + 1) * 1024);
            }
            const int cbase = s * cps * CHS;                        // first channel of this step
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                const int* q = xin + cbase + boff[p];
                s2_i4 bhi, blo;
                bhi[0] = q[0]; bhi[1] = q[1]; bhi[2] = q[IWHP]; bhi[3] = q[IWHP + 1];
                blo[0] = q[KIND]; blo[1] = q[KIND + 1]; blo[2] = q[KIND + IWHP]; blo[3] = q[KIND + IWHP + 1];
                const s2_h8 bh = __builtin_bit_cast(s2_h8, bhi), bl = __builtin_bit_cast(s2_h8, blo);
#pragma unroll
                for (int m = 0; m < NMT; ++m) {
                    if constexpr (MT == 32) {
                        acc[m][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh, acc[m][p], 0, 0, 0);
                        acc[m][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl, acc[m][p], 0, 0, 0);
                        acc[m][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh, acc[m][p], 0, 0, 0);
                    } else {
                        acc[m][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh, acc[m][p], 0, 0, 0);
                        acc[m][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl, acc[m][p], 0, 0, 0);
                        acc[m][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh, acc[m][p], 0, 0, 0);
                    }
                }
            }
        }
        if (more) store_inputs(nxt, pre);                         // split + LDS write of the registers prefetched above
        __syncthreads();
    }

    // ---- epilogue: undo the weight scale, bias, ReLU, store (lane = pixel column, registers = output channels) ----
    constexpr int NR = MT == 32 ? 16 : 4;
    const float inv_scale = a.wpack[0];
    const int64_t HWo = (int64_t)a.Hout * a.Wout;
    float* outn = a.out ? a.out + (int64_t)n * a.cout_store * HWo : nullptr;
    unsigned char* spkn = a.out_spk ? a.out_spk + (int64_t)n * ((a.cout_store + 7) >> 3) * 2 * HWo * 16 : nullptr;
    const int lk = lg;
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int oy = oy0 + wave * RPW + p / TPR;
        const int ox = ox0 + (p % TPR) * MT + lj;
        const bool pix_ok = oy < a.Hout && ox < a.Wout;
        const int64_t po = pix_ok ? (int64_t)oy * a.Wout + ox : 0;
#pragma unroll
        for (int m = 0; m < NMT; ++m) {
            float vv[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int co = MT == 32 ? m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk : m * 16 + lk * 4 + r;
                const int cb = co < a.cout ? co : a.cout - 1;
                float v = acc[m][p][r] * inv_scale + (a.bias ? a.bias[cb] : 0.0f);
                if (a.relu) v = fmaxf(v, 0.0f);
                vv[r] = v;
                if (outn && co < a.cout_store && pix_ok) outn[(int64_t)co * HWo + po] = v;
            }
            if (spkn) {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                // the range guard of the split, decided once per block of NR values and wave (common.h: fldr_guard_trips)
                float abs_sum = 0.0f;
#pragma unroll
                for (int r = 0; r < NR; ++r) abs_sum += fabsf(vv[r]);
                const bool guard = fldr_guard_trips(abs_sum);
                bool bad = false;                                                   // (one flag raise per tile of outputs: the raise is cold code, kept out of the unrolled body)
#pragma unroll
                for (int r0 = 0; r0 < NR; r0 += 4) {
                    const int co0 = MT == 32 ? m * 32 + 8 * (r0 >> 2) + 4 * lk : m * 16 + lk * 4;
                    h4 hi, lo;
                    {
                        float xs[4];
                        _Float16 hs[4], ls[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) xs[r] = co0 + r < a.cout_store ? vv[r0 + r] : 0.0f;
                        if (guard) {
                            {
#pragma unroll
                                for (int r = 0; r < 4; ++r) fldr_split_hl(xs[r], hs[r], ls[r], bad);
                            }
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) fldr_split_plain(xs[r], hs[r], ls[r]);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) { hi[r] = hs[r]; lo[r] = ls[r]; }
                    }
                    if ((co0 >> 3) < ((a.cout_store + 7) >> 3) && pix_ok) {     // every quad of a stored group: padding channels are written as zeros, never left uninitialised
                        unsigned char* q = spkn + ((int64_t)(co0 >> 3) * 2 * HWo + po) * 16 + ((co0 >> 2) & 1) * 8;
                        *reinterpret_cast<h4*>(q) = hi;
                        *reinterpret_cast<h4*>(q + HWo * 16) = lo;
                    }
                }
                if (guard) fldr_note_range(bad);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Persistent variant (enc1 / enc2: the weights of ALL chunks fit in LDS next to the two input stages).
//
// The kernel above is latency-bound, not HBM-bound: its 8 x 32 tiles are 7 short iterations (enc1) that each expose one
// global-load round trip (the next chunk is requested at the top of an iteration and needed at its end) and one weight
// DMA round trip before the barrier, plus a per-tile prologue (geometry, first chunk) and epilogue; measured on enc1:
// 405 us, 232 us with the input loads compiled out, unchanged without the MFMAs.  Here
//   * workgroups are persistent (2 per CU) and walk an XCD-contiguous list of tiles: the weights are fetched once per
//     workgroup and the (tile, chunk) iterations of all its tiles form ONE pipeline;
//   * the inputs of iteration j are requested during iteration j-2 (two register sets), split and written to LDS at the
//     end of iteration j-1 and consumed in iteration j: every load has a whole iteration to land;
//   * loads are unconditional from clamped offsets (one SGPR plane base + a 32-bit lane offset each) and masked afterwards.
// Same operand layout, same MFMA order as the kernel above: bit-identical results.
// ------------------------------------------------------------------------------------------------
//   * V4 (odd tile-grid shift, rows / planes 16-byte aligned: every 4K launch): the 68-float window [ix0 - 1, ix0 + 67) of a
//     row starts on a 16-byte boundary, and an item is one (row pair, aligned quad of columns): TWO 16-byte loads instead of
//     eight 4-byte ones.  This chip streams 16-byte lanes at 5.5-5.9 TB/s and 4-byte lanes at 3.9-4.1
//     (tools/ubench/plane_bw_bench), and enc1 sat at that second figure; the two extra floats of a row land in the padding
//     of the LDS lines.  Same LDS image, same MFMA order: bit-identical results.
template <int MT, int NMT, int PT, bool V4>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void conv4x4s2_pers_kernel(S2Args a) {
    // (two workgroups per CU by LDS = 2 waves per SIMD: the full 256-VGPR budget, no spills — a scratch reload would sit in
    // the same in-order counter as the prefetched inputs and drain them)
    using Cfg = S2Cfg<MT, NMT>;
    constexpr int KL = Cfg::KL, STEPS = Cfg::STEPS, IWHP = Cfg::IWHP, KIND = Cfg::KIND, CHS = Cfg::CHS;
    constexpr int TPR = S2_TW / MT, RPW = PT / TPR;
    static_assert(4 * RPW == S2_TH, "tile height");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int lj = lane & (MT - 1), lg = lane / MT;
    const int cin_pad = (a.cin + S2_CC - 1) / S2_CC * S2_CC;
    const int n_chunks = cin_pad / S2_CC;
    unsigned char* const wall = smem;                                         // [chunk][W_BYTES]
    unsigned char* const xst = smem + n_chunks * Cfg::W_BYTES;                // two input stages of X_DW dwords

    // tiles of this workgroup: XCD x owns tiles [x*tpx, (x+1)*tpx) of the N * n_tiles (sample-major) list
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int total_tiles = a.N * a.n_tiles;
    const int t_end = min((xcd + 1) * a.tiles_per_xcd, total_tiles);
    const int t_first = xcd * a.tiles_per_xcd + slot;
    if (t_first >= t_end) return;                                             // workgroup-uniform
    const int my_tiles = (t_end - t_first + a.wgs_per_xcd - 1) / a.wgs_per_xcd;
    const int total = my_tiles * n_chunks;

    // ---- weights: everything, once ----
    {
        const unsigned char* g = reinterpret_cast<const unsigned char*>(a.wpack + S2_HDR);
        const int wbytes = n_chunks * Cfg::W_BYTES;                           // multiple of 1 KB
        for (int piece = wave * 64; piece * 16 < wbytes; piece += 256)        // wave-uniform
            __builtin_amdgcn_global_load_lds((s2_gptr_t)(g + (piece + lane) * 16), (s2_lptr_t)(wall + piece * 16), 16, 0, 0);
    }

    // ---- per-channel plane pointers (sample 0) and batch strides in VGPR lanes (lane l = channel l; cin <= 64) ----
    unsigned long long ctab = 0ull;
    long long btab = 0;
    if (lane < a.cin) {
        int s = 0;
        while (s + 1 < a.n_src && lane >= a.src_cbegin[s + 1]) ++s;
        ctab = (unsigned long long)reinterpret_cast<uintptr_t>(a.src[s] + (int64_t)(lane - a.src_cbegin[s]) * a.src_cstride[s]);
        btab = a.src_bstride[s];
    }

    // ---- staging geometry ----
    // scalar path: an item = one column of one row pair (two 4-byte loads); V4: one aligned quad of columns of one row pair
    // (two 16-byte loads).  Tile-independent: the LDS dword(s) of item i; per tile: clamped byte offsets of its two rows and
    // their validity bits.
    constexpr int NIT = V4 ? 3 : S2_NI;                                       // items per lane and channel
    constexpr int QPR = 17;                                                   // V4: quads per row (68 floats)
    constexpr int SETN = V4 ? 8 * NIT + 2 : 2 * S2_NI + 2;                    // floats of a register set (+ the two validity words)
    int l_dw[NIT], it_pr2[NIT], it_x[NIT];                                    // V4: l_dw = even-plane dword of the quad, it_x = 4 * quad - 1
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int e = lane + 64 * i;
        if constexpr (V4) {
            const int pr = e / QPR, qd = e % QPR;
            it_pr2[i] = 2 * pr; it_x[i] = 4 * qd - 1;
            l_dw[i] = e < S2_RP * QPR ? (pr * 2) * IWHP + 2 * qd : -1;
        } else {
            const int pr = e / S2_IW, x = e % S2_IW;
            it_pr2[i] = 2 * pr; it_x[i] = x;
            l_dw[i] = e < S2_RP * S2_IW ? (pr * 2 + (x & 1)) * IWHP + (x >> 1) : -1;
        }
    }
    uint32_t voff0[NIT], voff1[NIT];
    unsigned iss_m0 = 0, iss_m1 = 0;
    int iss_k = 0, iss_c = 0, iss_n = 0;
    auto issue_geometry = [&]() __attribute__((always_inline)) {
        const int t = t_first + iss_k * a.wgs_per_xcd;
        iss_n = t / a.n_tiles;
        const int tile = t - iss_n * a.n_tiles;
        const int ty = tile / a.tiles_x;
        const int iy0 = ty * S2_TH * 2 - 1, ix0 = ((tile - ty * a.tiles_x) * S2_TW - a.x_shift) * 2 - 1;
        iss_m0 = 0; iss_m1 = 0;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int gy0 = iy0 + it_pr2[i], gy1 = gy0 + 1, gx = ix0 + it_x[i];
            // (V4: gx is a multiple of 4 and Win % 4 == 0, so a quad is inside the image or outside it as a whole)
            const bool okx = l_dw[i] >= 0 && gx >= 0 && gx < a.Win;
            const bool ok0 = okx && gy0 >= 0 && gy0 < a.Hin, ok1 = okx && gy1 >= 0 && gy1 < a.Hin;
            iss_m0 |= ok0 ? (1u << i) : 0u;
            iss_m1 |= ok1 ? (1u << i) : 0u;
            const int cx = okx ? gx : 0;                                      // (lanes without an item, columns outside: any valid address)
            voff0[i] = (uint32_t)(min(max(gy0, 0), a.Hin - 1) * a.Win + cx) * 4u;
            voff1[i] = (uint32_t)(min(max(gy1, 0), a.Hin - 1) * a.Win + cx) * 4u;
        }
    };
    // request the inputs of the issue side's (tile, chunk) into `dst` and step the issue side; wave w stages channel w
    // (the two validity words travel in dst[SETN - 2], dst[SETN - 1])
    auto issue_loads = [&](float (&dst)[SETN]) __attribute__((always_inline)) {
        const int c = iss_c * S2_CC + wave_u;                                 // wave-uniform
        const unsigned long long e = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(ctab >> 32), c & 63) << 32) |
                                     (unsigned)__builtin_amdgcn_readlane((int)(unsigned)ctab, c & 63);
        const long long bs = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)((unsigned long long)btab >> 32), c & 63) << 32) |
                                         (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(unsigned long long)btab, c & 63));
        const bool live = c < a.cin;                                          // padding channels of the last chunk read plane 0 and are masked
        const auto* base = (const __attribute__((address_space(1))) char*)(live ? e + (unsigned long long)(iss_n * bs) * 4ull : (unsigned long long)reinterpret_cast<uintptr_t>(a.src[0]));
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            if constexpr (V4) {
                const s2_f4 q0 = *reinterpret_cast<const __attribute__((address_space(1))) s2_f4*>(base + voff0[i]);
                const s2_f4 q1 = *reinterpret_cast<const __attribute__((address_space(1))) s2_f4*>(base + voff1[i]);
#pragma unroll
                for (int j = 0; j < 4; ++j) { dst[8 * i + j] = q0[j]; dst[8 * i + 4 + j] = q1[j]; }
            } else {
#if defined(S2_NT) && (S2_NT & 2)
                dst[2 * i] = __builtin_nontemporal_load(reinterpret_cast<const __attribute__((address_space(1))) float*>(base + voff0[i]));
                dst[2 * i + 1] = __builtin_nontemporal_load(reinterpret_cast<const __attribute__((address_space(1))) float*>(base + voff1[i]));
#else
                dst[2 * i] = *reinterpret_cast<const __attribute__((address_space(1))) float*>(base + voff0[i]);
                dst[2 * i + 1] = *reinterpret_cast<const __attribute__((address_space(1))) float*>(base + voff1[i]);
#endif
            }
        }
        dst[SETN - 2] = __uint_as_float(live ? iss_m0 : 0u); dst[SETN - 1] = __uint_as_float(live ? iss_m1 : 0u);
        if (++iss_c == n_chunks) { iss_c = 0; if (iss_k + 1 < my_tiles) { ++iss_k; issue_geometry(); } }      // (past the end: the last tile again)
    };
    // One staging call splits up to 24 values per thread.  The range guard of the split is decided once per call and wave: pass 1 masks the
    // values and sums their magnitudes, fldr_guard_trips (common.h) picks the guarded per-value split or the plain one for the whole call.
    auto store_inputs = [&](unsigned char* stage, const float (&src)[SETN]) __attribute__((always_inline)) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const unsigned m0 = __float_as_uint(src[SETN - 2]), m1 = __float_as_uint(src[SETN - 1]);
        h2* hi = reinterpret_cast<h2*>(stage) + wave * CHS;
        h2* lo = hi + KIND;
        constexpr int PER = V4 ? 8 : 2;
        float xs[NIT][PER];
        float abs_sum = 0.0f;
        // rows outside the image are zeroed by per-item mask bits; a wave whose live items are all inside (every tile away from the
        // borders) skips the selects (wave-uniform test)
        unsigned need = 0u;                                                   // this thread's live items
#pragma unroll
        for (int i = 0; i < NIT; ++i) need |= l_dw[i] >= 0 ? (1u << i) : 0u;
        const bool masked = __builtin_amdgcn_ballot_w64((m0 & m1 & need) != need) != 0ull;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            if constexpr (V4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xs[i][2 * j] = (!masked || ((m0 >> i) & 1u)) ? src[8 * i + j] : 0.0f;
                    xs[i][2 * j + 1] = (!masked || ((m1 >> i) & 1u)) ? src[8 * i + 4 + j] : 0.0f;
                }
            } else {
                xs[i][0] = (!masked || ((m0 >> i) & 1u)) ? src[2 * i] : 0.0f;
                xs[i][1] = (!masked || ((m1 >> i) & 1u)) ? src[2 * i + 1] : 0.0f;
            }
            if (l_dw[i] >= 0) {
#pragma unroll
                for (int k = 0; k < PER; ++k) abs_sum += fabsf(xs[i][k]);
            }
        }
        auto emit = [&](auto guardedc) __attribute__((always_inline)) {
            constexpr bool GUARDED = decltype(guardedc)::value;
            bool bad = false;                                                      // (one flag raise per emit: the raise is cold code, kept out of the unrolled body)
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                if (l_dw[i] < 0) continue;
                _Float16 hs[PER], ls[PER];
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    if constexpr (GUARDED) fldr_split_hl(xs[i][k], hs[k], ls[k], bad); else fldr_split_plain(xs[i][k], hs[k], ls[k]);
                }
                if constexpr (V4) {
                    // columns x = 4 q - 1 .. 4 q + 2 of the window: odd plane index 2 q - 1 (x = -1: the padding in front of the line),
                    // even 2 q, odd 2 q, even 2 q + 1 (x = 66: the padding behind it)
                    h2 h[4], l[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { h[j][0] = hs[2 * j]; h[j][1] = hs[2 * j + 1]; l[j][0] = ls[2 * j]; l[j][1] = ls[2 * j + 1]; }
                    const int ev = l_dw[i], od = l_dw[i] + IWHP - 1;
                    hi[ev] = h[1]; hi[ev + 1] = h[3]; hi[od] = h[0]; hi[od + 1] = h[2];
                    lo[ev] = l[1]; lo[ev + 1] = l[3]; lo[od] = l[0]; lo[od + 1] = l[2];
                } else {
                    h2 h, l;
                    h[0] = hs[0]; h[1] = hs[1]; l[0] = ls[0]; l[1] = ls[1];
                    hi[l_dw[i]] = h;
                    lo[l_dw[i]] = l;
                }
            }
            if constexpr (GUARDED) fldr_note_range(bad);
        };
        if (fldr_guard_trips(abs_sum)) emit(std::true_type{}); else emit(std::false_type{});
    };

    // ---- operand geometry (as above) ----
    const int cps = KL / 2;
    const int b_ch = cps == 2 ? (lg >> 1) : 0;
    const int b_rp = cps == 2 ? (lg & 1) : lg;
    int boff[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int r = wave * RPW + p / TPR, xl = (p % TPR) * MT + lj;
        boff[p] = b_ch * CHS + ((r + b_rp) * 2) * IWHP + xl;
    }
    typedef typename std::conditional<MT == 32, s2_f16, s2_f4>::type acc_t;
    constexpr int NR = MT == 32 ? 16 : 4;
    acc_t acc[NMT][PT];
#pragma unroll
    for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int r = 0; r < NR; ++r) acc[m][p][r] = 0.0f;

    const float inv_scale = a.wpack[0];
    const int64_t HWo = (int64_t)a.Hout * a.Wout;
    int cur_k = 0, cur_c = 0;
    float bias_r[NMT][NR];                                                    // this lane's output channels, fetched once
#pragma unroll
    for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int co = MT == 32 ? m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg : m * 16 + lg * 4 + r;
            bias_r[m][r] = a.bias ? a.bias[co < a.cout ? co : a.cout - 1] : 0.0f;
        }

    auto mfma_phase = [&](const unsigned char* xs, int ch) __attribute__((always_inline)) {
        const int* xin = reinterpret_cast<const int*>(xs);
        const unsigned char* win = wall + ch * Cfg::W_BYTES + lane * 16;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            s2_h8 ah[NMT], al[NMT];
#pragma unroll
            for (int m = 0; m < NMT; ++m) {
                ah[m] = *reinterpret_cast<const s2_h8*>(win + ((s * NMT + m) * 2 + 0) * 1024);
                al[m] = *reinterpret_cast<const s2_h8*>(win + ((s * NMT + m) * 2 + 1) * 1024);
            }
            const int cbase = s * cps * CHS;
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                const int* q = xin + cbase + boff[p];
                s2_i4 bhi, blo;
                bhi[0] = q[0]; bhi[1] = q[1]; bhi[2] = q[IWHP]; bhi[3] = q[IWHP + 1];
                blo[0] = q[KIND]; blo[1] = q[KIND + 1]; blo[2] = q[KIND + IWHP]; blo[3] = q[KIND + IWHP + 1];
                const s2_h8 bh = __builtin_bit_cast(s2_h8, bhi), bl = __builtin_bit_cast(s2_h8, blo);
#pragma unroll
                for (int m = 0; m < NMT; ++m) {
                    if constexpr (MT == 32) {
                        acc[m][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh, acc[m][p], 0, 0, 0);
                        acc[m][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl, acc[m][p], 0, 0, 0);
                        acc[m][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh, acc[m][p], 0, 0, 0);
                    } else {
                        acc[m][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh, acc[m][p], 0, 0, 0);
                        acc[m][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl, acc[m][p], 0, 0, 0);
                        acc[m][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh, acc[m][p], 0, 0, 0);
                    }
                }
            }
        }
    };
    // undo the weight scale, bias, ReLU, store (lane = pixel column, registers = output channels); clears the accumulators
    auto epilogue = [&]() __attribute__((always_inline)) {
        const int t = t_first + cur_k * a.wgs_per_xcd;
        const int n = t / a.n_tiles, tile = t - n * a.n_tiles;
        const int ty = tile / a.tiles_x;
        const int oy0 = ty * S2_TH, ox0 = (tile - ty * a.tiles_x) * S2_TW - a.x_shift;
        float* outn = a.out ? a.out + (int64_t)n * a.cout_store * HWo : nullptr;
        unsigned char* spkn = a.out_spk ? a.out_spk + (int64_t)n * ((a.cout_store + 7) >> 3) * 2 * HWo * 16 : nullptr;
        const int lk = lg;
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const int oy = oy0 + wave * RPW + p / TPR;
            const int ox = ox0 + (p % TPR) * MT + lj;
            const bool pix_ok = oy < a.Hout && ox >= 0 && ox < a.Wout;
            const int64_t po = pix_ok ? (int64_t)oy * a.Wout + ox : 0;
#pragma unroll
            for (int m = 0; m < NMT; ++m) {
                float vv[NR];
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int co = MT == 32 ? m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk : m * 16 + lk * 4 + r;
                    float v = acc[m][p][r] * inv_scale + bias_r[m][r];
                    if (a.relu) v = fmaxf(v, 0.0f);
                    vv[r] = v;
                    acc[m][p][r] = 0.0f;
#if defined(S2_NT) && (S2_NT & 1)
                    if (outn && co < a.cout_store && pix_ok) __builtin_nontemporal_store(v, &outn[(int64_t)co * HWo + po]);
#else
                    if (outn && co < a.cout_store && pix_ok) outn[(int64_t)co * HWo + po] = v;
#endif
                }
                if (spkn) {
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                    // the range guard of the split, decided once per block of NR values and wave (common.h: fldr_guard_trips)
                    float abs_sum = 0.0f;
#pragma unroll
                    for (int r = 0; r < NR; ++r) abs_sum += fabsf(vv[r]);
                    const bool guard = fldr_guard_trips(abs_sum);
                    bool bad = false;                                                   // (one flag raise per tile of outputs: the raise is cold code, kept out of the unrolled body)
#pragma unroll
                    for (int r0 = 0; r0 < NR; r0 += 4) {
                        const int co0 = MT == 32 ? m * 32 + 8 * (r0 >> 2) + 4 * lk : m * 16 + lk * 4;
                        h4 hi, lo;
                        {
                            float xs[4];
                            _Float16 hs[4], ls[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) xs[r] = co0 + r < a.cout_store ? vv[r0 + r] : 0.0f;
                            if (guard) {
                                {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) fldr_split_hl(xs[r], hs[r], ls[r], bad);
                                }
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; ++r) fldr_split_plain(xs[r], hs[r], ls[r]);
                            }
#pragma unroll
                            for (int r = 0; r < 4; ++r) { hi[r] = hs[r]; lo[r] = ls[r]; }
                        }
                        if ((co0 >> 3) < ((a.cout_store + 7) >> 3) && pix_ok) {     // every quad of a stored group: padding channels are written as zeros, never left uninitialised
                            unsigned char* q = spkn + ((int64_t)(co0 >> 3) * 2 * HWo + po) * 16 + ((co0 >> 2) & 1) * 8;
                            *reinterpret_cast<h4*>(q) = hi;
                            *reinterpret_cast<h4*>(q + HWo * 16) = lo;
                        }
                    }
                    if (guard) fldr_note_range(bad);
                }
            }
        }
    };

    // ---- pipeline: iteration j computes on stage j & 1; its inputs were requested in iteration j-2 ----
    // Every iteration issues its 20 loads UNCONDITIONALLY (past the end they re-read the last tile: harmless) and the
    // epilogue's stores come after the LDS write, so that between the request of a register set and its use there is
    // exactly one other request: the compiler's wait becomes a counted vmcnt(20) instead of a drain.
    float preA[SETN], preB[SETN];
    issue_geometry();
    issue_loads(preA);                                                        // iteration 0
    issue_loads(preB);                                                        // iteration 1
    store_inputs(xst, preA);
    __syncthreads();                                                          // (drains the weight DMA too)
    unsigned char* const xs0 = xst;
    unsigned char* const xs1 = xst + Cfg::X_DW * 4;
    for (int j = 0; j < total; j += 2) {
        // even iteration: stage 0; requests iteration j+2 into A (A held iteration j: consumed), writes iteration j+1 from B
        issue_loads(preA);
        mfma_phase(xs0, cur_c);
        store_inputs(xs1, preB);
        if (cur_c == n_chunks - 1) { epilogue(); cur_c = 0; ++cur_k; } else ++cur_c;
        __syncthreads();
        if (j + 1 >= total) break;
        // odd iteration: stage 1; requests j+3 into B, writes j+2 from A
        issue_loads(preB);
        mfma_phase(xs1, cur_c);
        store_inputs(xs0, preA);
        if (cur_c == n_chunks - 1) { epilogue(); cur_c = 0; ++cur_k; } else ++cur_c;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Persistent kernel on a SPLIT-PACKED source (round 3; enc2 reading enc1's packed twin: enc1 then writes no fp32 copy of its
// 16 half-resolution planes, -141 MB per 4K forward).  Same LDS image, weight pack, MFMA order and epilogue as the kernel above;
// what changes is the staging: a packed pixel is 16 bytes = the hi (or lo) halves of 8 channels, i.e. of TWO 4-channel chunks, so
// the pipeline runs in group iterations — {MFMA phases of chunks 2g and 2g+1 from the two LDS stages; barrier; write the next
// group's 8 channels into both stages from registers; request the group after that; barrier} — with one register set in flight
// (the loads of a group have the two MFMA phases of the previous one to land).  An item is one (row pair, column): four 16-byte
// loads (hi / lo x two rows), sixteen dwords written (8 channels x hi / lo, each dword = the two rows' halves, picked by
// v_perm_b32): no conversion work at all — the operands are the producer's hi / lo halves (the split this kernel's fp32-source
// twin would derive from hi + lo, except where lo was rounded up to a whole ulp of hi: same value, other split), so the results
// agree with it to fp32 accumulation rounding.  cin % 8 == 0, one source.
// ------------------------------------------------------------------------------------------------
template <int MT, int NMT, int PT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void conv4x4s2_pers_spk_kernel(S2Args a) {
    if (blockIdx.y) { a.wpack = a.wpack2; a.bias = a.bias2; a.out = a.out2; a.out_spk = a.out_spk2; }      // pair launch: the second half (uniform)
    using Cfg = S2Cfg<MT, NMT>;
    constexpr int KL = Cfg::KL, STEPS = Cfg::STEPS, IWHP = Cfg::IWHP, KIND = Cfg::KIND, CHS = Cfg::CHS;
    constexpr int TPR = S2_TW / MT, RPW = PT / TPR;
    static_assert(4 * RPW == S2_TH, "tile height");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lj = lane & (MT - 1), lg = lane / MT;
    const int n_chunks = a.cin / S2_CC, n_groups = a.cin / 8;                 // cin % 8 == 0 (host-checked)
    unsigned char* const wall = smem;                                         // [chunk][W_BYTES]
    unsigned char* const xst = smem + n_chunks * Cfg::W_BYTES;                // two input stages of X_DW dwords
    unsigned char* const xs0 = xst;
    unsigned char* const xs1 = xst + Cfg::X_DW * 4;

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int total_tiles = a.N * a.n_tiles;
    const int t_end = min((xcd + 1) * a.tiles_per_xcd, total_tiles);
    const int t_first = xcd * a.tiles_per_xcd + slot;
    if (t_first >= t_end) return;                                             // workgroup-uniform
    const int my_tiles = (t_end - t_first + a.wgs_per_xcd - 1) / a.wgs_per_xcd;
    const int total = my_tiles * n_groups;                                    // group iterations

    // ---- weights: everything, once ----
    {
        const unsigned char* g = reinterpret_cast<const unsigned char*>(a.wpack + S2_HDR);
        const int wbytes = n_chunks * Cfg::W_BYTES;                           // multiple of 1 KB
        for (int piece = wave * 64; piece * 16 < wbytes; piece += 256)        // wave-uniform
            __builtin_amdgcn_global_load_lds((s2_gptr_t)(g + (piece + lane) * 16), (s2_lptr_t)(wall + piece * 16), 16, 0, 0);
    }

    // ---- staging geometry: item e = thread + 256 i -> (row pair, column) of the 9 x 66 window ----
    constexpr int NIT = (S2_RP * S2_IW + 255) / 256;                          // 3
    int l_dw[NIT], it_pr2[NIT], it_x[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int e = tid + 256 * i;
        const int pr = e / S2_IW, x = e % S2_IW;
        it_pr2[i] = 2 * pr; it_x[i] = x;
        l_dw[i] = e < S2_RP * S2_IW ? (pr * 2 + (x & 1)) * IWHP + (x >> 1) : -1;
    }
    const int64_t HWi = (int64_t)a.Hin * a.Win;
    const int64_t src_b = a.src_bstride[0];                                   // BYTES between samples of the packed source
    const unsigned char* const src0 = reinterpret_cast<const unsigned char*>(a.src[0]);
    uint32_t voff0[NIT], voff1[NIT];                                          // byte offsets of the item's two pixels inside a (group, kind) plane
    uint32_t vmask[NIT];                                                      // 0x0000FFFF: row 0 inside the image, 0xFFFF0000: row 1
    int iss_k = 0, iss_g = 0, iss_n = 0;
    auto issue_geometry = [&]() __attribute__((always_inline)) {
        const int t = t_first + iss_k * a.wgs_per_xcd;
        iss_n = t / a.n_tiles;
        const int tile = t - iss_n * a.n_tiles;
        const int ty = tile / a.tiles_x;
        const int iy0 = ty * S2_TH * 2 - 1, ix0 = ((tile - ty * a.tiles_x) * S2_TW - a.x_shift) * 2 - 1;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int gy0 = iy0 + it_pr2[i], gy1 = gy0 + 1, gx = ix0 + it_x[i];
            const bool okx = l_dw[i] >= 0 && gx >= 0 && gx < a.Win;
            const bool ok0 = okx && gy0 >= 0 && gy0 < a.Hin, ok1 = okx && gy1 >= 0 && gy1 < a.Hin;
            vmask[i] = (ok0 ? 0x0000FFFFu : 0u) | (ok1 ? 0xFFFF0000u : 0u);
            const int cx = okx ? gx : 0;
            voff0[i] = (uint32_t)(min(max(gy0, 0), a.Hin - 1) * a.Win + cx) * 16u;
            voff1[i] = (uint32_t)(min(max(gy1, 0), a.Hin - 1) * a.Win + cx) * 16u;
        }
    };
    // request the 8 channels of the issue side's (tile, group) and step the issue side (past the end: the last tile again)
    s2_i4 R[NIT][4];                                                          // [item][hi row 0, hi row 1, lo row 0, lo row 1]
    uint32_t Rmask[NIT];
    auto issue_loads = [&]() __attribute__((always_inline)) {
        const auto* gh = (const __attribute__((address_space(1))) char*)(src0 + (int64_t)iss_n * src_b + (int64_t)(iss_g * 2) * HWi * 16);
        const auto* gl = gh + HWi * 16;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            R[i][0] = *reinterpret_cast<const __attribute__((address_space(1))) s2_i4*>(gh + voff0[i]);
            R[i][1] = *reinterpret_cast<const __attribute__((address_space(1))) s2_i4*>(gh + voff1[i]);
            R[i][2] = *reinterpret_cast<const __attribute__((address_space(1))) s2_i4*>(gl + voff0[i]);
            R[i][3] = *reinterpret_cast<const __attribute__((address_space(1))) s2_i4*>(gl + voff1[i]);
            Rmask[i] = vmask[i];
        }
        if (++iss_g == n_groups) { iss_g = 0; if (iss_k + 1 < my_tiles) { ++iss_k; issue_geometry(); } }
    };
    // the register set -> both LDS stages: channel c of the group goes to plane c & 3 of stage c >> 2
    auto store_inputs = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            if (l_dw[i] < 0) continue;
#pragma unroll
            for (int kind = 0; kind < 2; ++kind) {
                const s2_i4 r0 = R[i][2 * kind], r1 = R[i][2 * kind + 1];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    // halves c of the two rows: dword c >> 1, low half for even c (bytes 1:0), high half for odd c (bytes 3:2)
                    const uint32_t v = __builtin_amdgcn_perm((uint32_t)r1[c >> 1], (uint32_t)r0[c >> 1], (c & 1) ? 0x07060302u : 0x05040100u) & Rmask[i];
                    uint32_t* dst = reinterpret_cast<uint32_t*>((c >> 2) ? xs1 : xs0) + (c & 3) * CHS + kind * KIND + l_dw[i];
                    *dst = v;
                }
            }
        }
    };

    // ---- operand geometry, accumulators, bias, MFMA phase, epilogue: as in conv4x4s2_pers_kernel ----
    const int cps = KL / 2;
    const int b_ch = cps == 2 ? (lg >> 1) : 0;
    const int b_rp = cps == 2 ? (lg & 1) : lg;
    int boff[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int r = wave * RPW + p / TPR, xl = (p % TPR) * MT + lj;
        boff[p] = b_ch * CHS + ((r + b_rp) * 2) * IWHP + xl;
    }
    typedef typename std::conditional<MT == 32, s2_f16, s2_f4>::type acc_t;
    constexpr int NR = MT == 32 ? 16 : 4;
    acc_t acc[NMT][PT];
#pragma unroll
    for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int r = 0; r < NR; ++r) acc[m][p][r] = 0.0f;
    const float inv_scale = a.wpack[0];
    const int64_t HWo = (int64_t)a.Hout * a.Wout;
    int cur_k = 0, cur_g = 0;
    float bias_r[NMT][NR];
#pragma unroll
    for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int co = MT == 32 ? m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg : m * 16 + lg * 4 + r;
            bias_r[m][r] = a.bias ? a.bias[co < a.cout ? co : a.cout - 1] : 0.0f;
        }
    auto mfma_phase = [&](const unsigned char* xs, int ch) __attribute__((always_inline)) {
        const int* xin = reinterpret_cast<const int*>(xs);
        const unsigned char* win = wall + ch * Cfg::W_BYTES + lane * 16;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            s2_h8 ah[NMT], al[NMT];
#pragma unroll
            for (int m = 0; m < NMT; ++m) {
                ah[m] = *reinterpret_cast<const s2_h8*>(win + ((s * NMT + m) * 2 + 0) * 1024);
                al[m] = *reinterpret_cast<const s2_h8*>(win + ((s * NMT + m) * 2 + 1) * 1024);
            }
            const int cbase = s * cps * CHS;
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                const int* q = xin + cbase + boff[p];
                s2_i4 bhi, blo;
                bhi[0] = q[0]; bhi[1] = q[1]; bhi[2] = q[IWHP]; bhi[3] = q[IWHP + 1];
                blo[0] = q[KIND]; blo[1] = q[KIND + 1]; blo[2] = q[KIND + IWHP]; blo[3] = q[KIND + IWHP + 1];
                const s2_h8 bh = __builtin_bit_cast(s2_h8, bhi), bl = __builtin_bit_cast(s2_h8, blo);
#pragma unroll
                for (int m = 0; m < NMT; ++m) {
                    if constexpr (MT == 32) {
                        acc[m][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh, acc[m][p], 0, 0, 0);
                        acc[m][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl, acc[m][p], 0, 0, 0);
                        acc[m][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh, acc[m][p], 0, 0, 0);
                    } else {
                        acc[m][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh, acc[m][p], 0, 0, 0);
                        acc[m][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl, acc[m][p], 0, 0, 0);
                        acc[m][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh, acc[m][p], 0, 0, 0);
                    }
                }
            }
        }
    };
    auto epilogue = [&]() __attribute__((always_inline)) {
        const int t = t_first + cur_k * a.wgs_per_xcd;
        const int n = t / a.n_tiles, tile = t - n * a.n_tiles;
        const int ty = tile / a.tiles_x;
        const int oy0 = ty * S2_TH, ox0 = (tile - ty * a.tiles_x) * S2_TW - a.x_shift;
        float* outn = a.out ? a.out + (int64_t)n * a.cout_store * HWo : nullptr;
        unsigned char* spkn = a.out_spk ? a.out_spk + (int64_t)n * ((a.cout_store + 7) >> 3) * 2 * HWo * 16 : nullptr;
        const int lk = lg;
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const int oy = oy0 + wave * RPW + p / TPR;
            const int ox = ox0 + (p % TPR) * MT + lj;
            const bool pix_ok = oy < a.Hout && ox >= 0 && ox < a.Wout;
            const int64_t po = pix_ok ? (int64_t)oy * a.Wout + ox : 0;
#pragma unroll
            for (int m = 0; m < NMT; ++m) {
                float vv[NR];
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int co = MT == 32 ? m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk : m * 16 + lk * 4 + r;
                    float v = acc[m][p][r] * inv_scale + bias_r[m][r];
                    if (a.relu) v = fmaxf(v, 0.0f);
                    vv[r] = v;
                    acc[m][p][r] = 0.0f;
                    if (outn && co < a.cout_store && pix_ok) outn[(int64_t)co * HWo + po] = v;
                }
                if (spkn) {
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                    // the range guard of the split, decided once per block of NR values and wave (common.h: fldr_guard_trips)
                    float abs_sum = 0.0f;
#pragma unroll
                    for (int r = 0; r < NR; ++r) abs_sum += fabsf(vv[r]);
                    const bool guard = fldr_guard_trips(abs_sum);
                    bool bad = false;                                                   // (one flag raise per tile of outputs: the raise is cold code, kept out of the unrolled body)
#pragma unroll
                    for (int r0 = 0; r0 < NR; r0 += 4) {
                        const int co0 = MT == 32 ? m * 32 + 8 * (r0 >> 2) + 4 * lk : m * 16 + lk * 4;
                        h4 hi, lo;
                        {
                            float xs[4];
                            _Float16 hs[4], ls[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) xs[r] = co0 + r < a.cout_store ? vv[r0 + r] : 0.0f;
                            if (guard) {
                                {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) fldr_split_hl(xs[r], hs[r], ls[r], bad);
                                }
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; ++r) fldr_split_plain(xs[r], hs[r], ls[r]);
                            }
#pragma unroll
                            for (int r = 0; r < 4; ++r) { hi[r] = hs[r]; lo[r] = ls[r]; }
                        }
                        if ((co0 >> 3) < ((a.cout_store + 7) >> 3) && pix_ok) {
                            unsigned char* q = spkn + ((int64_t)(co0 >> 3) * 2 * HWo + po) * 16 + ((co0 >> 2) & 1) * 8;
                            *reinterpret_cast<h4*>(q) = hi;
                            *reinterpret_cast<h4*>(q + HWo * 16) = lo;
                        }
                    }
                    if (guard) fldr_note_range(bad);
                }
            }
        }
    };

    // ---- pipeline ----
    issue_geometry();
    issue_loads();                                                            // group iteration 0
    store_inputs();
    issue_loads();                                                            // group iteration 1 (or the last tile again)
    __syncthreads();                                                          // (drains the weight DMA too)
    for (int j = 0; j < total; ++j) {
#if !(defined(S2P_ABLATE) && S2P_ABLATE == 2)                             // diagnostic builds (tools/stamps/build_variant.sh): 1 no LDS staging, 2 no MFMAs, 3 no loads, 4 no epilogue
        mfma_phase(xs0, 2 * cur_g);
        mfma_phase(xs1, 2 * cur_g + 1);
#endif
        __syncthreads();                                                      // both stages consumed by everybody
        const bool tile_done = cur_g == n_groups - 1;
#if defined(S2P_ABLATE) && S2P_ABLATE == 1
        if (j + 1 < total) { issue_loads(); for (int i = 0; i < NIT; ++i) for (int k = 0; k < 4; ++k) asm volatile("" :: "v"(R[i][k])); }
#elif defined(S2P_ABLATE) && S2P_ABLATE == 3
        if (j + 1 < total) { store_inputs(); }
#else
        if (j + 1 < total) { store_inputs(); issue_loads(); }                 // group j + 1 into the stages; request group j + 2
#endif
#if defined(S2P_ABLATE) && S2P_ABLATE == 4
        if (tile_done) { cur_g = 0; ++cur_k; } else ++cur_g;
#else
        if (tile_done) { epilogue(); cur_g = 0; ++cur_k; } else ++cur_g;
#endif
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA kernel on a SPLIT-PACKED source, 17..32 output channels (round 4: enc2 16 -> 32 and the two 32-channel halves of enc3).
//
// The kernel above transposes the packed records into its fp32-source LDS image: 48 v_perm + 48 ds_write_b32 per thread and group
// iteration, ~700 vector instructions for 48 matrix instructions, two barriers, no operand read-ahead — ablations at 576x960
// (tools/kernel_bench.py s2spk): enc3 73.9 us = 33 us with the MFMAs compiled out + 41 us of matrix phase for 13-19 us of matrix work.
// A packed pixel IS a matrix operand: its 16-byte record holds 8 channels of one pixel, i.e. 8 consecutive k of
// v_mfma_f32_32x32x16_f16 when k = (tap of a pair, channel).  So here
//   * the (18 x 66 pixel) input window of a (tile, 8-channel group) goes to LDS by LDS-DMA in its natural order, no register staging
//     (a first version de-interleaved the columns by parity at DMA time so that an output row read 32 consecutive records: every DMA
//     instruction then gathered records 32 bytes apart and the kernel ran at half the LDS-DMA rate — 168 MB in 52 us for enc2);
//   * K = 16 = {taps (dy, 2 h) and (dy, 2 h + 1)} x 8 channels: the two lane halves read neighbouring records of the same window row —
//     one ds_read_b128 per operand (lanes 32 bytes apart: a two-way bank conflict, 8 LDS cycles instead of 4, affordable under 48
//     32-cycle MFMAs), 8 tap pairs x 3 split terms per group and 32-pixel block;
//   * two stages, ONE barrier per iteration: the next group's window streams in while this one's 48 MFMAs run with their operands read
//     two micro-steps ahead; the epilogue of a finished tile runs at the top of the next iteration (its stores have a whole iteration
//     to drain before the counted wait in front of the barrier).
// Weights: section D of the pack ([group][tap pair][hi, lo][lane][8 halves], fldr_conv_s2_prepack).  Different summation order than the
// kernels above (taps outer, channels inner): equal to them to fp32 accumulation rounding, tested against fp64.
// ------------------------------------------------------------------------------------------------
#define S2D_NREC (S2_IH * S2_IW)                       // 1188 records per (group, kind) plane: [window row][window column]
#define S2D_PLANE (S2D_NREC * 16)                      // 19,008 B
#define S2D_STAGE (2 * S2D_PLANE)                      // hi plane, lo plane
#ifdef S2D_NOFENCE
#define S2D_FENCE()
#else
#define S2D_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#ifndef S2D_NLOAD
#define S2D_NLOAD 8                                    // loader waves (half of them per plane)
#endif
#define S2D_NP (40 / S2D_NLOAD)                        // DMA pieces of 64 records per loader wave and stage (20 per plane, the last ones overlap)
static_assert((S2D_NLOAD / 2) * S2D_NP * 64 >= S2D_NREC, "DMA pieces cover a plane");
__global__ __launch_bounds__((4 + S2D_NLOAD) * 64) void conv4x4s2_dma_spk_kernel(S2Args a) {
    if (blockIdx.y) { a.wpack = a.wpack2; a.bias = a.bias2; a.out = a.out2; a.out_spk = a.out_spk2; }      // pair launch: the second half (uniform)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 31, lg = lane >> 5;
    const int n_groups = a.cin / 8;
    unsigned char* const wall = smem;                                         // [group][tap pair][kind][1 KB]
    unsigned char* const xst = smem + n_groups * 16384;                       // two stages

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int total_tiles = a.N * a.n_tiles;
    const int t_end = min((xcd + 1) * a.tiles_per_xcd, total_tiles);
    const int t_first = xcd * a.tiles_per_xcd + slot;
    if (t_first >= t_end) return;                                             // workgroup-uniform
    const int my_tiles = (t_end - t_first + a.wgs_per_xcd - 1) / a.wgs_per_xcd;
    const int total = my_tiles * n_groups;                                    // iterations
    // section D of the pack: behind the header and the [chunk][step][kind] section of the kernels above (cin * 512 floats)
    const unsigned char* const wsrc = reinterpret_cast<const unsigned char*>(a.wpack + S2_HDR + (int64_t)a.cin * 512);
    for (int piece = wave * 64; piece * 16 < n_groups * 16384; piece += (4 + S2D_NLOAD) * 64)  // wave-uniform
        __builtin_amdgcn_global_load_lds((s2_gptr_t)(wsrc + (piece + lane) * 16), (s2_lptr_t)(wall + piece * 16), 16, 0, 0);

    // Waves 0-3 are CONSUMERS (MFMAs + epilogue), waves 4-11 LOADERS: an LDS-DMA instruction holds its wave's issue port for ~250 cycles
    // here (a first version, every wave doing both, paid 10 of them = 1.25 us per iteration in front of its MFMAs; four loader waves of
    // 10 pieces each made the loaders' issue + landing time, 2.25 us, the iteration time).
    // ---- loader side: this wave's 5 pieces of plane `kind` ----
    const int lw = wave >= 4 ? wave - 4 : 0;
    const int kind = lw / (S2D_NLOAD / 2);
    int p_rec[S2D_NP];                                                        // first record of piece i (wave-uniform)
    int p_wr[S2D_NP], p_wc[S2D_NP];                                           // window (row, column) of this lane's record
#pragma unroll
    for (int i = 0; i < S2D_NP; ++i) {
        p_rec[i] = min(((lw % (S2D_NLOAD / 2)) * S2D_NP + i) * 64, S2D_NREC - 64);
        const int rec = p_rec[i] + lane;
        p_wr[i] = rec / S2_IW;
        p_wc[i] = rec - p_wr[i] * S2_IW;
    }
    const int64_t HWi = (int64_t)a.Hin * a.Win;
    const unsigned char* const src0 = reinterpret_cast<const unsigned char*>(a.src[0]);
    const unsigned char* const zero_blk = reinterpret_cast<const unsigned char*>(a.wpack + 4);   // 16 zero bytes (pack header)
    uint32_t voff[S2D_NP];                                                    // byte offset inside a (group, kind) plane; ~0u: zero record
    int iss_k = 0, iss_g = 0, iss_n = 0;
    auto issue_geometry = [&]() __attribute__((always_inline)) {
        const int t = t_first + iss_k * a.wgs_per_xcd;
        iss_n = t / a.n_tiles;
        const int tile = t - iss_n * a.n_tiles;
        const int ty = tile / a.tiles_x;
        const int iy0 = ty * S2_TH * 2 - 1, ix0 = (tile - ty * a.tiles_x) * S2_TW * 2 - 1;
#pragma unroll
        for (int i = 0; i < S2D_NP; ++i) {
            const int gy = iy0 + p_wr[i], gx = ix0 + p_wc[i];
            const bool ok = gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
            voff[i] = ok ? (uint32_t)(gy * a.Win + gx) * 16u : ~0u;
        }
    };
    // the window of the issue side's (tile, group) -> stage `st`; steps the issue side (past the end: the last tile again)
    auto issue_dma = [&](unsigned char* st) __attribute__((always_inline)) {
        const unsigned char* base = src0 + (int64_t)iss_n * a.src_bstride[0] + (int64_t)(iss_g * 2 + kind) * HWi * 16;
        unsigned char* dst = st + kind * S2D_PLANE;
#pragma unroll
        for (int i = 0; i < S2D_NP; ++i) {
            const unsigned char* p = voff[i] != ~0u ? base + voff[i] : zero_blk;
            __builtin_amdgcn_global_load_lds((s2_gptr_t)p, (s2_lptr_t)(dst + p_rec[i] * 16), 16, 0, 0);
        }
        if (++iss_g == n_groups) { iss_g = 0; if (iss_k + 1 < my_tiles) { ++iss_k; issue_geometry(); } }
    };

    // ---- consumer side: output rows 2 cw, 2 cw + 1 of the 8 x 32 tile; lane = (pixel lj, tap of the pair lg) ----
    const int cw = wave & 3;
    const uint32_t b_lane = (uint32_t)((cw * 4 * S2_IW + 2 * lj + lg) * 16);    // window row 4 wave (+ 2 p + dy), column 2 lj + lg (+ 2 (tap pair & 1))
    s2_f16 acc[2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;
    const float inv_scale = a.wpack[0];
    const int64_t HWo = (int64_t)a.Hout * a.Wout;
    int cur_k = 0, cur_g = 0;
    // Epilogue of tile k of this workgroup: scale, bias, ReLU, split, stores; clears acc.  (Emitting it in chunks between the next
    // iteration's MFMA micro-steps, on a copy of the accumulators, was built and measured SLOWER — enc2 73.6 vs 56.4 us with four loader
    // waves: the stores hold the wave's issue port and the matrix pipe starves behind them.)
    float bias_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = (r & 3) + 8 * (r >> 2) + 4 * lg;
        bias_r[r] = a.bias ? a.bias[co < a.cout ? co : a.cout - 1] : 0.0f;
    }
    auto epilogue = [&](int k) __attribute__((always_inline)) {
        const int t = t_first + k * a.wgs_per_xcd;
        const int n = t / a.n_tiles, tile = t - n * a.n_tiles;
        const int ty = tile / a.tiles_x;
        const int oy0 = ty * S2_TH, ox0 = (tile - ty * a.tiles_x) * S2_TW;
        float* outn = a.out ? a.out + (int64_t)n * a.cout_store * HWo : nullptr;
        unsigned char* spkn = a.out_spk ? a.out_spk + (int64_t)n * ((a.cout_store + 7) >> 3) * 2 * HWo * 16 : nullptr;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int oy = oy0 + cw * 2 + p, ox = ox0 + lj;
            const bool pix_ok = oy < a.Hout && ox < a.Wout;
            const int64_t po = pix_ok ? (int64_t)oy * a.Wout + ox : 0;
            float vv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (r & 3) + 8 * (r >> 2) + 4 * lg;
                float v = acc[p][r] * inv_scale + bias_r[r];
                if (a.relu) v = fmaxf(v, 0.0f);
                vv[r] = v;
                acc[p][r] = 0.0f;
                if (outn && co < a.cout_store && pix_ok) outn[(int64_t)co * HWo + po] = v;
            }
            if (spkn) {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                // the range guard of the split, decided once per block of 16 values and wave (common.h: fldr_guard_trips)
                float abs_sum = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) abs_sum += fabsf(vv[r]);
                const bool guard = fldr_guard_trips(abs_sum);
                bool bad = false;                                                   // (one flag raise per tile of outputs: the raise is cold code, kept out of the unrolled body)
#pragma unroll
                for (int r0 = 0; r0 < 16; r0 += 4) {
                    const int co0 = 8 * (r0 >> 2) + 4 * lg;
                    h4 hi, lo;
                    {
                        float xs[4];
                        _Float16 hs[4], ls[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) xs[r] = co0 + r < a.cout_store ? vv[r0 + r] : 0.0f;
                        if (guard) {
                            {
#pragma unroll
                                for (int r = 0; r < 4; ++r) fldr_split_hl(xs[r], hs[r], ls[r], bad);
                            }
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) fldr_split_plain(xs[r], hs[r], ls[r]);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) { hi[r] = hs[r]; lo[r] = ls[r]; }
                    }
                    if ((co0 >> 3) < ((a.cout_store + 7) >> 3) && pix_ok) {
                        unsigned char* q = spkn + ((int64_t)(co0 >> 3) * 2 * HWo + po) * 16 + ((co0 >> 2) & 1) * 8;
#if defined(S2D_ABLATE) && S2D_ABLATE == 4                                // diagnostic: the epilogue without its stores
                        asm volatile("" :: "v"(q), "v"(hi), "v"(lo));
#else
                        *reinterpret_cast<h4*>(q) = hi;
                        *reinterpret_cast<h4*>(q + HWo * 16) = lo;
#endif
                    }
                }
                if (guard) fldr_note_range(bad);
            }
        }
    };
    // the 48 MFMAs of group `g` on stage `st`: micro-step m = (tap pair m >> 1, pixel block m & 1); operands read two micro-steps ahead
    auto mfma_group = [&](const unsigned char* st, int g) __attribute__((always_inline)) {
        const unsigned char* wg = wall + g * 16384 + lane * 16;
        const unsigned char* xb = st + b_lane;
        s2_h8 Ah[2], Al[2], Bh[3], Bl[3];
        auto ld_a = [&](int tp) __attribute__((always_inline)) {
            Ah[tp & 1] = *reinterpret_cast<const s2_h8*>(wg + tp * 2048);
            Al[tp & 1] = *reinterpret_cast<const s2_h8*>(wg + tp * 2048 + 1024);
        };
        auto ld_b = [&](int m) __attribute__((always_inline)) {
            const int tp = m >> 1, p = m & 1;
            const int off = ((2 * p + (tp >> 1)) * S2_IW + 2 * (tp & 1)) * 16;
            Bh[m % 3] = *reinterpret_cast<const s2_h8*>(xb + off);
            Bl[m % 3] = *reinterpret_cast<const s2_h8*>(xb + off + S2D_PLANE);
        };
        ld_a(0); ld_b(0); ld_b(1);
        S2D_FENCE();
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            // (fences: left alone, the scheduler sinks the reads to one MFMA before their use and every micro-step waits an LDS round trip)
            if (m + 2 < 16) { if (!(m & 1)) ld_a((m >> 1) + 1); ld_b(m + 2); }
            const s2_h8 ah = Ah[(m >> 1) & 1], al = Al[(m >> 1) & 1], bh = Bh[m % 3], bl = Bl[m % 3];
            acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[m & 1], 0, 0, 0);
            acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[m & 1], 0, 0, 0);
            acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[m & 1], 0, 0, 0);
            S2D_FENCE();
        }
    };

    // ---- pipeline ----
    if (wave >= 4) {
        issue_geometry();
        issue_dma(xst);                                                       // iteration 0 -> stage 0
        __builtin_amdgcn_s_waitcnt(0x0F70);                                   // vmcnt(0): weights and the first window
        __syncthreads();
        for (int j = 0; j < total; ++j) {
#if !(defined(S2D_ABLATE) && S2D_ABLATE == 2)                             // diagnostic builds: 1 no MFMAs, 2 no DMA after the prologue, 3 no epilogue
            issue_dma(xst + ((j + 1) & 1) * S2D_STAGE);                       // the next iteration's window (past the end: the last tile again)
#endif
            __builtin_amdgcn_s_waitcnt(0x0F70);                               // vmcnt(0): my part of it has landed
            __syncthreads();
        }
        return;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                                       // vmcnt(0): my part of the weights
    __syncthreads();
    int done_k = -1;                                                          // tile whose accumulators wait for their epilogue
    for (int j = 0; j < total; ++j) {
        unsigned char* cur = xst + (j & 1) * S2D_STAGE;
        // the epilogue of the previous tile at the TOP of the iteration: its stores have a whole iteration to drain before the barrier
#if defined(S2D_ABLATE) && S2D_ABLATE == 3                                // diagnostic builds: 1 no MFMAs, 2 no DMA after the prologue, 3 no epilogue
        if (done_k >= 0) { for (int p = 0; p < 2; ++p) { asm volatile("" :: "v"(acc[p])); for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f; } done_k = -1; }
#else
        if (done_k >= 0) { epilogue(done_k); done_k = -1; }
#endif
#if !(defined(S2D_ABLATE) && S2D_ABLATE == 1)
        mfma_group(cur, cur_g);
#endif
        if (cur_g == n_groups - 1) { done_k = cur_k; cur_g = 0; ++cur_k; } else ++cur_g;
        __syncthreads();
    }
    if (done_k >= 0) epilogue(done_k);
}

FLDR_TU_STATUS(s2)

// ------------------------------------------------------------------------------------------------
// prepack: max|w| -> power-of-two scale -> hi/lo halves in A-operand order [chunk][step][m][kind][lane][8 halves]
// ------------------------------------------------------------------------------------------------
static inline void s2_geometry(int cout, int& mt, int& nmt) {
    if (cout <= 16) { mt = 16; nmt = 1; }
    else if (cout <= 32) { mt = 32; nmt = 1; }
    else { mt = 32; nmt = 2; }
}

__global__ void s2_absmax_kernel(const float* __restrict__ w, int64_t n, float* __restrict__ hdr) {
    __shared__ float red[256];
    float m = 0.0f;
    for (int64_t i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(w[i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x < S2_HDR) {
        const float mx = red[0];
        float scale = 1.0f;
        if (mx > 0.0f) scale = exp2f(floorf(log2f(8192.0f / mx)));
        const float v[S2_HDR] = {1.0f / scale, scale, mx, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        hdr[threadIdx.x] = v[threadIdx.x];
    }
}

__global__ void s2_prepack_kernel(const float* __restrict__ w, float* __restrict__ wp, int cout, int cin, int mt, int nmt,
                                  int64_t total_h8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one 8-half element per thread
    if (i >= total_h8) return;
    const float scale = wp[1];
    const int kl = 64 / mt, cps = kl / 2, steps = S2_CC / cps;
    const int lane = (int)(i % 64);
    const int kind = (int)((i / 64) % 2);
    const int m = (int)((i / 128) % nmt);
    const int s = (int)((i / (128 * nmt)) % steps);
    const int chunk = (int)(i / ((int64_t)128 * nmt * steps));
    const int li = lane % mt, lg = lane / mt;
    const int co = m * mt + li;
    const int c = chunk * S2_CC + s * cps + (cps == 2 ? (lg >> 1) : 0);
    const int rp = cps == 2 ? (lg & 1) : lg;
    s2_h8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int dy, dx;
        s2_tap(rp, j, dy, dx);
        const float x = (co < cout && c < cin) ? w[(((int64_t)co * cin + c) * 4 + dy) * 4 + dx] * scale : 0.0f;
        const _Float16 h = (_Float16)x;
        v[j] = kind == 0 ? h : (_Float16)(x - (float)h);
    }
    reinterpret_cast<s2_h8*>(wp + S2_HDR)[i] = v;
}

// Section D of the pack (conv4x4s2_dma_spk_kernel: 17..32 output channels, cin a multiple of 8, <= 64): [group of 8 channels][tap pair
// tp = dy * 2 + dx / 2][hi, lo][lane = tap-of-pair * 32 + output channel][8 channels] halves, cin * 512 floats behind the first section.
static inline bool s2_has_dma_section(int cout, int cin) { return cout > 16 && cout <= 32 && cin % 8 == 0 && cin <= 64; }
__global__ void s2_prepack_dma_kernel(const float* __restrict__ w, float* __restrict__ wp, float* __restrict__ dst, int cout, int cin, int64_t total_h8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one 8-half element per thread
    if (i >= total_h8) return;
    const float scale = wp[1];
    const int lane = (int)(i % 64), kind = (int)((i / 64) % 2), tp = (int)((i / 128) % 8), grp = (int)(i / 1024);
    const int co = lane & 31, dy = tp >> 1, dx = 2 * (tp & 1) + (lane >> 5);
    s2_h8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = grp * 8 + j;
        const float x = co < cout ? w[(((int64_t)co * cin + c) * 4 + dy) * 4 + dx] * scale : 0.0f;
        const _Float16 h = (_Float16)x;
        v[j] = kind == 0 ? h : (_Float16)(x - (float)h);
    }
    reinterpret_cast<s2_h8*>(dst)[i] = v;
}

static inline int64_t s2_first_section_floats(int cout, int cin) {
    int mt, nmt;
    s2_geometry(cout, mt, nmt);
    const int n_chunks = (cin + S2_CC - 1) / S2_CC, steps = S2_CC / ((64 / mt) / 2);
    return (int64_t)n_chunks * steps * nmt * 2 * 64 * 4;
}
extern "C" int64_t fldr_conv_s2_prepack_size(int cout, int cin) {
    if (cout <= 0 || cin <= 0 || cout > 64 || cin > 112) return FLDR_E_ARG;
    return S2_HDR + s2_first_section_floats(cout, cin) + (s2_has_dma_section(cout, cin) ? (int64_t)cin * 512 : 0);        // floats
}

extern "C" int fldr_conv_s2_prepack(const float* weight, float* wpack, int cout, int cin, fldr_stream_t stream) {
    FLDR_CHECK_ARG(weight && wpack);
    const int64_t total = fldr_conv_s2_prepack_size(cout, cin);
    if (total < 0) return (int)total;
    int mt, nmt;
    s2_geometry(cout, mt, nmt);
    const int64_t first = s2_first_section_floats(cout, cin), total_h8 = first / 4;
    hipLaunchKernelGGL(s2_absmax_kernel, dim3(1), dim3(256), 0, fldr_s(stream), weight, (int64_t)cout * cin * 16, wpack);
    hipLaunchKernelGGL(s2_prepack_kernel, dim3(fldr_cdiv(total_h8, 256)), dim3(256), 0, fldr_s(stream), weight, wpack, cout, cin, mt, nmt,
                       total_h8);
    if (s2_has_dma_section(cout, cin)) {
        static_assert(S2_CC * 128 == 512, "first section: cin * 512 floats for 17..32 output channels");
        const int64_t d_h8 = (int64_t)cin * 128;
        hipLaunchKernelGGL(s2_prepack_dma_kernel, dim3(fldr_cdiv(d_h8, 256)), dim3(256), 0, fldr_s(stream), weight, wpack, wpack + S2_HDR + first, cout, cin, d_h8);
    }
    FLDR_LAUNCH_RET();
}

template <int MT, int NMT, int PT>
static int s2_launch(S2Args& a, int N, hipStream_t s) {
    using Cfg = S2Cfg<MT, NMT>;
    static std::atomic<uint64_t> attr_done{0};
    if (int e = fldr_set_max_lds(reinterpret_cast<const void*>(&conv4x4s2_split_kernel<MT, NMT, PT>), Cfg::LDS_BYTES, attr_done)) return e;
    a.tiles_x = fldr_cdiv(a.Wout, S2_TW);
    const int tiles_y = fldr_cdiv(a.Hout, S2_TH);
    a.n_tiles = a.tiles_x * tiles_y;
    a.tiles_per_xcd = (a.n_tiles + 7) / 8;
    hipLaunchKernelGGL((conv4x4s2_split_kernel<MT, NMT, PT>), dim3(8 * a.tiles_per_xcd, N), dim3(256), Cfg::LDS_BYTES, s, a);
    FLDR_LAUNCH_RET();
}

// Persistent kernel: 2 workgroups per CU (LDS: all weights + two input stages), XCD-contiguous tile ranges.
static int g_s2_xshift = -1;                     // -1: automatic (15 on wide images); >= 0: forced (0 .. 31)
FLDR_HOOK int fldr_debug_s2_xshift(int v) { if (v >= -1 && v < S2_TW) g_s2_xshift = v; return g_s2_xshift; }
static int g_s2_persistent = 1;
FLDR_HOOK int fldr_debug_s2_persistent(int v) { if (v >= 0) g_s2_persistent = v; return g_s2_persistent; }

template <int MT, int NMT, int PT, bool V4>
static int s2_launch_pers2(S2Args& a, hipStream_t s, int lds_bytes) {
    static std::atomic<int> attr_bytes[64];                      // per device ordinal: the attribute is per device
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return (int)e;
    if (attr_bytes[dev & 63].load(std::memory_order_acquire) < lds_bytes) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv4x4s2_pers_kernel<MT, NMT, PT, V4>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return (int)e;
        attr_bytes[dev & 63].store(lds_bytes, std::memory_order_release);
    }
    hipLaunchKernelGGL((conv4x4s2_pers_kernel<MT, NMT, PT, V4>), dim3(8 * a.wgs_per_xcd), dim3(256), lds_bytes, s, a);
    FLDR_LAUNCH_RET();
}

static int g_s2_vec4 = 1;                        // 16-byte staging loads where the geometry allows (0: always the 4-byte path)
FLDR_HOOK int fldr_debug_s2_vec4(int v) { if (v == 0 || v == 1) g_s2_vec4 = v; return g_s2_vec4; }

template <int MT, int NMT, int PT>
static int s2_launch_pers(S2Args& a, int N, hipStream_t s, int lds_bytes) {
    // Tile grid shifted left by 15 output columns on wide images: a tile's 66 input columns then start ONE float into a
    // 128-byte line ([64 t - 31, 64 t + 34]) and touch 3 lines per row instead of the 4 of the unshifted span
    // [64 t - 1, 64 t + 64] (one float each into the lines left and right).  PMC at 4K, enc1: the L2 fetched 1,629 MB for
    // 920 MB of planes and the kernel ran at the fabric's ~6.4 TB/s; one extra, partly filled tile column pays for it.
    a.x_shift = g_s2_xshift >= 0 ? g_s2_xshift : (a.Wout >= 256 ? 15 : 0);
    a.tiles_x = fldr_cdiv(a.Wout + a.x_shift, S2_TW);
    a.n_tiles = a.tiles_x * fldr_cdiv(a.Hout, S2_TH);
    a.N = N;
    const int64_t total = (int64_t)N * a.n_tiles;
    if (total >= (1ll << 30)) return FLDR_E_SHAPE;
    a.tiles_per_xcd = (int)((total + 7) / 8);
    a.wgs_per_xcd = a.tiles_per_xcd < 64 ? a.tiles_per_xcd : 64;
    // 16-byte staging: the window start ix0 - 1 = 64 t - 2 x_shift - 2 is a multiple of 4 floats for odd shifts; rows and
    // planes must keep that alignment
    bool v4 = g_s2_vec4 && (a.x_shift & 1) && (a.Win & 3) == 0;
    for (int k = 0; k < a.n_src && v4; ++k)
        v4 = (reinterpret_cast<uintptr_t>(a.src[k]) & 15) == 0 && (a.src_bstride[k] & 3) == 0 && (a.src_cstride[k] & 3) == 0;
    return v4 ? s2_launch_pers2<MT, NMT, PT, true>(a, s, lds_bytes) : s2_launch_pers2<MT, NMT, PT, false>(a, s, lds_bytes);
}

// Same descriptor as fldr_conv2d (ksize 4, stride 2; no up2 sources, no residual); d->wpack from fldr_conv_s2_prepack.
extern "C" int fldr_conv2d_s2_split(const fldr_conv_desc* d, fldr_stream_t stream) {
    FLDR_CHECK_ARG(d && d->wpack && (d->out || d->out_spk) && d->n_src >= 1 && d->n_src <= FLDR_CONV_MAX_SRC);
    FLDR_CHECK_ARG(d->N > 0 && d->cin > 0 && d->cin <= 112 && d->cout > 0 && d->cout <= 64 && !d->residual);
    FLDR_CHECK_ARG(d->cout_store > 0 && d->cout_store <= d->cout && d->ksize == 4 && d->stride == 2);
    if (d->Hout != (d->Hin + 2 - 4) / 2 + 1 || d->Wout != (d->Win + 2 - 4) / 2 + 1) return FLDR_E_SHAPE;
    S2Args a;
    int csum = 0;
    for (int s = 0; s < FLDR_CONV_MAX_SRC; ++s) {
        const bool live = s < d->n_src;
        if (live) { FLDR_CHECK_ARG(d->src[s] && d->src_c[s] > 0 && !d->src_up2[s]); }
        a.src[s] = live ? d->src[s] : nullptr;
        a.src_bstride[s] = live ? d->src_bstride[s] : 0;
        a.src_cstride[s] = live ? (d->src_cstride[s] ? d->src_cstride[s] : (int64_t)d->Hin * d->Win) : 0;
        a.src_cbegin[s] = csum;
        if (live) csum += d->src_c[s];
    }
    a.src_cbegin[FLDR_CONV_MAX_SRC] = csum;
    if (csum != d->cin) return FLDR_E_SHAPE;
    a.n_src = d->n_src;
    a.wpack = d->wpack; a.bias = d->bias; a.out = d->out; a.out_spk = reinterpret_cast<unsigned char*>(d->out_spk);
    a.cin = d->cin; a.cout = d->cout; a.cout_store = d->cout_store;
    a.Hin = d->Hin; a.Win = d->Win; a.Hout = d->Hout; a.Wout = d->Wout; a.relu = d->relu; a.tiles_x = 0;
    int mt, nmt;
    s2_geometry(d->cout, mt, nmt);
    hipStream_t s = fldr_s(stream);
    if ((int64_t)d->Hin * d->Win * 4 >= (1ll << 32)) return FLDR_E_SHAPE;
    // persistent kernel when every chunk's weights fit in LDS next to the input stages with two workgroups per CU
    const int n_chunks = (d->cin + S2_CC - 1) / S2_CC;
    if (g_s2_persistent && d->cin <= 64) {
        if (mt == 16) { const int lds = n_chunks * S2Cfg<16, 1>::W_BYTES + 2 * S2Cfg<16, 1>::X_DW * 4; if (lds <= 80 * 1024) return s2_launch_pers<16, 1, 4>(a, d->N, s, lds); }
        else if (nmt == 1) { const int lds = n_chunks * S2Cfg<32, 1>::W_BYTES + 2 * S2Cfg<32, 1>::X_DW * 4; if (lds <= 80 * 1024) return s2_launch_pers<32, 1, 2>(a, d->N, s, lds); }
    }
    if (mt == 16) return s2_launch<16, 1, 4>(a, d->N, s);
    if (nmt == 1) return s2_launch<32, 1, 2>(a, d->N, s);
    return s2_launch<32, 2, 2>(a, d->N, s);
}

// The stride-2 4x4 convolution on a split-packed source (fldr_conv2d_s2_split's arithmetic and outputs; d->src[0] = the packed
// tensor, d->src_c[0] = cin with cin % 8 == 0, d->src_bstride[0] in BYTES (0 for N = 1); d->wpack from fldr_conv_s2_prepack).
// Persistent kernel only: cin <= 64 and every chunk's weights in LDS; FLDR_E_SHAPE otherwise.
template <int MT, int NMT, int PT>
static int s2_launch_pers_spk(S2Args& a, int N, hipStream_t s, int lds_bytes, int pair = 1) {
    static std::atomic<uint64_t> attr_done{0};
    if (int e = fldr_set_max_lds(reinterpret_cast<const void*>(&conv4x4s2_pers_spk_kernel<MT, NMT, PT>), lds_bytes, attr_done)) return e;
    a.x_shift = 0;                                                            // (pixels are 16-byte records: every window start is aligned)
    a.tiles_x = fldr_cdiv(a.Wout, S2_TW);
    a.n_tiles = a.tiles_x * fldr_cdiv(a.Hout, S2_TH);
    a.N = N;
    const int64_t total = (int64_t)N * a.n_tiles;
    if (total >= (1ll << 30)) return FLDR_E_SHAPE;
    a.tiles_per_xcd = (int)((total + 7) / 8);
    const int cap = (lds_bytes > 80 * 1024 ? 32 : 64) / pair;                // workgroups per XCD that fit: one or two per CU (a pair launch: shared by the two problems)
    a.wgs_per_xcd = a.tiles_per_xcd < cap ? a.tiles_per_xcd : cap;
    hipLaunchKernelGGL((conv4x4s2_pers_spk_kernel<MT, NMT, PT>), dim3(8 * a.wgs_per_xcd, pair), dim3(256), lds_bytes, s, a);
    FLDR_LAUNCH_RET();
}

// The LDS-DMA kernel (17..32 output channels): all groups' weights + two window stages, one workgroup per CU.
#ifndef S2_DMA_DEFAULT
#define S2_DMA_DEFAULT 1
#endif
static int g_s2_dma = S2_DMA_DEFAULT;
FLDR_HOOK int fldr_debug_s2_dma(int v) { if (v == 0 || v == 1) g_s2_dma = v; return g_s2_dma; }
static int s2_launch_dma_spk(S2Args& a, int N, hipStream_t s, int lds_bytes, int pair) {
    static std::atomic<uint64_t> attr_done{0};
    if (int e = fldr_set_max_lds(reinterpret_cast<const void*>(&conv4x4s2_dma_spk_kernel), lds_bytes, attr_done)) return e;
    a.x_shift = 0;
    a.tiles_x = fldr_cdiv(a.Wout, S2_TW);
    a.n_tiles = a.tiles_x * fldr_cdiv(a.Hout, S2_TH);
    a.N = N;
    const int64_t total = (int64_t)N * a.n_tiles;
    if (total >= (1ll << 30)) return FLDR_E_SHAPE;
    a.tiles_per_xcd = (int)((total + 7) / 8);
    const int cap = 32 / pair;                                               // one workgroup per CU (a pair launch: shared by the two problems)
    a.wgs_per_xcd = a.tiles_per_xcd < cap ? a.tiles_per_xcd : cap;
    hipLaunchKernelGGL(conv4x4s2_dma_spk_kernel, dim3(8 * a.wgs_per_xcd, pair), dim3((4 + S2D_NLOAD) * 64), lds_bytes, s, a);
    FLDR_LAUNCH_RET();
}

static int s2_spk_run(const fldr_conv_desc* d, const fldr_conv_desc* d2, fldr_stream_t stream) {
    FLDR_CHECK_ARG(d && d->wpack && (d->out || d->out_spk) && d->n_src == 1 && d->src[0] && !d->src_up2[0] && !d->residual);
    FLDR_CHECK_ARG(d->N > 0 && d->cin > 0 && d->cin <= 64 && (d->cin & 7) == 0 && d->src_c[0] == d->cin && d->cout > 0 && d->cout <= 64);
    FLDR_CHECK_ARG(d->cout_store > 0 && d->cout_store <= d->cout && d->ksize == 4 && d->stride == 2);
    if (d->Hout != (d->Hin + 2 - 4) / 2 + 1 || d->Wout != (d->Win + 2 - 4) / 2 + 1) return FLDR_E_SHAPE;
    if ((int64_t)d->Hin * d->Win * 16 >= (1ll << 32)) return FLDR_E_SHAPE;
    S2Args a;
    for (int s = 0; s < FLDR_CONV_MAX_SRC; ++s) { a.src[s] = nullptr; a.src_bstride[s] = 0; a.src_cstride[s] = 0; a.src_cbegin[s] = 0; }
    a.src[0] = d->src[0]; a.src_bstride[0] = d->src_bstride[0]; a.src_cbegin[FLDR_CONV_MAX_SRC] = d->cin; a.n_src = 1;
    a.wpack = d->wpack; a.bias = d->bias; a.out = d->out; a.out_spk = reinterpret_cast<unsigned char*>(d->out_spk);
    a.wpack2 = nullptr; a.bias2 = nullptr; a.out2 = nullptr; a.out_spk2 = nullptr;
    int pair = 1;
    if (d2) {                                                            // the same convolution geometry on the same source, other weights / outputs
        FLDR_CHECK_ARG(d2->wpack && (d2->out || d2->out_spk) && d2->n_src == 1 && d2->src[0] == d->src[0] && d2->src_bstride[0] == d->src_bstride[0]);
        FLDR_CHECK_ARG(d2->N == d->N && d2->cin == d->cin && d2->src_c[0] == d->cin && d2->cout == d->cout && d2->cout_store == d->cout_store && !d2->residual);
        FLDR_CHECK_ARG(d2->Hin == d->Hin && d2->Win == d->Win && d2->Hout == d->Hout && d2->Wout == d->Wout && d2->relu == d->relu && d2->ksize == 4 && d2->stride == 2);
        FLDR_CHECK_ARG(!d2->bias == !d->bias && !d2->out == !d->out && !d2->out_spk == !d->out_spk);
        a.wpack2 = d2->wpack; a.bias2 = d2->bias; a.out2 = d2->out; a.out_spk2 = reinterpret_cast<unsigned char*>(d2->out_spk);
        pair = 2;
    }
    a.cin = d->cin; a.cout = d->cout; a.cout_store = d->cout_store;
    a.Hin = d->Hin; a.Win = d->Win; a.Hout = d->Hout; a.Wout = d->Wout; a.relu = d->relu; a.tiles_x = 0;
    int mt, nmt;
    s2_geometry(d->cout, mt, nmt);
    const int n_chunks = d->cin / S2_CC;
    hipStream_t s = fldr_s(stream);
    if (mt == 16) { const int lds = n_chunks * S2Cfg<16, 1>::W_BYTES + 2 * S2Cfg<16, 1>::X_DW * 4; if (lds <= 80 * 1024) return s2_launch_pers_spk<16, 1, 4>(a, d->N, s, lds, pair); }
    else if (nmt == 1) {
        const int lds_dma = (d->cin / 8) * 16384 + 2 * S2D_STAGE;
        if (g_s2_dma && s2_has_dma_section(d->cout, d->cin) && lds_dma <= 160 * 1024) return s2_launch_dma_spk(a, d->N, s, lds_dma, pair);
        const int lds = n_chunks * S2Cfg<32, 1>::W_BYTES + 2 * S2Cfg<32, 1>::X_DW * 4;
        if (lds <= 156 * 1024) return s2_launch_pers_spk<32, 1, 2>(a, d->N, s, lds, pair);   // (> 80 KB: one workgroup per CU)
    }
    return FLDR_E_SHAPE;
}

extern "C" int fldr_conv2d_s2_spk(const fldr_conv_desc* d, fldr_stream_t stream) { return s2_spk_run(d, nullptr, stream); }

// Two stride-2 convolutions of the SAME packed source with the same geometry in ONE launch (gridDim.y = 2): the two 32-channel halves of
// enc3 (32 -> 64; one half's 64 KB of weights per workgroup in LDS, fLDRnet.py:617) — the bits of two fldr_conv2d_s2_spk calls.
extern "C" int fldr_conv2d_s2_spk_pair(const fldr_conv_desc* d0, const fldr_conv_desc* d1, fldr_stream_t stream) {
    FLDR_CHECK_ARG(d0 && d1);
    return s2_spk_run(d0, d1, stream);
}
