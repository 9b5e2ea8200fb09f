// 3x3 / stride 1 / pad 1 convolutions on SPLIT-PACKED activations: persistent, LDS-DMA fed, 3 x fp16-split MFMA.
//
// Arithmetic: exactly that of conv_split_kernels.hip (x = hi + lo in fp16, hi*hi + hi*lo + lo*hi on
// v_mfma_f32_16x16x32_f16 with fp32 accumulation, same tap/channel order), so results are bit-identical to that
// kernel.  What changes is where the hi/lo split happens and how operands travel:
//
//   * Activations between convolutions live in HBM in the "split-packed" (SPK) layout
//         [N][G = ceil(C/8)][kind: hi, lo][H*W][8 halves]            (16 B per pixel, group and kind)
//     which costs the same 4 B per element as fp32 but is already the MFMA B-operand image: the PRODUCER's epilogue
//     splits each value once (fldr_spk_pack does it for fp32 NCHW tensors that come from other kernels), and the
//     consumer's staging is pure LDS-DMA (global_load_lds_dwordx4, one pixel x 8 channels per lane) — no VGPRs, no
//     VALU, no per-chunk vmcnt(0).  Measured on the previous kernel (tools/stamps, 96->96 @288x480): of 27.5 us per
//     workgroup, 7.2 us were exposed register staging + split arithmetic, ~10 us prologue/epilogue/launch latency
//     and 7.2 us the MFMA loop itself.
//   * Workgroups are PERSISTENT (one per CU; 156 KB of LDS): each walks a list of (sample, tile, output-channel
//     group) units and the (unit, chunk) iterations form ONE software pipeline — the DMA of iteration g+2 is issued
//     at the top of iteration g whatever unit it belongs to, so the fetch latency of a unit's first chunk and the
//     store tail of the previous unit hide under MFMAs instead of being paid per workgroup.
//   * Per iteration (16 input channels): 3-stage LDS ring of {weights in A-operand order (hi, lo), input tile
//     10 x 34 px as 4 planes [hi g0, hi g1, lo g0, lo g1] of 16-B elements}; the only wait is a counted
//     s_waitcnt vmcnt(K) (K = the DMA instructions this wave issued for iteration g+2) in front of one raw s_barrier.
//
// Zero padding: out-of-image pixels and padding channel groups read a 16-B block of zeros in the weight header.
#include "common.h"
#include <hip/hip_fp16.h>

#ifdef SPK_STAMPS
// Diagnostic build only (tools/stamps): per-phase s_memtime sums of waves 0 and 4 of two workgroups, written to a
// buffer no kernel reads.
__device__ unsigned long long fldr_spk_stamp_buf[4 * 8];
#define KSTAMP(var) unsigned long long var; { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
FLDR_HOOK int fldr_debug_read_spk_stamps(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fldr_spk_stamp_buf), sizeof(unsigned long long) * 32);
}
#else
#define KSTAMP(var)
#endif

#include "spk_common.h"
#include <mutex>

#ifdef FLDR_TEST_HOOKS        // the barrier pipeline: round 1's kernel, kept as the bit-exact cross-check of the ring pipeline (test build only)
template <int NMT, int TERMS, bool HAS_RES>
__global__ __launch_bounds__(512, 2) void conv3x3_spk_kernel(SpkArgs a) {
    using Cfg = SpkCfg<NMT>;
    constexpr int MTOT = 16 * NMT;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lg = lane >> 4;
    const int n_chunks = a.n_chunks;

    // Units of this workgroup.  Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2): XCD x owns
    // the contiguous unit range [x*upx, (x+1)*upx) (unit = (sample, tile, group), group fastest), so the output
    // groups of a tile and vertically adjacent tiles meet in one L2.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int u_end = min((xcd + 1) * a.units_per_xcd, a.n_units);
    const int u_first = xcd * a.units_per_xcd + slot;
    if (u_first >= u_end) return;                                        // workgroup-uniform
    const int my_units = (u_end - u_first + a.wgs_per_xcd - 1) / a.wgs_per_xcd;
    const int total = my_units * n_chunks;

    // The two waves of a SIMD are skewed: waves 4-7 ("early", MFMA priority) enter an iteration's MFMA steps right
    // after the barrier and do their bookkeeping (epilogue stores, address generation of the next DMA) after their
    // last MFMA, while they would otherwise idle at the barrier; waves 0-3 ("late") do theirs first, under the early
    // wave's MFMAs.  Measured before the skew: ~1,200 cycles per iteration in which neither wave issued an MFMA.
    const bool early = wave >= 4;                                        // wave-uniform
    if (early) __builtin_amdgcn_s_setprio(1);

    // ---- issue side ----
    // Every wave issues the same K_DMA = NWI + 3 LDS-DMA instructions per iteration, all lanes active, no branches
    // (they are scheduled BETWEEN the MFMAs of the iteration, see the pipeline): NWI weight pieces (1 KB each; the
    // waves that would run past the slab re-fetch its last pieces instead) and three 64-pixel pieces of one of the 4
    // input planes (pieces [0,64) [64,128) [128,192) or [192,256) [256,320) [288,352): the overlap of the last two
    // writes the same bytes twice instead of masking lanes).
    constexpr int K_DMA = Cfg::NWI + 3;
    const int ip = wave >> 1;                                            // LDS plane: kind = ip >> 1, group of the chunk = ip & 1
    const int ikind = ip >> 1, igrp = ip & 1;
    const char* zero_blk = reinterpret_cast<const char*>(a.wpack + 4);
    int w_piece[Cfg::NWI], x_piece[3];
    uint32_t w_voff[Cfg::NWI];
    // Weight source: the pack holds [pack group][chunk][step][pack_nmt blocks][kind] 1-KB blocks; this workgroup's
    // output group (constant over its units, see cbase below) is blocks [msel, msel + NMT) of pack group pgrp.
    const int grp0 = u_first % a.groups, sub = a.pack_nmt / NMT;
    const int pgrp = grp0 / sub, msel = (grp0 - pgrp * sub) * NMT;
    const int pack_w_bytes = SPK_STEPS * a.pack_nmt * 2 * 1024;
#pragma unroll
    for (int i = 0; i < Cfg::NWI; ++i) {
        w_piece[i] = min(i * 512 + wave * 64, Cfg::PIECES - 64);
        const int b = w_piece[i] >> 6, step = b / (2 * NMT), mk = b - step * 2 * NMT;          // LDS block (step, m, kind)
        w_voff[i] = (uint32_t)((step * a.pack_nmt + msel) * 2 + mk) * 1024u + (uint32_t)lane * 16u;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) x_piece[i] = min(((wave & 1) * 3 + i) * 64, SPK_PLANE / 16 - 64);
    // The input-group table lives in VGPR lanes (lane l = group l, already advanced to this wave's hi or lo plane):
    // one v_readlane per iteration instead of dependent scalar loads from the kernel arguments (measured: the scalar
    // loads + waits cost ~1,000 cycles per iteration in which neither wave of a SIMD issued MFMAs).
    unsigned long long tab_ptr;
    long long tab_bs;
    {
        const auto* kt = (const __attribute__((address_space(4))) unsigned long long*)__builtin_amdgcn_kernarg_segment_ptr();
        const int l = lane < SPK_MAX_GROUPS ? lane : 0;
        unsigned long long e = kt[l];
        tab_bs = (long long)kt[SPK_MAX_GROUPS + l];
        const bool up2 = (e & 1ull) != 0ull;
        const long long plane = up2 ? (long long)(a.H >> 1) * (a.W >> 1) * 16 : (long long)a.H * a.W * 16;
        if (e != 0ull && ikind) e += (unsigned long long)plane;
        tab_ptr = e;
    }
    int iss_u = u_first, iss_c = 0;
    uint32_t g_full[3], g_half[3];                                       // byte offsets in a plane; ~0u = outside the image
    int iss_n = 0;
    const char* const iss_w = reinterpret_cast<const char*>(a.wpack + SPK_HDR) + (int64_t)pgrp * n_chunks * pack_w_bytes;
    auto issue_geometry = [&]() {
        const int t = spk_div(iss_u, a.m_groups, a.groups);
        iss_n = spk_div(t, a.m_tiles, a.n_tiles);
        const int tile = t - iss_n * a.n_tiles;
        const int ty = spk_div(tile, a.m_tiles_x, a.tiles_x);
        const int oy0 = ty * SPK_TH, ox0 = (tile - ty * a.tiles_x) * SPK_TW;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int e = x_piece[i] + lane;
            const int y = e / SPK_IW, x = e % SPK_IW;
            const int gy = oy0 - 1 + y, gx = ox0 - 1 + x;
            const bool ok = e < SPK_IH * SPK_IW && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            g_full[i] = ok ? (uint32_t)(gy * a.W + gx) * 16u : ~0u;
            g_half[i] = ok ? (uint32_t)((gy >> 1) * (a.W >> 1) + (gx >> 1)) * 16u : ~0u;
        }
    };
    const char* wbase = nullptr;                                         // workgroup-uniform source of the next weight slab
    const char* dptr[3];                                                 // per-lane source addresses of the next input pieces
    auto issue_prepare = [&]() {
        // Past the workgroup's last iteration the same instructions run against dummy sources (weight slab 0, the zero
        // block) into a ring stage nobody reads any more, so that the counted vmcnt wait holds in every iteration.
        const bool live = iss_u < u_end;                                 // workgroup-uniform
#if defined(SPK_ABLATE) && (SPK_ABLATE == 4 || SPK_ABLATE == 5)
        wbase = reinterpret_cast<const char*>(a.wpack + SPK_HDR);        // diagnostic: always the same (cached) slab
#else
        wbase = live ? iss_w + (int64_t)iss_c * pack_w_bytes : reinterpret_cast<const char*>(a.wpack + SPK_HDR);
#endif
        const int gi = live ? iss_c * 2 + igrp : 0;
        const uint32_t e_lo = __builtin_amdgcn_readlane((int)(uint32_t)tab_ptr, gi), e_hi = __builtin_amdgcn_readlane((int)(uint32_t)(tab_ptr >> 32), gi);
        const uint32_t b_lo = __builtin_amdgcn_readlane((int)(uint32_t)tab_bs, gi), b_hi = __builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)tab_bs >> 32), gi);
        const unsigned long long e = ((unsigned long long)e_hi << 32) | e_lo;
        const long long bs = (long long)(((unsigned long long)b_hi << 32) | b_lo);
        const bool nul = e == 0ull || !live, up2 = (e & 1ull) != 0ull;
        const char* base = reinterpret_cast<const char*>(static_cast<uintptr_t>(e & ~1ull)) + (int64_t)iss_n * bs;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#if defined(SPK_ABLATE) && (SPK_ABLATE == 3 || SPK_ABLATE == 5)
            dptr[i] = zero_blk;
#else
            const uint32_t off = up2 ? g_half[i] : g_full[i];
            dptr[i] = (off != ~0u && !nul) ? base + off : zero_blk;
#endif
        }
        if (live && ++iss_c == n_chunks) { iss_c = 0; iss_u += a.wgs_per_xcd; if (iss_u < u_end) issue_geometry(); }
    };
    auto issue_fire = [&](int j, unsigned char* stage) {                 // j compile-time after unrolling
        if (j < Cfg::NWI) {
            const int jj = j < Cfg::NWI ? j : 0;
            __builtin_amdgcn_global_load_lds((kgptr_t)(wbase + w_voff[jj]), (klptr_t)(stage + w_piece[jj] * 16), 16, 0, 0);
        } else {
            const int jj = j >= Cfg::NWI ? j - Cfg::NWI : 0;
            __builtin_amdgcn_global_load_lds((kgptr_t)dptr[jj], (klptr_t)(stage + Cfg::W_BYTES + ip * SPK_PLANE + x_piece[jj] * 16), 16, 0, 0);
        }
    };

    // ---- compute side ----
    int boff[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) boff[p] = Cfg::W_BYTES + (lg & 1) * SPK_PLANE + (wave * SPK_IW + p * 16 + lj) * 16;
    const int tap_sel = lg >> 1;
    f4 acc[NMT][2];
#pragma unroll
    for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int p = 0; p < 2; ++p) acc[m][p] = f4{0.0f, 0.0f, 0.0f, 0.0f};
    int cur_u = u_first, cur_c = 0;
    const float inv_scale = a.wpack[0];
    const int64_t HW = (int64_t)a.H * a.W;
    const int gout = (a.cout_store + 7) >> 3;
    // The output-channel group is the same for every unit of a workgroup (the host keeps wgs_per_xcd a multiple of
    // `groups`), so the bias is fetched once.
    const int cbase = grp0 * MTOT;
    float bias_r[NMT][4];
#pragma unroll
    for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int co = cbase + m * 16 + lg * 4 + r;
            co = co < a.cout ? co : a.cout - 1;
            bias_r[m][r] = a.bias ? a.bias[co] : 0.0f;
        }

    // Epilogue in three parts so that no memory latency sits between the last MFMA of a unit and the barrier:
    //   residual_prefetch  at the top of the unit's last iteration, BEFORE that iteration's DMA is issued (loads retire
    //                      in order, so waiting for it never drains the DMA);
    //   finish             after the last MFMA: scale, bias, ReLU, residual, hi/lo split — registers only;
    //   store              after the barrier, at the top of the next iteration (the stores are then older than the
    //                      next DMA and have a whole iteration to retire before the next counted wait).
    float res_r[NMT][2][4];
    float ov[NMT][2][4];
    h4 ohi[NMT][2], olo[NMT][2];
    uint32_t st_po[2];                                  // pixel index of my two output pixels (~0u when out of the image)
    int st_n = 0;
    // byte offsets below fit 32 bits (host-checked: cout_store*H*W*4 and the packed sample size < 2^32)
    const uint32_t HW32 = (uint32_t)HW;
    auto unit_pixels = [&](int u, uint32_t (&po)[2], int& n) {
        const int t = spk_div(u, a.m_groups, a.groups);
        n = spk_div(t, a.m_tiles, a.n_tiles);
        const int tile = t - n * a.n_tiles;
        const int ty = spk_div(tile, a.m_tiles_x, a.tiles_x);
        const int oy = ty * SPK_TH + wave;
        const int ox0 = (tile - ty * a.tiles_x) * SPK_TW;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int ox = ox0 + p * 16 + lj;
            po[p] = (oy < a.H && ox < a.W) ? (uint32_t)(oy * a.W + ox) : ~0u;
        }
    };
    auto residual_prefetch = [&]() {
        uint32_t po[2]; int n;
        unit_pixels(cur_u, po, n);
        const float* resn = a.residual + (int64_t)n * a.cout_store * HW;
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int co = cbase + m * 16 + lg * 4 + r;
                    co = co < a.cout_store ? co : a.cout_store - 1;
                    res_r[m][p][r] = resn[(uint32_t)co * HW32 + (po[p] != ~0u ? po[p] : 0u)];
                }
    };
    bool range_bad = false;
    auto finish = [&]() {
        unit_pixels(cur_u, st_po, st_n);
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[m][p][r] * inv_scale + bias_r[m][r];
                    if (a.relu) v = fmaxf(v, 0.0f);
                    if constexpr (HAS_RES) v += res_r[m][p][r];
                    ov[m][p][r] = v;
                    acc[m][p][r] = 0.0f;
                    const float x = cbase + m * 16 + lg * 4 + r < a.cout_store ? v : 0.0f;
                    _Float16 h, l;
                    spk_split(x, h, l, range_bad);
                    ohi[m][p][r] = h; olo[m][p][r] = l;
                }
            }
    };
    auto store = [&]() {
        if (a.out_f32) {
            char* outn = reinterpret_cast<char*>(a.out_f32 + (int64_t)st_n * a.cout_store * HW);
            const bool quads = !(a.cout_store & 3);                      // whole quads of channels: one predicate per 4 stores
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int m = 0; m < NMT; ++m) {
                    const int co0 = cbase + m * 16 + lg * 4;
                    const uint32_t off = ((uint32_t)co0 * HW32 + st_po[p]) * 4u;
                    if (quads) {
                        if (co0 < a.cout_store && st_po[p] != ~0u) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) *reinterpret_cast<float*>(outn + (off + (uint32_t)r * HW32 * 4u)) = ov[m][p][r];
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (co0 + r < a.cout_store && st_po[p] != ~0u) *reinterpret_cast<float*>(outn + (off + (uint32_t)r * HW32 * 4u)) = ov[m][p][r];
                    }
                }
        }
        if (a.out_spk) {
            char* spkn = reinterpret_cast<char*>(a.out_spk) + (int64_t)st_n * a.out_spk_bstride;
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int m = 0; m < NMT; ++m) {
                    const int go = (cbase + m * 16 + lg * 4) >> 3;
                    if (go < gout && st_po[p] != ~0u) {
                        const uint32_t off = ((uint32_t)go * 2u * HW32 + st_po[p]) * 16u + (uint32_t)(lg & 1) * 8u;
                        *reinterpret_cast<h4*>(spkn + off) = ohi[m][p];
                        *reinterpret_cast<h4*>(spkn + (off + HW32 * 16u)) = olo[m][p];
                    }
                }
        }
    };

    // ---- pipeline ----
    issue_geometry();
#pragma unroll
    for (int st = 0; st < 2; ++st) {                   // (stage 1: a dummy zero-block fill when the workgroup has one iteration)
        issue_prepare();
#pragma unroll
        for (int j = 0; j < K_DMA; ++j) issue_fire(j, smem + st * Cfg::STAGE);
    }
    if (early) issue_prepare();                        // iteration 0 fires the DMA of iteration 2
    __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): both prologue stages have landed
    __builtin_amdgcn_s_barrier();

    int st_cur = 0, st_iss = 2;
    bool store_pending = false;
#ifdef SPK_STAMPS
    unsigned long long ks_top = 0, ks_steps = 0, ks_fin = 0, ks_wait = 0, ks_bar = 0;
    KSTAMP(k_begin)
#endif
    for (int g = 0; g < total; ++g) {
        KSTAMP(k0)
        const bool last = cur_c == n_chunks - 1;       // workgroup-uniform
        if (!early) {
#if !defined(SPK_ABLATE) || SPK_ABLATE != 7
            if (store_pending) { store(); store_pending = false; }
#endif
            issue_prepare();                           // addresses of iteration g+2's DMA (dummy fills past the end); the
                                                       // K_DMA instructions themselves go between the MFMAs below
        }
        if constexpr (HAS_RES) { if (last) residual_prefetch(); }
        unsigned char* sdst = smem + st_iss * Cfg::STAGE;
        const unsigned char* sb = smem + st_cur * Cfg::STAGE;
        const unsigned char* win = sb + lane * 16;
        h8 bh[2][2], bl[2][2], ah[2][NMT], al[2][NMT];
        auto ld = [&](int buf, int s) {
            // taps of step s: 2s and 2s+1 (tap 9 = the zero-weight pad tap: re-reads tap 8's pixels, finite values).
            // Issue order = consumption order of the term-major MFMA sequence (hi x hi, hi x lo, lo x hi), so the
            // counted lgkmcnt waits let the first MFMAs of a step start before its last operands have arrived.
            const int tA = 2 * s, tB = 2 * s + 1 < 9 ? 2 * s + 1 : 8;
            const int offA = ((tA / 3) * SPK_IW + tA % 3) * 16, offB = ((tB / 3) * SPK_IW + tB % 3) * 16;
            const int toff = tap_sel ? offB : offA;
#pragma unroll
            for (int p = 0; p < 2; ++p) bh[buf][p] = *reinterpret_cast<const h8*>(sb + boff[p] + toff);
#pragma unroll
            for (int m = 0; m < NMT; ++m) ah[buf][m] = *reinterpret_cast<const h8*>(win + ((s * NMT + m) * 2 + 0) * 1024);
            if constexpr (TERMS > 1) {
#pragma unroll
                for (int p = 0; p < 2; ++p) bl[buf][p] = *reinterpret_cast<const h8*>(sb + 2 * SPK_PLANE + boff[p] + toff);
#pragma unroll
                for (int m = 0; m < NMT; ++m) al[buf][m] = *reinterpret_cast<const h8*>(win + ((s * NMT + m) * 2 + 1) * 1024);
            }
        };
        constexpr int N_MFMA = 2 * TERMS * NMT, N_DS = TERMS > 1 ? 4 + 2 * NMT : 2 + NMT;
        constexpr int N_TAIL = N_MFMA >= 12 ? 4 : (N_MFMA >= 6 ? 2 : 0);
        KSTAMP(k1)
        ld(0, 0);
        __builtin_amdgcn_sched_barrier(0);                           // keep step 0's reads out of the interleave pattern below
#pragma unroll
        for (int s = 0; s < SPK_STEPS; ++s) {
            // this step's share of the DMA instructions: an LDS-DMA instruction holds its wave's issue for 60-180
            // cycles; one at a time between MFMAs, the other wave of the SIMD keeps the matrix pipe busy meanwhile
            const int j0 = (s * K_DMA + SPK_STEPS - 1) / SPK_STEPS, j1 = ((s + 1) * K_DMA + SPK_STEPS - 1) / SPK_STEPS;
#pragma unroll
            for (int j = 0; j < K_DMA; ++j)
                if (j >= j0 && j < j1) issue_fire(j, sdst);
#if defined(SPK_ABLATE) && SPK_ABLATE == 1
            if (s + 1 < SPK_STEPS) { for (int p = 0; p < 2; ++p) { bh[(s + 1) & 1][p] = bh[s & 1][p]; bl[(s + 1) & 1][p] = bl[s & 1][p]; }
                                    for (int m = 0; m < NMT; ++m) { ah[(s + 1) & 1][m] = ah[s & 1][m]; al[(s + 1) & 1][m] = al[s & 1][m]; } }
#else
            if (s + 1 < SPK_STEPS) ld((s + 1) & 1, s + 1);           // lands while this step's MFMAs run
#endif
#pragma unroll
            for (int term = 0; term < TERMS; ++term)
#pragma unroll
                for (int m = 0; m < NMT; ++m)
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const h8 av = term == 2 ? al[s & 1][m] : ah[s & 1][m];
                        const h8 bv = term == 1 ? bl[s & 1][p] : bh[s & 1][p];
#if defined(SPK_ABLATE) && SPK_ABLATE == 2
                        asm volatile("" :: "v"(av), "v"(bv));
#else
                        acc[m][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc[m][p], 0, 0, 0);
#endif
                    }
            // spread the next step's LDS reads evenly between this step's MFMAs
            spk_step_pattern<N_MFMA, N_DS, N_TAIL, K_DMA>(s);
        }
        KSTAMP(k2)
        if (last) {
            finish();
            store_pending = true;
            cur_c = 0; cur_u += a.wgs_per_xcd;
        } else {
            ++cur_c;
        }
        KSTAMP(k3)
        // Iteration g+1's DMA (issued one iteration ago) must have landed; this iteration's may stay in flight.  Loads
        // retire in order, so "at most K_MIN of my vector-memory operations outstanding" implies it.  An early wave
        // that has just stored a unit drains everything instead (it has ~1,500 idle cycles before the late waves arrive).
        if (early) {
#if !defined(SPK_ABLATE) || SPK_ABLATE != 7
            if (store_pending) { store(); store_pending = false; __builtin_amdgcn_s_waitcnt(0x0F70); }
#endif
            issue_prepare();
        }
        __builtin_amdgcn_s_waitcnt(0x0F70 | Cfg::K_MIN);
        KSTAMP(k4)
        __builtin_amdgcn_s_barrier();
        KSTAMP(k5)
#ifdef SPK_STAMPS
        ks_top += k1 - k0; ks_steps += k2 - k1; ks_fin += k3 - k2; ks_wait += k4 - k3; ks_bar += k5 - k4;
#endif
        st_cur = st_cur == 2 ? 0 : st_cur + 1;
        st_iss = st_iss == 2 ? 0 : st_iss + 1;
    }
    if (store_pending) store();
    if (a.out_spk) fldr_note_range(range_bad);
#ifdef SPK_STAMPS
    KSTAMP(k_end)
    if ((blockIdx.x == 0 || blockIdx.x == 101) && (wave == 0 || wave == 4) && lane == 0) {
        unsigned long long* o = fldr_spk_stamp_buf + ((blockIdx.x == 0 ? 0 : 2) + (wave == 0 ? 0 : 1)) * 8;
        o[0] = ks_top; o[1] = ks_steps; o[2] = ks_fin; o[3] = ks_wait; o[4] = ks_bar; o[5] = total; o[6] = k_end - k_begin;
    }
#endif
}
#endif  // FLDR_TEST_HOOKS

// ------------------------------------------------------------------------------------------------
// fp32 NCHW <-> SPK
// ------------------------------------------------------------------------------------------------
__global__ void spk_pack_kernel(const float* __restrict__ src, int64_t src_bstride, unsigned char* __restrict__ dst,
                                int64_t dst_bstride, int C, int64_t HW) {
    const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int g = blockIdx.y, n = blockIdx.z;
    if (pix >= HW) return;
    const float* s = src + (int64_t)n * src_bstride + (int64_t)g * 8 * HW + pix;
    float x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = g * 8 + k < C ? s[(int64_t)k * HW] : 0.0f;
    h8 hi, lo;
    bool bad = false;
    { _Float16 hs[8], ls[8]; fldr_split_hl_group(x, hs, ls, bad);
#pragma unroll
      for (int k = 0; k < 8; ++k) { hi[k] = hs[k]; lo[k] = ls[k]; } }
    fldr_note_range(bad);
    unsigned char* d = dst + (int64_t)n * dst_bstride + ((int64_t)g * 2 * HW + pix) * 16;
    *reinterpret_cast<h8*>(d) = hi;
    *reinterpret_cast<h8*>(d + HW * 16) = lo;
}

__global__ void spk_unpack_kernel(const unsigned char* __restrict__ src, int64_t src_bstride, float* __restrict__ dst,
                                  int C, int64_t HW) {
    const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int g = blockIdx.y, n = blockIdx.z;
    if (pix >= HW) return;
    const unsigned char* s = src + (int64_t)n * src_bstride + ((int64_t)g * 2 * HW + pix) * 16;
    const h8 hi = *reinterpret_cast<const h8*>(s), lo = *reinterpret_cast<const h8*>(s + HW * 16);
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (g * 8 + k < C) dst[((int64_t)n * C + g * 8 + k) * HW + pix] = (float)hi[k] + (float)lo[k];
}

FLDR_TU_STATUS(spk)

// ---- status block of the current device (common.h: fldr_status_block) -------------------------------------------------------
namespace {
struct DevStatus { fldr_status_block* host = nullptr; fldr_status_block* host_dev = nullptr; float* poison = nullptr; };
std::mutex g_status_mu;
DevStatus g_status[64];
// allocate + bind on first use per device; returns null on failure
DevStatus* status_of_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(g_status_mu);
    DevStatus& st = g_status[dev];
    if (st.host) return &st;
    fldr_status_block* h = nullptr;
    fldr_status_block* hd = nullptr;
    float* p = nullptr;
    if (hipHostMalloc(reinterpret_cast<void**>(&h), sizeof(fldr_status_block), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return nullptr;
    *h = fldr_status_block{};
    bool ok = hipHostGetDevicePointer(reinterpret_cast<void**>(&hd), h, 0) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&p), 256) == hipSuccess &&
              hipMemset(p, 0, 256) == hipSuccess;
    int (*binders[])(fldr_status_block*, float*) = {fldr_status_bind_spk, fldr_status_bind_ring, fldr_status_bind_conv, fldr_status_bind_s2, fldr_status_bind_split,
                                                    fldr_status_bind_warp, fldr_status_bind_gather, fldr_status_bind_acc64, fldr_status_bind_dec23};
    for (auto b : binders) ok = ok && b(hd, p) == 0;
    ok = ok && hipDeviceSynchronize() == hipSuccess;
    if (!ok) { if (p) (void)hipFree(p); (void)hipHostFree(h); return nullptr; }
    st.host_dev = hd; st.poison = p; st.host = h;
    return &st;
}
}  // namespace

const float* fldr_status_poison_ptr(void) {
    DevStatus* st = status_of_current_device();
    return st ? st->poison : nullptr;
}

// Host-visible status of the current device: *host_words points at two ints in pinned host memory that the library's kernels set to 1
// (system-scope stores) when [0] an activation was saturated by the fp16 split (fldr_range_status), [1] a bounded ring wait expired
// (fldr_ring_status) — readable at any time without synchronising; cleared by the reset forms of those two calls.  The first call on a
// device allocates the block and binds it into the library's kernels (synchronises; not during a stream capture).
extern "C" int fldr_status_word(const volatile int** host_words) {
    FLDR_CHECK_ARG(host_words);
    DevStatus* st = status_of_current_device();
    if (!st) return FLDR_E_STATUS;
    *host_words = &st->host->range;
    return 0;
}

// Sticky range status of the split-precision convolutions on the current device since the last reset: 1 = a value beyond +-65504
// (or a NaN) was split — and saturated (common.h: fldr_split_hl) —, 0 = clean, negative on a HIP error.  Synchronises.
extern "C" int fldr_range_status(int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    int v = 0;
    int (*readers[])(int) = {fldr_range_read_spk, fldr_range_read_ring, fldr_range_read_conv, fldr_range_read_s2, fldr_range_read_split, fldr_range_read_warp, fldr_range_read_gather, fldr_range_read_acc64, fldr_range_read_dec23};
    for (auto r : readers) { const int x = r(reset); if (x < 0) return x; v |= x ? 1 : 0; }
    if (reset) {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) { std::lock_guard<std::mutex> lk(g_status_mu); if (g_status[dev].host) g_status[dev].host->range = 0; }
    }
    return v;
}
// Expired waits of the loader / consumer ring (conv_ring_kernels.hip) since the last reset; its own entry point so that a caller
// testing fldr_range_status() != 0 never mistakes a library fault for out-of-range data.  Synchronises.
extern "C" int fldr_ring_status(int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    const int v = fldr_ring_timeouts_read(reset);
    if (reset && v >= 0) {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
            std::lock_guard<std::mutex> lk(g_status_mu);
            if (g_status[dev].host) {
                g_status[dev].host->ring = 0;
                if (hipMemset(g_status[dev].poison, 0, 4) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return -1;
            }
        }
    }
    return v;
}

// sizeof of the descriptor structs as this library was compiled (binding self-check: tests/test_host_cpu.py)
extern "C" int fldr_sizeof_desc(int which) {
    switch (which) {
        case 0: return (int)sizeof(fldr_conv_desc);
        case 1: return (int)sizeof(fldr_spk_conv_desc);
        case 2: return (int)sizeof(fldr_prep_desc);
        case 3: return (int)sizeof(fldr_pca_level);
        case 4: return (int)sizeof(fldr_splat_acc_desc);
        case 5: return (int)sizeof(fldr_splat_gather_desc);
        default: return FLDR_E_ARG;
    }
}

extern "C" int64_t fldr_spk_bytes(int C, int H, int W) {
    if (C <= 0 || H <= 0 || W <= 0) return FLDR_E_ARG;
    return (int64_t)((C + 7) / 8) * 2 * H * W * 16;
}

extern "C" int fldr_spk_pack(const float* src, int64_t src_bstride, void* dst, int N, int C, int H, int W, fldr_stream_t stream) {
    FLDR_CHECK_ARG(src && dst && N > 0 && C > 0 && H > 0 && W > 0);
    const int64_t HW = (int64_t)H * W;
    hipLaunchKernelGGL(spk_pack_kernel, dim3(fldr_cdiv(HW, 256), (C + 7) / 8, N), dim3(256), 0, fldr_s(stream), src, src_bstride,
                       reinterpret_cast<unsigned char*>(dst), fldr_spk_bytes(C, H, W), C, HW);
    FLDR_LAUNCH_RET();
}

extern "C" int fldr_spk_unpack(const void* src, float* dst, int N, int C, int H, int W, fldr_stream_t stream) {
    FLDR_CHECK_ARG(src && dst && N > 0 && C > 0 && H > 0 && W > 0);
    const int64_t HW = (int64_t)H * W;
    hipLaunchKernelGGL(spk_unpack_kernel, dim3(fldr_cdiv(HW, 256), (C + 7) / 8, N), dim3(256), 0, fldr_s(stream),
                       reinterpret_cast<const unsigned char*>(src), fldr_spk_bytes(C, H, W), dst, C, HW);
    FLDR_LAUNCH_RET();
}

// ------------------------------------------------------------------------------------------------
// weights: same body layout as conv_split_kernels.hip ([group][chunk][step][m][kind][lane][8 halves]) behind an
// 8-float header whose second half is the zero block
// ------------------------------------------------------------------------------------------------
__global__ void spk_absmax_kernel(const float* __restrict__ w, int64_t n, float* __restrict__ hdr) {
    __shared__ float red[256];
    float m = 0.0f;
    for (int64_t i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(w[i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x < SPK_HDR) {
        const float mx = red[0];
        float scale = 1.0f;                                  // largest power of two with mx * scale <= 8192
        if (mx > 0.0f) scale = exp2f(floorf(log2f(8192.0f / mx)));
        const float v[SPK_HDR] = {1.0f / scale, scale, mx, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        hdr[threadIdx.x] = v[threadIdx.x];
    }
}

__global__ void spk_prepack_kernel(const float* __restrict__ w, float* __restrict__ wp, int cout, int cin, int nmt,
                                   int n_chunks, int64_t total_h8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total_h8) return;
    const float scale = wp[1];
    const int lane = (int)(i % 64);
    const int kind = (int)((i / 64) % 2);
    const int m = (int)((i / 128) % nmt);
    const int s = (int)((i / (128 * nmt)) % SPK_STEPS);
    const int ch = (int)((i / (128 * nmt * SPK_STEPS)) % n_chunks);
    const int gr = (int)(i / ((int64_t)128 * nmt * SPK_STEPS * n_chunks));
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
    reinterpret_cast<h8*>(wp + SPK_HDR)[i] = v;
}

// section R32: [group of 32 outputs][chunk][tap][hi, lo][lane = channel-group-of-the-chunk * 32 + output][8 channels]
__global__ void spk_prepack32_kernel(const float* __restrict__ w, const float* __restrict__ hdr, float* __restrict__ dst, int cout, int cin,
                                     int n_chunks, int64_t total_h8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total_h8) return;
    const float scale = hdr[1];
    const int lane = (int)(i % 64), kind = (int)((i / 64) % 2), tap = (int)((i / 128) % 9);
    const int ch = (int)((i / (128 * 9)) % n_chunks), gr = (int)(i / ((int64_t)128 * 9 * n_chunks));
    const int co = gr * 32 + (lane & 31);
    h8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = ch * 16 + (lane >> 5) * 8 + j;
        const float x = (co < cout && c < cin) ? w[((int64_t)co * cin + c) * 9 + tap] * scale : 0.0f;
        const _Float16 h = (_Float16)x;
        v[j] = kind == 0 ? h : (_Float16)(x - (float)h);
    }
    reinterpret_cast<h8*>(dst)[i] = v;
}

extern "C" int64_t fldr_conv_spk_prepack_size(int cout, int cin) {
    if (cout <= 0 || cin <= 0 || cout > 96 || cin > SPK_MAX_GROUPS * 8) return FLDR_E_ARG;
    return SPK_HDR + spk_first_section_floats(cout, cin) + spk_r32_section_floats(cout, cin);
}

extern "C" int fldr_conv_spk_prepack(const float* weight, float* wpack, int cout, int cin, fldr_stream_t stream) {
    FLDR_CHECK_ARG(weight && wpack);
    const int64_t total = fldr_conv_spk_prepack_size(cout, cin);
    if (total < 0) return (int)total;
    int nmt, groups;
    spk_geometry(cout, nmt, groups);
    const int n_chunks = (cin + 15) / 16;
    const int64_t first = spk_first_section_floats(cout, cin), total_h8 = first / 4;
    hipLaunchKernelGGL(spk_absmax_kernel, dim3(1), dim3(256), 0, fldr_s(stream), weight, (int64_t)cout * cin * 9, wpack);
    hipLaunchKernelGGL(spk_prepack_kernel, dim3(fldr_cdiv(total_h8, 256)), dim3(256), 0, fldr_s(stream), weight, wpack, cout, cin,
                       nmt, n_chunks, total_h8);
    if (spk_has_r32_section(cout)) {
        const int64_t r_h8 = spk_r32_section_floats(cout, cin) / 4;
        hipLaunchKernelGGL(spk_prepack32_kernel, dim3(fldr_cdiv(r_h8, 256)), dim3(256), 0, fldr_s(stream), weight, wpack, wpack + SPK_HDR + first,
                           cout, cin, n_chunks, r_h8);
    }
    FLDR_LAUNCH_RET();
}

// Persistent workgroups per XCD (32 = one per CU); fldr_debug_spk_wgs_per_xcd changes it for occupancy experiments.
static int g_spk_wgs_per_xcd = 32;
static int g_spk_small_units = 96;                 // launches with at most this many units use 16-channel sub-groups (-1: never)
FLDR_HOOK int fldr_debug_spk_small_units(int v) { if (v != 0) g_spk_small_units = v; return g_spk_small_units; }
FLDR_HOOK int fldr_debug_spk_wgs_per_xcd(int v) { if (v > 0) g_spk_wgs_per_xcd = v; return g_spk_wgs_per_xcd; }
// Pipeline variant: 1 (default) = loader / consumer ring without per-iteration barriers (conv_ring_kernels.hip),
// 0 = the barrier pipeline of this file.  Same arithmetic, bit-identical results.
static int g_spk_variant = 1;
FLDR_HOOK int fldr_debug_spk_variant(int v) { if (v >= 0) g_spk_variant = v; return g_spk_variant; }

#ifdef FLDR_TEST_HOOKS
template <int NMT, int TERMS, bool HAS_RES>
static int spk_launch2(SpkArgs& a, int N, hipStream_t s) {
    using Cfg = SpkCfg<NMT>;
    static std::atomic<uint64_t> attr_done{0};
    if (int e = fldr_set_max_lds(reinterpret_cast<const void*>(&conv3x3_spk_kernel<NMT, TERMS, HAS_RES>), Cfg::LDS_BYTES, attr_done)) return e;
    if (int e = spk_fill_geometry(a, N, g_spk_wgs_per_xcd)) return e;
    hipLaunchKernelGGL((conv3x3_spk_kernel<NMT, TERMS, HAS_RES>), dim3(8 * a.wgs_per_xcd), dim3(512), Cfg::LDS_BYTES, s, a);
    FLDR_LAUNCH_RET();
}

template <int NMT, int TERMS>
static int spk_launch(SpkArgs& a, int N, hipStream_t s) {
    return a.residual ? spk_launch2<NMT, TERMS, true>(a, N, s) : spk_launch2<NMT, TERMS, false>(a, N, s);
}
#endif  // FLDR_TEST_HOOKS

extern "C" int fldr_conv2d_spk(const fldr_spk_conv_desc* d, fldr_stream_t stream) {
    FLDR_CHECK_ARG(d && d->wpack && (d->out_f32 || d->out_spk) && d->n_src >= 1 && d->n_src <= FLDR_CONV_MAX_SRC);
    FLDR_CHECK_ARG(d->N > 0 && d->cin > 0 && d->cin <= SPK_MAX_GROUPS * 8 && d->cout > 0 && d->cout <= 96);
    FLDR_CHECK_ARG(d->cout_store > 0 && d->cout_store <= d->cout && d->H > 0 && d->W > 0);
    FLDR_CHECK_ARG(!d->residual || d->out_f32 || (d->precision & 2));           // (an fp32 residual comes with the fp32 output)
    FLDR_CHECK_ARG(!(d->precision & ~3) && (!(d->precision & 2) || (d->residual && g_spk_variant == 1)));
    if ((int64_t)d->cout_store * d->H * d->W * 4 >= (1ll << 32) || fldr_spk_bytes(96, d->H, d->W) >= (1ll << 32)) return FLDR_E_SHAPE;
    SpkArgs a;
    a.res_spk = (d->precision & 2) ? 1 : 0;
    int gsum = 0, csum = 0;
    for (int s = 0; s < d->n_src; ++s) {
        FLDR_CHECK_ARG(d->src[s] && d->src_c[s] > 0);
        if (d->src_up2[s] && ((d->H | d->W) & 1)) return FLDR_E_SHAPE;
        if (s + 1 < d->n_src && (d->src_c[s] & 7)) return FLDR_E_SHAPE;      // only the last source may have a partial group
        const int ng = (d->src_c[s] + 7) / 8;
        if (gsum + ng > SPK_MAX_GROUPS) return FLDR_E_ARG;
        const int64_t plane = d->src_up2[s] ? (int64_t)(d->H >> 1) * (d->W >> 1) * 16 : (int64_t)d->H * d->W * 16;
        for (int g = 0; g < ng; ++g) {
            a.grp_ptr[gsum + g] = (unsigned long long)(reinterpret_cast<uintptr_t>(d->src[s]) + (uint64_t)g * 2 * plane) | (d->src_up2[s] ? 1ull : 0ull);
            a.grp_bstride[gsum + g] = d->src_bstride[s];
        }
        gsum += ng; csum += d->src_c[s];
    }
    if (csum != d->cin) return FLDR_E_SHAPE;
    for (int g = gsum; g < SPK_MAX_GROUPS; ++g) { a.grp_ptr[g] = 0ull; a.grp_bstride[g] = 0; }
    a.n_levels = 0;
    a.wpack = d->wpack; a.bias = d->bias; a.residual = d->residual;
    a.out_f32 = d->out_f32; a.out_spk = reinterpret_cast<unsigned char*>(d->out_spk);
    a.out_spk_bstride = fldr_spk_bytes(d->cout_store, d->H, d->W);
    a.n_chunks = (d->cin + 15) / 16; a.cout = d->cout; a.cout_store = d->cout_store;
    a.H = d->H; a.W = d->W; a.relu = d->relu;
    int nmt, groups;
    spk_geometry(d->cout, nmt, groups);
    a.groups = groups; a.pack_nmt = nmt;
    a.w32_off = spk_has_r32_section(d->cout) ? (SPK_HDR + spk_first_section_floats(d->cout, d->cin)) * 4 : 0;
    // Small launches (the coarse pyramid levels): fewer units than CUs, and every workgroup would stream the weights
    // of 48 output channels on its own (166 KB for 96 inputs, ~8 us at one CU's DMA rate).  Run the 16-channel kernel
    // on sub-groups of the same weight pack instead: 3x the workgroups, a third of the weight stream each; the
    // results are the same bits (each 16-channel block accumulates independently in the same order).
    if (nmt > 1) {
        const int64_t units = (int64_t)d->N * fldr_cdiv(d->W, SPK_TW) * fldr_cdiv(d->H, SPK_TH) * groups;
        if (units <= g_spk_small_units) { a.groups = (d->cout + 15) / 16; nmt = 1; }
    }
    hipStream_t s = fldr_s(stream);
    if (g_spk_variant == 1) return fldr_spk_ring_dispatch(a, d->N, nmt, (d->precision & 1) ? 1 : 3, g_spk_wgs_per_xcd, s);
#ifndef FLDR_TEST_HOOKS
    return FLDR_E_ARG;                                               // (unreachable: the variant switch is a test-build hook)
#else
    if (d->precision & 1) {
        if (nmt == 1) return spk_launch<1, 1>(a, d->N, s);
        if (nmt == 2) return spk_launch<2, 1>(a, d->N, s);
        return spk_launch<3, 1>(a, d->N, s);
    }
    if (nmt == 1) return spk_launch<1, 3>(a, d->N, s);
    if (nmt == 2) return spk_launch<2, 3>(a, d->N, s);
    return spk_launch<3, 3>(a, d->N, s);
#endif
}

#ifdef FLDR_TEST_HOOKS
// EXPERIMENT (round 6): the convolution of `d` on conv3x3_ringrow_kernel (conv_ring_kernels.hip: ring item = 32 channels x one kernel row).
// wrow: fldr_debug_ringrow_prepack's section.  cin % 32 == 0, cout % 48 == 0 or % 32 == 0, every channel stored, packed output only,
// no residual, full-resolution or nearest-x2 sources as fldr_conv2d_spk.
int fldr_spk_ringrow_dispatch(SpkArgs& a, const float* wrow, int N, int wgs_per_xcd_max, hipStream_t s);
FLDR_HOOK int fldr_debug_conv2d_ringrow(const fldr_spk_conv_desc* d, const float* wrow, fldr_stream_t stream) {
    FLDR_CHECK_ARG(d && wrow && d->wpack && d->out_spk && !d->out_f32 && !d->residual && d->n_src >= 1 && d->n_src <= FLDR_CONV_MAX_SRC);
    FLDR_CHECK_ARG(d->N > 0 && d->cin > 0 && d->cin % 32 == 0 && d->cin <= SPK_MAX_GROUPS * 8 && d->cout > 0 && d->cout <= 96 && d->cout_store == d->cout);
    FLDR_CHECK_ARG((d->cout % 48 == 0 || d->cout % 32 == 0) && d->precision == 0 && d->H > 0 && d->W > 0);
    if (fldr_spk_bytes(96, d->H, d->W) >= (1ll << 32)) return FLDR_E_SHAPE;
    SpkArgs a;
    a.res_spk = 0;
    int gsum = 0;
    for (int s = 0; s < d->n_src; ++s) {
        FLDR_CHECK_ARG(d->src[s] && d->src_c[s] > 0 && !(d->src_c[s] & 7));
        if (d->src_up2[s] && ((d->H | d->W) & 1)) return FLDR_E_SHAPE;
        const int ng = d->src_c[s] / 8;
        if (gsum + ng > SPK_MAX_GROUPS) return FLDR_E_ARG;
        const int64_t plane = d->src_up2[s] ? (int64_t)(d->H >> 1) * (d->W >> 1) * 16 : (int64_t)d->H * d->W * 16;
        for (int g = 0; g < ng; ++g) {
            a.grp_ptr[gsum + g] = (unsigned long long)(reinterpret_cast<uintptr_t>(d->src[s]) + (uint64_t)g * 2 * plane) | (d->src_up2[s] ? 1ull : 0ull);
            a.grp_bstride[gsum + g] = d->src_bstride[s];
        }
        gsum += ng;
    }
    if (gsum * 8 != d->cin) return FLDR_E_SHAPE;
    for (int g = gsum; g < SPK_MAX_GROUPS; ++g) { a.grp_ptr[g] = 0ull; a.grp_bstride[g] = 0; }
    a.n_levels = 0;
    a.wpack = d->wpack; a.bias = d->bias; a.residual = nullptr; a.out_f32 = nullptr;
    a.out_spk = reinterpret_cast<unsigned char*>(d->out_spk);
    a.out_spk_bstride = fldr_spk_bytes(d->cout_store, d->H, d->W);
    a.n_chunks = d->cin / 16; a.cout = d->cout; a.cout_store = d->cout_store;
    a.H = d->H; a.W = d->W; a.relu = d->relu; a.pack_nmt = 0; a.w32_off = 0;
    return fldr_spk_ringrow_dispatch(a, wrow, d->N, g_spk_wgs_per_xcd, fldr_s(stream));
}
#endif

// The same convolution (same weights, bias, ReLU) over SEVERAL inputs of different sizes in ONE launch of the ring pipeline:
// rec_ctx_ds.0 / .2 over the six pyramid levels (fLDRnet.py:148-162 runs them level by level: 12 launches, ten of them too
// small to fill the chip).  descs[l]: one sample (N = 1), one packed source of cin channels, the same wpack / bias / relu /
// cout / cout_store / precision in every entry; residual / out_f32 / out_spk given for all entries or for none.  Units are
// numbered level after level (largest first is best: descs[0] should be the finest level) and dealt to the persistent
// workgroups exactly like the units of a single launch; results are the bits of n_levels separate fldr_conv2d_spk calls.
extern "C" int fldr_conv2d_spk_levels(const fldr_spk_conv_desc* descs, int n_levels, fldr_stream_t stream) {
    FLDR_CHECK_ARG(descs && n_levels >= 1 && n_levels <= SPK_MAX_LEVELS);
    if (n_levels == 1) return fldr_conv2d_spk(descs, stream);
    if (g_spk_variant != 1) return FLDR_E_ARG;                       // ring pipeline only
    const fldr_spk_conv_desc& d0 = descs[0];
    FLDR_CHECK_ARG(d0.wpack && (d0.out_f32 || d0.out_spk) && d0.n_src == 1 && d0.src[0] && d0.N == 1);
    FLDR_CHECK_ARG(d0.cin > 0 && d0.cin <= SPK_MAX_GROUPS * 8 && d0.cout > 0 && d0.cout <= 96 && d0.cout_store > 0 && d0.cout_store <= d0.cout);
    FLDR_CHECK_ARG(!d0.residual || d0.out_f32 || (d0.precision & 2));
    FLDR_CHECK_ARG(!(d0.precision & ~3) && (!(d0.precision & 2) || d0.residual));
    SpkArgs a;
    a.res_spk = (d0.precision & 2) ? 1 : 0;
    const int ng = (d0.cin + 7) / 8;
    for (int g = 0; g < SPK_MAX_GROUPS; ++g) { a.grp_ptr[g] = g < ng ? (unsigned long long)reinterpret_cast<uintptr_t>(d0.src[0]) : 0ull; a.grp_bstride[g] = 0; }
    a.wpack = d0.wpack; a.bias = d0.bias; a.residual = d0.residual;
    a.out_f32 = d0.out_f32; a.out_spk = reinterpret_cast<unsigned char*>(d0.out_spk);
    a.out_spk_bstride = 0;
    a.n_chunks = (d0.cin + 15) / 16; a.cout = d0.cout; a.cout_store = d0.cout_store;
    a.H = d0.H; a.W = d0.W; a.relu = d0.relu;
    int nmt, groups;
    spk_geometry(d0.cout, nmt, groups);
    a.groups = groups; a.pack_nmt = nmt; a.w32_off = 0;
    a.n_levels = n_levels;
    int64_t units = 0;
    for (int l = 0; l < n_levels; ++l) {
        const fldr_spk_conv_desc& d = descs[l];
        FLDR_CHECK_ARG(d.n_src == 1 && d.src[0] && d.N == 1 && d.H > 0 && d.W > 0 && !d.src_up2[0] && d.src_c[0] == d0.cin);
        FLDR_CHECK_ARG(d.wpack == d0.wpack && d.bias == d0.bias && d.cin == d0.cin && d.cout == d0.cout && d.cout_store == d0.cout_store);
        FLDR_CHECK_ARG(d.relu == d0.relu && d.precision == d0.precision);
        FLDR_CHECK_ARG(!d.residual == !d0.residual && !d.out_f32 == !d0.out_f32 && !d.out_spk == !d0.out_spk);
        if ((int64_t)d.cout_store * d.H * d.W * 4 >= (1ll << 32) || fldr_spk_bytes(96, d.H, d.W) >= (1ll << 32)) return FLDR_E_SHAPE;
        SpkArgs::SpkLevel& v = a.lv[l];
        v.H = d.H; v.W = d.W; v.tiles_x = fldr_cdiv(d.W, SPK_TW); v.n_tiles = v.tiles_x * fldr_cdiv(d.H, SPK_TH);
        v.unit0 = (int32_t)units; v.pad = 0; v.m_tiles_x = 0; v.pad2 = 0;
        v.in_off = reinterpret_cast<const char*>(d.src[0]) - reinterpret_cast<const char*>(d0.src[0]);
        v.out_spk_off = d.out_spk ? reinterpret_cast<const char*>(d.out_spk) - reinterpret_cast<const char*>(d0.out_spk) : 0;
        v.out_f32_off = d.out_f32 ? reinterpret_cast<const char*>(d.out_f32) - reinterpret_cast<const char*>(d0.out_f32) : 0;
        v.res_off = d.residual ? reinterpret_cast<const char*>(d.residual) - reinterpret_cast<const char*>(d0.residual) : 0;
        units += (int64_t)v.n_tiles * groups;
    }
    for (int l = n_levels; l < SPK_MAX_LEVELS; ++l) { a.lv[l] = a.lv[n_levels - 1]; a.lv[l].unit0 = 0x7fffffff; }
    if (units >= (1 << 28)) return FLDR_E_SHAPE;
    return fldr_spk_ring_dispatch_levels(a, (int)units, nmt, (d0.precision & 1) ? 1 : 3, g_spk_wgs_per_xcd, fldr_s(stream));
}
