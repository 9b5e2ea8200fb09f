// 3x3 / stride 1 / pad 1 convolutions on split-packed activations: LOADER / CONSUMER RING (no per-iteration barrier).
//
// Same arithmetic, operand layouts, weight pack, unit walk and LDS stage image as conv_spk_kernels.hip (x = hi + lo in
// fp16; hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16, fp32 accumulation, chunk -> tap pair -> term order per
// accumulator), so results are bit-identical to it and to conv_split_kernels.hip.  What changes is who does what:
//
//   * The barrier pipeline let all 8 waves issue LDS-DMA between their MFMAs and closed every 16-channel iteration with
//     one s_barrier.  Its phase stamps showed the cost: an LDS-DMA instruction holds its wave's
//     issue port for 60-180 cycles (7 per wave and iteration), the two waves of a SIMD had to be skewed to hide
//     bookkeeping, and the early wave then idled ~1,800 of 4,400 cycles per iteration at the barrier — the matrix pipe
//     was busy 47 % of the time.
//   * Here the first NC waves are CONSUMERS (NC = 8: one tile row each, two per SIMD, so that one's epilogue, poll and
//     LDS latencies hide under the other's MFMAs — the default; NC = 4: two rows each, 30 % fewer LDS operand reads per
//     MFMA but nothing to hide its epilogue under: measured 3-5 % slower) and never touch global memory inside the
//     loop except for the epilogue of a unit; the last four waves are LOADERS (one per SIMD): loader w streams
//     plane w of the input tile (hi g0, hi g1, lo g0, lo g1) and a quarter of the weight slab of every iteration by
//     LDS-DMA.  Their hand-off is a 3-slot ring with two LDS words per slot (MI355X_MICROARCH.md, ring-gemm):
//         FULL[slot] += 1  by each loader once its part of the fill has landed (counted s_waitcnt vmcnt),
//         FREE[slot] += 1  by each consumer after its last operand read of the slot;
//     a consumer spins on FULL[slot] >= 4 * use, a loader on FREE[slot] >= 4 * (use - 1) before refilling.  Waves
//     therefore run up to two iterations apart instead of meeting at a barrier 6 times per unit.
//   * Every spin is bounded (RING_SPIN_LIMIT polls, then the wave gives up, counts the event in fldr_ring_timeouts and
//     runs on): the grid always drains.
#include "spk_common.h"

#ifndef RING_SLOTS
#define RING_SLOTS 3                            // stages of the LDS ring (2: 105 KB for the 48-channel kernel, 55 KB of the CU left to other kernels)
#endif
#define RING_NLOAD 4
// Resident-weight variant (test build; template flag RW; 16-output-channel launches of <= RING_RW_MAX_CHUNKS input chunks: dec2 48 -> 16,
// conv_flow2.8 / conv_flow_bottom.8): the whole weight pack of the workgroup's output group is copied into LDS once (10 KB per chunk),
// a ring stage is the four input planes only (22 KB instead of 32 KB) and the ring has RING_RW_SLOTS stages with two fills in flight
// per loader wave instead of one (MI355X_MICROARCH.md, ring-gemm: "at least three slots more than the K-steps a loader keeps in flight").
#define RING_RW_MAX_CHUNKS 4
#ifndef RING_RW_SLOTS
#define RING_RW_SLOTS 5
#endif
#ifndef RING_RW_INFLIGHT
#define RING_RW_INFLIGHT 2
#endif
#ifndef RING_NT_STORES
#define RING_NT_STORES 0                        // 1: packed outputs of the fast epilogue leave with non-temporal stores — alone 48 -> 48 24.7 -> 22.8 us, dec1 91.7 -> 87.1, 96 -> 96 equal; in the forward the next layer reads them: bench 484.6-487.2 vs 488.1-489.6 with 0
#endif
#ifndef RING_PD_THIN
#define RING_PD_THIN 1                          // operand read-ahead (steps) of the 16-output-channel kernels (2: measured equal, +20-40 registers)
#endif
#define RING_SPIN_LIMIT (1 << 21)
#ifndef RING_LOADER_PRIO
#define RING_LOADER_PRIO 0                      // wave priority of the loaders (0 / 1 / 3 measured equal within noise)
#endif
#ifndef RING_FIN_PRIO
#define RING_FIN_PRIO 0                         // wave priority of a consumer during its epilogue (0 = unchanged)
#endif

#ifdef RING_STAMPS
// Diagnostic build only (tools/stamps): per-phase s_memtime sums of consumer wave 0 and loader wave 4 of two workgroups.
__device__ unsigned long long fldr_ring_stamp_buf[4 * 8];
__device__ unsigned long long fldr_ring_trace[4 * 24 * 4];            // [wave slot][iteration < 24][event] absolute s_memtime, workgroup 0
FLDR_HOOK int fldr_debug_read_ring_trace(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fldr_ring_trace), sizeof(unsigned long long) * 4 * 24 * 4);
}
#define RSTAMP(var) unsigned long long var; { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
FLDR_HOOK int fldr_debug_read_ring_stamps(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fldr_ring_stamp_buf), sizeof(unsigned long long) * 32);
}
#else
#define RSTAMP(var)
#endif

__device__ int fldr_ring_timeouts;
// Polls a bounded wait makes before it gives up: a KERNEL ARGUMENT (SpkArgs::spin_limit, set by the launchers from g_ring_spin_limit) —
// as a __device__ variable it cost every launch a global-memory round trip in its prologue or in front of the second poll of its first
// wait: +0.7 us on each of the 39 ring launches of a forward (tools/small_conv_probe.py, round 6).  Test build:
// fldr_debug_ring_spin_limit(0) makes every wait that is not satisfied at once expire — the way the fault path (counter, status block,
// poisoned outputs) is exercised.
static int g_ring_spin_limit = RING_SPIN_LIMIT;
FLDR_HOOK int fldr_debug_ring_spin_limit(int v) {
    g_ring_spin_limit = v < 0 ? RING_SPIN_LIMIT : v;
    return g_ring_spin_limit;
}
FLDR_HOOK int fldr_debug_ring_timeouts(void) {
    int v = -1;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(fldr_ring_timeouts), sizeof(int)) != hipSuccess) return -1;
    return v;
}

FLDR_TU_STATUS(ring)
// Product-visible status of the bounded ring waits (fldr_range_status bit 1): number of waits that expired — a wave then ran on
// with operands that had not landed, i.e. a convolution may have produced wrong output — since load / the last reset.
int fldr_ring_timeouts_read(int reset) {
    int v = -1;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(fldr_ring_timeouts), sizeof(int)) != hipSuccess) return -1;
    if (v > 0 && reset) { const int z = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(fldr_ring_timeouts), &z, sizeof(int)) != hipSuccess) return -1; }
    return v;
}

// TW: tile width in pixels (32, or 16 for launches whose 8 x 32 tiles fill the last round of persistent workgroups badly:
// see ring_pick_tile_width).  The LDS input plane is (8 + 2) x (TW + 2) pixels of 16 bytes, padded to a multiple of 256 bytes.
template <int NMT, int TW = SPK_TW, bool RW = false>
struct RingCfg {
    static constexpr int W_BYTES = SPK_STEPS * NMT * 2 * 1024;
    static constexpr int NBLK = SPK_STEPS * NMT * 2;                    // 1-KB weight blocks per chunk: (step, m, kind)
    static constexpr int NWL = (NBLK + RING_NLOAD - 1) / RING_NLOAD;    // weight blocks per loader wave
    static constexpr int IW = TW + 2;                                   // input tile width (pixels)
    static constexpr int CB = TW / 16;                                  // 16-pixel column blocks per tile row
    static constexpr int PLANE = (SPK_IH * IW * 16 + 255) / 256 * 256;  // bytes per LDS plane (TW = 32: 5632 = SPK_PLANE)
    static constexpr int NXI = (SPK_IH * IW + 63) / 64;                 // 64-pixel DMA pieces per input plane (the last ones overlap)
    static constexpr int K_DMA = (RW ? 0 : NWL) + NXI;                  // DMA instructions per loader wave and fill
    static constexpr int IN_OFF = RW ? 0 : W_BYTES;                     // the four input planes inside a stage
    static constexpr int STAGE = IN_OFF + 4 * PLANE;
    static constexpr int SLOTS = RW ? RING_RW_SLOTS : RING_SLOTS;
    static constexpr int INFLIGHT = RW ? RING_RW_INFLIGHT : 1;          // fills a loader wave has in flight behind the one it is issuing
    static constexpr int WRES_OFF = SLOTS * STAGE;                      // RW: the weight slabs of all chunks
    static constexpr int CTR_OFF = WRES_OFF + (RW ? RING_RW_MAX_CHUNKS * W_BYTES : 0);   // FULL[8] at +0, FREE[8] at +32
    static constexpr int BIAS_OFF = CTR_OFF + 64;                       // bias of the workgroup's 16 * NMT output channels (fp32)
    static constexpr int LDS_BYTES = BIAS_OFF + 64 * NMT;
    static_assert(SLOTS <= 8 && INFLIGHT + 2 <= SLOTS && INFLIGHT * K_DMA <= 15, "ring shape");
    static_assert(TW == 16 || TW == 32, "tile width");
    static_assert(TW != SPK_TW || PLANE == SPK_PLANE, "the 32-pixel plane is the barrier pipeline's");
    static_assert(PLANE / 16 >= 64 && NXI * 64 >= SPK_IH * IW, "DMA pieces cover the plane");
    static_assert(K_DMA <= 15, "counted vmcnt wait uses the 4 low bits");
    static_assert(LDS_BYTES <= 160 * 1024, "ring does not fit the LDS");
};

__device__ __forceinline__ uint32_t ring_peek(uint32_t lds_addr) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_addr) : "memory");
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

// Spin until the counter at lds_addr reaches `target` (counters only grow).  Bounded: see the file header.  An expired wait is a library
// fault and is never silent.  The WAVE is faulted from then on: `limit` (wave-uniform, an SGPR the wave holds anyway) becomes negative.
// A faulted wave never waits again (every later wait returns after one look: the launch stays bounded), and
//   * a faulted CONSUMER writes NaN for every unit it still finishes (its accumulators may hold products of operands that had not landed)
//     and stops releasing slots, so the loaders cannot overwrite a slot on its behalf: they expire on FREE in turn;
//   * a faulted LOADER retires at once (it never fills a slot that has not been released), so every consumer expires on FULL in turn;
//   * where it leaves its loop, a faulted wave REPORTS (ring_report_fault): the event is counted (fldr_ring_status) and stored into the
//     host-visible status block and the device's frame poison (common.h: fldr_status_raise_ring) — before the kernel ends, so every frame
//     synthesised behind this launch is NaN.
// A unit is therefore stored as values only by a consumer all of whose waits were satisfied: its operands had landed and were not
// overwritten.  What this costs a launch that does not fault was measured on the same box against the round-5 kernels
// (profiles/r06_ring_wait_ab.txt) and shaped the code:
//   * no shared poison word: the epilogue's test is one scalar compare (a word in LDS, read in every epilogue, cost the residual kernels 14
//     spilled registers: +18 % on the 48-channel residual launches of the 4K forward);
//   * the report is NOT inlined behind the wait: ~40 instructions of cold code between every wait and the code that follows it cost each
//     of the ~35 small launches of a forward 0.35 us (one more instruction-cache miss per wait site in a launch whose every fetch is cold);
//   * eight polls per trip of the loop, each with its own exit, as the compiler unrolled it by itself while the limit was a constant
//     (rolled — counter, compare and two more branches between polls — every ring launch took 0.3 us longer);
//   * every value of the loop is wave-uniform: scalar loop control.
__device__ __forceinline__ void ring_wait_ge(uint32_t lds_addr, uint32_t target, int& limit) {
    if (ring_peek(lds_addr) >= target) return;
    bool landed = false;
    for (int spins = 0; spins < limit && !landed; spins += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            __builtin_amdgcn_s_sleep(1);
            if (ring_peek(lds_addr) >= target) { landed = true; break; }
        }
    }
#ifndef RING_AB_NOFAULT                                                  // (A/B builds only: what the fault bookkeeping costs a launch)
    if (!landed) limit = -1;
#endif
}
__device__ __forceinline__ void ring_report_fault(int lane) {
    if (lane == 0) {
        atomicAdd(&fldr_ring_timeouts, 1);
        fldr_status_raise_ring();
    }
}

// One lane adds 1.  LDS instructions of a wave execute in issue order, so everything the wave read from (or, for a
// loader after its counted vmcnt wait, everything its DMA wrote to) the slot is done when the add lands.
__device__ __forceinline__ void ring_signal(uint32_t lds_addr, int lane) {
    __builtin_amdgcn_sched_barrier(0);
    if (lane == 0) asm volatile("ds_add_u32 %0, %1" :: "v"(lds_addr), "v"(1u) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// ML: multi-level launch (SpkArgs::lv: units of several images of different sizes, one source tensor each; rec_ctx_ds over the
// pyramid levels).  A template parameter, not a kernel argument: the single-level kernels keep exactly their registers (with
// the level geometry as run-time state they needed a scratch reservation, and that costs ~2 us per launch).
template <int NMT, int TERMS, bool HAS_RES, int NC, int TW, bool ML = false, bool RW = false>
__global__ __launch_bounds__((NC + RING_NLOAD) * 64) void conv3x3_ring_kernel(SpkArgs a) {
    // NC consumer waves (4: two tile rows each, one consumer per SIMD; 8: one row each, two per SIMD)
    using Cfg = RingCfg<NMT, TW, RW>;
    constexpr int SLOTS = Cfg::SLOTS;
    constexpr int CB = Cfg::CB;
    constexpr int RING_NCONS = NC;
    constexpr int MTOT = 16 * NMT;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_chunks = a.n_chunks;
    const uint32_t ctr = (uint32_t)(uintptr_t)(klptr_t)(smem + Cfg::CTR_OFF);   // LDS byte address of FULL[0]; FREE[s] at +32 + 4 s
#ifdef RING_AB_CONSTLIMIT                                               // A/B builds only: the limit as a constant instead of a kernel argument
    int spin_limit = RING_SPIN_LIMIT;
#else
    int spin_limit = a.spin_limit;                                        // negative once a wait of this wave has expired (ring_wait_ge)
#endif

    if (tid < 16) reinterpret_cast<uint32_t*>(smem + Cfg::CTR_OFF)[tid] = 0u;
    if (tid >= 64 && tid < 64 + 16 * NMT) {                               // (unit -> output group as below: constant over the workgroup)
        const int u0 = (blockIdx.x & 7) * a.units_per_xcd + (blockIdx.x >> 3);
        int co = (u0 % a.groups) * 16 * NMT + (tid - 64);
        co = co < a.cout ? co : a.cout - 1;
        reinterpret_cast<float*>(smem + Cfg::BIAS_OFF)[tid - 64] = a.bias ? a.bias[co] : 0.0f;
    }
    // Units of this workgroup: as in conv_spk_kernels.hip (XCD x owns the contiguous range [x*upx, (x+1)*upx)).
    const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
    const int u_end = min((xcd + 1) * a.units_per_xcd, a.n_units);
    const int u_first = xcd * a.units_per_xcd + slot_id;
    if constexpr (RW) {
        // the weight slabs of every chunk of this workgroup's output group, once: 1-KB blocks (chunk, step, m, kind) dealt to all waves
        const int g0 = (u_first < a.n_units ? u_first : 0) % a.groups;
        const int sub = a.pack_nmt / NMT, pgrp = g0 / sub, msel = (g0 - pgrp * sub) * NMT;
        const int pack_w_bytes = SPK_STEPS * a.pack_nmt * 2 * 1024;
        const char* wsrc = reinterpret_cast<const char*>(a.wpack + SPK_HDR) + (int64_t)pgrp * n_chunks * pack_w_bytes;
        for (int b = wave; b < n_chunks * Cfg::NBLK; b += NC + RING_NLOAD) {
            const int c = b / Cfg::NBLK, blk = b - c * Cfg::NBLK;
            const int step = blk / (2 * NMT), mk = blk - step * 2 * NMT;
            __builtin_amdgcn_global_load_lds((kgptr_t)(wsrc + (int64_t)c * pack_w_bytes + ((step * a.pack_nmt + msel) * 2 + mk) * 1024 + lane * 16),
                                             (klptr_t)(smem + Cfg::WRES_OFF + c * Cfg::W_BYTES + blk * 1024), 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                               // vmcnt(0): my blocks have landed
    }
    __syncthreads();                                                      // the only workgroup barrier of the kernel
    if (u_first >= u_end) return;                                        // workgroup-uniform
    const int my_units = (u_end - u_first + a.wgs_per_xcd - 1) / a.wgs_per_xcd;
    const int total = my_units * n_chunks;
    const int grp0 = u_first % a.groups;                                 // the output group is constant over a workgroup's units

    if (wave >= RING_NCONS) {
        // =========================================== loader ===========================================
        __builtin_amdgcn_s_setprio(RING_LOADER_PRIO);
        const int lw = wave - RING_NCONS;
        const int ip = lw, ikind = ip >> 1, igrp = ip & 1;               // LDS plane: kind = ip >> 1, group of the chunk = ip & 1
        const char* zero_blk = reinterpret_cast<const char*>(a.wpack + 4);
        const int sub = a.pack_nmt / NMT;
        const int pgrp = grp0 / sub, msel = (grp0 - pgrp * sub) * NMT;
        const int pack_w_bytes = SPK_STEPS * a.pack_nmt * 2 * 1024;
        int w_blk[Cfg::NWL], x_piece[Cfg::NXI];
        uint32_t w_voff[Cfg::NWL];
#pragma unroll
        for (int i = 0; i < Cfg::NWL; ++i) {
            w_blk[i] = min(lw * Cfg::NWL + i, Cfg::NBLK - 1);            // (past the slab: re-fetch its last block)
            const int step = w_blk[i] / (2 * NMT), mk = w_blk[i] - step * 2 * NMT;
            w_voff[i] = (uint32_t)((step * a.pack_nmt + msel) * 2 + mk) * 1024u + (uint32_t)lane * 16u;
        }
#pragma unroll
        for (int i = 0; i < Cfg::NXI; ++i) x_piece[i] = min(i * 64, Cfg::PLANE / 16 - 64);
        // input-group table in VGPR lanes (lane l = group l, advanced to this wave's hi or lo plane)
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
        int iss_u = u_first, iss_c = 0, iss_n = 0;
        uint32_t g_full[Cfg::NXI], g_half[Cfg::NXI];                      // byte offsets in a plane; ~0u = outside the image
        // multi-level launch: base of the unit's level-l source (its group g, kind k plane lies (2 g + k) * ml_plane bytes further)
        unsigned long long ml_base = 0ull;
        long long ml_plane = 0;
        auto issue_geometry = [&]() {
            int uH = a.H, uW = a.W, tiles_x = a.tiles_x, tile, ty;
            const int t = spk_div(iss_u, a.m_groups, a.groups);
            if constexpr (ML) {
                int l = 0;
                for (int k = 1; k < a.n_levels; ++k) l = t * a.groups >= a.lv[k].unit0 ? k : l;
                uH = a.lv[l].H; uW = a.lv[l].W; tiles_x = a.lv[l].tiles_x;
                tile = t - a.lv[l].unit0 / a.groups;
                iss_n = 0;
                ml_base = a.grp_ptr[0] + (unsigned long long)a.lv[l].in_off;
                ml_plane = (long long)uH * uW * 16;
                ty = tile / tiles_x;
            } else {
                iss_n = spk_div(t, a.m_tiles, a.n_tiles);
                tile = t - iss_n * a.n_tiles;
                ty = spk_div(tile, a.m_tiles_x, a.tiles_x);
            }
            const int oy0 = ty * SPK_TH, ox0 = (tile - ty * tiles_x) * TW;
#pragma unroll
            for (int i = 0; i < Cfg::NXI; ++i) {
                const int e = x_piece[i] + lane;
                const int y = e / Cfg::IW, x = e % Cfg::IW;
                const int gy = oy0 - 1 + y, gx = ox0 - 1 + x;
                const bool ok = e < SPK_IH * Cfg::IW && gy >= 0 && gy < uH && gx >= 0 && gx < uW;
                g_full[i] = ok ? (uint32_t)(gy * uW + gx) * 16u : ~0u;
                g_half[i] = ok ? (uint32_t)((gy >> 1) * (uW >> 1) + (gx >> 1)) * 16u : ~0u;
            }
        };
        const char* const iss_w = reinterpret_cast<const char*>(a.wpack + SPK_HDR) + (int64_t)pgrp * n_chunks * pack_w_bytes;
        issue_geometry();
        int st = 0;                                                       // slot of fill k
        uint32_t free_target = 0;                                         // RING_NCONS * (uses of the slot so far)
#ifdef RING_STAMPS
        unsigned long long ls_prep = 0, ls_free = 0, ls_fire = 0, ls_land = 0;
        RSTAMP(l_begin)
#endif
        for (int k = 0; k < total; ++k) {
            RSTAMP(l0)
            // addresses of fill k (before the FREE wait: they do not depend on it)
            const char* wbase = iss_w + (int64_t)iss_c * pack_w_bytes;
            const int gi = iss_c * 2 + igrp;
            const uint32_t e_lo = __builtin_amdgcn_readlane((int)(uint32_t)tab_ptr, gi), e_hi = __builtin_amdgcn_readlane((int)(uint32_t)(tab_ptr >> 32), gi);
            const uint32_t b_lo = __builtin_amdgcn_readlane((int)(uint32_t)tab_bs, gi), b_hi = __builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)tab_bs >> 32), gi);
            unsigned long long e = ((unsigned long long)e_hi << 32) | e_lo;
            const long long bs = (long long)(((unsigned long long)b_hi << 32) | b_lo);
            if constexpr (ML) e = e == 0ull ? 0ull : ml_base + (unsigned long long)((2 * gi + ikind) * ml_plane);   // (padding groups stay null)
            const bool nul = e == 0ull, up2 = (e & 1ull) != 0ull;
            const char* base = reinterpret_cast<const char*>(static_cast<uintptr_t>(e & ~1ull)) + (int64_t)iss_n * bs;
            const char* dptr[Cfg::NXI];
#pragma unroll
            for (int i = 0; i < Cfg::NXI; ++i) {
                const uint32_t off = up2 ? g_half[i] : g_full[i];
                dptr[i] = (off != ~0u && !nul) ? base + off : zero_blk;
            }
            if (++iss_c == n_chunks) { iss_c = 0; iss_u += a.wgs_per_xcd; if (iss_u < u_end) issue_geometry(); }
            RSTAMP(l1)
            if (free_target) ring_wait_ge(ctr + 32 + 4 * st, free_target, spin_limit);
            if (spin_limit < 0) break;                                    // expired: this loader retires (below), nothing is filled over a slot in use
            RSTAMP(l2)
            unsigned char* stage = smem + st * Cfg::STAGE;
#if defined(RING_ABLATE) && RING_ABLATE == 3                          // diagnostic: no DMA traffic after the prologue fills
            if (k >= SLOTS) { wbase = reinterpret_cast<const char*>(a.wpack + SPK_HDR); for (int i = 0; i < Cfg::NXI; ++i) dptr[i] = zero_blk; }
#endif
            if constexpr (!RW) {
#pragma unroll
                for (int i = 0; i < Cfg::NWL; ++i)
                    __builtin_amdgcn_global_load_lds((kgptr_t)(wbase + w_voff[i]), (klptr_t)(stage + w_blk[i] * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < Cfg::NXI; ++i)
                __builtin_amdgcn_global_load_lds((kgptr_t)dptr[i], (klptr_t)(stage + Cfg::IN_OFF + ip * Cfg::PLANE + x_piece[i] * 16), 16, 0, 0);
            RSTAMP(l3)
            if (k >= Cfg::INFLIGHT) {                                     // fill k-INFLIGHT has landed once at most INFLIGHT * K_DMA of my loads are outstanding
                __builtin_amdgcn_s_waitcnt(0x0F70 | (Cfg::INFLIGHT * Cfg::K_DMA));
                ring_signal(ctr + 4 * ((st + SLOTS - Cfg::INFLIGHT) % SLOTS), lane);
            }
            RSTAMP(l4)
#ifdef RING_STAMPS
            ls_prep += l1 - l0; ls_free += l2 - l1; ls_fire += l3 - l2; ls_land += l4 - l3;
            if (blockIdx.x == 0 && (lw == 0 || lw == 3) && lane == 0 && k < 24) {
                unsigned long long* tr = fldr_ring_trace + ((lw == 0 ? 2 : 3) * 24 + k) * 4;
                tr[0] = l1; tr[1] = l2; tr[2] = l3; tr[3] = l4;
            }
#endif
            if (++st == SLOTS) { st = 0; free_target += RING_NCONS; }
        }
        if (spin_limit < 0) { ring_report_fault(lane); __builtin_amdgcn_s_waitcnt(0x0F70); return; }   // retired on an expired wait: report; its last fills are never announced
        // the last INFLIGHT fills (st = slot after the last fill), oldest first
        if constexpr (Cfg::INFLIGHT == 2) {
            if (total >= 2) {
                __builtin_amdgcn_s_waitcnt(0x0F70 | Cfg::K_DMA);
                ring_signal(ctr + 4 * ((st + SLOTS - 2) % SLOTS), lane);
            }
        } else {
            static_assert(Cfg::INFLIGHT == 1, "drain sequence");
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                               // vmcnt(0)
        ring_signal(ctr + 4 * ((st + SLOTS - 1) % SLOTS), lane);
#ifdef RING_STAMPS
        RSTAMP(l_end)
        if ((blockIdx.x == 0 || blockIdx.x == 101) && wave == RING_NCONS && lane == 0) {
            unsigned long long* o = fldr_ring_stamp_buf + ((blockIdx.x == 0 ? 0 : 2) + 1) * 8;
            o[0] = ls_prep; o[1] = ls_free; o[2] = ls_fire; o[3] = ls_land; o[5] = total; o[6] = l_end - l_begin;
        }
#endif
        return;
    }

    // ============================================= consumer =============================================
    constexpr int ROWS = SPK_TH / NC;                                     // tile rows per consumer wave
    const int cw = wave;                                                  // rows ROWS * cw ... of the 8 x 32 tile
    const int lj = lane & 15, lg = lane >> 4;
    constexpr int NQ = CB * ROWS;                                         // pixel blocks: q = row_in_wave * CB + column block
    int boff[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        boff[q] = Cfg::IN_OFF + (lg & 1) * Cfg::PLANE + ((ROWS * cw + q / CB) * Cfg::IW + (q % CB) * 16 + lj) * 16;
    const int tap_sel = lg >> 1;
    f4 acc[NMT][NQ];
#pragma unroll
    for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[m][q] = f4{0.0f, 0.0f, 0.0f, 0.0f};
    int cur_u = u_first, cur_c = 0;
    const float inv_scale = a.wpack[0];
    // geometry of the unit being finished: constants of the kernel, or (ML) the values of the unit's level, set by unit_decode
    int uH = a.H, uW = a.W;
    int64_t HW = (int64_t)a.H * a.W;
    uint32_t HW32 = (uint32_t)HW;                                         // byte offsets below fit 32 bits (host-checked)
    int64_t u_spk_off = 0, u_f32_off = 0, u_res_off = 0;                  // byte offsets of the unit's level (ML)
    // (sample, tile row, tile column) of unit u
    auto unit_decode = [&](int u, int& n, int& ty, int& tx) __attribute__((always_inline)) {
        const int t = spk_div(u, a.m_groups, a.groups);
        if constexpr (ML) {
            int l = 0;
            for (int k = 1; k < a.n_levels; ++k) l = t * a.groups >= a.lv[k].unit0 ? k : l;
            uH = a.lv[l].H; uW = a.lv[l].W; HW = (int64_t)uH * uW; HW32 = (uint32_t)HW;
            u_spk_off = a.lv[l].out_spk_off; u_f32_off = a.lv[l].out_f32_off; u_res_off = a.lv[l].res_off;
            const int tile = t - a.lv[l].unit0 / a.groups;
            n = 0; ty = tile / a.lv[l].tiles_x; tx = tile - ty * a.lv[l].tiles_x;
        } else {
            n = spk_div(t, a.m_tiles, a.n_tiles);
            const int tile = t - n * a.n_tiles;
            ty = spk_div(tile, a.m_tiles_x, a.tiles_x); tx = tile - ty * a.tiles_x;
        }
    };
    const int gout = (a.cout_store + 7) >> 3;
    const int cbase = grp0 * MTOT;
    // bias of this lane's 4 channels per 16-channel block, in registers
    float bias_r[NMT][4];
#pragma unroll
    for (int m = 0; m < NMT; ++m) {
        const f4 bv = *reinterpret_cast<const f4*>(smem + Cfg::BIAS_OFF + (m * 16 + lg * 4) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) bias_r[m][r] = bv[r];
    }
    auto bias_of = [&](int m) __attribute__((always_inline)) -> f4 { return f4{bias_r[m][0], bias_r[m][1], bias_r[m][2], bias_r[m][3]}; };
    auto unit_pixels = [&](int u, uint32_t (&po)[NQ], int& n) {
        int ty, tx;
        unit_decode(u, n, ty, tx);
        const int ox0 = tx * TW;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int oy = ty * SPK_TH + ROWS * cw + q / CB;
            const int ox = ox0 + (q % CB) * 16 + lj;
            po[q] = (oy < uH && ox < uW) ? (uint32_t)(oy * uW + ox) : ~0u;
        }
    };
    // The residual of the unit's outputs, prefetched one iteration ahead: fp32 NCHW (4 loads of 4 bytes per block and lane), or —
    // a.res_spk, round 4: rec_ctx_ds.2 adds the PCA features, which then exist split-packed only — the hi and lo halves of the
    // lane's 4 channels (2 loads of 8 bytes), value = hi + lo: the fp32 value up to the split's 2^-22 relative rounding of lo.
    float res_r[HAS_RES ? NMT : 1][HAS_RES ? NQ : 1][4];
    auto residual_prefetch = [&]() {
        uint32_t po[NQ]; int n;
        unit_pixels(cur_u, po, n);
        if (a.res_spk) {
            const char* resn = reinterpret_cast<const char*>(a.residual) + u_res_off + (int64_t)n * ((a.cout_store + 7) >> 3) * 2 * HW * 16;
#pragma unroll
            for (int m = 0; m < NMT; ++m)
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    int co0 = cbase + m * 16 + lg * 4;
                    co0 = co0 < a.cout_store ? co0 : 0;                 // (channels past cout_store are never stored)
                    const uint32_t off = ((uint32_t)(co0 >> 3) * 2u * HW32 + (po[q] != ~0u ? po[q] : 0u)) * 16u + (uint32_t)((co0 >> 2) & 1) * 8u;
                    const h4 hi = *reinterpret_cast<const h4*>(resn + off), lo = *reinterpret_cast<const h4*>(resn + (off + HW32 * 16u));
#pragma unroll
                    for (int r = 0; r < 4; ++r) res_r[HAS_RES ? m : 0][HAS_RES ? q : 0][r] = (float)hi[r] + (float)lo[r];
                }
            return;
        }
        const float* resn = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.residual) + u_res_off) + (int64_t)n * a.cout_store * HW;
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int co = cbase + m * 16 + lg * 4 + r;
                    co = co < a.cout_store ? co : a.cout_store - 1;
                    res_r[HAS_RES ? m : 0][HAS_RES ? q : 0][r] = resn[(uint32_t)co * HW32 + (po[q] != ~0u ? po[q] : 0u)];
                }
    };
    // Epilogue of a unit: scale, bias, ReLU, residual, hi/lo split and the stores, block by block (registers of a block
    // are dead once it is stored; the stores drain under the next unit's MFMAs — a consumer never waits on vmcnt
    // except for the residual it prefetched one iteration earlier).  Two paths, chosen per unit by a wave-uniform test:
    // the FAST one (tile inside the image, every channel of the group stored — all of the 4K forward's big launches)
    // has no predication at all: ~20 VALU instructions and the stores per (pixel block, 16-channel block); the general
    // one predicates every store.  ReLU is max(v, floor) with floor = 0 or -FLT_MAX: no select.
    const float relu_floor = a.relu ? 0.0f : -3.402823466e+38f;
    // (no kernel-wide range flag: a per-lane bool carried through the iteration loop costs a 64-bit SGPR mask — this kernel spills SGPRs —
    // and mask bookkeeping at every merge; the guarded paths are rare and note the event in the sticky flag on the spot)
    const bool grp_full = cbase + MTOT <= a.cout_store && !(a.cout_store & 7);      // wave-uniform, constant over the kernel
#ifdef RING_STAMPS
    unsigned long long cs_f_dec = 0, cs_f_fast = 0, cs_f_b0 = 0, fq0 = 0;
#endif
    auto finish_store = [&]() {
        RSTAMP(f0)
        // (opaque copies of the lane coordinates: without them the compiler hoists this path's per-block offsets and
        // predicates out of the iteration loop and keeps ~20 VGPRs live across the MFMA steps, or spills them)
        int lj = lane & 15, lg = lane >> 4;
        asm volatile("" : "+v"(lj), "+v"(lg));
        int n, ty, tx;
        unit_decode(cur_u, n, ty, tx);
        const int oy0 = ty * SPK_TH + ROWS * cw, ox0 = tx * TW;
        char* outn = a.out_f32 ? reinterpret_cast<char*>(a.out_f32 + (int64_t)n * a.cout_store * HW) + u_f32_off : nullptr;
        char* spkn = a.out_spk ? reinterpret_cast<char*>(a.out_spk) + (int64_t)n * a.out_spk_bstride + u_spk_off : nullptr;
        const bool inside = oy0 + ROWS <= uH && ox0 + TW <= uW;      // wave-uniform
        // a wait of this wave expired (ring_wait_ge): its accumulators may hold products of operands that had not landed — the unit is
        // written as NaN (general path below) instead
        const bool poisoned = spin_limit < 0;                            // wave-uniform, scalar
        RSTAMP(f1)
        if (grp_full && inside) {
            const uint32_t p0 = (uint32_t)(oy0 * uW + ox0 + lj);
            // pass 1: the finished values in place of the accumulators, and per lane the sum of their magnitudes — the range guard of the
            // split is decided ONCE per unit and wave (common.h: fldr_guard_trips) instead of a compare and two clamps per value
            float abs_sum = 0.0f;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int m = 0; m < NMT; ++m) {
                    const f4 bsv = bias_of(m);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = fmaxf(acc[m][q][r] * inv_scale + bsv[r], relu_floor);
                        if constexpr (HAS_RES) v += res_r[HAS_RES ? m : 0][HAS_RES ? q : 0][r];
                        acc[m][q][r] = v;
                        abs_sum += fabsf(v);
                    }
                }
            // pass 2: split and store
            auto emit = [&](auto guardedc) __attribute__((always_inline)) {
                constexpr bool GUARDED = decltype(guardedc)::value;
                bool range_bad = false;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const uint32_t pq = p0 + (uint32_t)((q / CB) * uW + (q % CB) * 16);
#ifdef RING_STAMPS
                    if (q == 1) { RSTAMP(fq) fq0 = fq; }
#endif
#pragma unroll
                    for (int m = 0; m < NMT; ++m) {
                        const int co0 = cbase + m * 16 + lg * 4;
                        float ov[4];
                        h4 ohi, olo;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            ov[r] = acc[m][q][r];
                            acc[m][q][r] = 0.0f;
                            _Float16 h, l;
                            if constexpr (GUARDED) spk_split(ov[r], h, l, range_bad); else fldr_split_plain(ov[r], h, l);
                            ohi[r] = h; olo[r] = l;
                        }
                        if (outn) {
                            const uint32_t off = ((uint32_t)co0 * HW32 + pq) * 4u;
#pragma unroll
                            for (int r = 0; r < 4; ++r) *reinterpret_cast<float*>(outn + (off + (uint32_t)r * HW32 * 4u)) = ov[r];
                        }
                        if (spkn) {
#if defined(RING_ABLATE) && RING_ABLATE == 6                          // diagnostic: the finish without its packed stores
                            asm volatile("" :: "v"(ohi), "v"(olo));
#else
                            const uint32_t off = ((uint32_t)(co0 >> 3) * 2u * HW32 + pq) * 16u + (uint32_t)(lg & 1) * 8u;
#if RING_NT_STORES
                            __builtin_nontemporal_store(ohi, reinterpret_cast<h4*>(spkn + off));
                            __builtin_nontemporal_store(olo, reinterpret_cast<h4*>(spkn + (off + HW32 * 16u)));
#else
                            *reinterpret_cast<h4*>(spkn + off) = ohi;
                            *reinterpret_cast<h4*>(spkn + (off + HW32 * 16u)) = olo;
#endif
#endif
                        }
                    }
                }
                if constexpr (GUARDED) { if (a.out_spk) fldr_note_range(range_bad); }
            };
            if (!poisoned) {
                if (fldr_guard_trips(abs_sum)) emit(std::true_type{}); else emit(std::false_type{});
#ifdef RING_STAMPS
                { RSTAMP(f2) cs_f_dec += f1 - f0; cs_f_fast += f2 - f1; cs_f_b0 += fq0 - f1; }
#endif
                return;
            }
            // (poisoned: the general path below writes NaN for every value of the unit and clears the accumulators)
        }
        uint32_t po[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int oy = oy0 + q / CB, ox = ox0 + (q % CB) * 16 + lj;
            po[q] = (oy < uH && ox < uW) ? (uint32_t)(oy * uW + ox) : ~0u;
        }
        const bool quads = !(a.cout_store & 3);                          // whole quads of channels: one predicate per 4 stores
        bool range_bad = false;                                          // (one flag raise per unit: the raise is cold code, kept out of the unrolled body)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int m = 0; m < NMT; ++m) {
                const int co0 = cbase + m * 16 + lg * 4;
                float ov[4];
                h4 ohi, olo;
                const f4 bsv = bias_of(m);
                float xs[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = fmaxf(acc[m][q][r] * inv_scale + bsv[r], relu_floor);
                    if constexpr (HAS_RES) v += res_r[HAS_RES ? m : 0][HAS_RES ? q : 0][r];
                    ov[r] = v;
                    acc[m][q][r] = 0.0f;
                    xs[r] = co0 + r < a.cout_store ? v : 0.0f;
                }
                {
                    _Float16 h[4], l[4];
                    fldr_split_hl_group(xs, h, l, range_bad);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { ohi[r] = h[r]; olo[r] = l[r]; }
                }
                if (poisoned) {                                          // (past the split's range guard, which would saturate a NaN)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { ov[r] = __builtin_nanf(""); ohi[r] = (_Float16)__builtin_nanf(""); olo[r] = (_Float16)__builtin_nanf(""); }
                }
                if (outn) {
                    const uint32_t off = ((uint32_t)co0 * HW32 + po[q]) * 4u;
                    if (quads) {
                        if (co0 < a.cout_store && po[q] != ~0u) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) *reinterpret_cast<float*>(outn + (off + (uint32_t)r * HW32 * 4u)) = ov[r];
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (co0 + r < a.cout_store && po[q] != ~0u) *reinterpret_cast<float*>(outn + (off + (uint32_t)r * HW32 * 4u)) = ov[r];
                    }
                }
                if (spkn) {
                    const int go = co0 >> 3;
                    if (go < gout && po[q] != ~0u) {
                        const uint32_t off = ((uint32_t)go * 2u * HW32 + po[q]) * 16u + (uint32_t)(lg & 1) * 8u;
                        *reinterpret_cast<h4*>(spkn + off) = ohi;
                        *reinterpret_cast<h4*>(spkn + (off + HW32 * 16u)) = olo;
                    }
                }
            }
        if (a.out_spk && spin_limit >= 0) fldr_note_range(range_bad);    // (a poisoned unit's accumulators are garbage: not a range event)
    };

    int st_cur = 0;
    uint32_t full_target = RING_NLOAD;                                    // RING_NLOAD * (use index of the slot + 1)
#ifdef RING_STAMPS
    unsigned long long cs_wait = 0, cs_steps = 0, cs_fin = 0;
    RSTAMP(c_begin)
#endif
    // the MFMA steps of one iteration on slot st_cur
    auto steps = [&]() __attribute__((always_inline)) {
        const unsigned char* sb = smem + st_cur * Cfg::STAGE;
        const unsigned char* win = (RW ? smem + Cfg::WRES_OFF + cur_c * Cfg::W_BYTES : sb) + lane * 16;
        // Operand registers.  The lo weights are single-buffered (they feed only the last third of a step's MFMAs and are
        // fetched at its start).
        // PD: how many steps ahead the operand reads run (RING_PD_THIN for the 16-output-channel kernels, whose 3 NQ matrix instructions per
        // step do not cover an LDS round trip).  Reading two steps ahead, every operand triple-buffered, was built and measured EQUAL (48 -> 16
        // at 1152x1920: 135.3 vs 134.6 us): those kernels are bound by the LDS array (one 1-KB pixel operand read feeds 1.5 matrix
        // instructions: 8 waves x 6 reads per 192 matrix cycles = 250 of 256 B/clk) and by their epilogue, not by read latency.
        constexpr int PD = (NMT == 1 && TERMS == 3) ? RING_PD_THIN : 1;
        constexpr int NB = PD + 1;
        h8 bh[NB][NQ], bl[NB][NQ], ah[NB][NMT], al[PD > 1 ? NB : 1][NMT];
        auto tap_off = [&](int s) {
            // taps of step s: 2s and 2s+1 (tap 9 = the zero-weight pad tap: re-reads tap 8's pixels, finite values)
            const int tA = 2 * s, tB = 2 * s + 1 < 9 ? 2 * s + 1 : 8;
            const int offA = ((tA / 3) * Cfg::IW + tA % 3) * 16, offB = ((tB / 3) * Cfg::IW + tB % 3) * 16;
            return tap_sel ? offB : offA;
        };
        auto ld_bh = [&](int buf, int s) {
            const int toff = tap_off(s);
#pragma unroll
            for (int q = 0; q < NQ; ++q) bh[buf][q] = *reinterpret_cast<const h8*>(sb + boff[q] + toff);
        };
        auto ld_ah = [&](int buf, int s) {
#pragma unroll
            for (int m = 0; m < NMT; ++m) ah[buf][m] = *reinterpret_cast<const h8*>(win + ((s * NMT + m) * 2 + 0) * 1024);
        };
        auto ld_bl = [&](int buf, int s) {
            if constexpr (TERMS > 1) {
                const int toff = tap_off(s);
#pragma unroll
                for (int q = 0; q < NQ; ++q) bl[buf][q] = *reinterpret_cast<const h8*>(sb + 2 * Cfg::PLANE + boff[q] + toff);
            }
        };
        auto ld_al = [&](int s, int buf = 0) {
            if constexpr (TERMS > 1) {
#pragma unroll
                for (int m = 0; m < NMT; ++m) al[buf][m] = *reinterpret_cast<const h8*>(win + ((s * NMT + m) * 2 + 1) * 1024);
            }
        };
        // issue order = consumption order of the term-major MFMA sequence (hi x hi, hi x lo, lo x hi)
        auto ld = [&](int buf, int s) { ld_bh(buf, s); ld_ah(buf, s); ld_bl(buf, s); };
        constexpr int N_MFMA = NQ * TERMS * NMT, N_DS = TERMS > 1 ? 2 * NQ + 2 * NMT : NQ + NMT;
#ifdef RING_NTAIL
        constexpr int N_TAIL = N_MFMA >= 12 ? RING_NTAIL : (N_MFMA >= 6 ? 2 : 0);
#else
        constexpr int N_TAIL = N_MFMA >= 12 ? 4 : (N_MFMA >= 6 ? 2 : 0);
#endif
        ld(0, 0);
        ld_al(0);
        if constexpr (PD > 1) { ld(1, 1); ld_al(1, 1); }
        __builtin_amdgcn_sched_barrier(0);                               // keep step 0's reads out of the interleave pattern below
#pragma unroll
        for (int s = 0; s < SPK_STEPS; ++s) {
#if defined(RING_ABLATE) && RING_ABLATE == 1                          // diagnostic: operands read once per iteration
            if (s + 1 < SPK_STEPS) { for (int q = 0; q < NQ; ++q) { bh[(s + 1) & 1][q] = bh[s & 1][q]; bl[(s + 1) & 1][q] = bl[s & 1][q]; }
                                              for (int m = 0; m < NMT; ++m) { ah[(s + 1) & 1][m] = ah[s & 1][m]; } }
#else
            if constexpr (PD > 1) {
                if (s + PD < SPK_STEPS) { ld((s + PD) % NB, s + PD); ld_al(s + PD, (s + PD) % NB); }   // lands while this and the next step's MFMAs run
            } else {
                if (s > 0) ld_al(s);                                     // (after the previous step's last lo-weight MFMA in program order)
                if (s + 1 < SPK_STEPS) ld((s + 1) & 1, s + 1);            // lands while this step's MFMAs run
            }
#endif
#pragma unroll
            for (int term = 0; term < TERMS; ++term) {
#pragma unroll
                for (int m = 0; m < NMT; ++m)
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const h8 av = term == 2 ? al[PD > 1 ? s % NB : 0][m] : ah[s % NB][m];
                        const h8 bv = term == 1 ? bl[s % NB][q] : bh[s % NB][q];
#if defined(RING_ABLATE) && RING_ABLATE == 2                          // diagnostic: no MFMAs
                        asm volatile("" :: "v"(av), "v"(bv));
#else
                        acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc[m][q], 0, 0, 0);
#endif
                    }
            }
            // spread the next step's LDS reads evenly between this step's MFMAs
            spk_step_pattern_n<N_MFMA, N_DS, N_TAIL, 0>(s + PD < SPK_STEPS);
        }
    };
    // one iteration's hand-shake around `steps`: wait for the slot, run, release it, step to the next slot
    int g = 0;
#ifdef RING_STAMPS
    unsigned long long c0 = 0, c1 = 0, c2 = 0;
#endif
    auto iter_begin = [&]() __attribute__((always_inline)) {
#ifdef RING_STAMPS
        { RSTAMP(t) c0 = t; }
#endif
#if defined(RING_ABLATE) && RING_ABLATE == 5                          // diagnostic: consumers never look at FULL after the first fills (timing only: stale operands)
        if (g < SLOTS)
#endif
        ring_wait_ge(ctr + 4 * st_cur, full_target, spin_limit);
#ifdef RING_STAMPS
        { RSTAMP(t) c1 = t; }
#endif
    };
    auto iter_end = [&]() __attribute__((always_inline)) {
        if (spin_limit >= 0) ring_signal(ctr + 32 + 4 * st_cur, lane);                        // all my operand reads of the slot are issued: FREE
#ifdef RING_STAMPS
        { RSTAMP(t) c2 = t; }
#endif
    };
    auto iter_close = [&]() __attribute__((always_inline)) {
#ifdef RING_STAMPS
        RSTAMP(c3)
        cs_wait += c1 - c0; cs_steps += c2 - c1; cs_fin += c3 - c2;
        if (blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0 && g < 24) {
            unsigned long long* tr = fldr_ring_trace + ((wave == 0 ? 0 : 1) * 24 + g) * 4;
            tr[0] = c0; tr[1] = c1; tr[2] = c2; tr[3] = c3;
        }
#endif
        if (++st_cur == SLOTS) { st_cur = 0; full_target += RING_NLOAD; }
        ++g;
    };
    while (g < total) {
        const bool last = cur_c == n_chunks - 1;                          // workgroup-uniform
        if constexpr (HAS_RES) { if (last) residual_prefetch(); }
        iter_begin();
        steps();
        iter_end();
        if (last) {
            if constexpr (HAS_RES) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the residual prefetched at the top
            if (RING_FIN_PRIO) __builtin_amdgcn_s_setprio(RING_FIN_PRIO);
#if defined(RING_ABLATE) && RING_ABLATE == 4                          // diagnostic: no epilogue (accumulators cleared, nothing stored)
            for (int m = 0; m < NMT; ++m) for (int q = 0; q < NQ; ++q) { asm volatile("" :: "v"(acc[m][q])); acc[m][q] = f4{0.0f, 0.0f, 0.0f, 0.0f}; }
            cur_c = 0; cur_u += a.wgs_per_xcd;
            iter_close();
#else
            finish_store();
            cur_c = 0; cur_u += a.wgs_per_xcd;
            iter_close();
#endif
            if (RING_FIN_PRIO) __builtin_amdgcn_s_setprio(0);
        } else {
            ++cur_c;
            iter_close();
        }
    }
    if (spin_limit < 0) ring_report_fault(lane);                          // a wait of this consumer expired: its units were written as NaN; report
#ifdef RING_STAMPS
    RSTAMP(c_end)
    if ((blockIdx.x == 0 || blockIdx.x == 101) && wave == 0 && lane == 0) {
        unsigned long long* o = fldr_ring_stamp_buf + (blockIdx.x == 0 ? 0 : 2) * 8;
        o[0] = cs_wait; o[1] = cs_steps; o[2] = cs_fin; o[3] = cs_f_dec; o[4] = cs_f_fast; o[7] = cs_f_b0; o[5] = total; o[6] = c_end - c_begin;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// 32x32x16 variant (round 4; the previous review's item 2a): output channels in groups of 32, one v_mfma_f32_32x32x16_f16 tile per
// consumer wave (its tile row of 32 pixels x the unit's 32 output channels).
//   * K = 16 = ONE tap x the chunk's two 8-channel groups (lane half = group): 9 matrix instructions per chunk and split term,
//     no zero-weight pad tap (the 16x16x32 kernel spends 10 tap slots on 9 taps);
//   * a stage = 18 KB of weights ([tap][hi, lo] 1-KB blocks of this group and chunk) + the four input planes, unpadded (21,760 B):
//     40,192 B, so the ring has FOUR slots (the 48-channel kernel: 3 x 52.5 KB) — MI355X_MICROARCH.md, ring-gemm: a ring one slot
//     short of (fills in flight + 3) took 7-23 % longer;
//   * operands read two taps ahead (triple-buffered: 48 registers), one accumulator tile (16 registers).
// Summation order per output: chunk -> tap -> (hi hi, hi lo, lo hi) — not the 16x16x32 kernels' (tap pairs, term-major), so results
// agree with them to fp32 accumulation rounding only; error vs fp64 tested against the exact-fp32 kernel's.
// Scope of this first version: single-level launches, packed output only, every channel stored (cout_store == cout, cout % 32 == 0),
// no residual.  Loader, unit walk, hand-shake and timeouts: as in conv3x3_ring_kernel.
// ------------------------------------------------------------------------------------------------
struct Ring32Cfg {
    static constexpr int NBLK = 18;                                      // 1-KB weight blocks per chunk: (tap, kind)
    static constexpr int W_BYTES = NBLK * 1024;
    static constexpr int NWL = (NBLK + RING_NLOAD - 1) / RING_NLOAD;     // 5
    static constexpr int IW = SPK_TW + 2;
    static constexpr int PLANE = SPK_IH * IW * 16;                       // 5,440 B: no padding (the lane halves of an operand read different planes, no bank relation needed)
    static constexpr int NXI = (SPK_IH * IW + 63) / 64;                  // 6
    static constexpr int K_DMA = NWL + NXI;                              // 11
    static constexpr int STAGE = W_BYTES + 4 * PLANE;                    // 40,192
    static constexpr int SLOTS = 4;
    static constexpr int CTR_OFF = SLOTS * STAGE;                        // FULL[8] at +0, FREE[8] at +32
    static constexpr int BIAS_OFF = CTR_OFF + 64;
    static constexpr int LDS_BYTES = BIAS_OFF + 128;
    static_assert(K_DMA <= 15 && LDS_BYTES <= 160 * 1024, "ring32 shape");
};

__global__ __launch_bounds__((8 + RING_NLOAD) * 64) void conv3x3_ring32_kernel(SpkArgs a) {
    using Cfg = Ring32Cfg;
    constexpr int SLOTS = Cfg::SLOTS, NC = 8;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_chunks = a.n_chunks;
    const uint32_t ctr = (uint32_t)(uintptr_t)(klptr_t)(smem + Cfg::CTR_OFF);
#ifdef RING_AB_CONSTLIMIT                                               // A/B builds only: the limit as a constant instead of a kernel argument
    int spin_limit = RING_SPIN_LIMIT;
#else
    int spin_limit = a.spin_limit;                                        // negative once a wait of this wave has expired (ring_wait_ge)
#endif
    if (tid < 16) reinterpret_cast<uint32_t*>(smem + Cfg::CTR_OFF)[tid] = 0u;
    if (tid >= 64 && tid < 96) {
        const int u0 = (blockIdx.x & 7) * a.units_per_xcd + (blockIdx.x >> 3);
        const int co = (u0 % a.groups) * 32 + (tid - 64);
        reinterpret_cast<float*>(smem + Cfg::BIAS_OFF)[tid - 64] = (a.bias && co < a.cout) ? a.bias[co] : 0.0f;
    }
    __syncthreads();
    const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
    const int u_end = min((xcd + 1) * a.units_per_xcd, a.n_units);
    const int u_first = xcd * a.units_per_xcd + slot_id;
    if (u_first >= u_end) return;
    const int my_units = (u_end - u_first + a.wgs_per_xcd - 1) / a.wgs_per_xcd;
    const int total = my_units * n_chunks;
    const int grp0 = u_first % a.groups;

    if (wave >= NC) {
        // =========================================== loader ===========================================
        const int lw = wave - NC;
        const int ip = lw, ikind = ip >> 1, igrp = ip & 1;
        const char* zero_blk = reinterpret_cast<const char*>(a.wpack + 4);
        int w_blk[Cfg::NWL], x_piece[Cfg::NXI];
#pragma unroll
        for (int i = 0; i < Cfg::NWL; ++i) w_blk[i] = min(lw * Cfg::NWL + i, Cfg::NBLK - 1);
#pragma unroll
        for (int i = 0; i < Cfg::NXI; ++i) x_piece[i] = min(i * 64, Cfg::PLANE / 16 - 64);
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
        int iss_u = u_first, iss_c = 0, iss_n = 0;
        uint32_t g_full[Cfg::NXI], g_half[Cfg::NXI];
        auto issue_geometry = [&]() {
            const int t = spk_div(iss_u, a.m_groups, a.groups);
            iss_n = spk_div(t, a.m_tiles, a.n_tiles);
            const int tile = t - iss_n * a.n_tiles;
            const int ty = spk_div(tile, a.m_tiles_x, a.tiles_x);
            const int oy0 = ty * SPK_TH, ox0 = (tile - ty * a.tiles_x) * SPK_TW;
#pragma unroll
            for (int i = 0; i < Cfg::NXI; ++i) {
                const int e = x_piece[i] + lane;
                const int y = e / Cfg::IW, x = e % Cfg::IW;
                const int gy = oy0 - 1 + y, gx = ox0 - 1 + x;
                const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                g_full[i] = ok ? (uint32_t)(gy * a.W + gx) * 16u : ~0u;
                g_half[i] = ok ? (uint32_t)((gy >> 1) * (a.W >> 1) + (gx >> 1)) * 16u : ~0u;
            }
        };
        // section R32 of the pack: [group of 32 outputs][chunk][tap][hi, lo] 1-KB blocks behind the first section
        const char* const iss_w = reinterpret_cast<const char*>(a.wpack) + a.w32_off + (int64_t)grp0 * n_chunks * Cfg::W_BYTES;
        issue_geometry();
        int st = 0;
        uint32_t free_target = 0;
        for (int k = 0; k < total; ++k) {
            const char* wbase = iss_w + (int64_t)iss_c * Cfg::W_BYTES + lane * 16;
            const int gi = iss_c * 2 + igrp;
            const uint32_t e_lo = __builtin_amdgcn_readlane((int)(uint32_t)tab_ptr, gi), e_hi = __builtin_amdgcn_readlane((int)(uint32_t)(tab_ptr >> 32), gi);
            const uint32_t b_lo = __builtin_amdgcn_readlane((int)(uint32_t)tab_bs, gi), b_hi = __builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)tab_bs >> 32), gi);
            const unsigned long long e = ((unsigned long long)e_hi << 32) | e_lo;
            const long long bs = (long long)(((unsigned long long)b_hi << 32) | b_lo);
            const bool nul = e == 0ull, up2 = (e & 1ull) != 0ull;
            const char* base = reinterpret_cast<const char*>(static_cast<uintptr_t>(e & ~1ull)) + (int64_t)iss_n * bs;
            const char* dptr[Cfg::NXI];
#pragma unroll
            for (int i = 0; i < Cfg::NXI; ++i) {
                const uint32_t off = up2 ? g_half[i] : g_full[i];
                dptr[i] = (off != ~0u && !nul) ? base + off : zero_blk;
            }
            if (++iss_c == n_chunks) { iss_c = 0; iss_u += a.wgs_per_xcd; if (iss_u < u_end) issue_geometry(); }
            if (free_target) ring_wait_ge(ctr + 32 + 4 * st, free_target, spin_limit);
            if (spin_limit < 0) break;                                    // expired: this loader retires (below), nothing is filled over a slot in use
            unsigned char* stage = smem + st * Cfg::STAGE;
#pragma unroll
            for (int i = 0; i < Cfg::NWL; ++i)
                __builtin_amdgcn_global_load_lds((kgptr_t)(wbase + w_blk[i] * 1024), (klptr_t)(stage + w_blk[i] * 1024), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < Cfg::NXI; ++i)
                __builtin_amdgcn_global_load_lds((kgptr_t)dptr[i], (klptr_t)(stage + Cfg::W_BYTES + ip * Cfg::PLANE + x_piece[i] * 16), 16, 0, 0);
            if (k > 0) {
                __builtin_amdgcn_s_waitcnt(0x0F70 | Cfg::K_DMA);
                ring_signal(ctr + 4 * ((st + SLOTS - 1) % SLOTS), lane);
            }
            if (++st == SLOTS) { st = 0; free_target += NC; }
        }
        if (spin_limit < 0) { ring_report_fault(lane); __builtin_amdgcn_s_waitcnt(0x0F70); return; }   // retired on an expired wait: report; its last fills are never announced
        __builtin_amdgcn_s_waitcnt(0x0F70);
        ring_signal(ctr + 4 * ((st + SLOTS - 1) % SLOTS), lane);
        return;
    }

    // ============================================= consumer =============================================
    typedef float f16v __attribute__((ext_vector_type(16)));
    const int cw = wave;                                                  // tile row
    const int lj = lane & 31, lh = lane >> 5;
    const uint32_t b_lane = (uint32_t)(Cfg::W_BYTES + lh * Cfg::PLANE + (cw * Cfg::IW + lj) * 16);
    f16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    int cur_u = u_first, cur_c = 0;
    const float inv_scale = a.wpack[0];
    const int64_t HW = (int64_t)a.H * a.W;
    const uint32_t HW32 = (uint32_t)HW;
    float bias_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bias_r[r] = reinterpret_cast<const float*>(smem + Cfg::BIAS_OFF)[(r & 3) + 8 * (r >> 2) + 4 * lh];
    const float relu_floor = a.relu ? 0.0f : -3.402823466e+38f;
    auto finish_store = [&]() {
        const int t = spk_div(cur_u, a.m_groups, a.groups);
        const int n = spk_div(t, a.m_tiles, a.n_tiles);
        const int tile = t - n * a.n_tiles;
        const int ty = spk_div(tile, a.m_tiles_x, a.tiles_x), tx = tile - ty * a.tiles_x;
        const int oy = ty * SPK_TH + cw, ox = tx * SPK_TW + lj;
        const bool ok = oy < a.H && ox < a.W;
        unsigned char* spkn = a.out_spk + (int64_t)n * a.out_spk_bstride;
        const uint32_t pq = ok ? (uint32_t)(oy * a.W + ox) : 0u;
        float abs_sum = 0.0f;                                             // (the range guard of the split: once per unit and wave)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[r] = fmaxf(acc[r] * inv_scale + bias_r[r], relu_floor);
            abs_sum += fabsf(acc[r]);
        }
        const bool poisoned = spin_limit < 0;                             // a wait of this wave expired (ring_wait_ge): the unit is written as NaN
        const bool guard = fldr_guard_trips(abs_sum);
        bool bad = false;
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += 4) {
            const int co0 = grp0 * 32 + 8 * (r0 >> 2) + 4 * lh;
            h4 ohi, olo;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[r0 + r];
                acc[r0 + r] = 0.0f;
                _Float16 h, l;
                if (guard) spk_split(v, h, l, bad); else fldr_split_plain(v, h, l);
                ohi[r] = h; olo[r] = l;
            }
            if (poisoned) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { ohi[r] = (_Float16)__builtin_nanf(""); olo[r] = (_Float16)__builtin_nanf(""); }
            }
            if (ok) {
                const uint32_t off = ((uint32_t)(co0 >> 3) * 2u * HW32 + pq) * 16u + (uint32_t)lh * 8u;
                *reinterpret_cast<h4*>(spkn + off) = ohi;
                *reinterpret_cast<h4*>(spkn + (off + HW32 * 16u)) = olo;
            }
        }
        if (guard && !poisoned) fldr_note_range(bad);
    };
    int st_cur = 0, g = 0;
    uint32_t full_target = RING_NLOAD;
    while (g < total) {
        const bool last = cur_c == n_chunks - 1;
        ring_wait_ge(ctr + 4 * st_cur, full_target, spin_limit);
        {
            const unsigned char* sb = smem + st_cur * Cfg::STAGE;
            const unsigned char* wb = sb + lane * 16;
            const unsigned char* xb = sb + b_lane;
            h8 Ah[3], Al[3], Bh[3], Bl[3];
            auto ld = [&](int t) __attribute__((always_inline)) {
                const int off = ((t / 3) * Cfg::IW + t % 3) * 16;
                Ah[t % 3] = *reinterpret_cast<const h8*>(wb + (2 * t) * 1024);
                Al[t % 3] = *reinterpret_cast<const h8*>(wb + (2 * t + 1) * 1024);
                Bh[t % 3] = *reinterpret_cast<const h8*>(xb + off);
                Bl[t % 3] = *reinterpret_cast<const h8*>(xb + 2 * Cfg::PLANE + off);
            };
            ld(0); ld(1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                // (the fences keep the reads two taps ahead of their use: left alone, the scheduler sinks them to one MFMA before it —
                // 76 registers instead of 100, and an exposed LDS round trip per tap)
                if (t + 2 < 9) ld(t + 2);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[t % 3], Bh[t % 3], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[t % 3], Bl[t % 3], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[t % 3], Bh[t % 3], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (spin_limit >= 0) ring_signal(ctr + 32 + 4 * st_cur, lane);                        // all my operand reads of the slot are issued: FREE
        if (last) { finish_store(); cur_c = 0; cur_u += a.wgs_per_xcd; } else ++cur_c;
        if (++st_cur == SLOTS) { st_cur = 0; full_target += RING_NLOAD; }
        ++g;
    }
    if (spin_limit < 0) ring_report_fault(lane);                          // a wait of this consumer expired: its units were written as NaN; report
}

#ifdef FLDR_TEST_HOOKS
// ------------------------------------------------------------------------------------------------
// EXPERIMENT (round 6, test build only): ring item = (32 input channels, ONE kernel row).
// The 16x16x32 kernel above spends 5 K = 32 steps on the 9 taps x 16 channels of a chunk: the tenth half-step multiplies a zero
// pad tap (10 % of the issued matrix instructions).  Pairing tap 8 of two chunks couples two ring slots (round 3: no gain).  Here a
// K = 32 step is ONE tap x 32 channels (lane group lg = 8-channel group lg of the 32-channel block) and a ring item carries the
// three taps of one kernel row dy: 3 steps per item, 9 per 32 channels, 27 per 96 — no pad tap, no coupling between items.
//   stage = weights of (32 channels, dy): 3 taps x NMT x [hi, lo] 1-KB blocks (18 KB at 48 outputs)
//         + 8 planes (4 groups x hi / lo) of the 8 window rows this dy needs x 34 pixels (4,352 B each): 53,248 B -> 3 slots.
// The price: every window row is staged once per kernel row it serves (24 row-loads per 32 channels instead of 20: +66 % LDS-DMA
// bytes per unit with the weights counted) and a unit is 9 ring items instead of 6 (hand-shakes, operand pipeline restarts).
// Summation order: 32-channel block -> dy -> dx -> (hi hi, hi lo, lo hi): equal to the other kernels to fp32 accumulation rounding.
// Scope: cin % 32 == 0, cout % (16 NMT) == 0 all stored, packed output only, no residual, single level.
// ------------------------------------------------------------------------------------------------
template <int NMT>
struct RingRowCfg {
    static constexpr int NBLK = 3 * NMT * 2;                              // 1-KB weight blocks per item: (dx, m, kind)
    static constexpr int W_BYTES = NBLK * 1024;
    static constexpr int NWL = (NBLK + RING_NLOAD - 1) / RING_NLOAD;
    static constexpr int IW = SPK_TW + 2;
    static constexpr int PLANE = SPK_TH * IW * 16;                        // 4,352 B: the 8 rows of one kernel row's window
    static constexpr int NXI = (SPK_TH * IW + 63) / 64;                   // 5 pieces per plane (the last one overlaps)
    static constexpr int K_DMA = NWL + 2 * NXI;                           // a loader wave: its quarter of the weights + the hi and lo plane of ONE group
    static constexpr int STAGE = W_BYTES + 8 * PLANE;
    static constexpr int SLOTS = 3;
    static constexpr int CTR_OFF = SLOTS * STAGE;
    static constexpr int BIAS_OFF = CTR_OFF + 64;
    static constexpr int LDS_BYTES = BIAS_OFF + 64 * NMT;
    static_assert(K_DMA <= 15 && LDS_BYTES <= 160 * 1024 && PLANE % 256 == 0, "ring-row shape");
};

template <int NMT>
__global__ __launch_bounds__((8 + RING_NLOAD) * 64) void conv3x3_ringrow_kernel(SpkArgs a, const float* __restrict__ wrow) {
    using Cfg = RingRowCfg<NMT>;
    constexpr int SLOTS = Cfg::SLOTS, NC = 8, NQ = 2;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n32 = a.n_chunks;                                           // (the launcher passes 32-channel blocks here)
    const int n_items = 3 * n32;
    const uint32_t ctr = (uint32_t)(uintptr_t)(klptr_t)(smem + Cfg::CTR_OFF);
#ifdef RING_AB_CONSTLIMIT                                               // A/B builds only: the limit as a constant instead of a kernel argument
    int spin_limit = RING_SPIN_LIMIT;
#else
    int spin_limit = a.spin_limit;                                        // negative once a wait of this wave has expired (ring_wait_ge)
#endif
    if (tid < 16) reinterpret_cast<uint32_t*>(smem + Cfg::CTR_OFF)[tid] = 0u;
    if (tid >= 64 && tid < 64 + 16 * NMT) {
        const int u0 = (blockIdx.x & 7) * a.units_per_xcd + (blockIdx.x >> 3);
        const int co = (u0 % a.groups) * 16 * NMT + (tid - 64);
        reinterpret_cast<float*>(smem + Cfg::BIAS_OFF)[tid - 64] = (a.bias && co < a.cout) ? a.bias[co] : 0.0f;
    }
    __syncthreads();
    const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
    const int u_end = min((xcd + 1) * a.units_per_xcd, a.n_units);
    const int u_first = xcd * a.units_per_xcd + slot_id;
    if (u_first >= u_end) return;
    const int my_units = (u_end - u_first + a.wgs_per_xcd - 1) / a.wgs_per_xcd;
    const int total = my_units * n_items;
    const int grp0 = u_first % a.groups;

    if (wave >= NC) {
        // =========================================== loader: group lw of every 32-channel block ===========================================
        const int lw = wave - NC;
        const char* zero_blk = reinterpret_cast<const char*>(a.wpack + 4);
        int w_blk[Cfg::NWL], x_piece[Cfg::NXI];
#pragma unroll
        for (int i = 0; i < Cfg::NWL; ++i) w_blk[i] = min(lw * Cfg::NWL + i, Cfg::NBLK - 1);
#pragma unroll
        for (int i = 0; i < Cfg::NXI; ++i) x_piece[i] = min(i * 64, Cfg::PLANE / 16 - 64);
        unsigned long long tab_ptr;
        long long tab_bs;
        {
            const auto* kt = (const __attribute__((address_space(4))) unsigned long long*)__builtin_amdgcn_kernarg_segment_ptr();
            const int l = lane < SPK_MAX_GROUPS ? lane : 0;
            tab_ptr = kt[l];
            tab_bs = (long long)kt[SPK_MAX_GROUPS + l];
        }
        int iss_u = u_first, iss_i = 0, iss_n = 0;
        int gy0[Cfg::NXI], gx0[Cfg::NXI];                                  // window row / column of this lane's slot of piece i (row for dy = 0)
        auto issue_geometry = [&]() {
            const int t = spk_div(iss_u, a.m_groups, a.groups);
            iss_n = spk_div(t, a.m_tiles, a.n_tiles);
            const int tile = t - iss_n * a.n_tiles;
            const int ty = spk_div(tile, a.m_tiles_x, a.tiles_x);
            const int oy0 = ty * SPK_TH, ox0 = (tile - ty * a.tiles_x) * SPK_TW;
#pragma unroll
            for (int i = 0; i < Cfg::NXI; ++i) {
                const int e = x_piece[i] + lane;
                const int y = e / Cfg::IW, x = e - y * Cfg::IW;
                gy0[i] = oy0 - 1 + y; gx0[i] = ox0 - 1 + x;
            }
        };
        const char* const iss_w = reinterpret_cast<const char*>(wrow) + (int64_t)grp0 * n_items * Cfg::W_BYTES;
        issue_geometry();
        int st = 0;
        uint32_t free_target = 0;
        for (int k = 0; k < total; ++k) {
            const int c32 = iss_i / 3, dy = iss_i - 3 * c32;
            const char* wbase = iss_w + (int64_t)iss_i * Cfg::W_BYTES + lane * 16;
            const int gi = c32 * 4 + lw;
            const uint32_t e_lo = __builtin_amdgcn_readlane((int)(uint32_t)tab_ptr, gi), e_hi = __builtin_amdgcn_readlane((int)(uint32_t)(tab_ptr >> 32), gi);
            const uint32_t b_lo = __builtin_amdgcn_readlane((int)(uint32_t)tab_bs, gi), b_hi = __builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)tab_bs >> 32), gi);
            const unsigned long long e = ((unsigned long long)e_hi << 32) | e_lo;
            const long long bs = (long long)(((unsigned long long)b_hi << 32) | b_lo);
            const bool nul = e == 0ull, up2 = (e & 1ull) != 0ull;
            const char* base = reinterpret_cast<const char*>(static_cast<uintptr_t>(e & ~1ull)) + (int64_t)iss_n * bs;
            const long long plane_b = up2 ? (long long)(a.H >> 1) * (a.W >> 1) * 16 : (long long)a.H * a.W * 16;      // hi -> lo plane of the group
            const char* dhi[Cfg::NXI];
            const char* dlo[Cfg::NXI];
#pragma unroll
            for (int i = 0; i < Cfg::NXI; ++i) {
                const int gy = gy0[i] + dy, gx = gx0[i];
                const bool ok = !nul && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                const uint32_t off = up2 ? (uint32_t)((gy >> 1) * (a.W >> 1) + (gx >> 1)) * 16u : (uint32_t)(gy * a.W + gx) * 16u;
                dhi[i] = ok ? base + off : zero_blk;
                dlo[i] = ok ? base + plane_b + off : zero_blk;
            }
            if (++iss_i == n_items) { iss_i = 0; iss_u += a.wgs_per_xcd; if (iss_u < u_end) issue_geometry(); }
            if (free_target) ring_wait_ge(ctr + 32 + 4 * st, free_target, spin_limit);
            if (spin_limit < 0) break;                                    // expired: this loader retires (below), nothing is filled over a slot in use
            unsigned char* stage = smem + st * Cfg::STAGE;
#pragma unroll
            for (int i = 0; i < Cfg::NWL; ++i)
                __builtin_amdgcn_global_load_lds((kgptr_t)(wbase + w_blk[i] * 1024), (klptr_t)(stage + w_blk[i] * 1024), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < Cfg::NXI; ++i) {
                __builtin_amdgcn_global_load_lds((kgptr_t)dhi[i], (klptr_t)(stage + Cfg::W_BYTES + lw * Cfg::PLANE + x_piece[i] * 16), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((kgptr_t)dlo[i], (klptr_t)(stage + Cfg::W_BYTES + (4 + lw) * Cfg::PLANE + x_piece[i] * 16), 16, 0, 0);
            }
            if (k > 0) {
                __builtin_amdgcn_s_waitcnt(0x0F70 | Cfg::K_DMA);
                ring_signal(ctr + 4 * ((st + SLOTS - 1) % SLOTS), lane);
            }
            if (++st == SLOTS) { st = 0; free_target += NC; }
        }
        if (spin_limit < 0) { ring_report_fault(lane); __builtin_amdgcn_s_waitcnt(0x0F70); return; }   // retired on an expired wait: report; its last fills are never announced
        __builtin_amdgcn_s_waitcnt(0x0F70);
        ring_signal(ctr + 4 * ((st + SLOTS - 1) % SLOTS), lane);
        return;
    }

    // ============================================= consumer: tile row cw =============================================
    const int cw = wave;
    const int lj = lane & 15, lg = lane >> 4;
    int boff[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) boff[q] = Cfg::W_BYTES + lg * Cfg::PLANE + (cw * Cfg::IW + q * 16 + lj) * 16;
    f4 acc[NMT][NQ];
#pragma unroll
    for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[m][q] = f4{0.0f, 0.0f, 0.0f, 0.0f};
    int cur_u = u_first, cur_i = 0;
    const float inv_scale = a.wpack[0];
    const int64_t HW = (int64_t)a.H * a.W;
    const uint32_t HW32 = (uint32_t)HW;
    float bias_r[NMT][4];
#pragma unroll
    for (int m = 0; m < NMT; ++m) {
        const f4 bv = *reinterpret_cast<const f4*>(smem + Cfg::BIAS_OFF + (m * 16 + lg * 4) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) bias_r[m][r] = bv[r];
    }
    const float relu_floor = a.relu ? 0.0f : -3.402823466e+38f;
    auto finish_store = [&]() {
        const int t = spk_div(cur_u, a.m_groups, a.groups);
        const int n = spk_div(t, a.m_tiles, a.n_tiles);
        const int tile = t - n * a.n_tiles;
        const int ty = spk_div(tile, a.m_tiles_x, a.tiles_x), tx = tile - ty * a.tiles_x;
        const int oy = ty * SPK_TH + cw;
        unsigned char* spkn = a.out_spk + (int64_t)n * a.out_spk_bstride;
        float abs_sum = 0.0f;
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[m][q][r] = fmaxf(acc[m][q][r] * inv_scale + bias_r[m][r], relu_floor);
                    abs_sum += fabsf(acc[m][q][r]);
                }
        const bool guard = fldr_guard_trips(abs_sum);
        const bool poisoned = spin_limit < 0;
        bool bad = false;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int ox = tx * SPK_TW + q * 16 + lj;
            const bool ok = oy < a.H && ox < a.W;
            const uint32_t pq = ok ? (uint32_t)(oy * a.W + ox) : 0u;
#pragma unroll
            for (int m = 0; m < NMT; ++m) {
                const int co0 = grp0 * 16 * NMT + m * 16 + lg * 4;
                h4 ohi, olo;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[m][q][r];
                    acc[m][q][r] = 0.0f;
                    _Float16 h, l;
                    if (guard) spk_split(v, h, l, bad); else fldr_split_plain(v, h, l);
                    ohi[r] = h; olo[r] = l;
                }
                if (poisoned) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { ohi[r] = (_Float16)__builtin_nanf(""); olo[r] = (_Float16)__builtin_nanf(""); }
                }
                if (ok) {
                    const uint32_t off = ((uint32_t)(co0 >> 3) * 2u * HW32 + pq) * 16u + (uint32_t)(lg & 1) * 8u;
                    *reinterpret_cast<h4*>(spkn + off) = ohi;
                    *reinterpret_cast<h4*>(spkn + (off + HW32 * 16u)) = olo;
                }
            }
        }
        if (guard && !poisoned) fldr_note_range(bad);
    };
    int st_cur = 0, g = 0;
    uint32_t full_target = RING_NLOAD;
    while (g < total) {
        const bool last = cur_i == n_items - 1;
        ring_wait_ge(ctr + 4 * st_cur, full_target, spin_limit);
        {
            const unsigned char* sb = smem + st_cur * Cfg::STAGE;
            const unsigned char* win = sb + lane * 16;
            h8 bh[2][NQ], bl[2][NQ], ah[2][NMT], al[NMT];
            auto ld = [&](int buf, int s) __attribute__((always_inline)) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) bh[buf][q] = *reinterpret_cast<const h8*>(sb + boff[q] + s * 16);
#pragma unroll
                for (int m = 0; m < NMT; ++m) ah[buf][m] = *reinterpret_cast<const h8*>(win + ((s * NMT + m) * 2 + 0) * 1024);
#pragma unroll
                for (int q = 0; q < NQ; ++q) bl[buf][q] = *reinterpret_cast<const h8*>(sb + 4 * Cfg::PLANE + boff[q] + s * 16);
            };
            auto ld_al = [&](int s) __attribute__((always_inline)) {
#pragma unroll
                for (int m = 0; m < NMT; ++m) al[m] = *reinterpret_cast<const h8*>(win + ((s * NMT + m) * 2 + 1) * 1024);
            };
            constexpr int N_MFMA = NQ * 3 * NMT, N_DS = 2 * NQ + 2 * NMT;
            constexpr int N_TAIL = N_MFMA >= 12 ? 4 : 2;
            ld(0, 0);
            ld_al(0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                if (s > 0) ld_al(s);
                if (s + 1 < 3) ld((s + 1) & 1, s + 1);
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int m = 0; m < NMT; ++m)
#pragma unroll
                        for (int q = 0; q < NQ; ++q) {
                            const h8 av = term == 2 ? al[m] : ah[s & 1][m];
                            const h8 bv = term == 1 ? bl[s & 1][q] : bh[s & 1][q];
                            acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc[m][q], 0, 0, 0);
                        }
                spk_step_pattern_n<N_MFMA, N_DS, N_TAIL, 0>(s + 1 < 3);
            }
        }
        if (spin_limit >= 0) ring_signal(ctr + 32 + 4 * st_cur, lane);
        if (last) { finish_store(); cur_i = 0; cur_u += a.wgs_per_xcd; } else ++cur_i;
        if (++st_cur == SLOTS) { st_cur = 0; full_target += RING_NLOAD; }
        ++g;
    }
    if (spin_limit < 0) ring_report_fault(lane);                          // a wait of this consumer expired: its units were written as NaN; report
}

// weight section of the experiment: [group of 16 NMT outputs][32-channel block][dy][dx][m][hi, lo][lane = channel group * 16 + output][8 channels]
__global__ void ringrow_prepack_kernel(const float* __restrict__ w, const float* __restrict__ hdr, float* __restrict__ dst, int cout, int cin, int nmt,
                                       int n32, int64_t total_h8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total_h8) return;
    const float scale = hdr[1];
    const int lane = (int)(i % 64), kind = (int)((i / 64) % 2), m = (int)((i / 128) % nmt), dx = (int)((i / (128 * nmt)) % 3);
    const int dy = (int)((i / (128 * nmt * 3)) % 3), c32 = (int)((i / (128 * nmt * 9)) % n32), gr = (int)(i / ((int64_t)128 * nmt * 9 * n32));
    const int co = gr * 16 * nmt + m * 16 + (lane & 15);
    h8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c32 * 32 + (lane >> 4) * 8 + j;
        const float x = (co < cout && c < cin) ? w[((int64_t)co * cin + c) * 9 + dy * 3 + dx] * scale : 0.0f;
        const _Float16 h = (_Float16)x;
        v[j] = kind == 0 ? h : (_Float16)(x - (float)h);
    }
    reinterpret_cast<h8*>(dst)[i] = v;
}
FLDR_HOOK int64_t fldr_debug_ringrow_pack_floats(int cout, int cin) {
    if (cin % 32 || (cout % 48 && cout % 32)) return FLDR_E_SHAPE;
    const int nmt = cout % 48 == 0 ? 3 : 2;
    return (int64_t)(cout / (16 * nmt)) * (cin / 32) * 9 * nmt * 2 * 256;
}
// wpack: the layer's ordinary pack (header: scale); wrow: fldr_debug_ringrow_pack_floats floats
FLDR_HOOK int fldr_debug_ringrow_prepack(const float* weight, const float* wpack, float* wrow, int cout, int cin, fldr_stream_t stream) {
    const int64_t nf = fldr_debug_ringrow_pack_floats(cout, cin);
    if (nf < 0) return (int)nf;
    const int nmt = cout % 48 == 0 ? 3 : 2;
    hipLaunchKernelGGL(ringrow_prepack_kernel, dim3(fldr_cdiv(nf / 4, 256)), dim3(256), 0, fldr_s(stream), weight, wpack, wrow, cout, cin, nmt, cin / 32, nf / 4);
    FLDR_LAUNCH_RET();
}
template <int NMT>
static int ringrow_launch(SpkArgs& a, const float* wrow, int N, int wgs_per_xcd_max, hipStream_t s) {
    using Cfg = RingRowCfg<NMT>;
    static std::atomic<uint64_t> attr_done{0};
    if (int e = fldr_set_max_lds(reinterpret_cast<const void*>(&conv3x3_ringrow_kernel<NMT>), Cfg::LDS_BYTES, attr_done)) return e;
    a.groups = a.cout / (16 * NMT);
    if (int e = spk_fill_geometry(a, N, wgs_per_xcd_max, SPK_TW)) return e;
    a.spin_limit = g_ring_spin_limit;
    hipLaunchKernelGGL(conv3x3_ringrow_kernel<NMT>, dim3(8 * a.wgs_per_xcd), dim3((8 + RING_NLOAD) * 64), Cfg::LDS_BYTES, s, a, wrow);
    FLDR_LAUNCH_RET();
}
// a: filled by fldr_conv2d_spk's front end (fldr_debug_conv2d_ringrow in conv_spk_kernels.hip)
int fldr_spk_ringrow_dispatch(SpkArgs& a, const float* wrow, int N, int wgs_per_xcd_max, hipStream_t s) {
    if (a.cout % 48 == 0) { a.n_chunks = a.n_chunks / 2; return ringrow_launch<3>(a, wrow, N, wgs_per_xcd_max, s); }
    a.n_chunks = a.n_chunks / 2;
    return ringrow_launch<2>(a, wrow, N, wgs_per_xcd_max, s);
}
#endif  // FLDR_TEST_HOOKS

static int g_ring_consumers = 8;
FLDR_HOOK int fldr_debug_ring_consumers(int v) { if (v == 4 || v == 8) g_ring_consumers = v; return g_ring_consumers; }

template <int NMT, int TERMS, bool HAS_RES, int NC, int TW, bool RW = false>
static int ring_launch3(SpkArgs& a, int N, int wgs_per_xcd_max, hipStream_t s) {
    using Cfg = RingCfg<NMT, TW, RW>;
    static std::atomic<uint64_t> attr_done{0};
    if (int e = fldr_set_max_lds(reinterpret_cast<const void*>(&conv3x3_ring_kernel<NMT, TERMS, HAS_RES, NC, TW, false, RW>), Cfg::LDS_BYTES, attr_done)) return e;
    if (int e = spk_fill_geometry(a, N, wgs_per_xcd_max, TW)) return e;
    a.spin_limit = g_ring_spin_limit;
    hipLaunchKernelGGL((conv3x3_ring_kernel<NMT, TERMS, HAS_RES, NC, TW, false, RW>), dim3(8 * a.wgs_per_xcd), dim3((NC + RING_NLOAD) * 64), Cfg::LDS_BYTES, s, a);
    FLDR_LAUNCH_RET();
}

// The resident-weight ring is bit-identical to the streamed-weight ring and measured EQUAL (48 -> 16 at 1152x1920: 134.1 vs 134.0 us with 8
// consumer waves, 131.0 vs 134.2 with 4; 48 -> 4 at 288x480: 13.0 vs 12.4; bench 484.6 vs 487.3 pairs/s): these launches are bound by their
// consumers (stamps at 1152x1920: the loader waits 1,300 of 2,200 cycles per fill for a FREE slot; a consumer spends 1,330 cycles per
// iteration in its steps on an LDS array that the pixel-operand reads alone fill, and 2,300 per unit in the epilogue).  Test build only.
#ifndef RING_RW_DEFAULT
#define RING_RW_DEFAULT 0
#endif
#ifdef FLDR_TEST_HOOKS
static int g_ring_resident = RING_RW_DEFAULT;
FLDR_HOOK int fldr_debug_ring_resident(int v) { if (v == 0 || v == 1) g_ring_resident = v; return g_ring_resident; }
#endif

// Tile width of a launch.  Persistent workgroups (one per CU) walk the units in rounds; a launch whose 8 x 32 tiles leave the
// last round nearly empty (the second pyramid level of a 4K pair: 288 units on 256 workgroups = two rounds for 1.13 rounds
// of work) runs 8 x 16 tiles instead when that is cheaper: twice the units, each costing RING_NARROW_COST of a wide one (half
// the matrix work under the same weight stream).  Same arithmetic per output pixel: bit-identical results.
#ifndef RING_NARROW_COST
#define RING_NARROW_COST 0.65                  // measured: 9.5 vs 14.7 us per round of 256 units (96 -> 96)
#endif
static int g_ring_tile_width = 0;                // 0: automatic; 16 / 32: forced
FLDR_HOOK int fldr_debug_ring_tile_width(int v) { if (v == 0 || v == 16 || v == 32) g_ring_tile_width = v; return g_ring_tile_width; }

static int ring_pick_tile_width(const SpkArgs& a, int N, int wgs_per_xcd_max) {
    if (g_ring_tile_width) return g_ring_tile_width;
    const int64_t wgs = 8ll * wgs_per_xcd_max;
    const int64_t ty = fldr_cdiv(a.H, SPK_TH);
    const int64_t u32 = (int64_t)N * fldr_cdiv(a.W, 32) * ty * a.groups, u16 = (int64_t)N * fldr_cdiv(a.W, 16) * ty * a.groups;
    if (u32 <= wgs / 2) return 32;                                       // launches that do not fill the chip either way
    const double c32 = (double)((u32 + wgs - 1) / wgs), c16 = (double)((u16 + wgs - 1) / wgs) * RING_NARROW_COST;
    return c16 < c32 ? 16 : 32;
}

template <int NMT, int TERMS, bool HAS_RES>
static int ring_launch2(SpkArgs& a, int N, int wpx, hipStream_t s) {
#ifdef FLDR_TEST_HOOKS
    if constexpr (NMT == 1 && TERMS == 3) {
        if (g_ring_resident && a.n_chunks <= RING_RW_MAX_CHUNKS && g_ring_tile_width != 16) {
            const int64_t units = (int64_t)N * fldr_cdiv(a.W, SPK_TW) * fldr_cdiv(a.H, SPK_TH) * a.groups;
            if (g_ring_consumers == 4 || (g_ring_tile_width == 0 && units >= 32ll * wpx)) return ring_launch3<NMT, TERMS, HAS_RES, 4, 32, true>(a, N, wpx, s);
            return ring_launch3<NMT, TERMS, HAS_RES, 8, 32, true>(a, N, wpx, s);
        }
    }
#endif
    if (g_ring_consumers == 4) return ring_launch3<NMT, TERMS, HAS_RES, 4, 32>(a, N, wpx, s);
    if constexpr (NMT == 1 && TERMS == 3) {
        // 16 output channels at a large resolution (dec2: 48 -> 16 at half the frame size): with one 16-channel block every MFMA
        // needs one LDS operand read when a wave owns one tile row (8 consumers), 0.83 when it owns two (4 consumers), and the
        // LDS array, not the matrix pipe, paces the kernel: 170.7 vs 151.1 us at 1152x1920 (tools/kernel_bench.py conv).
        const int64_t units = (int64_t)N * fldr_cdiv(a.W, SPK_TW) * fldr_cdiv(a.H, SPK_TH) * a.groups;
        if (g_ring_consumers == 8 && g_ring_tile_width == 0 && units >= 32ll * wpx) return ring_launch3<NMT, TERMS, HAS_RES, 4, 32>(a, N, wpx, s);
    }
    if constexpr (TERMS == 3) {
        if (ring_pick_tile_width(a, N, wpx) == 16) return ring_launch3<NMT, TERMS, HAS_RES, 8, 16>(a, N, wpx, s);
    }
    return ring_launch3<NMT, TERMS, HAS_RES, 8, 32>(a, N, wpx, s);
}

template <int NMT, int TERMS>
static int ring_launch(SpkArgs& a, int N, int wpx, hipStream_t s) {
    return a.residual ? ring_launch2<NMT, TERMS, true>(a, N, wpx, s) : ring_launch2<NMT, TERMS, false>(a, N, wpx, s);
}

// The 32x32x16 kernel where it applies: 64 or 96 output channels all stored, packed output only, no residual, launches that are not
// run as 16-channel sub-groups.
#ifndef RING32_DEFAULT
#define RING32_DEFAULT 1
#endif
static int g_ring32 = RING32_DEFAULT;
FLDR_HOOK int fldr_debug_ring32(int v) { if (v >= 0 && v <= 2) g_ring32 = v; return g_ring32; }
static int ring32_launch(SpkArgs& a, int N, int wgs_per_xcd_max, hipStream_t s) {
    static std::atomic<uint64_t> attr_done{0};
    if (int e = fldr_set_max_lds(reinterpret_cast<const void*>(&conv3x3_ring32_kernel), Ring32Cfg::LDS_BYTES, attr_done)) return e;
    a.groups = a.cout / 32;
    if (int e = spk_fill_geometry(a, N, wgs_per_xcd_max, SPK_TW)) return e;
    a.spin_limit = g_ring_spin_limit;
    hipLaunchKernelGGL(conv3x3_ring32_kernel, dim3(8 * a.wgs_per_xcd), dim3((8 + RING_NLOAD) * 64), Ring32Cfg::LDS_BYTES, s, a);
    FLDR_LAUNCH_RET();
}

// Measured (tools/kernel_bench.py conv, 96 -> 96): a round of 32-channel units costs ~0.8 of a round of 48-channel units (11.1-11.4 vs
// 12.6-14.2 us) for two thirds of the work — the kernel's 36 operand reads per 27 MFMAs keep the LDS array ~75 % busy — so it is NOT
// the faster kernel per unit of work (288x480: 78.0 vs 71.1 us).  It wins where its finer units fill the rounds of persistent workgroups
// better: the second pyramid level of a 4K pair, 144x240 = 405 units instead of 270 (or 540 narrow ones): 22.8 vs 25.2 us.
#ifndef RING32_ROUND_COST
#define RING32_ROUND_COST 0.8
#endif
static bool ring32_pays(const SpkArgs& a, int N, int wgs_per_xcd_max) {
    if (g_ring32 == 2) return true;                                       // forced (tests, kernel_bench)
    if (a.cout != 96) return false;                                      // (64 outputs: the 16x16x32 kernel's units are 32 channels already — 35.2 vs 36.5 us at 288x480)
    const int64_t wgs = 8ll * wgs_per_xcd_max, tx32 = fldr_cdiv(a.W, 32), ty = fldr_cdiv(a.H, SPK_TH);
    const int64_t u48 = (int64_t)N * tx32 * ty * a.groups, u16 = (int64_t)N * fldr_cdiv(a.W, 16) * ty * a.groups, u32 = (int64_t)N * tx32 * ty * (a.cout / 32);
    if (u32 <= wgs) return false;                                        // launches that do not fill the chip either way
    const double c48 = (double)((u48 + wgs - 1) / wgs), c16 = (double)((u16 + wgs - 1) / wgs) * RING_NARROW_COST;
    const double c32 = (double)((u32 + wgs - 1) / wgs) * RING32_ROUND_COST;
    return c32 < (c48 < c16 ? c48 : c16);
}

int fldr_spk_ring_dispatch(SpkArgs& a, int N, int nmt, int terms, int wgs_per_xcd_max, hipStream_t s) {
    if (g_ring32 && terms == 3 && nmt >= 2 && a.w32_off && a.cout_store == a.cout && !a.residual && !a.out_f32 && a.out_spk && g_ring_consumers == 8 &&
        g_ring_tile_width == 0 && ring32_pays(a, N, wgs_per_xcd_max))
        return ring32_launch(a, N, wgs_per_xcd_max, s);
    if (terms == 1) {
        if (nmt == 1) return ring_launch<1, 1>(a, N, wgs_per_xcd_max, s);
        if (nmt == 2) return ring_launch<2, 1>(a, N, wgs_per_xcd_max, s);
        return ring_launch<3, 1>(a, N, wgs_per_xcd_max, s);
    }
    if (nmt == 1) return ring_launch<1, 3>(a, N, wgs_per_xcd_max, s);
    if (nmt == 2) return ring_launch<2, 3>(a, N, wgs_per_xcd_max, s);
    return ring_launch<3, 3>(a, N, wgs_per_xcd_max, s);
}

// Multi-level launch: 8 x 32 tiles, 8 consumer waves; the unit geometry of spk_fill_geometry with the units of all levels.
template <int NMT, int TERMS, bool HAS_RES>
static int ring_launch_levels(SpkArgs& a, int n_units, int wgs_per_xcd_max, hipStream_t s) {
    using Cfg = RingCfg<NMT, 32>;
    static std::atomic<uint64_t> attr_done{0};
    if (int e = fldr_set_max_lds(reinterpret_cast<const void*>(&conv3x3_ring_kernel<NMT, TERMS, HAS_RES, 8, 32, true>), Cfg::LDS_BYTES, attr_done)) return e;
    a.tiles_x = 1; a.n_tiles = 1; a.m_tiles = 0; a.m_tiles_x = 0;           // (per level: a.lv)
    a.n_units = n_units;
    a.units_per_xcd = (a.n_units + 7) / 8;
    a.units_per_xcd = (a.units_per_xcd + a.groups - 1) / a.groups * a.groups;
    a.wgs_per_xcd = spk_right_size(a.units_per_xcd, wgs_per_xcd_max, a.groups);
    if (((int64_t)a.n_units + 8 * a.units_per_xcd) * a.groups >= (1ll << 32)) return FLDR_E_SHAPE;
    a.m_groups = (uint32_t)((1ull << 32) / (uint32_t)a.groups) + 1u;
    a.spin_limit = g_ring_spin_limit;
    hipLaunchKernelGGL((conv3x3_ring_kernel<NMT, TERMS, HAS_RES, 8, 32, true>), dim3(8 * a.wgs_per_xcd), dim3((8 + RING_NLOAD) * 64), Cfg::LDS_BYTES, s, a);
    FLDR_LAUNCH_RET();
}

int fldr_spk_ring_dispatch_levels(SpkArgs& a, int n_units, int nmt, int terms, int wgs_per_xcd_max, hipStream_t s) {
#define RING_LV(NMT_, T_) (a.residual ? ring_launch_levels<NMT_, T_, true>(a, n_units, wgs_per_xcd_max, s) : ring_launch_levels<NMT_, T_, false>(a, n_units, wgs_per_xcd_max, s))
    if (terms == 1) return nmt == 1 ? RING_LV(1, 1) : (nmt == 2 ? RING_LV(2, 1) : RING_LV(3, 1));
    return nmt == 1 ? RING_LV(1, 3) : (nmt == 2 ? RING_LV(2, 3) : RING_LV(3, 3));
#undef RING_LV
}
