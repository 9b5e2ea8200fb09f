// Definitions shared by the split-packed 3x3 convolution kernels (conv_spk_kernels.hip: barrier pipeline;
// conv_ring_kernels.hip: loader / consumer ring): argument block, LDS stage geometry, the hi/lo split and the MFMA /
// LDS-read interleave patterns.
#pragma once
#include "common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* kgptr_t;
typedef __attribute__((address_space(3))) void* klptr_t;

#define SPK_MAX_GROUPS 14                       // 112 input channels
#define SPK_MAX_LEVELS 8                        // pyramid levels of a multi-level launch
#define SPK_TH 8
#define SPK_TW 32
#define SPK_IH (SPK_TH + 2)
#define SPK_IW (SPK_TW + 2)
#define SPK_PLANE 5632                          // bytes per LDS plane: 340 px * 16 B rounded up to a multiple of 256
#define SPK_IN_BYTES (4 * SPK_PLANE)
#define SPK_STEPS 5                             // tap pairs per 16-channel chunk (9 taps + 1 zero tap)
#define SPK_HDR 8                               // floats before the packed weights: {1/scale, scale, max|w|, 0, 0,0,0,0}

struct SpkArgs {
    unsigned long long grp_ptr[SPK_MAX_GROUPS];   // hi plane of input group g, sample 0; bit 0 = stored at half resolution (nearest x2 read); 0 = padding group
    int64_t grp_bstride[SPK_MAX_GROUPS];          // bytes between samples
    const float* wpack;
    const float* bias;
    const float* residual;
    float* out_f32;                               // [N, cout_store, H, W] or null
    unsigned char* out_spk;                       // SPK tensor of cout_store channels or null
    int64_t out_spk_bstride;                      // bytes between samples
    int32_t n_chunks, cout, cout_store;
    int32_t H, W;
    int32_t relu;
    int32_t res_spk;                              // 1: `residual` is a split-packed tensor of cout_store channels (value = hi + lo), ring pipeline only
    int32_t tiles_x, n_tiles, groups;             // groups: output-channel groups of 16*NMT channels the launch is split into
    int32_t pack_nmt;                             // 16-channel blocks per weight-pack group (>= NMT, a multiple of it: small
                                                  // launches run the NMT=1 kernel on sub-groups of an NMT=3 pack)
    int32_t n_units, units_per_xcd, wgs_per_xcd;
    // Multi-level launch (ring kernel only; n_levels > 1): ONE sample and ONE source tensor per level, the same weights for all
    // (rec_ctx_ds over the six pyramid levels, fLDRnet.py:148-162).  Units [lv[l].unit0, lv[l+1].unit0) belong to level l; its
    // tensors lie at byte offsets from level 0's (grp_ptr[0] / out_spk / out_f32 / residual).  H, W, tiles_x, n_tiles, m_tiles,
    // m_tiles_x above are unused then.
    int32_t n_levels;
    int32_t spin_limit;                           // ring kernels: polls a bounded wait makes before it expires (conv_ring_kernels.hip sets it at launch).
                                                  // HERE (the padding in front of lv[], next to the scalars every prologue reads): at the end of the
                                                  // struct it was one more kernel-argument cache line, fetched when the first wait needed it
    struct SpkLevel {
        int32_t H, W, tiles_x, n_tiles, unit0, pad;
        uint32_t m_tiles_x, pad2;
        int64_t in_off, out_spk_off, out_f32_off, res_off;
    } lv[SPK_MAX_LEVELS];
    uint32_t m_groups, m_tiles, m_tiles_x;        // floor(2^32 / d) + 1: u / d == umulhi(u, m) for u * d < 2^32 (d > 1)
    int64_t w32_off;                              // conv3x3_ring32_kernel: byte offset of section R32 of the weight pack from wpack
};

// Output-channel geometry of the 16x16x32 kernels: 16-channel blocks per group (NMT), groups per launch.
static inline void spk_geometry(int cout, int& nmt, int& groups) {
    if (cout <= 16)      { nmt = 1; groups = 1; }
    else if (cout <= 32) { nmt = 2; groups = 1; }
    else if (cout <= 48) { nmt = 3; groups = 1; }
    else if (cout <= 64) { nmt = 2; groups = 2; }
    else                 { nmt = 3; groups = (cout + 47) / 48; }
}
// Weight pack: header, first section ([group][chunk][step][m][kind] 1-KB blocks of the 16x16x32 kernels), and — for 64 or 96 output
// channels — section R32 ([group of 32][chunk][tap][kind] 1-KB blocks of conv3x3_ring32_kernel).  Sizes in floats.
static inline int64_t spk_first_section_floats(int cout, int cin) {
    int nmt, groups;
    spk_geometry(cout, nmt, groups);
    return (int64_t)groups * ((cin + 15) / 16) * SPK_STEPS * nmt * 2 * 64 * 4;
}
static inline bool spk_has_r32_section(int cout) { return cout == 64 || cout == 96; }
static inline int64_t spk_r32_section_floats(int cout, int cin) { return spk_has_r32_section(cout) ? (int64_t)(cout / 32) * ((cin + 15) / 16) * 18 * 256 : 0; }

template <int NMT>
struct SpkCfg {
    static constexpr int W_BYTES = SPK_STEPS * NMT * 2 * 1024;          // one chunk of one output group, hi + lo
    static constexpr int PIECES = W_BYTES / 16;
    static constexpr int NWI = (PIECES + 511) / 512;
    static constexpr int STAGE = W_BYTES + SPK_IN_BYTES;
    static constexpr int LDS_BYTES = 3 * STAGE;
    static constexpr int K_MIN = 3 + NWI;                               // DMA instructions every wave issues per iteration
    static_assert(PIECES % 64 == 0, "weight slab must be a whole number of wave-wide DMA pieces");
};

__device__ __forceinline__ int spk_div(int u, uint32_t m, int d) {     // exact for 0 <= u, u * d < 2^32 (host-checked)
    return d == 1 ? u : (int)__umulhi((uint32_t)u, m);
}

__device__ __forceinline__ void spk_split(float x, _Float16& hi, _Float16& lo, bool& bad) { fldr_split_hl(x, hi, lo, bad); }

// Scheduling pattern of one MFMA step: N_DS groups of {a share of the N_MFMA matrix instructions, one LDS read}.
template <int N_MFMA, int N_DS, int I>
struct SpkInterleave {
    static __device__ __forceinline__ void run() {
        constexpr int cnt = (N_MFMA * (I + 1)) / N_DS - (N_MFMA * I) / N_DS;
        if constexpr (cnt > 0) __builtin_amdgcn_sched_group_barrier(0x008, cnt, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        SpkInterleave<N_MFMA, N_DS, I + 1>::run();
    }
};
template <int NV, int I>
struct SpkInterleaveV {                                             // {1 MFMA, 1 vector-memory instruction} x NV
    static __device__ __forceinline__ void run() {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        SpkInterleaveV<NV, I + 1>::run();
    }
};
template <int NV>
struct SpkInterleaveV<NV, NV> {
    static __device__ __forceinline__ void run() {}
};
template <int N_MFMA, int N_DS>
struct SpkInterleave<N_MFMA, N_DS, N_DS> {
    static __device__ __forceinline__ void run() {}
};

// One MFMA step: {1 MFMA, 1 DMA} per DMA instruction of the step, then the next step's LDS reads spread over all but
// the last N_TAIL MFMAs (which cover the latency of the last read: the wait in front of the next step is an
// lgkmcnt(0)); the last step has no reads.
template <int N_MFMA, int N_DS, int N_TAIL, int NV>
__device__ __forceinline__ void spk_step_pattern_n(bool reads) {
    SpkInterleaveV<NV, 0>::run();
    constexpr int rest = N_MFMA - NV;
    if (reads) {
        constexpr int tail = rest - N_TAIL >= N_DS / 2 ? N_TAIL : 0;
        SpkInterleave<rest - tail, N_DS, 0>::run();
        if constexpr (tail > 0) __builtin_amdgcn_sched_group_barrier(0x008, tail, 0);
    } else {
        if constexpr (rest > 0) __builtin_amdgcn_sched_group_barrier(0x008, rest, 0);
    }
}
template <int N_MFMA, int N_DS, int N_TAIL, int K_DMA>
__device__ __forceinline__ void spk_step_pattern(int s) {            // s is a constant after unrolling
    const int nv = ((s + 1) * K_DMA + SPK_STEPS - 1) / SPK_STEPS - (s * K_DMA + SPK_STEPS - 1) / SPK_STEPS;
    const bool reads = s + 1 < SPK_STEPS;
    if (nv == 0) spk_step_pattern_n<N_MFMA, N_DS, N_TAIL, 0>(reads);
    else if (nv == 1) spk_step_pattern_n<N_MFMA, N_DS, N_TAIL, 1>(reads);
    else spk_step_pattern_n<N_MFMA, N_DS, N_TAIL, 2>(reads);
}

// Persistent workgroups per XCD for `upx` units per XCD on at most `wpx_max` workgroups, a multiple of `groups` (one output group per
// workgroup: bias kept in registers).  RIGHT-SIZED (round 4): the launch takes R = ceil(upx / wpx_max) rounds whatever the grid, so it
// only starts the ceil(upx / R) workgroups that keep every workgroup busy for all R rounds — the 96 -> 96 convolution at 288x480 has
// 136 units per XCD = 4.25 rounds of 32: 28 workgroups of <= 5 units finish when 32 workgroups (8 with 5 units, 24 with 4) would, and the
// 32 CUs of the chip it leaves alone run the other streams' kernels meanwhile (a workgroup occupies its CU's whole LDS).
#ifndef SPK_RIGHT_SIZE
#define SPK_RIGHT_SIZE 1
#endif
static inline int spk_right_size(int upx, int wpx_max, int groups) {
    int w = upx < wpx_max ? upx : wpx_max;
    w = w / groups * groups;
    if (w < groups) w = groups;
#if SPK_RIGHT_SIZE
    const int rounds = (upx + w - 1) / w;
    int need = (upx + rounds - 1) / rounds;
    need = (need + groups - 1) / groups * groups;
    if (need < w) w = need;
#endif
    return w;
}

// Launch geometry shared by both pipelines: tiles, (sample, tile, group) units, XCD-contiguous unit ranges, persistent
// workgroups per XCD (a multiple of `groups`: one output group per workgroup) and the magic numbers of spk_div.
static inline int spk_fill_geometry(SpkArgs& a, int N, int wgs_per_xcd_max, int tile_w = SPK_TW) {
    a.tiles_x = fldr_cdiv(a.W, tile_w);
    a.n_tiles = a.tiles_x * fldr_cdiv(a.H, SPK_TH);
    a.n_units = N * a.n_tiles * a.groups;
    a.units_per_xcd = (a.n_units + 7) / 8;
    a.units_per_xcd = (a.units_per_xcd + a.groups - 1) / a.groups * a.groups;      // whole tiles per XCD
    a.wgs_per_xcd = spk_right_size(a.units_per_xcd, wgs_per_xcd_max, a.groups);
    const int64_t dmax = a.n_tiles > a.groups ? a.n_tiles : a.groups;
    if (((int64_t)a.n_units + 8 * a.units_per_xcd) * dmax >= (1ll << 32)) return FLDR_E_SHAPE;      // exactness of spk_div
    a.m_groups = (uint32_t)((1ull << 32) / (uint32_t)a.groups) + 1u;
    a.m_tiles = (uint32_t)((1ull << 32) / (uint32_t)a.n_tiles) + 1u;
    a.m_tiles_x = (uint32_t)((1ull << 32) / (uint32_t)a.tiles_x) + 1u;
    return 0;
}

// conv_ring_kernels.hip: the loader / consumer ring pipeline (nmt in {1,2,3}, terms in {1,3})
int fldr_spk_ring_dispatch(SpkArgs& a, int N, int nmt, int terms, int wgs_per_xcd_max, hipStream_t s);
// ... for a multi-level launch (a.n_levels > 1, a.lv filled, n_units = the sum over the levels)
int fldr_spk_ring_dispatch_levels(SpkArgs& a, int n_units, int nmt, int terms, int wgs_per_xcd_max, hipStream_t s);
