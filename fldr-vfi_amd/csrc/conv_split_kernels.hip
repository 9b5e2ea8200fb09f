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
    const float* src[FLDR_CONV_MAX_SRC];
    int64_t src_bstride[FLDR_CONV_MAX_SRC];
    int32_t src_cbegin[FLDR_CONV_MAX_SRC + 1];
    int32_t src_up2[FLDR_CONV_MAX_SRC];
    int32_t n_src;
    const float* wpack;        // {header, halves...}
    const float* bias;
    const float* residual;
    float* out;
    int32_t cin, cout, cout_store;
    int32_t H, W;
    int32_t relu;
    int32_t tiles_x;
    int32_t groups;
};

template <int NMT>
struct SplitCfg {
    static constexpr int W_BYTES = SP_STEPS * NMT * 2 * 1024;           // one chunk of one group, hi + lo
    static constexpr int PIECES = W_BYTES / 16;                         // 16-B LDS-DMA pieces (a multiple of 64)
    static constexpr int NWI = (PIECES + 511) / 512;                    // sweeps of the 512-thread workgroup
    static constexpr int STAGE_BYTES = W_BYTES + SP_IN_BYTES;
    static constexpr int TAB_BYTES = SP_MAX_CIN * 8;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES + TAB_BYTES;
    static_assert(PIECES % 64 == 0, "weight slab must be a whole number of wave-wide DMA pieces");
};

__device__ __forceinline__ void split8(const float (&x)[8], h8& hi, h8& lo) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const _Float16 h = (_Float16)x[k];
        hi[k] = h;
        lo[k] = (_Float16)(x[k] - (float)h);
    }
}

template <int NMT>
__global__ __launch_bounds__(512, 2) void conv3x3_split_kernel(SplitArgs a) {
    using Cfg = SplitCfg<NMT>;
    constexpr int MTOT = 16 * NMT;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n = blockIdx.y;
    const int grp = blockIdx.x % a.groups, tile = blockIdx.x / a.groups;
    const int cbase = grp * MTOT;
    const int oy0 = (tile / a.tiles_x) * SP_TH, ox0 = (tile % a.tiles_x) * SP_TW;
    const int lj = lane & 15, lg = lane >> 4;
    const int cin_pad = (a.cin + 15) / 16 * 16;
    const int n_chunks = cin_pad / 16;

    // channel table (plane pointer | half-resolution flag), built once
    unsigned long long* ctab = reinterpret_cast<unsigned long long*>(smem + 2 * Cfg::STAGE_BYTES);
    if (tid < SP_MAX_CIN) {
        unsigned long long e = 0;
        if (tid < a.cin) {
            int s = 0;
            while (s + 1 < a.n_src && tid >= a.src_cbegin[s + 1]) ++s;
            const int up2 = a.src_up2[s];
            const int64_t plane = up2 ? (int64_t)(a.H >> 1) * (a.W >> 1) : (int64_t)a.H * a.W;
            const float* base = a.src[s] + (int64_t)n * a.src_bstride[s] + (int64_t)(tid - a.src_cbegin[s]) * plane;
            e = (unsigned long long)reinterpret_cast<uintptr_t>(base) | (unsigned long long)(up2 ? 1 : 0);
        }
        ctab[tid] = e;
    }

    // staging geometry: waves 0-3 stage channels 0-7 of a chunk, waves 4-7 channels 8-15; 340 pixels over 256 threads
    const int scg = wave >> 2;                         // wave-uniform 8-channel plane
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

    auto issue_weights = [&](int chunk, unsigned char* stage) {
        const float* g = wsrc + (int64_t)chunk * (Cfg::W_BYTES / 4);
#pragma unroll
        for (int i = 0; i < Cfg::NWI; ++i) {
            const int piece = i * 512 + wave * 64;                                    // wave-uniform, x16 bytes
            if (piece < Cfg::PIECES)
                __builtin_amdgcn_global_load_lds((sgptr_t)(g + (piece + lane) * 4), (slptr_t)(stage + piece * 16), 16, 0, 0);
        }
    };
    auto load_inputs = [&](int chunk) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned long long e = ctab[chunk * 16 + scg * 8 + k];
            const bool live = e != 0ull;
            const bool up2 = (e & 1ull) != 0ull;
            const float* base = reinterpret_cast<const float*>(static_cast<uintptr_t>(e & ~1ull));
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float v = 0.0f;
                if (live && s_ok[i]) v = base[up2 ? g_half[i] : g_full[i]];
                pre[i][k] = v;
            }
        }
    };
    auto store_inputs = [&](unsigned char* stage) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (l_off[i] < 0) continue;
            h8 hi, lo;
            split8(pre[i], hi, lo);
            *reinterpret_cast<h8*>(stage + Cfg::W_BYTES + l_off[i]) = hi;
            *reinterpret_cast<h8*>(stage + Cfg::W_BYTES + 2 * SP_CG_STRIDE + l_off[i]) = lo;
        }
    };

    __syncthreads();                                   // channel table visible
    issue_weights(0, smem);
    load_inputs(0);
    store_inputs(smem);
    __syncthreads();

    for (int ch = 0; ch < n_chunks; ++ch) {
        unsigned char* cur = smem + (ch & 1) * Cfg::STAGE_BYTES;
        unsigned char* nxt = smem + ((ch + 1) & 1) * Cfg::STAGE_BYTES;
        const bool more = ch + 1 < n_chunks;
        const unsigned char* win = cur + lane * 16;
        const unsigned char* xin = cur + Cfg::W_BYTES;
#pragma unroll
        for (int s = 0; s < SP_STEPS; ++s) {
            if (s == 0 && more) issue_weights(ch + 1, nxt);
            if (s == 1 && more) load_inputs(ch + 1);
            if (s == SP_STEPS - 1 && more) store_inputs(nxt);
            // taps of this step: tA = 2s, tB = 2s+1 (tB = 9 is the zero-weight pad tap: read tap 8's pixels again)
            const int tA = 2 * s, tB = 2 * s + 1 < 9 ? 2 * s + 1 : 8;
            const int offA = ((tA / 3) * SP_IW + tA % 3) * 16, offB = ((tB / 3) * SP_IW + tB % 3) * 16;
            const int toff = tap_sel ? offB : offA;
            h8 bh[2], bl[2], ah[NMT], al[NMT];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                bh[p] = *reinterpret_cast<const h8*>(xin + boff[p] + toff);
                bl[p] = *reinterpret_cast<const h8*>(xin + 2 * SP_CG_STRIDE + boff[p] + toff);
            }
#pragma unroll
            for (int m = 0; m < NMT; ++m) {
                ah[m] = *reinterpret_cast<const h8*>(win + ((s * NMT + m) * 2 + 0) * 1024);
                al[m] = *reinterpret_cast<const h8*>(win + ((s * NMT + m) * 2 + 1) * 1024);
            }
#pragma unroll
            for (int m = 0; m < NMT; ++m)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    acc[m][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh[p], acc[m][p], 0, 0, 0);
                    acc[m][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl[p], acc[m][p], 0, 0, 0);
                    acc[m][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh[p], acc[m][p], 0, 0, 0);
                }
        }
        __syncthreads();
    }

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

template <int NMT>
static int split_launch(const SplitArgs& a, int N, hipStream_t s) {
    using Cfg = SplitCfg<NMT>;
    SplitArgs b = a;
    b.tiles_x = fldr_cdiv(a.W, SP_TW);
    const int tiles_y = fldr_cdiv(a.H, SP_TH);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_split_kernel<NMT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL((conv3x3_split_kernel<NMT>), dim3(b.tiles_x * tiles_y * b.groups, N), dim3(512), Cfg::LDS_BYTES, s, b);
    FLDR_LAUNCH_RET();
}

// Same descriptor as fldr_conv2d; only ksize 3 / stride 1; d->wpack must come from fldr_conv_split_prepack.
extern "C" int fldr_conv2d_split(const fldr_conv_desc* d, fldr_stream_t stream) {
    FLDR_CHECK_ARG(d && d->wpack && d->out && d->n_src >= 1 && d->n_src <= FLDR_CONV_MAX_SRC);
    FLDR_CHECK_ARG(d->N > 0 && d->cin > 0 && d->cin <= SP_MAX_CIN && d->cout > 0 && d->cout <= 96);
    FLDR_CHECK_ARG(d->cout_store > 0 && d->cout_store <= d->cout && d->ksize == 3 && d->stride == 1);
    if (d->Hout != d->Hin || d->Wout != d->Win) return FLDR_E_SHAPE;
    SplitArgs a;
    int csum = 0;
    for (int s = 0; s < FLDR_CONV_MAX_SRC; ++s) {
        const bool live = s < d->n_src;
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
    a.cin = d->cin; a.cout = d->cout; a.cout_store = d->cout_store;
    a.H = d->Hin; a.W = d->Win; a.relu = d->relu; a.tiles_x = 0;
    int nmt, groups;
    split_geometry(d->cout, nmt, groups);
    a.groups = groups;
    hipStream_t s = fldr_s(stream);
    if (nmt == 1) return split_launch<1>(a, d->N, s);
    if (nmt == 2) return split_launch<2>(a, d->N, s);
    return split_launch<3>(a, d->N, s);
}
