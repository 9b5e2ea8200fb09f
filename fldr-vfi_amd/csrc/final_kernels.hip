// Fused end of the network: PCARefineUNet.dec3 on the nearest-x2 upsampled dec2 output (fLDRnet.py:642-643)
// + occlusion softmax / T + 6-way blend in fp64 (fLDRnet.py:511-524), one kernel, refine_out never stored.
//
// Nearest x2 upsampling followed by a 3x3 / pad 1 convolution is, for each of the four output phases (a,b) of
// a low-resolution pixel (i,j), a 2x2 convolution over low-res rows {i-1+a, i+a} and columns {j-1+b, j+b} whose
// weights are sums of the 3x3 taps that fall on the same low-res pixel (rows: a=0 -> {w0, w1+w2}, a=1 ->
// {w0+w1, w2}; same for columns).  Zero padding of the upsampled image equals zero padding of the low-res one.
// That is 4 taps instead of 9 (2.25x fewer MACs) and no redundant reads.  With 16 -> 6 channels this layer
// is far too thin for the matrix cores (6 of 16 MFMA rows would be useful), so it runs on the vector ALUs with
// the phase weights as scalar (SGPR) operands: one thread = one low-res pixel = 2x2 output pixels x 6 logits,
// the 8x32 low-res tile (+halo) of all 16 input channels staged once in LDS.  Measured at 2304x3840 (cold HBM): 311 us,
// 222 us without the convolution, 235 us without the candidate loads — co-limited by the vector ALUs (768 packed fp32
// FMAs + ~670 fp64 operations per thread, ~600 of them the six fp64 exponentials of each pixel) and the 991 MB it
// moves (18 candidate planes in, the fp64 frame out).
#include "common.h"

#define D3_CIN 16
#define D3_COUT 6
#ifndef D3_TH
#define D3_TH 8
#define D3_TW 32
#endif

// weff[c][phase][co][tap], tap = dy2*2+dx2  (1536 floats)
__global__ void dec3_prepack_kernel(const float* __restrict__ w, float* __restrict__ weff) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= D3_CIN * 4 * D3_COUT * 4) return;
    const int tap = i & 3, co = (i >> 2) % D3_COUT, ph = (i / (4 * D3_COUT)) & 3, c = i / (16 * D3_COUT);
    const int a = ph >> 1, b = ph & 1, dy2 = tap >> 1, dx2 = tap & 1;
    // 3x3 rows covered by low-res row (a + dy2 - ... ): a=0: dy2=0 -> {0}, dy2=1 -> {1,2};  a=1: dy2=0 -> {0,1}, dy2=1 -> {2}
    const int r0 = a == 0 ? (dy2 == 0 ? 0 : 1) : (dy2 == 0 ? 0 : 2), r1 = a == 0 ? (dy2 == 0 ? 0 : 2) : (dy2 == 0 ? 1 : 2);
    const int c0 = b == 0 ? (dx2 == 0 ? 0 : 1) : (dx2 == 0 ? 0 : 2), c1 = b == 0 ? (dx2 == 0 ? 0 : 2) : (dx2 == 0 ? 1 : 2);
    const float* wk = w + ((int64_t)co * D3_CIN + c) * 9;
    float s = 0.0f;
    for (int r = r0; r <= r1; ++r)
        for (int q = c0; q <= c1; ++q) s += wk[r * 3 + q];
    weff[i] = s;
}

struct FinalArgs {
    const float* cand[6];
    int64_t bstride[6];
    int64_t cstride[6];          // floats between the 3 channel planes of a candidate
};

struct D3Grid { int tiles_x, per_sample, total, per_xcd; };

template <typename OUT>
__global__ __launch_bounds__(256) void dec3_synth_kernel(const float* __restrict__ d2, const float* __restrict__ weff,
                                                         const float* __restrict__ bias, FinalArgs cd,
                                                         const float* __restrict__ tv, double T, OUT* __restrict__ out,
                                                         float* __restrict__ refine_dbg, int H, int W, int xs, D3Grid gr) {
    // xs: the tile grid starts xs low-resolution columns left of the image (fldr_dec3_synth_strided)
    const int h = H >> 1, w = W >> 1;
    __shared__ float tile[D3_CIN][D3_TH + 2][D3_TW + 2];
    const int tid = threadIdx.x, tx = tid % D3_TW, ty = tid / D3_TW;
    // Tiles are dealt to the XCDs in contiguous row-major ranges (workgroup b runs on XCD b & 7): horizontally adjacent tiles
    // share the 128-byte lines that hold their halo columns of dec2's output, and then find them in the same L2 instead
    // of fetching them once per XCD.
    const int lin = gr.per_xcd ? (int)(blockIdx.x & 7) * gr.per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;   // (per_xcd 0: row-major, test hook)
    if ((gr.per_xcd && (int)(blockIdx.x >> 3) >= gr.per_xcd) || lin >= gr.total) return;      // workgroup-uniform
    const int n = lin / gr.per_sample, trem = lin - n * gr.per_sample;
    const int tyi = trem / gr.tiles_x;
    const int i0 = tyi * D3_TH, j0 = (trem - tyi * gr.tiles_x) * D3_TW - xs;
    const float* src = d2 + (int64_t)n * D3_CIN * h * w;
    constexpr int TILE_E = (D3_TH + 2) * (D3_TW + 2);
    // Staging: this thread's (up to two) slots of a channel's (TH+2) x (TW+2) tile are the same for all 16 channels, so the
    // index arithmetic is done once and every load is a scalar channel base + a precomputed 32-bit lane offset (the kernel
    // is bound by its vector-ALU instruction count: the flat element -> (channel, row, column) decode per load cost
    // ~400 instructions per thread).  Unconditional clamped loads, then mask (see common.h).
    static_assert(TILE_E <= 512, "two slots per thread");
    uint32_t goff[2];
    int lidx[2];
    bool inb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = tid + 256 * j;
        const int rr = r < TILE_E ? r : 0;
        const int gy = i0 - 1 + rr / (D3_TW + 2), gx = j0 - 1 + rr % (D3_TW + 2);
        const int cy = min(max(gy, 0), h - 1), cx = min(max(gx, 0), w - 1);
        goff[j] = (__umul24((uint32_t)cy, (uint32_t)w) + (uint32_t)cx) * 4u;
        lidx[j] = r < TILE_E ? r : -1;
        inb[j] = gy >= 0 && gy < h && gx >= 0 && gx < w;
    }
    float st[D3_CIN][2];
    const int64_t hw = (int64_t)h * w;
#pragma unroll
    for (int c = 0; c < D3_CIN; ++c) {
        const char* base = reinterpret_cast<const char*>(src + c * hw);          // uniform
#if defined(DEC3_ABLATE) && (DEC3_ABLATE & 1)                         // diagnostic: no reads of dec2's output
        st[c][0] = 0.25f; st[c][1] = 0.5f; (void)base;
#else
        st[c][0] = *reinterpret_cast<const float*>(base + goff[0]);
        st[c][1] = *reinterpret_cast<const float*>(base + goff[1]);
#endif
    }
#pragma unroll
    for (int c = 0; c < D3_CIN; ++c) {
        fldr_pin(st[c][0]); fldr_pin(st[c][1]);
        float* tc = &tile[c][0][0];
        tc[lidx[0]] = inb[0] ? st[c][0] : 0.0f;                                   // (tid < 256 <= TILE_E: slot 0 always exists)
        if (lidx[1] >= 0) tc[lidx[1]] = inb[1] ? st[c][1] : 0.0f;
    }
    __syncthreads();

    // Candidate pixels of this thread's 2x2 output quad: row a = 0 is requested BEFORE the convolution below and row 1
    // before row 0's fp64 tail, so that HBM keeps streaming under the ALU phases (clamped addresses for the threads of
    // a partial tile; they return before using them).
    const int li = i0 + ty, lj = j0 + tx;
    const int64_t HW = (int64_t)H * W;
    const int lic = min(li, h - 1), ljc = min(max(lj, 0), w - 1);
    float2 cv[2][6][3];
    auto load_cands = [&](int a) __attribute__((always_inline)) {
        const uint32_t pob = (__umul24((uint32_t)(2 * lic + a), (uint32_t)W) + (uint32_t)(2 * ljc)) * 4u;     // byte offset inside a plane (< 4 GB, host-checked)
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
            for (int ch = 0; ch < 3; ++ch)
                {
                    const float2* cp = reinterpret_cast<const float2*>(reinterpret_cast<const char*>(cd.cand[k] + (int64_t)n * cd.bstride[k] + (int64_t)ch * cd.cstride[k]) + pob);
#if defined(DEC3_ABLATE) && (DEC3_ABLATE & 2)                         // diagnostic: no candidate reads
                    cv[a][k][ch] = make_float2(0.1f * k, 0.2f * ch); (void)cp;
#elif defined(DEC3_NT) && (DEC3_NT & 1)
                    typedef float d3_f2 __attribute__((ext_vector_type(2)));
                    const d3_f2 t2 = __builtin_nontemporal_load(reinterpret_cast<const d3_f2*>(cp));          // read once: streaming hint
                    cv[a][k][ch] = make_float2(t2[0], t2[1]);
#else
                    cv[a][k][ch] = *cp;
#endif
                }
    };
    load_cands(0);

    float acc[4][D3_COUT];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int co = 0; co < D3_COUT; ++co) acc[ph][co] = bias[co];
#pragma unroll 2
    for (int c = 0; c < D3_CIN; ++c) {
        float x[3][3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) x[dy][dx] = tile[c][ty + dy][tx + dx];
        const float* wc = weff + c * (4 * D3_COUT * 4);      // wave-uniform -> scalar loads
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const int a = ph >> 1, b = ph & 1;
#pragma unroll
            for (int co = 0; co < D3_COUT; ++co) {
                const float* wq = wc + (ph * D3_COUT + co) * 4;
                float s = acc[ph][co];
                s = fmaf(wq[0], x[a][b], s);
                s = fmaf(wq[1], x[a][b + 1], s);
                s = fmaf(wq[2], x[a + 1][b], s);
                s = fmaf(wq[3], x[a + 1][b + 1], s);
                acc[ph][co] = s;
            }
        }
    }

#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) { if (a == 0) { fldr_pin(cv[0][k][ch].x); fldr_pin(cv[0][k][ch].y); } }
    load_cands(1);
    if (li >= h || lj >= w || lj < 0) return;
    const float t = tv[n];
    const double w1 = (double)t, w0 = (double)(1.0f - t);
    const double inv_T = 1.0 / T;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int64_t po = (int64_t)(2 * li + a) * W + 2 * lj;           // two horizontally adjacent output pixels
        double res[2][3];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma clang fp contract(off)
            const int ph = a * 2 + b;
            if (refine_dbg) {
#pragma unroll
                for (int co = 0; co < D3_COUT; ++co) refine_dbg[((int64_t)n * D3_COUT + co) * HW + po + b] = acc[ph][co];
            }
            // The vector ALUs' fp64 rate co-limits this kernel with HBM: the 15 fp64 divisions per pixel of the literal
            // formula (logit / T, exp / sum, blend / div) are 2 reciprocals and multiplications here (<= 1 ulp of fp64 per
            // factor, ten orders of magnitude below the fp32 inputs' own precision).
            double s[6], mx = -1.0e300;
#pragma unroll
            for (int k = 0; k < 6; ++k) { s[k] = (double)acc[ph][k] * inv_T; mx = s[k] > mx ? s[k] : mx; }
            double sum = 0.0;
#pragma unroll
#if defined(DEC3_ABLATE) && (DEC3_ABLATE & 4)                         // diagnostic: no exponentials
                        for (int k = 0; k < 6; ++k) { s[k] = s[k] - mx + 2.0; sum += s[k]; }
#else
                        for (int k = 0; k < 6; ++k) { s[k] = exp(s[k] - mx); sum += s[k]; }
#endif
            const double inv_sum = 1.0 / sum;
            double wo[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) wo[k] = ((k & 1) ? w1 : w0) * (s[k] * inv_sum);
            double div = ((wo[0] + wo[1]) + wo[2]) + wo[3];               // fLDRnet.py:517
            div = div + (wo[4] + wo[5]);                                   // :522
            const double inv_div = 1.0 / div;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                double v[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) v[k] = wo[k] * (double)(b ? cv[a][k][ch].y : cv[a][k][ch].x);
                double o = v[0] + v[1];                                    // :518
                o = o + (v[2] + v[3]);                                     // :520
                o = o + (v[4] + v[5]);                                     // :521
                res[b][ch] = o * inv_div;                                  // :524
            }
        }
        // both pixels of the pair in one 16-B (fp64) / 8-B (fp32) store per channel
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            char* o = reinterpret_cast<char*>(out + ((int64_t)n * 3 + ch) * HW) + (uint32_t)po * (uint32_t)sizeof(OUT);      // scalar plane base + 32-bit offset
#if defined(DEC3_NT) && (DEC3_NT & 2)
            typedef double d3_d2 __attribute__((ext_vector_type(2)));
            typedef float d3_f2o __attribute__((ext_vector_type(2)));
            if constexpr (sizeof(OUT) == 8) __builtin_nontemporal_store(d3_d2{res[0][ch], res[1][ch]}, reinterpret_cast<d3_d2*>(o));
            else __builtin_nontemporal_store(d3_f2o{(float)res[0][ch], (float)res[1][ch]}, reinterpret_cast<d3_f2o*>(o));
#else
            if constexpr (sizeof(OUT) == 8) *reinterpret_cast<double2*>(o) = make_double2(res[0][ch], res[1][ch]);
            else *reinterpret_cast<float2*>(o) = make_float2((float)res[0][ch], (float)res[1][ch]);
#endif
        }
    }
}

static int g_d3_xcd = 1;                         // 1: XCD-contiguous tile ranges (see the kernel); 0: row-major round-robin
FLDR_HOOK int fldr_debug_dec3_xcd(int v) { if (v == 0 || v == 1) g_d3_xcd = v; return g_d3_xcd; }
static int g_d3_xshift = -1;                     // -1: automatic (16 on wide frames); 0 .. 31: forced
FLDR_HOOK int fldr_debug_dec3_xshift(int v) { if (v >= -1 && v < D3_TW) g_d3_xshift = v; return g_d3_xshift; }

extern "C" int fldr_dec3_prepack(const float* weight, float* weff, fldr_stream_t stream) {
    FLDR_CHECK_ARG(weight && weff);
    hipLaunchKernelGGL(dec3_prepack_kernel, dim3(fldr_cdiv(D3_CIN * 4 * D3_COUT * 4, 256)), dim3(256), 0, fldr_s(stream), weight, weff);
    FLDR_LAUNCH_RET();
}

extern "C" int fldr_dec3_synth_strided(const float* d2, const float* weff, const float* bias, const float* const cand[6],
                                       const int64_t cand_bstride[6], const int64_t cand_cstride[6], const float* t,
                                       double T_param, double* out_f64, float* out_f32, float* refine_out_or_null, int N,
                                       int H, int W, fldr_stream_t stream) {
    FLDR_CHECK_ARG(d2 && weff && bias && cand && cand_bstride && cand_cstride && t && N > 0 && H > 0 && W > 0);
    FLDR_CHECK_ARG((out_f64 != nullptr) != (out_f32 != nullptr));
    if ((H | W) & 1) return FLDR_E_SHAPE;
    if ((int64_t)H * W * 8 >= (1ll << 32)) return FLDR_E_SHAPE;            // 32-bit byte offsets inside a plane
    FLDR_CHECK_ARG(((reinterpret_cast<uintptr_t>(out_f64) & 15) | (reinterpret_cast<uintptr_t>(out_f32) & 7)) == 0);     // pixel pairs are stored whole
    FinalArgs a;
    for (int k = 0; k < 6; ++k) {
        FLDR_CHECK_ARG(cand[k] && (((uintptr_t)cand[k]) & 7) == 0 && (cand_bstride[k] & 1) == 0 && (cand_cstride[k] & 1) == 0);
        a.cand[k] = cand[k]; a.bstride[k] = cand_bstride[k]; a.cstride[k] = cand_cstride[k];
    }
    // Tile grid shifted 16 low-resolution columns left on wide frames: a tile's 34 staged columns of dec2's output then
    // start 15 floats into a 128-byte line and touch 2 lines per row instead of 3 ([32 t - 1, 32 t + 32]: one float each
    // into the lines left and right; PMC at 4K: 526 MB fetched for the 141 MB of dec2's output), while the full-resolution
    // candidate loads and the frame stores (64 t - 32 ...) stay line-aligned.  One extra half-filled tile column.
    const int xs = g_d3_xshift >= 0 ? g_d3_xshift : (W >= 1024 ? 16 : 0);
    D3Grid gr;
    gr.tiles_x = fldr_cdiv(W / 2 + xs, D3_TW);
    gr.per_sample = gr.tiles_x * fldr_cdiv(H / 2, D3_TH);
    if ((int64_t)gr.per_sample * N > (1ll << 30)) return FLDR_E_SHAPE;
    gr.total = gr.per_sample * N;
    gr.per_xcd = g_d3_xcd ? (gr.total + 7) / 8 : 0;
    dim3 grid(g_d3_xcd ? 8 * gr.per_xcd : gr.total);
    if (out_f64) hipLaunchKernelGGL(dec3_synth_kernel<double>, grid, dim3(256), 0, fldr_s(stream), d2, weff, bias, a, t, T_param, out_f64, refine_out_or_null, H, W, xs, gr);
    else         hipLaunchKernelGGL(dec3_synth_kernel<float>, grid, dim3(256), 0, fldr_s(stream), d2, weff, bias, a, t, T_param, out_f32, refine_out_or_null, H, W, xs, gr);
    FLDR_LAUNCH_RET();
}

extern "C" int fldr_dec3_synth(const float* d2, const float* weff, const float* bias, const float* const cand[6],
                               const int64_t cand_bstride[6], const float* t, double T_param, double* out_f64, float* out_f32,
                               float* refine_out_or_null, int N, int H, int W, fldr_stream_t stream) {
    const int64_t HW = (int64_t)H * W;
    const int64_t cs[6] = {HW, HW, HW, HW, HW, HW};
    return fldr_dec3_synth_strided(d2, weff, bias, cand, cand_bstride, cs, t, T_param, out_f64, out_f32, refine_out_or_null, N, H, W, stream);
}
