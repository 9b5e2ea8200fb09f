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

// MF (round 3): the phase convolutions on the fp16 MATRIX cores with the 3 x fp16 split of the 3x3 convolutions (x = hi + lo,
// hi*hi + hi*lo + lo*hi, fp32 accumulation), reading dec2's SPLIT-PACKED output.  6 of 16 matrix rows are useful and every
// product is issued three times — and it is still the better deal: 96 v_mfma_f32_16x16x32_f16 per wave and tile (1,536 cycles of
// a pipe that was idle) replace 768 packed fp32 FMAs + 144 LDS reads on the vector ALUs (~3,600 cycles of the pipe the fp64
// tail needs); ablation of the vector kernel at 4K: 163 of its 237 us are arithmetic.  Per phase (a, b): D[co][pixel] +=
// A[co][k] * B[k][pixel], k = (tap (dy2, dx2), channel of an 8-channel group): an A operand is a 1-KB block of the prepacked
// table (fldr_dec3_prepack_spk: [phase][group][hi, lo][lane][8 halves], rows 6 .. 15 zero), a B operand one 16-byte LDS read of
// the packed tile (lane group = tap: the pixel at row + a + dy2, column + b + dx2).  The tile and the table arrive by LDS-DMA;
// the logits go back through the LDS to the thread-per-pixel layout of the fp64 tail, which is unchanged.
#define D3M_PLANE 5632                 // (TH + 2) x (TW + 2) pixels of 16 bytes, padded to 1-KB DMA pieces with an overlapping last one
#define D3M_TABLE (4 * 2 * 2 * 1024)   // A operands
#define D3M_HDR 16                     // floats before the table: {1 / scale, scale, max |w|, 0 ...}; floats 4 .. 7 are the DMA's zero block
typedef _Float16 d3_h8 __attribute__((ext_vector_type(8)));
typedef float d3_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* d3_gptr_t;
typedef __attribute__((address_space(3))) void* d3_lptr_t;

template <typename OUT, bool MF>
__global__ __launch_bounds__(256) void dec3_synth_kernel(const void* __restrict__ d2v, const float* __restrict__ weff,
                                                         const float* __restrict__ bias, FinalArgs cd,
                                                         const float* __restrict__ tv, const float* __restrict__ poison, double T, OUT* __restrict__ out,
                                                         float* __restrict__ refine_dbg, int H, int W, int xs, D3Grid gr) {
    // xs: the tile grid starts xs low-resolution columns left of the image (fldr_dec3_synth_strided)
    const int h = H >> 1, w = W >> 1;
    constexpr int LOGITS_BYTES = D3_COUT * D3_TH * D3_TW * 4;                // one phase: [co][pixel]
    constexpr int SMEM = MF ? 4 * D3M_PLANE + LOGITS_BYTES + D3M_TABLE : D3_CIN * (D3_TH + 2) * (D3_TW + 2) * 4;
    __shared__ __attribute__((aligned(1024))) unsigned char d3_smem[SMEM];
    float (*tile)[D3_TH + 2][D3_TW + 2] = reinterpret_cast<float (*)[D3_TH + 2][D3_TW + 2]>(d3_smem);
    const float* d2 = static_cast<const float*>(d2v);
    const int tid = threadIdx.x, tx = tid % D3_TW, ty = tid / D3_TW;
    // Tiles are dealt to the XCDs in contiguous row-major ranges (workgroup b runs on XCD b & 7): horizontally adjacent tiles
    // share the 128-byte lines that hold their halo columns of dec2's output, and then find them in the same L2 instead
    // of fetching them once per XCD.
    const int lin = gr.per_xcd ? (int)(blockIdx.x & 7) * gr.per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;   // (per_xcd 0: row-major, test hook)
    if ((gr.per_xcd && (int)(blockIdx.x >> 3) >= gr.per_xcd) || lin >= gr.total) return;      // workgroup-uniform
    const int n = lin / gr.per_sample, trem = lin - n * gr.per_sample;
    const int tyi = trem / gr.tiles_x;
    const int i0 = tyi * D3_TH, j0 = (trem - tyi * gr.tiles_x) * D3_TW - xs;
    constexpr int TILE_E = (D3_TH + 2) * (D3_TW + 2);
    if constexpr (MF) {
        // ---- the packed tile (4 planes: group 0 hi / lo, group 1 hi / lo) and the A table by LDS-DMA; pixels outside the image
        //      come from the zero block.  A DMA instruction fills 64 consecutive 16-byte slots. ----
        const int lane = tid & 63;
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const unsigned char* tab = reinterpret_cast<const unsigned char*>(weff);
        const unsigned char* zero = tab + 16;
        const int64_t plane_b = (int64_t)h * w * 16;
        const unsigned char* pk = static_cast<const unsigned char*>(d2v) + (int64_t)n * 4 * plane_b + (int64_t)wv * plane_b;   // wave wv stages plane wv
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int piece = i * 64 < D3M_PLANE / 16 - 64 ? i * 64 : D3M_PLANE / 16 - 64;
            const int e = piece + lane;
            const int gy = i0 - 1 + e / (D3_TW + 2), gx = j0 - 1 + e % (D3_TW + 2);
            const bool ok = e < TILE_E && gy >= 0 && gy < h && gx >= 0 && gx < w;
            const unsigned char* g = ok ? pk + ((int64_t)gy * w + gx) * 16 : zero;
            __builtin_amdgcn_global_load_lds((d3_gptr_t)g, (d3_lptr_t)(d3_smem + wv * D3M_PLANE + piece * 16), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < D3M_TABLE / 1024 / 4; ++i)
            __builtin_amdgcn_global_load_lds((d3_gptr_t)(tab + D3M_HDR * 4 + ((i * 4 + wv) * 64 + lane) * 16),
                                             (d3_lptr_t)(d3_smem + SMEM - D3M_TABLE + (i * 4 + wv) * 1024), 16, 0, 0);
    } else {
    const float* src = d2 + (int64_t)n * D3_CIN * h * w;
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
    }

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
    if constexpr (MF) {
        const int lane = tid & 63, lq = lane & 15, lg = lane >> 4;
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const float inv_scale = weff[0];
        __builtin_amdgcn_s_waitcnt(0x0F70);                                // vmcnt(0): my DMA pieces (and row 0 of the candidates) have landed
        __syncthreads();
        const unsigned char* tabl = d3_smem + SMEM - D3M_TABLE + lane * 16;
        // One phase at a time: 24 matrix instructions into four accumulators (pixel blocks: row 2 wv + (q >> 1), columns 16 (q & 1) ...),
        // then the phase's logits cross from the matrix layout (lane = pixel column, registers = 4 channels) to the tail's (thread =
        // pixel, registers = channels) through 6 KB of LDS.  A wave's matrix columns are exactly its own threads' pixels (rows 2 wv,
        // 2 wv + 1 = threads 64 wv ... 64 wv + 63), so the exchange needs no workgroup barrier.
        float* lg_s = reinterpret_cast<float*>(d3_smem + 4 * D3M_PLANE);   // [co][pixel = row * 32 + column]
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const int a = ph >> 1, b = ph & 1;
            d3_h8 wa[2][2];                                                // [group][hi, lo]
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int k = 0; k < 2; ++k) wa[g][k] = *reinterpret_cast<const d3_h8*>(tabl + ((ph * 2 + g) * 2 + k) * 1024);
            d3_f4 d[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // lane group lg = tap (dy2, dx2) = (lg >> 1, lg & 1): the pixel at tile row r + a + dy2, tile column c + b + dx2
                const int slot = (2 * wv + (q >> 1) + a + (lg >> 1)) * (D3_TW + 2) + 16 * (q & 1) + lq + b + (lg & 1);
                d3_f4 c4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const d3_h8 xh = *reinterpret_cast<const d3_h8*>(d3_smem + (2 * g) * D3M_PLANE + slot * 16);
                    const d3_h8 xl = *reinterpret_cast<const d3_h8*>(d3_smem + (2 * g + 1) * D3M_PLANE + slot * 16);
                    c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[g][0], xh, c4, 0, 0, 0);
                    c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[g][0], xl, c4, 0, 0, 0);
                    c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[g][1], xh, c4, 0, 0, 0);
                }
                d[q] = c4;
            }
            if (lg < 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = 4 * lg + r;                         // matrix row
                        if (co < D3_COUT) lg_s[co * (D3_TH * D3_TW) + (2 * wv + (q >> 1)) * D3_TW + 16 * (q & 1) + lq] = d[q][r];
                    }
            }
            // (lanes exchange data without a workgroup barrier: the wave-scope fences keep the compiler — which orders memory
            // operations per thread — from moving a thread's reads across other lanes' writes; LDS executes a wave's operations in order)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int co = 0; co < D3_COUT; ++co) acc[ph][co] = lg_s[co * (D3_TH * D3_TW) + tid] * inv_scale + bias[co];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    } else {
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
    }

#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) { if (a == 0) { fldr_pin(cv[0][k][ch].x); fldr_pin(cv[0][k][ch].y); } }
    load_cands(1);
    if (li >= h || lj >= w || lj < 0) return;
    const float t = tv[n] + *poison;                                     // (*poison: 0.0f, NaN once a ring wait expired — common.h)
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

static int d3_launch(bool mf, const void* d2, const float* weff, const float* bias, const float* const cand[6],
                     const int64_t cand_bstride[6], const int64_t cand_cstride[6], const float* t,
                     double T_param, double* out_f64, float* out_f32, float* refine_out_or_null, int N,
                     int H, int W, fldr_stream_t stream) {
    FLDR_CHECK_ARG(d2 && weff && bias && cand && cand_bstride && cand_cstride && t && N > 0 && H > 0 && W > 0);
    FLDR_CHECK_ARG((out_f64 != nullptr) != (out_f32 != nullptr));
    if ((H | W) & 1) return FLDR_E_SHAPE;
    if ((int64_t)H * W * 8 >= (1ll << 32)) return FLDR_E_SHAPE;            // 32-bit byte offsets inside a plane
    FLDR_CHECK_ARG(((reinterpret_cast<uintptr_t>(out_f64) & 15) | (reinterpret_cast<uintptr_t>(out_f32) & 7)) == 0);     // pixel pairs are stored whole
    const float* poison = fldr_status_poison_ptr();                        // (common.h: frames written after a ring fault are NaN)
    if (!poison) return FLDR_E_STATUS;
    FinalArgs a;
    for (int k = 0; k < 6; ++k) {
        FLDR_CHECK_ARG(cand[k] && (((uintptr_t)cand[k]) & 7) == 0 && (cand_bstride[k] & 1) == 0 && (cand_cstride[k] & 1) == 0);
        a.cand[k] = cand[k]; a.bstride[k] = cand_bstride[k]; a.cstride[k] = cand_cstride[k];
    }
    // Tile grid shifted 16 low-resolution columns left on wide frames: a tile's 34 staged columns of dec2's output then
    // start 15 floats into a 128-byte line and touch 2 lines per row instead of 3 ([32 t - 1, 32 t + 32]: one float each
    // into the lines left and right; PMC at 4K: 526 MB fetched for the 141 MB of dec2's output), while the full-resolution
    // candidate loads and the frame stores (64 t - 32 ...) stay line-aligned.  One extra half-filled tile column.
    // (the packed source's pixels are 16-byte records: no shift)
    const int xs = mf ? 0 : (g_d3_xshift >= 0 ? g_d3_xshift : (W >= 1024 ? 16 : 0));
    if (mf && ((reinterpret_cast<uintptr_t>(d2) | reinterpret_cast<uintptr_t>(weff)) & 15)) return FLDR_E_ARG;
    D3Grid gr;
    gr.tiles_x = fldr_cdiv(W / 2 + xs, D3_TW);
    gr.per_sample = gr.tiles_x * fldr_cdiv(H / 2, D3_TH);
    if ((int64_t)gr.per_sample * N > (1ll << 30)) return FLDR_E_SHAPE;
    gr.total = gr.per_sample * N;
    gr.per_xcd = g_d3_xcd ? (gr.total + 7) / 8 : 0;
    dim3 grid(g_d3_xcd ? 8 * gr.per_xcd : gr.total);
    if (mf) {
        if (out_f64) hipLaunchKernelGGL((dec3_synth_kernel<double, true>), grid, dim3(256), 0, fldr_s(stream), d2, weff, bias, a, t, poison, T_param, out_f64, refine_out_or_null, H, W, xs, gr);
        else         hipLaunchKernelGGL((dec3_synth_kernel<float, true>), grid, dim3(256), 0, fldr_s(stream), d2, weff, bias, a, t, poison, T_param, out_f32, refine_out_or_null, H, W, xs, gr);
    } else {
        if (out_f64) hipLaunchKernelGGL((dec3_synth_kernel<double, false>), grid, dim3(256), 0, fldr_s(stream), d2, weff, bias, a, t, poison, T_param, out_f64, refine_out_or_null, H, W, xs, gr);
        else         hipLaunchKernelGGL((dec3_synth_kernel<float, false>), grid, dim3(256), 0, fldr_s(stream), d2, weff, bias, a, t, poison, T_param, out_f32, refine_out_or_null, H, W, xs, gr);
    }
    FLDR_LAUNCH_RET();
}

extern "C" int fldr_dec3_synth_strided(const float* d2, const float* weff, const float* bias, const float* const cand[6],
                                       const int64_t cand_bstride[6], const int64_t cand_cstride[6], const float* t,
                                       double T_param, double* out_f64, float* out_f32, float* refine_out_or_null, int N,
                                       int H, int W, fldr_stream_t stream) {
    return d3_launch(false, d2, weff, bias, cand, cand_bstride, cand_cstride, t, T_param, out_f64, out_f32, refine_out_or_null, N, H, W, stream);
}

// The same operator on dec2's SPLIT-PACKED output (fldr_spk_bytes(16, H/2, W/2) bytes per sample), phase convolutions on the
// fp16 matrix cores (3 x fp16 split: fp32-equivalent logits, not the bits of the fp32-FMA kernel above); wm: fldr_dec3_prepack_spk.
extern "C" int fldr_dec3_synth_spk(const void* d2_spk, const float* wm, const float* bias, const float* const cand[6],
                                   const int64_t cand_bstride[6], const int64_t cand_cstride[6], const float* t,
                                   double T_param, double* out_f64, float* out_f32, float* refine_out_or_null, int N,
                                   int H, int W, fldr_stream_t stream) {
    return d3_launch(true, d2_spk, wm, bias, cand, cand_bstride, cand_cstride, t, T_param, out_f64, out_f32, refine_out_or_null, N, H, W, stream);
}

// wm: D3M_HDR floats {1 / scale, scale, max |w|, 0, zero block ...} + the A operands [phase][group][hi, lo][lane][8 halves]
// (lane = (matrix row = output channel, lane group = tap); element j = input channel 8 group + j): the per-phase 2x2 weights
// of dec3_prepack_kernel, scaled by a power of two into the fp16 range and split into hi + lo.
__global__ void dec3_prepack_spk_kernel(const float* __restrict__ w, float* __restrict__ wm) {
    __shared__ float we[D3_CIN * 4 * D3_COUT * 4];
    __shared__ float red[256];
    const int tid = threadIdx.x;
    float m = 0.0f;
    for (int i = tid; i < D3_CIN * 4 * D3_COUT * 4; i += 256) {
        const int tap = i & 3, co = (i >> 2) % D3_COUT, ph = (i / (4 * D3_COUT)) & 3, c = i / (16 * D3_COUT);
        const int a = ph >> 1, b = ph & 1, dy2 = tap >> 1, dx2 = tap & 1;
        const int r0 = a == 0 ? (dy2 == 0 ? 0 : 1) : (dy2 == 0 ? 0 : 2), r1 = a == 0 ? (dy2 == 0 ? 0 : 2) : (dy2 == 0 ? 1 : 2);
        const int c0 = b == 0 ? (dx2 == 0 ? 0 : 1) : (dx2 == 0 ? 0 : 2), c1 = b == 0 ? (dx2 == 0 ? 0 : 2) : (dx2 == 0 ? 1 : 2);
        const float* wk = w + ((int64_t)co * D3_CIN + c) * 9;
        float sum = 0.0f;
        for (int r = r0; r <= r1; ++r)
            for (int q = c0; q <= c1; ++q) sum += wk[r * 3 + q];
        we[i] = sum;
        m = fmaxf(m, fabsf(sum));
    }
    red[tid] = m;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) red[tid] = fmaxf(red[tid], red[tid + st]);
        __syncthreads();
    }
    const float mx = red[0];
    float scale = 1.0f;                                                  // largest power of two with mx * scale <= 8192 (as spk_absmax_kernel)
    if (mx > 0.0f) scale = exp2f(floorf(log2f(8192.0f / mx)));
    if (tid < D3M_HDR) wm[tid] = tid == 0 ? 1.0f / scale : (tid == 1 ? scale : (tid == 2 ? mx : 0.0f));
    d3_h8* frag = reinterpret_cast<d3_h8*>(wm + D3M_HDR);
    for (int i = tid; i < 16 * 64; i += 256) {
        const int lane = i & 63, blk = i >> 6;
        const int kind = blk & 1, g = (blk >> 1) & 1, ph = blk >> 2;
        const int co = lane & 15, tap = lane >> 4;
        d3_h8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = g * 8 + j;
            const float x = co < D3_COUT ? we[((c * 4 + ph) * D3_COUT + co) * 4 + tap] * scale : 0.0f;
            const _Float16 hh = (_Float16)x;
            v[j] = kind == 0 ? hh : (_Float16)(x - (float)hh);
        }
        frag[i] = v;
    }
}

extern "C" int64_t fldr_dec3_prepack_spk_size(void) { return D3M_HDR + D3M_TABLE / 4; }

extern "C" int fldr_dec3_prepack_spk(const float* weight, float* wm, fldr_stream_t stream) {
    FLDR_CHECK_ARG(weight && wm && (reinterpret_cast<uintptr_t>(wm) & 15) == 0);
    hipLaunchKernelGGL(dec3_prepack_spk_kernel, dim3(1), dim3(256), 0, fldr_s(stream), weight, wm);
    FLDR_LAUNCH_RET();
}

extern "C" int fldr_dec3_synth(const float* d2, const float* weff, const float* bias, const float* const cand[6],
                               const int64_t cand_bstride[6], const float* t, double T_param, double* out_f64, float* out_f32,
                               float* refine_out_or_null, int N, int H, int W, fldr_stream_t stream) {
    const int64_t HW = (int64_t)H * W;
    const int64_t cs[6] = {HW, HW, HW, HW, HW, HW};
    return fldr_dec3_synth_strided(d2, weff, bias, cand, cand_bstride, cs, t, T_param, out_f64, out_f32, refine_out_or_null, N, H, W, stream);
}
