// Backward operators of the two custom ops: softmax-splat backward (softSplat.py:54-158, launched at
// :259-318) and cost-volume backward (correlation.py:114-242, launched at :350-410).  Training itself is out of scope of
// this path; these make the operators usable under autograd (softSplat._FunctionSoftsplat / correlation._FunctionCorrelation).
// HBM/gather-bound: one thread per pixel, channels looped in registers so that the bilinear geometry (splat) or the 81
// displacement weights (correlation) are formed once per pixel.
#include "common.h"

// ---- splat: gradInput (kernel_Softsplat_updateGradInput) and gradFlow (kernel_Softsplat_updateGradFlow) in one pass ----
// gradInput[n,c,y,x] = sum_corner gradOutput[n,c,corner] * w_corner      (in-bounds corners only, order NW,NE,SW,SE)
// gradFlow[n,0,y,x]  = sum_c sum_corner in[c] * gradOutput[c,corner] * dw_corner/dx,  gradFlow[n,1,...] likewise d/dy,
// with the reference's factorisation and accumulation order (per channel: NW, NE, SW, SE).
__global__ __launch_bounds__(256) void splat_bwd_kernel(const float* __restrict__ in, const float* __restrict__ flow,
                                                        const float* __restrict__ gout, float* __restrict__ gin,
                                                        float* __restrict__ gflow, int C, int H, int W) {
#pragma clang fp contract(off)
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int n = blockIdx.z;
    if (x >= W || y >= H) return;
    const int64_t HW = (int64_t)H * W, pix = (int64_t)y * W + x;
    const float ox = (float)x + flow[(int64_t)n * 2 * HW + pix];
    const float oy = (float)y + flow[(int64_t)n * 2 * HW + HW + pix];
    float xf = floorf(ox), yf = floorf(oy);
    // corner coordinates as floats (softSplat.py:70-83); clamped only for the int conversion of wild flows
    const float x1f = xf + 1.0f, y1f = yf + 1.0f;
    const float w_nw = (x1f - ox) * (y1f - oy), w_ne = (ox - xf) * (y1f - oy);
    const float w_sw = (x1f - ox) * (oy - yf), w_se = (ox - xf) * (oy - yf);
    // d/dx and d/dy factors (softSplat.py:131-141)
    const float dx_nw = (-1.0f) * (y1f - oy), dx_ne = (+1.0f) * (y1f - oy), dx_sw = (-1.0f) * (oy - yf), dx_se = (+1.0f) * (oy - yf);
    const float dy_nw = (x1f - ox) * (-1.0f), dy_ne = (ox - xf) * (-1.0f), dy_sw = (x1f - ox) * (+1.0f), dy_se = (ox - xf) * (+1.0f);
    const float cxf = fminf(fmaxf(xf, -2.0f), (float)W + 1.0f), cyf = fminf(fmaxf(yf, -2.0f), (float)H + 1.0f);
    const int x0 = (int)cxf, y0 = (int)cyf;
    const bool x0v = x0 >= 0 && x0 < W, x1v = x0 + 1 >= 0 && x0 + 1 < W;
    const bool y0v = y0 >= 0 && y0 < H, y1v = y0 + 1 >= 0 && y0 + 1 < H;
    const bool vnw = x0v && y0v, vne = x1v && y0v, vsw = x0v && y1v, vse = x1v && y1v;
    const int xa = min(max(x0, 0), W - 1), xb = min(max(x0 + 1, 0), W - 1);
    const int ya = min(max(y0, 0), H - 1), yb = min(max(y0 + 1, 0), H - 1);
    const int64_t onw = (int64_t)ya * W + xa, one = (int64_t)ya * W + xb, osw = (int64_t)yb * W + xa, ose = (int64_t)yb * W + xb;
    float gfx = 0.0f, gfy = 0.0f;
    for (int c = 0; c < C; ++c) {
        const float* g = gout + ((int64_t)n * C + c) * HW;
        float gnw = g[onw], gne = g[one], gsw = g[osw], gse = g[ose];      // unconditional clamped loads, masked below
        fldr_pin(gnw); fldr_pin(gne); fldr_pin(gsw); fldr_pin(gse);
        if (gin) {
            float v = 0.0f;
            if (vnw) v += gnw * w_nw;
            if (vne) v += gne * w_ne;
            if (vsw) v += gsw * w_sw;
            if (vse) v += gse * w_se;
            gin[((int64_t)n * C + c) * HW + pix] = v;
        }
        if (gflow) {
            const float iv = in[((int64_t)n * C + c) * HW + pix];
            if (vnw) { gfx += iv * gnw * dx_nw; gfy += iv * gnw * dy_nw; }
            if (vne) { gfx += iv * gne * dx_ne; gfy += iv * gne * dy_ne; }
            if (vsw) { gfx += iv * gsw * dx_sw; gfy += iv * gsw * dy_sw; }
            if (vse) { gfx += iv * gse * dx_se; gfy += iv * gse * dy_se; }
        }
    }
    if (gflow) {
        gflow[(int64_t)n * 2 * HW + pix] = gfx;
        gflow[(int64_t)n * 2 * HW + HW + pix] = gfy;
    }
}

extern "C" int fldr_softsplat_bwd(const float* in, const float* flow, const float* grad_out, float* grad_in_or_null,
                                  float* grad_flow_or_null, int N, int C, int H, int W, fldr_stream_t stream) {
    FLDR_CHECK_ARG(in && flow && grad_out && (grad_in_or_null || grad_flow_or_null) && N > 0 && C > 0 && H > 0 && W > 0);
    dim3 grid(fldr_cdiv(W, 64), fldr_cdiv(H, 4), N);
    hipLaunchKernelGGL(splat_bwd_kernel, grid, dim3(256), 0, fldr_s(stream), in, flow, grad_out, grad_in_or_null, grad_flow_or_null, C, H, W);
    FLDR_LAUNCH_RET();
}

// ---- correlation: gradFirst (kernel_Correlation_updateGradFirst) / gradSecond (kernel_Correlation_updateGradSecond) ----
// gradFirst[n,c,y,x]  = (1/C) sum_{p,o in [-4,4]} gradOut[n,(p+4)*9+(o+4),y,x]     * second[n,c,y+p,x+o]   (zero padded)
// gradSecond[n,c,y,x] = (1/C) sum_{p,o}           gradOut[n,(p+4)*9+(o+4),y-p,x-o] * first[n,c,y-p,x-o]    (in-range positions)
// p outer / o inner, one fp32 accumulator, divided once — the reference's order.  SECOND: template flag.
template <bool SECOND>
__global__ __launch_bounds__(256) void corr_bwd_kernel(const float* __restrict__ other, const float* __restrict__ gout,
                                                       float* __restrict__ gdst, int C, int H, int W) {
#pragma clang fp contract(off)
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int n = blockIdx.z;
    if (x >= W || y >= H) return;
    const int64_t HW = (int64_t)H * W;
    // the 81 gradient weights of this pixel and the validity / offset of the 81 partner positions, once per pixel
    float g[81];
    const float* gn = gout + (int64_t)n * 81 * HW;
#pragma unroll
    for (int p = -4; p <= 4; ++p)
#pragma unroll
        for (int o = -4; o <= 4; ++o) {
            const int op = (p + 4) * 9 + (o + 4);
            const int yy = SECOND ? y - p : y + p, xx = SECOND ? x - o : x + o;
            const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
            // FIRST: gradOut at (y,x), partner second[y+p,x+o] (zero outside); SECOND: both taken at (y-p, x-o)
            const int64_t gpos = SECOND ? (ok ? (int64_t)yy * W + xx : 0) : (int64_t)y * W + x;
            const float v = gn[(int64_t)op * HW + gpos];
            g[op] = ok ? v : 0.0f;
        }
    for (int c = 0; c < C; ++c) {
        const float* oc = other + ((int64_t)n * C + c) * HW;
        float sum = 0.0f;
#pragma unroll
        for (int p = -4; p <= 4; ++p)
#pragma unroll
            for (int o = -4; o <= 4; ++o) {
                const int yy = SECOND ? y - p : y + p, xx = SECOND ? x - o : x + o;
                const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
                const float v = oc[ok ? (int64_t)yy * W + xx : 0];
                if (ok) sum += g[(p + 4) * 9 + (o + 4)] * v;
            }
        gdst[((int64_t)n * C + c) * HW + (int64_t)y * W + x] = sum / (float)C;
    }
}

extern "C" int fldr_correlation_bwd(const float* first, const float* second, const float* grad_out, float* grad_first_or_null,
                                    float* grad_second_or_null, int N, int C, int H, int W, fldr_stream_t stream) {
    FLDR_CHECK_ARG(first && second && grad_out && (grad_first_or_null || grad_second_or_null) && N > 0 && C > 0 && H > 0 && W > 0);
    dim3 grid(fldr_cdiv(W, 64), fldr_cdiv(H, 4), N);
    if (grad_first_or_null)
        hipLaunchKernelGGL(corr_bwd_kernel<false>, grid, dim3(256), 0, fldr_s(stream), second, grad_out, grad_first_or_null, C, H, W);
    if (grad_second_or_null)
        hipLaunchKernelGGL(corr_bwd_kernel<true>, grid, dim3(256), 0, fldr_s(stream), first, grad_out, grad_second_or_null, C, H, W);
    FLDR_LAUNCH_RET();
}
