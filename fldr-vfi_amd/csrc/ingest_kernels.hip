// Callers either side of the hot path, moved onto the GPU:
//   * ingest: uint8 frames -> [-1,1] fp32, reflect padding to a multiple of 2^S*8 and the direct (non-cascaded)
//     bicubic pyramid that main.py:840-856 / run_on_your_images.py:124-145 build on the CPU before uploading
//     283 MB of fp32 (here: 50 MB of uint8 go up, everything else is produced on the device);
//   * metrics: crop, de-normalise, round-half-even to the 8-bit grid and the squared error against the uint8
//     ground truth (main.py:885-911, utils.py:644-652, 685-688) instead of a 212 MB fp64 download.
#include "common.h"

// frames_u8 [B,T=2,3,H,W] (I0 then I1) -> out [B,3,2,Hp,Wp]: v = u8/255*2-1 (run_on_your_images.py:84),
// right/bottom reflect padding as F.pad(mode='reflect') on the [B,C*T,H,W] view (main.py:848)
__global__ __launch_bounds__(256) void ingest_kernel(const uint8_t* __restrict__ u8, float* __restrict__ out,
                                                     int H, int W, int Hp, int Wp) {
#pragma clang fp contract(off)
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    const int bct = blockIdx.z;                 // ((b*3 + c)*2 + t) in the output
    if (x >= Wp) return;
    const int t = bct & 1, c = (bct >> 1) % 3, b = bct / 6;
    const int sx = x < W ? x : 2 * (W - 1) - x;
    const int sy = y < H ? y : 2 * (H - 1) - y;
    const uint8_t v = u8[(((int64_t)(b * 2 + t) * 3 + c) * H + sy) * W + sx];
    float f = (float)v / 255.0f;
    f = f * 2.0f;
    out[((int64_t)bct * Hp + y) * Wp + x] = f - 1.0f;
}

// Level i of the pyramid from level 0: F.interpolate(scale_factor=2^-i, mode='bicubic', align_corners=False) (main.py:855).
// For integer down-factors s the source position (dst+0.5)*s-0.5 always has fraction 0.5, so the four cubic
// convolution weights (A = -0.75) are the constants (-3/32, 19/32, 19/32, -3/32); taps are index-clamped.
__global__ __launch_bounds__(256) void pyramid_bicubic_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                              int Hs, int Ws, int s) {
#pragma clang fp contract(off)
    const int Hd = Hs / s, Wd = Ws / s;
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    const int plane = blockIdx.z;
    if (x >= Wd) return;
    const float c0 = -0.09375f, c1 = 0.59375f;
    const int fx = x * s + s / 2 - 1, fy = y * s + s / 2 - 1;       // floor of the source position
    const float* p = src + (int64_t)plane * Hs * Ws;
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int iy = min(max(fy - 1 + j, 0), Hs - 1);
        const float* row = p + (int64_t)iy * Ws;
        const float v0 = row[min(max(fx - 1, 0), Ws - 1)], v1 = row[min(max(fx, 0), Ws - 1)];
        const float v2 = row[min(max(fx + 1, 0), Ws - 1)], v3 = row[min(max(fx + 2, 0), Ws - 1)];
        r[j] = ((v0 * c0 + v1 * c1) + v2 * c1) + v3 * c0;
    }
    dst[((int64_t)plane * Hd + y) * Wd + x] = ((r[0] * c0 + r[1] * c1) + r[2] * c1) + r[3] * c0;
}

// ---- ingest + the whole pyramid in ONE pass (round 4) ---------------------------------------------------------------------
// The two kernels above move 262 MB (ingest: one byte load and one 4-byte store per thread, >= 124 us at 4K) and then read level 0
// once per pyramid level (five launches, ~2.3 x 212 MB).  For power-of-two factors s <= 64 the 4x4 bicubic footprint of a level-i
// pixel lies inside one s x s cell of level 0 (taps at s x + s/2 - 2 ... s x + s/2 + 1), except for level 1 (one pixel of halo).
// So a workgroup stages ONE 64 x 64 tile (+ halo) of one level-0 plane in LDS — uint8 in, four pixels per dword load, normalised
// and reflect-padded on the way — writes it out with 16-byte stores and produces the tile's pixels of EVERY level from LDS:
// 50 MB in, 283 MB out, one launch.  Same expressions and operation order as ingest_kernel / pyramid_bicubic_kernel: the same bits.
#define IP_T 64                               // level-0 tile edge
#define IP_LW 72                              // LDS row: columns X0 - 4 ... X0 + 67 (16-byte aligned pieces)
#define IP_LH 66                              // LDS rows: Y0 - 1 ... Y0 + 64
#define IP_MAX_LEVELS 7                       // level 0 + factors 2 ... 64
struct IpArgs {
    const uint8_t* u8;
    float* lv[IP_MAX_LEVELS];
    int n_levels, H, W, Hp, Wp;
};
__device__ __forceinline__ float ip_norm(uint8_t v) {
#pragma clang fp contract(off)
    float f = (float)v / 255.0f;
    f = f * 2.0f;
    return f - 1.0f;
}
__global__ __launch_bounds__(256) void ingest_pyramid_kernel(IpArgs a) {
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) float tile[IP_LH * IP_LW];
    const int tid = threadIdx.x;
    const int X0 = blockIdx.x * IP_T, Y0 = blockIdx.y * IP_T;
    const int bct = blockIdx.z;                                          // ((b*3 + c)*2 + t) in the output
    const int t = bct & 1, c = (bct >> 1) % 3, b = bct / 6;
    const uint8_t* src = a.u8 + ((int64_t)(b * 2 + t) * 3 + c) * a.H * a.W;
    // stage: one item = 4 columns of one row; level-0 coordinates clamped to the padded image (the pyramid's tap clamp), then
    // reflected into the frame (F.pad(mode='reflect'), main.py:848)
    for (int e = tid; e < IP_LH * (IP_LW / 4); e += 256) {
        const int r = e / (IP_LW / 4), q = e - r * (IP_LW / 4);
        const int yc = min(max(Y0 - 1 + r, 0), a.Hp - 1);
        const int sy = yc < a.H ? yc : 2 * (a.H - 1) - yc;
        const int x = X0 - 4 + 4 * q;
        float4 v;
        if (x >= 0 && x + 3 < a.W && !(a.W & 3) && !((uintptr_t)a.u8 & 3)) {       // aligned dword of four frame pixels
            const uint32_t w = *reinterpret_cast<const uint32_t*>(src + (int64_t)sy * a.W + x);
            v = make_float4(ip_norm((uint8_t)w), ip_norm((uint8_t)(w >> 8)), ip_norm((uint8_t)(w >> 16)), ip_norm((uint8_t)(w >> 24)));
        } else {
            float f[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int xc = min(max(x + k, 0), a.Wp - 1);
                const int sx = xc < a.W ? xc : 2 * (a.W - 1) - xc;
                f[k] = ip_norm(src[(int64_t)sy * a.W + sx]);
            }
            v = make_float4(f[0], f[1], f[2], f[3]);
        }
        *reinterpret_cast<float4*>(tile + r * IP_LW + 4 * q) = v;
    }
    __syncthreads();
    // level 0: the tile itself (rows 1 .. 64, columns 4 .. 67 of the LDS image), 16-byte stores
    {
        float* out = a.lv[0] + (int64_t)bct * a.Hp * a.Wp;
        for (int e = tid; e < IP_T * (IP_T / 4); e += 256) {
            const int r = e / (IP_T / 4), q = e - r * (IP_T / 4);
            const int y = Y0 + r, x = X0 + 4 * q;
            if (y < a.Hp && x < a.Wp)                                    // (Wp % 4 == 0: host-checked)
                *reinterpret_cast<float4*>(out + (int64_t)y * a.Wp + x) = *reinterpret_cast<const float4*>(tile + (r + 1) * IP_LW + 4 + 4 * q);
        }
    }
    // levels 1 ..: pyramid_bicubic_kernel's arithmetic on the staged tile (cubic convolution, A = -0.75, fraction 0.5)
    const float c0 = -0.09375f, c1 = 0.59375f;
    for (int l = 1; l < a.n_levels; ++l) {
        const int s = 1 << l, n = IP_T >> l;                             // n x n outputs of this level in the tile
        const int Hd = a.Hp >> l, Wd = a.Wp >> l;
        float* out = a.lv[l] + (int64_t)bct * Hd * Wd;
        for (int e = tid; e < n * n; e += 256) {
            const int oy = e / n, ox = e - oy * n;
            const int gy = (Y0 >> l) + oy, gx = (X0 >> l) + ox;
            if (gy >= Hd || gx >= Wd) continue;
            const int lr = oy * s + s / 2 - 1, lc = ox * s + s / 2 + 2;  // LDS row / column of tap (0, 0): fy - 1 - (Y0 - 1), fx - 1 - (X0 - 4)
            float r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* row = tile + (lr + j) * IP_LW + lc;
                r[j] = ((row[0] * c0 + row[1] * c1) + row[2] * c1) + row[3] * c0;
            }
            out[(int64_t)gy * Wd + gx] = ((r[0] * c0 + r[1] * c1) + r[2] * c1) + r[3] * c0;
        }
    }
}

// frames_u8 [B,2,3,H,W] -> levels[i] = [B,3,2,Hp >> i,Wp >> i] fp32 for i < n_levels (1 .. 7): fldr_ingest_u8 + fldr_pyramid_bicubic for every
// level (main.py:840-856) in one launch, the same bits.  Hp, Wp multiples of 2^(n_levels-1) and of 4; pad < size.
extern "C" int fldr_ingest_pyramid_u8(const uint8_t* frames_u8, float* const* levels, int n_levels, int B, int H, int W, int Hp, int Wp,
                                      fldr_stream_t stream) {
    FLDR_CHECK_ARG(frames_u8 && levels && n_levels >= 1 && n_levels <= IP_MAX_LEVELS && B > 0 && H > 1 && W > 1 && Hp >= H && Wp >= W);
    if (Hp - H >= H || Wp - W >= W) return FLDR_E_SHAPE;            // reflect padding needs pad < size
    if ((Wp & 3) || (Hp & ((1 << (n_levels - 1)) - 1)) || (Wp & ((1 << (n_levels - 1)) - 1))) return FLDR_E_SHAPE;
    IpArgs a;
    a.u8 = frames_u8; a.n_levels = n_levels; a.H = H; a.W = W; a.Hp = Hp; a.Wp = Wp;
    for (int i = 0; i < IP_MAX_LEVELS; ++i) { a.lv[i] = i < n_levels ? levels[i] : nullptr; FLDR_CHECK_ARG(i >= n_levels || levels[i]); }
    dim3 grid(fldr_cdiv(Wp, IP_T), fldr_cdiv(Hp, IP_T), B * 6);
    hipLaunchKernelGGL(ingest_pyramid_kernel, grid, dim3(256), 0, fldr_s(stream), a);
    FLDR_LAUNCH_RET();
}

extern "C" int fldr_ingest_u8(const uint8_t* frames_u8, float* level0, int B, int H, int W, int Hp, int Wp,
                              fldr_stream_t stream) {
    FLDR_CHECK_ARG(frames_u8 && level0 && B > 0 && H > 1 && W > 1 && Hp >= H && Wp >= W);
    if (Hp - H >= H || Wp - W >= W) return FLDR_E_SHAPE;            // reflect padding needs pad < size
    dim3 grid(fldr_cdiv(Wp, 256), Hp, B * 6);
    hipLaunchKernelGGL(ingest_kernel, grid, dim3(256), 0, fldr_s(stream), frames_u8, level0, H, W, Hp, Wp);
    FLDR_LAUNCH_RET();
}

extern "C" int fldr_pyramid_bicubic(const float* level0, float* level_i, int planes, int Hp, int Wp, int factor,
                                    fldr_stream_t stream) {
    FLDR_CHECK_ARG(level0 && level_i && planes > 0 && Hp > 0 && Wp > 0 && factor >= 2);
    if (Hp % factor || Wp % factor || (factor & (factor - 1))) return FLDR_E_SHAPE;
    dim3 grid(fldr_cdiv(Wp / factor, 256), Hp / factor, planes);
    hipLaunchKernelGGL(pyramid_bicubic_kernel, grid, dim3(256), 0, fldr_s(stream), level0, level_i, Hp, Wp, factor);
    FLDR_LAUNCH_RET();
}

// pred [B,3,Hp,Wp] (fp64 or fp32, [-1,1]) cropped to H x W -> rounded 8-bit value (np.around: half to even),
// optional uint8 image out_u8 [B,3,H,W]; sse[b] += sum (target - rounded)^2 against target_u8 [B,3,H,W] when given.
template <typename T>
__global__ __launch_bounds__(256) void metrics_kernel(const T* __restrict__ pred, const uint8_t* __restrict__ target,
                                                      uint8_t* __restrict__ out_u8, double* __restrict__ sse,
                                                      int H, int W, int Hp, int Wp) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    const int bc = blockIdx.z;
    double e = 0.0;
    if (x < W) {
        double v = ((double)pred[((int64_t)bc * Hp + y) * Wp + x] + 1.0) / 2.0;        // utils.py:685-688
        v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
        const double q = rint(v * 255.0);                                             // np.around (main.py:894)
        const int64_t o = ((int64_t)bc * H + y) * W + x;
        if (out_u8) out_u8[o] = (uint8_t)q;
        if (target) { const double d = (double)target[o] - q; e = d * d; }
    }
    if (target) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off);
        __shared__ double part[4];
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = e;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(sse + bc / 3, (part[0] + part[1]) + (part[2] + part[3]));
    }
}

extern "C" int fldr_frame_metrics(const void* pred, int pred_is_f64, const uint8_t* target_u8_or_null, uint8_t* out_u8_or_null,
                                  double* sse_zeroed_or_null, int B, int H, int W, int Hp, int Wp, fldr_stream_t stream) {
    FLDR_CHECK_ARG(pred && B > 0 && H > 0 && W > 0 && Hp >= H && Wp >= W);
    FLDR_CHECK_ARG((target_u8_or_null == nullptr) == (sse_zeroed_or_null == nullptr));
    dim3 grid(fldr_cdiv(W, 256), H, B * 3);
    if (pred_is_f64) hipLaunchKernelGGL(metrics_kernel<double>, grid, dim3(256), 0, fldr_s(stream), (const double*)pred, target_u8_or_null, out_u8_or_null, sse_zeroed_or_null, H, W, Hp, Wp);
    else             hipLaunchKernelGGL(metrics_kernel<float>, grid, dim3(256), 0, fldr_s(stream), (const float*)pred, target_u8_or_null, out_u8_or_null, sse_zeroed_or_null, H, W, Hp, Wp);
    FLDR_LAUNCH_RET();
}


// ------------------------------------------------------------------------------------------------
// SSIM of the Y channel (utils.ssim_bgr, utils.py:662-669, called at main.py:911) on the device.
// pred / target: uint8 [B,3,H,W], channel 0 = B, 1 = G, 2 = R (cv2 order, utils.py:237-251).  Y = 0.2568 R + 0.5041 G +
// 0.0979 B + 16 in fp64 (utils.py:690-711); scikit-image's structural_similarity defaults: 7x7 uniform window, sample
// covariance, K1 = 0.01, K2 = 0.03, data_range = max(Y_pred) - min(Y_pred), mean over the map cropped by 3 pixels.
// Pass 1 writes both Y planes (fp64) and reduces min / max of Y_pred; pass 2 evaluates the map from LDS tiles and
// accumulates its sum.  ws per sample: [2 H W] Y planes, then {min, max, sum, unused}.
// ------------------------------------------------------------------------------------------------
__global__ void ssim_init_kernel(double* ws, int64_t stride, int B, int64_t planes) {
    if ((int)threadIdx.x < B) { double* s = ws + threadIdx.x * stride + planes; s[0] = 1.0e300; s[1] = -1.0e300; s[2] = 0.0; s[3] = 0.0; }
}

__global__ __launch_bounds__(256) void ssim_y_kernel(const uint8_t* __restrict__ pred, const uint8_t* __restrict__ target,
                                                     double* __restrict__ ws, int64_t stride, int64_t HW) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    double yp = 0.0, lo = 1.0e300, hi = -1.0e300;
    if (i < HW) {
        const uint8_t* p = pred + (int64_t)b * 3 * HW + i;
        const uint8_t* t = target + (int64_t)b * 3 * HW + i;
        // np.dot of the (R, G, B) row with T[0]: ((R t0 + G t1) + B t2), then + 16 — products and sums in fp64
        yp = ((double)p[2 * HW] * 0.256788235294118 + (double)p[HW] * 0.504129411764706) + (double)p[0] * 0.097905882352941 + 16.0;
        const double yt = ((double)t[2 * HW] * 0.256788235294118 + (double)t[HW] * 0.504129411764706) + (double)t[0] * 0.097905882352941 + 16.0;
        double* w = ws + (int64_t)b * stride;
        w[i] = yt;                                                   // plane 0: Y_true, plane 1: Y_pred
        w[HW + i] = yp;
        lo = hi = yp;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ol = __shfl_xor(lo, off), oh = __shfl_xor(hi, off);
        lo = ol < lo ? ol : lo; hi = oh > hi ? oh : hi;
    }
    if ((threadIdx.x & 63) == 0) {
        double* s = ws + (int64_t)b * stride + 2 * HW;
        (void)__hip_atomic_fetch_min(s, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        (void)__hip_atomic_fetch_max(s + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

#define SSIM_TW 32
#define SSIM_TH 8
__global__ __launch_bounds__(256) void ssim_map_kernel(double* __restrict__ ws, int64_t stride, int H, int W) {
#pragma clang fp contract(off)
    __shared__ double tx[SSIM_TH + 6][SSIM_TW + 6], ty[SSIM_TH + 6][SSIM_TW + 6];
    const int b = blockIdx.z;
    const int64_t HW = (int64_t)H * W;
    const double* X = ws + (int64_t)b * stride;
    const double* Y = X + HW;
    double* stat = ws + (int64_t)b * stride + 2 * HW;
    const int x0 = blockIdx.x * SSIM_TW + 3, y0 = blockIdx.y * SSIM_TH + 3;       // first output pixel of the tile (interior only)
    for (int e = threadIdx.x; e < (SSIM_TH + 6) * (SSIM_TW + 6); e += 256) {
        const int r = e / (SSIM_TW + 6), c = e - r * (SSIM_TW + 6);
        const int gy = min(y0 - 3 + r, H - 1), gx = min(x0 - 3 + c, W - 1);      // clamped taps are only read by masked outputs
        tx[r][c] = X[(int64_t)gy * W + gx];
        ty[r][c] = Y[(int64_t)gy * W + gx];
    }
    __syncthreads();
    const int lx = threadIdx.x & (SSIM_TW - 1), ly = threadIdx.x / SSIM_TW;
    const int ox = x0 + lx, oy = y0 + ly;
    double s = 0.0;
    if (ox < W - 3 && oy < H - 3) {
        double sx = 0.0, sy = 0.0, sxx = 0.0, syy = 0.0, sxy = 0.0;
        for (int r = 0; r < 7; ++r)
#pragma unroll
            for (int c = 0; c < 7; ++c) {
                const double a = tx[ly + r][lx + c], d = ty[ly + r][lx + c];
                sx += a; sy += d; sxx += a * a; syy += d * d; sxy += a * d;
            }
        const double R = stat[1] - stat[0];
        const double C1 = (0.01 * R) * (0.01 * R), C2 = (0.03 * R) * (0.03 * R);
        const double ux = sx / 49.0, uy = sy / 49.0, uxx = sxx / 49.0, uyy = syy / 49.0, uxy = sxy / 49.0;
        const double cn = 49.0 / 48.0;
        const double vx = cn * (uxx - ux * ux), vy = cn * (uyy - uy * uy), vxy = cn * (uxy - ux * uy);
        s = ((2.0 * ux * uy + C1) * (2.0 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(stat + 2, (part[0] + part[1]) + (part[2] + part[3]));
}

extern "C" int64_t fldr_ssim_y_ws_doubles(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return FLDR_E_ARG;
    return (int64_t)B * (2ll * H * W + 4);
}

extern "C" int fldr_ssim_y_u8(const uint8_t* pred_u8, const uint8_t* target_u8, double* ws, int B, int H, int W,
                              fldr_stream_t stream) {
    FLDR_CHECK_ARG(pred_u8 && target_u8 && ws && B > 0 && B <= 64);
    if (H < 7 || W < 7) return FLDR_E_SHAPE;                            // the 7x7 window must fit (skimage raises too)
    const int64_t HW = (int64_t)H * W, stride = 2 * HW + 4;
    hipStream_t s = fldr_s(stream);
    hipLaunchKernelGGL(ssim_init_kernel, dim3(1), dim3(64), 0, s, ws, stride, B, 2 * HW);
    hipLaunchKernelGGL(ssim_y_kernel, dim3(fldr_cdiv(HW, 256), B), dim3(256), 0, s, pred_u8, target_u8, ws, stride, HW);
    hipLaunchKernelGGL(ssim_map_kernel, dim3(fldr_cdiv(W - 6, SSIM_TW), fldr_cdiv(H - 6, SSIM_TH), B), dim3(256), 0, s, ws, stride, H, W);
    FLDR_LAUNCH_RET();
}
