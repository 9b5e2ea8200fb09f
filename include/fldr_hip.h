/*
 * fldr_hip.h — C ABI of libfldr_hip.so, the MI355X (gfx950) kernels behind the
 * fLDRnet per-frame-pair inference path.
 *
 * Every entry point is `extern "C"`, takes plain device pointers and sizes, enqueues
 * work on the HIP stream it is given (NULL = the null stream) and returns 0 on success or a
 * negative FLDR_E_* / positive hipError_t code; nothing throws, nothing synchronises, nothing
 * allocates (callers own all buffers, as in the reference: softSplat.py:231-234,
 * correlation.py:297-305).  All tensors are contiguous NCHW fp32 unless noted.
 *
 * Each declaration cites the reference interface (file:line in visinf/fldr-vfi) it replaces.
 * The reference binds its kernels through CuPy (`cupy.RawModule(...).get_function`,
 * softSplat.py:215-218); the binding a maintainer would add instead is the ctypes stub in
 * INTEGRATION.md (and fldr-vfi_amd/fldr_hip.py is exactly that stub).
 */
#ifndef FLDR_HIP_H
#define FLDR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLDR_VERSION 105          /* major*10000 + minor*100 + patch; 102: fldr_pca_level.raw_ws (48-byte elements), fldr_sizeof_desc(3..5);
                                     103: fldr_range_status is 0 / 1 again, the ring status has its own entry (fldr_ring_status),
                                     fldr_enc1_fused; 104: fldr_dec23_prepack / fldr_dec23_synth (with the rounded 8-bit frame as a
                                     third output form); 105: fldr_status_word (status readable without synchronising; frames after a ring fault are NaN)
                                     — a caller built against an older header must be rebuilt */

#define FLDR_E_ARG   (-1)         /* bad argument (null pointer, non-positive size, unsupported shape) */
#define FLDR_E_SHAPE (-2)         /* shape constraint violated (e.g. H,W not multiples of 8 for the PCA) */
#define FLDR_E_STATUS (-3)        /* the status block of the device could not be allocated / bound (fldr_status_word) */

/* The product library is built with -fvisibility=hidden: the functions below are its whole dynamic symbol table. */
#define FLDR_API __attribute__((visibility("default")))

typedef void* fldr_stream_t;      /* hipStream_t; torch.cuda.current_stream().cuda_stream */

FLDR_API int         fldr_version(void);
FLDR_API const char* fldr_error_string(int code);

/* ------------------------------------------------------------------------------------------
 * Softmax splatting — replaces softSplat.py.
 * ------------------------------------------------------------------------------------------ */

/* kernel_Softsplat_updateOutput (softSplat.py:12-52) as launched by _FunctionSoftsplat.forward
 * (softSplat.py:222-258): out[n,c,y',x'] += in[n,c,y,x] * bilinear weight of (x+fx, y+fy).
 * `out_zeroed` must be zero on entry (the reference allocates it with new_zeros, :234).
 * flow is [N,2,H,W], channel 0 = x displacement, 1 = y displacement, in pixels. */
FLDR_API int fldr_softsplat_fwd(const float* in, const float* flow, float* out_zeroed,
                       int N, int C, int H, int W, fldr_stream_t stream);

/* FunctionSoftsplat (softSplat.py:320-352) end to end.  mode: 0 summation, 1 average, 2 linear,
 * 3 softmax.  metric is [N,1,H,W] or NULL (softmax with NULL metric = weight 1, :335-336).
 * scratch: N*(C+1)*H*W floats (the (C+1)-channel accumulator); contents are overwritten.
 * out: [N,C,H,W].  Includes the reference's pre-scale (x+1)/2 (softmax only, :334), the
 * zero-norm -> 1 rule (:346) and the post-scale (x-0.5)*2 (every mode, :349). */
FLDR_API int fldr_softsplat_fused(const float* img, const float* flow, const float* metric_or_null,
                         float* out, float* scratch, int N, int C, int H, int W, int mode,
                         fldr_stream_t stream);

/* The two feature splats of a pyramid level (fLDRnet.py:386-387; one sample each, no metric; mode 0, 1 or 3) with one memset
 * and one normalisation launch: scratch = 2 * (C + 1) * H * W floats, out_spk = packed batch of two (sample 0 = problem a).
 * Identical results to two fldr_softsplat_fused_spk calls. */
FLDR_API int fldr_softsplat_pair_spk(const float* img_a, const float* flow_a, const float* img_b, const float* flow_b, void* out_spk,
                            float* scratch, int C, int H, int W, int mode, fldr_stream_t stream);

/* FunctionSoftsplat of FEATURE MAPS as a deterministic gather (no atomics, no accumulator, no memset, no separate
 * normalisation pass; run-to-run identical results): every destination pixel collects the sources whose bilinear footprint
 * covers it, found through flow bounds per 16x16 source tile — exact for any flow, fast for the smooth flows of video.  Up
 * to two (image, flow) problems of the same shape per call (the two directions of fLDRnet.py:386-387).  C <= 48; meant
 * for maps of at most 4096 tiles of 16x16 (FLDR_E_SHAPE beyond: use fldr_softsplat_fused / _tile for frames).
 * mode: 0 summation, 1 average, 2 linear, 3 softmax (metric may be NULL: weight 1).  ws: fldr_softsplat_gather_ws_floats. */
typedef struct fldr_splat_gather_desc {
    const float* img[2];        /* [N,C,H,W] fp32, channel planes contiguous */
    int64_t      img_bstride[2];/* floats between samples */
    const float* flow[2];       /* [N,2,H,W] */
    int64_t      flow_bstride[2];
    const float* metric[2];     /* [N,1,H,W] contiguous, or NULL */
    float*       out_f32[2];    /* [N,C,H,W] or NULL */
    void*        out_spk[2];    /* packed [N,C,H,W] (fldr_spk_bytes per sample) or NULL */
    float*       ws;
    int32_t      ndir, N, C, H, W, mode;
} fldr_splat_gather_desc;
FLDR_API int64_t fldr_softsplat_gather_ws_floats(int ndir, int N, int H, int W);
FLDR_API int fldr_softsplat_gather(const fldr_splat_gather_desc* desc, fldr_stream_t stream);

/* FunctionSoftsplat (softSplat.py:320-352) with destination-owned tiles and fp64 LDS atomics (csrc/splat_acc64_kernels.hip):
 * every workgroup owns an output tile as fp64 accumulators in LDS, the sources that can reach it (flow-bounds tables, as for
 * the retired fldr_softsplat_tile of the test build) add the reference kernel's fp32 corner products (softSplat.py:40-51) with ds_add_f64, and the
 * normalised tile is written once: no global atomics, no accumulator tensor / memset / normalisation pass; exact for any
 * flow; independent of the summation order to ~1e-16 (the reference's own fp32 atomics are unordered).  One or
 * two problems of the same shape per call (the two image splats of fLDRnet.py:449-450, the two feature splats of :386-387).
 * img: sample n, channel c at img + n*img_bstride + c*img_cstride floats (cstride 0 = H*W); flow [N,2,H,W], samples
 * flow_bstride floats apart (0 = 2*H*W); metric [N,1,H,W] contiguous or NULL; ws: fldr_softsplat_tile_ws_floats(N,H,W)
 * floats per problem (flags bit 0: it already holds a bounds table, e.g. from fldr_splat_bounds_upsampled; bit 2, and
 * automatically for maps of at most 2304 pixels when no table is passed: no tables, every tile walks the whole map, ws unused);
 * out_f32 [N,C,H,W] and / or out_spk (packed, C > 3 only; fldr_spk_bytes per sample). */
/* flags bit 1: ws[0] holds the tables of BOTH problems as written by fldr_splat_bounds_upsampled_pair (ONE launch for the two
 * flows of a pyramid level: pair 1 = the level-0 image splats, problem 0: flow = up(t * flow_01) * mul, problem 1:
 * up((1 - t) * flow_10) * mul, fLDRnet.py:404-405; pair 2 = the feature splats, problem 0: up(flow_10) * mul, problem 1:
 * up(flow_01) * mul, :384-387).  flow_l [N,4,h,w], samples lo_bstride floats apart (0 = 4*h*w); ws: 2 *
 * fldr_softsplat_tile_ws_floats(N,H,W) floats. */
FLDR_API int fldr_splat_bounds_upsampled_pair(const float* flow_l, int64_t lo_bstride, const float* t, int pair, float mul, float* ws,
                                     int N, int h, int w, int H, int W, fldr_stream_t stream);
typedef struct fldr_splat_acc_desc {
    const float* img[2];
    int64_t      img_bstride[2], img_cstride[2];
    const float* flow[2];
    int64_t      flow_bstride[2];
    const float* metric[2];
    float*       ws[2];
    float*       out_f32[2];
    void*        out_spk[2];
    int32_t      nprob, N, C, H, W, mode, flags, reserved;
} fldr_splat_acc_desc;
FLDR_API int fldr_softsplat_acc64(const fldr_splat_acc_desc* desc, fldr_stream_t stream);

/* fldr_softsplat_fused with the result in the split-packed layout of the convolution section (fldr_spk_bytes(C,H,W) bytes
 * per sample) instead of fp32 NCHW: the warped feature maps of fLDRnet.py:386-387 are read by conv_flow1 only. */
FLDR_API int fldr_softsplat_fused_spk(const float* img, const float* flow, const float* metric_or_null, void* out_spk,
                             float* scratch, int N, int C, int H, int W, int mode, fldr_stream_t stream);

/* Floats of ONE flow-bounds table (intervals of the flow per 64x4 block and 256x64 super-block: what selects a destination tile's
 * candidate sources) for N samples of an H x W map: the workspace unit of fldr_splat_bounds_upsampled[_pair] / fldr_softsplat_acc64.
 * (The destination-owned splats of rounds 1-2 that shared these tables — fldr_softsplat_tile* — live in the test build only since
 * round 4: include/fldr_hip_test_hooks.h.) */
FLDR_API int64_t fldr_softsplat_tile_ws_floats(int N, int H, int W);

/* The level-0 image splats (fLDRnet.py:449-450) take flows that are bilinear upsamplings of a low-resolution field
 * (flow_t = F.interpolate(scale * flow_lo, (H, W)) * mul, fLDRnet.py:404-405,419-422).  Their bounds table follows from the
 * low-resolution field alone (interpolation is a convex combination; the intervals are widened by 2e-6 of their magnitude for
 * its roundings), so the pre-pass over the two full-resolution flow planes is not needed:
 *   fldr_splat_bounds_upsampled  fills ws (fldr_softsplat_tile_ws_floats(N,H,W) floats) from flow_lo — sample n at
 *                                flow_lo + n*lo_bstride, [2,h,w] contiguous; scale_mode 0: 1, 1: t[n], 2: 1 - t[n];
 * Any table whose block / super-block intervals contain the flow values of their pixels gives the exact result (the table only
 * selects candidate sources). */
FLDR_API int fldr_splat_bounds_upsampled(const float* flow_lo, int64_t lo_bstride, const float* t_or_null, int scale_mode, float mul,
                                float* ws, int N, int h, int w, int H, int W, fldr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * PWC cost volume — replaces OpticalFlow/correlation.py (forward only).
 * ------------------------------------------------------------------------------------------ */

/* _FunctionCorrelation.forward (correlation.py:294-348; kernels :17-112):
 * out[n,(dy+4)*9+(dx+4),y,x] = (1/C) sum_c a[n,c,y,x] * b[n,c,y+dy,x+dx], zero padded, dy,dx in [-4,4].
 * a, b: [N,C,H,W]; out: [N,81,H,W].  No rearranged NHWC copies are needed (the reference's
 * rbot0/rbot1, :297-298, exist only for its one-block-per-pixel kernel). */
FLDR_API int fldr_correlation_fwd(const float* a, const float* b, float* out,
                         int N, int C, int H, int W, fldr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Backward operators — replace the backward kernels of softSplat.py / correlation.py.
 * ------------------------------------------------------------------------------------------ */

/* _FunctionSoftsplat.backward (softSplat.py:259-318; kernels :54-158): gradients of fldr_softsplat_fwd with respect to its
 * input ([N,C,H,W]) and flow ([N,2,H,W]); either output may be NULL.  One pass produces both. */
FLDR_API int fldr_softsplat_bwd(const float* in, const float* flow, const float* grad_out, float* grad_in_or_null,
                       float* grad_flow_or_null, int N, int C, int H, int W, fldr_stream_t stream);

/* _FunctionCorrelation.backward (correlation.py:350-410; kernels :114-242): gradients of fldr_correlation_fwd with respect
 * to first / second ([N,C,H,W]); grad_out [N,81,H,W]; either output may be NULL. */
FLDR_API int fldr_correlation_bwd(const float* first, const float* second, const float* grad_out, float* grad_first_or_null,
                         float* grad_second_or_null, int N, int C, int H, int W, fldr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Low-dimensional feature projection — replaces pca_comp.to_pca_diff (pca_comp.py:473-528).
 * ------------------------------------------------------------------------------------------ */

/* planes [P,H,W] fp32 (P = 6*B planes in the order c*2+t, fLDRnet.py:146); ev [K,64], mean [64],
 * meanvec [K] fp64 (K <= 16).  y[p*K+k,by,bx] = sum_{i,j} (x[p,8by+i,8bx+j]-mean[8i+j])*ev[k,8i+j]/meanvec[k]
 * in fp64; then the GLOBAL min/max over all P*K*H/8*W/8 values maps y to [-1,1] (pca_comp.py:521-526).
 * out_f32: [P*K,H/8,W/8] (the caller's .float() of fLDRnet.py:146); out_f64_or_null: same shape in
 * fp64 (the function's own return value) when wanted.  minmax_ws: 2 doubles of workspace; on
 * completion it holds {min, max}.  H and W must be multiples of 8 (pca_comp.py:486-487). */
FLDR_API int fldr_pca_project(const float* planes, const double* ev, const double* mean, const double* meanvec,
                     float* out_f32, double* out_f64_or_null, double* minmax_ws,
                     int P, int K, int H, int W, fldr_stream_t stream);

/* The same projection with ONE pass over the planes: the raw fp64 projections are parked in out_f64 (required) and
 * rescaled in place by a streaming kernel that also emits the fp32 cast and, when out_spk is given, its split-packed
 * twin (fldr_spk_bytes(P*K, H/8, W/8) bytes; see the convolution section).  Identical results. */
FLDR_API int fldr_pca_project_stream(const float* planes, const double* ev, const double* mean, const double* meanvec,
                            float* out_f32_or_null, double* out_f64, void* out_spk_or_null, double* minmax_ws,
                            int P, int K, int H, int W, fldr_stream_t stream);

/* All pyramid levels of a forward in two launches (the six to_pca_diff calls of fLDRnet.py:133-146): pass A reduces the
 * per-level min / max, pass B recomputes the projection and emits the fp32 cast and / or its split-packed twin — no fp64
 * intermediate in memory.  Same arithmetic as fldr_pca_project per level (bit-identical for the vector kernel).  `table` comes from
 * fldr_pca_prepack (fldr_pca_table_size(K) doubles: coefficients pixel-major, mean, meanvec and its reciprocals);
 * minmax_ws: 32 * n_levels doubles (every bound on a 128-byte line of its own: the reduction is hardware fp64 atomics at the
 * L2), on completion min of level l at [32 l], max at [32 l + 16].  n_levels <= 8; K in {4, 8, 16}.  K = 16 can run on the
 * fp64 matrix cores instead (the test build's pca-variant hook, include/fldr_hip_test_hooks.h): same arithmetic, pixels summed in another order — equal to the
 * per-level kernels to fp64 rounding, not bit for bit; measured no faster, so not the default. */
typedef struct fldr_pca_level {
    const float* planes;       /* [P, H, W] fp32, 16-byte aligned; H, W multiples of 8 */
    float*       out_f32;      /* [P*K, H/8, W/8] or NULL */
    void*        out_spk;      /* packed [1, P*K, H/8, W/8] (fldr_spk_bytes) or NULL */
    int32_t      P, H, W, reserved;
    double*      raw_ws;       /* P * (H/8) * (W/8) * K doubles of scratch (16-byte aligned) or NULL.  Given: the first pass parks the level's
                                  un-normalised projections there and the second rescales them instead of reading the planes again and
                                  recomputing (worth it on the big levels: 128 bytes per block instead of 256 + 64 K fp64 FMAs); same bits */
} fldr_pca_level;
FLDR_API int64_t fldr_pca_table_size(int K);
FLDR_API int fldr_pca_prepack(const double* ev, const double* mean, const double* meanvec, double* table, int K, fldr_stream_t stream);
FLDR_API int fldr_pca_project_pyramid(const fldr_pca_level* levels, int n_levels, const double* table, int K, double* minmax_ws,
                             fldr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Gathers and resizes — replace DCTVFInet.bwarp and the F.interpolate calls of fLDRnet.py.
 * ------------------------------------------------------------------------------------------ */

/* DCTVFInet.bwarp (fLDRnet.py:546-581): grid = pixel + flo, normalised with 2v/max(S-1,1)-1,
 * F.grid_sample(bilinear, zeros, align_corners=False), times the {0,1} validity mask when
 * withmask != 0 (mask = sampled ones, <0.999 -> 0, else 1).  No grid/ones tensors are built. */
FLDR_API int fldr_bwarp(const float* x, const float* flo, float* out,
               int N, int C, int H, int W, int withmask, fldr_stream_t stream);

/* flowback of fLDRnet.py:474-475: bwarp(sx*x, sf*flo) with per-sample scales chosen by x_mode / flo_mode
 * (0: 1, 1: t[n], 2: 1-t[n]); t is a device array of N floats.  The scaled tensors are never materialised. */
FLDR_API int fldr_bwarp_tscaled(const float* x, const float* flo, float* out, const float* t, int x_mode, int flo_mode,
                       int N, int C, int H, int W, int withmask, fldr_stream_t stream);

/* F.interpolate(mode='bilinear', align_corners=False) from [NC,h,w] to [NC,H,W], result multiplied
 * by `mul` (fLDRnet.py:384-385 with mul = W/w; :419-422 with mul = upscale). */
/* fldr_resize_bilinear for [N,C,h,w] with C <= 8 that also writes the split-packed twin of the result (one group). */
FLDR_API int fldr_resize_bilinear_spk(const float* in, float* out, void* out_spk, int N, int C, int h, int w, int H, int W, float mul,
                             fldr_stream_t stream);
FLDR_API int fldr_resize_bilinear(const float* in, float* out, int NC, int h, int w, int H, int W, float mul,
                         fldr_stream_t stream);
/* fldr_resize_bilinear_spk of a [N,4,h,w] level flow AND fldr_splat_bounds_upsampled_pair(pair 2) of the same flow in one launch
 * (fLDRnet.py:384-387: the upsampled flow and the tables of the two feature splats that follow); both results bit-identical to the
 * two calls.  ws: 2 * fldr_softsplat_tile_ws_floats(N, H, W) floats.  Needs W < 4 w (FLDR_E_SHAPE otherwise: use the two calls). */
FLDR_API int fldr_resize_bilinear_spk_bounds(const float* in, float* out, void* out_spk, float* ws, int N, int h, int w, int H, int W, float mul,
                                    fldr_stream_t stream);

/* Splat metric of fLDRnet.py:442-446: z[n,0,y,x] = mean_c( alpha * |self[n,c,y,x] - bwarp(other, flow)[n,c,y,x]| ).
 * self/other: [N,C,H,W]; flow [N,2,H,W]; z [N,1,H,W]. */
FLDR_API int fldr_zmetric(const float* self_img, const float* other_img, const float* flow, float alpha, float* z,
                 int N, int C, int H, int W, fldr_stream_t stream);

/* Everything of fLDRnet.py:400-479 between the level-0 flow and the UNet input that is not a splat, in one pass over the
 * frame: x`mul` bilinear flow upsampling (:419-422, never materialised), the splat metrics z0 / z1 (:442-446, optional:
 * both pointers or neither), flow_t0 = up(t*flow_01), flow_t1 = up((1-t)*flow_10) (:404-405), flowback_0 / flowback_1
 * (:474-475) and im0_tot / im1_tot (:478-479).  Bit-identical to the sequence fldr_resize_bilinear / fldr_zmetric /
 * fldr_bwarp_tscaled / fldr_bwarp it replaces (same device functions and operation order). */
typedef struct fldr_prep_desc {
    const float* flow_lo;            /* [N,4,h,w]: flow_10 (x,y), flow_01 (x,y) at the level-0 feature resolution */
    const float* I0; const float* I1;/* frames, sample n at I + n*bstride, each [3,H,W] contiguous */
    int64_t i0_bstride, i1_bstride;  /* floats */
    const float* t;                  /* [N] */
    float* z0; float* z1;            /* [N,1,H,W] or NULL */
    float* flow_t0; float* flow_t1; float* flowback_0; float* flowback_1;   /* [N,2,H,W] */
    float* im0_tot; float* im1_tot;  /* [N,3,H,W] */
    int32_t N, h, w, H, W;
    float mul;                       /* upscale factor H/h (the flow is multiplied by it, :420,:422) */
    float z_alpha0, z_alpha1;
    int32_t withmask;                /* not args.outMaskLess */
    float* ws;                       /* workspace: N*h*w*4 floats (the low-resolution flow, channel-interleaved), 16-B aligned */
    int64_t i0_cstride, i1_cstride;  /* floats between the channel planes of I0 / I1; 0 = H*W (contiguous [3,H,W]) */
    int32_t phase;                   /* 0 or 3: everything in one launch; 1: only z0 / z1 + flow_t0 / flow_t1 (what the splats need);
                                        2 | 4 = 6: only flowback_* + im*_tot, reusing the workspace filled by a phase-1 call — so that
                                        the consumer of those planes (enc1) can run right behind their producer */
    int32_t reserved;
} fldr_prep_desc;
FLDR_API int fldr_level0_prep(const fldr_prep_desc* desc, fldr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Convolutions — replace the nn.Conv2d stacks of fLDRnet.py (:44-49, :318-345, :611-617).
 * ------------------------------------------------------------------------------------------ */

#define FLDR_CONV_MAX_SRC 12

/* One convolution whose input is the channel-concatenation of up to 12 sources (torch.cat along
 * dim 1 is never materialised: fLDRnet.py:379, :389, :480, :633, :639).  A source flagged `up2`
 * is stored at half resolution and read through nearest-neighbour x2 upsampling
 * (nn.UpsamplingNearest2d, fLDRnet.py:592,632,638,642).
 * out = act(conv(cat(src...), weight) + bias) [+ residual], act = ReLU when relu != 0.
 * Only the first cout_store output channels are written (fLDRnet.py:380, [:, :4]). */
typedef struct fldr_conv_desc {
    const float* src[FLDR_CONV_MAX_SRC];       /* source s, sample n at src[s] + n*src_bstride[s] */
    int64_t      src_bstride[FLDR_CONV_MAX_SRC];/* batch stride in floats */
    int32_t      src_c[FLDR_CONV_MAX_SRC];     /* channels of source s */
    int32_t      src_up2[FLDR_CONV_MAX_SRC];   /* 1: stored [c, Hin/2, Win/2], read nearest x2 */
    int32_t      n_src;
    const float* wpack;        /* weights prepacked by fldr_conv_prepack */
    const float* bias;         /* [cout] or NULL */
    const float* residual;     /* [N,cout_store,Hout,Wout] added after the activation, or NULL */
    float*       out;          /* [N,cout_store,Hout,Wout] */
    int32_t N, cin, cout, cout_store;
    int32_t Hin, Win;          /* input size (after any x2 upsampling) */
    int32_t Hout, Wout;
    int32_t ksize;             /* 3 (stride 1, pad 1) or 4 (stride 2, pad 1) */
    int32_t stride;
    int32_t relu;
    int32_t precision;         /* 0: fp32 MFMA (exact fp32 products, default); 1: fp16 inputs, fp32 accumulate */
    void*   out_spk;           /* fldr_conv2d only: optional split-packed twin of the output (fldr_spk_* below; the layout
                                  the 3x3 convolutions consume); `out` may then be NULL.  NULL for fldr_conv2d_split. */
    int64_t src_cstride[FLDR_CONV_MAX_SRC];    /* fldr_conv2d_s2_split only: floats between the channel planes of source s;
                                                  0 = Hin*Win (contiguous).  Must be 0 for the other entry points. */
} fldr_conv_desc;

/* Number of floats fldr_conv_prepack writes for a [cout,cin,k,k] weight. */
FLDR_API int64_t fldr_conv_prepack_size(int cout, int cin, int ksize);
/* Repack nn.Conv2d weight [cout,cin,k,k] (fp32, device) into the kernel's layout. */
FLDR_API int fldr_conv_prepack(const float* weight, float* wpack, int cout, int cin, int ksize, fldr_stream_t stream);
FLDR_API int fldr_conv2d(const fldr_conv_desc* desc, fldr_stream_t stream);

/* The same convolution (3x3 / stride 1 only) on the fp16 matrix cores with fp32-equivalent accuracy: every operand is
 * split into fp16 hi + lo halves and each product is formed by three fp16 MFMAs with fp32 accumulation ("3 x fp16
 * split"; measured error vs fp64 <= that of the exact fp32 MFMA chain, see csrc/conv_split_kernels.hip).  Takes the
 * same descriptor; desc->wpack must come from fldr_conv_split_prepack (it also holds the power-of-two weight scale). */
FLDR_API int64_t fldr_conv_split_prepack_size(int cout, int cin);
FLDR_API int fldr_conv_split_prepack(const float* weight, float* wpack, int cout, int cin, fldr_stream_t stream);
FLDR_API int fldr_conv2d_split(const fldr_conv_desc* desc, fldr_stream_t stream);

/* The stride-2 4x4 convolutions (UNet encoders, fLDRnet.py:611-613) with the same 3 x fp16 split: fldr_conv_desc with ksize 4,
 * stride 2, no x2 sources, no residual, cout <= 64; wpack from fldr_conv_s2_prepack; `out` and / or `out_spk`. */
FLDR_API int64_t fldr_conv_s2_prepack_size(int cout, int cin);
FLDR_API int fldr_conv_s2_prepack(const float* weight, float* wpack, int cout, int cin, fldr_stream_t stream);
/* The same stride-2 4x4 convolution reading ONE split-packed source (an encoder reading the previous encoder's packed output:
 * the producer then needs no fp32 copy): desc->src[0] = the packed tensor, src_c[0] = cin (a multiple of 8, <= 64),
 * src_bstride[0] in BYTES; outputs as fldr_conv2d_s2_split.  The MFMA operands are the stored hi / lo halves themselves (what
 * fldr_conv2d_s2_split derives from the unpacked value hi + lo, except in the rare case that lo was rounded up to a whole ulp
 * of hi: same value, other split), so the results agree with that function to fp32 accumulation rounding.  Shapes whose
 * weights do not fit the persistent kernel's LDS return FLDR_E_SHAPE (use fldr_spk_unpack + fldr_conv2d_s2_split). */
FLDR_API int fldr_conv2d_s2_spk(const fldr_conv_desc* desc, fldr_stream_t stream);
/* Two such convolutions of the SAME packed source (same geometry, channel counts, relu; other weights / bias / outputs) in ONE launch:
 * the two 32-channel halves of enc3 (fLDRnet.py:617).  The bits of two fldr_conv2d_s2_spk calls. */
FLDR_API int fldr_conv2d_s2_spk_pair(const fldr_conv_desc* desc0, const fldr_conv_desc* desc1, fldr_stream_t stream);
FLDR_API int fldr_conv2d_s2_split(const fldr_conv_desc* desc, fldr_stream_t stream);

/* Split-packed ("SPK") activations: the layout convolution outputs take when their consumer is another convolution.
 * A logical [N,C,H,W] fp32 tensor is stored as [N][G=ceil(C/8)][hi,lo][H*W][8 x fp16] (x = hi + lo, 22 significant
 * bits; 4 B per element like fp32): the producer splits each value once and the consumer's staging becomes pure
 * LDS-DMA.  fldr_conv2d_spk is the same convolution as fldr_conv2d_split (bit-identical results: same split, same
 * MFMA order) with packed sources, a persistent software-pipelined workgroup per CU, and either or both of an fp32
 * NCHW output (with the optional residual) and a packed output.  Replaces the same nn.Conv2d stacks. */
typedef struct fldr_spk_conv_desc {
    const void*  src[FLDR_CONV_MAX_SRC];        /* packed source s (fldr_spk_pack or a previous out_spk) */
    int64_t      src_bstride[FLDR_CONV_MAX_SRC];/* bytes between samples */
    int32_t      src_c[FLDR_CONV_MAX_SRC];      /* channels; every source but the last must have a multiple of 8 */
    int32_t      src_up2[FLDR_CONV_MAX_SRC];    /* 1: stored at [H/2, W/2], read nearest x2 */
    int32_t      n_src;
    const float* wpack;        /* from fldr_conv_spk_prepack */
    const float* bias;         /* [cout] or NULL */
    const float* residual;     /* added after the activation, or NULL: fp32 [N,cout_store,H,W] (needs out_f32), or — precision bit 1 — a split-packed
                                  tensor of cout_store channels (fldr_spk_bytes per sample): value = hi + lo, i.e. the packed fp32 value up to the
                                  split's rounding of lo (<= 2^-22 relative); cout_store % 4 == 0 */
    float*       out_f32;      /* [N,cout_store,H,W] or NULL */
    void*        out_spk;      /* packed [N, cout_store channels, H, W] (fldr_spk_bytes per sample) or NULL */
    int32_t N, cin, cout, cout_store;
    int32_t H, W;
    int32_t relu;
    int32_t precision;         /* bit 0: 0 = 3 x fp16 split (fp32-equivalent), 1 = hi halves only (plain fp16 inputs); bit 1: `residual` is split-packed */
} fldr_spk_conv_desc;

FLDR_API int64_t fldr_spk_bytes(int C, int H, int W);                       /* bytes of one packed sample */
FLDR_API int fldr_spk_pack(const float* src, int64_t src_bstride_floats, void* dst, int N, int C, int H, int W, fldr_stream_t stream);
FLDR_API int fldr_spk_unpack(const void* src, float* dst, int N, int C, int H, int W, fldr_stream_t stream);   /* hi + lo, tests */
FLDR_API int64_t fldr_conv_spk_prepack_size(int cout, int cin);             /* floats */
FLDR_API int fldr_conv_spk_prepack(const float* weight, float* wpack, int cout, int cin, fldr_stream_t stream);
FLDR_API int fldr_conv2d_spk(const fldr_spk_conv_desc* desc, fldr_stream_t stream);
/* The same convolution (shared wpack / bias / relu / channel counts / precision) over n_levels (<= 8) inputs of different sizes in ONE
 * launch of the ring pipeline — rec_ctx_ds over the pyramid levels (fLDRnet.py:148-162 runs it level by level).  Every entry:
 * N = 1, one packed source; residual / out_f32 / out_spk for all entries or none.  Results are the bits of n_levels separate
 * fldr_conv2d_spk calls. */
FLDR_API int fldr_conv2d_spk_levels(const fldr_spk_conv_desc* descs, int n_levels, fldr_stream_t stream);
FLDR_API int fldr_sizeof_desc(int which);                                   /* 0: sizeof(fldr_conv_desc), 1: fldr_spk_conv_desc, 2: fldr_prep_desc, 3: fldr_pca_level,
                                                                      4: fldr_splat_acc_desc, 5: fldr_splat_gather_desc — binding self-check: a binding compares its own struct sizes */
/* Range status of the fp16 hi/lo splits behind the split-precision convolutions (fp32-equivalent for |x| <= 65504; up to
 * 131008 the excess is kept to fp16 precision; beyond that, and for NaN inputs, values SATURATE to a finite number —
 * never inf / NaN out of finite inputs): returns 1 if that happened on the current device since the last reset, 0 if not,
 * negative on a HIP error.  The remedy for such data is exact fp32 (precision = 0 descriptors).  Synchronises the device. */
FLDR_API int fldr_range_status(int reset);
/* Status of the bounded waits inside the loader / consumer ring of the 3x3 convolutions: the number of waits that expired on
 * the current device since the last reset (never observed; > 0 would mean a wave ran on with operands that had not landed, so the
 * outputs since the last reset must not be trusted — a library fault, not a data problem), 0 = clean, negative on a HIP error
 * (including a failed reset).  Synchronises the device. */
FLDR_API int fldr_ring_status(int reset);
/* The same two conditions WITHOUT a synchronisation, for callers that drive frame after frame and never stop to ask (the reference's
 * CUDA kernels abort the process on a device-side fault, softSplat.py:25-26; this library keeps running, so it must not be silent):
 * *host_words receives a pointer to two ints in pinned host memory, [0] = 1 once an activation was saturated by the fp16 split, [1] = 1
 * once a bounded ring wait expired; the kernels store them with system scope when the event happens, the host may read them at any time
 * (fldr-vfi_amd/fLDRnet.py reads them on entry of every forward: a fault of forward n raises in forward n + 1).  Sticky until the
 * reset forms of fldr_range_status / fldr_ring_status.  In addition, every frame written (fldr_dec23_synth, fldr_dec3_synth*,
 * fldr_synth_tail) after a ring wait expired is NaN (8-bit output form: 0) until that reset — a convolution that ran on with operands
 * that had not landed also writes NaN instead of its result.  First call per device: allocates and binds the block (synchronises; do
 * not make it during a stream capture). */
FLDR_API int fldr_status_word(const volatile int** host_words);

/* ------------------------------------------------------------------------------------------
 * Occlusion softmax + frame synthesis — replaces fLDRnet.py:511-524.
 * ------------------------------------------------------------------------------------------ */

/* occ = softmax(refine[:, 0:6] / T) over channels; out = sum_k w_k occ_k cand_k / sum_k w_k occ_k with
 * w = (1-t, t, 1-t, t, 1-t, t) and cand = (warped0, warped1, im0_tot, im1_tot, I0, I1), all in fp64
 * (T_param is a 1-D double tensor, so the reference's tail is fp64: SURVEY F3).
 * refine [N,6,H,W]; the six candidates [N,3,H,W] with batch strides cand_bstride[k] (floats);
 * t: per-sample fp32 value t[n].  Exactly one of out_f64 / out_f32 may be NULL. */
FLDR_API int fldr_synth_tail(const float* refine, const float* const cand[6], const int64_t cand_bstride[6],
                    const float* t, double T_param, double* out_f64, float* out_f32,
                    int N, int H, int W, fldr_stream_t stream);

/* Fused PCARefineUNet.dec3 (3x3, 16 -> 6, on the nearest-x2 upsampled dec2 output, fLDRnet.py:642-643) + the tail above
 * (fLDRnet.py:511-524): refine_out is never stored.  d2: dec2 output [N,16,H/2,W/2]; weff: dec3 weights repacked by
 * fldr_dec3_prepack ([6,16,3,3] -> 1536 floats of per-phase 2x2 weights); bias [6]; cand/t/T/out as fldr_synth_tail.
 * refine_out_or_null: optional [N,6,H,W] debug/verification output of the logits. H, W even; candidate rows 8-B aligned. */
FLDR_API int fldr_dec3_prepack(const float* weight, float* weff, fldr_stream_t stream);
FLDR_API int fldr_dec3_synth(const float* d2, const float* weff, const float* bias, const float* const cand[6],
                    const int64_t cand_bstride[6], const float* t, double T_param, double* out_f64, float* out_f32,
                    float* refine_out_or_null, int N, int H, int W, fldr_stream_t stream);
/* The same with channel-strided candidates: channel ch of candidate k, sample n at cand[k] + n*cand_bstride[k] +
 * ch*cand_cstride[k] (floats; even). */
FLDR_API int fldr_dec3_synth_strided(const float* d2, const float* weff, const float* bias, const float* const cand[6],
                            const int64_t cand_bstride[6], const int64_t cand_cstride[6], const float* t, double T_param,
                            double* out_f64, float* out_f32, float* refine_out_or_null, int N, int H, int W,
                            fldr_stream_t stream);
/* The same fused operator on dec2's SPLIT-PACKED output (fldr_spk_bytes(16, H/2, W/2) bytes per sample): the phase convolutions run
 * on the fp16 matrix cores with the 3 x fp16 split of the 3x3 convolutions (fp32-equivalent logits; not the bits of the fp32-FMA
 * kernel of fldr_dec3_synth), the fp64 tail is the same code.  wm: fldr_dec3_prepack_spk (fldr_dec3_prepack_spk_size() floats,
 * 16-byte aligned).  The model's default since round 3 (fLDRnet.py:642-643 + :511-524). */
FLDR_API int64_t fldr_dec3_prepack_spk_size(void);
FLDR_API int fldr_dec3_prepack_spk(const float* weight, float* wm, fldr_stream_t stream);
FLDR_API int fldr_dec3_synth_spk(const void* d2_spk, const float* wm, const float* bias, const float* const cand[6],
                        const int64_t cand_bstride[6], const int64_t cand_cstride[6], const float* t, double T_param,
                        double* out_f64, float* out_f32, float* refine_out_or_null, int N, int H, int W, fldr_stream_t stream);

/* dec2 -> dec3 -> softmax / blend in ONE persistent kernel (csrc/dec23_kernels.hip; fLDRnet.py:638-643, 511-524): dec2 = ReLU(conv3x3(
 * cat(nearest-x2(dec1), enc1)) + bias), 48 -> 16 channels at half resolution, is produced tile by tile in LDS by four waves of a
 * persistent workgroup while eight others run fldr_dec3_synth_spk's matrix phases and the fp64 tail on the previous tile: dec2's output
 * (141 MB at 4K) is never written or read back.  dec1_spk: packed 32 channels at [H/4, W/4]; enc1_spk: packed 16 channels at [H/2, W/2] (whole tensors:
 * fldr_spk_bytes per sample); w2pack: fldr_dec23_prepack(dec2.weight [16,48,3,3]) (fldr_dec23_prepack_size floats, 16-byte aligned);
 * w3m: fldr_dec3_prepack_spk(dec3.weight); candidates / t / T / outputs as fldr_dec3_synth_spk.  H, W multiples of 4.  Candidates may be
 * views (per-candidate strides, as I0 / I1 are planes of the frame-pair tensor).  Results agree with fldr_conv2d_spk + fldr_dec3_synth_spk
 * to fp32 accumulation rounding in dec2 (tap-major summation) and fp64 rounding in the tail (unnormalised softmax weights, fma blend:
 * the same quotient).  Exactly one of out_f64 / out_f32 / out_u8: out_u8 [N,3,H_u8,W_u8] is the frame cropped to H_u8 x W_u8 (W_u8
 * even), rounded to 8 bits with fldr_frame_metrics' arithmetic (utils.py:685-688, np.around) straight from the fp64 blend — the
 * uint8-in / uint8-out callers (run_on_your_images.py:100-109) then never write or re-read the 212 MB fp64 frame. */
FLDR_API int64_t fldr_dec23_prepack_size(void);
FLDR_API int fldr_dec23_prepack(const float* dec2_weight, float* wpack, fldr_stream_t stream);
FLDR_API int fldr_dec23_synth(const void* dec1_spk, const void* enc1_spk, const float* w2pack, const float* bias2, const float* w3m, const float* bias3,
                     const float* const cand[6], const int64_t cand_bstride[6], const int64_t cand_cstride[6], const float* t, double T_param,
                     double* out_f64, float* out_f32, uint8_t* out_u8, int H_u8, int W_u8, int N, int H, int W, fldr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Callers either side of the path, on the device (the reference does these on the CPU).
 * ------------------------------------------------------------------------------------------ */

/* frames_u8 [B,2,3,H,W] (I0, I1) -> level0 [B,3,2,Hp,Wp] fp32: x/255*2-1 (run_on_your_images.py:84) with right/bottom
 * reflect padding (main.py:840-849).  Hp, Wp: padded size (multiples of 2^S_tst*8), pad < size. */
FLDR_API int fldr_ingest_u8(const uint8_t* frames_u8, float* level0, int B, int H, int W, int Hp, int Wp, fldr_stream_t stream);

/* One pyramid level: F.interpolate(level0 planes, scale_factor=1/factor, mode='bicubic', align_corners=False)
 * (main.py:855-856), factor a power of two >= 2.  level0 [planes,Hp,Wp] -> level_i [planes,Hp/factor,Wp/factor]. */
FLDR_API int fldr_pyramid_bicubic(const float* level0, float* level_i, int planes, int Hp, int Wp, int factor, fldr_stream_t stream);

/* Both of the above for every level in ONE launch (one read of the uint8 frames, every level written from a staged 64 x 64 tile):
 * levels[i] = [B,3,2,Hp >> i,Wp >> i] for i < n_levels <= 7; Hp, Wp multiples of 2^(n_levels-1) and of 4.  The bits of fldr_ingest_u8 +
 * fldr_pyramid_bicubic (main.py:840-856). */
FLDR_API int fldr_ingest_pyramid_u8(const uint8_t* frames_u8, float* const* levels, int n_levels, int B, int H, int W, int Hp, int Wp,
                           fldr_stream_t stream);

/* main.py:885-911 on the device: crop pred [B,3,Hp,Wp] (fp64 if pred_is_f64 else fp32) to H x W, (x+1)/2 clipped to
 * [0,1] * 255, rounded half-to-even; optionally written as uint8 [B,3,H,W]; when target_u8 [B,3,H,W] is given,
 * sse[b] (zeroed by the caller) accumulates the squared error, so PSNR = 10 log10(255^2 * 3HW / sse[b]). */
FLDR_API int fldr_frame_metrics(const void* pred, int pred_is_f64, const uint8_t* target_u8_or_null, uint8_t* out_u8_or_null,
                       double* sse_zeroed_or_null, int B, int H, int W, int Hp, int Wp, fldr_stream_t stream);

/* utils.ssim_bgr (utils.py:662-669; main.py:911) on the device: SSIM of the Y channel of two uint8 images [B,3,H,W]
 * in cv2 channel order (0 = B, 1 = G, 2 = R) — the rounded frame fldr_frame_metrics writes and the ground truth — with
 * scikit-image's structural_similarity defaults (7x7 uniform window, sample covariance, K1 0.01, K2 0.03, data_range =
 * max - min of Y_pred, map cropped by 3 pixels), all in fp64.  ws: fldr_ssim_y_ws_doubles(B,H,W) doubles; on completion
 * sample b's statistics sit at ws[b * (2 H W + 4) + 2 H W ...] = {min Y_pred, max Y_pred, sum of the SSIM map, 0}:
 * SSIM = sum / ((H - 6) (W - 6)).  H, W >= 7. */
FLDR_API int64_t fldr_ssim_y_ws_doubles(int B, int H, int W);
FLDR_API int fldr_ssim_y_u8(const uint8_t* pred_u8, const uint8_t* target_u8, double* ws, int B, int H, int W, fldr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FLDR_HIP_H */
