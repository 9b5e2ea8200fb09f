/* Tuning and cross-check hooks of libfldr_hip_test.so (the build with -DFLDR_TEST_HOOKS; `make -C fldr-vfi_amd/csrc hooks`).
 * NOT part of the integration ABI: the product library libfldr_hip.so exports none of these, and the retired kernel generations
 * some of them select (the barrier-pipeline 3x3 convolution, the LDS-f32-atomic tile splat) are not even compiled into it.
 * Used by tests/ and tools/ through fldr_hip.test_hooks(). */
#ifndef FLDR_HIP_TEST_HOOKS_H
#define FLDR_HIP_TEST_HOOKS_H
#include "fldr_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

FLDR_API int fldr_debug_pca_variant(int v);                                /* K = 16: 0 (default) the scalar-fed vector kernel (bit-identical to the per-level kernels), 1 fp64 matrix cores; other: query */
FLDR_API int fldr_debug_pca_workgroups(int v);                             /* tuning hook: persistent workgroups of the two pyramid passes (default 512); 0: query */
FLDR_API int fldr_debug_s2_persistent(int v);                                /* tuning hook: 1 (default) persistent-workgroup kernel where the weights fit, 0 per-tile kernel; < 0 query */
FLDR_API int fldr_debug_s2_xshift(int v);                                  /* tuning hook: left shift (output columns) of the persistent stride-2 kernel's tile grid; -1 (default): 15 on wide images */
FLDR_API int fldr_debug_s2_vec4(int v);                                   /* tuning hook: 1 (default) 16-byte staging loads in the persistent stride-2 encoder where the geometry allows, 0 never; other: query.  Bit-identical results */
FLDR_API int fldr_debug_dec3_xshift(int v);                                /* tuning hook: left shift (low-resolution columns) of fldr_dec3_synth's tile grid; -1 (default): 16 on wide frames */
FLDR_API int fldr_debug_splat_quad(int v);                                 /* image splat walk of fldr_softsplat_acc64: 1 (default) runs of four pixels per thread where W % 4 == 0 and the planes are 16-byte aligned, 0 one pixel per item; other: query */
FLDR_API int fldr_debug_dec3_xcd(int v);                                   /* tile order of fldr_dec3_synth: 1 (default) contiguous tile ranges per XCD, 0 row-major round-robin; other: query.  Identical results */
FLDR_API int fldr_debug_spk_small_units(int v);                              /* tuning hook: launches of <= v units run as 16-channel sub-groups (default 96; -1: never; 0: query) */
FLDR_API int fldr_debug_spk_wgs_per_xcd(int v);                             /* tuning hook: persistent workgroups per XCD (default 32) */
FLDR_API int fldr_debug_spk_variant(int v);                                 /* pipeline of fldr_conv2d_spk: 1 (default) loader / consumer ring, 0 barrier pipeline; -1: query.  Bit-identical results */
FLDR_API int fldr_debug_ring_resident(int v);                               /* 1: resident-weight ring (5 slots, two fills in flight) for 16-output-channel launches of <= 4 input chunks; 0 (default): streamed weights; other: query */
FLDR_API int fldr_debug_ring32(int v);                                      /* the 32x32x16 ring kernel for 64 / 96 output channels (packed output, no residual): 1 (default) where its cost model says it fills the workgroup rounds better, 2 wherever it applies, 0 never; other: query */
FLDR_API int fldr_debug_ring_consumers(int v);                              /* tuning hook of the ring pipeline: 8 (default; two consumer waves per SIMD) or 4 consumer waves; other: query */
FLDR_API int fldr_debug_corr_variant(int v);                                /* cost volume staging: 1 (default) LDS-DMA double buffer where W % 4 == 0, 0 synchronous; other: query.  Bit-identical results */
FLDR_API int fldr_debug_corr_xcd(int v);                                    /* tile order of the LDS-DMA cost-volume kernel: 1 (default) contiguous tile ranges per XCD, 0 row-major; other: query.  Identical results */
FLDR_API int fldr_debug_corr_chunk(int v);                                  /* channels per staged chunk of the LDS-DMA cost-volume kernel: 8 (default) or 16; other: query */
FLDR_API int fldr_debug_ring_tile_width(int v);                             /* tuning hook of the ring pipeline: 0 (default) automatic per launch, 16 / 32 forced; other: query.  Bit-identical results */
FLDR_API int fldr_debug_ring_spin_limit(int v);                             /* polls a bounded ring wait makes before it expires (default 2^21; < 0: restore the default); 0 makes every wait that is not satisfied at once expire: exercises the fault path */
FLDR_API int64_t fldr_debug_ringrow_pack_floats(int cout, int cin);         /* round-6 experiment (ring item = 32 channels x one kernel row, no pad tap): floats of its weight section, < 0: shape not covered */
FLDR_API int fldr_debug_ringrow_prepack(const float* weight, const float* wpack, float* wrow, int cout, int cin, fldr_stream_t stream);
FLDR_API int fldr_debug_conv2d_ringrow(const fldr_spk_conv_desc* desc, const float* wrow, fldr_stream_t stream);
FLDR_API int fldr_debug_busy_partner(float* out, int workgroups, int lds_bytes, int iters, int kind, fldr_stream_t stream);   /* concurrency tests: `workgroups` x 256 threads that hold `lds_bytes` of LDS each and loop `iters` times over kind 0 s_sleep, 1 matrix instructions, 2 vector FMAs, 3 scalar adds, 4 LDS reads; out: workgroups * 256 floats (csrc/test_partner_kernels.hip) */
FLDR_API int fldr_debug_ring_timeouts(void);                                /* number of bounded ring waits that expired since load (0 unless a kernel misbehaved); synchronises */
FLDR_API int fldr_debug_splat_tile_variant(int v);                          /* fldr_softsplat_tile: 1 (default) claim-and-add bands, 0 the LDS-f32-atomic tiles; other: query */
FLDR_API int fldr_debug_pca_variant(int v);                                 /* fldr_pca_project_pyramid: 0 (default) vector fp64 kernel, 1 fp64 matrix-core kernel; other: query */
FLDR_API int fldr_debug_pca_workgroups(int v);                              /* persistent workgroups of the pyramid PCA (0: query) */
FLDR_API int fldr_debug_s2_persistent(int v);                               /* stride-2 encoders: 1 (default) persistent kernel, 0 per-tile kernel */
FLDR_API int fldr_debug_s2_xshift(int v);                                   /* tile-grid shift of the persistent stride-2 kernel (output columns; -1: default) */
FLDR_API int fldr_debug_s2_vec4(int v);                                     /* 16-byte staging loads of the persistent stride-2 kernel: 1 (default) / 0 */
FLDR_API int fldr_debug_s2_dma(int v);                                      /* packed-source stride-2 encoders of 17..32 output channels: 1 (default) the LDS-DMA kernel, 0 the register-staged kernel; other: query */
FLDR_API int fldr_debug_dec3_xshift(int v);                                 /* tile-grid shift of dec3_synth (low-resolution columns; -1: default) */
FLDR_API int fldr_debug_splat_group_fold(int v);                            /* fldr_softsplat_acc64, > 3 channels: 1 all channel groups of a tile in one workgroup where the map is large enough, 0 (default) one group per workgroup; other: query.  Same results */
FLDR_API int fldr_debug_conv_occupancy(int* out4);                          /* occupancy query of the fp32-MFMA convolution kernels */

/* The destination-owned splats of rounds 1-2 (csrc/splat_tile_kernels.hip: claim-and-add bands without atomics; the LDS-f32-atomic
 * tiles behind fldr_debug_splat_tile_variant(0)), retired from the product in round 4 — every splat of the forward and
 * FunctionSoftsplat run on fldr_softsplat_acc64.  Kept as cross-checks of it (tests, tools/stress_shapes.py).  FunctionSoftsplat
 * end to end; ws: fldr_softsplat_tile_ws_floats(N,H,W) floats; _strided reads sample n, channel c at img + n*img_bstride +
 * c*img_cstride; _prebounded takes the bounds table in ws as given (fldr_splat_bounds_upsampled). */
FLDR_API int fldr_softsplat_tile(const float* img, const float* flow, const float* metric_or_null, float* out, float* ws,
                        int N, int C, int H, int W, int mode, fldr_stream_t stream);
FLDR_API int fldr_softsplat_tile_strided(const float* img, int64_t img_bstride, int64_t img_cstride, const float* flow,
                                const float* metric_or_null, float* out, float* ws, int N, int C, int H, int W, int mode,
                                fldr_stream_t stream);
FLDR_API int fldr_softsplat_tile_prebounded(const float* img, int64_t img_bstride, int64_t img_cstride, const float* flow,
                                   const float* metric_or_null, float* out, float* ws, int N, int C, int H, int W, int mode,
                                   fldr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
