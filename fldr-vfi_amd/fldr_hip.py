"""ctypes binding of libfldr_hip.so (C ABI: include/fldr_hip.h).

This is the stub that replaces the reference's CuPy launch plumbing
(`cupy_kernel` / `cupy_launch`, softSplat.py:160-218, correlation.py:245-289):
no source specialisation, no JIT, kernels are enqueued on torch's current HIP
stream.  There is NO fallback: if the shared library is missing or a tensor is
not a CUDA(HIP) tensor the call raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FLDR_LIB") or os.path.join(_HERE, "libfldr_hip.so")     # FLDR_LIB: an experimental build (A/B measurements)

MAX_SRC = 12
_c_float_p = ctypes.c_void_p


class ConvDesc(ctypes.Structure):
    _fields_ = [
        ("src", ctypes.c_void_p * MAX_SRC),
        ("src_bstride", ctypes.c_int64 * MAX_SRC),
        ("src_c", ctypes.c_int32 * MAX_SRC),
        ("src_up2", ctypes.c_int32 * MAX_SRC),
        ("n_src", ctypes.c_int32),
        ("wpack", ctypes.c_void_p),
        ("bias", ctypes.c_void_p),
        ("residual", ctypes.c_void_p),
        ("out", ctypes.c_void_p),
        ("N", ctypes.c_int32), ("cin", ctypes.c_int32), ("cout", ctypes.c_int32), ("cout_store", ctypes.c_int32),
        ("Hin", ctypes.c_int32), ("Win", ctypes.c_int32), ("Hout", ctypes.c_int32), ("Wout", ctypes.c_int32),
        ("ksize", ctypes.c_int32), ("stride", ctypes.c_int32), ("relu", ctypes.c_int32), ("precision", ctypes.c_int32),
        ("out_spk", ctypes.c_void_p),
        ("src_cstride", ctypes.c_int64 * MAX_SRC),
    ]


class PrepDesc(ctypes.Structure):
    _fields_ = [
        ("flow_lo", ctypes.c_void_p), ("I0", ctypes.c_void_p), ("I1", ctypes.c_void_p),
        ("i0_bstride", ctypes.c_int64), ("i1_bstride", ctypes.c_int64),
        ("t", ctypes.c_void_p), ("z0", ctypes.c_void_p), ("z1", ctypes.c_void_p),
        ("flow_t0", ctypes.c_void_p), ("flow_t1", ctypes.c_void_p), ("flowback_0", ctypes.c_void_p), ("flowback_1", ctypes.c_void_p),
        ("im0_tot", ctypes.c_void_p), ("im1_tot", ctypes.c_void_p),
        ("N", ctypes.c_int32), ("h", ctypes.c_int32), ("w", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
        ("mul", ctypes.c_float), ("z_alpha0", ctypes.c_float), ("z_alpha1", ctypes.c_float), ("withmask", ctypes.c_int32),
        ("ws", ctypes.c_void_p),
        ("i0_cstride", ctypes.c_int64), ("i1_cstride", ctypes.c_int64),
        ("phase", ctypes.c_int32), ("reserved", ctypes.c_int32),
    ]


class PcaLevel(ctypes.Structure):
    _fields_ = [("planes", ctypes.c_void_p), ("out_f32", ctypes.c_void_p), ("out_spk", ctypes.c_void_p),
                ("P", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("raw_ws", ctypes.c_void_p)]


class SplatGatherDesc(ctypes.Structure):
    _fields_ = [("img", ctypes.c_void_p * 2), ("img_bstride", ctypes.c_int64 * 2), ("flow", ctypes.c_void_p * 2),
                ("flow_bstride", ctypes.c_int64 * 2), ("metric", ctypes.c_void_p * 2), ("out_f32", ctypes.c_void_p * 2),
                ("out_spk", ctypes.c_void_p * 2), ("ws", ctypes.c_void_p),
                ("ndir", ctypes.c_int32), ("N", ctypes.c_int32), ("C", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
                ("mode", ctypes.c_int32)]


class SplatAccDesc(ctypes.Structure):
    _fields_ = [("img", ctypes.c_void_p * 2), ("img_bstride", ctypes.c_int64 * 2), ("img_cstride", ctypes.c_int64 * 2),
                ("flow", ctypes.c_void_p * 2), ("flow_bstride", ctypes.c_int64 * 2), ("metric", ctypes.c_void_p * 2),
                ("ws", ctypes.c_void_p * 2), ("out_f32", ctypes.c_void_p * 2), ("out_spk", ctypes.c_void_p * 2),
                ("nprob", ctypes.c_int32), ("N", ctypes.c_int32), ("C", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
                ("mode", ctypes.c_int32), ("flags", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class SpkConvDesc(ctypes.Structure):
    _fields_ = [
        ("src", ctypes.c_void_p * MAX_SRC),
        ("src_bstride", ctypes.c_int64 * MAX_SRC),
        ("src_c", ctypes.c_int32 * MAX_SRC),
        ("src_up2", ctypes.c_int32 * MAX_SRC),
        ("n_src", ctypes.c_int32),
        ("wpack", ctypes.c_void_p),
        ("bias", ctypes.c_void_p),
        ("residual", ctypes.c_void_p),
        ("out_f32", ctypes.c_void_p),
        ("out_spk", ctypes.c_void_p),
        ("N", ctypes.c_int32), ("cin", ctypes.c_int32), ("cout", ctypes.c_int32), ("cout_store", ctypes.c_int32),
        ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("relu", ctypes.c_int32), ("precision", ctypes.c_int32),
    ]


_SIGNATURES = {
    "fldr_version": (ctypes.c_int, []),
    "fldr_error_string": (ctypes.c_char_p, [ctypes.c_int]),
    "fldr_softsplat_fwd": (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "fldr_softsplat_fused": (ctypes.c_int, [_c_float_p] * 5 + [ctypes.c_int] * 5 + [ctypes.c_void_p]),
    "fldr_softsplat_fused_spk": (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_void_p, _c_float_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]),
    "fldr_softsplat_pair_spk": (ctypes.c_int, [_c_float_p] * 4 + [ctypes.c_void_p, _c_float_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "fldr_resize_bilinear_spk": (ctypes.c_int, [_c_float_p] * 2 + [ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_float, ctypes.c_void_p]),
    "fldr_resize_bilinear_spk_bounds": (ctypes.c_int, [_c_float_p] * 2 + [ctypes.c_void_p, _c_float_p] + [ctypes.c_int] * 5 + [ctypes.c_float, ctypes.c_void_p]),
    "fldr_softsplat_gather_ws_floats": (ctypes.c_int64, [ctypes.c_int] * 4),
    "fldr_softsplat_gather": (ctypes.c_int, [ctypes.POINTER(SplatGatherDesc), ctypes.c_void_p]),
    "fldr_softsplat_tile_ws_floats": (ctypes.c_int64, [ctypes.c_int] * 3),
    "fldr_softsplat_acc64": (ctypes.c_int, [ctypes.POINTER(SplatAccDesc), ctypes.c_void_p]),
    "fldr_splat_bounds_upsampled_pair": (ctypes.c_int, [_c_float_p, ctypes.c_int64, _c_float_p, ctypes.c_int, ctypes.c_float, _c_float_p]
                                         + [ctypes.c_int] * 5 + [ctypes.c_void_p]),
    "fldr_softsplat_tile": (ctypes.c_int, [_c_float_p] * 5 + [ctypes.c_int] * 5 + [ctypes.c_void_p]),
    "fldr_softsplat_tile_strided": (ctypes.c_int, [_c_float_p, ctypes.c_int64, ctypes.c_int64] + [_c_float_p] * 4 + [ctypes.c_int] * 5
                                    + [ctypes.c_void_p]),
    "fldr_softsplat_tile_prebounded": (ctypes.c_int, [_c_float_p, ctypes.c_int64, ctypes.c_int64] + [_c_float_p] * 4 + [ctypes.c_int] * 5
                                       + [ctypes.c_void_p]),
    "fldr_splat_bounds_upsampled": (ctypes.c_int, [_c_float_p, ctypes.c_int64, _c_float_p, ctypes.c_int, ctypes.c_float, _c_float_p]
                                    + [ctypes.c_int] * 5 + [ctypes.c_void_p]),
    "fldr_correlation_fwd": (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "fldr_debug_corr_variant": (ctypes.c_int, [ctypes.c_int]),
    "fldr_debug_splat_group_fold": (ctypes.c_int, [ctypes.c_int]),
    "fldr_debug_corr_chunk": (ctypes.c_int, [ctypes.c_int]),
    "fldr_debug_corr_xcd": (ctypes.c_int, [ctypes.c_int]),
    "fldr_softsplat_bwd": (ctypes.c_int, [_c_float_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "fldr_correlation_bwd": (ctypes.c_int, [_c_float_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "fldr_pca_project": (ctypes.c_int, [_c_float_p] * 7 + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "fldr_pca_project_stream": (ctypes.c_int, [_c_float_p] * 8 + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "fldr_pca_table_size": (ctypes.c_int64, [ctypes.c_int]),
    "fldr_pca_prepack": (ctypes.c_int, [_c_float_p] * 4 + [ctypes.c_int, ctypes.c_void_p]),
    "fldr_pca_project_pyramid": (ctypes.c_int, [ctypes.POINTER(PcaLevel), ctypes.c_int, _c_float_p, ctypes.c_int, _c_float_p, ctypes.c_void_p]),
    "fldr_debug_pca_workgroups": (ctypes.c_int, [ctypes.c_int]),
    "fldr_debug_pca_variant": (ctypes.c_int, [ctypes.c_int]),
    "fldr_debug_ring_tile_width": (ctypes.c_int, [ctypes.c_int]),
    "fldr_debug_ring_spin_limit": (ctypes.c_int, [ctypes.c_int]),
    "fldr_status_word": (ctypes.c_int, [ctypes.POINTER(ctypes.POINTER(ctypes.c_int))]),
    "fldr_debug_s2_vec4": (ctypes.c_int, [ctypes.c_int]),
    "fldr_bwarp": (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_int] * 5 + [ctypes.c_void_p]),
    "fldr_bwarp_tscaled": (ctypes.c_int, [_c_float_p] * 4 + [ctypes.c_int] * 7 + [ctypes.c_void_p]),
    "fldr_resize_bilinear": (ctypes.c_int, [_c_float_p] * 2 + [ctypes.c_int] * 5 + [ctypes.c_float, ctypes.c_void_p]),
    "fldr_level0_prep": (ctypes.c_int, [ctypes.POINTER(PrepDesc), ctypes.c_void_p]),
    "fldr_zmetric": (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_float, _c_float_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "fldr_conv_prepack_size": (ctypes.c_int64, [ctypes.c_int] * 3),
    "fldr_conv_prepack": (ctypes.c_int, [_c_float_p] * 2 + [ctypes.c_int] * 3 + [ctypes.c_void_p]),
    "fldr_conv2d": (ctypes.c_int, [ctypes.POINTER(ConvDesc), ctypes.c_void_p]),
    "fldr_conv_split_prepack_size": (ctypes.c_int64, [ctypes.c_int] * 2),
    "fldr_conv_split_prepack": (ctypes.c_int, [_c_float_p] * 2 + [ctypes.c_int] * 2 + [ctypes.c_void_p]),
    "fldr_conv2d_split": (ctypes.c_int, [ctypes.POINTER(ConvDesc), ctypes.c_void_p]),
    "fldr_conv_s2_prepack_size": (ctypes.c_int64, [ctypes.c_int] * 2),
    "fldr_conv_s2_prepack": (ctypes.c_int, [_c_float_p] * 2 + [ctypes.c_int] * 2 + [ctypes.c_void_p]),
    "fldr_conv2d_s2_split": (ctypes.c_int, [ctypes.POINTER(ConvDesc), ctypes.c_void_p]),
    "fldr_conv2d_s2_spk": (ctypes.c_int, [ctypes.POINTER(ConvDesc), ctypes.c_void_p]),
    "fldr_conv2d_s2_spk_pair": (ctypes.c_int, [ctypes.POINTER(ConvDesc), ctypes.POINTER(ConvDesc), ctypes.c_void_p]),
    "fldr_debug_s2_persistent": (ctypes.c_int, [ctypes.c_int]),
    "fldr_debug_s2_dma": (ctypes.c_int, [ctypes.c_int]),
    "fldr_debug_s2_xshift": (ctypes.c_int, [ctypes.c_int]),
    "fldr_debug_dec3_xshift": (ctypes.c_int, [ctypes.c_int]),
    "fldr_debug_dec3_xcd": (ctypes.c_int, [ctypes.c_int]),
    "fldr_debug_splat_quad": (ctypes.c_int, [ctypes.c_int]),
    "fldr_spk_bytes": (ctypes.c_int64, [ctypes.c_int] * 3),
    "fldr_spk_pack": (ctypes.c_int, [_c_float_p, ctypes.c_int64, ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "fldr_spk_unpack": (ctypes.c_int, [ctypes.c_void_p, _c_float_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "fldr_conv_spk_prepack_size": (ctypes.c_int64, [ctypes.c_int] * 2),
    "fldr_conv_spk_prepack": (ctypes.c_int, [_c_float_p] * 2 + [ctypes.c_int] * 2 + [ctypes.c_void_p]),
    "fldr_conv2d_spk": (ctypes.c_int, [ctypes.POINTER(SpkConvDesc), ctypes.c_void_p]),
    "fldr_conv2d_spk_levels": (ctypes.c_int, [ctypes.POINTER(SpkConvDesc), ctypes.c_int, ctypes.c_void_p]),
    "fldr_debug_spk_wgs_per_xcd": (ctypes.c_int, [ctypes.c_int]),
    "fldr_range_status": (ctypes.c_int, [ctypes.c_int]),
    "fldr_ring_status": (ctypes.c_int, [ctypes.c_int]),
    "fldr_debug_spk_small_units": (ctypes.c_int, [ctypes.c_int]),
    "fldr_debug_spk_variant": (ctypes.c_int, [ctypes.c_int]),
    "fldr_debug_ring_consumers": (ctypes.c_int, [ctypes.c_int]),
    "fldr_debug_ring32": (ctypes.c_int, [ctypes.c_int]),
    "fldr_debug_ring_resident": (ctypes.c_int, [ctypes.c_int]),
    "fldr_debug_ring_timeouts": (ctypes.c_int, []),
    "fldr_debug_busy_partner": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "fldr_sizeof_desc": (ctypes.c_int, [ctypes.c_int]),
    "fldr_synth_tail": (ctypes.c_int, [_c_float_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64),
                                       _c_float_p, ctypes.c_double, _c_float_p, _c_float_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p]),
    "fldr_dec3_prepack": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_void_p]),
    "fldr_dec3_synth": (ctypes.c_int, [_c_float_p] * 3 + [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64),
                                       _c_float_p, ctypes.c_double, _c_float_p, _c_float_p, _c_float_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p]),
    "fldr_dec3_synth_strided": (ctypes.c_int, [_c_float_p] * 3 + [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64),
                                               ctypes.POINTER(ctypes.c_int64), _c_float_p, ctypes.c_double, _c_float_p, _c_float_p,
                                               _c_float_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p]),
    "fldr_dec3_prepack_spk_size": (ctypes.c_int64, []),
    "fldr_dec3_prepack_spk": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_void_p]),
    "fldr_dec3_synth_spk": (ctypes.c_int, [ctypes.c_void_p, _c_float_p, _c_float_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64),
                                           ctypes.POINTER(ctypes.c_int64), _c_float_p, ctypes.c_double, _c_float_p, _c_float_p,
                                           _c_float_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p]),
    "fldr_dec23_prepack_size": (ctypes.c_int64, []),
    "fldr_dec23_prepack": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_void_p]),
    "fldr_dec23_synth": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p, ctypes.POINTER(ctypes.c_void_p),
                                        ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), _c_float_p, ctypes.c_double, _c_float_p, _c_float_p,
                                        ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]),
    "fldr_ingest_u8": (ctypes.c_int, [ctypes.c_void_p, _c_float_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]),
    "fldr_pyramid_bicubic": (ctypes.c_int, [_c_float_p, _c_float_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "fldr_ingest_pyramid_u8": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)] + [ctypes.c_int] * 6 + [ctypes.c_void_p]),
    "fldr_ssim_y_ws_doubles": (ctypes.c_int64, [ctypes.c_int] * 3),
    "fldr_ssim_y_u8": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p]),
    "fldr_frame_metrics": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
                           + [ctypes.c_int] * 5 + [ctypes.c_void_p]),
}

# The integration ABI (include/fldr_hip.h: what libfldr_hip.so exports) and the tuning / cross-check hooks that only the test
# build libfldr_hip_test.so has (include/fldr_hip_test_hooks.h).
_TEST_BUILD_ONLY = ("fldr_softsplat_tile", "fldr_softsplat_tile_strided", "fldr_softsplat_tile_prebounded")    # retired splat generation
EXPORTS = tuple(n for n in _SIGNATURES if not n.startswith("fldr_debug_") and n not in _TEST_BUILD_ONLY)
HOOKS = tuple(n for n in _SIGNATURES if n.startswith("fldr_debug_") or n in _TEST_BUILD_ONLY)
TEST_LIB_PATH = os.path.join(_HERE, "libfldr_hip_test.so")
ABI_VERSION = 105                # include/fldr_hip.h: FLDR_VERSION
# entry points the DEFAULT 4K forward / bench / harness call: a variant library (FLDR_LIB) that lacks one of them fails at load
_DEFAULT_PATH = ("fldr_version", "fldr_error_string", "fldr_sizeof_desc", "fldr_range_status", "fldr_ring_status", "fldr_status_word", "fldr_pca_prepack",
                 "fldr_pca_project_pyramid", "fldr_conv2d_spk", "fldr_conv2d_spk_levels", "fldr_conv_spk_prepack", "fldr_conv2d_s2_split",
                 "fldr_conv2d_s2_spk", "fldr_conv2d_s2_spk_pair", "fldr_conv_s2_prepack", "fldr_softsplat_acc64", "fldr_level0_prep",
                 "fldr_splat_bounds_upsampled_pair", "fldr_resize_bilinear_spk", "fldr_resize_bilinear_spk_bounds", "fldr_dec3_prepack_spk",
                 "fldr_dec3_synth_spk", "fldr_spk_pack", "fldr_dec23_prepack_size", "fldr_dec23_prepack", "fldr_dec23_synth")
_lib = None
_hooks_lib = None


def _load(path, want_hooks):
    if not os.path.exists(path):
        raise ImportError("%s is missing — build it with `make -C fldr-vfi_amd/csrc` (or `python -c 'import __graft_entry__ as g; "
                          "g.build()'`). There is no CPU/eager fallback." % path)
    l = ctypes.CDLL(path)
    variant = bool(os.environ.get("FLDR_LIB"))           # an experimental / older build selected for an A/B measurement
    missing = []
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(l, name)
        except AttributeError:
            # hooks exist in the test build only; a variant build may lack newer entry points — but never one the default path calls
            if name in HOOKS and (not want_hooks or variant):
                continue
            if variant and name not in _DEFAULT_PATH:
                missing.append(name)
                continue
            raise ImportError("%s does not export %s, which the default forward calls: rebuild it" % (path, name))
        fn.restype = res
        fn.argtypes = args
    # binding self-check, ALWAYS: this file's struct mirrors against the structs the library was compiled with (fldr_sizeof_desc exists
    # in every build since ABI 102) — a stale .so next to a newer binding (or the reverse) fails here instead of corrupting memory.
    # An index the library does not know yet (FLDR_E_ARG) is tolerated for variant builds only.
    for which, cls in enumerate((ConvDesc, SpkConvDesc, PrepDesc, PcaLevel, SplatAccDesc, SplatGatherDesc)):
        got = l.fldr_sizeof_desc(which)
        if got != ctypes.sizeof(cls) and not (variant and got < 0):
            raise ImportError("%s: sizeof(%s) is %d in the library, %d in this binding" % (path, cls.__name__, got, ctypes.sizeof(cls)))
    if l.fldr_version() != ABI_VERSION:
        msg = "%s reports ABI version %d, this binding is written for %d" % (path, l.fldr_version(), ABI_VERSION)
        if not variant:
            raise ImportError(msg + ": rebuild it")
        import warnings
        warnings.warn(msg + " (FLDR_LIB variant: struct sizes agree, continuing)" + (("; missing entry points: " + ", ".join(missing)) if missing else ""))
    return l


def lib():
    """The loaded PRODUCT library; raises (never falls back) when it has not been built.  Inside `with test_hooks():` the test
    build (the same kernels + the fldr_debug_* hooks and the retired cross-check kernels)."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH, want_hooks=False)
    return _lib


def busy_partner(out, workgroups=512, lds_bytes=1024, iters=40000, kind=2):
    """Concurrency tests / tools only (the TEST build's fldr_debug_busy_partner, whichever library the other calls go to): launch a kernel
    on the current stream that only occupies the compute units — `workgroups` x 256 threads holding `lds_bytes` of LDS each, looping
    `iters` times over kind 0 s_sleep, 1 matrix instructions, 2 vector FMAs, 3 scalar adds, 4 LDS reads.  `out`: >= workgroups * 256 floats."""
    global _hooks_lib
    if _hooks_lib is None:
        _hooks_lib = _load(os.environ.get("FLDR_LIB") or TEST_LIB_PATH, want_hooks=True)
    if out.dtype != torch.float32 or out.numel() < workgroups * 256:
        raise ValueError("busy_partner: out must hold workgroups * 256 floats")
    _check(_hooks_lib.fldr_debug_busy_partner(_dev(out, "out"), workgroups, lds_bytes, iters, kind, _stream()), "fldr_debug_busy_partner")


class test_hooks:
    """Context manager for tests / tools: route every call of this module to libfldr_hip_test.so, whose fldr_debug_* hooks select
    kernel variants and tuning values.  Environment hooks (FLDR_RING_CONSUMERS, FLDR_SPK_VARIANT, FLDR_PCA_WORKGROUPS,
    FLDR_PCA_VARIANT, FLDR_RING_TILE_WIDTH, FLDR_S2_VEC4) are applied when the test build is entered."""

    def __enter__(self):
        global _lib, _hooks_lib
        if _hooks_lib is None:
            _hooks_lib = _load(os.environ.get("FLDR_LIB") or TEST_LIB_PATH, want_hooks=True)
            for env, hook in (("FLDR_RING_CONSUMERS", "fldr_debug_ring_consumers"), ("FLDR_SPK_VARIANT", "fldr_debug_spk_variant"),
                              ("FLDR_PCA_WORKGROUPS", "fldr_debug_pca_workgroups"), ("FLDR_PCA_VARIANT", "fldr_debug_pca_variant"),
                              ("FLDR_RING_TILE_WIDTH", "fldr_debug_ring_tile_width"), ("FLDR_S2_VEC4", "fldr_debug_s2_vec4")):
                if os.environ.get(env) and hasattr(_hooks_lib, hook):
                    getattr(_hooks_lib, hook)(int(os.environ[env]))
        self._prev = _lib
        _lib = _hooks_lib
        return _hooks_lib

    def __exit__(self, *exc):
        global _lib
        _lib = self._prev
        return False


def enter_test_hooks():
    """tools/: switch this process to the test build for good (idempotent)."""
    if _lib is None or _lib is not _hooks_lib:
        test_hooks().__enter__()
    return _lib


def spk_variant():
    """Pipeline of fldr_conv2d_spk: 1 = loader / consumer ring (the only one in the product library), 0 = barrier pipeline
    (selectable in the test build)."""
    fn = getattr(lib(), "fldr_debug_spk_variant", None)
    return 1 if fn is None else fn(-1)


class FldrError(RuntimeError):
    pass


def _check(code, what):
    if code != 0:
        raise FldrError("%s failed: %s (code %d)" % (what, lib().fldr_error_string(code).decode(), code))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """Handle of torch's current HIP stream.  Called once per launch (~90 per forward): the raw getter costs ~0.3 us against
    ~3 us for building a torch.cuda.Stream object (the coarse pyramid levels are paced by the host's time per launch)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, name, dtype=torch.float32):
    if not t.is_cuda:
        raise NotImplementedError("%s: fldr_hip has no CPU path (the reference has none either, softSplat.py:251-252)" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return ctypes.c_void_p(t.data_ptr())


def _planes(t, name):
    """A [N,C,H,W] fp32 device tensor as (tensor, batch stride, channel stride) in floats for the entry points that take
    channel-strided images: views whose [H,W] planes are contiguous (x_l[0][:, :, 0] of the [B,3,2,H,W] frames,
    fLDRnet.py:130-131) pass through, anything else is copied."""
    if not t.is_cuda:
        raise NotImplementedError("%s: fldr_hip has no CPU path (the reference has none either, softSplat.py:251-252)" % name)
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (name, t.dtype))
    N, C, H, W = t.shape
    if not ((W == 1 or t.stride(3) == 1) and (H == 1 or t.stride(2) == W)):
        t = t.contiguous()
    return t, (t.stride(0) if N > 1 else 0), (t.stride(1) if C > 1 else H * W)


# ---------------------------------------------------------------------------------------------
# operators
# ---------------------------------------------------------------------------------------------

def softsplat_fwd(inp, flow):
    """_FunctionSoftsplat.forward (softSplat.py:222-258)."""
    N, C, H, W = inp.shape
    assert flow.shape[1] == 2 and flow.shape[2] == H and flow.shape[3] == W      # softSplat.py:227-229
    inp, flow = inp.contiguous(), flow.contiguous()
    out = inp.new_zeros(N, C, H, W)
    _check(lib().fldr_softsplat_fwd(_dev(inp, "input"), _dev(flow, "flow"), _dev(out, "output"), N, C, H, W, _stream()),
           "fldr_softsplat_fwd")
    return out


def softsplat_bwd(inp, flow, grad_out, need_input=True, need_flow=True):
    """_FunctionSoftsplat.backward (softSplat.py:259-318) -> (gradInput or None, gradFlow or None)."""
    N, C, H, W = inp.shape
    inp, flow, grad_out = inp.contiguous(), flow.contiguous(), grad_out.contiguous()
    gi = torch.empty_like(inp) if need_input else None
    gf = torch.empty_like(flow) if need_flow else None
    if gi is None and gf is None:
        return None, None
    _check(lib().fldr_softsplat_bwd(_dev(inp, "input"), _dev(flow, "flow"), _dev(grad_out, "gradOutput"),
                                    _dev(gi, "gradInput") if gi is not None else None,
                                    _dev(gf, "gradFlow") if gf is not None else None, N, C, H, W, _stream()), "fldr_softsplat_bwd")
    return gi, gf


def correlation_bwd(first, second, grad_out, need_first=True, need_second=True):
    """_FunctionCorrelation.backward (correlation.py:350-410) -> (gradFirst or None, gradSecond or None)."""
    N, C, H, W = first.shape
    first, second, grad_out = first.contiguous(), second.contiguous(), grad_out.contiguous()
    g1 = torch.empty_like(first) if need_first else None
    g2 = torch.empty_like(second) if need_second else None
    if g1 is None and g2 is None:
        return None, None
    _check(lib().fldr_correlation_bwd(_dev(first, "first"), _dev(second, "second"), _dev(grad_out, "gradOutput"),
                                      _dev(g1, "gradFirst") if g1 is not None else None,
                                      _dev(g2, "gradSecond") if g2 is not None else None, N, C, H, W, _stream()), "fldr_correlation_bwd")
    return g1, g2


_MODES = {"summation": 0, "average": 1, "linear": 2, "softmax": 3}


# Forward-splat kernels (FLDR_SPLAT = auto | acc64 | strip | tile):
#   "acc64" (= auto, default since round 3): destination-owned tiles with fp64 LDS atomics (splat_acc64_kernels.hip): every
#            splat of the forward and FunctionSoftsplat;
#   "strip": global float atomics with in-register merging + a normalisation pass (warp_kernels.hip): the raw operator
#            _FunctionSoftsplat (fldr_softsplat_fwd accumulates into the caller's tensor) and fldr_softsplat_fused;
#   "tile":  the destination-owned bands of rounds 1-2 (no atomics, claim rounds; splat_tile_kernels.hip) — a retired generation,
#            compiled into the TEST build only since round 4 (cross-check of acc64): selecting it outside fldr_hip.test_hooks() raises.
# Environment switches of the product path (everything else below is a module attribute that tests / tools flip to reach a cross-check
# kernel, not a deployment knob): FLDR_SPLAT (operator-level splat kernel), FLDR_SPLAT_FEATURES=gather (deterministic feature splats),
# FLDR_PCA_F32=0 (PCA features split-packed only: opt-in, not the parity configuration), FLDR_DEC23=0 (dec2 and dec3 as two kernels), FLDR_CONV_PRECISION (split | fp32 | fp16), FLDR_LIB (an
# experimental build of the library).
SPLAT_KERNEL = os.environ.get("FLDR_SPLAT", "auto")
# Warped feature maps of the flow estimator (fLDRnet.py:386-387): "acc64" (default since round 3) = destination-owned tiles of fp64 LDS
# accumulators, both directions per launch (splat_acc64_kernels.hip; deterministic); "gather" = the atomic-free gather of
# splat_gather_kernels.hip (opt-in: ~2x slower, every match costs a 48-load body whose latency the few waves of a feature map cannot hide).
SPLAT_FEATURES = os.environ.get("FLDR_SPLAT_FEATURES", "acc64")
if SPLAT_FEATURES not in ("acc64", "gather"):
    raise ValueError("FLDR_SPLAT_FEATURES=%r: expected 'acc64' (default) or 'gather' (the scatter kernel 'strip' of rounds 1-2 is retired)" % SPLAT_FEATURES)
if SPLAT_KERNEL not in ("auto", "acc64", "strip", "tile"):
    raise ValueError("FLDR_SPLAT=%r: expected auto | acc64 | strip | tile" % SPLAT_KERNEL)
# Switches of earlier rounds that no longer exist: setting one has no effect, say so once instead of silently ignoring it.
_RETIRED_ENV = ("FLDR_PREP_SPLIT", "FLDR_LEVEL_BATCH", "FLDR_ENC3_SPLIT", "FLDR_ENC3_PAIR", "FLDR_S2_SPK", "FLDR_SPLAT_BOUNDS", "FLDR_DEC3_MFMA",
                "FLDR_RESIZE_BOUNDS", "FLDR_PCA_RAW_MIN_BYTES", "FLDR_INGEST_FUSED")
for _n in _RETIRED_ENV:
    if _n in os.environ:
        import warnings
        warnings.warn("%s is set but is no longer a switch of fldr_hip (retired; see DESIGN.md section 1 for the current ones)" % _n, stacklevel=2)
# rec_ctx_ds of all pyramid levels in two launches (fldr_conv2d_spk_levels) instead of two per level
LEVEL_BATCH = True
# 1 (default; the parity configuration): the PCA rescale launch writes the features as fp32 next to the split-packed tensor and
# rec_ctx_ds.2 adds that fp32 tensor (tracks the oracle's features to the last bits).  FLDR_PCA_F32=0 (opt-in, +0.4 % throughput):
# the features exist split-packed only and rec_ctx_ds.2 adds hi + lo of the packed feature (the fp32 feature up to 2^-22 relative,
# 2.4e-7 for |x| <= 1; 71 instead of 141 MB written per 4K forward).  Every golden bound holds in both modes; the opt-in mode is NOT
# the parity default because on the strong-non-rigid-motion 4K stress pair that 2.4e-7 flips one nearly empty target cell of a feature
# splat and a 24 x 46 px patch of the frame then differs from the oracle by up to 0.145 (tests/test_gpu_fullsize.py reports it).
PCA_F32 = os.environ.get("FLDR_PCA_F32", "1") != "0"


# Bounds table of the level-0 image splats: "lowres" (default) = from the low-resolution flow the upsampled flow_t is made of
# (fldr_splat_bounds_upsampled: conservative intervals, 3 us instead of a 23 us pass over the full-resolution planes);
# "exact" = the pre-pass over the full-resolution flow.  Same results up to fp32 summation order.
SPLAT_BOUNDS = "lowres"


def splat_bounds_upsampled(flow_lo, t, scale_mode, mul, H, W):
    """Bounds workspace for softsplat_fused(..., bounds_ws=...) of flow = interpolate(scale * flow_lo, (H, W)) * mul.
    flow_lo [N,2,h,w] (may be a channel slice of the [N,4,h,w] level flow); scale_mode 0: 1, 1: t[n], 2: 1 - t[n]."""
    N, two, h, w = flow_lo.shape
    assert two == 2 and flow_lo.stride(3) == 1 and flow_lo.stride(2) == w and flow_lo.stride(1) == h * w
    ws = torch.empty(lib().fldr_softsplat_tile_ws_floats(N, H, W), device=flow_lo.device, dtype=torch.float32)
    if t is not None:
        t = t.reshape(N).contiguous().float()
    _check(lib().fldr_splat_bounds_upsampled(ctypes.c_void_p(flow_lo.data_ptr()), flow_lo.stride(0) if N > 1 else 2 * h * w,
                                             _dev(t, "t") if t is not None else None, int(scale_mode), float(mul), _dev(ws, "ws"),
                                             N, h, w, H, W, _stream()), "fldr_splat_bounds_upsampled")
    return ws


def splat_bounds_upsampled_pair(flow_l, t, pair, mul, H, W):
    """Both bounds tables of a two-problem softsplat_acc64 call in one launch (pass the result as bounds_ws): flow_l [N,4,h,w] =
    the flow of a pyramid level; pair = "images" (problem 0: flow = up(t * flow_l[:, 2:]) * mul, problem 1:
    up((1 - t) * flow_l[:, :2]) * mul) or "features" (problem 0: up(flow_l[:, :2]) * mul, problem 1: up(flow_l[:, 2:]) * mul)."""
    N, four, h, w = flow_l.shape
    assert four == 4 and flow_l.stride(3) == 1 and flow_l.stride(2) == w and flow_l.stride(1) == h * w
    ws = torch.empty(2 * lib().fldr_softsplat_tile_ws_floats(N, H, W), device=flow_l.device, dtype=torch.float32)
    if t is not None:
        t = t.reshape(N).contiguous().float()
    if not flow_l.is_cuda or flow_l.dtype != torch.float32:
        raise TypeError("flow_l must be a float32 device tensor")
    _check(lib().fldr_splat_bounds_upsampled_pair(ctypes.c_void_p(flow_l.data_ptr()), flow_l.stride(0) if N > 1 else 0, _dev(t, "t") if t is not None else None,
                                                  {"images": 1, "features": 2}[pair], float(mul), _dev(ws, "ws"), N, h, w, H, W, _stream()),
           "fldr_splat_bounds_upsampled_pair")
    return ws


def softsplat_fused(img, flow, metric, mode, out=None, scratch=None, kernel=None, want_spk=False, out_spk=None, bounds_ws=None):
    """FunctionSoftsplat (softSplat.py:320-352).  want_spk: return the result split-packed (Spk) instead of fp32 NCHW
    (written into `out_spk`, a Spk of the same shape, when given).  bounds_ws: a bounds table from splat_bounds_upsampled
    (tile kernel only) instead of the exact pre-pass over `flow`."""
    N, C, H, W = img.shape
    assert flow.shape[1] == 2 and flow.shape[2] == H and flow.shape[3] == W
    flow = flow.contiguous()
    if metric is not None:
        metric = metric.contiguous()
    kern = kernel or SPLAT_KERNEL
    if kern == "auto":
        kern = "acc64"
    if kern == "acc64":
        if want_spk and C <= 3:
            kern = "strip"
        else:
            r = softsplat_acc64([img], [flow], None if metric is None else [metric], mode, want_f32=not want_spk, want_spk=want_spk,
                                bounds_ws=None if bounds_ws is None else [bounds_ws])[0]
            if out is not None and not want_spk:
                out.copy_(r)
                return out
            if out_spk is not None and want_spk:
                raise ValueError("softsplat_fused(kernel='acc64') allocates its packed output")
            return r
    if kern == "tile" and not want_spk:
        img, ibs, ics = _planes(img, "img")
        ws = bounds_ws if bounds_ws is not None else torch.empty(lib().fldr_softsplat_tile_ws_floats(N, H, W), device=img.device, dtype=torch.float32)
        if out is None:
            out = torch.empty(N, C, H, W, device=img.device, dtype=torch.float32)
        if not hasattr(lib(), "fldr_softsplat_tile_strided"):
            raise FldrError("the band splat (kernel='tile' / FLDR_SPLAT=tile) is a retired generation kept in the test build only: "
                            "enter fldr_hip.test_hooks() (tests, tools) — the product splats run on fldr_softsplat_acc64")
        fn = lib().fldr_softsplat_tile_prebounded if bounds_ws is not None else lib().fldr_softsplat_tile_strided
        _check(fn(ctypes.c_void_p(img.data_ptr()), ibs, ics, _dev(flow, "flow"), _dev(metric, "metric") if metric is not None else None,
                  _dev(out, "out"), _dev(ws, "ws"), N, C, H, W, _MODES[mode], _stream()), "fldr_softsplat_tile")
        return out
    img = img.contiguous()
    if want_spk:
        ca = C + (0 if mode == "summation" else 1)
        scratch = torch.empty(N * ca * H * W, device=img.device, dtype=torch.float32)
        outp = _spk_alloc(N, C, H, W, img.device) if out_spk is None else out_spk
        assert outp.shape == (N, C, H, W) and (N == 1 or outp.bstride == ((C + 7) // 8) * 2 * H * W * 16)
        _check(lib().fldr_softsplat_fused_spk(_dev(img, "img"), _dev(flow, "flow"),
                                              _dev(metric, "metric") if metric is not None else None,
                                              ctypes.c_void_p(outp.ptr), _dev(scratch, "scratch"), N, C, H, W,
                                              _MODES[mode], _stream()), "fldr_softsplat_fused_spk")
        return outp
    ca = C + (0 if mode == "summation" else 1)
    if scratch is None:
        scratch = torch.empty(N * ca * H * W, device=img.device, dtype=torch.float32)
    if out is None:
        out = torch.empty(N, C, H, W, device=img.device, dtype=torch.float32)
    _check(lib().fldr_softsplat_fused(_dev(img, "img"), _dev(flow, "flow"),
                                      _dev(metric, "metric") if metric is not None else None,
                                      _dev(out, "out"), _dev(scratch, "scratch"), N, C, H, W, _MODES[mode], _stream()),
           "fldr_softsplat_fused")
    return out


def softsplat_acc64(imgs, flows, metrics=None, mode="softmax", want_f32=True, want_spk=False, bounds_ws=None, spk_batch=False):
    """FunctionSoftsplat (softSplat.py:320-352) of one or two (img [N,C,H,W], flow [N,2,H,W]) problems of the same shape with
    destination-owned tiles and fp64 LDS atomics (fldr_softsplat_acc64): no global atomics, accumulator, memset or
    normalisation pass.  img may be a view with contiguous [H,W] planes (x_l[0][:, :, 0]); flow a batch-strided channel slice
    (up[:, :2]) whose [2,H,W] block is contiguous.  bounds_ws: per-problem bounds tables from splat_bounds_upsampled.
    spk_batch (N == 1, two problems): the packed results form ONE batch of two (sample k = problem k).
    -> list of fp32 tensors, of Spk tensors, or of (fp32, Spk); with spk_batch the two-sample Spk."""
    nd = len(imgs)
    assert 1 <= nd <= 2 and len(flows) == nd
    N, C, H, W = imgs[0].shape
    assert want_f32 or want_spk
    assert not want_spk or C > 3, "packed output needs the 16-channel configuration (C > 3)"
    d = SplatAccDesc()
    keep, outs = [], []
    batch = None
    if spk_batch:
        assert nd == 2 and N == 1 and want_spk and not want_f32
        batch = _spk_alloc(2, C, H, W, imgs[0].device)
    for k in range(nd):
        im, fl = imgs[k], flows[k]
        assert im.shape == (N, C, H, W) and fl.shape == (N, 2, H, W)
        im, ibs, ics = _planes(im, "img")
        if not fl.is_cuda:
            raise NotImplementedError("fldr softsplat has no CPU path (the reference has none either, softSplat.py:251-252)")
        if fl.dtype != torch.float32:
            raise TypeError("flow must be float32")
        if not fl[0].is_contiguous():
            fl = fl.contiguous()
        keep += [im, fl]
        d.img[k], d.img_bstride[k], d.img_cstride[k] = im.data_ptr(), ibs, ics
        d.flow[k], d.flow_bstride[k] = fl.data_ptr(), (fl.stride(0) if N > 1 else 0)
        mt = metrics[k] if metrics is not None else None
        if mt is not None:
            assert mt.shape == (N, 1, H, W)
            mt = mt.contiguous()
            _dev(mt, "metric")
            keep.append(mt)
        d.metric[k] = mt.data_ptr() if mt is not None else None
        if torch.is_tensor(bounds_ws):                       # one table for both problems (splat_bounds_upsampled_pair)
            ws = bounds_ws
        else:
            ws = bounds_ws[k] if bounds_ws is not None else (torch.empty(lib().fldr_softsplat_tile_ws_floats(N, H, W), device=im.device, dtype=torch.float32)
                                                            if H * W > 2304 else None)      # (small maps: no tables, see the header)
        keep.append(ws)
        d.ws[k] = ws.data_ptr() if ws is not None else None
        o32 = torch.empty(N, C, H, W, device=im.device, dtype=torch.float32) if want_f32 else None
        if batch is not None:
            osp = batch.sample(k)
        else:
            osp = _spk_alloc(N, C, H, W, im.device) if want_spk else None
        d.out_f32[k] = o32.data_ptr() if o32 is not None else None
        d.out_spk[k] = osp.ptr if osp is not None else None
        outs.append((o32, osp) if (want_f32 and want_spk) else (osp if want_spk else o32))
    d.nprob, d.N, d.C, d.H, d.W, d.mode = nd, N, C, H, W, _MODES[mode]
    d.flags = 0 if bounds_ws is None else (2 if torch.is_tensor(bounds_ws) else 1)
    assert not torch.is_tensor(bounds_ws) or nd == 2
    _check(lib().fldr_softsplat_acc64(ctypes.byref(d), _stream()), "fldr_softsplat_acc64")
    return batch if batch is not None else outs


def softsplat_gather(imgs, flows, metrics=None, mode="softmax", want_f32=False, want_spk=True):
    """FunctionSoftsplat of one or two (img [N,C,H,W], flow [N,2,H,W]) problems of the same shape as a deterministic gather
    (feature maps: C <= 48, at most 4096 tiles of 16x16).  img / flow may be batch-strided channel slices of larger tensors
    (feat[:, 48:], up[:, :2]): each sample's [C,H,W] / [2,H,W] block must be contiguous.  -> list of fp32 tensors, list of
    Spk tensors, or list of (fp32, Spk)."""
    nd = len(imgs)
    assert 1 <= nd <= 2 and len(flows) == nd
    N, C, H, W = imgs[0].shape
    d = SplatGatherDesc()
    keep, outs = [], []
    for k in range(nd):
        im, fl = imgs[k], flows[k]
        assert im.shape == (N, C, H, W) and fl.shape == (N, 2, H, W)
        if not im.is_cuda:
            raise NotImplementedError("fldr softsplat has no CPU path (the reference has none either, softSplat.py:251-252)")
        if im.dtype != torch.float32 or fl.dtype != torch.float32:
            raise TypeError("softsplat_gather takes float32 tensors")
        if not im[0].is_contiguous():
            im = im.contiguous()
        if not fl[0].is_contiguous():
            fl = fl.contiguous()
        keep += [im, fl]
        d.img[k], d.img_bstride[k] = im.data_ptr(), (im.stride(0) if N > 1 else 0)
        d.flow[k], d.flow_bstride[k] = fl.data_ptr(), (fl.stride(0) if N > 1 else 0)
        mt = metrics[k] if metrics is not None else None
        if mt is not None:
            mt = mt.contiguous()
            assert mt.shape == (N, 1, H, W)
            keep.append(mt)
        d.metric[k] = mt.data_ptr() if mt is not None else None
        o32 = torch.empty(N, C, H, W, device=im.device, dtype=torch.float32) if want_f32 else None
        osp = _spk_alloc(N, C, H, W, im.device) if want_spk else None
        d.out_f32[k] = o32.data_ptr() if o32 is not None else None
        d.out_spk[k] = osp.buf.data_ptr() if osp is not None else None
        outs.append((o32, osp) if (want_f32 and want_spk) else (osp if want_spk else o32))
    n = lib().fldr_softsplat_gather_ws_floats(nd, N, H, W)
    ws = torch.empty(max(int(n), 4), device=imgs[0].device, dtype=torch.float32)
    d.ws = ws.data_ptr()
    d.ndir, d.N, d.C, d.H, d.W, d.mode = nd, N, C, H, W, _MODES[mode]
    _check(lib().fldr_softsplat_gather(ctypes.byref(d), _stream()), "fldr_softsplat_gather")
    return outs


def softsplat_pair_spk(img_a, flow_a, img_b, flow_b, mode="softmax"):
    """The two feature splats of a level (one sample each, no metric) -> ONE packed batch of two (sample 0 = problem a); one
    memset + one normalisation launch for both (fldr_softsplat_pair_spk)."""
    N, C, H, W = img_a.shape
    assert N == 1 and img_b.shape == img_a.shape and flow_a.shape == (1, 2, H, W) and flow_b.shape == (1, 2, H, W)
    ia, ib, fa, fb = img_a.contiguous(), img_b.contiguous(), flow_a.contiguous(), flow_b.contiguous()
    ca = C + (0 if mode == "summation" else 1)
    scratch = torch.empty(2 * ca * H * W, device=ia.device, dtype=torch.float32)
    out = _spk_alloc(2, C, H, W, ia.device)
    _check(lib().fldr_softsplat_pair_spk(_dev(ia, "img"), _dev(fa, "flow"), _dev(ib, "img"), _dev(fb, "flow"), ctypes.c_void_p(out.ptr),
                                         _dev(scratch, "scratch"), C, H, W, _MODES[mode], _stream()), "fldr_softsplat_pair_spk")
    return out


def correlation_fwd(a, b):
    N, C, H, W = a.shape
    assert b.shape == a.shape
    out = torch.empty(N, 81, H, W, device=a.device, dtype=torch.float32)
    _check(lib().fldr_correlation_fwd(_dev(a, "first"), _dev(b, "second"), _dev(out, "out"), N, C, H, W, _stream()),
           "fldr_correlation_fwd")
    return out


def pca_project(planes, ev, mean, meanvec, want_f64=False, want_f32=True):
    P, H, W = planes.shape
    K = ev.shape[0]
    planes = planes.contiguous()
    o32 = torch.empty(P * K, H // 8, W // 8, device=planes.device, dtype=torch.float32) if want_f32 else None
    o64 = torch.empty(P * K, H // 8, W // 8, device=planes.device, dtype=torch.float64) if want_f64 else None
    mm = torch.empty(2, device=planes.device, dtype=torch.float64)
    code = lib().fldr_pca_project(_dev(planes, "planes"), _dev(ev, "EV", torch.float64), _dev(mean, "mean", torch.float64),
                                  _dev(meanvec, "mean_vec", torch.float64),
                                  _dev(o32, "out") if want_f32 else None,
                                  _dev(o64, "out64", torch.float64) if want_f64 else None,
                                  _dev(mm, "minmax", torch.float64), P, K, H, W, _stream())
    if code == -2:
        raise Exception("in to_pca_diff the image is not padded right." + str(H) + " " + str(W))   # pca_comp.py:487
    _check(code, "fldr_pca_project")
    return o32, o64, mm


def pca_project_stream(planes, ev, mean, meanvec, want_spk=False):
    """One-pass projection: -> (fp32 [P*K,h,w], fp64 [P*K,h,w], minmax, Spk of shape [1,P*K,h,w] or None)."""
    P, H, W = planes.shape
    K = ev.shape[0]
    planes = planes.contiguous()
    h, w = H // 8, W // 8
    o32 = torch.empty(P * K, h, w, device=planes.device, dtype=torch.float32)
    o64 = torch.empty(P * K, h, w, device=planes.device, dtype=torch.float64)
    mm = torch.empty(2, device=planes.device, dtype=torch.float64)
    spk = _spk_alloc(1, P * K, h, w, planes.device) if want_spk else None
    code = lib().fldr_pca_project_stream(_dev(planes, "planes"), _dev(ev, "EV", torch.float64), _dev(mean, "mean", torch.float64),
                                         _dev(meanvec, "mean_vec", torch.float64), _dev(o32, "out"), _dev(o64, "out64", torch.float64),
                                         ctypes.c_void_p(spk.buf.data_ptr()) if want_spk else None,
                                         _dev(mm, "minmax", torch.float64), P, K, H, W, _stream())
    if code == -2:
        raise Exception("in to_pca_diff the image is not padded right." + str(H) + " " + str(W))   # pca_comp.py:487
    _check(code, "fldr_pca_project_stream")
    return o32, o64, mm, spk


_PCA_TABLES = {}


def pca_table(ev, mean, meanvec):
    """Prepacked projection table for pca_project_pyramid, cached per (storage address, version) of the three parameters
    (callers pass fresh views such as EV8.detach()[:k] on every forward, so the cache cannot live on the tensor object)."""
    K = ev.shape[0]
    key = (ev.data_ptr(), ev._version, mean.data_ptr(), mean._version, meanvec.data_ptr(), meanvec._version, K, ev.device.index)
    hit = _PCA_TABLES.get(key)
    if hit is not None:
        return hit
    n = lib().fldr_pca_table_size(K)
    if n < 0:
        raise FldrError("unsupported number of PCA components %d" % K)
    tab = torch.empty(n, device=ev.device, dtype=torch.float64)
    _check(lib().fldr_pca_prepack(_dev(ev.detach().contiguous(), "EV", torch.float64), _dev(mean.detach().contiguous(), "mean", torch.float64),
                                  _dev(meanvec.detach().contiguous(), "mean_vec", torch.float64), _dev(tab, "table", torch.float64), K, _stream()),
           "fldr_pca_prepack")
    _prepack_done()
    if len(_PCA_TABLES) >= 8:
        _PCA_TABLES.clear()
    _PCA_TABLES[key] = tab
    return tab


ENC3_SPLIT = True    # enc3 as two 32-channel persistent problems on enc2's packed output
ENC3_PAIR = True      # ... in ONE launch (fldr_conv2d_s2_spk_pair); 0: two launches
DEC3_MFMA = True      # the fused dec3 + blend kernel reads dec2's packed output (matrix-core phase convolutions)
PCA_RAW_MIN_BYTES = 0      # levels of at least this many projection bytes are parked between the two passes (4K pyramid: 196.9 us none, 186.7 from 4 MB, 168.4 all)


def pca_project_pyramid(planes_list, ev, mean, meanvec, want_f32=True, want_spk=False, raw_min_bytes=None):
    """to_pca_diff(...).float() of every pyramid level in two launches (fLDRnet.py:133-146).  planes_list: [P,H_l,W_l]
    fp32 tensors.  -> (list of fp32 [P*K,h,w] or None, list of Spk [1,P*K,h,w] or None, minmax [n_levels,2])."""
    ev = ev.detach()
    K = ev.shape[0]
    tab = pca_table(ev, mean.detach(), meanvec.detach())
    n = len(planes_list)
    arr = (PcaLevel * n)()
    outs32, outsp, keep = [], [], []
    for i, pl in enumerate(planes_list):
        P, H, W = pl.shape
        if H % 8 or W % 8:
            raise Exception("in to_pca_diff the image is not padded right." + str(H) + " " + str(W))   # pca_comp.py:487
        pl = pl.contiguous()
        keep.append(pl)
        o32 = torch.empty(P * K, H // 8, W // 8, device=pl.device, dtype=torch.float32) if want_f32 else None
        osp = _spk_alloc(1, P * K, H // 8, W // 8, pl.device) if want_spk else None
        outs32.append(o32)
        outsp.append(osp)
        arr[i].planes = _dev(pl, "planes").value
        arr[i].out_f32 = o32.data_ptr() if o32 is not None else None
        arr[i].out_spk = osp.buf.data_ptr() if osp is not None else None
        arr[i].P, arr[i].H, arr[i].W = P, H, W
        nb = P * (H // 8) * (W // 8)
        if nb * K * 8 >= (PCA_RAW_MIN_BYTES if raw_min_bytes is None else raw_min_bytes):        # one read of the frames: the projections parked as fp64 between the passes (default: every level)
            raw = torch.empty(nb * K, device=pl.device, dtype=torch.float64)
            keep.append(raw)
            arr[i].raw_ws = raw.data_ptr()
    mm = torch.empty(n, 32, device=planes_list[0].device, dtype=torch.float64)      # each bound on a 128-byte line of its own
    _check(lib().fldr_pca_project_pyramid(arr, n, _dev(tab, "table", torch.float64), K, _dev(mm, "minmax", torch.float64), _stream()),
           "fldr_pca_project_pyramid")
    return (outs32 if want_f32 else None), (outsp if want_spk else None), mm[:, ::16]


def bwarp(x, flo, withmask=True):
    N, C, H, W = x.shape
    assert flo.shape == (N, 2, H, W)
    x, flo = x.contiguous(), flo.contiguous()
    out = torch.empty_like(x)
    _check(lib().fldr_bwarp(_dev(x, "x"), _dev(flo, "flo"), _dev(out, "out"), N, C, H, W, int(bool(withmask)), _stream()),
           "fldr_bwarp")
    return out


_TMODE = {None: 0, "1": 0, "t": 1, "1-t": 2}


def bwarp_tscaled(x, flo, t, x_scale, flo_scale, withmask=True):
    """bwarp(sx * x, sf * flo), sx/sf in {None, 't', '1-t'} per sample (fLDRnet.py:474-475)."""
    N, C, H, W = x.shape
    assert flo.shape == (N, 2, H, W)
    x, flo = x.contiguous(), flo.contiguous()
    t = t.reshape(N).contiguous().float()
    out = torch.empty_like(x)
    _check(lib().fldr_bwarp_tscaled(_dev(x, "x"), _dev(flo, "flo"), _dev(out, "out"), _dev(t, "t"), _TMODE[x_scale],
                                    _TMODE[flo_scale], N, C, H, W, int(bool(withmask)), _stream()), "fldr_bwarp_tscaled")
    return out


def resize_bilinear(x, H, W, mul=1.0):
    N, C, h, w = x.shape
    x = x.contiguous()
    out = torch.empty(N, C, H, W, device=x.device, dtype=torch.float32)
    _check(lib().fldr_resize_bilinear(_dev(x, "in"), _dev(out, "out"), N * C, h, w, H, W, float(mul), _stream()),
           "fldr_resize_bilinear")
    return out


def resize_bilinear_spk(x, H, W, mul=1.0):
    """resize_bilinear for C <= 8 channels that also returns the split-packed twin: -> (fp32 [N,C,H,W], Spk)."""
    N, C, h, w = x.shape
    assert C <= 8
    x = x.contiguous()
    out = torch.empty(N, C, H, W, device=x.device, dtype=torch.float32)
    sp = _spk_alloc(N, C, H, W, x.device)
    _check(lib().fldr_resize_bilinear_spk(_dev(x, "in"), _dev(out, "out"), ctypes.c_void_p(sp.ptr), N, C, h, w, H, W, float(mul), _stream()),
           "fldr_resize_bilinear_spk")
    return out, sp


# The upsampled level flow and the bounds tables of the feature splats that consume it in ONE launch (fldr_resize_bilinear_spk_bounds);
# 0: the two launches.  Bit-identical either way.
RESIZE_BOUNDS = True


def resize_bilinear_spk_bounds(x, H, W, mul=1.0):
    """resize_bilinear_spk(x, H, W, mul) + splat_bounds_upsampled_pair(x, None, "features", mul, H, W) of a [N,4,h,w] level flow
    in one launch: -> (fp32 [N,4,H,W], Spk, bounds workspace).  Falls back to the two launches when W >= 4 w."""
    N, C, h, w = x.shape
    assert C == 4
    x = x.contiguous()
    if not RESIZE_BOUNDS or W >= 4 * w:
        out, sp = resize_bilinear_spk(x, H, W, mul)
        return out, sp, splat_bounds_upsampled_pair(x, None, "features", mul, H, W)
    out = torch.empty(N, C, H, W, device=x.device, dtype=torch.float32)
    sp = _spk_alloc(N, C, H, W, x.device)
    ws = torch.empty(2 * lib().fldr_softsplat_tile_ws_floats(N, H, W), device=x.device, dtype=torch.float32)
    _check(lib().fldr_resize_bilinear_spk_bounds(_dev(x, "in"), _dev(out, "out"), ctypes.c_void_p(sp.ptr), _dev(ws, "ws"), N, h, w, H, W,
                                                 float(mul), _stream()), "fldr_resize_bilinear_spk_bounds")
    return out, sp, ws


def level0_prep(flow_lo, I0, I1, t, H, W, za0, za1, withmask=True, want_z=True, phase=3, state=None):
    """fLDRnet.py:400-479 minus the splats in one kernel.  flow_lo [N,4,h,w]; I0 / I1 [N,3,H,W] (batch-strided views of
    the [N,3,2,H,W] level-0 tensor, e.g. x[:, :, 0], are read in place: batch and channel strides are passed down).
    -> dict(z0, z1 (None unless want_z), flow_t0, flow_t1, flowback_0, flowback_1, im0_tot, im1_tot).
    phase=1 only fills z0 / z1 / flow_t0 / flow_t1 (what the splats need); a second call with phase=2 and state=<the dict
    returned by the first> fills flowback_* / im*_tot: the model runs the splats in between, so that enc1 reads those planes
    right behind their producer (Infinity-Cache resident) instead of after 1.3 GB of splat traffic."""
    if state is not None:
        d, out = state["_desc"], state
        d.phase = 2 | 4
        _check(lib().fldr_level0_prep(ctypes.byref(d), _stream()), "fldr_level0_prep")
        return out
    N, four, h, w = flow_lo.shape
    assert four == 4 and I0.shape == (N, 3, H, W) and I1.shape == (N, 3, H, W)
    flow_lo = flow_lo.contiguous()
    I0, i0b, i0c = _planes(I0, "I0")
    I1, i1b, i1c = _planes(I1, "I1")
    dev = flow_lo.device
    e = lambda c: torch.empty(N, c, H, W, device=dev, dtype=torch.float32)
    out = {"z0": e(1) if want_z else None, "z1": e(1) if want_z else None, "flow_t0": e(2), "flow_t1": e(2),
           "flowback_0": e(2), "flowback_1": e(2), "im0_tot": e(3), "im1_tot": e(3)}
    t = t.reshape(N).contiguous().float()
    d = PrepDesc()
    d.flow_lo = _dev(flow_lo, "flow_lo").value
    d.I0, d.I1 = I0.data_ptr(), I1.data_ptr()
    d.i0_bstride, d.i1_bstride, d.i0_cstride, d.i1_cstride = i0b, i1b, i0c, i1c
    d.t = _dev(t, "t").value
    for k, v in out.items():
        setattr(d, k, v.data_ptr() if v is not None else None)
    d.N, d.h, d.w, d.H, d.W = N, h, w, H, W
    d.mul, d.z_alpha0, d.z_alpha1, d.withmask = float(H / h), float(za0), float(za1), int(bool(withmask))
    ws = torch.empty(N * h * w * 4, device=dev, dtype=torch.float32)
    d.ws = ws.data_ptr()
    d.phase = phase
    _check(lib().fldr_level0_prep(ctypes.byref(d), _stream()), "fldr_level0_prep")
    out["_keep"] = (I0, I1, t, ws, flow_lo)
    out["_desc"] = d
    return out


def zmetric(self_img, other_img, flow, alpha):
    N, C, H, W = self_img.shape
    self_img, other_img, flow = self_img.contiguous(), other_img.contiguous(), flow.contiguous()
    z = torch.empty(N, 1, H, W, device=self_img.device, dtype=torch.float32)
    _check(lib().fldr_zmetric(_dev(self_img, "self"), _dev(other_img, "other"), _dev(flow, "flow"), float(alpha),
                              _dev(z, "z"), N, C, H, W, _stream()), "fldr_zmetric")
    return z


def _prepack_done():
    """A packed weight / table is cached and may next be consumed from a DIFFERENT stream (several pairs in flight):
    wait for the one-time prepack kernels here rather than racing on first use."""
    torch.cuda.current_stream().synchronize()


def conv_prepack(weight):
    """Repack an nn.Conv2d weight for fldr_conv2d.  The packed copy is cached ON the tensor object (an
    allocator may hand the same address to a different weight, so the address alone is not a key)."""
    hit = getattr(weight, "_fldr_pack", None)
    if hit is not None and hit[0] == (weight._version, weight.data_ptr()):
        return hit[1]
    cout, cin, k, k2 = weight.shape
    assert k == k2
    n = lib().fldr_conv_prepack_size(cout, cin, k)
    if n < 0:
        raise FldrError("unsupported convolution shape %s" % (tuple(weight.shape),))
    w = weight.detach().contiguous()
    wp = torch.empty(n, device=weight.device, dtype=torch.float32)
    _check(lib().fldr_conv_prepack(_dev(w, "weight"), _dev(wp, "wpack"), cout, cin, k, _stream()), "fldr_conv_prepack")
    _prepack_done()
    weight._fldr_pack = ((weight._version, weight.data_ptr()), wp)
    return wp


# Arithmetic of the 3x3 stride-1 convolutions: "split" = 3 x fp16-split MFMA (fp32-equivalent accuracy, default),
# "fp32" = exact fp32 MFMA, "fp16" = fp16 inputs with fp32 accumulation (BASELINE config 5; NOT fp32-equivalent).
# Stride-2 4x4 convolutions always use the exact fp32 MFMA kernel.
CONV_PRECISION = os.environ.get("FLDR_CONV_PRECISION", "split")


def range_status(reset=True):
    """True if an activation left the range of the fp16 hi/lo split (|x| >= 65504, or NaN) since the last reset: such
    values SATURATE (finite, never inf / NaN from finite inputs) and are flagged here.  Synchronises the device."""
    return bool(device_status(reset) & STATUS_RANGE)


STATUS_RANGE, STATUS_RING_TIMEOUT = 1, 2


def device_status(reset=True):
    """Bit 0 (STATUS_RANGE): fldr_range_status — an activation was saturated by the fp16 split; bit 1 (STATUS_RING_TIMEOUT):
    fldr_ring_status — a bounded wait of the convolution ring expired (that convolution's output is not to be trusted).  Two entry
    points of the C ABI (a data problem and a library fault have different remedies); synchronises the device."""
    r = lib().fldr_range_status(int(bool(reset)))
    if r < 0:
        raise FldrError("fldr_range_status failed (%d)" % r)
    to = lib().fldr_ring_status(int(bool(reset)))
    if to < 0:
        raise FldrError("fldr_ring_status failed (%d)" % to)
    return (STATUS_RANGE if r else 0) | (STATUS_RING_TIMEOUT if to else 0)


def check_range():
    """Raise if the split-precision convolutions saturated an activation (see range_status); the remedy is
    FLDR_CONV_PRECISION=fp32 (exact fp32 MFMA everywhere)."""
    v = device_status(reset=True)
    if v & STATUS_RING_TIMEOUT:
        raise FldrError("a bounded wait of the 3x3 convolution ring expired: a wave ran on with operands that had not landed, "
                        "the outputs since the last check are not to be trusted")
    if v & STATUS_RANGE:
        raise FldrError("an activation exceeded the fp16 split range (|x| >= 65504 or NaN) and was saturated; "
                        "rerun with FLDR_CONV_PRECISION=fp32")


_status_words = {}          # (library handle id, device ordinal) -> ctypes pointer to the two host words of fldr_status_word


def status_words():
    """The current device's host-visible status words of the loaded library ([0] range, [1] ring; fldr_status_word): pinned host memory
    the kernels store into when the event happens.  The first call per (library, device) allocates and binds the block (synchronises)."""
    l = lib()
    key = (id(l), torch.cuda.current_device())
    p = _status_words.get(key)
    if p is None:
        p = ctypes.POINTER(ctypes.c_int)()
        _check(l.fldr_status_word(ctypes.byref(p)), "fldr_status_word")
        _status_words[key] = p
    return p


def poll_status():
    """The fault flags WITHOUT a synchronisation (what DCTXVFInet.forward does on entry): raises — through check_range(), which then
    synchronises and resets — if a kernel of an EARLIER call stored a flag; free otherwise (two reads of host memory).  A fault of forward
    n therefore raises at the latest in forward n + 1; frames written after a ring fault are NaN in any case (include/fldr_hip.h)."""
    p = status_words()
    if p[0] or p[1]:
        check_range()
        raise FldrError("a device-side fault flag was set (status words %d, %d) but the device status reads clean" % (p[0], p[1]))


def use_spk():
    """True when the 3x3 stride-1 convolutions run on split-packed activations (every precision but exact fp32)."""
    return CONV_PRECISION in ("split", "fp16")


def conv_split_prepack(weight):
    hit = getattr(weight, "_fldr_pack_split", None)
    if hit is not None and hit[0] == (weight._version, weight.data_ptr()):
        return hit[1]
    cout, cin, k, _ = weight.shape
    n = lib().fldr_conv_split_prepack_size(cout, cin)
    if n < 0:
        raise FldrError("unsupported convolution shape %s" % (tuple(weight.shape),))
    w = weight.detach().contiguous()
    wp = torch.empty(n, device=weight.device, dtype=torch.float32)
    _check(lib().fldr_conv_split_prepack(_dev(w, "weight"), _dev(wp, "wpack"), cout, cin, _stream()), "fldr_conv_split_prepack")
    _prepack_done()
    weight._fldr_pack_split = ((weight._version, weight.data_ptr()), wp)
    return wp


def conv_s2_prepack(weight):
    hit = getattr(weight, "_fldr_pack_s2", None)
    if hit is not None and hit[0] == (weight._version, weight.data_ptr()):
        return hit[1]
    cout, cin, k, _ = weight.shape
    n = lib().fldr_conv_s2_prepack_size(cout, cin)
    if n < 0:
        raise FldrError("unsupported convolution shape %s" % (tuple(weight.shape),))
    w = weight.detach().contiguous()
    wp = torch.empty(n, device=weight.device, dtype=torch.float32)
    _check(lib().fldr_conv_s2_prepack(_dev(w, "weight"), _dev(wp, "wpack"), cout, cin, _stream()), "fldr_conv_s2_prepack")
    _prepack_done()
    weight._fldr_pack_s2 = ((weight._version, weight.data_ptr()), wp)
    return wp


def conv2d(srcs, weight, bias, stride=1, relu=False, residual=None, cout_store=None, up2=None, out=None, precision=None,
           want_f32=True, want_spk=False):
    """conv(cat(srcs, 1)) with optional fused nearest-x2 read per source, ReLU and post-activation residual.
    want_spk (exact fp32-MFMA kernels only, i.e. the stride-2 encoders): also / only emit the split-packed twin of the
    output for a following conv2d_spk; returns fp32, Spk or (fp32, Spk) like conv2d_spk.

    srcs: list of [N,c_s,H_s,W_s] fp32 tensors whose (N, c, h, w) block may be a batch-strided view
    (e.g. feat[:, :48]) as long as each sample's [c,h,w] block is contiguous."""
    cout, cin, k, _ = weight.shape
    up2 = up2 or [False] * len(srcs)
    N = srcs[0].shape[0]
    Hin = srcs[0].shape[2] * (2 if up2[0] else 1)
    Win = srcs[0].shape[3] * (2 if up2[0] else 1)
    d = ConvDesc()
    keep = []
    csum = 0
    prec = precision or CONV_PRECISION
    if prec not in ("split", "fp32", "fp16"):
        raise ValueError("precision must be split, fp32 or fp16")
    split = prec in ("split", "fp16") and k == 3 and stride == 1
    # stride-2 4x4 encoders: 3 x fp16 split as well unless exact fp32 is asked for ("fp16" has no hi-only variant here)
    s2 = prec in ("split", "fp16") and k == 4 and stride == 2 and cout <= 64 and residual is None and not any(up2)
    for i, (s, u) in enumerate(zip(srcs, up2)):
        if not s.is_cuda:
            raise NotImplementedError("fldr conv2d has no CPU path")
        if s.dtype != torch.float32:
            raise TypeError("conv source must be float32")
        if s2:
            s, _, d.src_cstride[i] = _planes(s, "conv source")          # channel-strided views (I0 / I1) are read in place
        elif s[0].is_contiguous() is False:
            s = s.contiguous()
        keep.append(s)
        assert s.shape[0] == N and s.shape[2] * (2 if u else 1) == Hin and s.shape[3] * (2 if u else 1) == Win
        d.src[i] = s.data_ptr()
        d.src_bstride[i] = s.stride(0) if N > 1 else 0
        d.src_c[i] = s.shape[1]
        d.src_up2[i] = int(bool(u))
        csum += s.shape[1]
    assert csum == cin, "concatenated channels %d != weight cin %d" % (csum, cin)
    d.n_src = len(srcs)
    if k == 3 and stride == 1:
        Hout, Wout = Hin, Win
    elif k == 4 and stride == 2:
        Hout, Wout = (Hin + 2 - 4) // 2 + 1, (Win + 2 - 4) // 2 + 1
    else:
        raise FldrError("unsupported convolution k=%d stride=%d" % (k, stride))
    cs = cout if cout_store is None else cout_store
    if want_spk and split:
        raise FldrError("a split-packed output of a 3x3 stride-1 convolution comes from conv2d_spk")
    want_f32 = want_f32 or not want_spk or residual is not None
    outp = _spk_alloc(N, cs, Hout, Wout, srcs[0].device) if want_spk else None
    if out is None and want_f32:
        out = torch.empty(N, cs, Hout, Wout, device=srcs[0].device, dtype=torch.float32)
    wp = conv_split_prepack(weight) if split else (conv_s2_prepack(weight) if s2 else conv_prepack(weight))
    d.wpack = wp.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    if residual is not None:
        residual = residual.contiguous()
        assert residual.shape == out.shape
        d.residual = residual.data_ptr()
    d.out = _dev(out, "out").value if out is not None else None
    d.out_spk = outp.buf.data_ptr() if outp is not None else None
    d.N, d.cin, d.cout, d.cout_store = N, cin, cout, cs
    d.Hin, d.Win, d.Hout, d.Wout = Hin, Win, Hout, Wout
    d.ksize, d.stride, d.relu, d.precision = k, stride, int(bool(relu)), (1 if (split and prec == "fp16") else 0)
    if split:
        _check(lib().fldr_conv2d_split(ctypes.byref(d), _stream()), "fldr_conv2d_split")
    elif s2:
        _check(lib().fldr_conv2d_s2_split(ctypes.byref(d), _stream()), "fldr_conv2d_s2_split")
    else:
        _check(lib().fldr_conv2d(ctypes.byref(d), _stream()), "fldr_conv2d")
    if want_spk:
        return (out, outp) if out is not None else outp
    return out


# ---------------------------------------------------------------------------------------------
# split-packed activations (include/fldr_hip.h: fldr_spk_*): the layout between convolutions
# ---------------------------------------------------------------------------------------------

class Spk:
    """A logical [N,C,H,W] fp32 activation stored split-packed: [N][ceil(C/8)][hi,lo][H*W][8 x fp16]."""
    __slots__ = ("buf", "shape", "offset", "bstride")

    def __init__(self, buf, shape, offset=0, bstride=None):
        self.buf, self.shape, self.offset = buf, tuple(shape), offset
        N, C, H, W = self.shape
        self.bstride = ((C + 7) // 8) * 2 * H * W * 16 if bstride is None else bstride      # bytes between samples

    @property
    def device(self):
        return self.buf.device

    @property
    def ptr(self):
        return self.buf.data_ptr() + self.offset

    def narrow(self, c0, c):
        """Channels [c0, c0+c) as a view (the feat[:, :48] / feat[:, 48:] split of fLDRnet.py:368-370): c0 must be a
        multiple of 8; a view that does not end on a group boundary must end at the tensor's last channel."""
        N, C, H, W = self.shape
        assert c0 % 8 == 0 and c > 0 and c0 + c <= C and (c % 8 == 0 or c0 + c == C)
        return Spk(self.buf, (N, c, H, W), self.offset + (c0 // 8) * 2 * H * W * 16, self.bstride)

    def sample(self, n):
        """Sample n as a one-sample view."""
        N, C, H, W = self.shape
        assert 0 <= n < N
        return Spk(self.buf, (1, C, H, W), self.offset + n * self.bstride, self.bstride)

    def channel_halves(self):
        """A one-sample tensor of 2c channels seen as TWO samples of c channels (sample 0 = the first half): lets the two
        convolutions of fLDRnet.py:389 that share their weights run as one batch-of-2 launch."""
        N, C, H, W = self.shape
        assert N == 1 and C % 16 == 0
        return Spk(self.buf, (2, C // 2, H, W), self.offset, (C // 16) * 2 * H * W * 16)

    def float(self):
        """hi + lo as an fp32 NCHW tensor (22 significant bits; tests and debugging)."""
        N, C, H, W = self.shape
        assert self.offset == 0 and self.bstride == ((C + 7) // 8) * 2 * H * W * 16, "float() of a channel view is not supported"
        out = torch.empty(N, C, H, W, device=self.buf.device, dtype=torch.float32)
        _check(lib().fldr_spk_unpack(ctypes.c_void_p(self.buf.data_ptr()), _dev(out, "out"), N, C, H, W, _stream()), "fldr_spk_unpack")
        return out


def _spk_alloc(N, C, H, W, device):
    nb = lib().fldr_spk_bytes(C, H, W)
    return Spk(torch.empty(N * nb // 2, device=device, dtype=torch.float16), (N, C, H, W))


def spk_pack(x):
    """fp32 [N,C,H,W] (batch-strided views allowed) -> Spk."""
    if isinstance(x, Spk):
        return x
    if not x.is_cuda:
        raise NotImplementedError("fldr spk_pack has no CPU path")
    if x.dtype != torch.float32:
        raise TypeError("spk_pack source must be float32")
    if not x[0].is_contiguous():
        x = x.contiguous()
    N, C, H, W = x.shape
    out = _spk_alloc(N, C, H, W, x.device)
    _check(lib().fldr_spk_pack(ctypes.c_void_p(x.data_ptr()), x.stride(0) if N > 1 else 0, ctypes.c_void_p(out.buf.data_ptr()),
                               N, C, H, W, _stream()), "fldr_spk_pack")
    return out


def conv_spk_prepack(weight):
    hit = getattr(weight, "_fldr_pack_spk", None)
    if hit is not None and hit[0] == (weight._version, weight.data_ptr()):
        return hit[1]
    cout, cin, k, _ = weight.shape
    n = lib().fldr_conv_spk_prepack_size(cout, cin)
    if n < 0:
        raise FldrError("unsupported convolution shape %s" % (tuple(weight.shape),))
    w = weight.detach().contiguous()
    wp = torch.empty(n, device=weight.device, dtype=torch.float32)
    _check(lib().fldr_conv_spk_prepack(_dev(w, "weight"), _dev(wp, "wpack"), cout, cin, _stream()), "fldr_conv_spk_prepack")
    _prepack_done()
    weight._fldr_pack_spk = ((weight._version, weight.data_ptr()), wp)
    return wp


def conv2d_spk(srcs, weight, bias, relu=False, residual=None, cout_store=None, up2=None, want_f32=True, want_spk=False,
               precision=None):
    """3x3 / stride-1 conv(cat(srcs, 1)) on split-packed sources (fp32 tensors are packed on the fly).  Returns the fp32
    NCHW tensor, the Spk tensor, or (fp32, Spk) when both are asked for.  Same arithmetic as conv2d(precision='split')."""
    cout, cin, k, _ = weight.shape
    assert k == 3
    up2 = up2 or [False] * len(srcs)
    packed = [spk_pack(s) for s in srcs]
    N = packed[0].shape[0]
    H = packed[0].shape[2] * (2 if up2[0] else 1)
    W = packed[0].shape[3] * (2 if up2[0] else 1)
    d = SpkConvDesc()
    csum = 0
    for i, (s, u) in enumerate(zip(packed, up2)):
        n, c, h, w = s.shape
        assert n == N and h * (2 if u else 1) == H and w * (2 if u else 1) == W
        if i + 1 < len(packed) and c % 8:
            raise FldrError("every packed source but the last needs a multiple of 8 channels (got %d)" % c)
        d.src[i] = s.ptr
        d.src_bstride[i] = s.bstride if N > 1 else 0
        d.src_c[i] = c
        d.src_up2[i] = int(bool(u))
        csum += c
    assert csum == cin, "concatenated channels %d != weight cin %d" % (csum, cin)
    d.n_src = len(packed)
    cs = cout if cout_store is None else cout_store
    dev = packed[0].device
    wp = conv_spk_prepack(weight)
    d.wpack = wp.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    res_packed = isinstance(residual, Spk)          # packed residual (value = hi + lo): no fp32 output needed for it
    out32 = torch.empty(N, cs, H, W, device=dev, dtype=torch.float32) if (want_f32 or (residual is not None and not res_packed)) else None
    outp = _spk_alloc(N, cs, H, W, dev) if want_spk else None
    if res_packed:
        assert tuple(residual.shape) == (N, cs, H, W)
        d.residual = residual.ptr
    elif residual is not None:
        residual = residual.contiguous()
        assert residual.shape == out32.shape
        d.residual = residual.data_ptr()
    d.out_f32 = out32.data_ptr() if out32 is not None else None
    d.out_spk = outp.buf.data_ptr() if outp is not None else None
    prec = precision or CONV_PRECISION
    d.N, d.cin, d.cout, d.cout_store, d.H, d.W = N, cin, cout, cs, H, W
    d.relu, d.precision = int(bool(relu)), (1 if prec == "fp16" else 0) | (2 if res_packed else 0)
    _check(lib().fldr_conv2d_spk(ctypes.byref(d), _stream()), "fldr_conv2d_spk")
    if want_f32 and want_spk:
        return out32, outp
    return outp if want_spk else out32


def conv2d_s2_spk(src, weight, bias, relu=False, want_f32=True, want_spk=False):
    """The stride-2 4x4 convolution on ONE split-packed source (fldr_conv2d_s2_spk: an encoder reading the previous encoder's packed
    output, whose fp32 copy then need not exist).  -> fp32, Spk or (fp32, Spk); equal to conv2d(..., stride=2) on the unpacked
    values up to fp32 accumulation rounding.  Raises FldrError (shape) where the persistent kernel does not apply: check s2_spk_ok first."""
    cout, cin, k, _ = weight.shape
    assert k == 4 and isinstance(src, Spk) and src.shape[1] == cin
    N, _, Hin, Win = src.shape
    Hout, Wout = (Hin + 2 - 4) // 2 + 1, (Win + 2 - 4) // 2 + 1
    d = ConvDesc()
    d.src[0], d.src_bstride[0], d.src_c[0], d.src_up2[0], d.n_src = src.ptr, (src.bstride if N > 1 else 0), cin, 0, 1
    want_f32 = want_f32 or not want_spk
    out = torch.empty(N, cout, Hout, Wout, device=src.device, dtype=torch.float32) if want_f32 else None
    outp = _spk_alloc(N, cout, Hout, Wout, src.device) if want_spk else None
    wp = conv_s2_prepack(weight)
    d.wpack = wp.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.out = out.data_ptr() if out is not None else None
    d.out_spk = outp.buf.data_ptr() if outp is not None else None
    d.N, d.cin, d.cout, d.cout_store = N, cin, cout, cout
    d.Hin, d.Win, d.Hout, d.Wout = Hin, Win, Hout, Wout
    d.ksize, d.stride, d.relu, d.precision = 4, 2, int(bool(relu)), 0
    _check(lib().fldr_conv2d_s2_spk(ctypes.byref(d), _stream()), "fldr_conv2d_s2_spk")
    if want_spk:
        return (out, outp) if out is not None else outp
    return out


def conv2d_s2_spk_pair(src, halves, relu=False):
    """Two stride-2 4x4 convolutions of the SAME packed source in ONE launch (fldr_conv2d_s2_spk_pair): halves = [(weight, bias), (weight,
    bias)] with equal shapes — enc3's two 32-channel halves.  -> [Spk, Spk], the bits of two conv2d_s2_spk(..., want_spk=True) calls."""
    assert isinstance(src, Spk) and len(halves) == 2
    N, cin, Hin, Win = src.shape
    Hout, Wout = (Hin + 2 - 4) // 2 + 1, (Win + 2 - 4) // 2 + 1
    descs, outs, keep = [], [], []
    for weight, bias in halves:
        cout, ci, k, _ = weight.shape
        assert k == 4 and ci == cin and tuple(weight.shape) == tuple(halves[0][0].shape)
        d = ConvDesc()
        d.src[0], d.src_bstride[0], d.src_c[0], d.src_up2[0], d.n_src = src.ptr, (src.bstride if N > 1 else 0), cin, 0, 1
        outp = _spk_alloc(N, cout, Hout, Wout, src.device)
        wp = conv_s2_prepack(weight)
        keep.append(wp)
        d.wpack = wp.data_ptr()
        d.bias = bias.data_ptr() if bias is not None else None
        d.out, d.out_spk = None, outp.buf.data_ptr()
        d.N, d.cin, d.cout, d.cout_store = N, cin, cout, cout
        d.Hin, d.Win, d.Hout, d.Wout = Hin, Win, Hout, Wout
        d.ksize, d.stride, d.relu, d.precision = 4, 2, int(bool(relu)), 0
        descs.append(d)
        outs.append(outp)
    _check(lib().fldr_conv2d_s2_spk_pair(ctypes.byref(descs[0]), ctypes.byref(descs[1]), _stream()), "fldr_conv2d_s2_spk_pair")
    return outs


def s2_spk_ok(weight):
    """Does fldr_conv2d_s2_spk take this layer (cin a multiple of 8, <= 64; all chunks' weights + two stages within 80 KB of LDS)?"""
    cout, cin, k, _ = weight.shape
    if k != 4 or cin % 8 or cin > 64 or cout > 32 or CONV_PRECISION == "fp32":
        return False
    w_bytes = (2 if cout <= 16 else 4) * 2 * 1024                  # S2Cfg<MT, 1>::W_BYTES: steps x (hi, lo) KB
    return (cin // 4) * w_bytes + 2 * 4 * 1440 * 4 <= (80 if cout <= 16 else 156) * 1024      # (17..32 outputs: one workgroup per CU above 80 KB)


def conv2d_spk_levels(srcs, weight, bias, relu=False, residuals=None, want_f32=True, want_spk=False, precision=None):
    """The same 3x3 convolution over a LIST of one-sample packed tensors of different sizes (the pyramid levels) in ONE launch
    (fldr_conv2d_spk_levels).  -> list of fp32 tensors, of Spk tensors, or of (fp32, Spk): the bits of per-level conv2d_spk calls."""
    cout, cin, k, _ = weight.shape
    assert k == 3 and 1 <= len(srcs) <= 8
    packed = [spk_pack(x) for x in srcs]
    descs = (SpkConvDesc * len(packed))()
    wp = conv_spk_prepack(weight)
    prec = precision or CONV_PRECISION
    outs, keep = [], []
    for l, sp in enumerate(packed):
        n, c, h, w = sp.shape
        assert n == 1 and c == cin
        d = descs[l]
        d.src[0], d.src_bstride[0], d.src_c[0], d.src_up2[0], d.n_src = sp.ptr, 0, c, 0, 1
        d.wpack = wp.data_ptr()
        d.bias = bias.data_ptr() if bias is not None else None
        res = residuals[l] if residuals is not None else None
        res_packed = isinstance(res, Spk)           # split-packed residual (value = hi + lo; every level or none)
        o32 = torch.empty(1, cout, h, w, device=sp.device, dtype=torch.float32) if (want_f32 or (res is not None and not res_packed)) else None
        osp = _spk_alloc(1, cout, h, w, sp.device) if want_spk else None
        if res_packed:
            assert tuple(res.shape) == (1, cout, h, w)
            keep.append(res)
            d.residual = res.ptr
        elif res is not None:
            res = res.contiguous()
            assert res.shape == o32.shape
            keep.append(res)
            d.residual = res.data_ptr()
        d.out_f32 = o32.data_ptr() if o32 is not None else None
        d.out_spk = osp.buf.data_ptr() if osp is not None else None
        d.N, d.cin, d.cout, d.cout_store, d.H, d.W = 1, cin, cout, cout, h, w
        d.relu, d.precision = int(bool(relu)), (1 if prec == "fp16" else 0) | (2 if res_packed else 0)
        outs.append((o32, osp) if (want_f32 and want_spk) else (osp if want_spk else o32))
    _check(lib().fldr_conv2d_spk_levels(descs, len(packed), _stream()), "fldr_conv2d_spk_levels")
    return outs


def synth_tail(refine, cands, t, T_param, out_dtype=torch.float64):
    N, six, H, W = refine.shape
    assert six == 6 and len(cands) == 6
    refine = refine.contiguous()
    ptrs = (ctypes.c_void_p * 6)()
    strides = (ctypes.c_int64 * 6)()
    keep = []
    for k, c in enumerate(cands):
        assert c.shape == (N, 3, H, W)
        if not c[0].is_contiguous():
            c = c.contiguous()
        keep.append(c)
        ptrs[k] = _dev(c[0], "candidate").value
        strides[k] = c.stride(0) if N > 1 else 0
    t = t.reshape(N).contiguous().float()
    out = torch.empty(N, 3, H, W, device=refine.device, dtype=out_dtype)
    o64 = _dev(out, "out", torch.float64) if out_dtype == torch.float64 else None
    o32 = _dev(out, "out", torch.float32) if out_dtype == torch.float32 else None
    _check(lib().fldr_synth_tail(_dev(refine, "refine"), ptrs, strides, _dev(t, "t"), float(T_param), o64, o32,
                                 N, H, W, _stream()), "fldr_synth_tail")
    return out


def dec3_synth(d2, weight, bias, cands, t, T_param, out_dtype=torch.float64, want_refine=False):
    """Fused dec3 (on the nearest-x2 upsampled dec2 output) + softmax/T + blend (fLDRnet.py:642-643, 511-524).  d2: dec2's
    output as an fp32 tensor (phase convolutions in fp32 on the vector ALUs) or split-packed (Spk: on the fp16 matrix cores with
    the 3 x fp16 split, the model's path)."""
    N, cin, h, w = d2.shape
    assert tuple(weight.shape) == (6, 16, 3, 3) and cin == 16 and len(cands) == 6
    H, W = 2 * h, 2 * w
    packed = isinstance(d2, Spk)
    key = "_fldr_dec3m" if packed else "_fldr_dec3"
    hit = getattr(weight, key, None)
    if hit is None or hit[0] != (weight._version, weight.data_ptr()):
        if packed:
            weff = torch.empty(int(lib().fldr_dec3_prepack_spk_size()), device=weight.device, dtype=torch.float32)
            _check(lib().fldr_dec3_prepack_spk(_dev(weight.detach().contiguous(), "weight"), _dev(weff, "wm"), _stream()), "fldr_dec3_prepack_spk")
        else:
            weff = torch.empty(1536, device=weight.device, dtype=torch.float32)
            _check(lib().fldr_dec3_prepack(_dev(weight.detach().contiguous(), "weight"), _dev(weff, "weff"), _stream()), "fldr_dec3_prepack")
        _prepack_done()
        setattr(weight, key, ((weight._version, weight.data_ptr()), weff))
        hit = getattr(weight, key)
    weff = hit[1]
    if packed:
        assert d2.bstride == 4 * h * w * 16, "dec3_synth: the packed source must be a whole 16-channel tensor"
    else:
        d2 = d2.contiguous()
    ptrs = (ctypes.c_void_p * 6)()
    strides = (ctypes.c_int64 * 6)()
    cstrides = (ctypes.c_int64 * 6)()
    keep = []
    for k, c in enumerate(cands):
        assert c.shape == (N, 3, H, W)
        c, strides[k], cstrides[k] = _planes(c, "candidate")
        keep.append(c)
        ptrs[k] = c.data_ptr()
    t = t.reshape(N).contiguous().float()
    out = torch.empty(N, 3, H, W, device=d2.device, dtype=out_dtype)
    refine = torch.empty(N, 6, H, W, device=d2.device, dtype=torch.float32) if want_refine else None
    o64 = _dev(out, "out", torch.float64) if out_dtype == torch.float64 else None
    o32 = _dev(out, "out", torch.float32) if out_dtype == torch.float32 else None
    if packed:
        _check(lib().fldr_dec3_synth_spk(ctypes.c_void_p(d2.ptr), _dev(weff, "wm"), _dev(bias.detach(), "bias"), ptrs, strides, cstrides,
                                         _dev(t, "t"), float(T_param), o64, o32, _dev(refine, "refine") if want_refine else None,
                                         N, H, W, _stream()), "fldr_dec3_synth_spk")
    else:
        _check(lib().fldr_dec3_synth_strided(_dev(d2, "d2"), _dev(weff, "weff"), _dev(bias.detach(), "bias"), ptrs, strides, cstrides,
                                             _dev(t, "t"), float(T_param), o64, o32, _dev(refine, "refine") if want_refine else None,
                                             N, H, W, _stream()), "fldr_dec3_synth_strided")
    return (out, refine) if want_refine else out


# dec2 -> dec3 -> blend in one persistent producer / consumer kernel (fldr_dec23_synth): dec2's packed output never reaches HBM
# (FLDR_DEC23=0: conv2d_spk + dec3_synth, the form the fused kernel is tested against)
DEC23_FUSED = os.environ.get("FLDR_DEC23", "1") != "0"


def dec23_synth(dec1p, enc1p, w2, b2, w3, b3, cands, t, T_param, out_dtype=torch.float64, u8_crop=None):
    """PCARefineUNet.dec2 (on cat(nearest-x2(dec1), enc1), ReLU) + dec3 (on the nearest-x2 upsampled result) + softmax / T + blend
    (fLDRnet.py:638-643, 511-524) in one kernel.  dec1p: Spk [N,32,H/4,W/4]; enc1p: Spk [N,16,H/2,W/2]; w2 [16,48,3,3], w3 [6,16,3,3].
    u8_crop = (Hc, Wc) (Wc even): instead of the fp64 / fp32 frame, the frame cropped to Hc x Wc and rounded to 8 bits
    (frame_metrics' arithmetic) as a uint8 tensor [N,3,Hc,Wc]."""
    N, c1, h4, w4 = dec1p.shape
    N2, c2, h, w = enc1p.shape
    assert isinstance(dec1p, Spk) and isinstance(enc1p, Spk) and N == N2 and c1 == 32 and c2 == 16 and h == 2 * h4 and w == 2 * w4
    assert tuple(w2.shape) == (16, 48, 3, 3) and tuple(w3.shape) == (6, 16, 3, 3) and len(cands) == 6
    assert dec1p.bstride == 8 * h4 * w4 * 16 and enc1p.bstride == 4 * h * w * 16, "dec23_synth: the packed sources must be whole tensors"
    H, W = 2 * h, 2 * w
    hit = getattr(w2, "_fldr_dec23", None)
    if hit is None or hit[0] != (w2._version, w2.data_ptr()):
        wp = torch.empty(int(lib().fldr_dec23_prepack_size()), device=w2.device, dtype=torch.float32)
        _check(lib().fldr_dec23_prepack(_dev(w2.detach().contiguous(), "weight"), _dev(wp, "wpack"), _stream()), "fldr_dec23_prepack")
        _prepack_done()
        w2._fldr_dec23 = hit = ((w2._version, w2.data_ptr()), wp)
    w2p = hit[1]
    hit3 = getattr(w3, "_fldr_dec3m", None)
    if hit3 is None or hit3[0] != (w3._version, w3.data_ptr()):
        wm = torch.empty(int(lib().fldr_dec3_prepack_spk_size()), device=w3.device, dtype=torch.float32)
        _check(lib().fldr_dec3_prepack_spk(_dev(w3.detach().contiguous(), "weight"), _dev(wm, "wm"), _stream()), "fldr_dec3_prepack_spk")
        _prepack_done()
        w3._fldr_dec3m = hit3 = ((w3._version, w3.data_ptr()), wm)
    w3m = hit3[1]
    ptrs = (ctypes.c_void_p * 6)()
    strides = (ctypes.c_int64 * 6)()
    cstrides = (ctypes.c_int64 * 6)()
    keep = []
    for k, c in enumerate(cands):
        assert c.shape == (N, 3, H, W)
        c, strides[k], cstrides[k] = _planes(c, "candidate")
        keep.append(c)
        ptrs[k] = c.data_ptr()
    t = t.reshape(N).contiguous().float()
    o64 = o32 = o8 = None
    hc = wc = 0
    if u8_crop is not None:
        hc, wc = int(u8_crop[0]), int(u8_crop[1])
        assert 0 < hc <= H and 0 < wc <= W and wc % 2 == 0
        out = torch.empty(N, 3, hc, wc, device=w2.device, dtype=torch.uint8)
        o8 = _dev(out, "out", torch.uint8)
    else:
        out = torch.empty(N, 3, H, W, device=w2.device, dtype=out_dtype)
        o64 = _dev(out, "out", torch.float64) if out_dtype == torch.float64 else None
        o32 = _dev(out, "out", torch.float32) if out_dtype == torch.float32 else None
    _check(lib().fldr_dec23_synth(ctypes.c_void_p(dec1p.ptr), ctypes.c_void_p(enc1p.ptr), _dev(w2p, "w2pack"), _dev(b2.detach(), "bias2"), _dev(w3m, "w3m"),
                                  _dev(b3.detach(), "bias3"), ptrs, strides, cstrides, _dev(t, "t"), float(T_param), o64, o32, o8, hc, wc, N, H, W, _stream()),
           "fldr_dec23_synth")
    return out


INGEST_FUSED = True     # ingest + all pyramid levels in one launch (0: one launch per level)


def ingest_pyramid(frames_u8, n_levels=6):
    """uint8 frames [B,2,3,H,W] on the device -> the model's normInput list (level i: [B,3,2,Hp/2^i,Wp/2^i] fp32):
    normalisation, reflect padding and the direct bicubic pyramid of main.py:840-856, all on the GPU."""
    B, T, C, H, W = frames_u8.shape
    assert T == 2 and C == 3
    div = (2 ** (n_levels - 1)) * 8
    Hp, Wp = (H + div - 1) // div * div, (W + div - 1) // div * div
    u8 = frames_u8.contiguous()
    if INGEST_FUSED and n_levels <= 7 and Wp % 4 == 0:
        # one launch: the uint8 frames are read once, every level is written from a staged 64 x 64 tile (fldr_ingest_pyramid_u8)
        pyr = [torch.empty(B, 3, 2, Hp >> i, Wp >> i, device=u8.device, dtype=torch.float32) for i in range(n_levels)]
        ptrs = (ctypes.c_void_p * n_levels)(*[p.data_ptr() for p in pyr])
        _check(lib().fldr_ingest_pyramid_u8(_dev(u8, "frames", torch.uint8), ptrs, n_levels, B, H, W, Hp, Wp, _stream()), "fldr_ingest_pyramid_u8")
        return pyr
    lv0 = torch.empty(B, 3, 2, Hp, Wp, device=u8.device, dtype=torch.float32)
    code = lib().fldr_ingest_u8(_dev(u8, "frames", torch.uint8), _dev(lv0, "level0"), B, H, W, Hp, Wp, _stream())
    _check(code, "fldr_ingest_u8")
    pyr = [lv0]
    for i in range(1, n_levels):
        f = 2 ** i
        lv = torch.empty(B, 3, 2, Hp // f, Wp // f, device=u8.device, dtype=torch.float32)
        _check(lib().fldr_pyramid_bicubic(_dev(lv0, "level0"), _dev(lv, "level"), B * 6, Hp, Wp, f, _stream()), "fldr_pyramid_bicubic")
        pyr.append(lv)
    return pyr


def frame_metrics(pred, H, W, target_u8=None, want_u8=False):
    """pred [B,3,Hp,Wp] (fp64/fp32) -> (sse [B] fp64 device tensor or None, uint8 image [B,3,H,W] or None)."""
    B, C, Hp, Wp = pred.shape
    assert C == 3 and pred.dtype in (torch.float64, torch.float32)
    pred = pred.contiguous()
    sse = torch.zeros(B, device=pred.device, dtype=torch.float64) if target_u8 is not None else None
    img = torch.empty(B, 3, H, W, device=pred.device, dtype=torch.uint8) if want_u8 else None
    if target_u8 is not None:
        target_u8 = target_u8.contiguous()
        assert target_u8.shape == (B, 3, H, W)
    _check(lib().fldr_frame_metrics(_dev(pred, "pred", pred.dtype), int(pred.dtype == torch.float64),
                                    _dev(target_u8, "target", torch.uint8) if target_u8 is not None else None,
                                    _dev(img, "image", torch.uint8) if want_u8 else None,
                                    _dev(sse, "sse", torch.float64) if sse is not None else None,
                                    B, H, W, Hp, Wp, _stream()), "fldr_frame_metrics")
    return sse, img


def ssim_y_u8(pred_u8, target_u8):
    """utils.ssim_bgr (utils.py:662-669) on the device: uint8 [B,3,H,W] images in cv2 channel order -> SSIM-Y per sample
    as a [B] fp64 DEVICE tensor (no synchronisation)."""
    B, C, H, W = pred_u8.shape
    assert C == 3 and target_u8.shape == pred_u8.shape
    pred_u8, target_u8 = pred_u8.contiguous(), target_u8.contiguous()
    n = lib().fldr_ssim_y_ws_doubles(B, H, W)
    if n < 0:
        raise FldrError("bad SSIM shape")
    ws = torch.empty(n, device=pred_u8.device, dtype=torch.float64)
    code = lib().fldr_ssim_y_u8(_dev(pred_u8, "pred", torch.uint8), _dev(target_u8, "target", torch.uint8),
                                _dev(ws, "ws", torch.float64), B, H, W, _stream())
    if code == -2:
        raise ValueError("win_size exceeds image extent (7x7 window)")
    _check(code, "fldr_ssim_y_u8")
    stats = ws.view(B, 2 * H * W + 4)[:, 2 * H * W:]
    return stats[:, 2] / float((H - 6) * (W - 6))
