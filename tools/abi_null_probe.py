#!/usr/bin/env python3
"""Every compute entry point of include/fldr_hip.h called with (a) null pointers and zero sizes and (b) a zero-initialised descriptor where it
takes one: each must come back with a negative FLDR_E_* code — no launch, no crash.  Prints one JSON object {name: [code_a, code_b]}.
Run in a process of its own (tests/test_gpu_parity.py::test_every_entry_point_rejects_null_arguments starts it through the GPU-clean
launcher): a missing argument check shows up as a segmentation fault of THIS process, not of the test session."""
import ctypes
import json
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_hip as hip

SKIP = {"fldr_version", "fldr_error_string", "fldr_range_status", "fldr_ring_status"}      # no arguments to get wrong / plain status reads
lib = ctypes.CDLL(hip.LIB_PATH)
out = {}
for name, (res, args) in sorted(hip._SIGNATURES.items()):
    if name.startswith("fldr_debug_") or name in SKIP or not hasattr(lib, name):
        continue
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = res, args

    def zero(t, with_desc):
        if isinstance(t, type) and issubclass(t, ctypes._Pointer):
            target = t._type_
            if with_desc and isinstance(target, type) and issubclass(target, ctypes.Structure):
                return ctypes.pointer(target())                # zero-initialised descriptor
            return None
        if t is ctypes.c_void_p or t is ctypes.c_char_p:
            return None
        if t in (ctypes.c_float, ctypes.c_double):
            return 0.0
        return 0
    codes = []
    for with_desc in (False, True):
        r = fn(*[zero(t, with_desc) for t in args])
        codes.append(int(r) if r is not None else None)
    out[name] = codes
print(json.dumps(out))
