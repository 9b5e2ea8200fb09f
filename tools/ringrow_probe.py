#!/usr/bin/env python3
"""Round-6 experiment: the 3x3 ring convolution with ring items of (32 channels x one kernel row) — 9 K = 32 steps per 32 channels, no
zero pad tap (conv3x3_ringrow_kernel, test build) — against the product kernel (16-channel chunks, 5 steps each) at the layer shapes
of a 4K forward it covers: values (fp32 accumulation rounding apart) and time on rotating inputs."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_hip as hip
dev = torch.device("cuda:0")
L = hip.enter_test_hooks()
L.fldr_debug_ringrow_pack_floats.restype = ctypes.c_int64
L.fldr_debug_ringrow_pack_floats.argtypes = [ctypes.c_int, ctypes.c_int]
L.fldr_debug_ringrow_prepack.restype = ctypes.c_int
L.fldr_debug_ringrow_prepack.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 2 + [ctypes.c_void_p]
L.fldr_debug_conv2d_ringrow.restype = ctypes.c_int
L.fldr_debug_conv2d_ringrow.argtypes = [ctypes.POINTER(hip.SpkConvDesc), ctypes.c_void_p, ctypes.c_void_p]
torch.manual_seed(0)


def row_conv(srcs, weight, bias, relu, up2, wrow):
    cout, cin = weight.shape[:2]
    N = srcs[0].shape[0]
    H = srcs[0].shape[2] * (2 if up2[0] else 1)
    W = srcs[0].shape[3] * (2 if up2[0] else 1)
    d = hip.SpkConvDesc()
    for i, (s, u) in enumerate(zip(srcs, up2)):
        d.src[i] = s.ptr; d.src_bstride[i] = s.bstride if N > 1 else 0; d.src_c[i] = s.shape[1]; d.src_up2[i] = int(u)
    d.n_src = len(srcs)
    d.wpack = hip.conv_spk_prepack(weight).data_ptr()
    d.bias = bias.data_ptr()
    out = hip._spk_alloc(N, cout, H, W, dev)
    d.out_spk = out.buf.data_ptr()
    d.N, d.cin, d.cout, d.cout_store, d.H, d.W, d.relu, d.precision = N, cin, cout, cout, H, W, int(relu), 0
    rc = L.fldr_debug_conv2d_ringrow(ctypes.byref(d), wrow.data_ptr(), hip._stream())
    assert rc == 0, rc
    return out


def timeit(fn, n=24):
    for i in range(4): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


CASES = [  # name, source channel lists (with up2 flags), cout, N, H, W, relu
    ("96->96 @288x480", [(96, False)], 96, 1, 288, 480, True),
    ("96->48 @288x480 (conv_flow2.4)", [(96, False)], 48, 1, 288, 480, True),
    ("48+48->48 x2 @288x480 (conv_flow1 pair)", [(48, False), (48, False)], 48, 2, 288, 480, False),
    ("64->64 @288x480 (dec0)", [(64, False)], 64, 1, 288, 480, True),
    ("64up2+32->32 @576x960 (dec1)", [(64, True), (32, False)], 32, 1, 576, 960, True),
    ("96->96 @144x240", [(96, False)], 96, 1, 144, 240, True),
    ("96->96 @100x203 (ragged)", [(96, False)], 96, 1, 100, 203, True),
]
for name, parts, cout, N, H, W, relu in CASES:
    cin = sum(c for c, _ in parts)
    up2 = [u for _, u in parts]
    wt = (torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5).contiguous()
    bs = torch.randn(cout, device=dev) * 0.1
    nf = L.fldr_debug_ringrow_pack_floats(cout, cin)
    assert nf > 0, (name, nf)
    wrow = torch.empty(nf, device=dev, dtype=torch.float32)
    rc = L.fldr_debug_ringrow_prepack(wt.data_ptr(), hip.conv_spk_prepack(wt).data_ptr(), wrow.data_ptr(), cout, cin, hip._stream())
    assert rc == 0
    sets = []
    for k in range(4):
        sets.append([hip.spk_pack(torch.rand(N, c, H // 2 if u else H, W // 2 if u else W, device=dev) * 2 - 1) for c, u in parts])
    ref = hip.conv2d_spk(sets[0], wt, bs, relu=relu, up2=up2, want_f32=False, want_spk=True).float()
    got = row_conv(sets[0], wt, bs, relu, up2, wrow).float()
    err = (ref - got).abs().max().item()
    t_ref = timeit(lambda i: hip.conv2d_spk(sets[i % 4], wt, bs, relu=relu, up2=up2, want_f32=False, want_spk=True))
    t_row = timeit(lambda i: row_conv(sets[i % 4], wt, bs, relu, up2, wrow))
    t_ref2 = timeit(lambda i: hip.conv2d_spk(sets[i % 4], wt, bs, relu=relu, up2=up2, want_f32=False, want_spk=True))
    t_row2 = timeit(lambda i: row_conv(sets[i % 4], wt, bs, relu, up2, wrow))
    print("%-44s product %.1f / %.1f us   row items %.1f / %.1f us   max |diff| %.2e (|ref| <= %.1f)" % (name, t_ref, t_ref2, t_row, t_row2, err, ref.abs().max().item()), flush=True)
hip.check_range()
