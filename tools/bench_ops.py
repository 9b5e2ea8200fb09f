#!/usr/bin/env python3
"""Per-kernel micro-benchmark on the GPU box: every convolution shape of one 4K forward (and the other hot
kernels) timed in isolation with HIP events; prints achieved TFLOP/s / GB/s.
Usage: python tools/bench_ops.py [conv|mem|all]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_hip as hip  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


H, W = 2304, 3840
h, w = H // 8, W // 8
CONVS = [  # name, parts, cout, k, stride, (Hin, Win), up2
    ("rec/bottom/flow2.2 96->96 L0", [96], 96, 3, 1, (h, w), None),
    ("flow2.0 100->96 L0", [48, 48, 4], 96, 3, 1, (h, w), None),
    ("flow1 96->48 L0", [48, 48], 48, 3, 1, (h, w), None),
    ("flow2.6 48->48 L0", [48], 48, 3, 1, (h, w), None),
    ("flow2.8 48->4 L0", [48], 4, 3, 1, (h, w), None),
    ("96->96 L1", [96], 96, 3, 1, (h // 2, w // 2), None),
    ("96->96 L2", [96], 96, 3, 1, (h // 4, w // 4), None),
    ("96->96 L3", [96], 96, 3, 1, (h // 8, w // 8), None),
    ("enc1 26->16 s2", [3, 3, 3, 3, 2, 2, 2, 2, 3, 3], 16, 4, 2, (H, W), None),
    ("enc2 16->32 s2", [16], 32, 4, 2, (H // 2, W // 2), None),
    ("enc3 32->64 s2", [32], 64, 4, 2, (H // 4, W // 4), None),
    ("dec0 64->64", [64], 64, 3, 1, (H // 8, W // 8), None),
    ("dec1 up(64)+32->32", [64, 32], 32, 3, 1, (H // 4, W // 4), [True, False]),
    ("dec2 up(32)+16->16", [32, 16], 16, 3, 1, (H // 2, W // 2), [True, False]),
    ("dec3 up(16)->6", [16], 6, 3, 1, (H, W), [True]),
]


def bench_convs():
    tot = 0.0
    for name, parts, cout, k, stride, (Hi, Wi), up2 in CONVS:
        up2 = up2 or [False] * len(parts)
        srcs = [torch.rand(1, c, Hi // 2 if u else Hi, Wi // 2 if u else Wi, device=dev) * 2 - 1 for c, u in zip(parts, up2)]
        cin = sum(parts)
        wt = torch.randn(cout, cin, k, k, device=dev) / math.sqrt(cin * k * k)
        bs = torch.randn(cout, device=dev)
        Ho, Wo = (Hi, Wi) if k == 3 else (Hi // 2, Wi // 2)
        out = torch.empty(1, cout, Ho, Wo, device=dev)
        ms = timeit(lambda: hip.conv2d(srcs, wt, bs, stride=stride, relu=True, up2=up2, out=out))
        fl = 2.0 * cin * cout * k * k * Ho * Wo
        tot += ms
        print("%-30s %8.1f us  %7.2f TFLOP/s  (%5.1f%% of 157.3)" % (name, ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100))
    print("sum of listed convs: %.3f ms" % tot)


def bench_mem():
    I0 = torch.rand(1, 3, H, W, device=dev) * 2 - 1
    I1 = torch.rand(1, 3, H, W, device=dev) * 2 - 1
    lo = (torch.rand(1, 8, h, w, device=dev) - 0.5) * 4
    big = hip.resize_bilinear(lo, H, W, mul=8.0)
    f0, f1 = big[:, 0:2].contiguous(), big[:, 2:4].contiguous()
    t = torch.tensor([0.5], device=dev)
    z = hip.zmetric(I0, I1, f0, -1.89)
    scratch = torch.empty(4 * H * W, device=dev)
    outb = torch.empty(1, 3, H, W, device=dev)
    HW = H * W
    rows = [
        ("resize x8 (8 planes)", lambda: hip.resize_bilinear(lo, H, W, mul=8.0), (8 * HW) * 4),
        ("zmetric", lambda: hip.zmetric(I0, I1, f0, -1.89), (3 + 3 + 2 + 1) * HW * 4),
        ("softsplat image (C=3,+z)", lambda: hip.softsplat_fused(I0, f0, z, "softmax", out=outb, scratch=scratch), (3 + 2 + 1 + 3) * HW * 4),
        ("bwarp C=3", lambda: hip.bwarp(I0, f0), (3 + 2 + 3) * HW * 4),
        ("bwarp_tscaled C=2", lambda: hip.bwarp_tscaled(f0, f1, t, "t", "1-t"), (2 + 2 + 2) * HW * 4),
    ]
    feat = torch.rand(1, 48, h, w, device=dev)
    ff = (torch.rand(1, 2, h, w, device=dev) - 0.5) * 4
    rows.append(("softsplat feat L0 (C=48)", lambda: hip.softsplat_fused(feat, ff, None, "softmax"), (48 + 2 + 48) * h * w * 4))
    planes = torch.rand(6, H, W, device=dev)
    ev = torch.randn(16, 64, device=dev, dtype=torch.float64)
    mean = torch.randn(64, device=dev, dtype=torch.float64)
    mv = torch.rand(16, device=dev, dtype=torch.float64) + 0.5
    rows.append(("pca L0", lambda: hip.pca_project(planes, ev, mean, mv), (6 * HW + 96 * h * w) * 4))
    refine = torch.randn(1, 6, H, W, device=dev)
    rows.append(("synth_tail fp64 out", lambda: hip.synth_tail(refine, [I0, I1, I0, I1, I0, I1], t, 1.56), (6 + 18) * HW * 4 + 3 * HW * 8))
    for name, fn, byts in rows:
        ms = timeit(fn)
        print("%-30s %8.1f us  %7.1f GB/s algorithmic" % (name, ms * 1e3, byts / ms / 1e6))


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("conv", "all"):
        bench_convs()
    if what in ("mem", "all"):
        bench_mem()
