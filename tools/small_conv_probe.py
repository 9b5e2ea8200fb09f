"""The ring convolution on the maps of the coarse pyramid levels (the ~10 us launches of a forward): time per launch, default dispatch."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import torch
import fldr_hip as hip
dev = torch.device("cuda:0")
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (cin, cout) in ((100, 96), (96, 64), (64, 32), (32, 2)):
    row = []
    for (h, w) in ((9, 15), (18, 30), (36, 60), (72, 120)):
        xp = hip.spk_pack(torch.rand(2, cin, h, w, device=dev)); wt = torch.randn(cout, cin, 3, 3, device=dev) / 30; b = torch.randn(cout, device=dev)
        g = torch.cuda.CUDAGraph()
        fn = lambda: hip.conv2d_spk([xp], wt, b, relu=True, want_f32=cout < 8, want_spk=cout >= 8)
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(10): fn()
        row.append("%dx%d %.2f" % (h, w, timeit(g.replay, 30) / 10))
    print("%3d->%2d (N=2), us per launch in a 10-launch graph: " % (cin, cout) + " | ".join(row), flush=True)
