import os, sys, math, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_hip as hip
dev = torch.device("cuda:0")
def timeit(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
wt = torch.randn(96, 96, 3, 3, device=dev) / 30; b = torch.randn(96, device=dev)
w = 480
for k in (1, 4, 8, 9, 12, 17, 18, 25, 26, 34, 51, 72):
    h = 4 * k
    x = torch.rand(1, 96, h, w, device=dev); out = torch.empty(1, 96, h, w, device=dev)
    us = timeit(lambda: hip.conv2d([x], wt, b, relu=True, out=out))
    blocks = 15 * k * 2
    print("rows %4d blocks %5d (%.2f/CU) %8.1f us   %6.1f TF/s" % (h, blocks, blocks / 256, us, 2 * 96 * 96 * 9 * h * w / us / 1e6))
