"""Stride-2 encoders: persistent kernel (fldr_debug_s2_persistent(1)) against the per-tile kernel (0): bit-identity and us."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_hip as hip
dev = torch.device("cuda:0")
torch.manual_seed(0)
def timeit(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (parts, cout, H, W, N) in [([3, 3, 2, 5], 16, 40, 72, 1), ([16], 32, 34, 70, 2), ([26], 16, 50, 38, 3), ([5], 16, 18, 66, 1),
                               ([3, 3, 3, 3, 2, 2, 2, 2, 3, 3], 16, 2304, 3840, 1), ([16], 32, 1152, 1920, 1), ([32], 64, 576, 960, 1)]:
    srcs = [torch.randn(N, c, H, W, device=dev) for c in parts]
    cin = sum(parts)
    wt = torch.randn(cout, cin, 4, 4, device=dev) / (cin * 16) ** 0.5
    b = torch.randn(cout, device=dev)
    run = lambda: hip.conv2d(srcs, wt, b, stride=2, relu=True, precision="split", want_spk=True)
    res = []
    for mode in (0, 1):
        hip.lib().fldr_debug_s2_persistent(mode)
        o, sp = run(); torch.cuda.synchronize()
        res.append((o.clone(), sp.buf.clone(), timeit(run)))
    print(parts, cout, (H, W), "N", N, "identical", torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]),
          "| per-tile %.1f us, persistent %.1f us" % (res[0][2], res[1][2]), flush=True)
hip.lib().fldr_debug_s2_persistent(1)
