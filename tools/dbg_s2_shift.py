"""Stride-2 encoder convolutions at the 4K shapes: tile-grid shift 0 / 15 / 31 of the persistent kernel (us per launch)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_hip as hip
dev = torch.device("cuda:0"); torch.manual_seed(0); L = hip.lib()
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (cin, cout, h, w) in [(26, 16, 2304, 3840), (16, 32, 1152, 1920), (26, 16, 2304, 4096)]:
    xs = [torch.rand(1, cin, h, w, device=dev) * 2 - 1 for _ in range(3)]      # rotate inputs: 3 x 0.9 GB > Infinity Cache
    wt = torch.randn(cout, cin, 4, 4, device=dev) / 20; b = torch.randn(cout, device=dev)
    row = []
    for sh in (0, 15, 31, 7):
        L.fldr_debug_s2_xshift(sh)
        k = [0]
        def f():
            k[0] += 1
            hip.conv2d([xs[k[0] % 3]], wt, b, stride=2, relu=True, precision="split", want_f32=True, want_spk=True)
        row.append("shift %2d: %.1f us" % (sh, timeit(f)))
    print("%d->%d @%dx%d  " % (cin, cout, h, w) + " | ".join(row), flush=True)
L.fldr_debug_s2_xshift(-1)
