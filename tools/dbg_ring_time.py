"""Timing of fldr_conv2d_spk (ring pipeline, 8 and 4 consumers; barrier pipeline) on the layer shapes of a 4K forward; LIB=<path> selects an experimental library."""
import os, sys, torch
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..", "fldr-vfi_amd"))
import fldr_hip as hip
if os.environ.get("LIB"): hip.LIB_PATH = os.environ["LIB"]
dev = torch.device("cuda:0"); torch.manual_seed(0); L = hip.lib()
def timeit(fn, n=40):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
row = []
for (cin, cout, h, w) in [(96, 96, 288, 480), (96, 96, 144, 240), (96, 48, 288, 480), (48, 48, 288, 480), (48, 16, 1152, 1920), (96, 32, 576, 960), (64, 64, 288, 480)]:
    x = torch.rand(1, cin, h, w, device=dev); xp = hip.spk_pack(x); w2 = torch.randn(cout, cin, 3, 3, device=dev) / 30
    t = []
    for var in (0, 1, 2):
        L.fldr_debug_spk_variant(min(var, 1)); L.fldr_debug_ring_consumers(8 if var == 1 else 4)
        t.append(timeit(lambda: hip.conv2d_spk([xp], w2, None, relu=True, want_f32=False, want_spk=True)))
    row.append("%d->%d@%dx%d %.1f/%.1f/%.1f" % (cin, cout, h, w, t[0], t[1], t[2]))
print(os.environ.get("LIB", "product").split("/")[-1], "(barrier/ring8/ring4 us):", " | ".join(row), " timeouts", L.fldr_debug_ring_timeouts(), flush=True)
