import os, sys, torch
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..", "..", "fldr-vfi_amd"))
import fldr_hip as hip
which = sys.argv[1]
if which != "full":
    hip.LIB_PATH = os.path.join(R, "libfldr_%s.so" % which)
dev = torch.device("cuda:0")
def timeit(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
wt = torch.randn(96, 96, 3, 3, device=dev) / 30; b = torch.randn(96, device=dev)
for (h, w) in [(36, 60), (288, 480), (576, 960)]:
    x = torch.rand(1, 96, h, w, device=dev); xp = hip.spk_pack(x)
    print(which, (h, w), "%.1f us" % timeit(lambda: hip.conv2d_spk([xp], wt, b, relu=True, want_f32=False, want_spk=True)))
