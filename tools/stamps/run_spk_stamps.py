import ctypes, os, sys, torch
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..", "..", "fldr-vfi_amd"))
import fldr_hip as hip
hip.LIB_PATH = os.path.join(R, "libfldr_sstamp.so")
dev = torch.device("cuda:0")
wt = torch.randn(96, 96, 3, 3, device=dev) / 30; b = torch.randn(96, device=dev)
for (h, w) in [(36, 60), (288, 480), (576, 960)]:
    x = torch.rand(1, 96, h, w, device=dev); xp = hip.spk_pack(x)
    for _ in range(3): hip.conv2d_spk([xp], wt, b, relu=True, want_f32=False, want_spk=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); hip.conv2d_spk([xp], wt, b, relu=True, want_f32=False, want_spk=True); e1.record(); torch.cuda.synchronize()
    buf = (ctypes.c_uint64 * 32)()
    hip.lib().fldr_debug_read_spk_stamps.argtypes = [ctypes.c_void_p]
    hip.lib().fldr_debug_read_spk_stamps(buf)
    print((h, w), "launch %.1f us" % (e0.elapsed_time(e1) * 1e3))
    for k, name in enumerate(("wg0 wave0", "wg0 wave4", "wg101 wave0", "wg101 wave4")):
        v = buf[k * 8: k * 8 + 7]; n = max(1, v[5])
        print("  %-12s iters %3d | per iteration cycles: top %5d steps %5d finish %5d vmcnt-wait %5d barrier %5d | loop total %d cycles"
              % (name, v[5], v[0] / n, v[1] / n, v[2] / n, v[3] / n, v[4] / n, v[6]))
